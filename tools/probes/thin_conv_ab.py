import os, sys, time, torch
sys.path[:0] = ["/root/repo", "/root/repo/fusion-method-for-video-frame-interpolation_amd"]
from vfi_amd import ops
dev = torch.device("cuda:0")
for (n, cin, cout, h, w) in [(1, 25, 25, 1088, 1920), (2, 25, 25, 1088, 1920), (1, 64, 25, 544, 960), (1, 32, 32, 1088, 1920), (1, 6, 32, 1088, 1920)]:
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    pc = ops.PackedConv(wt, torch.randn(cout, device=dev))
    x = torch.randn(n, cin, h, w, device=dev)
    for _ in range(3): ops.conv2d(x, pc, "zeros", "relu")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.conv2d(x, pc, "zeros", "relu")
    torch.cuda.synchronize(); print(os.environ.get("VFI_CONV_WINOGRAD", "1"), (n, cin, cout, h, w), round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms", flush=True)
