"""Probe: where a wave of conv3x3_winograd4m_kernel spends its cycles (library built with -DW4M_STAMPS as
vfi_amd/libvfi_stamps.so: s_memtime around the body statements and the item epilogues)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ["VFI_HIP_LIBRARY"] = os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "vfi_amd", "libvfi_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamps"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops, _lib
dev = torch.device("cuda:0")
cases = [("pn.l7b 64->64 x3 1080p", 3, 64, 64, 1080, 1920, "reflect", "elu"), ("head 28->25 1088x1920", 1, 25, 25, 1088, 1920, "zeros", None)]
if os.environ.get("W4M_CASES") == "strides":       # does the channel-plane stride (a multiple of 4 KiB at 1080p) matter to the stores?
    cases = [("64->64 x3 1080x1920 (plane = 2025 x 4 KiB)", 3, 64, 64, 1080, 1920, "zeros", "relu"),
             ("64->64 x3 1080x1984 (plane = 2092.5 x 4 KiB)", 3, 64, 64, 1080, 1984, "zeros", "relu"),
             ("64->64 x3 1081x1984 (plane = odd multiple of 64 B)", 3, 64, 64, 1081, 1984, "zeros", "relu")]
for name, n, cin, cout, h, w, pad, act in cases:
    x = torch.randn((n, cin, h, w), device=dev)
    pc = ops.PackedConv(torch.randn((cout, cin, 3, 3)) / (cin * 9) ** 0.5, torch.zeros(cout), device=dev)
    for _ in range(3):
        ops.conv2d(x, pc, pad, act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d(x, pc, pad, act); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 4 * 10))()
    assert _lib.lib().vfi_debug_w4m_stamps(buf, 256 * 4 * 10) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 10).astype(np.float64)
    body, between, epi, nb, ne, total = (a[..., k] for k in range(6))
    ms = e0.elapsed_time(e1)
    print(f"{name}: {ms:.3f} ms; per wave: total {total.mean():.0f} cycles -> clock {total.mean() / ms / 1e3:.0f} MHz (if the kernel fills the launch)")
    print(f"  bodies {nb.mean():.0f} x {body.sum() / nb.sum():.0f} cycles; between bodies {between.sum() / nb.sum():.0f} cycles per body; "
          f"epilogues {ne.mean():.1f} x {epi.sum() / ne.sum():.0f} cycles; unaccounted {(total - body - between - epi).mean():.0f}")
    print(f"  inside an epilogue: accumulator reads + first pass {a[..., 6].sum() / ne.sum():.0f} cycles, second pass + bias + activation + stores {a[..., 7].sum() / ne.sum():.0f}")
    tl, nl = a[..., 8].sum(), a[..., 9].sum()
    print(f"  bodies inside the looping statement: {nl / a[..., 3].sum():.2f} of all, {tl / max(nl, 1):.0f} cycles each; single-chunk statements {(body.sum() - tl) / max(nb.sum() - nl, 1):.0f} cycles each")
    for wv in range(4):
        print(f"  wave {wv}: body {body[:, wv].sum() / nb[:, wv].sum():.0f} between {between[:, wv].sum() / nb[:, wv].sum():.0f} epi {epi[:, wv].sum() / ne[:, wv].sum():.0f}")
