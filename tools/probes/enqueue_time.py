"""Probe: host time to enqueue one 1080p frame (eager) vs its GPU time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
import bench
dev = torch.device("cuda:0")
runners, _ = bench.build_runner(dev, 1)
f0, f2 = torch.rand(3, 1080, 1920, device=dev), torch.rand(3, 1080, 1920, device=dev)
for _ in range(3):
    runners[0](f0, f2, output_baseline=True)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    runners[0](f0, f2, output_baseline=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms")
