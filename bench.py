#!/usr/bin/env python3
"""Headline benchmark: interpolated frames/s of the full fused per-frame path at 1920x1080.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = ONE interpolated frame per rank (the whole sequence of reference
src/fusion_net/interpolate_twoframe.py:148-330 with output_baseline, as src/evaluation always sets it:
4x AdaCoF (three of them as one batch), 2 pyramid analyses + 2 syntheses + 2 radial-filter shortcuts (the reference's
4 analyses and 5 syntheses, see vfi_pyr_apply_filter / _pair), PhaseNet, uncertainty maps incl. the 50x50 median,
FusionNet) on synthetic frame pairs already resident in HBM.  Frames of a clip shard across ranks with no data-path
collective (weak scaling: fixed work per GPU); weights are broadcast once from rank 0 over RCCL.
Rank 0 prints ONE JSON line.

`python bench.py --gpus N` without a torchrun environment spawns its N ranks itself (a child
`python -m torch.distributed.run ... bench.py`, started before this process has made any GPU call) and exits with
the child's code.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def synthetic_pairs(n_pairs, h, w, device, seed=0):
    """Band-limited translating textures (oracle.synth is test infrastructure; this is the product-side
    generator of the same kind of content, built on the device)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    yy, xx = yy.to(device), xx.to(device)
    pairs = []
    for p in range(n_pairs):
        waves = [(float(torch.rand(1, generator=g)) * 0.23 + 0.02, float(torch.rand(1, generator=g)) * np.pi,
                  float(torch.rand(1, generator=g)) * 2 * np.pi, torch.rand(3, generator=g) * 0.7 + 0.3) for _ in range(6)]

        def render(dy, dx):
            img = torch.zeros((3, h, w), device=device)
            for freq, ang, ph, col in waves:
                arg = freq * np.pi * (np.cos(ang) * (xx - dx) + np.sin(ang) * (yy - dy)) + ph
                img += col.to(device).view(3, 1, 1) * torch.sin(arg)
            img = img / 12.0 * 1.2 + 0.5 + 0.15 * ((xx - dx) / w - 0.5) + 0.1 * ((yy - dy) / h - 0.5)
            return img.clamp_(0, 1).contiguous()
        pairs.append((render(0.0, 0.0), render(3.5, -2.25)))
    return pairs


def build_runner(device, n_streams=1, seed=0):
    import types
    from vfi_amd.adacof.models import Model
    from vfi_amd.fusion_net.fusion_net import FusionNet
    from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator
    from vfi_amd.phase_net.phase_net import PhaseNet
    from vfi_amd import shard
    torch.manual_seed(seed)      # random-init weights of the real architectures (no checkpoints offline)
    adacof = Model(types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1,
                                         gpu_id=device.index or 0))
    adacof.eval()
    fusion = FusionNet().to(device)
    fusion.eval()
    proto = PhaseNet(types.SimpleNamespace(height=17, nbands=4), device)
    n = shard.broadcast_module_states([adacof, fusion, proto], src=0)     # RCCL broadcast over xGMI (one buffer)
    state = proto.state_dict()
    # one interpolator (own pyramid plan / workspace / PhaseNet state) per in-flight frame; weights are shared
    return [FusionInterpolator(adacof, fusion, state, device) for _ in range(n_streams)], n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Cores this process can actually run on at once: the affinity mask, cut down to the cgroup's CPU quota where one is
    set (the GPU box gives one GPU's job a 16-core share of a much larger host; threads beyond the share only fight
    over it -- a 1080p oracle frame then takes more than 7 minutes instead of ~70 s).  -> (cores, how they were found)"""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = f"sched_getaffinity: {avail}"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # cgroup v1
                q, p = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        how += f", cgroup cpu quota: {quota:g}"
        avail = max(1, min(avail, int(quota + 0.5)))
    return avail, how


def cpu_baseline(hw=(1080, 1920), runs=3, warmup_hw=(256, 256), threads=None):
    """The oracle pipeline (CPU restatement of the same path, oracle/pipeline_cpu.py) on the benchmark workload ITSELF: one
    fused frame with output_baseline at the benchmark size after a small warm-up frame (thread pools, first-touch);
    frames/s = 1 / median over `runs` timed runs -- BASELINE.md section 3's "1 warm-up + 3 runs, median" protocol by
    default (~70 s per run on the GPU box's host share, so the CPU leg takes ~4 minutes).  Threads = every core this
    process can use (usable_cores).  A line per run goes to stderr so a long CPU leg never looks hung."""
    from oracle import pipeline_cpu, synth
    avail, how = usable_cores()
    if threads:
        how += f", --cpu-threads {threads}"
    threads = max(1, min(threads or avail, avail))
    torch.set_num_threads(threads)
    print(f"[bench] cpu_baseline: {threads} threads ({how})", file=sys.stderr, flush=True)
    weights = pipeline_cpu.seeded_weights(0)
    w0, _, w2 = (torch.from_numpy(x) for x in synth.translating_pair(1, *warmup_hw))
    pipeline_cpu.interp(w0, w2, weights, output_baseline=True)
    h, w = hw
    f0, _, f2 = (torch.from_numpy(x) for x in synth.translating_pair(0, h, w))
    times, stages = [], []
    for i in range(runs):
        st = {}
        t0 = time.perf_counter()
        pipeline_cpu.interp(f0, f2, weights, output_baseline=True, timings=st)
        times.append(time.perf_counter() - t0)
        stages.append(st)
        print(f"[bench] cpu_baseline run {i + 1}/{runs}: {times[-1]:.1f} s", file=sys.stderr, flush=True)
    med = float(np.median(times))
    st = stages[int(np.argsort(times)[len(times) // 2])]
    return {"value": 1.0 / med, "unit": "frames/s", "cores": threads, "cores_note": f"threads used: {threads} ({how})",
            "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{runs} timed fused frame(s) at {w}x{h} (the benchmark workload itself) after a {warmup_hw[1]}x{warmup_hw[0]} "
                      f"warm-up frame; median {med:.1f} s of {[round(t, 1) for t in times]} on {threads} threads",
            "stage_seconds": {k: round(v, 3) for k, v in st.items()}}


def spawn_ranks(n):
    """Parent of a bare `python bench.py --gpus N`: starts the N ranks through torch.distributed.run and relays their
    exit code.  Nothing in this process has touched HIP (device_count() does not initialise the runtime)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not (have >= 1 and os.environ.get("VFI_BENCH_SHARE_GPUS") == "1"):     # (rehearsal: ranks share the GPUs)
        sys.exit(f"bench.py: --gpus {n} but this node shows {have} GPU(s)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # per-rank stderr logs (stderr still streams to this console, prefixed with its rank; stdout -- rank 0's ONE JSON line --
    # is left alone): when a rank fails, the launcher's summary names it, and the stderr of the rank that failed FIRST is
    # repeated below it
    import tempfile
    with tempfile.TemporaryDirectory(prefix="vfi_bench_logs_") as logs:
        cmd[3:3] = ["--log-dir", logs, "--tee", "2"]
        rc = subprocess.call(cmd, env=env)
        if rc != 0:
            rank, tail = first_failing_rank_log(logs)
            if rank is not None:
                print(f"[bench] exit code {rc}; stderr of the first failing rank ({rank}):\n{tail}", file=sys.stderr, flush=True)
    sys.exit(rc)


def first_failing_rank_log(log_dir, lines=40):
    """(rank, last lines of its stderr) of the rank whose stderr log shows a failure and stopped growing first, from a
    torch.distributed.run --log-dir tree (<run id>/attempt_N/<rank>/stderr.log); (None, "") when there is none."""
    found = []
    for root, _dirs, files in os.walk(log_dir):
        if "stderr.log" in files and os.path.basename(root).isdigit():
            path = os.path.join(root, "stderr.log")
            try:
                with open(path, errors="replace") as f:
                    text = f.read()
            except OSError:
                continue
            if "[vfi shard]" in text or "Traceback" in text or "Error" in text:
                found.append((os.path.getmtime(path), int(os.path.basename(root)), text))
    if not found:
        return None, ""
    _, rank, text = min(found)
    return rank, "\n".join(text.rstrip().splitlines()[-lines:])


TRAFFIC_FILE = os.path.join("profiles", "r04_traffic.json")
KERNEL_SOURCES = {"conv3x3_winograd4m_kernel": "vfi_conv_winograd4m.hip", "conv3x3_winograd4_kernel": "vfi_conv_winograd4.hip",
                  "conv3x3_winograd_kernel": "vfi_conv_winograd.hip"}


def kernel_source_hash(kernel):
    """sha256[:16] of the source file a kernel lives in: the PMC passes behind `roofline.traffic` cannot run inside this
    benchmark (they need separate rocprofv3 --pmc runs), so the committed figures carry the hash of the source they were
    taken on and are dropped when the tree has moved on."""
    src = KERNEL_SOURCES.get(kernel)
    if not src:
        return None
    try:
        with open(os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "csrc", src), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE
    runs of this benchmark, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py writes the file) -- only
    when they were taken on the kernel source of this tree."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            rows = json.load(f)
    except (OSError, ValueError):
        return None, {"source": TRAFFIC_FILE, "dropped": "file missing"}
    row = rows.get(kernel)
    if not row:
        return None, {"source": TRAFFIC_FILE, "dropped": "no entry for this kernel"}
    have = kernel_source_hash(kernel)
    if row.get("kernel_source_sha16") != have:
        return None, {"source": TRAFFIC_FILE, "dropped": f"counters were taken on source {row.get('kernel_source_sha16')}, the tree has {have}"}
    return row["bytes_per_launch"], {"source": TRAFFIC_FILE, **{k: v for k, v in row.items() if k != "bytes_per_launch"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--streams", type=int, default=3,
                    help="frames in flight per GPU (independent frames of the clip on separate HIP streams)")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: capture each in-flight frame's ~700 launches into a hipGraph (torch.cuda.CUDAGraph) and replay it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU oracle leg (default: every core the process can use -- affinity mask cut to the cgroup quota)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions of --steps steps, back to back with the same barriers; `value` is the FIRST, all of them go to value_repeats")
    ap.add_argument("--cpu-baseline-runs", type=int, default=3,
                    help="timed full-size runs of the CPU oracle (median reported); 3 = BASELINE.md's protocol, ~4 minutes")
    ap.add_argument("--steps-720p", type=int, default=20,
                    help="extra timed steps at 1280x720 after the headline run (north_star asks for both sizes); 0 = skip")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                               # never returns

    from vfi_amd import _lib, shard
    try:
        rank, local_rank, world = shard.init_distributed()
    except shard.ShardError:
        sys.exit(3)                                          # (rank and backend error are on stderr already)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    local_dev = local_rank % torch.cuda.device_count()     # (== local_rank on a real node; lets 2 ranks rehearse on 1 GPU)
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    h, w = args.height, args.width

    try:
        runners, n_weights = build_runner(device, args.streams)        # (RCCL's first collective: the weight broadcast)
    except shard.ShardError:
        sys.exit(4)
    streams = [torch.cuda.Stream(device=device) for _ in runners]
    pairs = synthetic_pairs(4, h, w, device, seed=rank)
    torch.cuda.synchronize()

    def step(i):
        # frames of a clip are independent: frame i runs on stream i % S, so one frame's small kernels
        # (coarse pyramid levels, deep U-Net levels) overlap with another frame's large ones
        f0, f2 = pairs[i % len(pairs)]
        k = i % len(runners)
        with torch.cuda.stream(streams[k]):
            return runners[k](f0, f2, output_baseline=True)["final"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not steps: one untimed frame per in-flight slot creates that slot's pyramid plans / tables and (once) the
    # packed weights, whatever --warmup says
    for k in range(len(runners)):
        step(k)
    barrier()
    for i in range(args.warmup):
        step(i)
    barrier()
    if args.graph:
        # every library call only enqueues on the given stream (no allocation / synchronisation), so a whole frame can
        # be captured once per in-flight slot and replayed; inputs are copied into the captured buffers
        slots = []
        for k, r in enumerate(runners):
            f0s, f2s = torch.empty_like(pairs[0][0]), torch.empty_like(pairs[0][1])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(streams[k]):
                f0s.copy_(pairs[0][0]); f2s.copy_(pairs[0][1])
                streams[k].synchronize()
                with torch.cuda.graph(g, stream=streams[k]):
                    out = r(f0s, f2s, output_baseline=True)["final"]
            slots.append((g, f0s, f2s, out))
        eager_step = step

        def step(i):                                         # noqa: F811
            f0, f2 = pairs[i % len(pairs)]
            k = i % len(slots)
            g, f0s, f2s, out = slots[k]
            with torch.cuda.stream(streams[k]):
                f0s.copy_(f0, non_blocking=True); f2s.copy_(f2, non_blocking=True)
                g.replay()
            return out
        for i in range(2):
            step(i)
        barrier()
    def timed_region():
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        barrier()
        return shard.reduce_counters(args.steps, time.perf_counter() - t0, device)

    try:
        frames, elapsed = timed_region()
        # the same region again (same barriers, max over ranks each): how far `value` moves from run to run on this box --
        # a gain smaller than value_spread is not a measurement
        repeats = [frames / elapsed] + [(lambda fe: fe[0] / fe[1])(timed_region()) for _ in range(max(0, args.repeats - 1))]
    except shard.ShardError:
        sys.exit(5)

    # second size north_star names: the same fused frame at 1280x720, same barriers, same in-flight scheme (eager)
    value_720p = None
    if args.steps_720p > 0 and (h, w) != (720, 1280):
        pairs720 = synthetic_pairs(4, 720, 1280, device, seed=rank)

        def step720(i):
            f0, f2 = pairs720[i % len(pairs720)]
            k = i % len(runners)
            with torch.cuda.stream(streams[k]):
                return runners[k](f0, f2, output_baseline=True)["final"]
        for k in range(len(runners) + 2):      # plans / tables of the new size, then two warm-up frames
            step720(k)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps_720p):
            step720(i)
        barrier()
        f720, e720 = shard.reduce_counters(args.steps_720p, time.perf_counter() - t0, device)
        value_720p = f720 / e720

    line = None
    if rank == 0:
        print(f"[bench] timed region: {frames} frames in {elapsed:.2f} s", file=sys.stderr, flush=True)
        line = {"metric": "interpolated frames/sec at 1080p", "value": frames / elapsed, "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"full fused frame (PhaseNet + 4x AdaCoF + uncertainty maps + FusionNet, output_baseline) "
                                       f"at {w}x{h}, BASELINE.json configs[3]; one frame pair per rank per step",
                           "frame": [h, w], "weights": f"random-init, {n_weights} params broadcast from rank 0",
                           "sharding": f"frame pairs over {world} rank(s), no data-path collective",
                           "frames_in_flight_per_gpu": args.streams, "hip_graph": bool(args.graph)}}
        line["value_repeats"] = repeats
        line["value_spread"] = (max(repeats) - min(repeats)) / float(np.median(repeats))
        line["value_repeats_note"] = (f"{len(repeats)} timed regions of {args.steps} steps back to back, same barriers; `value` is the first; "
                                      "value_spread = (max - min) / median")
        if value_720p is not None:
            line["value_720p"] = value_720p
            line["value_720p_note"] = f"same fused frame at 1280x720, {args.steps_720p} timed steps after the headline run, frames/s of all ranks"
        if not args.no_profile:
            # Per-kernel algorithmic work / measured launch duration: HIP events on the launch stream around every library
            # call, over ONE frame running ALONE on the device on one stream after the timed region.  (With two frames in
            # flight the events of two streams overlap in time and would double-count, and a kernel's duration depends on
            # what the other stream happens to run beside it; `rocprofv3 --kernel-trace --stats -- python3 bench.py
            # --streams 1` reproduces these averages, profiles/.)  Sum over a frame of avg_launch_ms x launches stays
            # below the single-stream frame time by construction.
            prof_step = eager_step if args.graph else step

            def profile(n_frames):
                torch.cuda.synchronize()
                _lib.PROFILE = _lib.Recorder()
                for i in range(n_frames):
                    prof_step(i)
                agg = _lib.PROFILE.summary()
                _lib.PROFILE = None
                return agg

            saved, runners[:] = list(runners), runners[:1]
            profile(1)                                        # (first profiled frame also creates the event objects)
            t0 = time.perf_counter()
            agg = profile(1)
            alone_ms = (time.perf_counter() - t0) * 1e3
            runners[:] = saved
            convs = {k: v for k, v in agg.items() if v["kind"] == "flop"}
            dom = max(convs, key=lambda k: convs[k]["seconds"])
            d = convs[dom]
            tf = d["work"] / d["seconds"] / 1e12
            traffic, traffic_note = measured_traffic(dom)
            line["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "launches_per_frame": d["calls"],
                                "avg_launch_ms": d["seconds"] / d["calls"] * 1e3, "kernel_ms_per_frame": d["seconds"] * 1e3,
                                "frame_ms_alone_one_stream": alone_ms,
                                "share_of_kernel_time": d["seconds"] / sum(v["seconds"] for v in agg.values()),
                                "regime": "one frame alone on the device, one stream (HIP events on the launch stream)",
                                "algorithmic_gflop_per_launch": d["work"] / d["calls"] / 1e9}
            if traffic_note:
                line["roofline"]["traffic_note"] = traffic_note
            line["roofline"]["rocprof_kernel_name"] = dom
            if "winograd4" in dom:
                # `achieved` counts the Winograd algorithm's own multiply-adds (36 per 4x4 outputs and channel pair: what
                # the matrix cores execute); the same convolutions done directly are 144
                line["roofline"]["flops_counted"] = "Winograd F(4x4,3x3): 2*N*Cin*Cout*36*(H*W/16) per launch"
                line["roofline"]["direct_conv_equivalent_tflops"] = tf * 4.0
            elif "winograd" in dom:
                # (16 per 2x2 outputs and channel pair; directly: 36)
                line["roofline"]["flops_counted"] = "Winograd F(2x2,3x3): 2*N*Cin*Cout*16*(H*W/4) per launch"
                line["roofline"]["direct_conv_equivalent_tflops"] = tf * 2.25
            # north_star's pyramid metric by name: the two pyramid entry points against the HBM roofline on their
            # algorithmic bytes (72 * N * H * W bytes per call with all levels and both residuals)
            line["roofline_pyramid"] = {
                k: {"bound": "hbm", "achieved": v["work"] / v["seconds"] / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": v["work"] / v["seconds"] / 1e9 / PEAK_HBM_GBS, "calls_per_frame": v["calls"],
                    "ms_per_frame": v["seconds"] * 1e3, "algorithmic_mb_per_call": v["work"] / v["calls"] / 1e6}
                for k, v in agg.items() if k in ("pyr_analyze", "pyr_synthesize") and v["seconds"] > 0}
            line["roofline_other"] = []
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["seconds"]):
                if k == dom or not v["kind"]:
                    continue
                rate = v["work"] / v["seconds"]
                peak, unit, div = (PEAK_FP32_MFMA_TFLOPS, "TFLOP/s", 1e12) if v["kind"] == "flop" else (PEAK_HBM_GBS, "GB/s", 1e9)
                line["roofline_other"].append({"kernel": k, "bound": "mfma" if v["kind"] == "flop" else "hbm",
                                               "achieved": rate / div, "peak": peak, "unit": unit, "frac": rate / div / peak,
                                               "calls": v["calls"], "ms_per_frame": v["seconds"] * 1e3})
            line["stage_ms_isolated"] = {k: round(v["seconds"] * 1e3, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["seconds"])[:14]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline((h, w), runs=args.cpu_baseline_runs, threads=args.cpu_threads or None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
