#!/bin/bash
# Regenerates the bench lines and rocprofv3 kernel statistics kept under profiles/ (run on the GPU box, from the repo root;
# results land in gpurun_out/ and are copied to profiles/ by hand).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
python bench.py > gpurun_out/r02_bench_default.json
python bench.py --height 720 --width 1280 --no-cpu-baseline > gpurun_out/r02_bench_720p.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -o p3 -- python3 bench.py --no-cpu-baseline --steps 10 > gpurun_out/r02_bench_default_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -o p1 -- python3 bench.py --no-cpu-baseline --steps 10 --streams 1 > gpurun_out/r02_bench_streams1_under_rocprof.json
find gpurun_out/prof3 gpurun_out/prof1 -name "*kernel_stats.csv" | xargs ls -la
find gpurun_out/prof3 gpurun_out/prof1 -name "*kernel_trace.csv" -delete
