# development aid: one GPU call = conv tests on the current build, then the stamps probe and the conv microbench (M = 32 / M = 16)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/conv_tests.txt 2>&1; rc=$?; echo rc=$rc; tail -5 gpurun_out/conv_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/probes/w4m_stamps.py stamps 2>&1 | grep -v amdgpu | tee gpurun_out/w4m_stamps.txt
for rep in 1 2; do for m in 1 0; do echo "== M32=$m rep $rep"; VFI_CONV_WINOGRAD4M=$m timeout -k 10 200 python tools/microbench.py --what conv --iters 20 2>&1 | grep -v amdgpu | grep -E "pn.l7|heads|conv1.2|conv2|head7|total"; done; done 2>&1 | tee gpurun_out/conv_ab.txt
