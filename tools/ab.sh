#!/bin/bash
# Development aid: run the conv microbench on several library builds (vfi_amd/libvfi_<tag>.so) in ONE gpurun call,
# interleaved, so that device-to-device and clock differences cancel.  usage: tools/ab.sh "A B C" [reps]
P=$GRAFT_REPO_ROOT/fusion-method-for-video-frame-interpolation_amd/vfi_amd
for rep in $(seq 1 ${2:-2}); do
  for tag in $1; do
    echo "== $tag (rep $rep)"
    VFI_HIP_LIBRARY=$P/libvfi_$tag.so timeout -k 10 200 python $GRAFT_REPO_ROOT/tools/microbench.py --what conv --iters 20 2>&1 | grep -v amdgpu | grep -E "pn.l7|heads|conv1.2|conv3 |conv5|total"
  done
done
