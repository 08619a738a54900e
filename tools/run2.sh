# development aid: the conv microbench on several library builds (vfi_amd/libvfi_<tag>.so) in one GPU call
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/fusion-method-for-video-frame-interpolation_amd/vfi_amd
for rep in 1 2; do for tag in $TAGS; do echo "== $tag rep $rep"; VFI_HIP_LIBRARY=$P/libvfi_$tag.so timeout -k 10 200 python tools/microbench.py --what conv --iters 20 --filter "${FILTER:-pn.l7b}" 2>&1 | grep -v amdgpu | grep -E "pn.l7|heads|conv1.2|conv2|head7"; done; done 2>&1 | tee gpurun_out/conv_elim.txt
