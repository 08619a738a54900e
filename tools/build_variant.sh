#!/bin/bash
# development aid: vfi_amd/libvfi_<tag>.so = the current objects with vfi_conv_winograd4m.hip recompiled with extra flags /
# generator environment (W4M_ELIM=...), e.g.  tools/build_variant.sh stamps -DW4M_STAMPS
set -e
tag=$1; shift
cd "$(dirname "$0")/../fusion-method-for-video-frame-interpolation_amd/csrc"
python ../../tools/gen_wino4m.py /tmp/w4m_body_$tag.h
mkdir -p /tmp/w4m_$tag && cp /tmp/w4m_body_$tag.h /tmp/w4m_$tag/vfi_conv_winograd4m_body.h
cp vfi_conv_winograd4m.hip /tmp/w4m_$tag/          # (so that the quoted include finds the variant's body header first)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-cuda-compat -I../../include -I. -fno-slp-vectorize "$@" -c /tmp/w4m_$tag/vfi_conv_winograd4m.hip -o /tmp/w4m_$tag/w4m.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../vfi_amd/libvfi_$tag.so $(ls *.o | grep -v vfi_conv_winograd4m.o) /tmp/w4m_$tag/w4m.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
echo built libvfi_$tag.so
