#!/bin/bash
# Timing-only experiment builds of the fused kernel (results are WRONG by construction): which resource bounds it?
#   usage (build container): tools/exp_fused.sh build ; then on the GPU box: tools/exp_fused.sh run
cd "$(dirname "$0")/../icon_amd/csrc" || exit 1
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
OBJS="mesh_build.o query_kernels.o mlp_kernels.o mlp_f16x3.o mlp_mx6.o mcubes.o mc_device.o vis_kernels.o sort_points.o mesh_cc.o vox_kernels.o"
if [ "$1" = build ]; then
  for v in NODMA FASTACT "FASTACT -DICON_EXP_NODMA"; do
    tag=$(echo $v | tr -d ' ' | sed 's/-DICON_EXP_/_/g')
    $HIPCC $FLAGS -DICON_EXP_$v -c fused_f16x3.hip -o /tmp/fused_$tag.o && $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libicon_exp_$tag.so $OBJS /tmp/fused_$tag.o && echo built $tag
  done
else
  cd ../..
  for lib in "" icon_amd/libicon_exp_NODMA.so icon_amd/libicon_exp_FASTACT.so icon_amd/libicon_exp_FASTACT_NODMA.so; do
    ICON_AMD_LIB=$([ -n "$lib" ] && echo $PWD/$lib) python tools/time_fused.py 10 2>&1 | tail -1
  done
fi
