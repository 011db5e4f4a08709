#!/bin/bash
# Timing-only variants of the f16x3 MLP bodies (ICON_EXP_* switches: WRONG results, each prices one component of the kernel
# by removing it).  The switches are NOT in the product sources: tools/probes/exp_r03/ holds the round-3 snapshot of
# mlp_f16x3_device.h / mlp_f16x3.hip that carries them; the variants are built from that copy next to the real objects.
# Builds icon_amd/exp/libicon_amd_<tag>.so next to the real library:
#   tools/exp_fused.sh build            (here: hipcc cross-compiles)
#   tools/exp_fused.sh run > out.txt    (on the GPU box: tools/mlp_power_probe.py under every variant, ICON_AMD_LIB)
set -e
cd "$(dirname "$0")/.."
C=icon_amd/csrc
VARIANTS="${ICON_EXP_VARIANTS:-base:: act2x:-DICON_EXP_ACT2X prio:-DICON_EXP_PRIO}"
if [ "$1" = build ]; then
  make -s -C $C
  mkdir -p icon_amd/exp
  for v in $VARIANTS; do
    tag=${v%%:*}; flags=$(echo "${v#*:}" | sed 's/:$//; s/__/ /g; s/^://')
    ( mkdir -p icon_amd/exp/src_$tag && cp $C/common.h icon_amd/exp/src_$tag/ && cp tools/probes/exp_r03/mlp_f16x3_device.h tools/probes/exp_r03/mlp_f16x3.hip icon_amd/exp/src_$tag/ &&
      sed -i 's#"../../include/icon_amd.h"#"../../../include/icon_amd.h"#' icon_amd/exp/src_$tag/common.h &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c icon_amd/exp/src_$tag/mlp_f16x3.hip -o icon_amd/exp/mlp_f16x3_$tag.o
      objs=$(ls $C/*.o | grep -v "mlp_f16x3.o")
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o icon_amd/exp/libicon_amd_$tag.so $objs icon_amd/exp/mlp_f16x3_$tag.o ) &
  done
  wait
  ls -la icon_amd/exp/*.so
else
  for v in $VARIANTS; do
    tag=${v%%:*}
    echo "== $tag"
    ICON_AMD_LIB=$PWD/icon_amd/exp/libicon_amd_$tag.so python tools/mlp_power_probe.py f16x3 ${2:-4}
  done
fi
