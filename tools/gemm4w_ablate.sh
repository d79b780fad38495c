#!/bin/bash
# timing ablations of the four-wave GEMM (EXP_4W_* in csrc/gemm_4w.hip; results of the ablated builds are WRONG by construction).
#   tools/gemm4w_ablate.sh build    (CPU container: one experiment library per macro, only gemm_4w.hip is recompiled)
#   tools/gemm4w_ablate.sh          (GPU box: tools/gemm_yardstick.py with each library)
V="${VARIANTS:-EXP_4W_NO_DMA EXP_4W_NO_READS EXP_4W_NO_BARRIER EXP_4W_NO_EPI EXP_4W_DMA_NOP4}"
if [ "$1" = build ]; then
  mkdir -p g2vlm_amd/lib/exp
  for v in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -D$v -Iinclude -Ig2vlm_amd/csrc \
      -c g2vlm_amd/csrc/gemm_4w.hip -o g2vlm_amd/lib/exp/gemm_4w_$v.o || exit 1
    objs=$(ls g2vlm_amd/lib/obj/*.o | grep -v gemm_4w.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs g2vlm_amd/lib/exp/gemm_4w_$v.o -o g2vlm_amd/lib/exp/lib4w_$v.so || exit 1
    rm -f g2vlm_amd/lib/exp/gemm_4w_$v.o
  done
  exit 0
fi
echo "== base"; python3 tools/gemm_yardstick.py ${SHAPES:-} 2>&1 | grep "^M"
for v in $V; do
  echo "== $v"; G2V_LIB_PATH=$PWD/g2vlm_amd/lib/exp/lib4w_$v.so python3 tools/gemm_yardstick.py ${SHAPES:-} 2>&1 | grep "^M"
done
