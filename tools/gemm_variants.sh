#!/bin/bash
# timing ablations of the 8-phase GEMM (EXP_GEMM_* in csrc/gemm_8p.hip): one experiment library per macro, built on the spot
# (g2vlm_amd.build.build(extra_flags=..., out=...)); results of the ablated builds are WRONG by construction.
mkdir -p g2vlm_amd/lib/exp
for v in ${VARIANTS:-base EXP_GEMM_NO_EPI EXP_GEMM_HALF_LDS EXP_GEMM_NO_MFMA}; do
  if [ "$v" = base ]; then unset G2V_LIB_PATH; else
    python -c "from g2vlm_amd import build; build.build(extra_flags=['-D$v'], out='g2vlm_amd/lib/exp/lib_$v.so')" || exit 1
    export G2V_LIB_PATH=$PWD/g2vlm_amd/lib/exp/lib_$v.so
  fi
  echo "== $v"; python tools/gemm_shapes.py ${M:-10968} 2>&1 | grep "^M"
done
