#!/bin/bash
# timing experiments on the MoT attention kernel: one library per EXP_* macro (built by g2vlm_amd.build.build(extra_flags=..., out=...))
for v in ${VARIANTS:-base noexp nobar prio}; do
  if [ "$v" = base ]; then unset G2V_LIB_PATH; else export G2V_LIB_PATH=$PWD/g2vlm_amd/lib/exp/lib_$v.so; fi
  echo "== $v"; python tools/bench_kernels.py attn 2>&1 | grep "^mot"
done
