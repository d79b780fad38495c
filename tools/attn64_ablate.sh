#!/bin/bash
# timing ablations of flash_fwd64_kernel (results are WRONG by construction; only the time of form 1 is read): one experiment
# library per EXP64_* macro, built by g2vlm_amd.build.build(extra_flags=[...], out=g2vlm_amd/lib/exp/lib64_<name>.so)
for v in ${VARIANTS:-base nodma noexp novalu nolds nobar nothing}; do
  if [ "$v" = base ]; then unset G2V_LIB_PATH; else export G2V_LIB_PATH=$PWD/g2vlm_amd/lib/exp/lib64_$v.so; fi
  echo -n "$v: "; python tools/attn_ab.py mot 2>&1 | grep "^mot" | sed 's/form1 vs.*//'
done
