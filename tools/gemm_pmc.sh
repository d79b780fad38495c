#!/bin/bash
# rocprofv3 PMC pass (MFMA busy) of one GEMM shape in the eight-wave and the four-wave form (own process each, --kernel-trace only):
#   tools/gemm_pmc.sh <tag> [shapes...]   then the table: python3 tools/pmc_summary.py gpurun_out/pmc_<tag>_*
export TMPDIR=/tmp
tag=$1; shift
for c in ${@:-gu down o qkv sq8k}; do
  for f in staggered four_wave; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_${tag}_${c}_${f} -o p -- python3 tools/gemm_only.py $c $f 6 > /dev/null 2>&1
    echo "## $c $f"; python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_${c}_${f} | grep -i "gemm"
  done
done
