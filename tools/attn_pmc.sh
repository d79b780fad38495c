#!/bin/bash
# rocprofv3 PMC passes of the MoT prefill attention (one counter group per process, --kernel-trace only; never with sys/hip traces).
# Usage on the GPU box: tools/attn_pmc.sh <tag> [cases...] ; then python3 tools/pmc_summary.py gpurun_out/pmc_<tag>_<case>_* > profiles/...
export TMPDIR=/tmp
tag=$1; shift
for c in ${@:-mot c4rank}; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_${c}_fetch -o p -- python3 tools/attn_only.py $c 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_${c}_write -o p -- python3 tools/attn_only.py $c 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_${tag}_${c}_mfma -o p -- python3 tools/attn_only.py $c 6 > /dev/null 2>&1
  echo "## $c"; python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_${c}_fetch gpurun_out/pmc_${tag}_${c}_write gpurun_out/pmc_${tag}_${c}_mfma
done
