#!/bin/bash
# same-box A/B of the C3 step by GEMM form per shape class (G2V_GEMM_4W_MASK, csrc/gemm_8p.hip: bit 0 wide N, bit 1 long K, bit 2 rest;
# 0 = eight-wave everywhere, 7 = four-wave everywhere), interleaved passes
for i in $(seq ${REPS:-2}); do
  for m in ${MASKS:-0 7 1 2 4 6 3 5}; do
    echo -n "mask $m: "; G2V_GEMM_4W_MASK=$m python3 bench.py --no-cpu-baseline --decode-tokens 0 --overlap 1 --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], 'ms/step', d['value'], 'views/s | gate/up', d['roofline_gemm']['launch_ms'], 'ms | attention', d['roofline']['launch_ms'], 'ms')"
  done
done
