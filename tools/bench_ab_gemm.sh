#!/bin/bash
# same-box A/B of the C3 step by GEMM form per shape class (G2V_GEMM_4W_MASK, csrc/gemm_8p.hip: bit 0 wide N (gate/up), bit 1 long K
# (down, fc2), bit 2 other fp32-residual Linears (o-proj), bit 3 plain bf16 outputs (qkv), bit 4 GELU (fc1); 0 = eight-wave everywhere,
# 31 = four-wave everywhere, shipped default 6), interleaved passes:   MASKS="0 6 31" REPS=3 tools/bench_ab_gemm.sh
for i in $(seq ${REPS:-2}); do
  for m in ${MASKS:-0 6 31 1 2 4 8 16}; do
    echo -n "mask $m: "; G2V_GEMM_4W_MASK=$m python3 bench.py --no-cpu-baseline --decode-tokens 0 --overlap 1 --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], 'ms/step', d['value'], 'views/s | gate/up', d['roofline_gemm']['launch_ms'], 'ms | attention', d['roofline']['launch_ms'], 'ms')"
  done
done
