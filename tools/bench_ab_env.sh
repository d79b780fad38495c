#!/bin/bash
# same-box A/B of the C3 step between environment settings, interleaved:   tools/bench_ab_env.sh "G2V_ATTN_FORM=1" "G2V_ATTN_FORM=0" [reps]
for i in $(seq ${3:-3}); do
  for e in "$1" "$2"; do
    echo -n "$e: "; env $e python3 bench.py --no-cpu-baseline --decode-tokens 0 --overlap 1 --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], 'ms/step', d['value'], 'views/s | gate/up', d['roofline_gemm']['launch_ms'], 'ms | attention', d['roofline']['launch_ms'], 'ms')"
  done
done
