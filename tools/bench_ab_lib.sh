#!/bin/bash
# same-box A/B of the C3 step between the shipped library and an experiment library (G2V_LIB_PATH), interleaved
#   tools/bench_ab_lib.sh g2vlm_amd/lib/exp/lib_x.so [reps]
for i in $(seq ${2:-3}); do
  for l in "" "$1"; do
    echo -n "${l:-shipped}: "; G2V_LIB_PATH=${l:+$PWD/$l} python3 bench.py --no-cpu-baseline --decode-tokens 0 --overlap 1 --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], 'ms/step', d['value'], 'views/s | gate/up', d['roofline_gemm']['launch_ms'], 'ms | attention', d['roofline']['launch_ms'], 'ms')"
  done
done
