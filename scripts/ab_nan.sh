#!/bin/bash
# Diagnostic (GPU box): where the NaN tracks are found -- k_cull itself (default) or k_frenet_state's scan blocks
# (FOT_NAN_SCAN=eager) -- on ONE box: serial step, kernel times, the step with three plan calls in flight.
#   scripts/ab_nan.sh ["-DFOT_CULL_UNROLL=8"]
set -e
cd "$(dirname "$0")/.."
FLAGS="$1"
if [ -n "$FLAGS" ]; then
  make -C integrated_path_planning_amd/csrc clean > /dev/null
  make -C integrated_path_planning_amd/csrc EXTRA="$FLAGS" | grep -v hipcc
fi
one() {
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --no-parity --steps 100 --warmup 10 --repeats 3 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $FLAGS', 'headline', round(d['ms_per_step'],4), 'serial', round(d['serial']['ms_per_step'],4), d['serial']['kernel_ms'])"
}
for r in 1 2; do
  FOT_NAN_SCAN=eager one eager
  one lazy
done
