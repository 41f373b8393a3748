#!/bin/bash
# Diagnostic (GPU box): k_cull's gathers in flight (FOT_CULL_UNROLL) x track loads in flight (FOT_CULL_VU), lazy NaN check
set -e
cd "$(dirname "$0")/.."
one() {
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --no-parity --steps 100 --warmup 10 --repeats 3 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'headline', round(d['ms_per_step'],4), 'serial', round(d['serial']['ms_per_step'],4), d['serial']['kernel_ms'])"
}
for v in "4 4" "6 4" "10 4" "4 8" "4 2" "4 4"; do
  set -- $v
  make -C integrated_path_planning_amd/csrc clean > /dev/null
  make -C integrated_path_planning_amd/csrc EXTRA="-DFOT_CULL_UNROLL=$1 -DFOT_CULL_VU=$2" > /dev/null 2>&1
  one "lazy U$1 V$2"
  FOT_NAN_SCAN=eager one "eager U$1 V$2"
done
