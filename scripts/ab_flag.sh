#!/bin/bash
# Diagnostic (GPU box): the full step (four calls in flight) and the serial step of a build with extra flags against the
# default build, alternating, on one box.     scripts/ab_flag.sh "-DFOT_CULL_KG=8"
set -e
cd "$(dirname "$0")/.."
one() {
  make -C integrated_path_planning_amd/csrc clean > /dev/null
  make -C integrated_path_planning_amd/csrc EXTRA="$2" > /dev/null 2>&1
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --steps 100 --warmup 10 --repeats 3 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'headline', round(d['ms_per_step'],4), 'serial', round(d['serial']['ms_per_step'],4), d['serial']['kernel_ms'], 'parity', d['parity']['ok'])"
}
for r in 1 2; do
  one "[$1]" "$1"
  one "[default]" ""
done
