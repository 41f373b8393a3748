#!/bin/bash
# Diagnostic (GPU box): k_evaluate with 1 / 2 / 4 waves per workgroup on the serial bench workload
set -o pipefail
cd "$(dirname "$0")/.."
for w in 4 2 1 4 2 1; do
  FOT_EVAL_WPW=$w timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --no-parity --steps 100 --warmup 10 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpw $w', 'headline %.4f' % d['ms_per_step'], 'serial %.4f' % d['serial']['ms_per_step'], d['serial']['kernel_ms'])" || exit 1
done
