#!/bin/bash
# Diagnostic (GPU box): k_cull with groups of $1 (default 8) time steps per workgroup against the build's default (4), one box.
set -e
KG=${1:-8}
cd "$(dirname "$0")/.."
one() {
  make -C integrated_path_planning_amd/csrc clean > /dev/null
  make -C integrated_path_planning_amd/csrc EXTRA="$2" > /dev/null 2>&1
  for r in 1 2; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --steps 100 --warmup 10 --repeats 3 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'headline', round(d['ms_per_step'],4), 'serial', round(d['serial']['ms_per_step'],4), d['serial']['kernel_ms'], 'parity', d['parity']['ok'])"
  done
}
one "KG$KG" "-DFOT_CULL_KG=$KG"
timeout -k 10 400 python -m pytest tests/test_gpu_limits.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1
one "default" ""
