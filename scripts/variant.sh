#!/bin/bash
# Diagnostic (GPU box): rebuild libfot with extra compiler flags and time the serial bench workload.
#   scripts/variant.sh "-DFOT_EVAL_STATIC" [bench args]
set -e
cd "$(dirname "$0")/.."
FLAGS="$1"; shift || true
make -C integrated_path_planning_amd/csrc clean > /dev/null
make -C integrated_path_planning_amd/csrc EXTRA="$FLAGS" | grep -v hipcc
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --overlap 1 --steps 50 --warmup 5 "$@" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$FLAGS', d['serial']['ms_per_step'], d['serial']['kernel_ms'], d['parity'])"
