#!/bin/bash
# Diagnostic (GPU box): rebuild libfot with extra compiler flags and run both bench legs (serial + headline), parity on.
#   scripts/variant_full.sh "-DFOT_GROUP_TILES=2" [bench args]
set -e
cd "$(dirname "$0")/.."
FLAGS="$1"; shift || true
make -C integrated_path_planning_amd/csrc clean > /dev/null
make -C integrated_path_planning_amd/csrc EXTRA="$FLAGS" | grep -v hipcc | grep -v "^make" | cut -c1-150
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --steps 200 --warmup 20 "$@" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$FLAGS]', 'headline %.4f' % d['ms_per_step'], 'serial %.4f' % d['serial']['ms_per_step_without_event_pairs'], d['serial']['kernel_ms'], d['parity'])"
