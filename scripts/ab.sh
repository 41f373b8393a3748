#!/bin/bash
# Diagnostic (GPU box): A/B of two builds on the SAME box -- integrated_path_planning_amd/libfot_prev.so (A) against
# libfot.so (B), serial bench workload, alternating A B A B (boxes differ by a few percent, launches do not).
set -o pipefail
cd "$(dirname "$0")/.."
PK=integrated_path_planning_amd
cp $PK/libfot.so $PK/libfot_new.so
run() {
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --no-parity --steps 100 --warmup 10 "$@" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', 'headline %.4f' % d['ms_per_step'], 'serial %.4f' % d['serial']['ms_per_step'], d['serial']['kernel_ms'])"
}
for i in 1 2; do
  cp $PK/libfot_prev.so $PK/libfot.so; TAG=A run "$@" || exit 1
  cp $PK/libfot_new.so $PK/libfot.so; TAG=B run "$@" || exit 1
done
