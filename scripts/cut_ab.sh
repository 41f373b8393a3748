#!/bin/bash
# Diagnostic (GPU box): the per-wave cut (3 waves/SIMD) against the grouped cut (4 waves/SIMD) on the SAME box:
# throughput legs, 1-ego latencies and the batched closed loop.  FOT_TILE_CUT forces the cut (fot_setup.hpp).
set -o pipefail
cd "$(dirname "$0")/.."
for i in 1 2; do
for p in wave group; do
  export FOT_TILE_CUT=$p
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['latency']
print('cut $p', 'headline %.4f' % d['ms_per_step'], 'serial %.4f' % d['serial']['ms_per_step'], d['serial']['kernel_ms'], 'cfg2 %.4f cfg3 %.4f f2 %.4f f4 %.3f' % (l['config2']['p50_ms'], l['config3']['p50_ms'], l['f2_three_level_cycle']['one_launch_p50_ms'], l['f4_closed_loop']['ms_per_lock_step']), d['parity']['ok'])" || exit 1
done
done
