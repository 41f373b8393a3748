#!/bin/bash
# Diagnostic (GPU box): the step of config 4 with 2 .. 4 plan calls in flight (bench.py --overlap N, capped at 4), one box.
cd "$(dirname "$0")/.."
for n in 2 3 4 3; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-latency --no-parity --steps 100 --warmup 10 --repeats 3 --overlap $n | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $n', round(d['ms_per_step'],4), d['repeats']['ms_per_step'])"; done
