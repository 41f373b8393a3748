#!/bin/bash
# One-ego plan call on the device clock (GPU box): kernel durations and the gaps between them, next to the bare wall time.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/lat_trace
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 200 python3 scripts/latency_loop.py 1200 2>/dev/null | tee "$OUT/bare.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o run -- python3 scripts/latency_loop.py 400 > "$OUT/traced.txt" 2>&1
python3 scripts/latency_gaps.py "$OUT/trace" | tee "$OUT/gaps.json"
