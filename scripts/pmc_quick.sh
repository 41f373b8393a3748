#!/bin/bash
# Quick SQ counter pass of the serial bench workload (GPU box): prints per-launch means for k_evaluate / k_cull.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_quick
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --pmc ${FOT_PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS} --output-format csv -d "$OUT/sq" -o run -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-latency --no-parity --overlap 1 > "$OUT/sq.log" 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for p in glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "k_evaluate" in k or "k_cull" in k:
            a = acc[k.split("(")[0][-20:]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in acc.items():
    print(k, {c: round(x[0] / x[1] / 1e6, 2) for c, x in v.items()})
PY
