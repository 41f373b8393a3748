"""Device-side anatomy of the one-ego plan call from a rocprofv3 kernel trace of scripts/latency_loop.py: per call the
duration of the three kernels and the idle gaps between them (end of one kernel to start of the next on the device
clock).
   latency_gaps.py <dir with run_kernel_trace.csv>"""
import csv
import glob
import json
import os
import sys

import numpy as np

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])))
rows.sort()
short = lambda n: ("frenet" if "k_frenet_state" in n else "cull" if "k_cull" in n else "evaluate" if "k_evaluate" in n else None)
seq = [(s, e, short(n), g) for s, e, n, g in rows if short(n)]
calls = []
i = 0
while i + 2 < len(seq):
    a, b, c = seq[i], seq[i + 1], seq[i + 2]
    if (a[2], b[2], c[2]) == ("frenet", "cull", "evaluate"):
        calls.append((a[1] - a[0], b[0] - a[1], b[1] - b[0], c[0] - b[1], c[1] - c[0], c[1] - a[0], c[3]))
        i += 3
    else:
        i += 1
calls = np.array(calls, dtype=float).reshape(-1, 7) / 1e3       # us (grid size column aside)
out = {}
# latency_loop.py runs config 2, then config 3, the same number of calls each: first and second half of the trace
for name, sel in (("config2", calls[: len(calls) // 2]), ("config3", calls[len(calls) // 2:])):
    sel = sel[30:]                                              # (the first calls of a handle: warm-up)
    if len(sel) < 20:
        continue
    med = np.median(sel, axis=0)
    out[name] = {"calls": int(len(sel)), "k_frenet_state_us": med[0], "gap_1_us": med[1], "k_cull_us": med[2],
                 "gap_2_us": med[3], "k_evaluate_us": med[4], "first_start_to_last_end_us": med[5],
                 "kernels_sum_us": med[0] + med[2] + med[4]}
print(json.dumps(out, indent=1))
