#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/profile.sh into small files for profiles/.

usage: pmc_summary.py <gpurun_out/prof_TAG> <TAG>
writes <dir>/summary/<TAG>_kernel_stats.csv (copy of rocprofv3's kernel stats), <TAG>_pmc.json and traffic.json:
per kernel and per launch, FETCH_SIZE / WRITE_SIZE in bytes and the gfx950-corrected HBM bytes (FETCH_SIZE reports half
of a wide streaming read on gfx950 -> doubled, WRITE_SIZE exact; MI355X_MICROARCH.md, HBM section).
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("fot_kernels.hip", "fot_math.hpp", "fot_types.h", "fot_setup.hpp")      # == bench.py


def kernel_source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "integrated_path_planning_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def short(name):
    name = name.split("(")[0]
    for part in name.replace("void ", "").split("::"):
        if part.startswith("k_"):
            return part.split("<")[0]
    return name


def per_kernel(dirname, counter):
    """mean counter value per launch, in bytes (rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB... the unit is checked
    against the header: values are documented as kilobytes)"""
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row["Kernel_Name"])
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(root, "summary")
    os.makedirs(out, exist_ok=True)
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(out, f"{tag}_kernel_stats.csv"))
    fetch = per_kernel(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "write"), "WRITE_SIZE")
    KB = 1024.0                                   # FETCH_SIZE / WRITE_SIZE are reported in KiB
    pmc, traffic = {}, {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fb = fetch.get(k, (0.0, 0))[0] * KB
        wb = write.get(k, (0.0, 0))[0] * KB
        pmc[k] = {"launches_fetch_pass": fetch.get(k, (0, 0))[1], "launches_write_pass": write.get(k, (0, 0))[1],
                  "FETCH_SIZE_bytes_per_launch": fb, "WRITE_SIZE_bytes_per_launch": wb}
        traffic[k] = {"fetch_bytes_reported": fb, "write_bytes": wb, "hbm_bytes_gfx950_corrected": 2.0 * fb + wb}
    for path in sum((glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)
                     for d in ("sq", "sq2", "sq3", "sq4")), []):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k.startswith("k_"):
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, cs in acc.items():
            pmc.setdefault(k, {}).update({c: v[0] / v[1] for c, v in cs.items() if v[1]})
    # which build and workload the counters belong to: bench.py only trusts them for the same kernel sources
    kernel_ms = {}
    for path in glob.glob(os.path.join(out, f"{tag}_kernel_stats.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row.get("Name", ""))
                if k.startswith("k_") and row.get("AverageNs"):
                    kernel_ms[k] = float(row["AverageNs"]) * 1e-6
    meta = {"tag": tag, "source_hash": kernel_source_hash(), "kernel_ms": kernel_ms,
            "instances_per_launch": int(os.environ.get("FOT_PROFILE_INSTANCES", "256")),
            "bench_args": os.environ.get("FOT_PROFILE_ARGS", "")}
    pmc["_meta"] = meta
    traffic["_meta"] = meta
    json.dump(pmc, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    json.dump(pmc, open(os.path.join(out, "pmc.json"), "w"), indent=1)          # what bench.py reads (with traffic.json)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
