#!/bin/bash
# Diagnostic, on the GPU box only: rebuilds libfot with -DFOT_TIMELINE (per-wave timestamps in k_evaluate; the box's
# copy of the tree is scratch) and prints the wave timeline of the bench workload.  Do not time anything with this build.
set -e
cd "$(dirname "$0")/.."
make -C integrated_path_planning_amd/csrc clean > /dev/null
make -C integrated_path_planning_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DFOT_TIMELINE" > /dev/null
# (the per-wave cut: the instrumented build of the grouped kernel faults -- 128 VGPRs plus the stamps -- and is not used)
FOT_TILE_CUT=wave FOT_TIMELINE_SLOTS=3072 timeout -k 10 200 python3 scripts/timeline.py
