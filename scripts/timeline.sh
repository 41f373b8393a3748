#!/bin/bash
# Diagnostic, on the GPU box only: rebuilds libfot with -DFOT_TIMELINE (per-wave timestamps in k_evaluate; the box's
# copy of the tree is scratch) and prints the wave timeline of the bench workload.  Do not time anything with this build.
set -e
cd "$(dirname "$0")/.."
make -C integrated_path_planning_amd/csrc clean > /dev/null
# (make runs the ISA guard on this build too and fails it on a finding: scripts/isa_check_async.py)
make -C integrated_path_planning_amd/csrc EXTRA="-DFOT_TIMELINE" | grep -v hipcc
# the cut the handle picks (grouped for the default lattice); FOT_TILE_CUT=wave|group forces one
timeout -k 10 200 python3 scripts/timeline.py
