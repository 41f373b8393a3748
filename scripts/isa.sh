#!/bin/bash
# device ISA of fot_kernels.hip into /tmp/isa/fot.s + register / spill summary per kernel; extra args go to hipcc
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=on --offload-arch=gfx950 -Wno-unused-function -S --cuda-device-only "$@" \
    -o /tmp/isa/fot.s /root/repo/integrated_path_planning_amd/csrc/fot_kernels.hip 2>/dev/null
python3 - <<'PY'
import re
t = open('/tmp/isa/fot.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', t, re.S):
    body = m.group(2)
    g = lambda k: re.search(k + r':\s+(\d+)', body).group(1)
    name = re.sub(r'^_ZN3fot\d+', '', m.group(1))[:28]
    print(f"{name:30s} vgpr {g('.vgpr_count'):>4s} sgpr {g('.sgpr_count'):>4s} sspill {g('.sgpr_spill_count'):>3s} vspill {g('.vgpr_spill_count'):>3s} ")
PY
