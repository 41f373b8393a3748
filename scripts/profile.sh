#!/bin/bash
# Kernel trace + HBM / SQ counters of the bench workload, on the GPU box.  Separate rocprofv3 runs (kernel trace;
# FETCH_SIZE; WRITE_SIZE -- the two TCC counters do not fit one pass; two SQ passes), the program itself after `--`.
#   scripts/profile.sh <tag> [bench args...]      default: the SOLO profile (--overlap 1: one plan call in flight, so
#                                                 every per-launch figure is a launch that has the GPU to itself)
# Outputs under gpurun_out/prof_<tag>/; scripts/pmc_summary.py condenses them into gpurun_out/prof_<tag>/summary/
# (copy what is to be judged into profiles/).
set -e
TAG=${1:-run}
shift || true
EXTRA="${*:---overlap 1}"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-latency --no-parity $EXTRA"
export FOT_PROFILE_ARGS="$ARGS"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 $ARGS > "$OUT/trace.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- python3 $ARGS > "$OUT/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- python3 $ARGS > "$OUT/write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$OUT/sq" -o run -- python3 $ARGS > "$OUT/sq.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/sq2" -o run -- python3 $ARGS > "$OUT/sq2.log" 2>&1 || true
# the VALU instructions by class (what bench.py's roofline_issue weights with the issue costs of each class)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/sq3" -o run -- python3 $ARGS > "$OUT/sq3.log" 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU --output-format csv -d "$OUT/sq4" -o run -- python3 $ARGS > "$OUT/sq4.log" 2>&1 || true
python3 scripts/pmc_summary.py "$OUT" "$TAG"
