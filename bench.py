#!/usr/bin/env python3
"""bench.py -- candidate trajectories / second of the MI355X Frenet planner.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (obstacle preparation -> Frenet state ->
lattice -> Frenet->Cartesian + checks -> collision -> argmin -> selected paths)
over one batch of synthetic instances whose obstacle tensors are already
resident in HBM.

Workload
  N = 1 : BASELINE.json config 4 -- 256 independent ego instances per batch,
          each a 2240-candidate lattice (5 s horizon, dt = 0.1 s) checked
          against a 20-sample x 30-pedestrian x 51-step prediction distribution
          (fp32), eps = 0.
  N > 1 : BASELINE.json config 5 -- 512 instances per GPU (4096 over 8 GPUs,
          contiguous shards), selected-path records all-gathered with RCCL
          inside the step (weak scaling).  `--instances-per-gpu` overrides.
  The timed loop rotates `--batches` (default 8) DISTINCT batches, i.e. more
  than 256 MiB of obstacle tensors, so that no step finds its inputs in the
  Infinity Cache.

`--gpus N` without WORLD_SIZE in the environment starts the N ranks itself
(child `torch.distributed.run`, before this process touches the GPU); under
`torch.distributed.run` it is a plain rank.

Two timed legs of exactly K steps each (barrier + synchronize on both sides):
the SERIAL leg (one plan call in flight) yields `serial`, `kernel_ms` and the
`roofline` of the dominant kernel -- a launch has the GPU to itself there, so
kernel_ms[dominant] <= serial.ms_per_step -- and the HEADLINE leg (`--overlap`
plan calls in flight, default 4: how a server keeps batches in flight) yields
`value` / `ms_per_step`.  `--overlap 1` runs the serial leg only.

Prints ONE JSON line on rank 0 with the extra objects "roofline" (dominant
kernel, HIP events on its stream inside the timed serial leg), "roofline_issue"
(VALU issue bound from the committed PMC profile), "roofline_valu" (SURVEY
8(d)'s nominal brute-force flop count against the fp32 vector peak) and
"cpu_baseline" (the CPU oracle on this host).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SIMD = 1024                    # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                 # MI355X peak engine clock
# Issue cost of one wave64 vector instruction, in cycles, by class.  The guide's figures (MI355X_MICROARCH.md constants
# table: v_fma_f32 2, transcendental 8, fp64 at half rate) are the fallback; what roofline_issue uses is what
# scripts/micro/issue_bench.hip MEASURED on this part (profiles/r03_issue_costs.jsonl: 131 k independent instructions
# per wave at 1 / 2 / 4 / 8 waves per SIMD), relative to v_add_f32 = 2 and interpolated at the number of resident waves
# the committed profile of the kernel reports (measured_issue_cycles below): fp64 arithmetic comes out at 2.8 - 3.2
# rather than 4, fp64 reciprocals at 10 - 12 rather than 16.
ISSUE_CYCLES = {"f64": 4.0, "trans_f64": 16.0, "f32": 2.0, "trans_f32": 8.0, "int": 2.0, "cvt": 4.0, "other": 3.5}
ISSUE_OPS = {"f64": ("v_fma_f64", "v_mul_f64", "v_add_f64"), "trans_f64": ("v_rcp_f64", "v_rsq_f64"),
             "f32": ("v_fma_f32", "v_mul_f32", "v_add_f32"), "trans_f32": ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"),
             "int": ("v_and_b32",),
             # what the class counters leave over: compares, three-source VOP3, 64-bit moves, packed float32, lane reads,
             # plain moves (v_cndmask_b32 is not in the mean: its microbenchmark chains through VCC)
             "other": ("v_cmp_f64", "v_cmp_f32", "v_min3_f32", "v_lshl_or_b32", "v_mov_b64", "v_pk_fma_f32",
                       "v_readlane_b32", "v_mov_b32")}


def measured_issue_cycles(resident_waves):
    """Class weights from profiles/r03_issue_costs.jsonl at `resident_waves` per SIMD (log-linear between the measured
    occupancies), normalised to v_add_f32 = 2 cycles; None if the file is missing."""
    try:
        rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r03_issue_costs.jsonl")) if l.strip()]
    except Exception:
        return None
    cost = {}
    for r in rows:
        cost.setdefault(r["op"], {})[int(r["waves_per_simd"])] = float(r["cycles_per_instr_per_simd"])

    def at(op, w):
        pts = sorted(cost[op].items())
        w = min(max(w, pts[0][0]), pts[-1][0])
        for (w0, c0), (w1, c1) in zip(pts, pts[1:]):
            if w0 <= w <= w1:
                f = (np.log(w) - np.log(w0)) / (np.log(w1) - np.log(w0))
                return c0 + f * (c1 - c0)
        return pts[-1][1]

    base = at("v_add_f32", resident_waves)
    out = {c: 2.0 * float(np.mean([at(o, resident_waves) for o in ops if o in cost])) / base for c, ops in ISSUE_OPS.items()}
    out["cvt"] = out["f64"]                                      # (not measured: conversions to / from fp64 run at its rate)
    return out
ISSUE_CLASSES = {"f64": ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"),
                 "trans_f64": ("SQ_INSTS_VALU_TRANS_F64",),
                 "f32": ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"),
                 "trans_f32": ("SQ_INSTS_VALU_TRANS_F32",),
                 "int": ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"),
                 "cvt": ("SQ_INSTS_VALU_CVT",)}
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, MI355X_MICROARCH.md "8 TB/s peak (spec)"
# algorithmic HBM bytes per candidate of config 3/4/5
B_ALG = 119.0
KERNEL_SOURCES = ("fot_kernels.hip", "fot_math.hpp", "fot_types.h", "fot_setup.hpp")


def kernel_source_hash():
    """Identifies the kernel sources a committed PMC profile was taken on (scripts/pmc_summary.py stores the same)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "integrated_path_planning_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def usable_cores():
    """Cores this process may actually keep busy: the affinity mask, capped by the cgroup CPU quota (a container that
    sees 256 logical cores but is throttled to a 16-core share gains nothing from 256 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--instances-per-gpu", type=int, default=0,
                    help="0 = 256 at --gpus 1 (config 4), 512 at --gpus > 1 (config 5's shard)")
    ap.add_argument("--batches", type=int, default=8, help="distinct batches rotated through the timed loop")
    ap.add_argument("--overlap", type=int, default=4,
                    help="plan calls in flight in the headline leg: consecutive steps alternate between this many "
                         "handles/streams (1 = strictly serial, one leg only)")
    ap.add_argument("--dyn-layout", choices=("spt", "tsp"), default="spt",
                    help="layout of the obstacle tensors: spt = the reference's [S][P][T][2] (default), tsp = time-major "
                         "[T][S][P][2] as libfot's own resampler can write it (FOT_DYN_LAYOUT_TSP)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed K-step headline leg is run this many times; the line reports the median repeat")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="timing diagnostics only (FOT_EVAL_ABLATE / FOT_CULL_ABLATE builds give wrong results)")
    ap.add_argument("--cpu-instances", type=int, default=256,
                    help="instances of the same workload timed on the single-thread CPU oracle (~53 ms each)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with no rank environment: start the N ranks as CHILD processes (this process has not
    touched the GPU and never will) and pass their output and exit code through."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


class Leg:
    """One timed configuration of the step loop: `n_ov` handles on `n_ov` streams, batches rotated."""

    def __init__(self, n_ov, make_planner, dev, batches, out_bytes, pg, rehearse, n_inst=0, wire_bytes=0):
        import torch
        self.torch, self.dev, self.n_ov, self.pg, self.rehearse = torch, dev, n_ov, pg, rehearse
        self.n_inst = n_inst
        self.wire_dev = [torch.zeros(max(n_inst * wire_bytes, 1), dtype=torch.uint8, device=dev)
                         for _ in range(n_ov)] if rehearse else None
        self.planners = [make_planner() for _ in range(n_ov)]
        # explicit streams only: a NULL stream handed to fot_plan_batch_device means "the handle's own stream", which
        # neither torch's default stream nor the collective that follows would wait for
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(n_ov)]
        self.outs = [torch.zeros(out_bytes, dtype=torch.uint8, device=dev) for _ in range(n_ov)]
        self.batches = batches
        self.count = 0

    def step(self):
        torch = self.torch
        i = self.count
        self.count += 1
        b = i % self.n_ov
        bstruct = self.batches[i % len(self.batches)]
        with torch.cuda.stream(self.streams[b]):
            if self.pg is None:
                self.planners[b].plan_packed_device(bstruct, self.outs[b].data_ptr(), self.streams[b].cuda_stream)
                return
            j, send, _ = self.pg.slot()                         # free again: the gather that last read it is done
            st = self.streams[b].cuda_stream
            self.planners[b].plan_packed_device(bstruct, self.outs[b].data_ptr(), st)
            # the records travel in the compact wire form (fot_pack_records_device, same stream)
            if self.rehearse:
                self.planners[b].pack_records_device(self.n_inst, self.outs[b].data_ptr(), self.wire_dev[b].data_ptr(), st)
                send.copy_(self.wire_dev[b])                    # (synchronous D2H: rehearsal only)
            else:
                self.planners[b].pack_records_device(self.n_inst, self.outs[b].data_ptr(), send.data_ptr(), st)
            self.pg.launch(j)

    def fence(self):
        import torch.distributed as dist
        if self.pg is not None:
            self.pg.drain()                                     # every gather of the timed steps is inside the region
            dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def run(self, warmup, steps, profile):
        """W untimed + exactly K timed steps; returns (seconds, per-kernel HIP-event totals)."""
        for _ in range(warmup):
            self.step()
        self.fence()
        if profile:
            for p_ in self.planners:
                p_.profile(True)
                p_.profile_read(reset=True)
        first = self.count
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        elapsed = time.perf_counter() - t0
        prof = {}
        if profile:
            for p_ in self.planners:
                for k_, v_ in p_.profile_read(reset=True).items():
                    a_ = prof.setdefault(k_, {"launches": 0, "total_ms": 0.0})
                    a_["launches"] += v_["launches"]; a_["total_ms"] += v_["total_ms"]
                p_.profile(False)
        return elapsed, prof, first

    def close(self):
        for p_ in self.planners:
            p_.close()


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(self_launch(args))                             # nothing above has touched the GPU

    import torch
    import torch.distributed as dist
    from integrated_path_planning_amd import _abi, synthetic as syn
    from integrated_path_planning_amd.batch import PackedBatch, PlanRequest, request_from_instance
    from integrated_path_planning_amd.distributed import PipelinedAllGather
    from integrated_path_planning_amd.planner import BatchPlanner

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    # FOT_BENCH_REHEARSE=1: dry run of the N > 1 control flow on a box with ONE GPU -- every rank on cuda:0, records
    # gathered through host memory with "gloo".  Not a measurement (the line says so), never used by the driver.
    rehearse = os.environ.get("FOT_BENCH_REHEARSE") == "1" and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI

    n_inst = args.instances_per_gpu or (256 if world == 1 else 512)
    n_rot = max(1, args.batches)
    kw = syn.CONFIG3_PLANNER
    wp = (syn.STRAIGHT_WX, syn.STRAIGHT_WY)
    # batch b of the rotation = seeds [b * world * n_inst, (b+1) * world * n_inst), contiguous shards over the ranks:
    # batch 0 is config 4 (seeds 0..255) at N = 1 and config 5 (seeds 0..4095, 512 per rank) at N = 8
    reqs_rot, packed, dyn_dev, bstructs = [], [], [], []
    for b in range(n_rot):
        s0 = b * world * n_inst + rank * n_inst
        reqs = [request_from_instance(syn.config3_instance(s)) for s in range(s0, s0 + n_inst)]
        pb = PackedBatch(reqs, obstacle_dtype=np.float32, dyn_layout_tsp=args.dyn_layout == "tsp")
        d = torch.from_numpy(pb.dyn_xy).to(dev)                 # obstacle tensors resident in HBM
        reqs_rot.append(reqs); packed.append(pb); dyn_dev.append(d)
        bstructs.append(pb.with_device_obstacles(None, d.data_ptr()))
    obstacle_mb = sum(p.dyn_xy.nbytes for p in packed) / 1e6
    out_bytes = n_inst * _abi.RESULT_BYTES

    def make_planner():
        return BatchPlanner(waypoints=wp, device=local_rank, **kw)

    from integrated_path_planning_amd.distributed import pack_records_host, wire_record_bytes
    wire_bytes = wire_record_bytes(int(round(5.0 / kw["dt"])) + 1)         # CONFIG3_PLANNER: max_t 5 s

    def make_pg(depth):
        # N > 1: the selected-path records of every rank are all-gathered (RCCL over xGMI) in every step, in the compact
        # wire form; the gather of step i runs on the collective's stream while step i+1 plans into the other buffers
        return PipelinedAllGather(n_inst * wire_bytes, world, torch.device("cpu") if rehearse else dev,
                                  depth=depth) if world > 1 else None

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- serial leg: one plan call in flight; per-kernel HIP events -> kernel_ms, roofline
    n_ov = max(1, min(args.overlap, 4))
    pg1 = make_pg(2)
    leg1 = Leg(1, make_planner, dev, bstructs, out_bytes, pg1, rehearse, n_inst, wire_bytes)
    el1, prof, first1 = leg1.run(args.warmup, args.steps, profile=True)
    el1 = reduce_max(el1)
    # the same leg once more without the event pairs around every launch (they cost a few microseconds per kernel)
    el1_plain, _, _ = leg1.run(2, args.steps, profile=False)
    el1_plain = reduce_max(el1_plain)
    # ---- headline leg
    if n_ov > 1:
        pg = make_pg(max(2, n_ov))
        leg = Leg(n_ov, make_planner, dev, bstructs, out_bytes, pg, rehearse, n_inst, wire_bytes)
        # the K-step leg `--repeats` times (W warm-up steps before the first, each repeat bracketed by the barrier +
        # synchronise of Leg.fence and reduced with MAX over the ranks): the line carries the MEDIAN repeat, min / max beside
        runs = []
        for r_ in range(max(1, args.repeats)):
            el_, _, first_ = leg.run(args.warmup if r_ == 0 else 0, args.steps, profile=False)
            runs.append((reduce_max(el_), first_))
        elapsed, first = sorted(runs)[(len(runs) - 1) // 2]
        repeats_ms = [e_ / args.steps * 1e3 for e_, _ in runs]
    else:
        pg, leg, elapsed, first = pg1, leg1, el1, first1
        repeats_ms = [el1 / args.steps * 1e3]

    # ---- candidates actually generated: one untimed pass per rotation batch, counted from its result records
    bp = leg1.planners[0]
    stream = leg1.streams[0]
    torch.cuda.set_stream(stream)                               # everything below (copies, collectives) follows it
    chk = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
    cand_batch, recs0 = [], None
    for b in range(n_rot):
        bp.plan_packed_device(bstructs[b], chk.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        rh = chk.cpu().numpy()
        recs = (_abi.Result * n_inst).from_buffer_copy(rh.tobytes())
        cand_batch.append(sum(int(r.n_cand) for r in recs))
        if b == 0:
            recs0, recs0_host = recs, rh.copy()

    def cand_of_leg(first_step):
        tot = torch.tensor([sum(cand_batch[(first_step + i) % n_rot] for i in range(args.steps))], dtype=torch.int64,
                           device="cpu" if rehearse else dev)
        if world > 1:
            dist.all_reduce(tot)
        return int(tot.item())

    cand_total = cand_of_leg(first)
    cand_serial = cand_of_leg(first1)
    value = cand_total / elapsed

    gathered_ok = None
    if world > 1:
        # one more gathered step on batch 0: every rank's slice of the gathered tensor must hold that rank's records
        assert bp.n_total_samples * 60 + 176 <= wire_bytes
        j, send, recv = pg.slot()
        bp.plan_packed_device(bstructs[0], chk.data_ptr(), stream.cuda_stream)
        if rehearse:
            wtmp = torch.zeros(n_inst * wire_bytes, dtype=torch.uint8, device=dev)
            bp.pack_records_device(n_inst, chk.data_ptr(), wtmp.data_ptr(), stream.cuda_stream)
            send.copy_(wtmp)
        else:
            send.zero_()
            bp.pack_records_device(n_inst, chk.data_ptr(), send.data_ptr(), stream.cuda_stream)
        pg.launch(j)
        g = pg.drain().cpu().numpy()
        torch.cuda.synchronize(dev)
        mine = g[rank * n_inst * wire_bytes:(rank + 1) * n_inst * wire_bytes]
        want = pack_records_host(recs0, n_inst, bp.n_total_samples)       # the host packer on this rank's own records
        from integrated_path_planning_amd.distributed import unpack_records
        allrec = unpack_records(g, world * n_inst, bp.n_total_samples)
        gathered_ok = bool(np.array_equal(mine, want) and all(r.n_cand > 0 for r in allrec))
        ok_t = torch.tensor([1 if gathered_ok else 0], dtype=torch.int64, device="cpu" if rehearse else dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        gathered_ok = bool(ok_t.item())
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events on its stream, inside the timed serial leg)
    dom = max(prof, key=lambda k: prof[k]["total_ms"])
    dom_ms = prof[dom]["total_ms"] / max(prof[dom]["launches"], 1)
    launches_per_step = max(prof[dom]["launches"] / max(args.steps, 1), 1.0)
    cand_launch = cand_batch[0] / launches_per_step
    alg_bytes = B_ALG * cand_launch
    src_hash = kernel_source_hash()

    def committed(name):
        """(per-kernel dict of a committed rocprofv3 --pmc summary, taken on these very kernel sources?)"""
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            return None, False
        return d, d.get("_meta", {}).get("source_hash") == src_hash

    def of_kernel(d, name):
        """entry of the profiled kernel that ran in slot `name` (the evaluation slot has variants: k_evaluate,
        k_evaluate_group, k_evaluate_split -- the profile of this very workload holds the one that ran)"""
        if not d:
            return {}, None
        keys = [name] + sorted(k for k in d if k.startswith(name + "_"))
        hit = max((k for k in keys if k in d), key=lambda k: float(d[k].get("SQ_INSTS_VALU", d[k].get("hbm_bytes_gfx950_corrected", 0.0))), default=None)
        return (d[hit], hit) if hit else ({}, None)

    traffic_d, traffic_fresh = committed("traffic.json")
    inst_launch = n_inst / launches_per_step                       # instances one launch of the dominant kernel handles
    prof_inst_t = int((traffic_d or {}).get("_meta", {}).get("instances_per_launch", 256))
    traffic = of_kernel(traffic_d, dom)[0].get("hbm_bytes_gfx950_corrected") if traffic_fresh else None
    if traffic is not None:
        traffic = traffic * inst_launch / prof_inst_t               # (counted per launch of prof_inst_t instances)
    roofline = {"kernel": dom, "bound": "hbm", "achieved": alg_bytes / (dom_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic,
                "avg_launch_ms": dom_ms, "algorithmic_bytes_per_launch": alg_bytes,
                "candidates_per_launch": cand_launch, "launches_per_step": launches_per_step,
                "measured_in": "serial leg (one plan call in flight)",
                "traffic_source": ("profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same kernel sources; "
                                   "counted on %d instances per launch, scaled to this launch's %d)" % (prof_inst_t, int(inst_launch))
                                   if traffic is not None else
                                   "null: profiles/traffic.json was taken on other kernel sources (hash mismatch)"
                                   if traffic_d else "null: no committed profile"),
                "note": "the path is not HBM-bound (SURVEY 8(d)); the binding resource is VALU issue, see roofline_issue"}
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    if traffic is not None:
        roofline["achieved_from_traffic"] = traffic / (dom_ms * 1e-3) / 1e9
    # What the kernel actually issues (rocprofv3 --pmc, profiles/pmc.json): its vector instructions BY CLASS, each class
    # weighted with its issue cost on a SIMD that holds two or more waves (ISSUE_CYCLES) -- the time the vector pipes
    # would need for exactly this instruction stream -- against the measured launch time.  Next to it, from the same
    # profile: how busy the VALU pipes were and how many waves a SIMD held on average (SQ_* count quad-cycles).
    issue = None
    pmc_d, pmc_fresh = committed("pmc.json")
    pmc_k, pmc_name = of_kernel(pmc_d, dom)
    if "SQ_INSTS_VALU" in pmc_k:
        prof_inst = int(pmc_d.get("_meta", {}).get("instances_per_launch", 256))
        scale = inst_launch / prof_inst                              # (the work is per instance)
        n_valu = float(pmc_k["SQ_INSTS_VALU"])
        by_class = {c: sum(float(pmc_k.get(k, 0.0)) for k in ks) for c, ks in ISSUE_CLASSES.items()}
        have_classes = any(k in pmc_k for ks in ISSUE_CLASSES.values() for k in ks)
        by_class["other"] = max(n_valu - sum(by_class.values()), 0.0) if have_classes else 0.0
        # class weights: measured on this part, at the occupancy the profile of this kernel shows (fallback: the guide's)
        prof_ms0 = pmc_d.get("_meta", {}).get("kernel_ms", {}).get(pmc_name)
        res_waves = (float(pmc_k["SQ_WAVE_CYCLES"]) * 4.0 / (N_SIMD * prof_ms0 * 1e-3 * CLOCK_HZ)
                     if prof_ms0 and "SQ_WAVE_CYCLES" in pmc_k else 2.0)
        weights = measured_issue_cycles(res_waves) or ISSUE_CYCLES
        if have_classes:
            cycles = sum(by_class[c] * weights[c] for c in by_class)
        else:                                                        # an old profile without the class counters
            cycles = n_valu * 4.0
        issue_ms = cycles * scale / (N_SIMD * CLOCK_HZ) * 1e3
        issue = {"kernel": dom, "bound": "valu_issue", "unit": "ms", "achieved": issue_ms, "peak": dom_ms,
                 "frac": issue_ms / dom_ms, "valu_instructions_per_launch": n_valu * scale,
                 "instructions_by_class": {c: v * scale for c, v in by_class.items()} if have_classes else None,
                 "cycles_per_instruction_by_class": {c: round(v, 3) for c, v in weights.items()} if have_classes else {"all": 4.0},
                 "weights_source": ("profiles/r03_issue_costs.jsonl (scripts/micro/issue_bench.hip), v_add_f32 = 2, at %.2f "
                                    "resident waves per SIMD" % res_waves) if weights is not ISSUE_CYCLES else "guide constants",
                 "mean_cycles_per_instruction": cycles / n_valu if n_valu else None,
                 "simds": N_SIMD, "clock_ghz": CLOCK_HZ / 1e9,
                 "profiled_kernel": pmc_name,
                 "source": "profiles/pmc.json" + (" (%s)" % pmc_d.get("_meta", {}).get("tag", "?")),
                 "source_matches_build": pmc_fresh,
                 "note": "sum over instruction classes of executed VALU wave-instructions x issue cycles of the class / "
                         "(1024 SIMDs x 2.4 GHz) over the measured launch time of the serial leg; counted on %d instances "
                         "per launch, scaled to this launch's %d" % (prof_inst, int(inst_launch))}
        prof_ms = pmc_d.get("_meta", {}).get("kernel_ms", {}).get(pmc_name)      # the launch time IN the profiled run
        if prof_ms and "SQ_ACTIVE_INST_VALU" in pmc_k and "SQ_WAVE_CYCLES" in pmc_k:
            simd_cycles = N_SIMD * prof_ms * 1e-3 * CLOCK_HZ
            issue["valu_busy"] = float(pmc_k["SQ_ACTIVE_INST_VALU"]) * 4.0 / simd_cycles
            issue["resident_waves_per_simd"] = float(pmc_k["SQ_WAVE_CYCLES"]) * 4.0 / simd_cycles
            issue["profiled_launch_ms"] = prof_ms
    # SURVEY 8(d)'s NOMINAL vector-flop count per candidate -- brute force, no early exit, no credit for the broad phase:
    # mean samples per candidate x (200 flops of lattice + conversion + checks by convention, + 5 per ego circle and
    # obstacle point) -- against the fp32 vector peak.  A fraction above 1 is the work the broad phase and the early exits
    # removed, not a kernel beating the hardware; the kernel's own bound is roofline_issue.
    mean_nt = 103075.0 / 2240.0                                     # default lattice (sum of N_t over the candidates)
    pts = float(np.mean([(0 if r.static is None else len(r.static)) +
                         (r.dist.shape[0] * r.dist.shape[1] if r.dist is not None else
                          (r.dyn.shape[0] if r.dyn is not None else 0)) for r in reqs_rot[0]]))
    f_alg = mean_nt * (200.0 + 5.0 * max(1, int(kw.get("n_circles", 1) or 1)) * pts)
    valu_nominal = {"kernel": dom, "bound": "valu_nominal", "unit": "TFLOP/s",
                    "achieved": cand_launch * f_alg / (dom_ms * 1e-3) / 1e12, "peak": 157.3,
                    "flops_per_candidate": f_alg, "obstacle_points_per_instance": pts,
                    "note": "SURVEY 8(d) convention: 46.0 samples x (200 + 5 x circles x obstacle points), brute force; "
                            "peak = fp32 vector (the collision inner loop runs in fp32 with fp64 confirmation); "
                            "> 1 means work removed by the broad phase / early exits"}
    valu_nominal["frac"] = valu_nominal["achieved"] / valu_nominal["peak"]
    kernels = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in prof.items() if v["launches"]}

    # ---- parity spot check against the oracle (checker only, outside the timed region)
    from oracle import oracle as orc
    from oracle.check import assert_record_matches_oracle, oracle_plan_for_request
    oparams = orc.make_params(**kw)
    osp = orc.Spline(syn.STRAIGHT_WX, syn.STRAIGHT_WY)
    n_check = 0 if args.no_parity else min(4, n_inst)
    for i in range(n_check):
        assert_record_matches_oracle(recs0[i], oracle_plan_for_request(orc, oparams, osp, reqs_rot[0][i]),
                                     label=f"inst {i}")

    # ---- CPU baseline: the oracle (a C port of the reference algorithm), bounded samples of the same workload
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        n_cpu = min(args.cpu_instances, n_inst)
        t1 = time.perf_counter()
        n_c = 0
        for i in range(n_cpu):
            n_c += oracle_plan_for_request(orc, oparams, osp, reqs_rot[0][i]).n_cand
        dt_cpu = time.perf_counter() - t1
        n_logical = os.cpu_count() or 1
        n_avail = usable_cores()
        cpu = {"value": n_c / dt_cpu, "unit": "candidates/s", "cores": 1, "kind": "port",
               "sample": f"first {n_cpu} instances of batch 0, oracle/fot_oracle.c single thread, {dt_cpu:.1f} s; "
                         f"host: nproc {n_logical}, {n_avail} usable by this process (affinity and cgroup quota)"}
        # the same port on every core this process may use: one instance per task (the C call releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        pool_reqs = [r for rq in reqs_rot for r in rq][:max(n_cpu, min(2048, 8 * n_avail))]
        t1 = time.perf_counter()
        with ThreadPoolExecutor(n_avail) as ex:
            n_mt = sum(ex.map(lambda rq: oracle_plan_for_request(orc, oparams, osp, rq).n_cand, pool_reqs))
        dt_mt = time.perf_counter() - t1
        cpu["threads"] = {"value": n_mt / dt_mt, "cores": n_avail, "nproc": n_logical, "seconds": dt_mt,
                          "sample": f"{len(pool_reqs)} instances of the rotation batches"}

    # ---- plan-step latency for one ego through the host-pointer API (H2D + kernels + D2H)
    latency = None
    host_api = None
    if not args.no_latency and world == 1:
        latency = {}
        for name, pk, mk in (("config2", syn.CONFIG2_PLANNER, syn.config2_instance),
                             ("config3", syn.CONFIG3_PLANNER, syn.config3_instance)):
            p1 = BatchPlanner(waypoints=wp, device=local_rank, **pk)
            pk8 = [PackedBatch([request_from_instance(mk(s))], np.float32) for s in range(8)]
            for b in pk8:
                p1.plan_packed(b)
            ts = []
            for it in range(1200):
                t1 = time.perf_counter()
                p1.plan_packed(pk8[it % 8])
                ts.append(time.perf_counter() - t1)
            ts = np.array(ts) * 1e3
            latency[name] = {"p50_ms": float(np.percentile(ts, 50)), "p95_ms": float(np.percentile(ts, 95)),
                             "calls": len(ts)}
            p1.close()
        # ---- rows f1 / f2 of SURVEY 8 (the stages either side of the path), one ego
        from integrated_path_planning_amd.prediction import PredictionResampler
        rng = np.random.default_rng(0)
        S, P, Lp = 20, 30, 12
        raw = rng.normal(0, 5, (S, Lp, P, 2)).astype(np.float32)
        p0 = rng.normal(0, 5, (P, 2))
        rs = PredictionResampler(bp)
        raw_dev = torch.from_numpy(raw).to(dev)
        obs_dev = torch.zeros((S, P, rs.n_dense + 1, 2), dtype=torch.float32, device=dev)
        ts = []
        for it in range(200):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            rs.resample_device(raw_dev.data_ptr(), np.float32, S, P, p0, p0, 0.2, obs_dev.data_ptr(), np.float32,
                               stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t1)
        t1 = time.perf_counter()
        for s_ in range(S):
            orc.process_prediction(raw[s_].astype(np.float64), p0, 0.2)
        latency["f1_resample_20x30"] = {"p50_ms": float(np.percentile(np.array(ts) * 1e3, 50)),
                                        "cpu_port_ms": (time.perf_counter() - t1) * 1e3}
        p3 = BatchPlanner(waypoints=wp, device=local_rank, **syn.CONFIG3_PLANNER)
        inst3 = syn.config3_instance(1)                       # a NO_PATH instance: the reference would retry twice
        lvl = [dict(), dict(target_speed=0.8 * syn.TARGET_SPEED, overrides=dict(max_accel=3.0, max_speed=11.0)),
               dict(target_speed=0.0, overrides=dict(max_accel=6.0, max_lat_accel=6.0), max_stop_distance=8.0)]
        reqs3 = [PlanRequest(*inst3.ego, dist=inst3.dist, chain_prev_s=bool(j), **kw_) for j, kw_ in enumerate(lvl)]
        one_launch = PackedBatch(reqs3, np.float32)
        seq = [PackedBatch([PlanRequest(*inst3.ego, dist=inst3.dist, **kw_)], np.float32) for kw_ in lvl]
        t_spec, t_seq = [], []
        for it in range(200):
            t1 = time.perf_counter(); p3.plan_packed(one_launch); t_spec.append(time.perf_counter() - t1)
            t1 = time.perf_counter()
            for b_ in seq:
                p3.plan_packed(b_)
            t_seq.append(time.perf_counter() - t1)
        latency["f2_three_level_cycle"] = {"one_launch_p50_ms": float(np.percentile(np.array(t_spec) * 1e3, 50)),
                                           "three_calls_p50_ms": float(np.percentile(np.array(t_seq) * 1e3, 50))}
        p3.close()
        # row f4: whole closed-loop episodes in lock-step (fixture = pedestrian tracks + scenario of the reference run)
        epi = os.path.join(ROOT, "tests", "golden", "closed_loop", "reference_cv_episodes.npz")
        if os.path.exists(epi):
            from integrated_path_planning_amd.closed_loop import BatchedClosedLoop
            z = np.load(epi, allow_pickle=False)
            cfg_ = json.loads(str(z["meta"]))["config"]
            for n_epi, key_ in ((64, "f4_closed_loop"), (256, "f4_closed_loop_256")):
                walls = []
                for _ in range(3):                               # whole runs: the first one also pays for fresh memory
                    with BatchedClosedLoop(cfg_, [z["base_ped_traj"]] * n_epi, device=local_rank) as loop:
                        t1 = time.perf_counter()
                        hists = loop.run()
                        walls.append(time.perf_counter() - t1)
                        steps_ = len(hists[0])
                wall = float(np.median(walls))
                latency[key_] = {
                    "episodes": n_epi, "lock_steps": steps_, "ms_per_lock_step": wall / steps_ * 1e3,
                    "runs_ms_per_lock_step": [w / steps_ * 1e3 for w in walls],
                    "episode_steps_per_s": n_epi * steps_ / wall,
                    "note": "scenario_01 (1261-candidate lattice, 14 pedestrians, cv predictor), %d copies advanced together, "
                            "one libfot call per lock step (fot_loop_step); the reference simulator takes ~131 ms per step "
                            "of ONE episode in the build container" % n_epi}
            # ... and in the headline workload's form: 256 episodes, every step a 20-sample prediction DISTRIBUTION of 30
            # pedestrians per episode handed over as a device tensor (a torch stand-in for the Social-GAN forward passes,
            # which are out of scope: constant velocity + per-sample velocity noise + a random walk, float32, on the
            # GPU), resampled and planned against under the chance constraint inside ONE libfot call per lock step
            n_epi_d, S_d, P_d, steps_d = 256, 20, 30, 60
            cfg_d = dict(cfg_, distribution_aware_planning=True, prediction_method="sgan")
            rng_d = np.random.default_rng(4)
            t_fr = np.arange(400)[:, None, None] * cfg_d["dt"]
            tracks_d = []
            for _ in range(n_epi_d):
                p0 = np.column_stack([rng_d.uniform(-10.0, 90.0, P_d), rng_d.uniform(-25.0, 25.0, P_d)])
                e0_ = np.asarray(cfg_d["ego_initial_state"], float)[:2]
                near_ = np.linalg.norm(p0 - e0_, axis=1) < 8.0      # (nobody starts on top of the ego)
                p0[near_, 1] += np.where(p0[near_, 1] >= e0_[1], 10.0, -10.0)
                hd, sp_ = rng_d.uniform(0.0, 2.0 * np.pi, P_d), rng_d.normal(1.3, 0.2, P_d)
                tracks_d.append(p0[None] + np.column_stack([sp_ * np.cos(hd), sp_ * np.sin(hd)])[None] * t_fr)
            n_ped_d = n_epi_d * P_d
            L_d = int(cfg_d["pred_len"])
            g_ = torch.Generator(device=dev); g_.manual_seed(11)
            dv_d = torch.randn(S_d, 1, n_ped_d, 2, device=dev, generator=g_) * 0.3
            walk_d = torch.cumsum(torch.randn(S_d, L_d, n_ped_d, 2, device=dev, generator=g_) * 0.05, dim=1)
            tk_d = (torch.arange(1, L_d + 1, device=dev, dtype=torch.float32) * 0.4).view(1, L_d, 1, 1)
            t_src = [0.0, 0]

            def device_samples(last, prev):
                t1_ = time.perf_counter()
                la = torch.from_numpy(last.astype(np.float32)).to(dev)
                ve = (la - torch.from_numpy(prev.astype(np.float32)).to(dev)) / 0.4
                m_ = la.shape[0]                                   # (pedestrians of the episodes still running)
                out_ = (la.view(1, 1, -1, 2) + (ve.view(1, 1, -1, 2) + dv_d[:, :, :m_]) * tk_d + walk_d[:, :, :m_]).contiguous()
                torch.cuda.current_stream(dev).synchronize()
                t_src[0] += time.perf_counter() - t1_; t_src[1] += 1
                return out_

            walls_d = []
            for _ in range(3):
                with BatchedClosedLoop(cfg_d, tracks_d, device=local_rank, sample_source=device_samples,
                                       device_samples=True) as loop:
                    for _w in range(5):
                        loop.step()
                    t_src[0], t_src[1] = 0.0, 0
                    t1 = time.perf_counter()
                    ran = [loop.step() for _s in range(steps_d)]
                    walls_d.append((time.perf_counter() - t1, float(np.mean(ran)), t_src[0] / max(t_src[1], 1)))
            wd = sorted(walls_d)[1]
            latency["f4_closed_loop_dist"] = {
                "episodes": n_epi_d, "prediction_samples": S_d, "pedestrians_per_episode": P_d, "lock_steps": steps_d,
                "ms_per_lock_step": wd[0] / steps_d * 1e3, "episodes_running_mean": wd[1],
                "runs_ms_per_lock_step": [w[0] / steps_d * 1e3 for w in walls_d],
                "of_which_sample_source_ms": wd[2] * 1e3,
                "episode_steps_per_s": wd[1] * steps_d / wd[0],
                "note": "scenario_01's planner (1261-candidate lattice) and fail-safe loop, 256 episodes with their own 30 "
                        "scripted pedestrians, distribution_aware_planning: 20 raw samples x 12 steps of all 7680 pedestrians "
                        "as ONE float32 device tensor per lock step (torch stand-in for the Social-GAN forward passes, "
                        "timed beside), resampled to the 51-step grid and planned against inside fot_loop_step -- the "
                        "samples never cross PCIe"}
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            bp.plan_packed(packed[0])
            ts.append(time.perf_counter() - t1)
        host_api = {"candidates_per_s": cand_batch[0] / float(np.median(ts)),
                    "note": "fot_plan_batch with pageable host buffers: H2D of the obstacle tensors and D2H of the records included"}
        # config 5's shard size on this one GPU: the like-for-like base of the N > 1 lines (512 instances per GPU)
        if n_inst != 512:
            reqs5 = [request_from_instance(syn.config3_instance(s)) for s in range(1536, 2048)]   # rank 3's shard
            pb5 = PackedBatch(reqs5, obstacle_dtype=np.float32)
            d5 = torch.from_numpy(pb5.dyn_xy).to(dev)
            leg5 = Leg(n_ov, make_planner, dev, [pb5.with_device_obstacles(None, d5.data_ptr())],
                       512 * _abi.RESULT_BYTES, None, False)
            k5 = max(20, args.steps // 2)
            el5, _, _ = leg5.run(max(3, args.warmup // 4), k5, profile=False)
            rh5 = leg5.outs[0].cpu().numpy()
            c5 = sum(int(r.n_cand) for r in (_abi.Result * 512).from_buffer_copy(rh5.tobytes()))
            host_api["config5_shard_on_one_gpu"] = {
                "instances": 512, "seeds": "1536..2047", "steps": k5, "ms_per_step": el5 / k5 * 1e3,
                "candidates_per_s": c5 * k5 / el5, "plan_calls_in_flight": n_ov,
                "note": "one fixed batch (123 MB of obstacle tensors, below the Infinity Cache size)"}
            leg5.close()

    # the same serial leg on time-major tensors ([T][S][P][2], what fot_resample_predictions writes with FOT_OUT_TMAJOR)
    layout_tsp = None
    if world == 1 and args.dyn_layout == "spt" and not args.no_latency:
        pbt = [PackedBatch(rq, obstacle_dtype=np.float32, dyn_layout_tsp=True) for rq in reqs_rot]
        dt_ = [torch.from_numpy(p_.dyn_xy).to(dev) for p_ in pbt]
        legt = Leg(1, make_planner, dev, [p_.with_device_obstacles(None, d_.data_ptr()) for p_, d_ in zip(pbt, dt_)],
                   out_bytes, None, False)
        kt = max(20, args.steps // 2)
        elt, proft, _ = legt.run(max(3, args.warmup // 2), kt, profile=True)
        layout_tsp = {"ms_per_step": elt / kt * 1e3, "steps": kt, "plan_calls_in_flight": 1,
                      "kernel_ms": {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in proft.items()
                                    if v["launches"]},
                      "note": "obstacle tensors laid out [T][S][P][2] (FOT_DYN_LAYOUT_TSP); the headline uses the "
                              "reference's [S][P][T][2]"}
        legt.close()
        del dt_, pbt

    cfg_name = "config4" if (world == 1 and n_inst == 256) else "config5" if n_inst == 512 else "config4-like"
    line = {
        "metric": "candidate trajectories/sec", "value": value, "unit": "candidates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "repeats": {"n": len(repeats_ms), "ms_per_step": [round(v, 5) for v in repeats_ms], "min": min(repeats_ms),
                    "max": max(repeats_ms), "reported": "median repeat (value, ms_per_step): each repeat is exactly "
                    "`steps` timed steps between a barrier + synchronise on both sides, MAX over the ranks"},
        "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not rehearse else "synthetic (REHEARSAL of the N>1 control flow on one GPU: not a measurement)",
        "config": {"workload": "%s: %d ego instances/GPU x 2240-candidate lattice (5 s, dt 0.1 s), "
                               "20-sample x 30-pedestrian x 51-step fp32 prediction distribution (layout %s), eps=0; "
                               "%d distinct batches rotated (%.0f MB of obstacle tensors per GPU)"
                               % (cfg_name, n_inst, "[S][P][T][2]" if args.dyn_layout == "spt" else "[T][S][P][2]", n_rot,
                                  obstacle_mb),
                   "instances_per_gpu": n_inst, "instances_total": n_inst * world, "rotation_batches": n_rot,
                   "plan_calls_in_flight": n_ov, "candidates_per_step": cand_total // args.steps,
                   "parallelism": "instances sharded over %d GPU(s), RCCL all-gather of %d-byte wire records (fot_result: "
                                  "%d bytes)" % (world, wire_bytes, _abi.RESULT_BYTES)},
        "layout_tsp_serial": layout_tsp,
        "serial": {"ms_per_step": el1 / args.steps * 1e3, "value": cand_serial / el1, "plan_calls_in_flight": 1,
                   "steps": args.steps, "kernel_ms": kernels,
                   "ms_per_step_without_event_pairs": el1_plain / args.steps * 1e3},
        "roofline": roofline, "roofline_issue": issue, "roofline_valu": valu_nominal, "kernel_ms": kernels,
        "kernel_source_hash": src_hash,
        "cpu_baseline": cpu, "latency": latency, "host_api": host_api,
        "parity": {"instances_checked_against_oracle": n_check, "ok": n_check > 0, "all_gather_ok": gathered_ok},
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
