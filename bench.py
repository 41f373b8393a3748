#!/usr/bin/env python3
"""bench.py -- candidate trajectories / second of the MI355X Frenet planner.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (obstacle preparation -> Frenet state ->
lattice -> Frenet->Cartesian + checks -> collision -> argmin -> selected paths)
over one batch of synthetic instances whose obstacle tensors are already
resident in HBM.  Workload per GPU = BASELINE.json config 4: 256 independent ego
instances, each a 2240-candidate lattice (5 s horizon, dt = 0.1 s) checked
against a 20-sample x 30-pedestrian x 51-step prediction distribution (fp32).
For N > 1 every rank plans its own 256 instances (weak scaling) and the selected
path records are all-gathered with RCCL inside the step.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra
objects "roofline" (dominant kernel, measured with HIP events on its stream
inside the timed region) and "cpu_baseline" (the CPU oracle on this host).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_SIMD = 1024                    # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                 # MI355X peak engine clock
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, MI355X_MICROARCH.md "HBM3E peak BW"
VALU_FP64_PEAK_TF = 78.6       # fp64 vector peak = half the 157.3 TF fp32 vector peak
# algorithmic HBM bytes and nominal flops per candidate of config 3/4/5
B_ALG = 119.0
F_ALG = 147.2e3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--instances-per-gpu", type=int, default=256)
    ap.add_argument("--overlap", type=int, default=2,
                    help="plan calls in flight: consecutive steps alternate between this many handles/streams, so the "
                         "launch gaps and the draining tail of one step are filled by the next (1 = strictly serial)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="timing diagnostics only (FOT_EVAL_ABLATE / FOT_CULL_ABLATE builds give wrong results)")
    ap.add_argument("--cpu-instances", type=int, default=256,
                    help="instances of the same workload timed on the CPU oracle (~53 ms each)")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from integrated_path_planning_amd import _abi, synthetic as syn
    from integrated_path_planning_amd.batch import PackedBatch
    from integrated_path_planning_amd.planner import BatchPlanner
    from helpers import request_from_instance

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    n_inst = args.instances_per_gpu
    seeds = range(rank * n_inst, (rank + 1) * n_inst)
    kw = syn.CONFIG3_PLANNER
    reqs = [request_from_instance(syn.config3_instance(s)) for s in seeds]
    pb = PackedBatch(reqs, obstacle_dtype=np.float32)
    bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=local_rank, **kw)

    # obstacle tensors + result records resident in HBM (torch = device memory + streams only)
    dyn_dev = torch.from_numpy(pb.dyn_xy).to(dev)
    static_dev = torch.from_numpy(pb.static_xy).to(dev) if pb.static_xy.size else None
    out_dev = torch.zeros(n_inst * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    # N > 1: the selected-path records of every rank are all-gathered (RCCL over xGMI) in every step; the gather of step i
    # runs on the collective's stream while step i+1 plans into the other buffer pair (distributed.PipelinedAllGather)
    from integrated_path_planning_amd.distributed import PipelinedAllGather
    pg = PipelinedAllGather(n_inst * _abi.RESULT_BYTES, world, torch.device("cpu") if rehearse else dev,
                            depth=max(2, min(args.overlap, 4))) if world > 1 else None     # slot j <-> stream j
    bstruct = pb.with_device_obstacles(static_dev.data_ptr() if static_dev is not None else None, dyn_dev.data_ptr())
    stream = torch.cuda.current_stream(dev)
    # steps are independent plan calls: by default they alternate between two handles (each with its own
    # workspace) on two streams, the way a server keeps batches in flight (--overlap 3 is faster still, but then two
    # launches of the dominant kernel share the GPU and its per-launch time no longer says anything about the kernel)
    n_ov = max(1, min(args.overlap, 4))
    planners = [bp] + [BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=local_rank, **kw)
                       for _ in range(n_ov - 1)]
    streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(n_ov - 1)]
    outs = [out_dev] + [torch.zeros_like(out_dev) for _ in range(n_ov - 1)]
    counter = [0]

    def step():
        b = counter[0] % n_ov
        counter[0] += 1
        with torch.cuda.stream(streams[b]):
            if pg is None:
                planners[b].plan_packed_device(bstruct, outs[b].data_ptr(), streams[b].cuda_stream)
                return
            j, out, _ = pg.slot()                               # free again: the gather that last read it is done
            if rehearse:
                planners[b].plan_packed_device(bstruct, outs[b].data_ptr(), streams[b].cuda_stream)
                out.copy_(outs[b])                              # (synchronous D2H: rehearsal only)
            else:
                planners[b].plan_packed_device(bstruct, out.data_ptr(), streams[b].cuda_stream)
            pg.launch(j)

    def fence():
        if pg is not None:
            pg.drain()                                          # every gather of the timed steps is inside the region
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    for p_ in planners:
        p_.profile(True)
        p_.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = {}
    for p_ in planners:                                         # per-kernel HIP-event totals over all handles
        for k_, v_ in p_.profile_read(reset=True).items():
            a_ = prof.setdefault(k_, {"launches": 0, "total_ms": 0.0})
            a_["launches"] += v_["launches"]; a_["total_ms"] += v_["total_ms"]
        p_.profile(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # candidates actually generated (from the result records)
    gathered = None
    if pg is not None:
        gathered = pg.drain()
        out_dev = pg.send[(pg.step - 1) % pg.depth]
    recs_host = out_dev.cpu().numpy()
    recs = (_abi.Result * n_inst).from_buffer_copy(recs_host.tobytes())
    cand_local = sum(int(r.n_cand) for r in recs)
    cand_total = torch.tensor([cand_local], dtype=torch.int64, device="cpu" if rehearse else dev)
    if world > 1:
        dist.all_reduce(cand_total)
    cand_total = int(cand_total.item())
    value = cand_total * args.steps / elapsed

    gathered_ok = None
    if world > 1:
        # every rank's slice of the gathered tensor must hold that rank's records
        g = gathered.cpu().numpy()
        rb = _abi.RESULT_BYTES
        mine = g[rank * n_inst * rb:(rank + 1) * n_inst * rb]
        allrec = (_abi.Result * (world * n_inst)).from_buffer_copy(g.tobytes())
        gathered_ok = bool(np.array_equal(mine, recs_host) and all(r.n_cand > 0 for r in allrec))
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events on its stream, inside the timed region)
    dom = max(prof, key=lambda k: prof[k]["total_ms"])
    dom_ms = prof[dom]["total_ms"] / max(prof[dom]["launches"], 1)
    # candidates of one launch: a step splits the batch over the handle's lanes (one launch of each kernel per lane)
    launches_per_step = max(prof[dom]["launches"] / max(args.steps, 1), 1.0)
    cand_launch = cand_local / launches_per_step
    alg_bytes = B_ALG * cand_launch
    # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled as the
    # gfx950 note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE); null when no profile is committed for it
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_gfx950_corrected")
        except Exception:
            traffic = None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": alg_bytes / (dom_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic,
                "avg_launch_ms": dom_ms, "algorithmic_bytes_per_launch": alg_bytes,
                "candidates_per_launch": cand_launch, "launches_per_step": launches_per_step}
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    valu = {"kernel": dom, "bound": "valu_fp64", "achieved": F_ALG * cand_launch / (dom_ms * 1e-3) / 1e12,
            "peak": VALU_FP64_PEAK_TF, "unit": "TFLOP/s",
            "note": "nominal brute-force flops of SURVEY 8(d) (every candidate sample x every obstacle point); the "
                    "broad phase skips ~96% of those pair tests, so this can exceed the peak"}
    valu["frac"] = valu["achieved"] / valu["peak"]
    # what the kernel actually issues (rocprofv3 --pmc SQ_INSTS_VALU of the same command, profiles/): the time its vector
    # instructions alone would take at one wave64 instruction per 4 cycles per SIMD, against the measured launch time
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))[dom]     # copy of the latest <tag>_pmc.json
        n_valu = float(pmc["SQ_INSTS_VALU"])
        issue_ms = n_valu * 4.0 / (N_SIMD * CLOCK_HZ) * 1e3
        valu["issue_bound"] = {"source": "profiles/pmc.json", "valu_instructions_per_launch": n_valu,
                               "cycles_per_instruction": 4, "simds": N_SIMD, "clock_ghz": CLOCK_HZ / 1e9,
                               "bound_ms": issue_ms, "frac": issue_ms / dom_ms,
                               "note": "valid for the default workload (256 instances per launch), which the profile ran"}
    except Exception:
        pass
    kernels = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in prof.items() if v["launches"]}

    # ---- parity spot check against the oracle (checker only, outside the timed region)
    from oracle import oracle as orc
    from helpers import assert_record_matches_oracle, oracle_plan_for_request
    oparams = orc.make_params(**kw)
    osp = orc.Spline(syn.STRAIGHT_WX, syn.STRAIGHT_WY)
    n_check = 0 if args.no_parity else min(4, n_inst)
    for i in range(n_check):
        assert_record_matches_oracle(recs[i], oracle_plan_for_request(orc, oparams, osp, reqs[i]), label=f"inst {i}")

    # ---- CPU baseline: the oracle (a C port of the reference algorithm), single thread, bounded sample
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        n_cpu = min(args.cpu_instances, n_inst)
        t1 = time.perf_counter()
        n_c = 0
        for i in range(n_cpu):
            n_c += oracle_plan_for_request(orc, oparams, osp, reqs[i]).n_cand
        dt_cpu = time.perf_counter() - t1
        cpu = {"value": n_c / dt_cpu, "unit": "candidates/s", "cores": 1, "kind": "port",
               "sample": f"first {n_cpu} instances of the same batch, oracle/fot_oracle.c single thread, "
                         f"{dt_cpu:.1f} s; host has {os.cpu_count()} logical cores"}
        # the same port on the box's CPU share: one instance per task, threads (the C call releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        n_thr = min(16, os.cpu_count() or 1)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(n_thr) as ex:
            n_mt = sum(ex.map(lambda rq: oracle_plan_for_request(orc, oparams, osp, rq).n_cand, reqs[:n_cpu]))
        dt_mt = time.perf_counter() - t1
        cpu["threads"] = {"value": n_mt / dt_mt, "cores": n_thr, "seconds": dt_mt}

    # ---- plan-step latency for one ego through the host-pointer API (H2D + kernels + D2H)
    latency = None
    host_api = None
    if not args.no_latency and world == 1:
        latency = {}
        for name, pk, mk in (("config2", syn.CONFIG2_PLANNER, syn.config2_instance),
                             ("config3", syn.CONFIG3_PLANNER, syn.config3_instance)):
            p1 = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=local_rank, **pk)
            packed = [PackedBatch([request_from_instance(mk(s))], np.float32) for s in range(8)]
            for b in packed:
                p1.plan_packed(b)
            ts = []
            for it in range(400):
                t1 = time.perf_counter()
                p1.plan_packed(packed[it % 8])
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
        p3 = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=local_rank, **syn.CONFIG3_PLANNER)
        inst3 = syn.config3_instance(1)                       # a NO_PATH instance: the reference would retry twice
        lvl = [dict(), dict(target_speed=0.8 * syn.TARGET_SPEED, overrides=dict(max_accel=3.0, max_speed=11.0)),
               dict(target_speed=0.0, overrides=dict(max_accel=6.0, max_lat_accel=6.0), max_stop_distance=8.0)]
        from integrated_path_planning_amd.batch import PlanRequest
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
            n_epi = 64
            with BatchedClosedLoop(cfg_, [z["base_ped_traj"]] * n_epi, device=local_rank) as loop:
                t1 = time.perf_counter()
                hists = loop.run()
                wall = time.perf_counter() - t1
            steps_ = len(hists[0])
            latency["f4_closed_loop"] = {
                "episodes": n_epi, "lock_steps": steps_, "ms_per_lock_step": wall / steps_ * 1e3,
                "episode_steps_per_s": n_epi * steps_ / wall,
                "note": "scenario_01 (1261-candidate lattice, 14 pedestrians, cv predictor), 64 copies advanced together; "
                        "the reference simulator takes ~131 ms per step of ONE episode in the build container"}
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            bp.plan_packed(pb)
            ts.append(time.perf_counter() - t1)
        host_api = {"candidates_per_s": cand_local / float(np.median(ts)),
                    "note": "fot_plan_batch with pageable host buffers: H2D of the obstacle tensors and D2H of the records included"}

    line = {
        "metric": "candidate trajectories/sec", "value": value, "unit": "candidates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not rehearse else "synthetic (REHEARSAL of the N>1 control flow on one GPU: not a measurement)",
        "config": {"workload": "config4: %d ego instances/GPU x 2240-candidate lattice (5 s, dt 0.1 s), "
                               "20-sample x 30-pedestrian x 51-step fp32 prediction distribution, eps=0" % n_inst,
                   "instances_per_gpu": n_inst, "plan_calls_in_flight": n_ov, "candidates_per_step": cand_total,
                   "parallelism": "instances sharded over %d GPU(s), RCCL all-gather of %d-byte path records"
                                  % (world, _abi.RESULT_BYTES)},
        "roofline": roofline, "roofline_valu": valu, "kernel_ms": kernels,
        "cpu_baseline": cpu, "latency": latency, "host_api": host_api,
        "parity": {"instances_checked_against_oracle": n_check, "ok": n_check > 0, "all_gather_ok": gathered_ok},
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
