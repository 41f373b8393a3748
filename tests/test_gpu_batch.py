"""GPU parity of batched planning against the CPU oracle, plus size-independent properties
at BASELINE.json's full sizes (config 4: 256 instances x 2240 candidates x 20x30x51 obstacles)."""
import ctypes as C

import numpy as np
import pytest

from helpers import (EVAL_PATHS, TIGHT, assert_record_matches_oracle, oracle_plan_for_request, request_from_instance,
                     set_eval_path)
from integrated_path_planning_amd import _abi, synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch, PlanRequest
from integrated_path_planning_amd.footprint import EgoFootprint
from integrated_path_planning_amd.planner import BatchPlanner
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

WP = (syn.STRAIGHT_WX, syn.STRAIGHT_WY)


def _oracle(kw, wp=WP):
    okw = dict(kw)
    fp = okw.pop("footprint", None)
    if fp is not None:
        okw["footprint_offsets"] = list(fp.offsets)
        okw["footprint_radius"] = fp.radius
    return orc.make_params(**okw), orc.Spline(*wp)


def _check_all(res, reqs, kw, wp=WP, dtype=np.float64, table_every=0, bp=None):
    params, sp = _oracle(kw, wp)
    for i, rq in enumerate(reqs):
        want = oracle_plan_for_request(orc, params, sp, _rounded(rq, dtype), table=bool(table_every) and i % table_every == 0)
        assert_record_matches_oracle(res.records[i], want, label=f"inst {i}")
        if table_every and i % table_every == 0:
            cost, status, keep, nt = bp.candidates(i)
            np.testing.assert_array_equal(status, want.cand_status, err_msg=f"inst {i} status table")
            np.testing.assert_array_equal(keep, want.cand_keep)
            np.testing.assert_allclose(cost, want.cand_cost, rtol=TIGHT, atol=TIGHT)


def _rounded(rq: PlanRequest, dtype):
    """The oracle must see the obstacle values the device sees (float32 round trip when dtype is float32)."""
    if dtype == np.float64:
        return rq
    r = PlanRequest(**rq.__dict__)
    for f in ("static", "dyn", "dist"):
        v = getattr(r, f)
        if v is not None:
            setattr(r, f, np.asarray(v).astype(np.float32).astype(np.float64))
    return r


def test_config2_seeds_vs_oracle():
    """BASELINE config 2: default lattice + 10 static points, seeds 0..31 in one launch."""
    kw = syn.CONFIG2_PLANNER
    bp = BatchPlanner(waypoints=WP, **kw)
    reqs = [request_from_instance(syn.config2_instance(s)) for s in range(32)]
    res = bp.plan_batch(reqs)
    _check_all(res, reqs, kw, table_every=8, bp=bp)


def test_config3_seeds_vs_oracle_fp32_inputs():
    """BASELINE config 3: 20x30x51 distribution, float32 obstacle tensor, seeds 0..23 in one launch.
    Both outcomes (path found / NO_PATH) occur and must match."""
    kw = syn.CONFIG3_PLANNER
    bp = BatchPlanner(waypoints=WP, **kw)
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(24)]
    res = bp.plan_batch(reqs, obstacle_dtype=np.float32)
    _check_all(res, reqs, kw, dtype=np.float32, table_every=6, bp=bp)
    statuses = {res.status(i) for i in range(len(reqs))}
    assert statuses == {_abi.PLAN_OK, _abi.PLAN_NO_PATH}


def test_curved_reference_vs_oracle():
    kw = dict(max_speed=10.0, max_accel=2.0, max_curvature=0.2, dt=0.1, d_road_w=0.5, max_road_width=3.0,
              robot_radius=1.0, obstacle_radius=0.2, k_j=1.0, k_t=1.0)
    wp = (syn.CURVED_WX, syn.CURVED_WY)
    bp = BatchPlanner(waypoints=wp, **kw)
    rng = np.random.default_rng(7)
    reqs = []
    for i in range(16):
        s = rng.uniform(2.0, 30.0)
        x, y, yaw, _, _ = [a[0] for a in orc.Spline(*wp).eval([s])]
        reqs.append(PlanRequest(x + rng.normal(0, 0.3), y + rng.normal(0, 0.3), yaw + rng.normal(0, 0.05),
                                rng.uniform(0.5, 6.0), rng.uniform(-1, 1), target_speed=rng.uniform(2.0, 6.0),
                                last_kappa=rng.normal(0, 0.02), prev_s=float(s + rng.normal(0, 1.0)) if i % 2 else None,
                                static=rng.uniform(-10, 5, (6, 2))))
    res = bp.plan_batch(reqs)
    _check_all(res, reqs, kw, wp=wp, table_every=4, bp=bp)


def test_ragged_batch_vs_oracle():
    """One launch mixing every input shape plan() accepts: no obstacles, static only, single-sample
    dynamic, distributions of different S/P/T, overrides, stop directive, cached nearest point."""
    kw = dict(syn.CONFIG3_PLANNER, chance_epsilon=0.1, collision_margin_inflation=1.15)
    bp = BatchPlanner(waypoints=WP, **kw)
    c3 = [syn.config3_instance(100 + i) for i in range(8)]
    c2 = [syn.config2_instance(200 + i) for i in range(4)]
    e = lambda inst: dict(x=inst.ego[0], y=inst.ego[1], yaw=inst.ego[2], v=inst.ego[3], a=inst.ego[4])
    reqs = [
        PlanRequest(**e(c3[0])),
        PlanRequest(**e(c2[0]), static=c2[0].static),
        PlanRequest(**e(c3[1]), dyn=c3[1].dist[0].astype(np.float64)),
        PlanRequest(**e(c3[2]), dist=c3[2].dist.astype(np.float64)),
        PlanRequest(**e(c3[3]), dist=c3[3].dist[:7, :11, :33].astype(np.float64), static=c2[1].static),
        PlanRequest(**e(c3[4]), dist=c3[4].dist[:3, :29].astype(np.float64), target_speed=4.0,
                    overrides=dict(max_accel=3.0, max_speed=11.0)),
        PlanRequest(**e(c3[5]), dyn=c3[5].dist[2, :5, :1].astype(np.float64), target_speed=0.0,
                    overrides=dict(max_accel=6.0, max_lat_accel=6.0), max_stop_distance=5.0),
        PlanRequest(**e(c3[6]), dist=c3[6].dist.astype(np.float64), prev_s=float(c3[6].ego[0]) + 0.4, last_kappa=0.01),
        PlanRequest(x=12.0, y=0.3, yaw=0.05, v=0.0, a=0.0, dist=c3[7].dist[:5].astype(np.float64)),
        PlanRequest(x=99.5, y=0.0, yaw=0.0, v=5.0, a=0.0),
        PlanRequest(**e(c2[2]), static=c2[2].static, target_speed=12.5),
        PlanRequest(**e(c2[3]), static=np.empty((0, 2)), dyn=np.empty((0, 0, 2))),
    ]
    params, sp = _oracle(kw)
    wants = [oracle_plan_for_request(orc, params, sp, rq, table=True) for rq in reqs]
    for path in EVAL_PATHS:                              # the oracle once, every evaluation kernel against it
        set_eval_path(bp, path)
        res = bp.plan_batch(reqs)
        for i, want in enumerate(wants):
            label = f"inst {i} [{path}]"
            assert_record_matches_oracle(res.records[i], want, label=label)
            cost, status, keep, nt = bp.candidates(i)
            np.testing.assert_array_equal(status, want.cand_status, err_msg=label)
            np.testing.assert_array_equal(keep, want.cand_keep, err_msg=label)
            np.testing.assert_allclose(cost, want.cand_cost, rtol=TIGHT, atol=TIGHT, err_msg=label)


def test_footprint_batch_vs_oracle():
    fp = EgoFootprint.multi_circle(4.5, 1.8, 5)
    kw = dict(syn.CONFIG3_PLANNER, footprint=fp)
    bp = BatchPlanner(waypoints=WP, **kw)
    reqs = [request_from_instance(syn.config3_instance(300 + s, S=6, P=20)) for s in range(6)]
    for i, r in enumerate(reqs):
        r.static = syn.config2_instance(300 + i).static
    res = bp.plan_batch(reqs, obstacle_dtype=np.float32)
    _check_all(res, reqs, kw, dtype=np.float32, table_every=2, bp=bp)


def test_device_api_matches_host_api():
    """fot_plan_batch_device on HBM-resident tensors (the path bench.py times) == fot_plan_batch."""
    import torch
    kw = syn.CONFIG3_PLANNER
    bp = BatchPlanner(waypoints=WP, **kw)
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(12)]
    pb = PackedBatch(reqs, np.float32)
    host = bp.plan_packed(pb)
    dev = torch.device("cuda", 0)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    out = torch.zeros(len(reqs) * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    got = out.cpu().numpy().tobytes()
    assert got == bytes(host.records)[: len(got)]


@pytest.fixture(scope="module")
def config4():
    """BASELINE config 4 at full size: 256 instances in one launch (573 440 candidates)."""
    kw = syn.CONFIG3_PLANNER
    bp = BatchPlanner(waypoints=WP, **kw)
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(256)]
    pb = PackedBatch(reqs, np.float32)
    res = bp.plan_packed(pb)
    return bp, reqs, pb, res, kw


def test_full_size_accounting(config4):
    bp, reqs, pb, res, kw = config4
    n_cand = syn.lattice_size()
    assert n_cand == 2240
    for i in range(len(reqs)):
        r = res.records[i]
        assert r.n_cand == n_cand
        assert sum(r.stats[:8]) <= n_cand                      # silently dropped candidates are not counted
        assert (r.status == _abi.PLAN_OK) == (r.stats[_abi.ST_OK] > 0)
        if r.status == _abi.PLAN_OK:
            assert 0 <= r.best_index < n_cand and 2 <= r.n_keep <= 51 and np.isfinite(r.cost)


def test_full_size_deterministic_and_batch_independent(config4):
    """Same inputs -> bit-identical records; an instance's result does not depend on its batch neighbours
    (planned alone, in a permuted batch, in a sub-batch)."""
    bp, reqs, pb, res, kw = config4
    again = bp.plan_packed(pb)
    assert bytes(again.records) == bytes(res.records)
    rb = _abi.RESULT_BYTES
    ref = bytes(res.records)
    perm = np.random.default_rng(0).permutation(len(reqs))
    shuffled = bp.plan_batch([reqs[j] for j in perm], obstacle_dtype=np.float32)
    sb = bytes(shuffled.records)
    for pos, j in enumerate(perm):
        assert sb[pos * rb:(pos + 1) * rb] == ref[j * rb:(j + 1) * rb], f"instance {j} changed when permuted"
    for j in (0, 17, 255):
        alone = bp.plan_batch([reqs[j]], obstacle_dtype=np.float32)
        assert bytes(alone.records)[:rb] == ref[j * rb:(j + 1) * rb]


@pytest.mark.parametrize("lanes", [2, 3])
def test_lane_split_gives_identical_records(config4, lanes, monkeypatch):
    """FOT_LANES > 1 splits a large batch over internal streams; the records, the candidate tables and the debug
    entry points (which must find the lane an instance ran on) are those of the single-stream run."""
    bp, reqs, pb, res, kw = config4
    monkeypatch.setenv("FOT_LANES", str(lanes))
    split = BatchPlanner(waypoints=WP, **kw)                     # the knob is read when the handle is created
    monkeypatch.delenv("FOT_LANES")
    got = split.plan_packed(pb)
    assert bytes(got.records) == bytes(res.records)
    bp.plan_packed(pb)
    for inst in (0, 100, 128, 200, 255):
        a, b = split.candidates(inst), bp.candidates(inst)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y, err_msg=f"inst {inst}")
    split.close()


def test_full_size_selected_path_is_feasible_and_minimal(config4):
    """The selected candidate is 'ok' in the candidate table and no 'ok' candidate is cheaper or earlier at
    equal cost (first strict minimum, frenet_planner.py:1254-1257)."""
    bp, reqs, pb, res, kw = config4
    bp.plan_packed(pb)
    for i in range(0, len(reqs), 16):
        r = res.records[i]
        cost, status, keep, nt = bp.candidates(i)
        ok = np.flatnonzero(status == _abi.ST_OK)
        if r.status != _abi.PLAN_OK:
            assert len(ok) == 0
            continue
        assert status[r.best_index] == _abi.ST_OK
        assert r.best_index == ok[np.argmin(cost[ok])]
        assert r.cost == cost[r.best_index]
        for k in range(8):
            assert r.stats[k] == int(np.sum(status == k))


def test_full_size_sample_vs_oracle(config4):
    """A spread sample of the 256 instances against the oracle (the oracle needs ~50 ms per instance); every third of
    them (8 instances) with the whole per-candidate table -- status, kept length, cost of all 2240 candidates -- of
    the very launch bench.py times (256 instances, k_evaluate_group)."""
    bp, reqs, pb, res, kw = config4
    again = bp.plan_packed(pb)                           # (other tests plan smaller batches on this handle)
    assert bytes(again.records) == bytes(res.records)
    params, sp = _oracle(kw)
    n_tables = 0
    for n, i in enumerate(range(0, 256, 11)):
        table = n % 3 == 0
        want = oracle_plan_for_request(orc, params, sp, _rounded(reqs[i], np.float32), table=table)
        assert_record_matches_oracle(res.records[i], want, label=f"inst {i}")
        if table:
            cost, status, keep, nt = bp.candidates(i)
            np.testing.assert_array_equal(status, want.cand_status, err_msg=f"inst {i} status table")
            np.testing.assert_array_equal(keep, want.cand_keep, err_msg=f"inst {i}")
            np.testing.assert_allclose(cost, want.cand_cost, rtol=TIGHT, atol=TIGHT, err_msg=f"inst {i}")
            n_tables += 1
    assert n_tables >= 8


@pytest.mark.parametrize("eval_path", ["wave", "split-wave"])
def test_full_size_other_kernels_give_identical_tables(config4, eval_path):
    """The same 256-instance launch through the per-wave kernel (k_evaluate) and through time segments of per-wave
    tiles: records byte-identical to the grouped launch, candidate tables of 8 instances equal."""
    bp, reqs, pb, res, kw = config4
    set_eval_path(bp, "auto")
    assert bytes(bp.plan_packed(pb).records) == bytes(res.records)
    ref_tables = {i: bp.candidates(i) for i in range(0, 256, 37)}
    set_eval_path(bp, eval_path)
    try:
        other = bp.plan_packed(pb)
        assert bytes(other.records) == bytes(res.records)
        for i, (c0, s0, k0, n0) in ref_tables.items():
            c, s_, k, n = bp.candidates(i)
            np.testing.assert_array_equal(s_, s0, err_msg=f"inst {i}")
            np.testing.assert_array_equal(k, k0, err_msg=f"inst {i}")
            np.testing.assert_array_equal(c, c0, err_msg=f"inst {i}")
    finally:
        set_eval_path(bp, "auto")


def test_obstacle_translation_invariance():
    """Moving an obstacle that is far from every candidate does not change anything; moving the world origin
    (ego, path, obstacles shifted by a power-of-two offset) reproduces the same selection."""
    kw = syn.CONFIG3_PLANNER
    inst = syn.config3_instance(3)
    rq = request_from_instance(inst)
    bp = BatchPlanner(waypoints=WP, **kw)
    base = bp.plan_batch([rq]).records[0]
    far = PlanRequest(**rq.__dict__)
    extra = np.zeros((rq.dist.shape[0], 1, rq.dist.shape[2], 2))
    extra[..., 0] = rq.x + 10.0
    extra[..., 1] = 500.0                                        # a pedestrian half a kilometre off the road
    far.dist = np.concatenate([np.asarray(rq.dist, dtype=np.float64), extra], axis=1)
    moved = bp.plan_batch([far]).records[0]
    assert moved.best_index == base.best_index and moved.cost == base.cost
    assert list(moved.stats) == list(base.stats)
    shift = 64.0
    bp2 = BatchPlanner(waypoints=(syn.STRAIGHT_WX + shift, syn.STRAIGHT_WY), **kw)
    sh = PlanRequest(**rq.__dict__)
    sh.x = rq.x + shift
    sh.dist = np.array(rq.dist, dtype=np.float64) + np.array([shift, 0.0])
    r2 = bp2.plan_batch([sh]).records[0]
    assert r2.best_index == base.best_index and r2.status == base.status
    np.testing.assert_allclose(r2.cost, base.cost, rtol=1e-9)
    n = base.n_keep
    np.testing.assert_allclose(np.array(r2.x[:n]) - shift, np.array(base.x[:n]), atol=1e-9)


@pytest.fixture(scope="module")
def config5_shard():
    """BASELINE config 5's shard of rank 3: 512 instances (seeds 1536..2047 of the 4096) in one launch on one GPU --
    1 146 880 candidates, 125 MB of float32 obstacle tensors."""
    kw = syn.CONFIG3_PLANNER
    bp = BatchPlanner(waypoints=WP, **kw)
    from integrated_path_planning_amd.distributed import shard_bounds
    lo, hi = shard_bounds(4096, 8)[3]
    assert (lo, hi) == (1536, 2048)
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(lo, hi)]
    pb = PackedBatch(reqs, np.float32)
    res = bp.plan_packed(pb)
    return bp, reqs, pb, res, kw


def test_config5_shard_accounting_and_determinism(config5_shard):
    bp, reqs, pb, res, kw = config5_shard
    assert len(reqs) == 512
    n_cand = syn.lattice_size()
    n_ok = 0
    for i in range(len(reqs)):
        r = res.records[i]
        assert r.n_cand == n_cand
        assert sum(r.stats[:8]) <= n_cand
        assert (r.status == _abi.PLAN_OK) == (r.stats[_abi.ST_OK] > 0)
        if r.status == _abi.PLAN_OK:
            n_ok += 1
            assert 0 <= r.best_index < n_cand and 2 <= r.n_keep <= 51 and np.isfinite(r.cost)
    assert 0 < n_ok < len(reqs)                                  # both outcomes occur in the shard
    again = bp.plan_packed(pb)
    assert bytes(again.records) == bytes(res.records)
    # the shard's records do not depend on the shard they are planned in: the second half alone, on a fresh handle
    rb = _abi.RESULT_BYTES
    half = BatchPlanner(waypoints=WP, **kw).plan_batch(reqs[256:], obstacle_dtype=np.float32)
    assert bytes(half.records)[: 256 * rb] == bytes(res.records)[256 * rb: 512 * rb]


def test_config5_shard_selection_property(config5_shard):
    """first strict minimum over the 'ok' candidates, histogram == candidate table (every 32nd instance)"""
    bp, reqs, pb, res, kw = config5_shard
    bp.plan_packed(pb)
    for i in range(0, len(reqs), 32):
        r = res.records[i]
        cost, status, keep, nt = bp.candidates(i)
        ok = np.flatnonzero(status == _abi.ST_OK)
        if r.status != _abi.PLAN_OK:
            assert len(ok) == 0
            continue
        assert r.best_index == ok[np.argmin(cost[ok])] and r.cost == cost[r.best_index]
        for k in range(8):
            assert r.stats[k] == int(np.sum(status == k))


def test_config5_shard_sample_vs_oracle(config5_shard):
    """every 16th instance of the shard against the oracle (selected path, cost, histogram)"""
    bp, reqs, pb, res, kw = config5_shard
    params, sp = _oracle(kw)
    for i in range(0, 512, 16):
        want = oracle_plan_for_request(orc, params, sp, _rounded(reqs[i], np.float32))
        assert_record_matches_oracle(res.records[i], want, label=f"seed {1536 + i}")


def test_one_handle_alternating_streams_is_ordered():
    """One handle owns ONE workspace: plan calls enqueued on different caller streams must not overlap on it.  Ten
    plan calls alternate between two streams without any host synchronisation, each followed by a device-resident
    prediction resample on the other stream (shared scratch); every result equals that of a serial run."""
    import torch
    from integrated_path_planning_amd.prediction import PredictionResampler
    kw = syn.CONFIG3_PLANNER
    dev = torch.device("cuda", 0)
    bp = BatchPlanner(waypoints=WP, **kw)
    batches = [PackedBatch([request_from_instance(syn.config3_instance(40 * b + s)) for s in range(40)], np.float32)
               for b in range(4)]
    want = [bytes(bp.plan_packed(pb).records) for pb in batches]
    dyn = [torch.from_numpy(pb.dyn_xy).to(dev) for pb in batches]
    structs = [pb.with_device_obstacles(None, d.data_ptr()) for pb, d in zip(batches, dyn)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = [torch.zeros(40 * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev) for _ in range(10)]
    rs = PredictionResampler(bp)
    rng = np.random.default_rng(5)
    raw = torch.from_numpy(rng.normal(0, 5, (20, 12, 30, 2)).astype(np.float32)).to(dev)
    p0 = rng.normal(0, 5, (30, 2))
    obs = [torch.zeros((20, 30, rs.n_dense + 1, 2), dtype=torch.float32, device=dev) for _ in range(10)]
    torch.cuda.synchronize(dev)
    for i in range(10):
        st = streams[i % 2]
        bp.plan_packed_device(structs[i % 4], outs[i].data_ptr(), st.cuda_stream)
        rs.resample_device(raw.data_ptr(), np.float32, 20, 30, p0, p0, 0.2, obs[i].data_ptr(), np.float32,
                           streams[(i + 1) % 2].cuda_stream)
    torch.cuda.synchronize(dev)
    for i in range(10):
        got = outs[i].cpu().numpy().tobytes()
        assert got == want[i % 4][: len(got)], f"plan call {i} differs from the serial run"
        assert torch.equal(obs[i], obs[0]), f"resample {i} was disturbed"


def test_time_major_obstacle_layout_gives_identical_records():
    """FOT_DYN_LAYOUT_TSP: the same tensors packed [T][S][P][2] -- records and candidate tables are those of the
    reference layout; a ragged batch (single-sample, distributions of different S/P/T, static points, none)."""
    kw = dict(syn.CONFIG3_PLANNER, chance_epsilon=0.1, collision_margin_inflation=1.15)
    bp = BatchPlanner(waypoints=WP, **kw)
    c3 = [syn.config3_instance(700 + i) for i in range(10)]
    e = lambda inst: dict(x=inst.ego[0], y=inst.ego[1], yaw=inst.ego[2], v=inst.ego[3], a=inst.ego[4])
    reqs = [PlanRequest(**e(c3[0]), dist=c3[0].dist),
            PlanRequest(**e(c3[1]), dyn=c3[1].dist[0]),
            PlanRequest(**e(c3[2]), dist=c3[2].dist[:7, :11, :33], static=syn.config2_instance(3).static),
            PlanRequest(**e(c3[3])),
            PlanRequest(**e(c3[4]), dist=c3[4].dist[:64 // 4, :, :1]),
            PlanRequest(**e(c3[5]), dist=np.concatenate([c3[5].dist] * 3, axis=1)),           # 90 pedestrians
            PlanRequest(**e(c3[6]), dyn=c3[6].dist[3, :5, :60 // 2])]
    reqs += [request_from_instance(c) for c in c3[7:]]
    ref = bp.plan_batch(reqs, obstacle_dtype=np.float32)
    tables = [bp.candidates(i) for i in range(len(reqs))]
    got = bp.plan_packed(PackedBatch(reqs, np.float32, dyn_layout_tsp=True))
    assert bytes(got.records) == bytes(ref.records)
    for i in range(len(reqs)):
        for a, b in zip(bp.candidates(i), tables[i]):
            np.testing.assert_array_equal(a, b, err_msg=f"inst {i}")
    got64 = bp.plan_packed(PackedBatch(reqs, np.float64, dyn_layout_tsp=True))
    ref64 = bp.plan_batch(reqs, obstacle_dtype=np.float64)
    assert bytes(got64.records) == bytes(ref64.records)


def test_device_wire_pack_equals_host_pack():
    """fot_pack_records_device on HBM-resident records == the host packer on the same records, byte for byte; and the
    unpacked records agree with the originals to float32 on the path samples, exactly on everything else."""
    import torch
    from integrated_path_planning_amd.distributed import pack_records_host, unpack_records, wire_record_bytes
    kw = syn.CONFIG3_PLANNER
    bp = BatchPlanner(waypoints=WP, **kw)
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(20)]
    pb = PackedBatch(reqs, np.float32)
    host = bp.plan_packed(pb)
    dev = torch.device("cuda", 0)
    nt = bp.n_total_samples
    assert nt == 51
    wb = wire_record_bytes(nt)
    rec_dev = torch.from_numpy(np.frombuffer(bytes(host.records), dtype=np.uint8).copy()).to(dev)
    wire_dev = torch.zeros(len(reqs) * wb, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    bp.pack_records_device(len(reqs), rec_dev.data_ptr(), wire_dev.data_ptr(), st.cuda_stream)
    st.synchronize()
    got = wire_dev.cpu().numpy()
    np.testing.assert_array_equal(got, pack_records_host(host.records, len(reqs), nt))
    back = unpack_records(got, len(reqs), nt)
    for i in range(len(reqs)):
        a, b = host.records[i], back[i]
        assert (a.status, a.best_index, a.n_cand, a.n_keep, a.cost) == (b.status, b.best_index, b.n_cand, b.n_keep, b.cost)
        assert list(a.stats) == list(b.stats) and a.new_last_kappa == b.new_last_kappa and a.new_prev_s == b.new_prev_s
        for f in _abi.PATH_FIELDS:                               # s, x, y: float32 offsets from the record's start state
            np.testing.assert_allclose(np.array(getattr(b, f)[: a.n_keep]), np.array(getattr(a, f)[: a.n_keep]),
                                       rtol=2.0 ** -23, atol=2.0 ** -24 * 100.0 if f in ("s", "x", "y") else 1e-30, err_msg=f)


def test_pack_on_the_handles_stream_waits_for_a_plan_on_a_caller_stream():
    """fot_pack_records_device with stream NULL (the handle's own stream) right behind fot_plan_batch_device on a
    caller's stream, no host synchronisation in between: the pack is ordered after the plan by the handle itself
    (order_begin / order_end), so the wire records are those of a synchronised run -- five times over, with the
    record buffer wiped before every plan so that a pack that ran early would ship zeros."""
    import torch
    from integrated_path_planning_amd.distributed import pack_records_host, wire_record_bytes
    kw = syn.CONFIG3_PLANNER
    bp = BatchPlanner(waypoints=WP, **kw)
    n = 192
    pb = PackedBatch([request_from_instance(syn.config3_instance(3000 + s)) for s in range(n)], np.float32)
    host = bp.plan_packed(pb)
    nt = bp.n_total_samples
    want = pack_records_host(host.records, n, nt)
    dev = torch.device("cuda", 0)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    struct = pb.with_device_obstacles(None, dyn.data_ptr())
    rec_dev = torch.zeros(n * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    wire_dev = torch.zeros(n * wire_record_bytes(nt), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    for rep in range(5):
        with torch.cuda.stream(st):
            rec_dev.zero_()
            wire_dev.zero_()
        bp.plan_packed_device(struct, rec_dev.data_ptr(), st.cuda_stream)
        bp.pack_records_device(n, rec_dev.data_ptr(), wire_dev.data_ptr(), None)
        bp.synchronize()                                                     # the handle's stream: the pack
        np.testing.assert_array_equal(wire_dev.cpu().numpy(), want, err_msg=f"repetition {rep}")


def test_wire_records_far_from_the_origin_and_sharded_planner():
    """A map frame, not a test track: path and ego 10-20 km from the origin.  The wire form (float32 OFFSETS for s, x, y)
    still holds the north star's 1e-5 -- and ShardedPlanner (world size 1: the whole pipeline of a rank without a process
    group: own stream, device pack, gather, unpack) returns what BatchPlanner returns."""
    from integrated_path_planning_amd.distributed import ShardedPlanner
    kw = syn.CONFIG3_PLANNER
    off = np.array([1.0e4, -2.0e4])
    wp = (syn.STRAIGHT_WX + off[0], syn.STRAIGHT_WY + off[1])
    reqs = []
    for s_ in range(6):
        rq = request_from_instance(syn.config3_instance(40 + s_, S=4, P=12))
        rq.x += off[0]; rq.y += off[1]
        rq.dist = rq.dist.astype(np.float64) + off
        reqs.append(rq)
    want = BatchPlanner(waypoints=wp, **kw).plan_batch(reqs, obstacle_dtype=np.float64)
    sp = ShardedPlanner(wp, 0, world=1, rank=0, **kw)
    got, _ = sp.plan(reqs, obstacle_dtype=np.float64)
    assert any(want.records[i].status == _abi.PLAN_OK for i in range(len(reqs)))
    for i in range(len(reqs)):
        a, b = want.records[i], got[i]
        assert (a.status, a.best_index, a.n_cand, a.n_keep, a.cost) == (b.status, b.best_index, b.n_cand, b.n_keep, b.cost)
        assert list(a.stats) == list(b.stats) and a.new_last_kappa == b.new_last_kappa and a.new_prev_s == b.new_prev_s
        assert list(a.frenet0) == list(b.frenet0) and list(a.ref0) == list(b.ref0)
        for f in _abi.PATH_FIELDS:
            np.testing.assert_allclose(np.array(getattr(b, f)[: a.n_keep]), np.array(getattr(a, f)[: a.n_keep]),
                                       rtol=2.0 ** -23, atol=1e-5 if f in ("s", "x", "y") else 1e-30, err_msg=f)
