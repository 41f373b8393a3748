"""Sizes at and beyond the comfortable range: obstacle sets larger than k_cull's LDS cache and than one 32-chunk pass of
the broad phase, the longest horizon the build supports, a chance-constraint budget with many samples -- all against
the oracle, candidate table included."""
import numpy as np
import pytest

from helpers import EVAL_PATHS, TIGHT, assert_record_matches_oracle, oracle_plan_for_request, set_eval_path
from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.batch import PlanRequest
from integrated_path_planning_amd.footprint import EgoFootprint
from integrated_path_planning_amd.planner import BatchPlanner
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

WX, WY = np.linspace(0.0, 120.0, 13), np.zeros(13)


def _check(bp, params, sp, reqs):
    res = bp.plan_batch(reqs)
    for i, rq in enumerate(reqs):
        want = oracle_plan_for_request(orc, params, sp, rq, table=True)
        cost, status, keep, nt = bp.candidates(i)
        np.testing.assert_array_equal(status, want.cand_status, err_msg=f"inst {i}")
        np.testing.assert_array_equal(keep, want.cand_keep, err_msg=f"inst {i}")
        np.testing.assert_allclose(cost, want.cand_cost, rtol=TIGHT, atol=TIGHT)
        assert_record_matches_oracle(res.records[i], want, label=f"inst {i}")
    return res


def _plan_resident(bp, pb):
    """The records of fot_plan_batch_device on tensors that lie in HBM already."""
    import torch
    dev = torch.device("cuda", 0)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    out = torch.zeros(pb.n * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(),
                          torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    return out.cpu().numpy().tobytes()


def test_dense_static_crowd_many_chunk_passes():
    """3000 static points in the corridor: ~100 chunks per time step (more than one 32-chunk pass), every strip full."""
    kw = dict(dt=0.2, max_road_width=3.0, d_road_w=1.0, robot_radius=0.3, obstacle_radius=0.1, max_t=4.4)
    rng = np.random.default_rng(5)
    pts = np.column_stack([rng.uniform(5, 60, 3000), rng.uniform(-6, 6, 3000)])
    pts = pts[np.abs(pts[:, 1]) > 0.2]                       # a free lane along the centre line, crowded either side
    bp = BatchPlanner(waypoints=(WX, WY), **kw)
    reqs = [PlanRequest(2.0, 0.1, 0.0, 6.0, 0.0, target_speed=7.0, static=pts),
            PlanRequest(10.0, -0.4, 0.05, 3.0, 0.5, target_speed=5.0, static=pts[::2])]
    res = _check(bp, orc.make_params(**kw), orc.Spline(WX, WY), reqs)
    assert res.records[0].stats[_abi.ST_COLLISION] > 0


def test_distribution_larger_than_cull_cache_with_budget():
    """64 samples x 80 pedestrians = 5120 points per step (far more than the 256 kept obstacles k_cull remembers per step), eps = 0.1
    -> up to 6 violating samples allowed: the exact per-sample counting path, with a 3-circle footprint."""
    fp = EgoFootprint.multi_circle(4.5, 1.8, 3)
    kw = dict(dt=0.2, max_road_width=2.0, d_road_w=1.0, robot_radius=1.0, obstacle_radius=0.2, chance_epsilon=0.1)
    okw = dict(kw, footprint_offsets=list(fp.offsets), footprint_radius=fp.radius)
    rng = np.random.default_rng(9)
    S, P, T = 64, 80, 26
    p0 = np.column_stack([rng.uniform(0, 70, P), rng.uniform(-12, 12, P)])
    vel = rng.normal(0, 1.0, (S, P, 1, 2))
    dist = p0[None, :, None, :] + vel * (np.arange(T) * 0.2)[None, None, :, None]
    bp = BatchPlanner(waypoints=(WX, WY), footprint=fp, **kw)
    reqs = [PlanRequest(3.0, 0.0, 0.0, 5.0, 0.0, target_speed=6.0, dist=dist),
            PlanRequest(20.0, 0.5, 0.0, 7.0, -0.5, target_speed=8.0, dist=dist[:, ::3])]
    _check(bp, orc.make_params(**okw), orc.Spline(WX, WY), reqs)


@pytest.mark.parametrize("S,P,n_block,n_extra,label", [(3, 30, 1, 0, "a few NaN tracks among the remembered ones"),
                                                       (4, 40, 10, 5, "more NaN tracks than the group remembers"),
                                                       (64, 80, 1, 0, "more tracks inside the boxes than the group lists"),
                                                       (64, 80, 10, 300, "both, and more kept obstacles than a step remembers")])
def test_nan_tracks_beyond_every_cache_of_the_lazy_check(S, P, n_block, n_extra, label):
    """k_cull looks for NaNs itself in the caller's [S][P][T] layout -- only in the tracks its boxes touch, up to 256 of
    them per group of steps, up to 32 NaN tracks remembered: every combination of what fits and what does not.  A crowd
    along the left road edge fills the lists; a line of pedestrians ACROSS the lane holds the NaNs (every sample of them,
    plus some of the crowd): with them the lane is free (frenet_planner.py:1211-1219: a track with a NaN anywhere is
    no obstacle at any step), without the NaNs it is blocked.  Time-major layout (flags from the scan blocks): same bytes."""
    from integrated_path_planning_amd.batch import PackedBatch
    kw = dict(dt=0.2, max_road_width=2.0, d_road_w=1.0, robot_radius=0.8, obstacle_radius=0.2, max_t=4.4)
    rng = np.random.default_rng(S * 100 + n_block)
    T = 23
    p0 = np.column_stack([rng.uniform(4, 40, P), rng.uniform(2.3, 3.0, P)])
    p0[:n_block] = np.column_stack([rng.uniform(22, 26, n_block), np.linspace(-3.2, 1.0, n_block)])
    vel = rng.normal(0, 0.05, (S, P, 1, 2))
    clean = p0[None, :, None, :] + vel * (np.arange(T) * 0.2)[None, None, :, None]
    dist = clean.copy()
    bad = [s * P + p for s in range(S) for p in range(n_block)]
    bad += list(rng.choice(np.setdiff1d(np.arange(S * P), bad), n_extra, replace=False))
    for j in bad:
        dist[j // P, j % P, rng.integers(0, T), rng.integers(0, 2)] = np.nan
    bp = BatchPlanner(waypoints=(WX, WY), **kw)
    reqs = [PlanRequest(2.0, 0.1, 0.0, 6.0, 0.0, target_speed=7.0, dist=dist),
            PlanRequest(12.0, -0.3, 0.02, 4.0, 0.3, target_speed=5.0, dist=dist[:, ::2]),
            PlanRequest(2.0, 0.1, 0.0, 6.0, 0.0, target_speed=7.0, dist=clean)]
    res = _check(bp, orc.make_params(**kw), orc.Spline(WX, WY), reqs)
    assert res.records[0].stats[_abi.ST_COLLISION] < res.records[2].stats[_abi.ST_COLLISION], label
    for dtype in (np.float32, np.float64):
        spt = bp.plan_packed(PackedBatch(reqs, dtype))
        tsp = bp.plan_packed(PackedBatch(reqs, dtype, dyn_layout_tsp=True))
        assert bytes(spt.records) == bytes(tsp.records), label
        # HBM-resident tensors (fot_plan_batch_device): no staging pass, so k_cull finds the NaN tracks of [S][P][T] itself
        for tsp_layout in (False, True):
            got = _plan_resident(bp, PackedBatch(reqs, dtype, dyn_layout_tsp=tsp_layout))
            assert got == bytes(spt.records)[: len(got)], f"{label} (resident, time-major {tsp_layout}, {dtype.__name__})"
    bp.close()


_NAN_CHILD = r"""
import faulthandler, hashlib, sys
faulthandler.dump_traceback_later(150, exit=True)
import numpy as np, torch
from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.batch import PackedBatch, PlanRequest
from integrated_path_planning_amd.planner import BatchPlanner
WX, WY = np.linspace(0.0, 120.0, 13), np.zeros(13)
kw = dict(dt=0.2, max_road_width=2.0, d_road_w=1.0, robot_radius=0.8, obstacle_radius=0.2, max_t=4.4)
rng = np.random.default_rng(410)
S, P, T, n_block = 4, 40, 23, 10
p0 = np.column_stack([rng.uniform(4, 40, P), rng.uniform(2.3, 3.0, P)])
p0[:n_block] = np.column_stack([rng.uniform(22, 26, n_block), np.linspace(-3.2, 1.0, n_block)])
dist = p0[None, :, None, :] + rng.normal(0, 0.05, (S, P, 1, 2)) * (np.arange(T) * 0.2)[None, None, :, None]
for j in [s * P + p for s in range(S) for p in range(n_block)]:
    dist[j // P, j % P, rng.integers(0, T), rng.integers(0, 2)] = np.nan
with BatchPlanner(waypoints=(WX, WY), **kw) as bp:
    reqs = [PlanRequest(2.0, 0.1, 0.0, 6.0, 0.0, target_speed=7.0, dist=dist),
            PlanRequest(12.0, -0.3, 0.02, 4.0, 0.3, target_speed=5.0, dist=dist[:, ::2])]
    dev = torch.device("cuda", 0)
    h = hashlib.sha256()
    for dtype in (np.float32, np.float64):
        pb = PackedBatch(reqs, dtype)
        dyn = torch.from_numpy(pb.dyn_xy).to(dev)
        out = torch.zeros(pb.n * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
        bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(),
                              torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        h.update(out.cpu().numpy().tobytes())
print("RECORDS", h.hexdigest(), flush=True)
"""


def test_scan_blocks_for_every_layout_give_the_same_records():
    """FOT_NAN_SCAN=eager (read once per process): k_frenet_state's scan blocks flag the NaN tracks of [S][P][T] tensors
    too, as before round 4, and k_cull looks nothing up itself -- same records as the default, on HBM-resident tensors
    with forty NaN tracks across the lane.  Two child processes, one per setting."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ("lazy", "eager"):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("FOT_NAN_SCAN", None)
        if mode == "eager":
            env["FOT_NAN_SCAN"] = "eager"
        r = subprocess.run([sys.executable, "-c", _NAN_CHILD], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, f"{mode}: {r.stdout[-2000:]}\n{r.stderr[-4000:]}"
        got[mode] = [l for l in r.stdout.splitlines() if l.startswith("RECORDS")][-1]
    assert got["lazy"] == got["eager"]


def test_nan_behind_the_64th_sample_of_a_track():
    """121 samples per track: the wave-wide look through a track takes two loads, the NaN sits in the second."""
    kw = dict(dt=0.05, max_road_width=2.0, d_road_w=1.0, robot_radius=0.8, obstacle_radius=0.2, min_t=5.0, max_t=6.0)
    rng = np.random.default_rng(77)
    S, P, T = 2, 12, 121
    p0 = np.column_stack([rng.uniform(4, 30, P), rng.uniform(-2.5, 2.5, P)])
    dist = p0[None, :, None, :] + rng.normal(0, 0.3, (S, P, 1, 2)) * (np.arange(T) * 0.05)[None, None, :, None]
    for j in range(0, S * P, 2):
        dist[j // P, j % P, rng.integers(64, T), 1] = np.nan
    bp = BatchPlanner(waypoints=(WX, WY), **kw)
    reqs = [PlanRequest(2.0, 0.0, 0.0, 5.0, 0.0, target_speed=6.0, dist=dist),
            PlanRequest(2.0, 0.0, 0.0, 5.0, 0.0, target_speed=6.0, dyn=dist[1])]
    _check(bp, orc.make_params(**kw), orc.Spline(WX, WY), reqs)
    from integrated_path_planning_amd.batch import PackedBatch
    for dtype in (np.float32, np.float64):
        want = bytes(bp.plan_packed(PackedBatch(reqs, dtype)).records)
        for tsp_layout in (False, True):
            got = _plan_resident(bp, PackedBatch(reqs, dtype, dyn_layout_tsp=tsp_layout))
            assert got == want[: len(got)], f"resident, time-major {tsp_layout}, {dtype.__name__}"
    bp.close()


@pytest.mark.parametrize("dt,max_t,n_samples", [(0.1, 6.3, 64), (0.1, 6.4, 65), (0.1, 12.7, 128), (0.1, 12.8, 129),
                                                (0.05, 12.75, 256)])
def test_longest_horizons(dt, max_t, n_samples):
    """Samples per candidate around the wave width and up to the limit of the build (FOT_MAX_NT = 256): 64 samples fill
    the lanes that hold the per-step values exactly once, 65 need the second block of steps for ONE step, 128 two full
    blocks, 129 a third for one step, 256 four.  The dynamic tensor is as long as the horizon, so the late steps do
    collide."""
    kw = dict(dt=dt, min_t=max_t - 4 * dt, max_t=max_t, max_road_width=1.0, d_road_w=0.5, robot_radius=1.0,
              obstacle_radius=0.2, max_speed=20.0)
    rng = np.random.default_rng(2)
    wx = np.arange(0.0, 301.0, 10.0)
    dyn = np.array([12.0, 1.0]) + rng.normal(0, 3.0, (18, 1, 2)) + np.cumsum(rng.normal(0, 0.08, (18, n_samples, 2)), axis=1)
    late = np.array([6.0 * (max_t - 1.0), 0.3]) + np.cumsum(rng.normal(0, 0.05, (6, n_samples, 2)), axis=1)   # met late in the walk
    dyn = np.concatenate([dyn, late])
    bp = BatchPlanner(waypoints=(wx, 0 * wx), **kw)
    reqs = [PlanRequest(1.0, 0.2, 0.0, 6.0, 0.2, target_speed=7.0, dyn=dyn),
            PlanRequest(4.0, -0.1, 0.0, 5.0, 0.0, target_speed=6.0, dist=np.stack([dyn, dyn + 0.3, dyn - 0.2]))]
    params, sp = orc.make_params(**kw), orc.Spline(wx, 0 * wx)
    for path in EVAL_PATHS:
        set_eval_path(bp, path)
        res = _check(bp, params, sp, reqs)
        if res.records[0].status == _abi.PLAN_OK:
            assert res.records[0].n_keep <= n_samples
    assert res.records[0].stats[_abi.ST_COLLISION] > 0
    if n_samples == 256:
        with pytest.raises(_abi.FotError) as ei:
            BatchPlanner(waypoints=(wx, 0 * wx), **dict(kw, max_t=12.8))   # 257 samples: the C ABI says so (FOT_ERR_UNSUPPORTED)
        assert ei.value.code == _abi.ERR_UNSUPPORTED


def test_drop_in_planner_never_raises_beyond_the_capacities():
    """The reference plans any lattice and its plan() never raises (frenet_planner.py:266-268).  Beyond the library's
    capacities the drop-in class keeps that contract: it warns once, plan() returns None with last_check_stats None, and
    last_error says which capacity -- for a planner that cannot be built (501 samples per candidate) and for a single
    call (65 prediction samples) on a planner that otherwise works."""
    from integrated_path_planning_amd.cubic_spline import CubicSpline2D
    from integrated_path_planning_amd.data_structures import EgoVehicleState
    from integrated_path_planning_amd.planner import FrenetPlanner
    csp = CubicSpline2D(list(WX), list(WY))
    ego = EgoVehicleState(5.0, 0.0, 0.0, 5.0, 0.0)
    with pytest.warns(RuntimeWarning, match="capacities"):
        too_fine = FrenetPlanner(csp, dt=0.01)
    assert too_fine.plan(ego, np.empty((0, 2))) is None and too_fine.last_check_stats is None
    assert "FOT_MAX_" in too_fine.last_error                        # (101 horizons and 501 samples: whichever is checked first)
    ok = FrenetPlanner(csp, dt=0.2)
    assert ok.plan(ego, np.empty((0, 2))) is not None and ok.last_error is None
    dist = np.full((65, 2, 26, 2), 400.0)
    with pytest.warns(RuntimeWarning, match="capacities"):
        assert ok.plan(ego, np.empty((0, 2)), dynamic_obstacles_distribution=dist) is None
    assert ok.last_check_stats is None and "sample" in ok.last_error.lower()
    assert ok.plan(ego, np.empty((0, 2)), dynamic_obstacles_distribution=dist[:64]) is not None and ok.last_error is None


def test_empty_and_degenerate_inputs():
    kw = dict(dt=0.2)
    bp = BatchPlanner(waypoints=(WX, WY), **kw)
    params, sp = orc.make_params(**kw), orc.Spline(WX, WY)
    assert len(bp.plan_batch([])) == 0
    reqs = [PlanRequest(5.0, 0.0, 0.0, 5.0, 0.0),                                      # no obstacles at all
            PlanRequest(5.0, 0.0, 0.0, 5.0, 0.0, static=np.empty((0, 2)), dyn=np.empty((0, 0, 2))),
            PlanRequest(5.0, 0.0, 0.0, 5.0, 0.0, dyn=np.full((3, 1, 2), 1e6)),          # everything far away, T = 1
            PlanRequest(500.0, 300.0, 0.0, 5.0, 0.0, static=np.array([[500.0, 300.0]]))]  # ego far off the path
    _check(bp, params, sp, reqs)


def test_more_profiles_than_the_cull_box_cache():
    """31 horizons x 9 terminal speeds = 279 longitudinal profiles: k_cull keeps the boxes of the first 96 in LDS and
    derives the others again where it needs them; k_evaluate blocks span more profiles than fit their LDS window."""
    kw = dict(dt=0.1, min_t=2.0, max_t=5.0, d_t_s=1.0, max_road_width=1.0, d_road_w=1.0, robot_radius=0.8,
              obstacle_radius=0.2, max_speed=12.0)
    rng = np.random.default_rng(4)
    dyn = np.array([25.0, 0.0]) + rng.normal(0, 6.0, (25, 1, 2)) + np.cumsum(rng.normal(0, 0.1, (25, 51, 2)), axis=1)
    static = np.column_stack([rng.uniform(5, 60, 40), rng.uniform(-3, 3, 40)])
    bp = BatchPlanner(waypoints=(WX, WY), **kw)
    reqs = [PlanRequest(2.0, 0.1, 0.0, 6.0, 0.0, target_speed=7.5, dyn=dyn, static=static),
            PlanRequest(8.0, -0.2, 0.02, 4.0, 0.3, target_speed=7.5, dyn=dyn)]
    res = _check(bp, orc.make_params(**kw), orc.Spline(WX, WY), reqs)
    assert res.records[0].n_cand > 279 * 3


def test_lattice_shapes_across_tile_boundaries():
    """k_evaluate's tiles (<= 64 candidates, <= 3 full-length profiles, LDS row budget) on lattices that stress their
    construction: a lateral grid wider than a wave (141 offsets: several tiles per profile), a single-offset grid (one
    candidate per profile: tiles end on the profile limit, not on 64 lanes), many brake-ladder entries, and instances
    with different terminal-speed grids (different tile tables) in one batch."""
    rng = np.random.default_rng(9)
    dyn = np.array([22.0, 0.5]) + rng.normal(0, 5.0, (12, 1, 2)) + np.cumsum(rng.normal(0, 0.1, (12, 41, 2)), axis=1)
    static = np.column_stack([rng.uniform(5, 50, 25), rng.uniform(-6, 6, 25)])
    cases = [
        dict(dt=0.1, min_t=3.8, max_t=4.0, max_road_width=7.0, d_road_w=0.1, robot_radius=0.8, obstacle_radius=0.2),
        dict(dt=0.1, min_t=3.0, max_t=4.0, max_road_width=0.2, d_road_w=0.5, robot_radius=0.8, obstacle_radius=0.2),
        dict(dt=0.2, min_t=6.4, max_t=6.6, max_road_width=2.0, d_road_w=0.5, robot_radius=0.8, obstacle_radius=0.2),
    ]
    for kw in cases:
        bp = BatchPlanner(waypoints=(WX, WY), **kw)
        reqs = [PlanRequest(3.0, 0.2, 0.01, 6.0, 0.2, target_speed=8.0, dyn=dyn, static=static),
                PlanRequest(9.0, -0.3, -0.02, 3.0, -0.4, target_speed=1.2, dyn=dyn),          # fewer terminal speeds
                PlanRequest(15.0, 0.0, 0.0, 0.05, 0.0, target_speed=4.1, static=static),      # standing: no brake ladder
                PlanRequest(5.0, 0.4, 0.03, 9.0, 0.0, target_speed=11.0, dyn=dyn[:, :7])]
        res = _check(bp, orc.make_params(**kw), orc.Spline(WX, WY), reqs)
        assert len({res.records[i].n_cand for i in range(len(reqs))}) >= 3
        bp.close()


def test_debug_hooks_argument_checks_and_frenet_given_state():
    """fot_debug_set_eval_segments rejects anything but 0..4; a FOT_EGO_IS_FRENET record plans from the state as given
    (no nearest-point search: new_prev_s is NaN), a state off the reference path gives "no Frenet state", and the
    records are the same whichever evaluation kernel walks them."""
    bp = BatchPlanner(waypoints=(WX, WY), dt=0.2)
    for bad in (-1, 5, 99):
        with pytest.raises(Exception):
            bp.set_eval_segments(bad)
    ego = PlanRequest(20.0, 0.6, 0.05, 5.0, 0.2)
    rec0 = bp.plan_batch([ego]).records[0]
    given = PlanRequest(*[float(v) for v in rec0.frenet0[:5]], last_kappa=float(rec0.frenet0[5]), is_frenet=True)
    outs = []
    for n_seg in (1, 4):
        bp.set_eval_segments(n_seg)
        r = bp.plan_batch([given]).records[0]
        outs.append(r)
        assert np.isnan(r.new_prev_s) and r.status == rec0.status and r.best_index == rec0.best_index
        np.testing.assert_array_equal(np.ctypeslib.as_array(r.frenet0), np.ctypeslib.as_array(rec0.frenet0))
        np.testing.assert_allclose(np.ctypeslib.as_array(r.x)[: r.n_keep], np.ctypeslib.as_array(rec0.x)[: rec0.n_keep],
                                   rtol=0, atol=1e-12)
    assert outs[0].cost == outs[1].cost and list(outs[0].stats) == list(outs[1].stats)
    bp.set_eval_segments(0)
    off = PlanRequest(500.0, 1.0, 0.0, 0.0, 0.0, is_frenet=True)                       # s beyond the 120 m path
    assert bp.plan_batch([off]).records[0].status == _abi.PLAN_C2F_FAILED
