"""fot_loop_plan / fot_loop_observe: the fused step's two calls against the separate entry points they
replace -- same kernels, so the same bytes -- and their argument checks."""
import numpy as np
import pytest

from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.planner import BatchPlanner
from integrated_path_planning_amd.prediction import PredictionResampler

pytestmark = pytest.mark.gpu

WP = (np.array([0.0, 20.0, 40.0, 60.0, 80.0]), np.array([0.0, 1.0, -1.0, 0.5, 0.0]))


def _same(a, b):
    """two structured arrays field by field (padding aside; NaN == NaN)"""
    assert a.dtype == b.dtype and a.shape == b.shape
    for name in a.dtype.names:
        if not name.startswith("_"):
            np.testing.assert_array_equal(a[name], b[name], err_msg=name)


def _requests(bp, egos, episodes, targets):
    req = np.zeros(len(egos), dtype=bp.LOOP_REQUEST_DT)
    for j, (e, ep, t) in enumerate(zip(egos, episodes, targets)):
        for col, f in enumerate(("x", "y", "yaw", "v", "a")):
            req["ego"][f][j] = e[col]
        req["overrides"][j] = (np.nan, np.nan, np.nan, np.nan)
        req["target_speed"][j], req["max_stop_distance"][j], req["episode"][j] = t, np.nan, ep
    return req


def test_loop_calls_equal_the_separate_calls():
    rng = np.random.default_rng(3)
    bp = BatchPlanner(waypoints=WP, dt=0.1, robot_radius=1.0, obstacle_radius=0.3)
    rs = PredictionResampler(bp, pred_len=12, sgan_dt=0.4, sim_dt=0.1, plan_horizon=5.0)
    static = np.array([[30.0, 4.0], [30.5, 4.0], [31.0, 4.0]])
    bp.loop_set_static(static)
    counts = [5, 0, 7]
    off = np.concatenate([[0], np.cumsum(counts)])
    pos = np.column_stack([rng.uniform(5, 45, off[-1]), rng.uniform(-4, 4, off[-1])])
    vel = rng.normal(0, 1.0, (off[-1], 2))
    obs = np.stack([pos - 0.4 * vel, pos + rng.normal(0, 0.01, pos.shape)]).astype(np.float32)
    prepend = np.array([True, False, False])
    egos = np.array([[2.0, 0.2, 0.05, 6.0, 0.1], [5.0, 0.0, 0.0, 4.0, 0.0], [8.0, -0.5, 0.0, 7.0, -0.2]])
    frame = dict(ped_off=off, ped_pos=pos, ped_vel=vel, obs_last=obs[1], obs_prev=obs[0], prepend=prepend, staleness=0.1,
                 pred_len=rs.pred_len, rp=rs.params, ego=egos[:, :4], ego_radius=1.0, ped_radius=0.3, use_footprint=False)
    req = _requests(bp, egos[[0, 1, 2, 2]], [0, 1, 2, 2], [8.0, 8.0, 8.0, 3.0])
    rec, m = bp.loop_plan(req, frame)
    rec = rec.copy()
    # --- the same through the separate entry points: prediction on the host, per episode
    dyn, d_off, dims, cursor = [], [], [], 0
    for e in range(3):
        lo, hi = off[e], off[e + 1]
        if hi > lo:
            p = rs.predict_cv(obs[:, lo:hi], staleness=0.1, float32_observations=True,
                              current=pos[lo:hi] if prepend[e] else None)
            dyn.append(p.reshape(-1, 2)); d_off.append(cursor); dims.append((1, 1, hi - lo, p.shape[1])); cursor += p.shape[0] * p.shape[1]
        else:
            d_off.append(cursor); dims.append((0, 1, 0, 1))
    ego = np.zeros(4, dtype=bp.EGO_DT)
    for col, f in enumerate(("x", "y", "yaw", "v", "a")):
        ego[f] = egos[[0, 1, 2, 2], col]
    want = bp.plan_arrays(ego, np.array([8.0, 8.0, 8.0, 3.0]), np.full((4, 4), np.nan), np.full(4, np.nan),
                          np.tile(static, (4, 1)), np.arange(5) * 3, np.concatenate(dyn), np.array(d_off)[[0, 1, 2, 2]],
                          np.array(dims)[[0, 1, 2, 2]])
    _same(rec, want)
    assert (rec["status"] == 0).any()
    want_m = bp.safety_metrics_cat(egos[:, :4], off, pos, vel, 1.0, 0.3, use_footprint=False)
    _same(m, want_m)
    # --- a second set of requests against the SAME tensor (no frame): the escalation retries of a step
    rec2, none = bp.loop_plan(_requests(bp, egos[[2]], [2], [0.0]))
    assert none is None
    want2 = bp.plan_arrays(ego[3:], np.array([0.0]), np.full((1, 4), np.nan), np.full(1, np.nan), static, np.array([0, 3]),
                           np.concatenate(dyn), np.array(d_off)[[2]], np.array(dims)[[2]])
    _same(rec2, want2)
    # --- observe: metrics of new states + nearest arc length, in one call and in two halves
    new = egos + np.array([0.6, 0.01, 0.0, 0.1, 0.0])
    prev_s = np.array([np.nan, 5.0, 8.2])
    am, s_now = bp.loop_observe(new, prev_s)
    _same(am, bp.safety_metrics_cat(new[:, :4], off, pos, vel, 1.0, 0.3, use_footprint=False))
    np.testing.assert_array_equal(s_now, bp.nearest_s_arrays(new[:, 0], new[:, 1], new[:, 2], new[:, 3], new[:, 4], prev_s))
    bm, s2 = bp.loop_observe_begin(new, prev_s)()
    _same(bm, am)
    np.testing.assert_array_equal(s2, s_now)
    # --- predictor not ready: the tensor is the current positions (T = 1)
    frame0 = dict(frame, obs_last=None, obs_prev=None)
    rec0, _ = bp.loop_plan(req[:3], frame0)
    want0 = bp.plan_arrays(ego[:3], np.full(3, 8.0), np.full((3, 4), np.nan), np.full(3, np.nan), np.tile(static, (3, 1)),
                           np.arange(4) * 3, pos, off[:-1], np.array([(1, 1, 5, 1), (0, 1, 0, 1), (1, 1, 7, 1)]))
    _same(rec0, want0)
    # --- gather: the first samples of the 15 arrays of chosen records as one block
    block = bp.gather_paths(rec, np.array([3, 0]), 20)
    for j, f in enumerate(_abi.PATH_FIELDS):
        np.testing.assert_array_equal(block[j], rec[f][[3, 0], :20])
    bp.close()


def test_loop_argument_checks():
    bp = BatchPlanner(waypoints=WP, dt=0.1)
    req = _requests(bp, np.array([[2.0, 0.0, 0.0, 5.0, 0.0]]), [0], [8.0])
    with pytest.raises(_abi.FotError, match="no frame"):
        bp.loop_plan(req)
    with pytest.raises(_abi.FotError, match="no frame"):
        bp.loop_observe(np.zeros((1, 5)), np.array([np.nan]))
    frame = dict(ped_off=np.array([0, 2]), ped_pos=np.zeros((2, 2)) + 30.0, ped_vel=np.zeros((2, 2)), obs_last=None,
                 ego=np.array([[2.0, 0.0, 0.0, 5.0]]), ego_radius=1.0, ped_radius=0.3)
    rec, m = bp.loop_plan(req, frame)
    assert len(rec) == 1 and len(m) == 1
    bad = req.copy()
    bad["episode"] = 1
    with pytest.raises(_abi.FotError, match="episode out of range"):
        bp.loop_plan(bad)
    with pytest.raises(_abi.FotError, match="one ego per episode"):
        bp.loop_observe(np.zeros((2, 5)), np.array([np.nan, np.nan]))
    with pytest.raises(_abi.FotError, match="non-decreasing"):
        bp.loop_plan(req, dict(frame, ped_off=np.array([1, 2])))
    with pytest.raises(_abi.FotError, match="no frame"):          # a rejected frame leaves none behind
        bp.loop_plan(req)
    with pytest.raises(_abi.FotError, match="no fot_loop_observe_begin"):
        bp._lib.fot_loop_observe_end.restype = int
        _abi.check(bp._h, bp._lib.fot_loop_observe_end(bp._h, None, None))
    with pytest.raises(_abi.FotError):
        bp.gather_paths(rec, np.array([-1]), 4)
    bp.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FOT_LOOP_FUZZ_SEEDS", "12"))))
def test_loop_calls_equal_the_separate_calls_on_random_frames(seed):
    """Random frames: 1 - 9 episodes with 0 - 12 pedestrians each, random prepend flags, staleness and footprint use,
    a predictor that is sometimes not ready, 0 - 40 static points, requests that revisit episodes (retries) with random
    targets / overrides / stop directives / cached arc lengths -- fot_loop_plan + fot_loop_observe against prediction,
    metrics, plan and nearest point through the separate entry points, field by field."""
    rng = np.random.default_rng(9000 + seed)
    fp = None
    if rng.random() < 0.4:
        from integrated_path_planning_amd.footprint import EgoFootprint
        fp = EgoFootprint.multi_circle(4.5, 2.0, int(rng.integers(2, 5)))
    bp = BatchPlanner(waypoints=WP, dt=float(rng.choice([0.1, 0.2])), robot_radius=1.0, obstacle_radius=0.3, footprint=fp,
                      max_accel=float(rng.uniform(2, 6)))
    rs = PredictionResampler(bp, pred_len=12, sgan_dt=0.4, sim_dt=bp.params.dt, plan_horizon=5.0)
    n_static = int(rng.integers(0, 40)) if rng.random() < 0.6 else 0
    static = np.column_stack([rng.uniform(5, 70, n_static), rng.uniform(-9, 9, n_static)]) if n_static else np.empty((0, 2))
    bp.loop_set_static(static)
    n_ep = int(rng.integers(1, 10))
    counts = rng.integers(0, 13, n_ep)
    off = np.concatenate([[0], np.cumsum(counts)])
    pos = np.column_stack([rng.uniform(0, 75, off[-1]), rng.uniform(-6, 6, off[-1])])
    vel = rng.normal(0, 1.2, (off[-1], 2))
    obs = np.stack([pos - 0.4 * vel, pos + rng.normal(0, 0.02, pos.shape)]).astype(np.float32)
    ready = rng.random() < 0.8
    prepend = rng.random(n_ep) < 0.6
    stale = float(rng.choice([0.0, 0.1, 0.2, 0.3]))
    egos = np.column_stack([rng.uniform(0, 50, n_ep), rng.normal(0, 0.5, n_ep), rng.normal(0, 0.1, n_ep),
                            rng.uniform(0, 9, n_ep), rng.uniform(-1, 1, n_ep)])
    use_fp = bool(rng.random() < 0.5)
    frame = dict(ped_off=off, ped_pos=pos, ped_vel=vel, ego=egos[:, :4], ego_radius=1.0, ped_radius=0.3,
                 use_footprint=use_fp)
    if ready:
        frame.update(obs_last=obs[1], obs_prev=obs[0], prepend=prepend, staleness=stale, pred_len=rs.pred_len, rp=rs.params)
    n_req = int(rng.integers(1, 2 * n_ep + 1))
    ep_of = np.concatenate([np.arange(n_ep), rng.integers(0, n_ep, max(n_req - n_ep, 0))])[:n_req]
    targets = np.where(rng.random(n_req) < 0.2, 0.0, rng.uniform(1, 9, n_req))
    req = _requests(bp, egos[ep_of], ep_of, targets)
    ov = np.where(rng.random((n_req, 4)) < 0.2, rng.uniform(1.5, 12.0, (n_req, 4)), np.nan)
    for j, f in enumerate(("max_speed", "max_accel", "max_curvature", "max_lat_accel")):
        req["overrides"][f] = ov[:, j]
    stop = np.where(rng.random(n_req) < 0.2, rng.uniform(0.1, 10.0, n_req), np.nan)
    req["max_stop_distance"] = stop
    prev_s = np.where(rng.random(n_req) < 0.5, np.clip(egos[ep_of, 0] + rng.normal(0, 1.5, n_req), 0, 79), np.nan)
    req["ego"]["has_prev_s"] = ~np.isnan(prev_s)
    req["ego"]["prev_s"] = np.where(np.isnan(prev_s), 0.0, prev_s)
    req["ego"]["last_kappa"] = rng.normal(0, 0.02, n_req)
    rec, m = bp.loop_plan(req, frame)
    rec = rec.copy()
    # --- the separate entry points
    dyn, d_off, dims, cursor = [], np.zeros(n_ep, np.int64), np.zeros((n_ep, 4), np.int64), 0
    for e in range(n_ep):
        lo, hi = off[e], off[e + 1]
        d_off[e] = cursor
        if hi == lo:
            dims[e] = (0, 1, 0, 1)
            continue
        if ready:
            p = rs.predict_cv(obs[:, lo:hi], staleness=stale, float32_observations=True,
                              current=pos[lo:hi] if prepend[e] else None)
        else:
            p = pos[lo:hi, None, :]
        dyn.append(p.reshape(-1, 2)); dims[e] = (1, 1, hi - lo, p.shape[1]); cursor += p.shape[0] * p.shape[1]
    ego = np.zeros(n_req, dtype=bp.EGO_DT)
    for col, f in enumerate(("x", "y", "yaw", "v", "a")):
        ego[f] = egos[ep_of, col]
    ego["last_kappa"], ego["has_prev_s"], ego["prev_s"] = req["ego"]["last_kappa"], req["ego"]["has_prev_s"], req["ego"]["prev_s"]
    want = bp.plan_arrays(ego, targets, ov, stop, np.tile(static, (n_req, 1)) if n_static else None,
                          np.arange(n_req + 1) * n_static if n_static else None,
                          np.concatenate(dyn) if dyn else None, d_off[ep_of] if dyn else None, dims[ep_of] if dyn else None)
    _same(rec, want)
    _same(m, bp.safety_metrics_cat(egos[:, :4], off, pos, vel, 1.0, 0.3, use_footprint=use_fp))
    new = egos + rng.normal(0, 0.3, egos.shape) * np.array([1, 0.1, 0.02, 0.2, 0.1])
    gps = np.where(rng.random(n_ep) < 0.5, np.clip(new[:, 0], 0, 79), np.nan)
    am, s_now = bp.loop_observe(new, gps)
    _same(am, bp.safety_metrics_cat(new[:, :4], off, pos, vel, 1.0, 0.3, use_footprint=use_fp))
    np.testing.assert_array_equal(s_now, bp.nearest_s_arrays(new[:, 0], new[:, 1], new[:, 2], new[:, 3], new[:, 4], gps))
    bp.close()
