"""Randomised parity: random planner parameters, reference paths, ego states and obstacle sets, GPU vs the oracle.

Every per-candidate status, length and cost must agree, as well as the selected path.  Seeds are fixed; a failure
message names the seed so the case can be replayed.
"""
import os

import numpy as np
import pytest

import eps_band
from helpers import EVAL_PATHS, NORTH_STAR_TOL, TIGHT, assert_record_matches_oracle, oracle_plan_for_request, set_eval_path
from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.batch import PackedBatch, PlanRequest
from integrated_path_planning_amd.footprint import EgoFootprint
from integrated_path_planning_amd.planner import BatchPlanner
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def random_path(rng):
    kind = rng.integers(0, 4)
    if kind == 0:                                   # straight, random heading
        L, th = rng.uniform(60, 150), rng.uniform(-np.pi, np.pi)
        t = np.linspace(0, L, rng.integers(2, 16))
        return t * np.cos(th) + rng.uniform(-50, 50), t * np.sin(th) + rng.uniform(-50, 50)
    if kind == 1:                                   # arc
        R, span = rng.uniform(8, 60), rng.uniform(0.5, 2.5)
        th = np.linspace(0, span, rng.integers(8, 40))
        sgn = rng.choice([-1.0, 1.0])
        return R * np.sin(th), sgn * R * (1 - np.cos(th))
    if kind == 2:                                   # gentle random walk in heading
        n = rng.integers(6, 25)
        hd = np.cumsum(rng.normal(0, 0.15, n))
        step = rng.uniform(4, 12, n)
        return np.concatenate([[0], np.cumsum(step * np.cos(hd))]), np.concatenate([[0], np.cumsum(step * np.sin(hd))])
    x = np.linspace(0, rng.uniform(60, 120), rng.integers(8, 30))          # sine lane
    return x, rng.uniform(0.5, 4) * np.sin(x / rng.uniform(8, 25))


def random_planner_kwargs(rng):
    dt = float(rng.choice([0.1, 0.2, 0.125]))
    min_t = float(rng.choice([2.0, 3.0, 4.0]))
    max_t = min_t + float(rng.choice([0.0, 0.5, 1.0]))
    kw = dict(dt=dt, min_t=min_t, max_t=max_t,
              max_speed=float(rng.uniform(6, 16)), max_accel=float(rng.uniform(1.5, 8)),
              max_curvature=float(rng.choice([0.2, 1.0, 10.0])), max_lat_accel=float(rng.uniform(2, 6)),
              d_road_w=float(rng.choice([0.3, 0.5, 1.0])), max_road_width=float(rng.uniform(1.0, 7.0)),
              robot_radius=float(rng.uniform(0.5, 2.0)), obstacle_radius=float(rng.uniform(0.1, 0.5)),
              d_t_s=float(rng.uniform(1.0, 2.5)), k_j=float(rng.choice([0.1, 1.0])), k_t=float(rng.choice([0.1, 1.0])),
              k_d=float(rng.uniform(0.5, 2)), k_s_dot=float(rng.uniform(0.5, 2)), k_lat=1.0, k_lon=float(rng.uniform(0.5, 1.5)),
              chance_epsilon=float(rng.choice([0.0, 0.0, 0.1, 0.3])),
              collision_margin_inflation=float(rng.choice([1.0, 1.2])))
    if rng.random() < 0.35:
        kw["footprint"] = EgoFootprint.multi_circle(float(rng.uniform(3.5, 5)), float(rng.uniform(1.6, 2.1)),
                                                    int(rng.integers(1, 6)))
    return kw


def random_request(rng, sp, kw, dense=False):
    s_end = sp.coeffs()[0][-1]
    s = rng.uniform(0.0, s_end) if rng.random() < 0.9 else rng.uniform(s_end - 3, s_end)
    x, y, yaw, _, _ = [a[0] for a in sp.eval([s])]
    off = rng.normal(0, 0.6)
    ex, ey = x - np.sin(yaw) * off + rng.normal(0, 0.05), y + np.cos(yaw) * off + rng.normal(0, 0.05)
    v = float(rng.choice([0.0, rng.uniform(0, 1), rng.uniform(1, kw["max_speed"])]))
    target = float(rng.choice([0.0, rng.uniform(1.0, kw["max_speed"])]))
    req = PlanRequest(float(ex), float(ey), float(yaw + rng.normal(0, 0.1)), v, float(rng.uniform(-1.5, 1.5)),
                      target_speed=target, last_kappa=float(rng.normal(0, 0.02)),
                      prev_s=float(np.clip(s + rng.normal(0, 2.0), 0, s_end)) if rng.random() < 0.5 else None)
    if rng.random() < 0.3:
        req.overrides = {k: float(kw[k] * rng.uniform(0.6, 2.0)) for k in ("max_accel", "max_speed", "max_lat_accel")
                         if rng.random() < 0.6} or None
    if rng.random() < 0.2:
        req.max_stop_distance = float(rng.uniform(0.05, 12.0))
    n_t = int(round(kw["max_t"] / kw["dt"])) + 1
    ax, ay, ayaw = sp.eval(np.clip(s + rng.uniform(0, 45, 64), 0, s_end))[:3]
    ahead = np.stack([ax, ay], axis=1)

    def beside(idx):                                # points along both road sides, some reaching into the lattice
        reach = kw["max_road_width"] + kw["robot_radius"] + kw["obstacle_radius"]
        off = rng.choice([-1.0, 1.0], len(idx)) * rng.uniform(0.6, 2.0, len(idx)) * reach
        return ahead[idx] + np.stack([-np.sin(ayaw[idx]) * off, np.cos(ayaw[idx]) * off], axis=1)
    mode = rng.integers(0, 4)
    if dense:                                       # crowded scenes: the broad phase's strips, lists and bins fill up
        mode = int(rng.integers(1, 3))
    if rng.random() < (0.8 if dense else 0.5):
        pick = ahead[rng.integers(0, 64, rng.integers(100, 700) if dense else rng.integers(1, 40))]
        req.static = beside(rng.integers(0, 64, len(pick))) + rng.normal(0, 0.3, pick.shape) if dense \
            else pick + rng.normal(0, 4.0, pick.shape)
    if mode in (1, 2):
        P = int(rng.integers(30, 90)) if dense else int(rng.integers(1, 25))
        T = int(rng.choice([1, n_t // 2, n_t, n_t + 3]))
        S = 1 if mode == 1 else (int(rng.integers(20, 64)) if dense else int(rng.integers(2, 24)))
        p0 = beside(rng.integers(0, 64, P)) if dense else ahead[rng.integers(0, 64, P)] + rng.normal(0, 5.0, (P, 2))
        vel = rng.normal(0, 0.3 if dense else 1.2, (S, P, 1, 2))
        t = (np.arange(T) * kw["dt"])[None, None, :, None]
        traj = p0[None, :, None, :] + vel * t + np.cumsum(rng.normal(0, 0.05, (S, P, T, 2)), axis=2)
        if rng.random() < 0.2:                      # a few NaN coordinates: their whole tracks stop being obstacles
            for _ in range(int(rng.integers(1, 4))):
                traj[rng.integers(0, S), rng.integers(0, P), rng.integers(0, T), rng.integers(0, 2)] = np.nan
        if mode == 1:
            req.dyn = traj[0]
        else:
            req.dist = traj
    return req


# FOT_FUZZ_SEEDS=N widens the sweep (default 40 seeds x 6 instances; a 1000-seed sweep was run once per build round)
# FOT_FUZZ_BASE=B moves the sweep to seeds B .. B+N-1
N_SEEDS = int(os.environ.get("FOT_FUZZ_SEEDS", "40"))
SEED_BASE = int(os.environ.get("FOT_FUZZ_BASE", "0"))
N_DENSE = int(os.environ.get("FOT_FUZZ_DENSE_SEEDS", "8"))
FORCE_SEGMENTS = int(os.environ.get("FOT_FUZZ_SEGMENTS", "0"))      # 1..4: every case with that many time segments


@pytest.mark.parametrize("seed", range(SEED_BASE, SEED_BASE + N_SEEDS))
def test_random_configuration(seed):
    run_seed(seed, n_inst=6, dense=False)


@pytest.mark.parametrize("seed", range(SEED_BASE, SEED_BASE + N_DENSE))
def test_random_crowded_configuration(seed):
    run_seed(500000 + seed, n_inst=3, dense=True)


def run_seed(seed, n_inst, dense):
    rng = np.random.default_rng(1000 + seed)
    wx, wy = random_path(rng)
    kw = random_planner_kwargs(rng)
    if dense:                                       # let most candidates reach the collision check
        kw.update(max_curvature=10.0, max_accel=max(kw["max_accel"], 5.0))
    okw = dict(kw)
    fp = okw.pop("footprint", None)
    if fp is not None:
        okw["footprint_offsets"], okw["footprint_radius"] = list(fp.offsets), fp.radius
    params, sp = orc.make_params(**okw), orc.Spline(wx, wy)
    bp = BatchPlanner(waypoints=(wx, wy), **kw)
    reqs = [random_request(rng, sp, kw, dense) for _ in range(n_inst)]
    wants = [oracle_plan_for_request(orc, params, sp, rq, table=True) for rq in reqs]
    # the oracle once, the library under every evaluation kernel it ships (FOT_FUZZ_SEGMENTS: that segment count only)
    paths = tuple(p_ for p_ in os.environ.get("FOT_FUZZ_PATHS", "").split(",") if p_) or EVAL_PATHS   # (sweeps: a subset)
    for path in (paths if not FORCE_SEGMENTS else ("forced",)):
        if FORCE_SEGMENTS:
            bp.set_eval_segments(FORCE_SEGMENTS)
        else:
            set_eval_path(bp, path)
        res = bp.plan_batch(reqs)
        for i, want in enumerate(wants):
            label = f"seed {seed} inst {i} [{path}]"
            cost, status, keep, nt = bp.candidates(i)
            assert len(cost) == want.n_cand, label
            np.testing.assert_array_equal(nt, want.cand_nt, err_msg=label)
            np.testing.assert_array_equal(keep, want.cand_keep, err_msg=label)
            eps_band.check_status_table(bp, i, status, want.cand_status, label)
            # (the start state is the oracle's bit for bit -- correctly rounded hypot / cube in the nearest-point search on
            #  both sides, oracle/check.py -- so every candidate's cost is held to the tight tolerance)
            np.testing.assert_allclose(cost, want.cand_cost, rtol=TIGHT, atol=TIGHT, err_msg=label)
            assert_record_matches_oracle(res.records[i], want, label=label)
        if path == "auto":
            # the same tensors handed over time-major ([T][S][P][2], what the device resampler can write) and as float32:
            # time-major float64 gives the very same records; float32 tensors the records of the float32-rounded inputs
            tsp = bp.plan_packed(PackedBatch(reqs, np.float64, dyn_layout_tsp=True))
            assert bytes(tsp.records) == bytes(res.records), f"seed {seed}: time-major layout changes the records"
            f32 = bp.plan_packed(PackedBatch(reqs, np.float32))
            f32t = bp.plan_packed(PackedBatch(reqs, np.float32, dyn_layout_tsp=True))
            assert bytes(f32t.records) == bytes(f32.records), f"seed {seed}: time-major float32 layout changes the records"
            # ... and through the asynchronous entry point, tensors and records resident in HBM (no pinned staging, no
            # record flags: the path of large batches and of a PyTorch producer)
            import torch
            pb = PackedBatch(reqs, np.float64)
            dev = torch.device("cuda", 0)
            st_t = torch.from_numpy(pb.static_xy).to(dev) if pb.static_xy.size else None
            dy_t = torch.from_numpy(pb.dyn_xy).to(dev) if pb.dyn_xy.size else None
            out_t = torch.zeros(len(reqs) * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
            stream = torch.cuda.Stream(device=dev)
            torch.cuda.synchronize(dev)
            bp.plan_packed_device(pb.with_device_obstacles(st_t.data_ptr() if st_t is not None else None,
                                                           dy_t.data_ptr() if dy_t is not None else None),
                                  out_t.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            assert out_t.cpu().numpy().tobytes() == bytes(res.records), f"seed {seed}: the device entry point differs"


@pytest.mark.parametrize("seed,dense", [(s, False) for s in range(900, 912)] + [(500900 + s, True) for s in range(4)])
def test_time_segments_agree(seed, dense):
    """The evaluation kernel walks a candidate's time range in one piece (batches) or in 2..4 segments merged
    afterwards (a handful of egos): every per-candidate table and the record must not depend on the cut."""
    rng = np.random.default_rng(1000 + seed)
    wx, wy = random_path(rng)
    kw = random_planner_kwargs(rng)
    if dense:
        kw.update(max_curvature=10.0, max_accel=max(kw["max_accel"], 5.0))
    okw = dict(kw)
    fp = okw.pop("footprint", None)
    if fp is not None:
        okw["footprint_offsets"], okw["footprint_radius"] = list(fp.offsets), fp.radius
    sp = orc.Spline(wx, wy)
    bp = BatchPlanner(waypoints=(wx, wy), **kw)
    reqs = [random_request(rng, sp, kw, dense) for _ in range(3)]
    tables, records = [], []
    for n_seg in (1, 2, 3, 4):
        bp.set_eval_segments(n_seg)
        res = bp.plan_batch(reqs)
        tables.append([bp.candidates(i) for i in range(len(reqs))])
        records.append(res.records)
    bp.set_eval_segments(0)
    for n_seg, tab, rec in zip((2, 3, 4), tables[1:], records[1:]):
        for i in range(len(reqs)):
            label = f"seed {seed} inst {i} segments {n_seg}"
            c1, s1, k1, n1 = tables[0][i]
            c, s, k, n = tab[i]
            np.testing.assert_array_equal(s, s1, err_msg=label)
            np.testing.assert_array_equal(k, k1, err_msg=label)
            np.testing.assert_array_equal(n, n1, err_msg=label)
            np.testing.assert_allclose(c, c1, rtol=1e-12, atol=0, err_msg=label)
            a, b = rec[i], records[0][i]
            assert (a.status, a.best_index, a.n_keep) == (b.status, b.best_index, b.n_keep), label
            assert list(a.stats) == list(b.stats), label
            if a.status == 0:
                for f in ("x", "y", "yaw", "v", "a", "c", "s", "d"):
                    np.testing.assert_array_equal(np.ctypeslib.as_array(getattr(a, f))[: a.n_keep],
                                                  np.ctypeslib.as_array(getattr(b, f))[: b.n_keep], err_msg=label)
