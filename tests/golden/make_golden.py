#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE planner (build container only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference]

The reference (mnhrk15/integrated_path_planning) is imported read-only from
``--ref``; its only missing dependency on the planner path, ``loguru`` (logging),
is replaced by an in-process no-op module.  Nothing of the reference is copied:
the outputs are data (inputs + expected outputs) written to tests/golden/*.npz.
Those fixtures are what travels to the GPU box; this script never runs there.

Each case file holds
  meta (JSON): planner kwargs, waypoints name, ego, planner state, target speed, overrides ...
  static / dyn / dist            obstacle tensors fed to plan()
  sp_*                           reference spline coefficients
  frenet0, ref0, prev_s_after    Frenet initial state and nearest-point result
  cand_cost/status/keep/nt       one row per generated candidate (generation order)
  stats                          last_check_stats as 8 ints (-1 = key absent), or all -2 when None
  best_index, best_cost, best_*  selected path (15 arrays), last_kappa_after
  probe_idx, probe_*             full 15-array dumps of a few candidates
"""
import argparse
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from integrated_path_planning_amd import synthetic as syn  # noqa: E402

STATUS_NAMES = ["max_speed_error", "max_accel_error", "max_curvature_error", "max_lat_accel_error",
                "road_bound_error", "collision_error", "ok", "stop_distance_error"]
ST_DROPPED = 8
FIELDS = ["t", "s", "s_d", "s_dd", "s_ddd", "d", "d_d", "d_dd", "d_ddd", "x", "y", "yaw", "v", "a", "c"]


def import_reference(ref_root):
    lg = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    lg.logger = _Logger()
    sys.modules["loguru"] = lg
    sys.path.insert(0, ref_root)
    from src.planning.frenet_planner import FrenetPlanner
    from src.planning.cubic_spline import CubicSpline2D
    from src.core.data_structures import EgoVehicleState
    from src.core.footprint import EgoFootprint
    return FrenetPlanner, CubicSpline2D, EgoVehicleState, EgoFootprint


def waypoints(name):
    if isinstance(name, dict):                       # a random case carries its own waypoints
        return np.asarray(name["wx"], float), np.asarray(name["wy"], float)
    if name == "straight":
        return syn.STRAIGHT_WX, syn.STRAIGHT_WY
    if name == "curved":
        return syn.CURVED_WX, syn.CURVED_WY
    if name == "arc5":
        th = np.linspace(0.0, 1.5 * np.pi, 60)
        return 5.0 * np.sin(th), 5.0 * (1.0 - np.cos(th))
    if name == "straight60":
        return np.linspace(0, 60, 25), np.zeros(25)
    if name == "scenario01":
        return np.arange(0.0, 61.0, 10.0), np.zeros(7)
    raise KeyError(name)


SCEN01 = dict(max_speed=10.0, max_accel=2.0, max_curvature=0.2, max_lat_accel=3.0, dt=0.1, d_road_w=0.3,
              max_road_width=2.7, robot_radius=1.0, obstacle_radius=0.2, min_t=4.0, max_t=5.0,
              d_t_s=5.0 / 3.6, k_j=1.0, k_t=1.0, k_d=1.0, k_s_dot=1.0, k_lat=1.0, k_lon=1.0)
SCEN03 = dict(SCEN01, d_road_w=0.5, max_road_width=3.0)
ARC = dict(max_speed=13.9, max_accel=8.0, max_curvature=10.0, dt=0.1, d_road_w=0.5, max_road_width=7.0,
           robot_radius=1.0, min_t=4.0, max_t=5.0, d_t_s=1.39)


def scen01_peds(t0=0.0, T=51, dt=0.1):
    """scripted constant-velocity pedestrians of scenario_01 (SURVEY 8(c) substitute)."""
    init = np.array([
        [18.0, 11.0, 0.0, -1.3], [19.0, 10.0, 0.0, -1.2], [20.0, 11.5, 0.0, -1.4], [21.0, 9.0, 0.0, -1.1],
        [22.0, 12.0, 0.0, -1.3], [28.0, -18.0, 0.0, 1.2], [29.0, -17.0, 0.0, 1.3], [30.0, -19.0, 0.0, 1.1],
        [31.0, -16.0, 0.0, 1.2], [32.0, -19.5, 0.0, 1.4], [15.0, 10.0, 0.2, -1.0], [35.0, -10.0, -0.2, 1.0],
        [25.0, 20.0, 0.0, -1.8], [40.0, -5.0, -0.5, 0.5]])
    t = t0 + np.arange(T) * dt
    return init[:, None, 0:2] + init[:, None, 2:4] * t[None, :, None]


def build_cases():
    cases = []

    def add(name, path, planner, ego, **kw):
        c = dict(name=name, path=path, planner=planner, ego=[float(v) for v in ego],
                 target_speed=kw.pop("target_speed", syn.TARGET_SPEED),
                 overrides=kw.pop("overrides", None), max_stop=kw.pop("max_stop", None),
                 prev_s=kw.pop("prev_s", None), last_kappa=kw.pop("last_kappa", 0.0),
                 footprint=kw.pop("footprint", None),
                 static=kw.pop("static", np.empty((0, 2))), dyn=kw.pop("dyn", None), dist=kw.pop("dist", None))
        assert not kw, kw
        cases.append(c)

    # --- config 2: default lattice, 10 static points ---
    for seed in range(4):
        inst = syn.config2_instance(seed)
        add(f"cfg2_s{seed}", "straight", syn.CONFIG2_PLANNER, inst.ego, static=inst.static)
    inst = syn.config2_instance(4)
    add("cfg2_noobs", "straight", syn.CONFIG2_PLANNER, inst.ego)
    # --- config 3: 20x30x51 distribution, eps = 0 ---
    for seed in range(3):
        inst = syn.config3_instance(seed)
        add(f"cfg3_s{seed}", "straight", syn.CONFIG3_PLANNER, inst.ego, dist=inst.dist.astype(np.float64),
            dyn=inst.dyn.astype(np.float64))
    # single-sample dynamic with margin inflation
    inst = syn.config3_instance(5)
    add("single_infl", "straight", dict(syn.CONFIG3_PLANNER, collision_margin_inflation=1.25), inst.ego,
        dyn=inst.dist[3, :14].astype(np.float64))
    # chance constraint eps = 0.1 (floor(0.1*20) = 2 violations allowed)
    inst = syn.config3_instance(6)
    add("chance_eps01", "straight", dict(syn.CONFIG3_PLANNER, chance_epsilon=0.1), inst.ego,
        dist=inst.dist.astype(np.float64))
    # 3-circle footprint + distribution + static
    inst = syn.config3_instance(7)
    st = syn.config2_instance(7).static
    add("footprint3", "straight", dict(syn.CONFIG3_PLANNER), inst.ego, dist=inst.dist.astype(np.float64),
        static=st + np.array([inst.ego[0] - syn.config2_instance(7).ego[0], 0.0]),
        footprint=dict(length=4.5, width=1.8, n=3))
    # 5-circle footprint, single sample
    inst = syn.config3_instance(8)
    add("footprint5_single", "straight", dict(syn.CONFIG3_PLANNER, collision_margin_inflation=1.1), inst.ego,
        dyn=inst.dist[0].astype(np.float64), footprint=dict(length=4.5, width=1.8, n=5))
    # CAUTION-like overrides
    inst = syn.config3_instance(9)
    add("caution", "straight", syn.CONFIG3_PLANNER, inst.ego, dist=inst.dist.astype(np.float64),
        target_speed=0.8 * syn.TARGET_SPEED, overrides=dict(max_accel=3.0, max_speed=0.8 * 50.0 / 3.6))
    # EMERGENCY-like: target 0, relaxed accel/lat-accel, stop-distance directive
    inst = syn.config3_instance(10)
    add("emergency_stop", "straight", syn.CONFIG3_PLANNER, inst.ego, dist=inst.dist.astype(np.float64),
        target_speed=0.0, overrides=dict(max_accel=6.0, max_lat_accel=6.0), max_stop=6.0)
    add("emergency_stop_tight", "straight", syn.CONFIG3_PLANNER, [10.0, 0.2, 0.01, 3.0, -0.5],
        target_speed=0.0, overrides=dict(max_accel=6.0, max_lat_accel=6.0), max_stop=2.5)
    # standstill (brake ladder gated off, low-speed curvature regime)
    add("standstill", "straight", syn.CONFIG2_PLANNER, [12.0, 0.3, 0.05, 0.0, 0.0])
    add("creep", "straight", syn.CONFIG2_PLANNER, [12.0, -0.4, -0.03, 0.05, 0.3], target_speed=2.0)
    # near the spline end: lockstep truncation
    add("trunc_end", "straight", syn.CONFIG2_PLANNER, [78.0, 0.1, 0.0, 7.0, 0.0])
    add("trunc_end60", "straight60", dict(ARC, max_speed=10.0), [45.0, 0.0, 0.0, 6.0, 0.0], target_speed=6.0)
    add("past_end", "straight", syn.CONFIG2_PLANNER, [99.5, 0.0, 0.0, 5.0, 0.0])
    # cached nearest-point window + previous-path curvature
    add("prev_s_window", "straight", syn.CONFIG2_PLANNER, [33.3, 0.7, 0.08, 6.0, 0.4], prev_s=32.7,
        last_kappa=0.013)
    add("prev_s_edge", "straight", syn.CONFIG2_PLANNER, [55.0, -0.5, 0.0, 5.0, 0.0], prev_s=30.0)
    # curved reference (scenario_03 waypoints)
    add("curved_a", "curved", SCEN03, [-25.0, 2.3, 0.02, 4.0, 0.2], target_speed=5.0)
    add("curved_b", "curved", SCEN03, [-9.0, 2.7, -0.05, 3.0, 0.0], target_speed=5.0, last_kappa=-0.02,
        static=np.array([[-2.0, 1.0], [1.0, -4.0], [3.5, -9.0]]))
    add("curved_c", "curved", dict(SCEN03, max_curvature=1.0, max_road_width=5.0), [-3.0, 1.6, -0.6, 2.5, 0.1],
        target_speed=4.0, prev_s=27.0)
    # radius-5 arc: singularity guard
    add("arc_singular", "arc5", ARC, [5.0 * np.sin(0.4), 5.0 * (1 - np.cos(0.4)), 0.4, 3.0, 0.0],
        target_speed=3.0)
    # scenario_01 lattice with 14 scripted CV pedestrians, single sample
    add("scen01_t0", "scenario01", SCEN01, [0.0, 0.0, 0.0, 5.0, 0.0], target_speed=6.0, dyn=scen01_peds(0.0))
    add("scen01_t8", "scenario01", SCEN01, [14.0, 0.1, 0.01, 4.2, -0.3], target_speed=6.0, dyn=scen01_peds(8.0),
        prev_s=13.6, last_kappa=0.004)
    add("scen01_dist", "scenario01", dict(SCEN01, chance_epsilon=0.05), [14.0, 0.1, 0.01, 4.2, -0.3],
        target_speed=4.8, overrides=dict(max_accel=3.0, max_speed=6.0),
        dist=np.stack([scen01_peds(8.0 + 0.15 * k) + 0.02 * k for k in range(20)]))
    # module defaults (dt = 0.2)
    add("defaults_dt02", "straight", dict(), [20.0, 0.0, 0.0, 6.0, 0.0],
        static=np.array([[35.0, 0.5], [42.0, -1.0]]))
    # dynamic tensor shorter than the horizon: time index clips to T-1
    inst = syn.config3_instance(11)
    add("short_T", "straight", syn.CONFIG3_PLANNER, inst.ego, dyn=inst.dist[0, :, :20].astype(np.float64))
    add("cur_pos_only", "straight", syn.CONFIG3_PLANNER, inst.ego, dyn=inst.dist[0, :, :1].astype(np.float64))
    # NaN samples: np.min / np.max over a pedestrian's track propagate the NaN, so _hits_dynamic drops that pedestrian
    # at EVERY time step (frenet_planner.py:1211-1219); a NaN static point never matches the box mask (:1186-1191)
    inst = syn.config3_instance(2)
    d = inst.dist.astype(np.float64).copy()
    d[:, [0, 3, 5, 7, 11, 13, 17, 19, 23, 29], 50, 1] = np.nan     # these pedestrians vanish from every sample
    d[2, 1, 0, 1] = np.nan                                        # ... and pedestrian 1 from sample 2 only
    add("nan_ped_dist", "straight", syn.CONFIG3_PLANNER, inst.ego, dist=d)
    dd = inst.dist[4].astype(np.float64).copy()
    dd[::2, 50, 1] = np.nan                                       # every other pedestrian: last sample NaN
    st = syn.config2_instance(2).static + np.array([inst.ego[0] - syn.config2_instance(2).ego[0], 0.0])
    st = np.concatenate([st, [[np.nan, 0.0], [inst.ego[0] + 12.0, np.nan]]])
    add("nan_ped_single", "straight", dict(syn.CONFIG3_PLANNER, collision_margin_inflation=1.2), inst.ego, dyn=dd,
        static=st)
    # --- round 2: other lattice shapes and option combinations
    inst = syn.config3_instance(5)
    add("two_speeds", "straight", syn.CONFIG3_PLANNER, inst.ego, target_speed=1.0,            # terminal speeds 1.0, 0.0 only
        dist=inst.dist[:8].astype(np.float64))
    inst = syn.config3_instance(6)
    add("overrides_tight", "straight", syn.CONFIG3_PLANNER, [22.0, -0.3, 0.02, 5.0, 0.4],
        overrides={"max_speed": 6.0, "max_accel": 1.2, "max_curvature": 0.15, "max_lat_accel": 1.0},
        dist=inst.dist[:6].astype(np.float64))
    inst = syn.config3_instance(7)
    add("footprint_chance_infl", "straight", dict(syn.CONFIG3_PLANNER, chance_epsilon=0.1, collision_margin_inflation=1.2),
        inst.ego, dist=inst.dist.astype(np.float64), footprint=dict(length=4.5, width=1.8, n=3))
    inst = syn.config3_instance(8, S=64, P=10)
    add("dist_s64", "straight", dict(syn.CONFIG3_PLANNER, chance_epsilon=0.05), inst.ego, dist=inst.dist.astype(np.float64))
    inst = syn.config3_instance(9)
    add("stop_directive_dist", "straight", syn.CONFIG3_PLANNER, [30.0, 0.2, 0.0, 3.0, -0.5], max_stop=6.0,
        target_speed=0.0, dist=inst.dist[:5].astype(np.float64))
    rng = np.random.default_rng(77)
    crowd = np.column_stack([rng.uniform(15.0, 75.0, 300), rng.choice([-1.0, 1.0], 300) * rng.uniform(1.5, 9.0, 300)])
    add("static_crowd", "straight", syn.CONFIG2_PLANNER, [12.0, 0.4, 0.03, 6.5, 0.1], static=crowd)
    add("curved_slow_turn", "curved", dict(SCEN03, max_curvature=1.0), [-6.0, 2.2, -0.9, 0.6, -0.4], target_speed=1.5,
        last_kappa=-0.2)
    # --- round 3: more than 64 samples per candidate / more than 32 horizons (FOT_MAX_NT 128, FOT_MAX_TI 64)
    inst = syn.config3_instance(12, S=6, P=20, T=101, dt=0.05)
    add("dt005_dist", "straight", dict(syn.CONFIG3_PLANNER, dt=0.05), inst.ego, dist=inst.dist.astype(np.float64))
    inst = syn.config3_instance(13, S=4, P=16)
    add("min_t1_dist", "straight", dict(syn.CONFIG3_PLANNER, min_t=1.0), inst.ego, dist=inst.dist.astype(np.float64),
        static=np.array([[inst.ego[0] + 9.0, 1.2], [inst.ego[0] + 17.0, -2.1]]))
    add("dt005_curved", "curved", dict(SCEN03, dt=0.05, max_curvature=1.0), [-20.0, 2.4, 0.0, 3.5, 0.1], target_speed=4.0,
        static=np.array([[-8.0, 1.5], [-1.0, -1.0], [2.0, -6.0]]), prev_s=9.0)
    inst = syn.config3_instance(14, S=1, P=8, T=101, dt=0.05)
    add("dt005_min_t2_single", "straight", dict(syn.CONFIG2_PLANNER, dt=0.05, min_t=2.0, max_road_width=3.0), inst.ego,
        dyn=inst.dist[0].astype(np.float64), target_speed=6.0)
    add("dt005_stop", "straight", dict(syn.CONFIG3_PLANNER, dt=0.05, max_road_width=3.0), [30.0, 0.2, 0.0, 3.0, -0.5],
        max_stop=6.0, target_speed=0.0, overrides=dict(max_accel=6.0, max_lat_accel=6.0))
    # --- round 4: 251 samples per candidate, 51 horizons (dt = 0.02 s; FOT_MAX_NT 256)
    inst = syn.config3_instance(15, S=3, P=12, T=251, dt=0.02)
    add("dt002_dist", "straight", dict(syn.CONFIG3_PLANNER, dt=0.02, max_road_width=2.0), inst.ego,
        dist=inst.dist.astype(np.float64))
    add("dt002_curved_stop", "curved", dict(SCEN03, dt=0.02, max_curvature=1.0, max_road_width=1.5), [-20.0, 2.4, 0.0, 3.5, 0.1],
        target_speed=0.0, max_stop=9.0, static=np.array([[-8.0, 1.5], [-1.0, -1.0]]), prev_s=9.0,
        overrides=dict(max_accel=6.0))
    return cases


def build_random_cases(n, seed0=7000):
    """Random configurations drawn by the generators of the GPU fuzz test (tests/test_gpu_fuzz.py: random path, planner
    arguments, ego, overrides, stop directive, static points, single-sample / distribution tensors) -- so that the
    ORACLE, which that fuzz compares the library with, is itself pinned to the reference on configurations of the same
    kind, not only on the hand-made cases above."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fuzz as fz
    from oracle import oracle as orc                                       # (samples points along the path, nothing else)
    cases = []
    k = 0
    while len(cases) < n:
        rng = np.random.default_rng(seed0 + k)
        k += 1
        wx, wy = fz.random_path(rng)
        kw = fz.random_planner_kwargs(rng)
        fp = kw.pop("footprint", None)
        if fp is not None:                                                  # (redrawn in the reference's own terms)
            fp = dict(length=float(rng.uniform(3.5, 5)), width=float(rng.uniform(1.6, 2.1)), n=int(rng.integers(1, 6)))
        rq = fz.random_request(rng, orc.Spline(wx, wy), kw, dense=bool(rng.random() < 0.15))
        for arr in (rq.static, rq.dyn, rq.dist):
            if arr is not None and np.isnan(arr).any():
                np.nan_to_num(arr, copy=False, nan=1.0e3)                   # (NaN tracks have their own cases)
        cases.append(dict(name=f"rnd_{len(cases):03d}", path=dict(wx=[float(v) for v in wx], wy=[float(v) for v in wy]),
                          planner=kw, ego=[rq.x, rq.y, rq.yaw, rq.v, rq.a], target_speed=rq.target_speed,
                          overrides=rq.overrides, max_stop=rq.max_stop_distance, prev_s=rq.prev_s,
                          last_kappa=rq.last_kappa, footprint=fp,
                          static=np.empty((0, 2)) if rq.static is None else rq.static, dyn=rq.dyn, dist=rq.dist))
    return cases


def build_fuzz_cases(seed_inst):
    """Instances of the GPU fuzz test itself (tests/test_gpu_fuzz.py run_seed: seed -> path, planner arguments, six
    requests) put before the REFERENCE: the cases a sweep singled out -- `crawl_<seed>_<inst>`: the library's curvature
    differs from the oracle's by more than 1e-8 at a sample just above the EPS_S_DOT gate -- so that the reference says
    which side, if either, holds its value."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fuzz as fz
    from integrated_path_planning_amd.footprint import EgoFootprint
    from oracle import oracle as orc
    cases = []
    for seed, inst in seed_inst:
        rng = np.random.default_rng(1000 + seed)
        wx, wy = fz.random_path(rng)
        drawn = {}
        orig = EgoFootprint.multi_circle.__func__

        def recording(cls, length, width, n, _orig=orig, _d=drawn):
            _d.update(length=float(length), width=float(width), n=int(n))
            return _orig(cls, length, width, n)
        EgoFootprint.multi_circle = classmethod(recording)
        try:
            kw = fz.random_planner_kwargs(rng)
        finally:
            EgoFootprint.multi_circle = classmethod(orig)
        fp = dict(drawn) if kw.pop("footprint", None) is not None else None
        sp = orc.Spline(wx, wy)
        rq = None
        for _ in range(inst + 1):
            rq = fz.random_request(rng, sp, kw, False)
        for arr in (rq.static, rq.dyn, rq.dist):
            assert arr is None or not np.isnan(arr).any(), "a NaN track: not a case for this generator"
        cases.append(dict(name=f"crawl_{seed}_{inst}", path=dict(wx=[float(v) for v in wx], wy=[float(v) for v in wy]),
                          planner=kw, ego=[rq.x, rq.y, rq.yaw, rq.v, rq.a], target_speed=rq.target_speed,
                          overrides=rq.overrides, max_stop=rq.max_stop_distance, prev_s=rq.prev_s,
                          last_kappa=rq.last_kappa, footprint=fp,
                          static=np.empty((0, 2)) if rq.static is None else rq.static, dyn=rq.dyn, dist=rq.dist))
    return cases


# what the 36 000-seed sweep of round 4 flagged (profiles/r04_fuzz36000.log, gpurun_out/r04_crawl.txt)
CRAWL_CASES = [(17857, 3), (19368, 3)]


def run_case(ref, case, out_dir):
    FrenetPlanner, CubicSpline2D, EgoVehicleState, EgoFootprint = ref
    wx, wy = waypoints(case["path"])
    csp = CubicSpline2D(list(wx), list(wy))
    kw = dict(case["planner"])
    if case["footprint"]:
        f = case["footprint"]
        kw["footprint"] = EgoFootprint.multi_circle(f["length"], f["width"], f["n"])
    planner = FrenetPlanner(csp, **kw)
    planner._last_kappa = case["last_kappa"]
    if case["prev_s"] is not None:
        planner.converter._prev_s = case["prev_s"]
    ego = EgoVehicleState(*case["ego"])
    static, dyn, dist = case["static"], case["dyn"], case["dist"]

    out = {}
    out["static"] = np.asarray(static, dtype=np.float64)
    out["dyn"] = np.empty((0, 0, 2)) if dyn is None else np.asarray(dyn)
    out["dist"] = np.empty((0, 0, 0, 2)) if dist is None else np.asarray(dist)
    out["wx"], out["wy"] = np.asarray(wx, float), np.asarray(wy, float)
    out["sp_s"] = np.asarray(csp.s, float)
    for ax, sp1 in (("x", csp.sx), ("y", csp.sy)):
        for nm in "abcd":
            out[f"sp_{nm}{ax}"] = np.asarray(getattr(sp1, nm), float)

    # --- the stages of plan() (frenet_planner.py:259-304), run one by one to capture per-candidate data
    planner.last_check_stats = None
    fs = planner._cartesian_to_frenet_state(ego)
    if fs is None and case["name"].startswith(("rnd_", "crawl_")):
        print(case["name"], "skipped: the conversion to the Frenet frame fails")
        return
    assert fs is not None
    out["frenet0"] = np.array([fs.s, fs.s_d, fs.s_dd, fs.d, fs.d_d, fs.d_dd])
    out["prev_s_after"] = np.array(planner.converter._prev_s)
    rs = fs.s
    rx, ry = csp.calc_position(rs)
    out["ref0"] = np.array([rs, float(rx), float(ry), float(csp.calc_yaw(rs)), float(csp.calc_curvature(rs)),
                            float(csp.calc_curvature_rate(rs))])
    fp_list = planner._generate_frenet_paths(fs, case["target_speed"])
    nts = np.array([len(fp.t) for fp in fp_list], dtype=np.int32)
    fp_list = planner._calc_global_paths(fp_list)
    fp_dict = planner._check_paths(fp_list, static, dyn, case["overrides"], dist)
    if case["max_stop"] is not None:
        planner._apply_stop_distance_filter(fp_dict, case["max_stop"])
    status = {}
    for key, lst in fp_dict.items():
        for fp in lst:
            status[id(fp)] = STATUS_NAMES.index(key)
    out["cand_cost"] = np.array([fp.cost for fp in fp_list])
    out["cand_status"] = np.array([status.get(id(fp), ST_DROPPED) for fp in fp_list], dtype=np.int8)
    out["cand_keep"] = np.array([len(fp.x) for fp in fp_list], dtype=np.int32)
    out["cand_nt"] = nts
    out["stats"] = np.array([len(fp_dict[k]) if k in fp_dict else -1 for k in STATUS_NAMES], dtype=np.int32)
    best = planner._select_best_path(fp_dict)
    if best is not None:
        bi = [i for i, fp in enumerate(fp_list) if fp is best][0]
        out["best_index"] = np.array(bi)
        out["best_cost"] = np.array(best.cost)
        for f in FIELDS:
            out["best_" + f] = np.asarray(getattr(best, f), float)
        out["last_kappa_after"] = np.array(float(best.c[1]) if len(best.c) > 1 else case["last_kappa"])
    else:
        out["best_index"] = np.array(-1)
        out["best_cost"] = np.array(np.inf)
        out["last_kappa_after"] = np.array(case["last_kappa"])
    n = len(fp_list)
    probes = sorted(set(i for i in [0, 1, n // 3, n // 2, (2 * n) // 3, n - 8, n - 1,
                                    int(out["best_index"])] if 0 <= i < n))
    out["probe_idx"] = np.array(probes, dtype=np.int32)
    longest = max(len(fp_list[i].t) for i in probes)
    width = 64 if longest <= 64 else 128 if longest <= 128 else 256         # (64 / 128: the fixtures of rounds 1-3, unchanged)
    for f in FIELDS:
        arr = np.full((len(probes), width), np.nan)
        for r, i in enumerate(probes):
            v = np.asarray(getattr(fp_list[i], f), float)
            arr[r, : len(v)] = v
        out["probe_" + f] = arr

    # --- cross-check: the public plan() entry gives the same answer from a fresh planner
    p2 = FrenetPlanner(csp, **kw)
    p2._last_kappa = case["last_kappa"]
    if case["prev_s"] is not None:
        p2.converter._prev_s = case["prev_s"]
    res = p2.plan(ego, static, dyn, target_speed=case["target_speed"], constraint_overrides=case["overrides"],
                  dynamic_obstacles_distribution=dist, max_stop_distance=case["max_stop"])
    assert (res is None) == (best is None)
    if res is not None:
        assert res.cost == best.cost and np.array_equal(res.x, best.x)
    assert p2.last_check_stats == {k: len(v) for k, v in fp_dict.items()}

    meta = {k: case[k] for k in ("name", "path", "planner", "ego", "target_speed", "overrides", "max_stop",
                                 "prev_s", "last_kappa", "footprint")}
    if case["footprint"]:
        fpt = kw["footprint"]
        meta["footprint_offsets"] = [float(v) for v in fpt.offsets]
        meta["footprint_radius"] = float(fpt.radius)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out_dir, case["name"] + ".npz"), **out)
    ok = int((out["cand_status"] == 6).sum())
    print(f"{case['name']:22s} n_cand={n:5d} ok={ok:5d} best={int(out['best_index']):5d} "
          f"cost={float(out['best_cost']):.6f} stats={out['stats'].tolist()}")


def run_time_cache(ref, out_dir):
    """The reference's polynomial builders (_build_time_cache, _build_longitudinal_profiles, _build_lateral_profiles,
    frenet_planner.py:586-701) on a few horizons and one Frenet state: what the shim's views of them must return."""
    FrenetPlanner, CubicSpline2D, _Ego, _Fp = ref
    sys.path.insert(0, "/root/reference")
    from src.core.data_structures import FrenetState
    wx, wy = waypoints("straight")
    out = {}
    for tag, dt in (("dt01", 0.1), ("dt005", 0.05)):
        pl = FrenetPlanner(CubicSpline2D(wx, wy), dt=dt)
        fs = FrenetState(s=12.5, s_d=6.25, s_dd=-0.75, d=0.4, d_d=-0.3, d_dd=0.125)
        tvs, dis = np.array([8.0, 6.5, 0.0]), np.array([-3.5, 0.0, 1.5])
        for T in (0.5, 1.0, 4.0, 4.7, 5.0):
            tc = pl._build_time_cache(T)
            key = f"{tag}_T{T}"
            out[key + "_t"] = tc.t; out[key + "_t5"] = tc.t5
            out[key + "_qa"] = tc.quartic_A_inv; out[key + "_qi"] = tc.quintic_A_inv
            lon = pl._build_longitudinal_profiles(fs, tvs, T, tc)
            lat = pl._build_lateral_profiles(fs, dis, T, tc)
            out[key + "_lon"] = np.stack([np.stack([p.s, p.s_d, p.s_dd, p.s_ddd]) for p in lon])
            out[key + "_lat"] = np.stack([np.stack([p.d, p.d_d, p.d_dd, p.d_ddd]) for p in lat])
    out["state"] = np.array([12.5, 6.25, -0.75, 0.4, -0.3, 0.125])
    out["tvs"], out["dis"] = np.array([8.0, 6.5, 0.0]), np.array([-3.5, 0.0, 1.5])
    os.makedirs(os.path.join(out_dir, "builders"), exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "builders", "time_cache.npz"), **out)
    print("builders/time_cache.npz", len(out), "arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None)
    ap.add_argument("--random", type=int, default=40, help="with --only rnd: how many random cases")
    args = ap.parse_args()
    ref = import_reference(args.ref)
    if args.only == "time_cache":
        run_time_cache(ref, HERE)
        return
    if args.only == "crawl":                         # the fuzz instances a sweep flagged
        for case in build_fuzz_cases(CRAWL_CASES):
            run_case(ref, case, HERE)
        return
    if args.only == "rnd":                           # the random block alone (--random N of them)
        for case in build_random_cases(args.random):
            run_case(ref, case, HERE)
        return
    for case in build_cases():
        if args.only and args.only not in case["name"]:
            continue
        run_case(ref, case, HERE)


if __name__ == "__main__":
    main()
