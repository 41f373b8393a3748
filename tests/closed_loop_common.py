"""Shared by the CPU and GPU closed-loop episode tests: fixture access, an oracle-backed stand-in engine
(test infrastructure: lets the host logic of BatchedClosedLoop run where there is no GPU) and the comparison."""
import json
import os
from types import SimpleNamespace

import numpy as np

from conftest import GOLDEN_DIR
from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.data_structures import FrenetPath
from oracle import oracle as orc

STATE_NAMES = {0: "NORMAL", 1: "CAUTION", 2: "EMERGENCY"}


def load_episodes():
    z = np.load(os.path.join(GOLDEN_DIR, "closed_loop", "reference_cv_episodes.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def load_dist_episodes():
    """distribution-aware episodes (tests/golden/make_closed_loop_distribution.py)"""
    z = np.load(os.path.join(GOLDEN_DIR, "closed_loop", "reference_dist_episodes.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def scenario_config(meta, name="base"):
    return dict(meta["variants"][name]["config"])


class OracleEngine:
    """BatchPlanner's methods on top of the CPU oracle -- for tests only."""

    def __init__(self, cfg):
        c = SimpleNamespace(**cfg)
        fp_kw = {}
        if cfg.get("ego_footprint", "circle") == "multi_circle":          # footprint_from_config (src/core/footprint.py)
            from integrated_path_planning_amd.footprint import EgoFootprint
            fp = EgoFootprint.multi_circle(cfg.get("vehicle_length", 4.5), cfg.get("vehicle_width", 2.0),
                                           int(cfg.get("ego_footprint_n_circles", 3)))
            fp_kw = dict(footprint_offsets=list(fp.offsets), footprint_radius=fp.radius)
        self.params = orc.make_params(
            **fp_kw,
            max_speed=c.ego_max_speed, max_accel=c.ego_max_accel, max_curvature=c.ego_max_curvature,
            max_lat_accel=cfg.get("ego_max_lat_accel", 3.0), dt=c.dt, d_road_w=c.d_road_w, max_road_width=c.max_road_width,
            robot_radius=cfg.get("ego_radius", 1.0), obstacle_radius=c.obstacle_radius, min_t=cfg.get("min_t", 4.0),
            max_t=cfg.get("max_t", 5.0), d_t_s=cfg.get("d_t_s", 5.0 / 3.6), k_j=c.k_j, k_t=c.k_t, k_d=c.k_d,
            k_s_dot=c.k_s_dot, k_lat=c.k_lat, k_lon=c.k_lon, chance_epsilon=cfg.get("chance_epsilon", 0.0),
            collision_margin_inflation=cfg.get("collision_margin_inflation", 1.0))
        self.sp = orc.Spline(np.asarray(c.reference_waypoints_x, float), np.asarray(c.reference_waypoints_y, float))
        self.calls = 0

    def path_coeffs(self):
        return self.sp.coeffs()

    def plan_batch(self, reqs):
        outs = []
        prev = None
        for r in reqs:
            ps = prev if r.chain_prev_s else r.prev_s
            ego = orc.make_ego(r.x, r.y, r.yaw, r.v, r.a, last_kappa=r.last_kappa, prev_s=ps)
            o = orc.plan(self.params, self.sp, ego, r.target_speed, r.overrides, r.max_stop_distance,
                         static=r.static, dyn=r.dyn, dist=r.dist)
            prev = o.new_prev_s
            outs.append(o)
            self.calls += 1
        recs = [SimpleNamespace(new_prev_s=o.new_prev_s, new_last_kappa=o.new_last_kappa, status=o.status) for o in outs]

        def path(j):
            o = outs[j]
            if o.status != 0:
                return None
            fp = FrenetPath(**{f: list(o.path[f]) for f in _abi.PATH_FIELDS})
            fp.cost = float(o.cost)
            return fp

        return SimpleNamespace(records=recs, stats=lambda j: outs[j].stats, path=path, outs=outs)

    # -- the array interfaces BatchedClosedLoop drives (planner.BatchPlanner.plan_arrays / safety_metrics_cat /
    #    nearest_s_arrays), on top of the methods above
    RESULT_DT = np.dtype(_abi.Result)
    EGO_DT = np.dtype(_abi.Ego)

    def plan_arrays(self, ego, target_speed, overrides, max_stop, static_xy, static_off, dyn_xy, dyn_off, dyn_dims):
        from integrated_path_planning_amd.batch import PlanRequest
        n = len(ego)
        keys = ("max_speed", "max_accel", "max_curvature", "max_lat_accel")
        reqs = []
        for i in range(n):
            ov = {k: float(overrides[i, j]) for j, k in enumerate(keys) if not np.isnan(overrides[i, j])} or None
            st = None
            if static_xy is not None:
                st = np.asarray(static_xy)[int(static_off[i]):int(static_off[i + 1])]
            dyn, dist = None, None
            if dyn_xy is not None and dyn_dims[i][0] == 1:
                P, T = int(dyn_dims[i][2]), int(dyn_dims[i][3])
                dyn = np.asarray(dyn_xy)[int(dyn_off[i]):int(dyn_off[i]) + P * T].reshape(P, T, 2)
            if dyn_xy is not None and dyn_dims[i][0] == 2:
                S, P, T = int(dyn_dims[i][1]), int(dyn_dims[i][2]), int(dyn_dims[i][3])
                dist = np.asarray(dyn_xy)[int(dyn_off[i]):int(dyn_off[i]) + S * P * T].reshape(S, P, T, 2)
            e = ego[i]
            reqs.append(PlanRequest(x=float(e["x"]), y=float(e["y"]), yaw=float(e["yaw"]), v=float(e["v"]), a=float(e["a"]),
                                    target_speed=float(target_speed[i]), last_kappa=float(e["last_kappa"]),
                                    prev_s=float(e["prev_s"]) if e["has_prev_s"] == 1 else None,
                                    chain_prev_s=bool(e["has_prev_s"] == 2), overrides=ov,
                                    max_stop_distance=None if np.isnan(max_stop[i]) else float(max_stop[i]),
                                    static=st, dyn=dyn, dist=dist))
        res = self.plan_batch(reqs)
        out = np.zeros(n, dtype=self.RESULT_DT)
        for i in range(n):
            o = res.outs[i]
            out["status"][i], out["best_index"][i], out["n_cand"][i] = o.status, o.best_index, o.n_cand
            out["new_prev_s"][i], out["new_last_kappa"][i] = o.new_prev_s, o.new_last_kappa
            out["cost"][i] = o.cost if o.status == 0 else np.inf
            if o.stats is not None:
                out["stats_valid"][i] = 1
                out["stats"][i] = [o.stats.get(k, 0) for k in _abi.STATUS_NAMES]
            if o.status == 0:
                k = len(o.path["x"])
                out["n_keep"][i] = k
                for f in _abi.PATH_FIELDS:
                    out[f][i, :k] = o.path[f]
        return out

    def safety_metrics_cat(self, egos, ped_off, ped_pos, ped_vel, ego_radius, ped_radius, use_footprint=True):
        pos = [ped_pos[int(ped_off[i]):int(ped_off[i + 1])] for i in range(len(egos))]
        vel = [ped_vel[int(ped_off[i]):int(ped_off[i + 1])] for i in range(len(egos))]
        return self.safety_metrics(egos, pos, vel, ego_radius, ped_radius, use_footprint)

    def nearest_s_arrays(self, x, y, yaw, v, a, prev_s):
        from integrated_path_planning_amd.batch import PlanRequest
        return self.frenet_states([PlanRequest(x[i], y[i], yaw[i], v[i], a[i],
                                               prev_s=None if np.isnan(prev_s[i]) else float(prev_s[i]))
                                   for i in range(len(x))])[2]

    def safety_metrics(self, egos, pos, vel, ego_radius, ped_radius, use_footprint=True):
        out = np.zeros(len(egos), dtype=[("min_distance", "f8"), ("ttc", "f8"), ("clearance", "f8"),
                                         ("clearance_ahead", "f8"), ("collision", "i4")])
        for i, e in enumerate(egos):
            m = orc.safety_metrics(self.params, ego_radius, ped_radius, e, pos[i], vel[i])
            out[i] = (m["min_distance"], m["ttc"], m["clearance"], m["clearance_ahead"], int(m["collision"]))
        return out

    def frenet_states(self, reqs):
        nps = []
        for r in reqs:
            ego = orc.make_ego(r.x, r.y, r.yaw, r.v, r.a, last_kappa=0.0, prev_s=r.prev_s)
            nps.append(orc.cartesian_to_frenet_state(self.sp, ego)[3])
        return None, None, np.array(nps), None


class OracleResampler:
    def __init__(self, cfg):
        self.kw = dict(pred_len=cfg["pred_len"], sgan_dt=0.4, sim_dt=cfg["dt"], plan_horizon=cfg.get("max_t", 5.0))

    def predict_cv(self, obs_traj, staleness=0.0, current=None, float32_observations=False):
        obs = np.asarray(obs_traj)
        return orc.predict_cv(obs[-1], obs[-2] if obs.shape[0] >= 2 else None, staleness,
                              float32_observations=float32_observations, **self.kw)


    def process_prediction(self, pred_traj, anchor_pos=None, staleness=0.0):
        pred = np.asarray(pred_traj, dtype=np.float64)
        kw = {k: v for k, v in self.kw.items() if k != "pred_len"}
        return np.stack([orc.process_prediction(p, anchor_pos, staleness, **kw) for p in pred])


def scripted_raw_sample(obs_last, obs_prev, k, n_samples, pred_len, sgan_dt):
    """Sample k of a scripted multi-sample predictor: [pred_len, P, 2] positions at the predictor's own step.

    Stands in for ONE Social-GAN forward pass in the reference-generated fixture of distribution-aware episodes (the
    network weights are not available offline) and in the tests that replay it: every pedestrian keeps walking with
    its last observed velocity, turned by a sample-dependent angle and scaled by a sample-dependent factor.  Plain
    NumPy float64 on both sides: the reference run calls it from its predictor's predict(), the tests hand it to
    BatchedClosedLoop as sample_source."""
    obs_last, obs_prev = np.asarray(obs_last, np.float64), np.asarray(obs_prev, np.float64)
    vel = (obs_last - obs_prev) / sgan_dt
    ang = 0.08 * (k - 0.5 * (n_samples - 1))
    gain = 1.0 + 0.06 * ((k % 3) - 1)
    c, s_ = np.cos(ang) * gain, np.sin(ang) * gain
    v = np.stack([c * vel[:, 0] - s_ * vel[:, 1], s_ * vel[:, 0] + c * vel[:, 1]], axis=1)
    steps = (np.arange(pred_len) + 1.0) * sgan_dt
    return obs_last[None, :, :] + steps[:, None, None] * v[None, :, :]


def scripted_sample_source(n_samples, pred_len, sgan_dt=0.4):
    return lambda obs_last, obs_prev: np.stack([scripted_raw_sample(obs_last, obs_prev, k, n_samples, pred_len, sgan_dt)
                                                for k in range(n_samples)])


def assert_episode_matches(hist, termination, ep, name, tol=1e-6):
    """hist: List[StepRecord] of one episode; ep: fixture dict; name: variant prefix."""
    meta = ep["meta"]["variants"][name]
    pre = name + "_"
    assert termination == meta["termination"], f"{name}: ended with {termination}, reference {meta['termination']}"
    assert len(hist) == meta["steps"], f"{name}: {len(hist)} steps, reference {meta['steps']}"
    ego = np.array([[r.ego.x, r.ego.y, r.ego.yaw, r.ego.v, r.ego.a, r.ego.jerk] for r in hist])
    want = ep[pre + "ego"]
    for col, f in enumerate(("x", "y", "yaw", "v", "a", "jerk")):
        np.testing.assert_allclose(ego[:, col], want[:, col], rtol=tol, atol=tol * (100 if f == "jerk" else 1),
                                   err_msg=f"{name} ego {f}")
    np.testing.assert_allclose([r.time for r in hist], ep[pre + "times"], atol=1e-9)
    states = [r.ego.state.name for r in hist]
    assert states == [STATE_NAMES[int(s)] for s in ep[pre + "state"]], f"{name} state sequence"
    m = np.array([[r.metrics["min_distance"], r.metrics["ttc"], r.metrics["clearance"], r.metrics["clearance_ahead"],
                   float(r.metrics["collision"]), r.metrics.get("n_collision_rejected", -1)] for r in hist])
    np.testing.assert_allclose(m[:, :4], ep[pre + "metrics"][:, :4], rtol=tol, atol=tol, err_msg=f"{name} metrics")
    np.testing.assert_array_equal(m[:, 4:], ep[pre + "metrics"][:, 4:], err_msg=f"{name} collision / rejected counts")
    plen = np.array([len(r.planned_path.x) if r.planned_path is not None else 0 for r in hist])
    np.testing.assert_array_equal(plen, ep[pre + "planned_len"], err_msg=f"{name} planned path lengths")
    cost = np.array([r.planned_path.cost if r.planned_path is not None else np.inf for r in hist])
    np.testing.assert_allclose(cost, ep[pre + "planned_cost"], rtol=tol, err_msg=f"{name} planned cost")
    for i, r in enumerate(hist):
        if r.planned_path is not None:
            np.testing.assert_allclose(r.planned_path.x, ep[pre + "planned_x"][i, :plen[i]], atol=tol, err_msg=f"{name} step {i}")
            np.testing.assert_allclose(r.planned_path.y, ep[pre + "planned_y"][i, :plen[i]], atol=tol, err_msg=f"{name} step {i}")
        shape = list(r.predicted_trajectories.shape) if r.predicted_trajectories is not None else [0, 0, 0]
        assert shape == list(ep[pre + "pred_shape"][i]), f"{name} step {i} prediction shape"
        if r.predicted_trajectories is not None:
            np.testing.assert_allclose(r.predicted_trajectories[0, :3].ravel(), ep[pre + "pred_first"][i], rtol=1e-12,
                                       atol=1e-12, err_msg=f"{name} step {i} prediction values")


def assert_npz_layout(arrays, meta_keys, n_steps):
    """trajectory.npz: the reference's keys, dtypes and shapes (integrated_simulator.py:959-982)."""
    assert set(arrays) == set(meta_keys)
    for k, (dtype, shape) in meta_keys.items():
        a = arrays[k]
        assert str(a.dtype) == dtype, f"{k}: dtype {a.dtype}, reference {dtype}"
        assert list(a.shape) == shape, f"{k}: shape {a.shape}, reference {shape}"
    assert len(arrays["times"]) == n_steps
