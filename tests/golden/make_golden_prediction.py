#!/usr/bin/env python3
"""Golden vectors for SURVEY 8(f1), the obstacle-tensor producer (build container only).

Runs the REFERENCE TrajectoryPredictor (method 'cv': no weights needed) read-only:
process_prediction on synthetic raw Social-GAN-shaped predictions, predict_cv, and
predict_single_best's closest-to-mean selection (predict() patched to replay prepared samples).
Writes tests/golden/prediction/cases.npz -- inputs and expected outputs only.
"""
import argparse
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    lg = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    lg.logger = _Logger()
    sys.modules["loguru"] = lg
    sys.path.insert(0, args.ref)
    import torch
    from src.prediction.trajectory_predictor import TrajectoryPredictor

    rng = np.random.default_rng(42)
    out = {}
    meta = []
    ci = 0
    for pred_len, plan_h, sgan_dt, sim_dt in [(12, 5.0, 0.4, 0.1), (12, 3.0, 0.4, 0.1), (8, 5.0, 0.4, 0.1),
                                               (12, 6.0, 0.4, 0.2), (12, 5.0, 0.4, 0.1)]:
        pr = TrajectoryPredictor(model_path=None, pred_len=pred_len, num_samples=1, sgan_dt=sgan_dt, sim_dt=sim_dt,
                                 plan_horizon=plan_h, method="cv")
        for staleness in (0.0, 0.1, 0.3):
            for with_anchor in (True, False):
                P = int(rng.integers(1, 9))
                p0 = rng.uniform(-20, 20, (P, 2))
                vel = rng.normal(0, 1.3, (P, 2))
                vel[0] = [4.0, -3.5]                                   # fast mover: tail clamp at 2.5 m/s
                steps = np.arange(1, pred_len + 1)[:, None, None] * sgan_dt
                pred = p0[None] + vel[None] * steps + np.cumsum(rng.normal(0, 0.08, (pred_len, P, 2)), axis=0)
                if P > 2:
                    pred[:, 1, :] = p0[1]                              # standing pedestrian: constant fill
                if P > 3:
                    pred[:, 2, 0] = 0.0                                # warm-up zeros on one axis
                    p0[2, 0] = 0.0
                if P > 4:
                    pred[:, 3, 1] = p0[3, 1] * (1 + 4e-6)              # inside np.allclose's rtol
                anchor = p0 if with_anchor else None
                dense = pr.process_prediction(pred.copy(), anchor_pos=None if anchor is None else anchor.copy(),
                                              staleness=staleness)
                out[f"c{ci}_pred"], out[f"c{ci}_dense"] = pred, dense
                out[f"c{ci}_anchor"] = p0 if with_anchor else np.empty((0, 2))
                # predict_cv from an observation window whose last two samples are p_prev, p0
                p_prev = p0 - vel * sgan_dt
                obs = torch.tensor(np.stack([p_prev, p0]), dtype=torch.float64)
                out[f"c{ci}_cv"] = pr.predict_cv(obs, staleness)
                out[f"c{ci}_cv1"] = pr.predict_cv(obs[-1:], staleness)
                out[f"c{ci}_prev"] = p_prev
                meta.append(dict(case=ci, pred_len=pred_len, plan_horizon=plan_h, sgan_dt=sgan_dt, sim_dt=sim_dt,
                                 staleness=staleness, with_anchor=with_anchor, P=P))
                ci += 1
    # closest-to-mean selection over S samples (predict_single_best :338-351)
    sel = []
    for j, (S, P, T) in enumerate([(20, 30, 50), (5, 3, 50), (2, 1, 30), (20, 14, 50)]):
        pr = TrajectoryPredictor(model_path=None, pred_len=12, num_samples=S, method="cv")
        base = rng.uniform(-10, 10, (P, 1, 2)) + np.cumsum(rng.normal(0.1, 0.05, (P, T, 2)), axis=1)
        samples = base[None] + np.cumsum(rng.normal(0, 0.05, (S, P, T, 2)), axis=2)
        it = iter(list(samples))
        pr.predict = lambda *a, **k: next(it)
        best, dist = pr.predict_single_best(None, None, None, staleness=0.0)
        bi = [i for i in range(S) if np.array_equal(samples[i], best)][0]
        out[f"s{j}_samples"] = samples
        sel.append(dict(case=j, best=bi))
    out["meta"] = np.array(json.dumps(dict(resample=meta, select=sel)))
    os.makedirs(os.path.join(HERE, "prediction"), exist_ok=True)
    path = os.path.join(HERE, "prediction", "cases.npz")
    np.savez_compressed(path, **out)
    print(f"{ci} resample cases, {len(sel)} selection cases, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
