"""the oracle's prediction resampling against reference-generated vectors (no GPU)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import oracle as orc


@pytest.fixture(scope="module")
def cases():
    z = np.load(os.path.join(GOLDEN_DIR, "prediction", "cases.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def test_process_prediction_and_cv(cases):
    for m in cases["meta"]["resample"]:
        c = m["case"]
        kw = dict(sgan_dt=m["sgan_dt"], sim_dt=m["sim_dt"], plan_horizon=m["plan_horizon"])
        anchor = cases[f"c{c}_anchor"] if m["with_anchor"] else None
        got = orc.process_prediction(cases[f"c{c}_pred"], anchor, m["staleness"], **kw)
        want = cases[f"c{c}_dense"]
        assert got.shape == want.shape, m
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, err_msg=str(m))
        cv = orc.predict_cv(_last_obs(cases, c, m), cases[f"c{c}_prev"], m["staleness"], pred_len=m["pred_len"], **kw)
        np.testing.assert_allclose(cv, cases[f"c{c}_cv"], rtol=1e-12, atol=1e-12, err_msg=str(m))
        cv1 = orc.predict_cv(_last_obs(cases, c, m), None, m["staleness"], pred_len=m["pred_len"], **kw)
        np.testing.assert_allclose(cv1, cases[f"c{c}_cv1"], rtol=1e-12, atol=1e-12, err_msg=str(m))


def _last_obs(cases, c, m):
    """last observation sample = prev + v*sgan_dt; recover it from the stored zero-velocity CV prediction"""
    return cases[f"c{c}_cv1"][:, 0, :]


def test_best_sample(cases):
    for m in cases["meta"]["select"]:
        best, dist = orc.best_sample(cases[f"s{m['case']}_samples"])
        assert best == m["best"], m


def test_predict_cv_float32_observations_follow_numpy_semantics():
    """The observer hands float32 tensors to predict_cv (observer.py:134): positions rounded to float32, the velocity
    (p_curr - p_prev) / sgan_dt evaluated in float32, the extrapolation in float64 (trajectory_predictor.py:203-228)."""
    rng = np.random.default_rng(11)
    obs = rng.normal(0, 20, (8, 13, 2))
    for stale in (0.0, 0.1, 0.30000000000000004):
        got = orc.predict_cv(obs[-1], obs[-2], stale, float32_observations=True)
        cur, prev = obs[-1].astype(np.float32), obs[-2].astype(np.float32)
        vel = (cur - prev) / 0.4                                   # float32 array / Python float -> float32
        assert vel.dtype == np.float32
        t = np.arange(0.1, 5.0 + 1e-9, 0.1) + stale
        want = cur[:, None, :] + vel[:, None, :] * t[None, :, None]   # float32 * float64 -> float64
        assert want.dtype == np.float64
        np.testing.assert_allclose(got, want, rtol=1e-15, atol=1e-13)
    f64 = orc.predict_cv(obs[-1], obs[-2], 0.1)
    assert np.abs(f64 - orc.predict_cv(obs[-1], obs[-2], 0.1, float32_observations=True)).max() > 1e-7
