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
