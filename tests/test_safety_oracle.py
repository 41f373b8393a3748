"""the oracle's safety metrics against reference-generated vectors (no GPU)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import oracle as orc


def load_cases():
    z = np.load(os.path.join(GOLDEN_DIR, "safety", "cases.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def oracle_params(meta):
    fp = meta["footprint"]
    if fp is None:
        return orc.make_params()
    return orc.make_params(footprint_offsets=fp["offsets"], footprint_radius=fp["radius"])


KEYS = ("min_distance", "collision", "ttc", "clearance", "clearance_ahead")


def test_oracle_matches_reference_metrics():
    cases = load_cases()
    assert len(cases["meta"]) == 70
    for i, m in enumerate(cases["meta"]):
        got = orc.safety_metrics(oracle_params(m), m["ego_radius"], m["ped_radius"], cases[f"c{i}_ego"],
                                 cases[f"c{i}_pos"], cases[f"c{i}_vel"])
        want = cases[f"c{i}_want"]
        assert got["collision"] == bool(want[1]), i
        for k, j in (("min_distance", 0), ("ttc", 2), ("clearance", 3), ("clearance_ahead", 4)):
            np.testing.assert_allclose(got[k], want[j], rtol=1e-12, atol=1e-12, err_msg=f"case {i} {k}")


def test_reference_test_values():
    """The literal expectations of the reference's tests/test_footprint.py:52-102 and test_smooth_braking.py:137-163."""
    cases = load_cases()
    w = [cases[f"c{i}_want"] for i in range(10)]
    assert w[0][0] == pytest.approx(3.0) and not w[0][1] and w[0][2] == pytest.approx(1.8 / 6.0)
    assert not w[1][1] and w[2][1] and w[2][0] == pytest.approx(0.7)
    assert not w[3][1] and not w[4][1]
    assert np.isinf(w[5][0]) and np.isinf(w[5][2]) and not w[5][1]
    assert w[7][3] == pytest.approx(0.3) and w[7][4] == pytest.approx(1.8)
    assert np.isinf(w[8][4]) and np.isfinite(w[8][3])
    assert w[9][4] == pytest.approx(0.8)
