"""SURVEY 8(f3) on the GPU: k_safety against the reference's vectors, single calls and one batched launch."""
import numpy as np
import pytest

from integrated_path_planning_amd.data_structures import EgoVehicleState
from integrated_path_planning_amd.footprint import EgoFootprint
from integrated_path_planning_amd.safety import SafetyMonitor, compute_safety_metrics_static
from oracle import oracle as orc
from test_safety_oracle import load_cases, oracle_params

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-12, atol=1e-12)


class _Peds:
    def __init__(self, pos, vel):
        self.positions, self.velocities = pos, vel


def _check(got, want, label):
    assert bool(got["collision"]) == bool(want[1]), label
    for k, j in (("min_distance", 0), ("ttc", 2), ("clearance", 3), ("clearance_ahead", 4)):
        np.testing.assert_allclose(float(got[k]), want[j], err_msg=f"{label} {k}", **TOL)


def test_drop_in_function_matches_reference():
    cases = load_cases()
    for i, m in enumerate(cases["meta"]):
        e = cases[f"c{i}_ego"]
        fp = m["footprint"]
        foot = None if fp is None else EgoFootprint.multi_circle(fp["length"], fp["width"], fp["n"])
        got = compute_safety_metrics_static(EgoVehicleState(x=e[0], y=e[1], yaw=e[2], v=e[3], a=0.0),
                                            _Peds(cases[f"c{i}_pos"], cases[f"c{i}_vel"]), m["ego_radius"],
                                            m["ped_radius"], footprint=foot)
        assert set(got) == {"min_distance", "collision", "ttc", "clearance", "clearance_ahead"}
        assert isinstance(got["collision"], bool)
        _check(got, cases[f"c{i}_want"], f"case {i}")


def test_batched_launch_matches_single_calls_and_oracle():
    cases = load_cases()
    idx = [i for i, m in enumerate(cases["meta"]) if m["footprint"] is None]
    mon = SafetyMonitor()
    got = mon.metrics_batch([cases[f"c{i}_ego"] for i in idx], [cases[f"c{i}_pos"] for i in idx],
                            [cases[f"c{i}_vel"] for i in idx], 1.1, 0.25)
    assert len(got) == len(idx)
    for r, i in zip(got, idx):
        want = orc.safety_metrics(orc.make_params(), 1.1, 0.25, cases[f"c{i}_ego"], cases[f"c{i}_pos"], cases[f"c{i}_vel"])
        _check(r, [want["min_distance"], want["collision"], want["ttc"], want["clearance"], want["clearance_ahead"]],
               f"batched {i}")


def test_large_ragged_batch_against_oracle():
    rng = np.random.default_rng(3)
    foot = EgoFootprint.multi_circle(4.6, 1.9, 4)
    params = orc.make_params(footprint_offsets=list(foot.offsets), footprint_radius=foot.radius)
    mon = SafetyMonitor(foot)
    n = 300
    egos = np.column_stack([rng.normal(0, 30, n), rng.normal(0, 30, n), rng.uniform(-np.pi, np.pi, n), rng.uniform(0, 10, n)])
    counts = rng.integers(0, 400, n)
    counts[:4] = (0, 1, 64, 65)
    pos = [egos[i, :2] + rng.normal(0, 12, (c, 2)) for i, c in enumerate(counts)]
    vel = [rng.normal(0, 1.2, (c, 2)) for c in counts]
    got = mon.metrics_batch(egos, pos, vel, 1.0, 0.3)
    for i in range(n):
        want = orc.safety_metrics(params, 1.0, 0.3, egos[i], pos[i], vel[i])
        _check(got[i], [want["min_distance"], want["collision"], want["ttc"], want["clearance"], want["clearance_ahead"]],
               f"ego {i} P {counts[i]}")


def test_empty_batch_and_bad_arguments():
    mon = SafetyMonitor()
    assert len(mon.metrics_batch(np.empty((0, 4)), [], [], 1.0, 0.2)) == 0
    with pytest.raises(ValueError):
        mon.metrics_batch([[0, 0, 0, 1]], [np.zeros((2, 2))], [np.zeros((3, 2))], 1.0, 0.2)
