"""The reference's time-anchor tests of the predictor (tests/test_prediction_anchor.py:30-155), restated against the
device resampler (``PredictionResampler`` -> fot_resample_predictions / fot_predict_cv): for a constant-velocity
pedestrian the dense prediction is the true future position at every observation phase (staleness 0 .. 0.3 s)."""
import numpy as np
import pytest

from integrated_path_planning_amd import synthetic as syn
from integrated_path_planning_amd.closed_loop import Observer
from integrated_path_planning_amd.planner import BatchPlanner
from integrated_path_planning_amd.prediction import PredictionResampler

pytestmark = pytest.mark.gpu

SIM_DT, SGAN_DT, PLAN_HORIZON = 0.1, 0.4, 5.0


@pytest.fixture(scope="module")
def predictor():
    engine = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), **syn.CONFIG3_PLANNER)
    return PredictionResampler(engine, pred_len=12, sgan_dt=SGAN_DT, sim_dt=SIM_DT, plan_horizon=PLAN_HORIZON)


def _run(predictor, staleness):
    p0, v = np.array([2.0, -1.0]), np.array([1.2, 0.5])
    pred = np.stack([p0 + v * (k * SGAN_DT) for k in range(1, 13)], axis=0)[:, None, :]       # (12, 1, 2)
    return predictor.process_prediction(pred, anchor_pos=p0[None, :], staleness=staleness), p0, v


def test_reanchored_grid_matches_true_future(predictor):                     # :47-63
    for j in range(4):
        staleness = j * SIM_DT
        dense, p0, v = _run(predictor, staleness)
        support_end = 12 * SGAN_DT - staleness
        for k in range(dense.shape[1]):
            t = (k + 1) * SIM_DT
            if t > support_end:
                break
            np.testing.assert_allclose(dense[0, k], p0 + v * (t + staleness), atol=1e-9, err_msg=f"staleness={staleness}, k={k}")


def test_no_left_clamp_with_anchor(predictor):                               # :65-71
    dense, p0, v = _run(predictor, 0.0)
    np.testing.assert_allclose(dense[0, 0], p0 + v * 0.1, atol=1e-9)
    np.testing.assert_allclose(dense[0, 2], p0 + v * 0.3, atol=1e-9)


def test_tail_extrapolation_continues_velocity(predictor):                   # :73-81
    dense, p0, v = _run(predictor, 0.3)
    k_last = dense.shape[1] - 1
    np.testing.assert_allclose(dense[0, k_last], p0 + v * ((k_last + 1) * SIM_DT + 0.3), atol=1e-9)


def test_zero_staleness_no_anchor_backward_compatible(predictor):            # :83-94
    p0, v = np.array([0.0, 0.0]), np.array([1.0, 0.0])
    pred = np.stack([p0 + v * (k * SGAN_DT) for k in range(1, 13)], axis=0)[:, None, :]
    dense = predictor.process_prediction(pred)
    np.testing.assert_allclose(dense[0, 3], p0 + v * 0.4, atol=1e-9)


def test_cv_origin_shifted_by_staleness(predictor):                          # :97-113
    obs = np.stack([np.array([[0.0, 0.0]]), np.array([[0.48, 0.0]])], axis=0)                  # (2, 1, 2), v = 1.2 m/s
    for j in range(4):
        staleness = j * SIM_DT
        dense = predictor.predict_cv(obs, staleness=staleness, float32_observations=True)
        for k in (0, 9, 49):
            t = (k + 1) * SIM_DT
            expected = np.float64(np.float32(0.48)) + np.float64(np.float32(0.48) / np.float32(0.4)) * (t + staleness)
            np.testing.assert_allclose(dense[0, k], [expected, 0.0], atol=1e-9, err_msg=f"staleness={staleness}, k={k}")
            np.testing.assert_allclose(dense[0, k], [0.48 + 1.2 * (t + staleness), 0.0], atol=1e-5)   # the reference's own bound


def test_dense_prediction_matches_truth_at_all_phases(predictor):            # :122-155
    speed = np.array([1.2, -0.4])
    observer = Observer(obs_len=8, dt=SIM_DT, sgan_dt=SGAN_DT)
    pos = lambda t: np.array([[speed[0] * t, speed[1] * t]])
    t = 0.0
    for _ in range(32):                                                      # warm-up: 32 updates -> 8 samples
        t = round(t + SIM_DT, 9)
        observer.update(pos(t), t)
    assert observer.is_ready
    for _ in range(8):                                                       # both sampling phases j = 0..3, twice
        t = round(t + SIM_DT, 9)
        observer.update(pos(t), t)
        hist = np.stack(list(observer.history), axis=0)
        staleness = t - observer.last_sample_time
        dense = predictor.predict_cv(hist, staleness=staleness, float32_observations=True)
        for k in (0, 3, 19, 39):
            np.testing.assert_allclose(dense[0, k], pos(t + (k + 1) * SIM_DT)[0], atol=1e-4,
                                       err_msg=f"t={t:.1f}, staleness={staleness:.1f}, k={k}")
