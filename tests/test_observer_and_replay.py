"""The pedestrian observer and the replay source of the closed loop (``closed_loop.Observer`` / ``ReplayPedestrians``)
against the reference's own tests of ``PedestrianObserver`` (tests/test_observer.py:29-110) and
``ReplayPedestrianSource`` (tests/test_replay_source.py:19-78): same scenarios and numbers, this build's classes."""
import numpy as np
import pytest

from integrated_path_planning_amd.closed_loop import Observer, ReplayPedestrians


def _pos(t, speed=1.2):
    return np.array([[speed * t, 0.0]])                                 # one pedestrian walking along +x


def _drive(obs, dt, n_steps, t0=0.0):
    for k in range(1, n_steps + 1):
        obs.update(_pos(t0 + k * dt), t0 + k * dt)


def test_samples_at_exact_sgan_dt_intervals():
    obs = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    _drive(obs, 0.1, 200)
    intervals = np.diff(np.array(obs.timestamps))
    assert len(intervals) == 7
    np.testing.assert_allclose(intervals, 0.4, atol=1e-9)


def test_sample_timestamps_on_sgan_grid():
    obs = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    sampled = []
    for k in range(1, 33):
        before = len(obs.history)
        obs.update(_pos(k * 0.1), k * 0.1)
        if len(obs.history) > before:
            sampled.append(k * 0.1)
    np.testing.assert_allclose(sampled, [0.4, 0.8, 1.2, 1.6, 2.0, 2.4, 2.8, 3.2], atol=1e-9)


def test_ready_after_warmup_step_count():
    obs = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    warmup = int(8 * 0.4 / 0.1)
    _drive(obs, 0.1, warmup - 1)
    assert not obs.is_ready
    obs.update(_pos(warmup * 0.1), warmup * 0.1)
    assert obs.is_ready


def test_apparent_velocity_matches_true_speed():
    obs = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    _drive(obs, 0.1, 100)
    traj = np.stack(list(obs.history), axis=0)
    np.testing.assert_allclose(np.linalg.norm(np.diff(traj[:, 0, :], axis=0), axis=1) / 0.4, 1.2, atol=1e-6)


def test_no_float_drift_over_long_run():
    obs = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    sampled = []
    for k in range(1, 1001):
        before = len(obs.history)
        obs.update(_pos(k * 0.1), k * 0.1)
        if len(obs.history) > before or (len(obs.history) == 8 and obs.timestamps[-1] == k * 0.1):
            sampled.append(k * 0.1)
    np.testing.assert_allclose(np.diff(sorted(set(sampled))), 0.4, atol=1e-9)


def test_dt_equal_to_sgan_dt_samples_every_step():
    obs = Observer(obs_len=8, dt=0.4, sgan_dt=0.4)
    _drive(obs, 0.4, 8)
    assert obs.is_ready
    np.testing.assert_allclose(np.diff(np.array(obs.timestamps)), 0.4, atol=1e-9)


def test_nonzero_start_time():
    obs = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    _drive(obs, 0.1, 100, t0=3.2)
    np.testing.assert_allclose(np.diff(np.array(obs.timestamps)), 0.4, atol=1e-9)


def test_reset_clears_accumulator_and_reference_time():
    obs = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    _drive(obs, 0.1, 10)
    obs.reset()
    assert len(obs.history) == 0 and obs.accumulated_time == 0.0 and obs._last_update_timestamp is None
    _drive(obs, 0.1, 32, t0=5.0)
    assert obs.is_ready
    np.testing.assert_allclose(np.diff(np.array(obs.timestamps)), 0.4, atol=1e-9)


def test_the_observer_keeps_copies():
    """(a frame handed to update() may be overwritten by the caller afterwards: the loop's frames are views)"""
    obs = Observer(obs_len=2, dt=0.4, sgan_dt=0.4)
    frame = np.array([[1.0, 2.0]])
    obs.update(frame, 0.4)
    frame[:] = 99.0
    np.testing.assert_array_equal(obs.history[-1], [[1.0, 2.0]])


# ---- replay source

def _traj():
    return np.array([[[0.0, 0.0], [5.0, 5.0]], [[1.0, 0.0], [5.0, 6.0]], [[2.0, 0.0], [5.0, 7.0]]])


def test_get_state_returns_current_frame():
    s0 = ReplayPedestrians(_traj(), dt=0.4).get_state()
    assert s0.n_peds == 2
    np.testing.assert_array_equal(s0.ids, [0, 1])                  # (replay_source.py:71-73, 98-106)
    np.testing.assert_allclose(s0.positions, [[0, 0], [5, 5]])
    assert s0.timestamp == pytest.approx(0.0)


def test_step_advances_frame_and_time():
    src = ReplayPedestrians(_traj(), dt=0.4)
    src.step()
    s1 = src.get_state()
    np.testing.assert_allclose(s1.positions, [[1, 0], [5, 6]])
    assert s1.timestamp == pytest.approx(0.4)


def test_step_clamps_position_but_time_advances():
    src = ReplayPedestrians(_traj(), dt=0.4)
    src.step(n=10)
    s = src.get_state()
    np.testing.assert_allclose(s.positions, [[2, 0], [5, 7]])
    assert s.timestamp == pytest.approx(10 * 0.4)


def test_velocities_finite_difference():
    s0 = ReplayPedestrians(_traj(), dt=0.4).get_state()
    np.testing.assert_allclose(s0.velocities, [[2.5, 0.0], [0.0, 2.5]])


def test_goals_default_to_final_position():
    np.testing.assert_allclose(ReplayPedestrians(_traj(), dt=0.4).get_state().goals, [[2, 0], [5, 7]])


def test_ego_state_is_ignored():
    src = ReplayPedestrians(_traj(), dt=0.4)
    src.step(ego_state=object())
    np.testing.assert_allclose(src.get_state().positions, [[1, 0], [5, 6]])


def test_rejects_bad_shape():
    with pytest.raises(ValueError):
        ReplayPedestrians(np.zeros((3, 2)), dt=0.4)


def test_supplied_velocities_and_reset():
    traj = _traj()
    src = ReplayPedestrians(traj, dt=0.4, velocities=np.ones_like(traj))
    np.testing.assert_allclose(src.get_state().velocities, np.ones((2, 2)))
    src.step(n=2)
    src.reset()
    assert src.get_state().timestamp == pytest.approx(0.0)
    np.testing.assert_allclose(src.get_state().positions, [[0, 0], [5, 5]])
