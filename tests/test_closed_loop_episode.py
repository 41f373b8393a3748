"""SURVEY 8(f4) without a GPU: the host logic of BatchedClosedLoop (pedestrian replay, observer, prepend, planning cycle,
ego update, emergency stop, termination, trajectory.npz layout) free-running against whole episodes of the reference
simulator, with oracle-backed stand-ins for the libfot calls."""
import numpy as np
import pytest

from closed_loop_common import (OracleEngine, OracleResampler, assert_episode_matches, assert_npz_layout, load_dist_episodes,
                                load_episodes, scenario_config, scripted_sample_source)
from integrated_path_planning_amd.closed_loop import BatchedClosedLoop, Observer, ReplayPedestrians


@pytest.fixture(scope="module")
def episodes():
    return load_episodes()


@pytest.mark.parametrize("name", ["fast", "shift", "base", "walls", "turn", "footprint", "inflate",
                                  "rnd0", "rnd1", "rnd2", "rnd3", "rnd4", "rnd5"])
def test_episodes_free_running(episodes, name):
    """scenario_01 in three pedestrian scripts, scenario_02 (static obstacle rectangles), scenario_03 (curved path),
    scenario_01 with the three-circle ego footprint (planner geometry and safety metrics) and with the planner's dynamic
    margin inflated by 1.2 (39 EMERGENCY steps, 34 steps without a path); six random pedestrian scripts on the three
    scenarios (796 steps together, 39 of them without a path)."""
    cfg = scenario_config(episodes["meta"], name)
    sim = BatchedClosedLoop(cfg, [episodes[name + "_ped_traj"]], engine=OracleEngine(cfg), resampler=OracleResampler(cfg))
    hist = sim.run()[0]
    assert_episode_matches(hist, sim.episodes[0].termination_reason, episodes, name)
    arrays = sim.trajectory_arrays(hist)
    assert_npz_layout(arrays, episodes["meta"]["variants"][name]["npz_keys"], len(hist))


@pytest.mark.parametrize("name", ["s4_eps0", "s5_best_only"])
def test_distribution_episodes_free_running(name):
    """A multi-sample predictor in front of the planner (scripted: closed_loop_common.scripted_raw_sample, the one the
    reference-generated fixture used in the place of a Social-GAN forward pass): samples resampled to the simulation
    step, the sample closest to the mean recorded as the prediction, and -- with distribution_aware_planning -- every
    plan() call checked against the whole distribution under the chance constraint (integrated_simulator.py:459-460,
    514-525)."""
    ep = load_dist_episodes()
    var = ep["meta"]["variants"][name]
    cfg = dict(var["config"])
    src = scripted_sample_source(var["n_samples"], cfg["pred_len"])
    sim = BatchedClosedLoop(cfg, [ep[name + "_ped_traj"]], engine=OracleEngine(cfg), resampler=OracleResampler(cfg),
                            sample_source=src)
    hist = sim.run()[0]
    assert_episode_matches(hist, sim.episodes[0].termination_reason, ep, name)


def test_two_episodes_in_lock_step_equal_their_solo_runs(episodes):
    """Episodes of different length in one batch: the finished one stops being stepped, the other is unaffected."""
    cfg = scenario_config(episodes["meta"])
    sim = BatchedClosedLoop(cfg, [episodes["fast_ped_traj"], episodes["shift_ped_traj"]], engine=OracleEngine(cfg),
                            resampler=OracleResampler(cfg))
    hists = sim.run()
    for h, ep, name in zip(hists, sim.episodes, ("fast", "shift")):
        assert_episode_matches(h, ep.termination_reason, episodes, name)


def test_save_results_writes_reference_layout(episodes, tmp_path):
    cfg = scenario_config(episodes["meta"])
    sim = BatchedClosedLoop(cfg, [episodes["fast_ped_traj"]], engine=OracleEngine(cfg), resampler=OracleResampler(cfg))
    sim.run(n_steps=5)
    (f,) = sim.save_results(str(tmp_path))
    z = np.load(f, allow_pickle=True)                      # object arrays, as in the reference's file; written above
    keys = episodes["meta"]["variants"]["fast"]["npz_keys"]
    assert set(z.files) == set(keys)
    assert z["ego_state"].dtype.kind == "U" and z["planned_x"].dtype == object and len(z["planned_x"]) == 5


def test_observer_samples_every_sgan_dt():
    """observer.py:52-86: 0.4 s sampling on a 0.1 s clock, leftover time carried (no drift)."""
    ob = Observer(obs_len=8, dt=0.1, sgan_dt=0.4)
    peds = ReplayPedestrians(np.zeros((100, 2, 2)), dt=0.1)
    stamps = []
    for _ in range(40):
        peds.step()
        n = len(ob.timestamps)
        ob.update(peds.positions, peds.time)
        if len(ob.timestamps) != n or (ob.timestamps and ob.timestamps[-1] == peds.time and peds.time not in stamps):
            stamps.append(peds.time)
    assert ob.is_ready and len(ob.history) == 8
    np.testing.assert_allclose(np.diff(list(ob.timestamps)), 0.4, atol=1e-9)


def test_replay_velocities_and_clamp():
    """replay_source.py:77-100: forward differences, last velocity repeated, head clamps at the last frame."""
    tr = np.cumsum(np.ones((5, 1, 2)), axis=0)
    p = ReplayPedestrians(tr, dt=0.5)
    np.testing.assert_allclose(p.velocities[:, 0, 0], 2.0)
    p.step(n=10)
    assert p._idx == 4 and abs(p.time - 5.0) < 1e-12
    np.testing.assert_allclose(p.goals, tr[-1])


def test_static_rectangles_expand_like_the_reference(episodes):
    """integrated_simulator.py:805-832: boundary points every 0.5 m, duplicates removed (count recorded from the reference)."""
    from integrated_path_planning_amd.closed_loop import expand_static_obstacles
    v = episodes["meta"]["variants"]["walls"]
    pts = expand_static_obstacles(v["config"]["static_obstacles"], step=0.5)
    assert len(pts) == v["n_static_points"] > 0
    assert np.array_equal(pts, np.unique(pts, axis=0))
    xmin, xmax, ymin, ymax = v["config"]["static_obstacles"][0]
    on_edge = (np.isclose(pts[:, 0], xmin) | np.isclose(pts[:, 0], xmax) | np.isclose(pts[:, 1], ymin) | np.isclose(pts[:, 1], ymax))
    first = (pts[:, 1] >= ymin - 1e-9) & (pts[:, 1] <= ymax + 0.5)
    assert on_edge[first & (pts[:, 0] <= xmax + 1e-9)].any()
    assert expand_static_obstacles(None).shape == (0, 2) and expand_static_obstacles([]).shape == (0, 2)
