"""SURVEY 8(f4) on the GPU: whole reference episodes free-running through BatchedClosedLoop on libfot -- every step one
prediction launch, two safety-metric launches, one plan batch with all escalation levels of all episodes, one
nearest-point launch."""
import time

import numpy as np
import pytest

from closed_loop_common import (assert_episode_matches, assert_npz_layout, load_dist_episodes, load_episodes, scenario_config,
                                scripted_sample_source)
from integrated_path_planning_amd.closed_loop import BatchedClosedLoop

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def episodes():
    return load_episodes()


def test_three_episodes_in_lock_step_match_the_reference(episodes, tmp_path):
    """274-step run to the goal (20 emergency stops, all three vehicle states) and two runs that end in a collision,
    advanced together: each must equal its own reference episode step by step."""
    cfg = scenario_config(episodes["meta"])
    names = ("base", "fast", "shift")
    sim = BatchedClosedLoop(cfg, [episodes[n + "_ped_traj"] for n in names])
    t0 = time.perf_counter()
    hists = sim.run()
    wall = time.perf_counter() - t0
    for h, ep, n in zip(hists, sim.episodes, names):
        assert_episode_matches(h, ep.termination_reason, episodes, n)
    files = sim.save_results(str(tmp_path))
    for f, n in zip(files, names):
        z = np.load(f, allow_pickle=True)                  # object arrays as in the reference's file; written just above
        assert_npz_layout({k: z[k] for k in z.files}, episodes["meta"]["variants"][n]["npz_keys"],
                          episodes["meta"]["variants"][n]["steps"])
    sim.close()
    print(f"3 episodes, {sum(len(h) for h in hists)} episode-steps in {wall:.2f} s")


@pytest.mark.parametrize("name", ["walls", "turn", "footprint", "inflate", "rnd0", "rnd1", "rnd2", "rnd3", "rnd4", "rnd5"])
def test_other_scenarios_match_the_reference(episodes, name):
    """scenario_02: corridor of static obstacle rectangles (expanded to boundary points); scenario_03: right turn;
    scenario_01 with the three-circle ego footprint; scenario_01 with the dynamic collision margin inflated by 1.2; six
    random pedestrian scripts (jittered starts, scaled and turned velocities) on the three scenarios."""
    cfg = scenario_config(episodes["meta"], name)
    with BatchedClosedLoop(cfg, [episodes[name + "_ped_traj"]] * 2) as sim:
        hists = sim.run()
        for h, ep in zip(hists, sim.episodes):
            assert_episode_matches(h, ep.termination_reason, episodes, name)


def test_replicated_episodes_are_identical(episodes):
    """32 copies of one episode in one batch: every copy must reproduce the reference (batch independence)."""
    cfg = scenario_config(episodes["meta"])
    sim = BatchedClosedLoop(cfg, [episodes["shift_ped_traj"]] * 32)
    hists = sim.run()
    for i in (0, 13, 31):
        assert_episode_matches(hists[i], sim.episodes[i].termination_reason, episodes, "shift")
    x0 = [r.ego.x for r in hists[0]]
    for h in hists[1:]:
        assert [r.ego.x for r in h] == x0
    sim.close()


@pytest.mark.parametrize("name", ["s6_eps02", "s4_eps0"])
def test_distribution_aware_episodes_with_samples_resident_in_hbm(name):
    """The same reference episodes with the predictor's raw samples handed over as a DEVICE tensor (what Social-GAN on
    PyTorch-ROCm leaves): resampled into the planner's tensor inside the step's one call (fot_loop_step with
    fot_loop_frame.dist_raw), never crossing PCIe -- against the reference, and step by step against the run whose
    samples travel through the host."""
    import torch
    ep = load_dist_episodes()
    var = ep["meta"]["variants"][name]
    cfg = dict(var["config"])
    host_src = scripted_sample_source(var["n_samples"], cfg["pred_len"])
    dev = torch.device("cuda", 0)
    dev_src = lambda last, prev: torch.from_numpy(np.ascontiguousarray(host_src(last, prev), dtype=np.float64)).to(dev)
    with BatchedClosedLoop(cfg, [ep[name + "_ped_traj"]] * 2, sample_source=dev_src, device_samples=True) as sim:
        assert sim._native and sim._device_samples
        hists = sim.run()
        for h, e in zip(hists, sim.episodes):
            assert_episode_matches(h, e.termination_reason, ep, name)
        h_dev = [list(h) for h in hists]
    with BatchedClosedLoop(cfg, [ep[name + "_ped_traj"]] * 2, sample_source=host_src) as sim:
        h_host = [list(h) for h in sim.run()]
    _same_histories(h_host, h_dev)


@pytest.mark.parametrize("name", ["s6_eps02", "s4_eps0", "s5_best_only"])
def test_distribution_aware_episodes_match_the_reference(name):
    """Rolling-horizon episodes of the headline workload's kind: every step a sampled prediction DISTRIBUTION of all
    pedestrians goes through the device resampler into the planner, which applies the chance constraint
    (floor(0.2 * 6) = 1 colliding sample allowed / none allowed / planning on the best sample only).  Reference episodes
    generated with the same scripted sample source in the place of the Social-GAN forward pass; two copies per batch."""
    ep = load_dist_episodes()
    var = ep["meta"]["variants"][name]
    cfg = dict(var["config"])
    src = scripted_sample_source(var["n_samples"], cfg["pred_len"])
    with BatchedClosedLoop(cfg, [ep[name + "_ped_traj"]] * 2, sample_source=src) as sim:
        hists = sim.run()
        for h, e in zip(hists, sim.episodes):
            assert_episode_matches(h, e.termination_reason, ep, name)
        assert [r.ego.x for r in hists[0]] == [r.ego.x for r in hists[1]]


def _same_histories(ha, hb):
    assert len(ha) == len(hb)
    for a, b in zip(ha, hb):
        assert len(a) == len(b)
        for ra, rb in zip(a, b):
            assert (ra.ego.x, ra.ego.y, ra.ego.yaw, ra.ego.v, ra.ego.a, ra.ego.jerk, ra.ego.state) == \
                   (rb.ego.x, rb.ego.y, rb.ego.yaw, rb.ego.v, rb.ego.a, rb.ego.jerk, rb.ego.state)
            assert ra.metrics == rb.metrics
            assert (ra.planned_path is None) == (rb.planned_path is None)
            if ra.planned_path is not None:
                np.testing.assert_array_equal(ra.planned_path.x, rb.planned_path.x)
                assert ra.planned_path.cost == rb.planned_path.cost
            assert (ra.predicted_trajectories is None) == (rb.predicted_trajectories is None)
            if ra.predicted_trajectories is not None:
                np.testing.assert_array_equal(ra.predicted_trajectories, rb.predicted_trajectories)


def test_two_calls_per_step_equal_five_calls_per_step(episodes):
    """The fused step (fot_loop_plan + fot_loop_observe, prediction resident in HBM) and the step of five separate calls
    (prediction through the host) are the same computation: the three reference episodes again, unfused, against the
    reference AND record by record against the fused run -- ego states, metrics, selected paths, and the predictions the
    fused run computes again when its history is read."""
    cfg = scenario_config(episodes["meta"])
    names = ("base", "fast", "shift")
    tracks = [episodes[n + "_ped_traj"] for n in names]
    with BatchedClosedLoop(cfg, tracks, fused=False) as plain:
        assert not plain._fused
        h_plain = plain.run()
        for h, ep, n in zip(h_plain, plain.episodes, names):
            assert_episode_matches(h, ep.termination_reason, episodes, n)
        h_plain = [list(h) for h in h_plain]
    with BatchedClosedLoop(cfg, tracks, fused="two-call") as fused:
        assert fused._fused and not fused._native
        h_fused = [list(h) for h in fused.run()]
    _same_histories(h_plain, h_fused)
    # ... and the whole step behind ONE call (fot_loop_step: episode state, fail-safe machine and retry loop in the library)
    with BatchedClosedLoop(cfg, tracks) as native:
        assert native._native
        h_native = [list(h) for h in native.run()]
    _same_histories(h_plain, h_native)


def test_fused_step_with_standing_and_walking_crowds(episodes):
    """Episodes that disagree on the prepend rule in one frame (integrated_simulator.py:503-511): a crowd that stands
    still (its prediction starts AT the current positions: nothing is prepended, T = n_dense), a walking crowd
    (T = n_dense + 1), one without pedestrians, a standing one again -- the tensor then holds blocks of different
    lengths, written by one launch per run of episodes.  Fused == unfused, record by record; histories are read after
    the loop was closed (the predictions of the fused run are materialised on close)."""
    cfg = scenario_config(episodes["meta"])
    walk = episodes["base_ped_traj"][:140]
    stand = np.repeat(walk[60:61], len(walk), axis=0)
    none = np.zeros((len(walk), 0, 2))
    tracks = [stand, walk, none, stand[:, :3], walk[:, ::2]]
    runs = []
    for fused in (False, "two-call", True):
        with BatchedClosedLoop(cfg, tracks, fused=fused) as sim:
            hists = sim.run(60)
        runs.append([list(h) for h in hists])
    _same_histories(runs[0], runs[1])
    _same_histories(runs[0], runs[2])
