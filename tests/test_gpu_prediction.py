"""SURVEY 8(f1) on the GPU: prediction resampling against the reference's vectors, and the device-resident
hand-over into the planner."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from helpers import assert_record_matches_oracle, oracle_plan_for_request
from integrated_path_planning_amd import _abi, synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch, PlanRequest
from integrated_path_planning_amd.planner import BatchPlanner
from integrated_path_planning_amd.prediction import PredictionResampler
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cases():
    z = np.load(os.path.join(GOLDEN_DIR, "prediction", "cases.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


@pytest.fixture(scope="module")
def engine():
    return BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), **syn.CONFIG3_PLANNER)


def test_process_prediction_matches_reference(cases, engine):
    for m in cases["meta"]["resample"]:
        c = m["case"]
        rs = PredictionResampler(engine, pred_len=m["pred_len"], sgan_dt=m["sgan_dt"], sim_dt=m["sim_dt"],
                                 plan_horizon=m["plan_horizon"])
        anchor = cases[f"c{c}_anchor"] if m["with_anchor"] else None
        got = rs.process_prediction(cases[f"c{c}_pred"], anchor_pos=anchor, staleness=m["staleness"])
        want = cases[f"c{c}_dense"]
        assert got.shape == want.shape, m
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, err_msg=str(m))
        last = cases[f"c{c}_cv1"][:, 0, :]
        obs = np.stack([cases[f"c{c}_prev"], last])
        np.testing.assert_allclose(rs.predict_cv(obs, m["staleness"]), cases[f"c{c}_cv"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(rs.predict_cv(obs[-1:], m["staleness"]), cases[f"c{c}_cv1"], rtol=1e-12, atol=1e-12)
        # current-position prepend (integrated_simulator.py:503-525): one more leading time step
        cur = last + 0.01
        pre = rs.process_prediction(cases[f"c{c}_pred"], anchor_pos=anchor, staleness=m["staleness"], current=cur)
        np.testing.assert_array_equal(pre[:, 0, :], cur)
        np.testing.assert_array_equal(pre[:, 1:, :], got)


def test_best_sample_matches_reference(cases, engine):
    """closest-to-mean pick (predict_single_best :346-351): reference index on the reference's own samples through
    the oracle, and the device distance sums against the oracle's on a resampled distribution."""
    rs = PredictionResampler(engine)
    for m in cases["meta"]["select"]:
        samples = cases[f"s{m['case']}_samples"]
        assert orc.best_sample(samples)[0] == m["best"]
        pred = np.transpose(samples[:, :, 3::4][:, :, :12], (0, 2, 1, 3))          # [S, 12, P, 2]: every 0.4 s
        dense, d_gpu = rs.process_prediction(pred, anchor_pos=samples[0, :, 0], staleness=0.0, want_sample_dist=True)
        best_ref, d_ref = orc.best_sample(dense)
        np.testing.assert_allclose(d_gpu, d_ref, rtol=1e-10)
        assert rs.best_sample(d_gpu) == best_ref


def test_device_resident_handover_into_planner(engine):
    """raw predictions in HBM -> fot_resample_predictions -> fot_plan_batch_device, no host copy of the tensor;
    same plan as the oracle fed with the oracle's own resampling."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    S, P, L = 20, 30, 12
    inst = syn.config3_instance(21)
    ego = inst.ego
    p0 = np.stack([rng.uniform(ego[0] - 5, ego[0] + 60, P), rng.uniform(-12, 12, P)], axis=1)
    vel = rng.normal(0, 1.2, (S, 1, P, 2)) * 0.3 + rng.normal(0, 1.0, (1, 1, P, 2))
    steps = (np.arange(1, L + 1) * 0.4)[None, :, None, None]
    raw = (p0[None, None] + vel * steps + np.cumsum(rng.normal(0, 0.05, (S, L, P, 2)), axis=1)).astype(np.float32)
    staleness, cur = 0.2, p0 + rng.normal(0, 0.02, p0.shape)
    rs = PredictionResampler(engine)
    T = rs.n_dense + 1
    raw_dev = torch.from_numpy(raw).to(dev)
    obs_dev = torch.zeros((S, P, T, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    t_out, dist = rs.resample_device(raw_dev.data_ptr(), np.float32, S, P, p0, cur, staleness, obs_dev.data_ptr(),
                                     np.float32, stream.cuda_stream, want_sample_dist=True)
    assert t_out == T == 51
    # plan straight from the device tensor
    req = PlanRequest(*ego, dist=np.zeros((S, P, T, 2), np.float32))          # shapes only; data stays on the device
    pb = PackedBatch([req], np.float32)
    out = torch.zeros(_abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    engine.plan_packed_device(pb.with_device_obstacles(None, obs_dev.data_ptr()), out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    rec = (_abi.Result * 1).from_buffer_copy(out.cpu().numpy().tobytes())[0]
    # checker: oracle resampling (float64) rounded to the float32 the device stored, then the oracle planner
    want_t = np.stack([orc.process_prediction(raw[s].astype(np.float64), p0, staleness) for s in range(S)])
    want_t = np.concatenate([np.broadcast_to(cur[None, :, None, :], (S, P, 1, 2)), want_t], axis=2)
    got_t = obs_dev.cpu().numpy()
    np.testing.assert_allclose(got_t, want_t, rtol=2e-7, atol=1e-6)          # float32 storage of a float64 result
    params, sp = orc.make_params(**syn.CONFIG3_PLANNER), orc.Spline(syn.STRAIGHT_WX, syn.STRAIGHT_WY)
    rq = PlanRequest(*ego, dist=got_t.astype(np.float64))
    assert_record_matches_oracle(rec, oracle_plan_for_request(orc, params, sp, rq), label="resample->plan")
    assert rs.best_sample(dist) == orc.best_sample(got_t.astype(np.float64))[0]

    # the same hand-over in the layout the broad phase wants: the resampler writes [T][S][P][2] (FOT_OUT_TMAJOR), the
    # batch says so (FOT_DYN_LAYOUT_TSP); tensor = the transposed one above bit for bit, plan record identical
    obs_tm = torch.zeros((T, S, P, 2), dtype=torch.float32, device=dev)
    t2, dist2 = rs.resample_device(raw_dev.data_ptr(), np.float32, S, P, p0, cur, staleness, obs_tm.data_ptr(),
                                   np.float32, stream.cuda_stream, want_sample_dist=True, t_major=True)
    assert t2 == T
    assert torch.equal(obs_tm, obs_dev.permute(2, 0, 1, 3).contiguous())
    np.testing.assert_array_equal(dist2, dist)
    pb_tm = PackedBatch([req], np.float32, dyn_layout_tsp=True)
    out_tm = torch.zeros(_abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    engine.plan_packed_device(pb_tm.with_device_obstacles(None, obs_tm.data_ptr()), out_tm.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    assert torch.equal(out_tm, out)


def test_predict_cv_float32_observation_mode(engine):
    rng = np.random.default_rng(12)
    obs = rng.normal(0, 20, (8, 21, 2))
    rs = PredictionResampler(engine, pred_len=12, sgan_dt=0.4, sim_dt=0.1, plan_horizon=5.0)
    for f32 in (False, True):
        got = rs.predict_cv(obs, staleness=0.2, float32_observations=f32)
        want = orc.predict_cv(obs[-1], obs[-2], 0.2, float32_observations=f32)
        np.testing.assert_allclose(got, want, rtol=1e-15, atol=1e-13)
