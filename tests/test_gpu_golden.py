"""GPU parity: libfot (through the C ABI) against the reference's golden vectors."""
import numpy as np
import pytest

import eps_band
from helpers import EVAL_PATHS, TIGHT, request_from_golden, set_eval_path, wrap_angle
from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.planner import BatchPlanner

pytestmark = pytest.mark.gpu


def _planner(g):
    kw = g.planner_kwargs()
    return BatchPlanner(waypoints=(g["wx"], g["wy"]), **kw)


@pytest.mark.parametrize("eval_path", EVAL_PATHS)
def test_golden_case(golden, eval_path):
    """Every reference case under every evaluation kernel the library ships (helpers.EVAL_PATHS)."""
    g = golden
    bp = _planner(g)
    set_eval_path(bp, eval_path)
    res = bp.plan_batch([request_from_golden(g)])
    r = res.records[0]
    np.testing.assert_allclose(np.array(r.frenet0[:]), g["frenet0"], rtol=TIGHT, atol=TIGHT)
    np.testing.assert_allclose(np.array(r.ref0[:]), g["ref0"], rtol=TIGHT, atol=TIGHT)
    np.testing.assert_allclose(r.new_prev_s, float(g["prev_s_after"]), atol=1e-9)

    cost, status, keep, nt = bp.candidates(0)
    assert len(cost) == len(g["cand_cost"]) == r.n_cand
    np.testing.assert_array_equal(nt, g["cand_nt"])
    np.testing.assert_array_equal(keep, g["cand_keep"])
    np.testing.assert_allclose(cost, g["cand_cost"], rtol=TIGHT, atol=TIGHT)
    eps_band.check_status_table(bp, 0, status, g["cand_status"], f"golden {g.name} [{eval_path}]")   # equal + margin bookkeeping

    stats = g["stats"]
    want = {_abi.STATUS_NAMES[i]: int(stats[i]) for i in range(8) if stats[i] >= 0}
    assert res.stats(0) == want
    bi = int(g["best_index"])
    assert r.best_index == bi
    path = res.path(0)
    if bi < 0:
        assert path is None and r.status == _abi.PLAN_NO_PATH
        assert r.new_last_kappa == g.meta["last_kappa"]
        return
    np.testing.assert_allclose(path.cost, float(g["best_cost"]), rtol=TIGHT)
    for f in _abi.PATH_FIELDS:
        got = np.array(getattr(path, f))
        exp = g["best_" + f]
        assert len(got) == len(exp), f
        if f == "yaw":
            np.testing.assert_allclose(wrap_angle(got - exp), 0.0, atol=TIGHT)
        elif f == "c":
            # curvature at a crawl (|s_d| < 0.05: the reference's own value moves with the last bit of s_d there; the
            # `crawl_*` goldens are such fuzz instances put before the reference -- oracle/check.py CRAWL_*)
            from oracle.check import CRAWL_C_TOL, CRAWL_S_DOT
            loose = np.where(np.abs(g["best_s_d"]) < CRAWL_S_DOT, CRAWL_C_TOL, 0.0)
            err = np.abs(got - exp)
            assert np.all(err <= TIGHT + TIGHT * np.abs(exp) + loose), (f, float(err.max()))
            if g.name.startswith("crawl_"):
                k = int(err.argmax())
                print(f"\n{g.name} [{eval_path}]: curvature sample {k} (s_d {g['best_s_d'][k]:.3e}): reference {exp[k]!r}, "
                      f"library {got[k]!r}, |difference| {err[k]:.3e} ({err[k] / abs(exp[k]):.3e} relative)")
        else:
            np.testing.assert_allclose(got, exp, rtol=TIGHT, atol=TIGHT, err_msg=f)
    lk_tol = TIGHT if abs(g["best_s_d"][1]) >= 0.05 else 1e-6      # (_last_kappa = c[1]: the same rule)
    np.testing.assert_allclose(r.new_last_kappa, float(g["last_kappa_after"]), rtol=lk_tol, atol=lk_tol)


def test_spline_matches_reference(golden):
    g = golden
    bp = _planner(g)
    got = bp.path_coeffs()
    for arr, key in zip(got, ["sp_s", "sp_ax", "sp_bx", "sp_cx", "sp_dx", "sp_ay", "sp_by", "sp_cy", "sp_dy"]):
        np.testing.assert_allclose(arr, g[key], rtol=1e-10, atol=1e-11, err_msg=key)


def test_golden_case_through_adopted_reference_spline(golden):
    """reference_path handed over as an object with the reference's attributes (fot_set_path_coeffs):
    the device spline is then the reference's own coefficients bit for bit."""
    from types import SimpleNamespace
    g = golden
    ref_like = SimpleNamespace(s=g["sp_s"].tolist(),
                               sx=SimpleNamespace(a=g["sp_ax"], b=g["sp_bx"], c=g["sp_cx"], d=g["sp_dx"]),
                               sy=SimpleNamespace(a=g["sp_ay"], b=g["sp_by"], c=g["sp_cy"], d=g["sp_dy"]))
    bp = BatchPlanner(reference_path=ref_like, **g.planner_kwargs())
    got = bp.path_coeffs()
    for arr, key in zip(got, ["sp_s", "sp_ax", "sp_bx", "sp_cx", "sp_dx", "sp_ay", "sp_by", "sp_cy", "sp_dy"]):
        np.testing.assert_array_equal(arr, g[key])
    res = bp.plan_batch([request_from_golden(g)])
    r = res.records[0]
    assert r.best_index == int(g["best_index"])
    cost, status, keep, nt = bp.candidates(0)
    np.testing.assert_array_equal(status, g["cand_status"].astype(np.int32))
    np.testing.assert_allclose(cost, g["cand_cost"], rtol=TIGHT, atol=TIGHT)
    if r.best_index >= 0:
        np.testing.assert_allclose(np.array(r.x[: r.n_keep]), g["best_x"], rtol=TIGHT, atol=TIGHT)


@pytest.mark.parametrize("name", ["nan_ped_dist", "nan_ped_single"])
@pytest.mark.parametrize("layout_tsp", [False, True], ids=["spt", "tsp"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_nan_tracks_through_the_device_entry(name, layout_tsp, dtype):
    """The reference ignores a pedestrian whose track holds ONE NaN coordinate at every time step
    (frenet_planner.py:1211-1219).  The rule lives in the library: the un-sanitised tensor, resident in HBM the way a
    PyTorch predictor would hand it over (fot_plan_batch_device), in both layouts and both element types, gives the
    reference's status table -- and differs from it when the partly-NaN tracks are made finite."""
    import torch
    from conftest import Golden
    from integrated_path_planning_amd.batch import PackedBatch
    g = Golden(name)
    bp = _planner(g)
    rq = request_from_golden(g)
    tensor = rq.dist if rq.dist is not None else rq.dyn
    bad = np.isnan(tensor).any(axis=(-1, -2))
    assert bad.any() and not np.isnan(tensor).all(axis=(-1, -2))[bad].all(), "the case needs a PARTLY NaN track"
    pb = PackedBatch([rq], dtype, dyn_layout_tsp=layout_tsp)
    dev = torch.device("cuda", 0)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    stat = torch.from_numpy(pb.static_xy).to(dev) if pb.static_xy.size else None
    out = torch.zeros(_abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    bp.plan_packed_device(pb.with_device_obstacles(stat.data_ptr() if stat is not None else None, dyn.data_ptr()),
                          out.data_ptr(), st.cuda_stream)
    st.synchronize()
    rec = _abi.Result.from_buffer_copy(out.cpu().numpy().tobytes())
    cost, status, keep, nt = bp.candidates(0)
    if dtype == np.float64:                              # (float32 obstacle values are not the golden's inputs)
        np.testing.assert_array_equal(status, g["cand_status"].astype(np.int32))
        assert rec.best_index == int(g["best_index"])
    # the oracle on exactly the values the device saw
    from helpers import oracle_plan_for_request
    from oracle import oracle as orc
    okw = g.planner_kwargs()
    rounded = request_from_golden(g)
    for f in ("static", "dyn", "dist"):
        v = getattr(rounded, f)
        if v is not None:
            setattr(rounded, f, np.asarray(v).astype(dtype).astype(np.float64))
    want = oracle_plan_for_request(orc, orc.make_params(**okw), orc.Spline(g["wx"], g["wy"]), rounded, table=True)
    np.testing.assert_array_equal(status, want.cand_status)
    assert rec.best_index == want.best_index
