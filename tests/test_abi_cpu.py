"""No-GPU checks of the drop-in boundary: libfot.so loads, exports every symbol include/fot.h declares,
the ctypes mirror has the C layout, and the host-side packing / failure behaviour is right."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_names, Golden
from integrated_path_planning_amd import _abi, synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch, PlanRequest
from integrated_path_planning_amd.params import make_params

HEADER = os.path.join(ROOT, "include", "fot.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fot_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _abi.lib()
    decl = declared_symbols()
    assert len(decl) >= 18
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/fot.h but not exported by libfot.so"
    assert sorted(_abi.SYMBOLS) == decl, "integrated_path_planning_amd/_abi.py SYMBOLS out of sync with include/fot.h"
    assert b"gfx950" in lib.fot_version()


def test_ctypes_layout_matches_c(tmp_path):
    """sizeof/offsetof from a C translation unit that includes include/fot.h."""
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "fot.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(fot_params), sizeof(fot_ego), sizeof(fot_overrides), sizeof(fot_result), sizeof(fot_batch));
  printf("%zu %zu %zu %zu %zu\n", offsetof(fot_result, cost), offsetof(fot_result, stats), offsetof(fot_result, frenet0),
         offsetof(fot_result, t), offsetof(fot_result, c));
  printf("%zu %zu %zu\n", offsetof(fot_batch, static_xy), offsetof(fot_batch, dyn_dims), offsetof(fot_params, footprint_offsets));
  printf("%zu %zu %zu %zu %zu\n", sizeof(fot_loop_frame), offsetof(fot_loop_frame, ped_off), offsetof(fot_loop_frame, prepend),
         offsetof(fot_loop_frame, staleness), offsetof(fot_loop_frame, rp));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(fot_loop_request), offsetof(fot_loop_request, ego.has_prev_s),
         offsetof(fot_loop_request, overrides), offsetof(fot_loop_request, target_speed),
         offsetof(fot_loop_request, max_stop_distance), offsetof(fot_loop_request, episode));
  return 0; }''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    got = [int(v) for v in out]
    R, B, P = _abi.Result, _abi.Batch, _abi.Params
    want = [C.sizeof(P), C.sizeof(_abi.Ego), C.sizeof(_abi.Overrides), C.sizeof(R), C.sizeof(B),
            R.cost.offset, R.stats.offset, R.frenet0.offset, R.t.offset, R.c.offset,
            B.static_xy.offset, B.dyn_dims.offset, P.footprint_offsets.offset]
    F, Q = _abi.LoopFrame, _abi.LoopRequest
    want += [C.sizeof(F), F.ped_off.offset, F.prepend.offset, F.staleness.offset, F.rp.offset,
             C.sizeof(Q), Q.ego.offset + _abi.Ego.has_prev_s.offset, Q.overrides.offset, Q.target_speed.offset,
             Q.max_stop_distance.offset, Q.episode.offset]
    assert got == want
    # closed_loop.py fills fot_loop_request column-wise through float64 / int32 views: 15 eight-byte slots
    assert got[-6:] == [120, 56, 64, 96, 104, 112]


def test_create_fails_loudly_without_gpu_or_succeeds_with_one():
    lib = _abi.lib()
    p = make_params(dt=0.1)
    h = C.c_void_p()
    rc = lib.fot_create(C.byref(p), -1, C.byref(h))
    if rc == _abi.OK:
        lib.fot_destroy(h)
    else:
        assert rc == _abi.ERR_HIP and not h.value
        assert b"HIP" in lib.fot_last_error(None) or b"device" in lib.fot_last_error(None)


def test_create_rejects_unsupported_configurations():
    lib = _abi.lib()
    h = C.c_void_p()
    for kw, code in ((dict(dt=0.01), _abi.ERR_UNSUPPORTED),          # 501 samples > FOT_MAX_NT (256)
                     (dict(dt=-1.0), _abi.ERR_INVALID),
                     (dict(dt=0.1, min_t=0.5, max_t=7.0), _abi.ERR_UNSUPPORTED)):   # 66 horizons > FOT_MAX_TI (64)
        p = make_params(**kw)
        assert lib.fot_create(C.byref(p), -1, C.byref(h)) == code, kw
        assert lib.fot_last_error(None)
    # what rounds 1-2 refused: 101 samples per candidate (dt = 0.05 s), 41 horizons (min_t = 1 s) -- the reference has
    # no such limit (frenet_planner.py:397-398, 586-617); without a GPU the call gets as far as the device
    # ... and round 4: 251 samples per candidate (dt = 0.02 s), 51 horizons
    for kw in (dict(dt=0.05), dict(dt=0.1, min_t=1.0, max_t=5.0), dict(dt=0.05, min_t=2.0, max_t=5.0), dict(dt=0.02)):
        p = make_params(**kw)
        rc = lib.fot_create(C.byref(p), -1, C.byref(h))
        assert rc in (_abi.OK, _abi.ERR_HIP), (kw, lib.fot_last_error(None))
        if rc == _abi.OK:
            lib.fot_destroy(h)


def test_packed_batch_layout():
    inst = [syn.config3_instance(0, S=3, P=4, T=5), syn.config2_instance(1, n_static=2), syn.config3_instance(2, S=2, P=1, T=7)]
    reqs = [PlanRequest(*inst[0].ego, dist=inst[0].dist), PlanRequest(*inst[1].ego, static=inst[1].static),
            PlanRequest(*inst[2].ego, dyn=inst[2].dist[0], overrides={"max_accel": 3.0}, max_stop_distance=4.0,
                        prev_s=1.5, last_kappa=0.25)]
    pb = PackedBatch(reqs, np.float32)
    assert pb.c.n_inst == 3 and pb.c.obstacle_dtype == _abi.F32
    assert pb.dyn_dims.tolist() == [[_abi.DYN_DISTRIBUTION, 3, 4, 5], [0, 0, 0, 0], [_abi.DYN_SINGLE, 1, 1, 7]]
    assert pb.dyn_off.tolist() == [0, 60, 60] and pb.dyn_xy.shape == (67, 2) and pb.dyn_xy.dtype == np.float32
    assert pb.static_off.tolist() == [0, 0, 2, 2] and pb.static_xy.shape == (2, 2)
    np.testing.assert_array_equal(pb.dyn_xy[:60].reshape(3, 4, 5, 2), inst[0].dist)
    assert np.isnan(pb.overrides[0].max_accel) and pb.overrides[2].max_accel == 3.0
    assert np.isnan(pb.max_stop[0]) and pb.max_stop[2] == 4.0
    assert pb.ego[2].has_prev_s == 1 and pb.ego[2].prev_s == 1.5 and pb.ego[2].last_kappa == 0.25
    assert pb.ego[0].has_prev_s == 0
    # the distribution wins over the single sample, bad shapes mean "no obstacles" (frenet_planner.py:1043-1047, 1205-1208)
    both = PackedBatch([PlanRequest(0, 0, 0, 1, 0, dyn=inst[0].dist[0], dist=inst[0].dist)])
    assert both.dyn_dims[0, 0] == _abi.DYN_DISTRIBUTION
    none = PackedBatch([PlanRequest(0, 0, 0, 1, 0, dyn=np.empty((0, 0, 2)), static=np.empty((0, 2)))])
    assert none.dyn_dims[0, 0] == _abi.DYN_NONE and not none.c.static_xy and not none.c.dyn_xy


def test_lattice_size_matches_reference_counts():
    """synthetic.lattice_size against the candidate counts the reference generated (golden cand tables)."""
    for name in golden_names():
        g = Golden(name)
        kw = g.meta["planner"]
        moving = g["frenet0"][1] > 0.1
        n = syn.lattice_size(g.meta["target_speed"], dt=kw.get("dt", 0.2), min_t=kw.get("min_t", 4.0),
                             max_t=kw.get("max_t", 5.0), d_t_s=kw.get("d_t_s", 5.0 / 3.6),
                             d_road_w=kw.get("d_road_w", 0.5), max_road_width=kw.get("max_road_width", 7.0), moving=moving)
        assert n == len(g["cand_cost"]), name


def test_reference_style_spline_object_is_accepted():
    """The shim reads knots and coefficients off any CubicSpline2D-like object (reference attributes
    .s list, .sx/.sy with a[n], b[n-1], c[n], d[n-1]; cubic_spline.py:30-45, 201-204)."""
    from types import SimpleNamespace
    from integrated_path_planning_amd.planner import spline_arrays
    g = Golden("curved_a")
    n = len(g["sp_s"])
    ref_like = SimpleNamespace(s=g["sp_s"].tolist(),
                               sx=SimpleNamespace(a=g["sp_ax"], b=g["sp_bx"], c=g["sp_cx"], d=g["sp_dx"]),
                               sy=SimpleNamespace(a=g["sp_ay"], b=g["sp_by"], c=g["sp_cy"], d=g["sp_dy"]))
    arrs = spline_arrays(ref_like)
    assert [len(a) for a in arrs] == [n, n, n - 1, n, n - 1, n, n - 1, n, n - 1]
    np.testing.assert_array_equal(arrs[3], g["sp_cx"])
    bad = SimpleNamespace(s=[0.0, 1.0], sx=SimpleNamespace(a=[0, 1], b=[1, 2], c=[0, 0], d=[0]), sy=ref_like.sy)
    with pytest.raises(ValueError):
        spline_arrays(bad)


def test_header_constants_match_the_python_mirror():
    import re
    from integrated_path_planning_amd import _abi
    text = open(os.path.join(os.path.dirname(__file__), "..", "include", "fot.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define (FOT_[A-Z_]+) (\d+)\b", text)}
    assert defs["FOT_PROFILE_KERNELS"] == _abi.PROFILE_KERNELS
    assert defs["FOT_MAX_NT"] == _abi.MAX_NT


def test_abi_info_matches_the_binding_and_the_header():
    """fot_abi_info: what the library was built with == the ctypes mirror == include/fot.h."""
    lib = _abi.lib()
    want = _abi.abi_expectation()
    got = (C.c_int32 * 64)()
    n = lib.fot_abi_info(64, got)
    text = open(HEADER).read()
    defs = {k: int(v) for k, v in re.findall(r"#define (FOT_[A-Z_]+) (\d+)\b", text)}
    assert n == defs["FOT_ABI_INFO_WORDS"] == len(want) == len(_abi.ABI_WORD_NAMES)
    assert list(got[:n]) == want
    assert got[0] == defs["FOT_ABI_VERSION"] == _abi.ABI_VERSION
    for name, value in (("FOT_MAX_NT", _abi.MAX_NT), ("FOT_MAX_CIRCLES", _abi.MAX_CIRCLES), ("FOT_MAX_TI", _abi.MAX_TI),
                        ("FOT_MAX_TV", _abi.MAX_TV), ("FOT_MAX_BRAKE", _abi.MAX_BRAKE), ("FOT_MAX_SAMPLES", _abi.MAX_SAMPLES),
                        ("FOT_MAX_PRED_LEN", _abi.MAX_PRED_LEN), ("FOT_PROFILE_KERNELS", _abi.PROFILE_KERNELS),
                        ("FOT_MARGIN_GROUPS", _abi.MARGIN_GROUPS)):
        assert defs[name] == value == got[_abi.ABI_WORD_NAMES.index(name)], name
    assert lib.fot_abi_info(3, got) == n                          # a short array is not written past its end
    assert lib.fot_abi_info(0, None) == n


def test_a_library_with_other_layouts_is_refused(tmp_path):
    """Round 3's abort (gpurun_out/ab_r3_select.log: `double free or corruption (out)` at bench.py's exit) was an OLDER
    libfot.so -- FOT_PROFILE_KERNELS 4 -- under a binding that allocated three slots: fot_profile_read wrote past both
    arrays.  The loader now compares fot_abi_info() with the ctypes mirror and refuses; a library without the symbol is
    refused as well."""
    want = _abi.abi_expectation()
    for idx, delta, frag in ((_abi.ABI_WORD_NAMES.index("FOT_PROFILE_KERNELS"), 1, "FOT_PROFILE_KERNELS"),
                             (_abi.ABI_WORD_NAMES.index("sizeof(fot_result)"), -7680, "sizeof(fot_result)"),
                             (0, -1, "FOT_ABI_VERSION")):
        vals = list(want)
        vals[idx] += delta
        src = tmp_path / f"fake{idx}.c"
        src.write_text("#include <stdint.h>\nint32_t fot_abi_info(int32_t cap, int32_t *out) { static const int32_t v[] = {"
                       + ",".join(map(str, vals)) + "}; for (int i = 0; i < %d && i < cap && out; ++i) out[i] = v[i]; return %d; }\n"
                       % (len(vals), len(vals)))
        so = tmp_path / f"fake{idx}.so"
        subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
        with pytest.raises(ImportError, match=re.escape(frag)):
            _abi._check_abi(C.CDLL(str(so)), str(so))
    src = tmp_path / "old.c"
    src.write_text("int fot_create(void) { return 0; }\n")
    so = tmp_path / "old.so"
    subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    with pytest.raises(ImportError, match="predates fot_abi_info"):
        _abi._check_abi(C.CDLL(str(so)), str(so))


def test_destroy_ignores_what_create_did_not_return():
    """fot_destroy is idempotent and never reads a pointer it does not own."""
    lib = _abi.lib()
    n0 = lib.fot_live_handles()
    junk = (C.c_char * 4096)()
    lib.fot_destroy(C.c_void_p(C.addressof(junk)))                 # not a handle: ignored
    lib.fot_destroy(None)
    assert lib.fot_live_handles() == n0


def test_exit_hook_closes_registered_owners():
    """_abi.close_all (the atexit hook) closes every owner that is still registered, once."""
    closed = []

    class Owner:
        def close(self):
            closed.append(self)
            _abi.unregister_owner(self)

    a, b = Owner(), Owner()
    _abi.register_owner(a)
    _abi.register_owner(b)
    b.close()
    _abi.close_all()
    assert closed.count(a) == 1 and closed.count(b) == 1
    _abi.close_all()
    assert len(closed) == 2
    import atexit
    atexit.unregister(_abi.close_all)                               # (registered exactly once, at import)
    atexit.register(_abi.close_all)
