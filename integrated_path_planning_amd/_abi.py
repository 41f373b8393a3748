"""ctypes mirror of include/fot.h and the loader of libfot.so.

The library is the product: if it is missing or cannot be loaded this module
raises -- there is no Python or CPU fallback.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import weakref

MAX_NT = 256
MAX_CIRCLES = 8
MAX_SAMPLES = 64

OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_PATH_SET = -1, -2, -3, -4
ST_SPEED, ST_ACCEL, ST_CURVATURE, ST_LAT_ACCEL, ST_ROAD, ST_COLLISION, ST_OK, ST_STOP_DISTANCE, ST_DROPPED = range(9)
PLAN_OK, PLAN_NO_PATH, PLAN_C2F_FAILED = 0, 1, 2
F32, F64 = 0, 1
DYN_NONE, DYN_SINGLE, DYN_DISTRIBUTION = 0, 1, 2
DYN_LAYOUT_TSP = 0x10                    # OR-ed into dyn_dims[i][0]: tensor laid out [T][S][P][2]
OUT_DEVICE, OUT_TMAJOR = 1, 2            # bits of the resampler's `on_device` argument

# reference's last_check_stats keys in fot_result.stats[] order (frenet_planner.py:910-918, 324)
STATUS_NAMES = ["max_speed_error", "max_accel_error", "max_curvature_error", "max_lat_accel_error",
                "road_bound_error", "collision_error", "ok", "stop_distance_error"]
# FrenetPath fields in fot_result order (data_structures.py:164-178)
PATH_FIELDS = ["t", "s", "s_d", "s_dd", "s_ddd", "d", "d_d", "d_dd", "d_ddd", "x", "y", "yaw", "v", "a", "c"]


class Params(C.Structure):
    _fields_ = [
        ("max_speed", C.c_double), ("max_accel", C.c_double), ("max_curvature", C.c_double),
        ("max_lat_accel", C.c_double),
        ("dt", C.c_double), ("d_road_w", C.c_double), ("max_road_width", C.c_double),
        ("robot_radius", C.c_double), ("obstacle_radius", C.c_double),
        ("min_t", C.c_double), ("max_t", C.c_double), ("d_t_s", C.c_double),
        ("k_j", C.c_double), ("k_t", C.c_double), ("k_d", C.c_double), ("k_s_dot", C.c_double),
        ("k_lat", C.c_double), ("k_lon", C.c_double),
        ("chance_epsilon", C.c_double), ("collision_margin_inflation", C.c_double),
        ("n_circles", C.c_int32), ("_pad", C.c_int32),
        ("footprint_radius", C.c_double),
        ("footprint_offsets", C.c_double * MAX_CIRCLES),
    ]


class Ego(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("yaw", C.c_double), ("v", C.c_double),
                ("a", C.c_double), ("last_kappa", C.c_double), ("prev_s", C.c_double),
                ("has_prev_s", C.c_int32), ("_pad", C.c_int32)]


class Overrides(C.Structure):
    _fields_ = [("max_speed", C.c_double), ("max_accel", C.c_double),
                ("max_curvature", C.c_double), ("max_lat_accel", C.c_double)]


_ARR = C.c_double * MAX_NT


class Result(C.Structure):
    _fields_ = ([("status", C.c_int32), ("best_index", C.c_int32), ("n_cand", C.c_int32), ("n_keep", C.c_int32),
                 ("cost", C.c_double), ("stats", C.c_int32 * 8), ("stats_valid", C.c_int32), ("_pad", C.c_int32),
                 ("new_last_kappa", C.c_double), ("new_prev_s", C.c_double),
                 ("frenet0", C.c_double * 6), ("ref0", C.c_double * 6)]
                + [(f, _ARR) for f in PATH_FIELDS])


class WireHeader(C.Structure):
    _fields_ = [("status", C.c_int32), ("best_index", C.c_int32), ("n_cand", C.c_int32), ("n_keep", C.c_int32),
                ("cost", C.c_double), ("stats", C.c_int32 * 8), ("stats_valid", C.c_int32), ("n_total", C.c_int32),
                ("new_last_kappa", C.c_double), ("new_prev_s", C.c_double),
                ("frenet0", C.c_double * 6), ("ref0", C.c_double * 6)]


class ResampleParams(C.Structure):
    _fields_ = [("sgan_dt", C.c_double), ("sim_dt", C.c_double), ("plan_horizon", C.c_double)]


class Safety(C.Structure):
    _fields_ = [("min_distance", C.c_double), ("ttc", C.c_double), ("clearance", C.c_double),
                ("clearance_ahead", C.c_double), ("collision", C.c_int32), ("_pad", C.c_int32)]


class LoopFrame(C.Structure):                                      # fot_loop_frame
    _fields_ = [("n_episodes", C.c_int32), ("pred_len", C.c_int32), ("use_footprint", C.c_int32), ("_pad", C.c_int32),
                ("ped_off", C.c_void_p), ("ped_pos", C.c_void_p), ("ped_vel", C.c_void_p),
                ("obs_last", C.c_void_p), ("obs_prev", C.c_void_p), ("prepend", C.c_void_p), ("ego", C.c_void_p),
                ("staleness", C.c_double), ("ego_radius", C.c_double), ("ped_radius", C.c_double),
                ("rp", ResampleParams), ("dist_raw", C.c_void_p), ("dist_S", C.c_int32), ("dist_dtype", C.c_int32)]


class LoopRequest(C.Structure):                                    # fot_loop_request
    _fields_ = [("ego", Ego), ("overrides", Overrides), ("target_speed", C.c_double),
                ("max_stop_distance", C.c_double), ("episode", C.c_int32), ("_pad", C.c_int32)]


class LoopConfig(C.Structure):                                     # fot_loop_config
    _fields_ = [(n, C.c_double) for n in
                ("dt", "target_speed", "max_accel", "emergency_decel", "clearance_caution", "clearance_emergency",
                 "trigger_clearance_caution", "trigger_time_headway", "envelope_decel", "envelope_standoff",
                 "caution_accel", "caution_speed", "caution_speed_mult", "emergency_accel", "emergency_lat_accel")] + \
               [("max_replan", C.c_int32), ("_pad", C.c_int32)]


class LoopStepOut(C.Structure):                                    # fot_loop_step_out
    _fields_ = [(n, C.c_void_p) for n in ("ego", "jerk", "state", "stats", "record", "keep", "cost", "before", "after",
                                          "s_now", "records")] + [("n_records", C.c_int32), ("_pad", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("n_inst", C.c_int32), ("obstacle_dtype", C.c_int32),
                ("ego", C.POINTER(Ego)), ("target_speed", C.POINTER(C.c_double)),
                ("overrides", C.POINTER(Overrides)), ("max_stop_distance", C.POINTER(C.c_double)),
                ("static_xy", C.c_void_p), ("static_off", C.POINTER(C.c_int32)),
                ("dyn_xy", C.c_void_p), ("dyn_off", C.POINTER(C.c_int64)), ("dyn_dims", C.POINTER(C.c_int32))]


RESULT_BYTES = C.sizeof(Result)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfot.so")

# every symbol include/fot.h declares
SYMBOLS = ["fot_version", "fot_abi_info", "fot_create", "fot_destroy", "fot_live_handles", "fot_last_error", "fot_set_path_waypoints",
           "fot_set_path_coeffs", "fot_get_path_coeffs", "fot_spline_eval", "fot_plan_batch",
           "fot_plan_batch_device", "fot_synchronize", "fot_frenet_state_batch", "fot_debug_candidates",
           "fot_debug_candidate_path", "fot_debug_margins", "fot_debug_set_eval_segments", "fot_debug_set_tile_cut", "fot_debug_time_info", "fot_check_collision_paths", "fot_check_paths", "fot_resample_n_dense", "fot_resample_predictions",
           "fot_predict_cv", "fot_safety_metrics_batch", "fot_loop_set_static", "fot_loop_plan", "fot_loop_observe", "fot_loop_observe_begin", "fot_loop_observe_end", "fot_loop_begin", "fot_loop_step", "fot_gather_paths", "fot_wire_n_total", "fot_wire_record_bytes",
           "fot_pack_records_device", "fot_pack_records_host", "fot_unpack_records", "fot_profile_enable", "fot_profile_read", "fot_profile_kernel_name"]
PROFILE_KERNELS = 3                      # FOT_PROFILE_KERNELS (include/fot.h)
ABI_VERSION = 4                          # FOT_ABI_VERSION
MAX_TI, MAX_TV, MAX_BRAKE, MAX_PRED_LEN = 64, 32, 32, 32
EGO_IS_FRENET = 3                        # FOT_EGO_IS_FRENET (fot_ego.has_prev_s)
MARGIN_GROUPS = 8                        # FOT_MARGIN_GROUPS
MARGIN_NAMES = ["speed", "accel", "curvature", "lat_accel", "road", "collision", "stop_filter", "structural"]

_lib = None


def abi_expectation():
    """What fot_abi_info() must report for THIS binding: version, structure sizes, array capacities (include/fot.h)."""
    return [ABI_VERSION, C.sizeof(Params), C.sizeof(Ego), C.sizeof(Overrides), C.sizeof(Result), C.sizeof(Batch),
            C.sizeof(ResampleParams), C.sizeof(Safety), C.sizeof(LoopFrame), C.sizeof(LoopRequest), C.sizeof(WireHeader),
            MAX_NT, MAX_CIRCLES, MAX_TI, MAX_TV, MAX_BRAKE, MAX_SAMPLES, MAX_PRED_LEN, PROFILE_KERNELS, MARGIN_GROUPS,
            C.sizeof(LoopConfig), C.sizeof(LoopStepOut)]


ABI_WORD_NAMES = ["FOT_ABI_VERSION", "sizeof(fot_params)", "sizeof(fot_ego)", "sizeof(fot_overrides)", "sizeof(fot_result)",
                  "sizeof(fot_batch)", "sizeof(fot_resample_params)", "sizeof(fot_safety)", "sizeof(fot_loop_frame)",
                  "sizeof(fot_loop_request)", "sizeof(fot_wire_header)", "FOT_MAX_NT", "FOT_MAX_CIRCLES", "FOT_MAX_TI",
                  "FOT_MAX_TV", "FOT_MAX_BRAKE", "FOT_MAX_SAMPLES", "FOT_MAX_PRED_LEN", "FOT_PROFILE_KERNELS",
                  "FOT_MARGIN_GROUPS", "sizeof(fot_loop_config)", "sizeof(fot_loop_step_out)"]


def _check_abi(L, path):
    """Refuse a library whose layouts differ from this binding's: an old libfot.so under new ctypes structures (or the
    other way round) writes past the caller's arrays instead of failing -- round 3's "double free or corruption" at
    process exit was exactly that (a library with four profile slots filling arrays of three)."""
    try:
        f = L.fot_abi_info
    except AttributeError:
        raise ImportError(f"{path} predates fot_abi_info (ABI {ABI_VERSION}): rebuild it with `make -C "
                          f"{os.path.join(_HERE, 'csrc')}`") from None
    f.argtypes = [C.c_int32, C.POINTER(C.c_int32)]
    f.restype = C.c_int32
    want = abi_expectation()
    got = (C.c_int32 * len(want))()
    n = f(len(want), got)
    bad = [f"{ABI_WORD_NAMES[i]}: library {got[i]}, binding {want[i]}" for i in range(min(n, len(want))) if got[i] != want[i]]
    if n != len(want):
        bad.append(f"the library reports {n} ABI words, the binding knows {len(want)}")
    if bad:
        raise ImportError(f"{path} does not match this binding (include/fot.h changed since it was built?): "
                          + "; ".join(bad))


# ---- handle lifetime: every owner of a libfot handle registers itself here; whatever is still open when the interpreter
# exits is closed by the atexit hook -- BEFORE module teardown and before the HIP runtime's own exit handlers, while
# streams, events and the context are intact.  (Round 3 let __del__ run fot_destroy during finalisation, or leaked the
# handle there; neither is deterministic.)
_live_owners = weakref.WeakSet()


def register_owner(obj):
    """``obj.close()`` will be called at interpreter exit unless it was closed (and unregistered) before."""
    _live_owners.add(obj)


def unregister_owner(obj):
    _live_owners.discard(obj)


def close_all():
    """Close every registered owner that is still open (the atexit hook; also callable from tests)."""
    for obj in list(_live_owners):
        try:
            obj.close()
        except Exception:
            pass
    _live_owners.clear()


atexit.register(close_all)


class FotError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfot error {code}: {msg}")
        self.code = code


hip_runtime_path = None                  # which HIP runtime lib() bound libfot to (diagnostics / tests)


def _needed_sonames(path):
    """DT_NEEDED entries of an ELF64 shared object (what the loader will ask for), read from the file itself."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return []
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    out = []
    for sec in secs:
        if sec[1] != 6:                                          # SHT_DYNAMIC
            continue
        strtab = secs[sec[6]]                                    # sh_link: its string table
        for off in range(sec[4], sec[4] + sec[5], 16):
            tag, val = struct.unpack_from("<qQ", data, off)
            if tag == 1:                                         # DT_NEEDED
                beg = strtab[4] + val
                out.append(data[beg:data.index(b"\0", beg)].decode())
            if tag == 0:
                break
    return out


def hip_soname():
    """The HIP runtime SONAME libfot.so was linked against (its DT_NEEDED), e.g. libamdhip64.so.7."""
    try:
        for n in _needed_sonames(LIB_PATH):
            if n.startswith("libamdhip64"):
                return n
    except OSError:
        pass
    return "libamdhip64.so"


def _bind_hip_runtime():
    """ONE HIP runtime per process, whatever the import order.

    PyTorch-ROCm ships its own HIP + HSA runtime in ``torch/lib`` and asks for it by FILE name (``libamdhip64.so``,
    found through its RPATH); libfot asks for a versioned SONAME (``hip_soname()``, e.g. ``libamdhip64.so.7``).  If libfot came first the loader would
    pick the system copy for it and later a second, different copy for torch -- two HSA runtimes in one process, and
    the second one finds no GPU.  So before libfot is opened: (1) a runtime that is already in the process is reused
    (the loader matches libfot's DT_NEEDED against its SONAME); (2) otherwise, when this interpreter has a torch
    installation, torch's copy is opened first -- ``import torch`` later resolves to the very same file; (3) otherwise
    the loader's normal search (the system ROCm) applies.  torch itself is NOT imported here."""
    global hip_runtime_path
    noload = getattr(os, "RTLD_NOLOAD", 4)
    for name in (hip_soname(), "libamdhip64.so"):
        try:
            C.CDLL(name, mode=noload | os.RTLD_NOW)
            hip_runtime_path = f"(already loaded: {name})"
            return
        except OSError:
            pass
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        tdir = list(spec.submodule_search_locations)[0] if spec and spec.submodule_search_locations else None
    except Exception:
        tdir = None
    if tdir:
        cand = os.path.join(tdir, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=os.RTLD_GLOBAL | os.RTLD_NOW)
            hip_runtime_path = cand
            return
    hip_runtime_path = "(loader default)"


def lib():
    """Load libfot.so (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc, gfx950). There is no CPU fallback.")
    _bind_hip_runtime()
    L = C.CDLL(LIB_PATH)
    _check_abi(L, LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.fot_version.restype = C.c_char_p
    L.fot_last_error.restype = C.c_char_p
    L.fot_last_error.argtypes = [vp]
    L.fot_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(vp)]
    L.fot_destroy.argtypes = [vp]
    L.fot_destroy.restype = None
    L.fot_live_handles.argtypes = []
    L.fot_live_handles.restype = C.c_int32
    L.fot_set_path_waypoints.argtypes = [vp, C.c_int32, dp, dp]
    L.fot_set_path_coeffs.argtypes = [vp, C.c_int32] + [dp] * 9
    L.fot_get_path_coeffs.argtypes = [vp, ip] + [dp] * 9
    L.fot_spline_eval.argtypes = [vp, C.c_int32] + [dp] * 6
    L.fot_plan_batch.argtypes = [vp, C.POINTER(Batch), C.POINTER(Result)]
    L.fot_plan_batch_device.argtypes = [vp, C.POINTER(Batch), vp, vp]
    L.fot_synchronize.argtypes = [vp]
    L.fot_frenet_state_batch.argtypes = [vp, C.c_int32, C.POINTER(Ego), dp, dp, dp, ip]
    L.fot_debug_candidates.argtypes = [vp, C.c_int32, C.c_int32, dp, ip, ip, ip]
    L.fot_check_collision_paths.argtypes = [vp, C.c_int32, ip, dp, dp, dp, dp, C.c_int32, dp,
                                            C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, ip]
    L.fot_debug_candidate_path.argtypes = [vp, C.c_int32, C.c_int32, dp, ip]
    L.fot_debug_margins.argtypes = [vp, C.c_int32, C.c_int32, dp]
    L.fot_debug_set_eval_segments.argtypes = [vp, C.c_int32]
    L.fot_debug_set_tile_cut.argtypes = [vp, C.c_int32]
    L.fot_debug_time_info.argtypes = [vp, C.c_double, ip, dp, dp]
    L.fot_check_paths.argtypes = [vp, C.c_int32, ip, ip] + [dp] * 9 + [C.POINTER(Overrides), C.c_double, C.c_int32, dp,
                                                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, ip]
    L.fot_resample_n_dense.argtypes = [C.POINTER(ResampleParams), C.c_int32]
    L.fot_resample_predictions.argtypes = [vp, C.POINTER(ResampleParams), C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32,
                                           dp, dp, C.c_double, vp, C.c_int32, C.c_int32, ip, dp, vp]
    L.fot_predict_cv.argtypes = [vp, C.POINTER(ResampleParams), C.c_int32, C.c_int32, vp, vp, C.c_int32, dp,
                                 C.c_double, vp, C.c_int32, C.c_int32, ip, vp]
    L.fot_safety_metrics_batch.argtypes = [vp, C.c_int32, dp, ip, dp, dp, C.c_double, C.c_double, C.c_int32, C.POINTER(Safety)]
    L.fot_loop_set_static.argtypes = [vp, C.c_int32, vp]
    L.fot_loop_plan.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
    L.fot_loop_observe.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
    L.fot_loop_observe_begin.argtypes = [vp, C.c_int32, vp, vp]
    L.fot_loop_observe_end.argtypes = [vp, vp, vp]
    L.fot_loop_begin.argtypes = [vp, C.c_int32, vp, vp]
    L.fot_loop_step.argtypes = [vp, vp, vp, vp]
    L.fot_gather_paths.argtypes = [vp, C.c_int32, vp, C.c_int32, vp]
    L.fot_wire_n_total.argtypes = [vp]
    L.fot_wire_record_bytes.argtypes = [C.c_int32]
    L.fot_pack_records_device.argtypes = [vp, C.c_int32, vp, vp, vp]
    L.fot_pack_records_host.argtypes = [C.c_int32, C.c_int32, vp, vp]
    L.fot_unpack_records.argtypes = [C.c_int32, C.c_int32, vp, vp]
    L.fot_profile_enable.argtypes = [vp, C.c_int]
    L.fot_profile_read.argtypes = [vp, C.c_int, C.c_int32, ip, dp]
    L.fot_profile_kernel_name.argtypes = [C.c_int]
    L.fot_profile_kernel_name.restype = C.c_char_p
    _lib = L
    return L


def check(handle, rc):
    if rc != OK:
        msg = lib().fot_last_error(handle)
        raise FotError(rc, msg.decode() if msg else "")
