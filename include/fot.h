/*
 * fot.h -- C ABI of libfot.so, the MI355X (gfx950) Frenet optimal-trajectory planner.
 *
 * Drop-in boundary for ONE hot path of mnhrk15/integrated_path_planning:
 * FrenetPlanner.plan() (reference src/planning/frenet_planner.py:227-304) and
 * everything it calls.  The reference is pure Python and has no FFI of its own;
 * each entry point below names the reference interface it replaces, and
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no exceptions across the boundary; every call returns FOT_OK (0)
 *     or a negative FOT_ERR_*; fot_last_error(h) has the message.
 *   - "no feasible trajectory" is NOT an error: fot_result.status says so
 *     (the reference returns None, frenet_planner.py:294-304).
 *   - the caller owns every buffer it passes; the library neither keeps nor
 *     modifies inputs (reference: integrated_simulator.py:698 passes copies).
 *   - one handle = one GPU + one stream + its own workspace.  Handles share no
 *     state, so one handle per host thread / per rank is safe; a single handle
 *     is not thread-safe, and its calls execute in the order they were made even
 *     when they are enqueued on different caller streams: every enqueue first waits
 *     (on the device) for the handle's previous one, because they share the workspace
 *     and the scratch buffers.  To keep several batches in flight use one handle per
 *     batch in flight.  (The reference planner is not thread-safe either: it carries
 *     _last_kappa and converter._prev_s, frenet_planner.py:218, coordinate_converter.py:283;
 *     here that state is explicit in fot_ego / fot_result.)
 *   - all arithmetic that decides a candidate's status is float64, like the reference.
 */
#ifndef FOT_H
#define FOT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOT_MAX_NT 256       /* samples per candidate: round(max_t/dt)+1 must be <= 256 (dt = 0.02 s at max_t = 5 s: 251);
                                also the stride of the 15 path arrays of fot_result -- the library only ever touches the
                                first round(max_t/dt)+1 entries of each */
#define FOT_MAX_CIRCLES 8    /* ego footprint circles (footprint.py:26) */
#define FOT_MAX_TI 64        /* time horizons  int((max_t-min_t)/dt)+1  (min_t = 1 s at max_t = 5 s, dt = 0.1 s: 41) */
#define FOT_MAX_TV 32        /* terminal speeds per horizon */
#define FOT_MAX_BRAKE 32     /* brake-ladder entries 0.5 s, 1.0 s, ... < min_t (frenet_planner.py:475): min_t <= 16.4 s */
#define FOT_MAX_SAMPLES 64   /* prediction samples S of a distribution */

/* error codes */
#define FOT_OK 0
#define FOT_ERR_INVALID (-1)      /* bad argument */
#define FOT_ERR_UNSUPPORTED (-2)  /* configuration exceeds a FOT_MAX_* limit */
#define FOT_ERR_HIP (-3)          /* HIP runtime failure (no device, out of memory, ...) */
#define FOT_ERR_NO_PATH_SET (-4)  /* fot_set_path_* not called yet */

/* candidate status == index into fot_result.stats[]; order of the reference's
 * last_check_stats dict (frenet_planner.py:910-918, 324) */
enum {
    FOT_ST_SPEED = 0, FOT_ST_ACCEL = 1, FOT_ST_CURVATURE = 2, FOT_ST_LAT_ACCEL = 3,
    FOT_ST_ROAD = 4, FOT_ST_COLLISION = 5, FOT_ST_OK = 6, FOT_ST_STOP_DISTANCE = 7,
    FOT_ST_DROPPED = 8           /* silently dropped, not counted (frenet_planner.py:933-956) */
};

/* fot_result.status */
enum {
    FOT_PLAN_OK = 0,             /* a path was selected */
    FOT_PLAN_NO_PATH = 1,        /* plan() would return None after the checks */
    FOT_PLAN_C2F_FAILED = 2      /* plan() would return None at frenet_planner.py:266-268 */
};

/* obstacle element type */
enum { FOT_F32 = 0, FOT_F64 = 1 };
/* fot_batch.dyn_dims[i][0] */
enum { FOT_DYN_NONE = 0, FOT_DYN_SINGLE = 1, FOT_DYN_DISTRIBUTION = 2 };
/* OR-ed into dyn_dims[i][0]: instance i's tensor is laid out [T][S][P][2] (all pedestrians of one time row
 * contiguous: what the broad phase reads, fully coalesced) instead of the reference's [S][P][T][2].
 * fot_resample_predictions / fot_predict_cv write it when asked to (FOT_OUT_TMAJOR). */
#define FOT_DYN_LAYOUT_TSP 0x10

/* replaces the constructor arguments of FrenetPlanner (frenet_planner.py:149-210) */
typedef struct fot_params {
    double max_speed, max_accel, max_curvature, max_lat_accel;
    double dt, d_road_w, max_road_width;
    double robot_radius, obstacle_radius;
    double min_t, max_t, d_t_s;
    double k_j, k_t, k_d, k_s_dot, k_lat, k_lon;
    double chance_epsilon, collision_margin_inflation;
    int32_t n_circles;           /* 0: single circle of robot_radius; else EgoFootprint (footprint.py:14-45) */
    int32_t _pad;
    double footprint_radius;
    double footprint_offsets[FOT_MAX_CIRCLES];
} fot_params;

/* replaces EgoVehicleState (data_structures.py:32-51) + the planner's cross-call state */
typedef struct fot_ego {
    double x, y, yaw, v, a;
    double last_kappa;           /* FrenetPlanner._last_kappa */
    double prev_s;               /* CoordinateConverter._prev_s */
    int32_t has_prev_s;          /* 0: first call (global nearest-point search); 1: prev_s valid;
                                    2 (FOT_PREV_S_CHAINED): prev_s := new_prev_s of the PREVIOUS instance of the
                                    batch, i.e. this instance is the next plan() call on the same planner object
                                    (the escalation retries of integrated_simulator.py:602-644 in one launch);
                                    3 (FOT_EGO_IS_FRENET): the record IS a Frenet state -- x, y, yaw, v, a, last_kappa
                                    hold s, s_d, s_dd, d, d_d, d_dd -- and the lattice is generated from it as given
                                    (what the reference's tests do through _generate_frenet_paths(FrenetState, ...),
                                    frenet_planner.py:376); no nearest-point search, new_prev_s comes back NaN */
    int32_t _pad;
} fot_ego;
#define FOT_PREV_S_CHAINED 2
#define FOT_EGO_IS_FRENET 3

/* replaces constraint_overrides (frenet_planner.py:921-930); NaN = key absent */
typedef struct fot_overrides {
    double max_speed, max_accel, max_curvature, max_lat_accel;
} fot_overrides;

/* replaces the returned FrenetPath (data_structures.py:149-220) + last_check_stats + state updates */
typedef struct fot_result {
    int32_t status;              /* FOT_PLAN_* */
    int32_t best_index;          /* candidate index in generation order (Ti -> tv -> di, brake ladder last); -1 */
    int32_t n_cand;              /* candidates generated */
    int32_t n_keep;              /* samples in the path arrays below */
    double cost;
    int32_t stats[8];            /* last_check_stats, FOT_ST_* order */
    int32_t stats_valid;         /* 0 when the reference leaves last_check_stats = None */
    int32_t _pad;
    double new_last_kappa;       /* value of _last_kappa after the call */
    double new_prev_s;           /* value of converter._prev_s after the call */
    double frenet0[6];           /* s, s_d, s_dd, d, d_d, d_dd of the ego (frenet_planner.py:371) */
    double ref0[6];              /* rs, rx, ry, rtheta, rkappa, rdkappa (coordinate_converter.py:308) */
    /* path arrays: entries [0, n_keep) hold the path, [n_keep, n_total) are written as zero, entries from
     * n_total = round(max_t/dt)+1 on are NEVER touched, neither in a device-resident record nor in the caller's host
     * record (a caller who compares whole records zero-fills its buffer once) */
    double t[FOT_MAX_NT], s[FOT_MAX_NT], s_d[FOT_MAX_NT], s_dd[FOT_MAX_NT], s_ddd[FOT_MAX_NT];
    double d[FOT_MAX_NT], d_d[FOT_MAX_NT], d_dd[FOT_MAX_NT], d_ddd[FOT_MAX_NT];
    double x[FOT_MAX_NT], y[FOT_MAX_NT], yaw[FOT_MAX_NT], v[FOT_MAX_NT], a[FOT_MAX_NT], c[FOT_MAX_NT];
} fot_result;

/* One batch of independent ego/scenario instances = the arguments of n_inst
 * plan() calls (frenet_planner.py:227-236).  The small per-instance arrays and
 * the shape metadata are ALWAYS host memory; the obstacle coordinates (and the
 * results) are host memory for fot_plan_batch and device memory for
 * fot_plan_batch_device. */
typedef struct fot_batch {
    int32_t n_inst;
    int32_t obstacle_dtype;              /* FOT_F32 | FOT_F64 */
    const fot_ego *ego;                  /* [n_inst] host */
    const double *target_speed;          /* [n_inst] host */
    const fot_overrides *overrides;      /* [n_inst] host, or NULL */
    const double *max_stop_distance;     /* [n_inst] host, NaN = None; or NULL */
    /* static_obstacles [Ns,2] per instance, concatenated; instance i owns points [static_off[i], static_off[i+1]) */
    const void *static_xy;               /* host | device */
    const int32_t *static_off;           /* [n_inst+1] host, or NULL (no static obstacles) */
    /* dynamic_obstacles [P,T,2] / dynamic_obstacles_distribution [S,P,T,2] per instance, concatenated;
     * instance i starts at point dyn_off[i]; dyn_dims[i] = {mode, S, P, T} (S = 1 for FOT_DYN_SINGLE).
     * Non-finite coordinates never hit, and a pedestrian whose track holds a NaN anywhere is no obstacle at ANY
     * time step, as in the reference (np.min / np.max in its box pre-filter, frenet_planner.py:1211-1219): the
     * library finds such tracks itself, whoever produced the tensor. */
    const void *dyn_xy;                  /* host | device */
    const int64_t *dyn_off;              /* [n_inst] host, or NULL (no dynamic obstacles) */
    const int32_t *dyn_dims;             /* [n_inst][4] host */
} fot_batch;

typedef struct fot_handle fot_handle;

const char *fot_version(void);

/* What the library was BUILT with, for a binding to check before its first real call: a binding whose structure layouts
 * or array capacities differ from the library's corrupts memory instead of failing (round 3: an older libfot.so with
 * four profile slots under a binding that allocated three aborted the process at exit with "double free or
 * corruption").  out[i], i < cap: FOT_ABI_VERSION, sizeof of fot_params, fot_ego, fot_overrides, fot_result, fot_batch,
 * fot_resample_params, fot_safety, fot_loop_frame, fot_loop_request, fot_wire_header, then FOT_MAX_NT, FOT_MAX_CIRCLES,
 * FOT_MAX_TI, FOT_MAX_TV, FOT_MAX_BRAKE, FOT_MAX_SAMPLES, FOT_MAX_PRED_LEN, FOT_PROFILE_KERNELS, FOT_MARGIN_GROUPS,
 * sizeof of fot_loop_config, fot_loop_step_out.
 * Returns the number of words the library knows (FOT_ABI_INFO_WORDS of ITS header). */
#define FOT_ABI_VERSION 4
#define FOT_ABI_INFO_WORDS 22
int32_t fot_abi_info(int32_t cap, int32_t *out);

/* FrenetPlanner.__init__ (frenet_planner.py:149-225).  device < 0: current device. */
int fot_create(const fot_params *params, int device, fot_handle **out);
/* Frees the handle.  Idempotent (a pointer fot_create did not return, or one already destroyed, is ignored) and
 * bounded: the handle's streams and the event behind its last enqueue are polled for at most FOT_DESTROY_TIMEOUT_MS
 * (default 5000); if they do not drain, or the HIP runtime is already shutting down, device and pinned memory are left
 * to the process teardown instead of being freed under running work.  Never blocks on a caller's stream. */
void fot_destroy(fot_handle *h);
/* handles created and not yet destroyed in this process (what a binding's exit hook still has to close) */
int32_t fot_live_handles(void);
const char *fot_last_error(const fot_handle *h);   /* h may be NULL: error of the last failed fot_create */

/* reference_path: CubicSpline2D(waypoints) (cubic_spline.py:190-213) built natively ... */
int fot_set_path_waypoints(fot_handle *h, int32_t n, const double *wx, const double *wy);
/* ... or adopted verbatim from an existing CubicSpline2D object: knots s[n] and the
 * CubicSpline1D coefficient arrays a[n], b[n-1], c[n], d[n-1] of sx and sy (cubic_spline.py:30-45) */
int fot_set_path_coeffs(fot_handle *h, int32_t n, const double *s,
                        const double *ax, const double *bx, const double *cx, const double *dx,
                        const double *ay, const double *by, const double *cy, const double *dy);
/* read the spline back (same array sizes); n_out receives the knot count; arrays may be NULL */
int fot_get_path_coeffs(const fot_handle *h, int32_t *n_out, double *s,
                        double *ax, double *bx, double *cx, double *dx,
                        double *ay, double *by, double *cy, double *dy);
/* CubicSpline2D.calc_position/calc_yaw/calc_curvature/calc_curvature_rate (cubic_spline.py:215-288)
 * evaluated on the device; NaN outside the domain.  Host arrays of n doubles. */
int fot_spline_eval(fot_handle *h, int32_t n, const double *s, double *x, double *y,
                    double *yaw, double *kappa, double *dkappa);

/* n_inst x FrenetPlanner.plan() with host-resident obstacles and results; synchronous */
int fot_plan_batch(fot_handle *h, const fot_batch *batch, fot_result *out);
/* same with device-resident obstacle coordinates and a device-resident fot_result[n_inst];
 * enqueued on `stream` (a hipStream_t; NULL = the handle's own stream) and NOT synchronised */
int fot_plan_batch_device(fot_handle *h, const fot_batch *batch, fot_result *out_dev, void *stream);
/* block until everything the handle enqueued on its own stream has finished */
int fot_synchronize(fot_handle *h);

/* FrenetPlanner._cartesian_to_frenet_state (frenet_planner.py:334-374) for n egos.
 * frenet[n][6], ref[n][6], new_prev_s[n], ok[n] (1 = converted) -- host arrays */
int fot_frenet_state_batch(fot_handle *h, int32_t n, const fot_ego *ego,
                           double *frenet, double *ref, double *new_prev_s, int32_t *ok);

/* Per-candidate table of instance `inst` of the most recent plan call on this handle
 * (what _check_paths put in each list, frenet_planner.py:932-991).  Arrays of `cap`
 * entries, any may be NULL; returns the number of candidates or a negative error. */
int fot_debug_candidates(fot_handle *h, int32_t inst, int32_t cap, double *cost,
                         int32_t *status, int32_t *keep, int32_t *n_t);

/* All 15 FrenetPath arrays of candidate `index` of instance `inst` of the most recent plan call, BEFORE
 * truncation (what _generate_frenet_paths + _calc_global_paths produce, frenet_planner.py:376-503, 736-889):
 * arrays[15][FOT_MAX_NT] in fot_result field order, *n_t = generated samples. */
int fot_debug_candidate_path(fot_handle *h, int32_t inst, int32_t index, double *arrays, int32_t *n_t);

/* Epsilon-band report for instance `inst` of the most recent plan call: per candidate and per group of decisions the
 * smallest relative distance |value - threshold| / |threshold| of every comparison made on it (speed :964, accel :966,
 * curvature incl. the 0.5 m/s gate and the low-speed rules :968/:995-1033, lateral acceleration :975, road :982,
 * collision radius :1198/:1233 against the obstacles the broad phase kept, stop filter :307-324, structural:
 * singularity :826, EPS_S_DOT :792, step length :955).  margins[cap][FOT_MARGIN_GROUPS]; +inf = no such decision.
 * A status that differs from the reference's with all margins far above float64 rounding is a logic error; a margin
 * at rounding level marks a decision that a re-association may flip.  Returns the number of candidates. */
#define FOT_MARGIN_GROUPS 8
int fot_debug_margins(fot_handle *h, int32_t inst, int32_t cap, double *margins);

/* Test hook.  A plan call of a few egos cuts every candidate's time range into up to 4 segments evaluated by
 * different waves and merged (the decisions are the same: every per-candidate accumulator of _check_paths /
 * _path_is_collision_free merges associatively); large batches walk it in one piece.  n_seg = 1..4 forces the number
 * of segments for the handle's later plan calls, 0 restores the choice by batch size. */
int fot_debug_set_eval_segments(fot_handle *h, int32_t n_seg);

/* Test hook.  How the handle cuts a lattice into tiles (the unit of work of the evaluation kernel): 0 = chosen by
 * the lattice (default), 1 = per-wave rows (k_evaluate: every wave stages the rows of its own tile), 2 = groups
 * (k_evaluate_group: four tiles share one row table).  Same decisions, byte-identical records either way; the GPU
 * tests run every reference case under both cuts and under 1..4 time segments.  Rebuilds the handle's tile table
 * (synchronises the device); applies to the handle's later plan calls. */
int fot_debug_set_tile_cut(fot_handle *h, int32_t cut);

/* FrenetPlanner._build_time_cache (frenet_planner.py:586-617) as the library holds it for a horizon of `time`
 * seconds: the sample count n_t = round(time / dt) + 1 and the closed-form inverses of the quartic / quintic
 * boundary-value matrices (row-major 2x2 and 3x3) that every lattice polynomial is solved with.  Host only. */
int fot_debug_time_info(const fot_handle *h, double time, int32_t *n_t, double *quartic_inv4, double *quintic_inv9);

/* FrenetPlanner._path_is_collision_free (frenet_planner.py:1035-1233) for n_paths externally
 * supplied paths against ONE obstacle set.  x, y, yaw, t: [n_paths][FOT_MAX_NT] host, len[n_paths];
 * static_xy [n_static][2] double host; dyn [S][P][T][2] double host with mode as in dyn_dims.
 * free_out[n_paths]: 1 = collision free. */
int fot_check_collision_paths(fot_handle *h, int32_t n_paths, const int32_t *len,
                              const double *x, const double *y, const double *yaw, const double *t,
                              int32_t n_static, const double *static_xy,
                              int32_t mode, int32_t S, int32_t P, int32_t T, const double *dyn,
                              int32_t *free_out);

/* FrenetPlanner._check_paths (frenet_planner.py:891-993) followed by _apply_stop_distance_filter (:307-324)
 * for n_paths externally supplied paths: arrays [n_paths][FOT_MAX_NT] host (yaw, d, s may be NULL = zeros),
 * len[n_paths], flags[n_paths] (bit0: x/y/yaw/s/d all present -> low-speed curvature rules apply, :1017-1022;
 * bit1: d present -> road-corridor test applies; NULL = both), overrides may be NULL, max_stop_distance NaN = None,
 * obstacle set as in fot_check_collision_paths.  status_out[n_paths]: FOT_ST_* (FOT_ST_OK = 'ok', FOT_ST_DROPPED =
 * silently skipped). */
int fot_check_paths(fot_handle *h, int32_t n_paths, const int32_t *len, const int32_t *flags,
                    const double *x, const double *y, const double *yaw, const double *v, const double *a,
                    const double *c, const double *d, const double *s, const double *t,
                    const fot_overrides *overrides, double max_stop_distance,
                    int32_t n_static, const double *static_xy,
                    int32_t mode, int32_t S, int32_t P, int32_t T, const double *dyn, int32_t *status_out);

/* ---- the obstacle-tensor producer in front of the planner ------------------------------
 * TrajectoryPredictor.process_prediction (trajectory_predictor.py:233-313) for S prediction samples, fused
 * with the current-position prepend of IntegratedSimulator._update_prediction (integrated_simulator.py:503-525):
 * raw Social-GAN predictions (0.4 s grid, anchored at the last observation) -> the planner's [S][P][T][2]
 * obstacle tensor on the dt grid, written where fot_plan_batch_device reads it (no host round trip).
 *   pred    [S][pred_len][P][2] (pred_dtype), host or device memory according to on_device
 *   anchor  [P][2] host, or NULL (no anchor point);  current [P][2] host, or NULL (no prepend)
 *   out     [S][P][T][2] (out_dtype), same memory space as pred; *T_out = n_dense (+1 with current)
 *   sample_dist [S] host or NULL: sum over (p, k) of |sample - sample mean|, whose first minimum is
 *               predict_single_best's representative sample (:346-351); requesting it synchronises
 * pred_len <= FOT_MAX_PRED_LEN, T <= FOT_MAX_NT.  stream: NULL = the handle's stream. */
#define FOT_MAX_PRED_LEN 32
typedef struct fot_resample_params {
    double sgan_dt, sim_dt, plan_horizon;
} fot_resample_params;
/* bits of the `on_device` argument below */
#define FOT_OUT_DEVICE 1     /* pred / out are device memory (else host memory) */
#define FOT_OUT_TMAJOR 2     /* out is written [T][S][P][2] (FOT_DYN_LAYOUT_TSP) instead of [S][P][T][2] */
int fot_resample_n_dense(const fot_resample_params *rp, int32_t pred_len);
int fot_resample_predictions(fot_handle *h, const fot_resample_params *rp, int32_t S, int32_t pred_len, int32_t P,
                             const void *pred, int32_t pred_dtype, const double *anchor, const double *current,
                             double staleness, void *out, int32_t out_dtype, int32_t on_device, int32_t *T_out,
                             double *sample_dist, void *stream);
/* TrajectoryPredictor.predict_cv (:188-231): obs_last / obs_prev [P][2] host (obs_prev NULL = zero velocity)
 * -> out [P][T][2] with the same prepend / memory-space conventions.  obs_dtype FOT_F32: the observations are float32
 * (what PedestrianObserver.get_observation hands over, observer.py:134) and the velocity is formed in float32 as NumPy
 * does for float32 arrays; FOT_F64: float64 observations, float64 velocity. */
int fot_predict_cv(fot_handle *h, const fot_resample_params *rp, int32_t pred_len, int32_t P,
                   const void *obs_last, const void *obs_prev, int32_t obs_dtype, const double *current,
                   double staleness, void *out, int32_t out_dtype, int32_t on_device, int32_t *T_out, void *stream);

/* ---- compute_safety_metrics_static (data_structures.py:301-388) for n egos in one launch ----
 * ego [n][4] = x, y, yaw, v; pedestrians of ego i: ped_pos / ped_vel [ped_off[i] .. ped_off[i+1])[2]  (host arrays).
 * use_footprint != 0: the handle's multi-circle footprint (footprint= argument of the reference); otherwise, or when the
 * handle has none, the single centre circle of ego_radius. */
typedef struct fot_safety {
    double min_distance, ttc, clearance, clearance_ahead;
    int32_t collision;
    int32_t _pad;
} fot_safety;
int fot_safety_metrics_batch(fot_handle *h, int32_t n, const double *ego, const int32_t *ped_off,
                             const double *ped_pos, const double *ped_vel, double ego_radius, double ped_radius,
                             int32_t use_footprint, fot_safety *out);

/* ---- the device work of one closed-loop step (IntegratedSimulator.step, integrated_simulator.py:678-747)
 * of n episodes in two calls with ONE synchronisation each; the prediction tensor never leaves HBM.
 *
 * fot_loop_plan, with a frame: _update_prediction's constant-velocity branch (:424-527 -> trajectory_predictor.py
 * :188-231) for the pedestrians of all episodes, written into the handle's own tensor -- per episode a [P_e][T_e][2]
 * float64 block, T_e = n_dense + prepend[e] -- and compute_safety_metrics_static on the current ego states (:529-560);
 * then n_req plan() calls (:562-600), request j against the block of episode req[j].episode.  Without a frame
 * (NULL) the requests run against the tensor of the previous call: the escalation retries of one step (:602-644).
 * *records points at n_req records in pinned host memory owned by the handle, valid until the next fot_loop_plan.
 *
 * fot_loop_observe: the metrics of the new ego states against the same frame's pedestrians (:864-870) and the arc
 * length of the nearest point of the reference path (the goal test's converter, :873-883), ego i = episode i:
 * ego5 [n][5] = x, y, yaw, v, a; prev_s [n], NaN = no cached arc length.  _begin enqueues the two launches and returns,
 * _end waits and hands the results over (any may be NULL); no other call on the handle in between.
 *
 * fot_loop_set_static: the static obstacle points every request of the loop sees (kept in HBM, one copy per request). */
typedef struct fot_loop_frame {
    int32_t n_episodes;
    int32_t pred_len;               /* of the predictor (trajectory_predictor.py:188) */
    int32_t use_footprint;          /* as fot_safety_metrics_batch */
    int32_t _pad;
    const int32_t *ped_off;         /* [n_episodes + 1]: pedestrians of episode e = rows [ped_off[e], ped_off[e+1]) */
    const double *ped_pos, *ped_vel;    /* [sum P][2] current positions / velocities */
    const float *obs_last, *obs_prev;   /* [sum P][2] the observer's last two samples; obs_last NULL: the predictor is
                                           not ready and the tensor is the current positions alone (T = 1, :495-498) */
    const uint8_t *prepend;         /* [n_episodes] 1: the current positions lead the episode's tracks (:503-511) */
    const double *ego;              /* [n_episodes][4] x, y, yaw, v for the metrics; NULL = no metrics */
    double staleness;               /* time since the observer's last sample (:463-470) */
    double ego_radius, ped_radius;
    fot_resample_params rp;
    /* Distribution-aware planning (integrated_simulator.py:459-460, 514-525, 622-630) without a host round trip: the raw
     * samples of a multi-sample predictor for ALL the frame's pedestrians, [dist_S][pred_len][sum P][2] in DEVICE memory
     * (what dist_S Social-GAN forward passes on PyTorch-ROCm leave), anchored at obs_last.  They are resampled on the
     * device (process_prediction, trajectory_predictor.py:233-313) straight into the handle's tensor, per episode a
     * [dist_S][P_e][n_dense + 1][2] block led by the current positions in EVERY sample, and every request of the step
     * plans against its episode's whole distribution under the handle's chance constraint.  NULL: the constant-velocity
     * predictor above.  dist_dtype FOT_F32 | FOT_F64; dist_S <= FOT_MAX_SAMPLES. */
    const void *dist_raw;
    int32_t dist_S, dist_dtype;
} fot_loop_frame;
typedef struct fot_loop_request {
    fot_ego ego;
    fot_overrides overrides;
    double target_speed;
    double max_stop_distance;       /* NaN = None */
    int32_t episode;                /* whose pedestrians */
    int32_t _pad;
} fot_loop_request;
int fot_loop_set_static(fot_handle *h, int32_t n_points, const double *xy);
int fot_loop_plan(fot_handle *h, const fot_loop_frame *frame, int32_t n_req, const fot_loop_request *req,
                  fot_safety *safety_out, const fot_result **records);
int fot_loop_observe(fot_handle *h, int32_t n, const double *ego5, const double *prev_s,
                     fot_safety *safety_out, double *new_prev_s);
int fot_loop_observe_begin(fot_handle *h, int32_t n, const double *ego5, const double *prev_s);
int fot_loop_observe_end(fot_handle *h, fot_safety *safety_out, double *new_prev_s);
/* The WHOLE lock step behind one call: the episodes' state lives in the handle (ego, the planner's nearest-point cache and
 * last curvature, the fail-safe state machine), a step is
 *   prediction + current metrics + the level-0 plan() of every running episode (fot_loop_plan with the frame)
 *   -> every further escalation level of the episodes whose first attempt failed, in one more launch
 *   -> the reference's retry loop replayed on the records (integrated_simulator.py:576-653, state_machine.py:116-247)
 *   -> ego update from the selected path's sample 1, or the emergency stop (:655-676, :749-802)
 *   -> metrics of the new states + the goal test's nearest point (fot_loop_observe).
 * fot_loop_config: IntegratedSimulator's and FailSafeStateMachine's constants as the reference resolves them from its
 * configuration (state_machine.py:32-98); emergency_decel NaN = 2 x max_accel.
 * fot_loop_begin: n_episodes slots, ego5 [n][5] = x, y, yaw, v, a; every machine NORMAL, no caches.
 * fot_loop_step: frame as for fot_loop_plan (its `ego` is ignored: the handle knows the egos), episode[i] = the slot
 * of the frame's episode i (distinct); out arrays of frame->n_episodes entries, any may be NULL.  out->records: the
 * step's records in pinned memory owned by the handle (level 0 of episode i = record i, escalation levels behind),
 * valid until the next loop call; out->record[i]: the record whose path episode i follows (-1: emergency stop). */
typedef struct fot_loop_config {
    double dt, target_speed, max_accel, emergency_decel;
    double clearance_caution, clearance_emergency;               /* recovery thresholds (combined radii subtracted) */
    double trigger_clearance_caution, trigger_time_headway;      /* preventive escalation */
    double envelope_decel, envelope_standoff;                    /* speed envelope on the clearance ahead */
    double caution_accel, caution_speed, caution_speed_mult;     /* CAUTION: overrides, target speed factor */
    double emergency_accel, emergency_lat_accel;                 /* EMERGENCY: overrides */
    int32_t max_replan;                                          /* integrated_simulator.py:383 */
    int32_t _pad;
} fot_loop_config;
typedef struct fot_loop_step_out {
    double *ego;                /* [n][5] the new ego states */
    double *jerk;               /* [n] */
    int32_t *state;             /* [n] 0 / 1 / 2 = NORMAL / CAUTION / EMERGENCY after the step */
    int32_t *stats;             /* [n][8] last_check_stats of the last plan() of the step; a row of -1: None */
    int32_t *record;            /* [n] */
    int32_t *keep;              /* [n] samples of the followed path (0: none) */
    double *cost;               /* [n] */
    fot_safety *before, *after; /* [n] metrics of the current / of the new ego states */
    double *s_now;              /* [n] arc length of the new state's nearest path point */
    const fot_result *records;  /* out */
    int32_t n_records;          /* out */
    int32_t _pad;
} fot_loop_step_out;
int fot_loop_begin(fot_handle *h, int32_t n_episodes, const fot_loop_config *cfg, const double *ego5);
int fot_loop_step(fot_handle *h, const fot_loop_frame *frame, const int32_t *episode, fot_loop_step_out *out);

/* Host utility (no GPU): the first kmax samples of the 15 path arrays of records[index[i]], i < n, as one dense block
 * out[15][n][kmax] in fot_result array order (t .. c) -- what a history keeps of a step's records. */
int fot_gather_paths(const fot_result *records, int32_t n, const int32_t *index, int32_t kmax, double *out);

/* ---- compact wire form of the records, for the all-gather of selected paths across GPUs ------------
 * fot_result is the host view (float64, FOT_MAX_NT slots per array: 15 536 bytes).  On the wire a record is
 *   fot_wire_header (176 bytes) | float path[15][n_total] (fot_result array order t .. c) | padding to 256 bytes
 * e.g. 3 328 bytes at n_total = 51.  The header keeps cost, the state updates and the Frenet start state in float64;
 * the path samples travel as float32, s / x / y as OFFSETS from the record's own start (frenet0[0], ref0[1], ref0[2]):
 * 2^-24 of the path's length (4e-6 m at 70 m) at any world coordinate, inside the 1e-5 the north star allows; the
 * other arrays are small numbers (d, speeds, curvature).  Samples past n_keep are zero.
 * n_total = round(max_t / dt) + 1 of the planner (fot_wire_n_total).  Pack / unpack on the host are pure format
 * conversions (no GPU); fot_pack_records_device converts device-resident records on `stream`. */
typedef struct fot_wire_header {
    int32_t status, best_index, n_cand, n_keep;
    double cost;
    int32_t stats[8];
    int32_t stats_valid, n_total;
    double new_last_kappa, new_prev_s;
    double frenet0[6], ref0[6];
} fot_wire_header;
int32_t fot_wire_n_total(const fot_handle *h);
int32_t fot_wire_record_bytes(int32_t n_total);
int fot_pack_records_device(fot_handle *h, int32_t n, const fot_result *records_dev, void *wire_dev, void *stream);
int fot_pack_records_host(int32_t n_total, int32_t n, const fot_result *records, void *wire);
int fot_unpack_records(int32_t n_total, int32_t n, const void *wire, fot_result *records);

/* ---- measurement (no reference counterpart: the reference times plan() with perf_counter,
 *      integrated_simulator.py:575-585) ----
 * With profiling on, every kernel launch of a plan call is bracketed by HIP events on the
 * stream it is launched on.  fot_profile_read waits for the recorded work, then returns, per
 * kernel, the number of launches and the summed device time in ms since the last reset. */
#define FOT_PROFILE_KERNELS 3
int fot_profile_enable(fot_handle *h, int on);
/* launches / total_ms: arrays of `cap` entries (entries past cap are not written); returns FOT_PROFILE_KERNELS of the
 * library, or a negative error */
int fot_profile_read(fot_handle *h, int reset, int32_t cap, int32_t *launches, double *total_ms);
const char *fot_profile_kernel_name(int index);

#ifdef __cplusplus
}
#endif
#endif /* FOT_H */
