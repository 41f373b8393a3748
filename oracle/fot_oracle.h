/*
 * fot_oracle.h -- CPU oracle for the Frenet optimal-trajectory planner hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, float64, single-thread
 * restatement of the reference algorithm (mnhrk15/integrated_path_planning,
 * src/planning/frenet_planner.py and friends).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it, and only as the
 * checker.  The product path (integrated_path_planning_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * Parity: PINNED.  tests/golden/ holds vectors produced by importing the
 * reference planner in the build container (tests/golden/make_golden.py);
 * tests/test_oracle_golden.py checks every stage of this file against them.
 */
#ifndef FOT_ORACLE_H
#define FOT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_NT 256         /* samples per candidate (round(max_t/dt)+1) */
#define ORC_MAX_CIRCLES 8

/* candidate status == index into stats[] (order of the reference's path_dict,
 * frenet_planner.py:910-918, + 'stop_distance_error' :324) */
enum {
    ORC_ST_SPEED = 0, ORC_ST_ACCEL = 1, ORC_ST_CURV = 2, ORC_ST_LATACC = 3,
    ORC_ST_ROAD = 4, ORC_ST_COLLISION = 5, ORC_ST_OK = 6, ORC_ST_STOPDIST = 7,
    ORC_ST_DROPPED = 8          /* silently dropped (:933-956), not counted */
};

enum { ORC_PLAN_OK = 0, ORC_PLAN_NO_PATH = 1, ORC_PLAN_C2F_FAILED = 2 };

typedef struct orc_params {
    double max_speed, max_accel, max_curvature, max_lat_accel;
    double dt, d_road_w, max_road_width;
    double robot_radius, obstacle_radius;
    double min_t, max_t, d_t_s;
    double k_j, k_t, k_d, k_s_dot, k_lat, k_lon;
    double chance_epsilon, collision_margin_inflation;
    int n_circles;              /* 0 => single circle of robot_radius */
    int _pad;
    double footprint_radius;
    double footprint_offsets[ORC_MAX_CIRCLES];
} orc_params;

typedef struct orc_spline orc_spline;

typedef struct orc_ego {
    double x, y, yaw, v, a;
    double last_kappa;          /* FrenetPlanner._last_kappa */
    double prev_s;              /* CoordinateConverter._prev_s */
    int has_prev_s;
    int _pad;
} orc_ego;

/* NaN = not overridden (frenet_planner.py:921-930) */
typedef struct orc_overrides {
    double max_speed, max_accel, max_curvature, max_lat_accel;
} orc_overrides;

/* obstacle set of ONE instance; coordinates are float64 (callers holding
 * float32 data widen it first -- exact) */
typedef struct orc_obstacles {
    const double *static_xy;    /* [n_static][2] */
    int n_static;
    int dyn_mode;               /* 0 none, 1 single [P][T][2], 2 distribution [S][P][T][2] */
    const double *dyn;
    int S, P, T;
} orc_obstacles;

typedef struct orc_result {
    int status;                 /* ORC_PLAN_* */
    int best_index;             /* candidate index in generation order, -1 if none */
    int n_cand;
    int n_keep;                 /* samples of the selected path */
    double cost;
    int stats[8];
    int stats_valid;            /* 0 when plan() returned before _check_paths (last_check_stats None) */
    int _pad;
    double new_last_kappa;
    double new_prev_s;
    double frenet0[6];          /* s, s_d, s_dd, d, d_d, d_dd */
    double ref0[6];             /* rs, rx, ry, rtheta, rkappa, rdkappa */
    /* selected path, 15 arrays in FrenetPath field order */
    double t[ORC_MAX_NT], s[ORC_MAX_NT], s_d[ORC_MAX_NT], s_dd[ORC_MAX_NT], s_ddd[ORC_MAX_NT];
    double d[ORC_MAX_NT], d_d[ORC_MAX_NT], d_dd[ORC_MAX_NT], d_ddd[ORC_MAX_NT];
    double x[ORC_MAX_NT], y[ORC_MAX_NT], yaw[ORC_MAX_NT], v[ORC_MAX_NT], a[ORC_MAX_NT], c[ORC_MAX_NT];
} orc_result;

/* per-candidate debug table (all optional, caller-allocated with >= orc_max_candidates entries) */
typedef struct orc_cand_table {
    double *cost;
    int32_t *status;
    int32_t *keep;
    int32_t *n_t;               /* untruncated sample count */
} orc_cand_table;

/* ---- spline (cubic_spline.py) ---- */
orc_spline *orc_spline_from_waypoints(int n, const double *wx, const double *wy);
orc_spline *orc_spline_from_coeffs(int n, const double *s,
                                   const double *ax, const double *bx, const double *cx, const double *dx,
                                   const double *ay, const double *by, const double *cy, const double *dy);
void orc_spline_free(orc_spline *sp);
int orc_spline_n(const orc_spline *sp);
/* out: coefficient arrays, knots[n], a[n], b[n-1], c[n], d[n-1] for x then y */
void orc_spline_coeffs(const orc_spline *sp, double *s, double *ax, double *bx, double *cx, double *dx,
                       double *ay, double *by, double *cy, double *dy);
/* out[7]: x, y, yaw, curvature, curvature_rate, (unused), (unused); NaN outside domain */
void orc_spline_eval(const orc_spline *sp, int n, const double *s, double *x, double *y,
                     double *yaw, double *kappa, double *dkappa);

/* ---- stage: ego -> Frenet state (frenet_planner.py:334-374) ---- */
/* returns 0 ok, 1 failed.  frenet[6], ref[6], new_prev_s */
int orc_cartesian_to_frenet_state(const orc_spline *sp, const orc_ego *ego,
                                  double *frenet, double *ref, double *new_prev_s);

/* ---- the whole hot path for one instance (frenet_planner.py:227-304) ---- */
int orc_max_candidates(const orc_params *p, double target_speed);
int orc_plan(const orc_params *p, const orc_spline *sp, const orc_ego *ego,
             double target_speed, const orc_overrides *ov, double max_stop_distance /* NaN = None */,
             const orc_obstacles *obs, orc_result *out, orc_cand_table *table /* may be NULL */);

/* the full 15 arrays of candidate `index` after _calc_global_paths (truncated to keep);
 * returns keep, or -1 if index out of range.  arrays[15][ORC_MAX_NT] in FrenetPath order */
int orc_candidate_path(const orc_params *p, const orc_spline *sp, const double *frenet0,
                       double target_speed, int index, double *arrays, double *cost);

/* batch driver used by bench.py's cpu_baseline: plans n_inst instances one after
 * another on the calling thread.  obstacle pointers are per instance. */
int orc_plan_batch(const orc_params *p, const orc_spline *sp, int n_inst, const orc_ego *ego,
                   const double *target_speed, const orc_overrides *ov, const double *max_stop,
                   const orc_obstacles *obs, orc_result *out);

/* collision check of an externally supplied path (frenet_planner.py:1035-1233);
 * returns 1 if collision-free */
int orc_path_collision_free(const orc_params *p, int n, const double *x, const double *y,
                            const double *yaw, const double *t, const orc_obstacles *obs);

/* ---- obstacle-tensor producer ----
 * TrajectoryPredictor.process_prediction (trajectory_predictor.py:233-313) for one sample:
 * pred [pred_len][P][2] raw predictions anchored at the last observation, anchor [P][2] or NULL,
 * out [P][n_dense][2].  Returns n_dense = len(arange(sim_dt, max(plan_horizon, pred_len*sgan_dt)+1e-9, sim_dt)). */
int orc_n_dense(double sgan_dt, double sim_dt, double plan_horizon, int pred_len);
int orc_process_prediction(double sgan_dt, double sim_dt, double plan_horizon, int pred_len, int P,
                           const double *pred, const double *anchor, double staleness, double *out);
/* predict_cv (:188-231): obs_last, obs_prev [P][2] (obs_prev NULL: zero velocity) -> out [P][n_dense][2] */
int orc_predict_cv(double sgan_dt, double sim_dt, double plan_horizon, int pred_len, int P,
                   const double *obs_last, const double *obs_prev, double staleness, double *out);
int orc_predict_cv_obs(double sgan_dt, double sim_dt, double plan_horizon, int pred_len, int P,
                       const double *obs_last, const double *obs_prev, int obs_f32, double staleness, double *out);
/* predict_single_best (:338-351): samples [S][P][T][2] -> index of the sample closest to the sample mean */
int orc_best_sample(int S, int P, int T, const double *samples, double *dist_out);

/* ---- compute_safety_metrics_static (data_structures.py:301-388) for one ego ----
 * ego = x, y, yaw, v; ped_pos / ped_vel [P][2]; footprint from p (n_circles 0: single circle of ego_radius).
 * out[5] = min_distance, collision (0/1), ttc, clearance, clearance_ahead */
void orc_safety_metrics(const orc_params *p, double ego_radius, double ped_radius, const double *ego, int P,
                        const double *ped_pos, const double *ped_vel, double *out);

#ifdef __cplusplus
}
#endif
#endif
