// fot_types.h -- plain-old-data shared by the host code and the gfx950 kernels.
#pragma once

#include <stdint.h>
#include "../../include/fot.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define FOT_HD __host__ __device__ __forceinline__
#else
#define FOT_HD inline
#endif

namespace fot {

constexpr int WAVE = 64;                 // gfx950 wavefront
constexpr int WAVES_PER_GROUP = 4;       // waves of one k_evaluate workgroup
// k_evaluate works on TILES, one per wave: up to 64 consecutive candidates of one instance whose longitudinal
// profiles fit one wave's share of LDS (tile_extent, fot_math.hpp).
constexpr int TILE_MAX_PROFILES = 8;     // profiles a tile may span (more brake-ladder entries: next tile)
constexpr int N_XCD = 8;
constexpr int LON_FIELDS = 10;           // rows of a GlobalTab table: s, s_d, s_dd, rx, ry, cos_r, sin_r, kappa_r, dkappa_r, 1/s_d
constexpr int ST_PENDING = FOT_ST_OK;    // passed the kinematic checks, collision check outstanding

struct d2 { double x, y; };
struct f2 { float x, y; };
constexpr float FAR32 = 1.0e18f;         // padding obstacle of the float32 broad-phase rows

// one time horizon: quartic / quintic boundary-value inverses (frenet_planner.py:586-617)
struct TimeInfo {
    double T;
    double qa[4];                        // inverse of [[3T^2,4T^3],[6T,12T^2]], row-major
    double qi[9];                        // inverse of the 3x3 quintic matrix, row-major
    int32_t n_t;                         // round(T/dt)+1 samples
    int32_t _pad;
};

// per-handle constants, resident in HBM, read through scalar loads
struct DevParams {
    double max_speed, max_accel, max_curvature, max_lat_accel;
    double dt, d_road_w, max_road_width, min_t, d_t_s;
    double k_j, k_t, k_d, k_s_dot, k_lat, k_lon;
    double sq_r, sq_r_dyn;               // squared combined radii (frenet_planner.py:1172-1175)
    double chance_epsilon;
    double road_lim;                     // max_road_width + 1e-9 (frenet_planner.py:982)
    double circ_off[FOT_MAX_CIRCLES];
    int32_t n_ti, n_di, n_side, n_brake; // n_brake = valid ladder entries
    int32_t n_total;                     // round(max_t/dt)+1
    int32_t n_circ;                      // expanded points per sample (>= 1)
    int32_t has_footprint;
    int32_t _pad;
    TimeInfo ti[FOT_MAX_TI];
    TimeInfo brake[FOT_MAX_BRAKE];
};

// cubic spline in HBM: 9 arrays of n doubles (b, d padded to n)
struct SplineView {
    const double *s, *ax, *bx, *cx, *dx, *ay, *by, *cy, *dy;
    int32_t n;
    int32_t _pad;
};

// host-built description of one instance
struct InstDesc {
    fot_ego ego;
    double target_speed;
    double max_stop;                     // NaN = none
    double lim_speed, lim_accel, lim_curv, lim_lat;
    double step_limit;                   // max(c_max_speed, max_speed)*dt*3 (frenet_planner.py:955)
    int32_t n_tv;
    int32_t n_down;                      // tv[k] = target - k*d_t_s for k <= n_down, else 0.0
    int32_t n_grid;                      // n_ti*n_tv*n_di
    int32_t cand_off;                    // first candidate slot (multiple of 64)
    int32_t n_cand_max;                  // n_grid + n_brake
    int32_t lon_off;                     // first longitudinal-profile slot
    int32_t n_static;
    int64_t static_off;                  // points into the caller's static_xy
    int32_t dyn_mode, S, P, T;
    int64_t dyn_off;                     // points into the caller's dyn_xy
    int64_t ent_off;                     // first broad-phase entry slot of this instance
    int32_t ent_cap;                     // entry slots per time step (multiple of 16): S*P + n_static rounded up
    int32_t tile0;                       // number of this instance's first tile in the batch (tile-range rows)
    int32_t n_tiles;                     // tiles of this instance
    int32_t shape_off;                   // its lattice shape's first entry in the handle's tile table
    int32_t dyn_tmajor;                  // 1: the dynamic tensor is [T][S][P][2] (FOT_DYN_LAYOUT_TSP), 0: [S][P][T][2]
    int32_t max_viol;                    // floor(eps*S)
    int32_t n_chained;                   // instances right behind this one that continue its nearest-point cache
    int32_t _pad;
    int64_t nan_off;                     // first of this instance's S * P track flags (fot_kernels.hip scan_nan_tracks)
};

// device-produced per-instance state
struct InstState {
    double frenet0[6];
    double ref0[6];
    double new_prev_s;
    int32_t c2f_ok;
    int32_t n_brake;                     // 0 when s_d <= BRAKE_MIN_SPEED
    int32_t n_cand;
    int32_t _pad;
};

// one longitudinal profile (Ti x tv grid entry or brake-ladder entry)
struct LonInfo {
    double a0, a1, a2, a3, a4;
    double Js;                           // sum s_ddd^2 over all samples
    double sd_last;                      // s_d at the last (untruncated) sample
    double T;
    int32_t n_t;                         // samples incl. brake padding
    int32_t n_eval;                      // samples on the polynomial (== n_t unless brake)
};

}  // namespace fot
