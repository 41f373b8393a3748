// fot_setup.hpp -- host-side planning of a batch: lattice dimensions, boundary-value inverses,
// natural cubic spline fit, per-instance descriptors and tile counts.
// Plain C++ (no HIP), shared by libfot.so and the CPU logic tests.
#pragma once

#include <cmath>
#include <string>
#include <cstdlib>
#include <vector>

#include "fot_types.h"
#include "fot_math.hpp"

namespace fot {

// ---------------------------------------------------------------------------
// per-handle constants  (reference: frenet_planner.py:149-225, 397-420, 469-475, 586-617, 1172-1175)
// ---------------------------------------------------------------------------

inline bool time_info(double T, double dt, TimeInfo &ti)
{
    ti.T = T;
    ti.n_t = (int)std::nearbyint(T / dt) + 1;
    if (ti.n_t < 1 || ti.n_t > FOT_MAX_NT) return false;
    const double T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
    // quartic: [[3T^2, 4T^3], [6T, 12T^2]]^-1
    {
        const double a = 3.0 * T2, b = 4.0 * T3, c = 6.0 * T, d = 12.0 * T2;
        const double det = a * d - b * c;
        ti.qa[0] = d / det; ti.qa[1] = -b / det; ti.qa[2] = -c / det; ti.qa[3] = a / det;
    }
    // quintic: [[T^3,T^4,T^5],[3T^2,4T^3,5T^4],[6T,12T^2,20T^3]]^-1 by cofactors
    {
        const double m[3][3] = { { T3, T4, T5 }, { 3.0 * T2, 4.0 * T3, 5.0 * T4 }, { 6.0 * T, 12.0 * T2, 20.0 * T3 } };
        double cof[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                cof[r][c] = m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1];   // cyclic => sign included
            }
        const double det = m[0][0] * cof[0][0] + m[0][1] * cof[0][1] + m[0][2] * cof[0][2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) ti.qi[r * 3 + c] = cof[c][r] / det;   // adjugate = cofactor^T
    }
    return std::isfinite(ti.qa[0]) && std::isfinite(ti.qi[0]);
}

inline int build_dev_params(const fot_params &p, DevParams &P, std::string &err)
{
    P = DevParams();
    if (!(p.dt > 0.0) || !(p.d_road_w > 0.0) || !(p.d_t_s > 0.0) || !(p.max_t > 0.0) || !(p.min_t > 0.0)) {
        err = "dt, d_road_w, d_t_s, min_t and max_t must be positive";
        return FOT_ERR_INVALID;
    }
    if (p.n_circles < 0 || p.n_circles > FOT_MAX_CIRCLES) {
        err = "n_circles must be in [0, FOT_MAX_CIRCLES]";
        return FOT_ERR_UNSUPPORTED;
    }
    P.max_speed = p.max_speed; P.max_accel = p.max_accel; P.max_curvature = p.max_curvature;
    P.max_lat_accel = p.max_lat_accel;
    P.dt = p.dt; P.d_road_w = p.d_road_w; P.max_road_width = p.max_road_width; P.min_t = p.min_t; P.d_t_s = p.d_t_s;
    P.road_lim = p.max_road_width + 1e-9;
    P.k_j = p.k_j; P.k_t = p.k_t; P.k_d = p.k_d; P.k_s_dot = p.k_s_dot; P.k_lat = p.k_lat; P.k_lon = p.k_lon;
    P.chance_epsilon = p.chance_epsilon;

    int n_ti = (int)((p.max_t - p.min_t) / p.dt + 1e-9) + 1;        // horizons min_t + i*dt, inclusive of max_t
    if (n_ti < 0) n_ti = 0;
    if (n_ti > FOT_MAX_TI) { err = "too many time horizons (FOT_MAX_TI)"; return FOT_ERR_UNSUPPORTED; }
    P.n_ti = n_ti;
    for (int i = 0; i < n_ti; ++i)
        if (!time_info(p.min_t + (double)i * p.dt, p.dt, P.ti[i])) {
            err = "a time horizon needs more than FOT_MAX_NT samples";
            return FOT_ERR_UNSUPPORTED;
        }
    P.n_total = (int)std::nearbyint(p.max_t / p.dt) + 1;
    if (P.n_total < 2 || P.n_total > FOT_MAX_NT) { err = "round(max_t/dt)+1 must be in [2, FOT_MAX_NT]"; return FOT_ERR_UNSUPPORTED; }

    P.n_side = (int)(p.max_road_width / p.d_road_w + 1e-9);
    P.n_di = P.n_side >= 0 ? 2 * P.n_side + 1 : 0;

    // brake ladder: horizons 0.5, 1.0, ... < min_t - 1e-9 that fit into n_total samples
    int n_ladder = (int)std::ceil((p.min_t - 1e-9 - 0.5) / 0.5);
    if (n_ladder < 0) n_ladder = 0;
    P.n_brake = 0;
    for (int j = 0; j < n_ladder; ++j) {
        TimeInfo tb;
        if (!time_info(0.5 + (double)j * 0.5, p.dt, tb)) continue;     // cannot happen below n_total
        if (P.n_total - tb.n_t < 0) continue;                           // n_pad < 0: skipped by the reference
        if (P.n_brake >= FOT_MAX_BRAKE) { err = "brake ladder longer than FOT_MAX_BRAKE"; return FOT_ERR_UNSUPPORTED; }
        P.brake[P.n_brake++] = tb;
    }

    P.has_footprint = p.n_circles > 0 ? 1 : 0;
    P.n_circ = p.n_circles > 0 ? p.n_circles : 1;
    for (int i = 0; i < p.n_circles; ++i) P.circ_off[i] = p.footprint_offsets[i];
    const double ego_r = p.n_circles > 0 ? p.footprint_radius : p.robot_radius;
    const double r = std::fmax(ego_r + p.obstacle_radius, 1e-6);
    const double r_dyn = r * p.collision_margin_inflation;
    P.sq_r = r * r;
    P.sq_r_dyn = r_dyn * r_dyn;
    return FOT_OK;
}

// ---------------------------------------------------------------------------
// natural cubic spline through waypoints, chord-length parameter
// (what CubicSpline2D builds, cubic_spline.py:23-45, 201-213; solved here as a tridiagonal system)
// ---------------------------------------------------------------------------

inline void fit_natural_spline(const std::vector<double> &x, const std::vector<double> &y,
                               std::vector<double> &a, std::vector<double> &b, std::vector<double> &c,
                               std::vector<double> &d)
{
    const int n = (int)x.size();
    a = y;
    b.assign(n, 0.0); c.assign(n, 0.0); d.assign(n, 0.0);
    std::vector<double> h(n - 1);
    for (int i = 0; i < n - 1; ++i) h[i] = x[i + 1] - x[i];
    // rows 1..n-2: h[i-1] c[i-1] + 2(h[i-1]+h[i]) c[i] + h[i] c[i+1] = rhs[i];  c[0] = c[n-1] = 0
    std::vector<double> diag(n, 1.0), rhs(n, 0.0), upper(n, 0.0);
    for (int i = 1; i < n - 1; ++i) {
        diag[i] = 2.0 * (h[i - 1] + h[i]);
        upper[i] = h[i];
        rhs[i] = 3.0 * (a[i + 1] - a[i]) / h[i] - 3.0 * (a[i] - a[i - 1]) / h[i - 1];
    }
    for (int i = 1; i < n - 1; ++i) {            // forward elimination (lower entry of row i is h[i-1])
        const double lower = i == 1 ? 0.0 : h[i - 1];   // c[0] = 0 removes the coupling of row 1 to row 0
        if (lower != 0.0) {
            const double w = lower / diag[i - 1];
            diag[i] -= w * upper[i - 1];
            rhs[i] -= w * rhs[i - 1];
        }
    }
    for (int i = n - 2; i >= 1; --i) {
        const double next = i == n - 2 ? 0.0 : c[i + 1];
        c[i] = (rhs[i] - upper[i] * next) / diag[i];
    }
    for (int i = 0; i < n - 1; ++i) {
        d[i] = (c[i + 1] - c[i]) / (3.0 * h[i]);
        b[i] = (a[i + 1] - a[i]) / h[i] - h[i] * (2.0 * c[i] + c[i + 1]) / 3.0;
    }
}

struct HostSpline {
    std::vector<double> s, ax, bx, cx, dx, ay, by, cy, dy;
    int n = 0;
};

inline int build_spline(int n, const double *wx, const double *wy, HostSpline &sp, std::string &err)
{
    if (n < 2 || !wx || !wy) { err = "a path needs at least 2 waypoints"; return FOT_ERR_INVALID; }
    sp.n = n;
    sp.s.assign(n, 0.0);
    double acc = 0.0;
    for (int i = 0; i < n - 1; ++i) {
        acc += std::hypot(wx[i + 1] - wx[i], wy[i + 1] - wy[i]);
        sp.s[i + 1] = acc;
    }
    for (int i = 0; i < n - 1; ++i)
        if (!(sp.s[i + 1] - sp.s[i] > 0.0)) { err = "consecutive waypoints must be distinct"; return FOT_ERR_INVALID; }
    std::vector<double> vx(wx, wx + n), vy(wy, wy + n);
    fit_natural_spline(sp.s, vx, sp.ax, sp.bx, sp.cx, sp.dx);
    fit_natural_spline(sp.s, vy, sp.ay, sp.by, sp.cy, sp.dy);
    return FOT_OK;
}

// ---------------------------------------------------------------------------
// batch layout
// ---------------------------------------------------------------------------

// Tile table of a handle: for every terminal-speed grid size n_tv (the one thing besides the planner constants that
// shapes an instance's lattice) the tiles it is cut into.  Built once per handle, resident in HBM.
struct TileShapes {
    int row_budget = 0;                            // rows a wave that stages one tile on its own needs at most
    int grouped = 0;                               // 1: groups of GROUP_TILES tiles share a row table (fot_math.hpp)
    int64_t n_real = 0;                            // tiles with candidates, over all shapes (cost of the cut)
    std::vector<int32_t> cand0, n;                 // all shapes back to back
    std::vector<int32_t> span;                     // first << 16 | last longitudinal profile of each tile (0 for padding)
    int32_t off[FOT_MAX_TV + 2] = { 0 };           // shape of n_tv: entries [off[n_tv], off[n_tv + 1])
    int tiles_of(int n_tv) const { return off[n_tv + 1] - off[n_tv]; }
};

inline InstDesc shape_desc(const DevParams &P, int n_tv)
{
    InstDesc D = InstDesc();
    D.n_tv = n_tv;
    D.n_grid = P.n_ti * n_tv * P.n_di;
    D.n_cand_max = D.n_grid + P.n_brake;
    return D;
}

// per-wave rows: every tile fits row_budget rows
inline void build_tile_shapes_wave(const DevParams &P, TileShapes &T)
{
    T = TileShapes();
    T.row_budget = tile_row_budget(P.n_total);
    for (int n_tv = 0; n_tv <= FOT_MAX_TV; ++n_tv) {
        T.off[n_tv] = (int32_t)T.cand0.size();
        if (n_tv == 0) continue;
        const InstDesc D = shape_desc(P, n_tv);
        for (int c = 0; c < D.n_cand_max;) {
            const int n = tile_extent(P, D, c, T.row_budget);
            T.cand0.push_back(c); T.n.push_back(n);
            c += n; ++T.n_real;
        }
    }
    T.off[FOT_MAX_TV + 1] = (int32_t)T.cand0.size();
}

// groups: greedy -- a group takes profiles while their rows fit GROUP_ROWS (and GROUP_MAX_PROFILES), its tiles take 64
// candidates each out of those profiles (at most TILE_MAX_PROFILES per tile: k_cull merges that many boxes per tile)
inline void build_tile_shapes_grouped(const DevParams &P, TileShapes &T)
{
    T = TileShapes();
    T.grouped = 1;
    for (int n_tv = 0; n_tv <= FOT_MAX_TV; ++n_tv) {
        T.off[n_tv] = (int32_t)T.cand0.size();
        if (n_tv == 0) continue;
        const InstDesc D = shape_desc(P, n_tv);
        const int n_grid_lon = P.n_ti * D.n_tv;
        int c = 0;
        while (c < D.n_cand_max) {
            int rows = 0, profs = 0, last_slot = -1;          // of the group so far
            for (int t = 0; t < GROUP_TILES; ++t) {
                const int c0 = c;
                int n = 0, tile_profs = 0, tile_rows = 0;
                while (n < WAVE && c < D.n_cand_max && tile_profs < TILE_MAX_PROFILES) {
                    int slot, left;                           // profile of candidate c, candidates of it from c on
                    if (c < D.n_grid) { slot = c / P.n_di; left = (slot + 1) * P.n_di - c; }
                    else { slot = n_grid_lon + (c - D.n_grid); left = 1; }
                    const int r = profile_rows(P, D, slot);
                    if (slot != last_slot) {                  // a profile the group does not hold yet
                        if (profs > 0 && (rows + r > GROUP_ROWS || profs + 1 > GROUP_MAX_PROFILES)) break;
                        rows += r; ++profs; last_slot = slot;
                    }
                    ++tile_profs; tile_rows += r;
                    const int take = left < WAVE - n ? left : WAVE - n;
                    n += take; c += take;
                }
                T.cand0.push_back(c0); T.n.push_back(n);
                if (n > 0) ++T.n_real;
                if (tile_rows > T.row_budget) T.row_budget = tile_rows;
            }
        }
    }
    T.off[FOT_MAX_TV + 1] = (int32_t)T.cand0.size();
}

// profiles each tile's candidates come from (k_cull merges their boxes per tile and step)
inline void fill_tile_spans(const DevParams &P, TileShapes &T)
{
    T.span.assign(T.cand0.size(), 0);
    for (int n_tv = 1; n_tv <= FOT_MAX_TV; ++n_tv) {
        const InstDesc D = shape_desc(P, n_tv);
        for (int t = T.off[n_tv]; t < T.off[n_tv + 1]; ++t) {
            if (T.n[(size_t)t] <= 0) continue;
            int s0, s1;
            wave_profile_span(P, D, P.n_ti * n_tv, T.cand0[(size_t)t], T.cand0[(size_t)t] + T.n[(size_t)t] - 1, s0, s1);
            T.span[(size_t)t] = (int32_t)(((uint32_t)s0 << 16) | (uint32_t)s1);
        }
    }
}

// The cut of the handle's lattice: groups (four waves per SIMD in k_evaluate) unless they make over 15 % more tiles
// than the per-wave cut -- with few lateral offsets per profile a tile fills its 64 lanes badly either way, and a
// group's sixteen profiles may then hold fewer candidates than four per-wave tiles.  `cut` forces one
// (fot_debug_set_tile_cut: the GPU tests run every golden under both).
enum { TILE_CUT_AUTO = 0, TILE_CUT_WAVE = 1, TILE_CUT_GROUP = 2 };
inline void build_tile_shapes(const DevParams &P, TileShapes &T, int cut = TILE_CUT_AUTO)
{
    TileShapes wave, grouped;
    build_tile_shapes_wave(P, wave);
    build_tile_shapes_grouped(P, grouped);
    bool use_groups = (double)grouped.n_real <= 1.15 * (double)wave.n_real;
    if (cut == TILE_CUT_WAVE) use_groups = false;
    if (cut == TILE_CUT_GROUP) use_groups = true;
    T = use_groups ? grouped : wave;
    fill_tile_spans(P, T);
}

struct BatchLayout {
    std::vector<InstDesc> desc;
    int n_inst = 0;
    int n_tiles = 0;              // tiles of the whole batch (k_evaluate's units of work: one wave each)
    int max_tiles = 0;            // most tiles of one instance
    int row_budget = 0;           // LDS rows per k_evaluate wave the tiles were cut for
    int grouped = 0;              // the tile table's cut (TileShapes::grouped)
    int64_t n_slots = 0;          // candidate slots (instances padded to multiples of 64)
    int64_t n_lon = 0;            // longitudinal profile slots
    int max_lon = 0;              // max profiles of one instance
    int64_t n_static = 0;         // extent of the caller's static_xy that is referenced (points)
    int64_t dyn_src_points = 0;   // extent of the caller's dyn_xy that is referenced (points)
    int64_t n_entries = 0;        // broad-phase entry slots in the batch (n_total * ent_cap per instance)
    int64_t n_tracks = 0;         // pedestrian tracks (S * P per instance, rounded up to 16): one NaN flag each
    int64_t max_dyn_bytes = 0;    // largest dynamic tensor of one instance (sizes the NaN scan)
    bool any_obstacles = false;
    bool any_tmajor = false;      // some instance's tensor is time-major (its NaN flags come from k_frenet_state's scan blocks)
};

inline int build_batch_layout(const fot_params &hp, const DevParams &P, const TileShapes &shapes, const fot_batch &b,
                              BatchLayout &L, std::string &err)
{
    L = BatchLayout();
    if (b.n_inst < 0) { err = "n_inst < 0"; return FOT_ERR_INVALID; }
    if (b.n_inst > 0 && (!b.ego || !b.target_speed)) { err = "ego / target_speed missing"; return FOT_ERR_INVALID; }
    if (b.obstacle_dtype != FOT_F32 && b.obstacle_dtype != FOT_F64) { err = "obstacle_dtype"; return FOT_ERR_INVALID; }
    L.n_inst = b.n_inst;
    L.desc.resize(b.n_inst);
    L.row_budget = shapes.row_budget;
    L.grouped = shapes.grouped;
    for (int i = 0; i < b.n_inst; ++i) {
        InstDesc &D = L.desc[i];
        D = InstDesc();
        D.ego = b.ego[i];
        if (D.ego.has_prev_s < 0 || D.ego.has_prev_s > FOT_EGO_IS_FRENET) { err = "has_prev_s must be 0 .. 3"; return FOT_ERR_INVALID; }
        if (i == 0 && D.ego.has_prev_s == FOT_PREV_S_CHAINED) { err = "the first instance cannot be chained"; return FOT_ERR_INVALID; }
        const double target = b.target_speed[i];
        D.target_speed = target;
        D.max_stop = b.max_stop_distance ? b.max_stop_distance[i] : NAN;
        D.lim_speed = hp.max_speed; D.lim_accel = hp.max_accel; D.lim_curv = hp.max_curvature; D.lim_lat = hp.max_lat_accel;
        if (b.overrides) {
            const fot_overrides &o = b.overrides[i];
            if (!std::isnan(o.max_speed)) D.lim_speed = o.max_speed;
            if (!std::isnan(o.max_accel)) D.lim_accel = o.max_accel;
            if (!std::isnan(o.max_curvature)) D.lim_curv = o.max_curvature;
            if (!std::isnan(o.max_lat_accel)) D.lim_lat = o.max_lat_accel;
        }
        D.step_limit = std::fmax(D.lim_speed, hp.max_speed) * hp.dt * 3.0;

        // terminal speed grid: target - k*d_t_s, k = 0..n_down, plus 0.0 unless already reached
        if (!std::isfinite(target)) { err = "target_speed must be finite"; return FOT_ERR_INVALID; }
        const int n_down = (int)(target / hp.d_t_s + 1e-9);
        if (n_down + 1 <= 0) { err = "target_speed too negative: empty terminal-speed grid"; return FOT_ERR_INVALID; }
        const double tv_last = target - (double)n_down * hp.d_t_s;
        D.n_down = n_down;
        D.n_tv = n_down + 1 + (tv_last > 1e-9 ? 1 : 0);
        if (D.n_tv > FOT_MAX_TV) { err = "terminal-speed grid larger than FOT_MAX_TV"; return FOT_ERR_UNSUPPORTED; }
        D.n_grid = P.n_ti * D.n_tv * P.n_di;
        D.n_cand_max = D.n_grid + P.n_brake;

        // tiles of this instance: its lattice shape's run of the handle's tile table
        const int n_tiles_i = shapes.tiles_of(D.n_tv);
        D.shape_off = shapes.off[D.n_tv];
        const int64_t slots_i = ((int64_t)D.n_cand_max + WAVE - 1) / WAVE * WAVE;
        if (L.n_slots + slots_i > 0x7fffffffLL || (int64_t)L.n_tiles + n_tiles_i > 0x7fffffffLL) {
            err = "batch too large"; return FOT_ERR_UNSUPPORTED;
        }
        D.cand_off = (int32_t)L.n_slots;
        D.tile0 = L.n_tiles;
        D.n_tiles = n_tiles_i;
        L.n_tiles += n_tiles_i;
        if (n_tiles_i > L.max_tiles) L.max_tiles = n_tiles_i;
        L.n_slots += slots_i;
        const int n_lon_i = P.n_ti * D.n_tv + P.n_brake;
        D.lon_off = (int32_t)L.n_lon;
        L.n_lon += n_lon_i;
        if (n_lon_i > L.max_lon) L.max_lon = n_lon_i;

        if (b.static_off) {
            const int64_t lo = b.static_off[i], hi = b.static_off[i + 1];
            if (lo < 0 || hi < lo) { err = "static_off must be non-decreasing"; return FOT_ERR_INVALID; }
            D.static_off = lo;
            D.n_static = (int32_t)(hi - lo);
            if (hi > L.n_static) L.n_static = hi;
        }
        D.dyn_mode = FOT_DYN_NONE;
        if (b.dyn_off && b.dyn_dims) {
            const int32_t *dm = b.dyn_dims + 4 * i;
            const int mode = dm[0] & ~FOT_DYN_LAYOUT_TSP;
            D.dyn_tmajor = (dm[0] & FOT_DYN_LAYOUT_TSP) ? 1 : 0;
            int S = dm[1], Pn = dm[2], T = dm[3];
            if (mode != FOT_DYN_NONE && mode != FOT_DYN_SINGLE && mode != FOT_DYN_DISTRIBUTION) {
                err = "dyn_dims mode"; return FOT_ERR_INVALID;
            }
            if (mode == FOT_DYN_SINGLE) S = 1;
            if (mode != FOT_DYN_NONE && S > 0 && Pn > 0 && T > 0) {
                if (S > FOT_MAX_SAMPLES) { err = "more prediction samples than FOT_MAX_SAMPLES"; return FOT_ERR_UNSUPPORTED; }
                if (b.dyn_off[i] < 0) { err = "dyn_off < 0"; return FOT_ERR_INVALID; }
                D.dyn_mode = mode; D.S = S; D.P = Pn; D.T = T;
                if (D.dyn_tmajor) L.any_tmajor = true;
                D.dyn_off = b.dyn_off[i];
                const int64_t pts = (int64_t)S * Pn * T;
                if (D.dyn_off + pts > L.dyn_src_points) L.dyn_src_points = D.dyn_off + pts;
                D.max_viol = mode == FOT_DYN_DISTRIBUTION ? (int)std::floor(hp.chance_epsilon * (double)S) : 0;
                D.nan_off = L.n_tracks;
                L.n_tracks += ((int64_t)S * Pn + 15) & ~(int64_t)15;
                const int64_t bytes = pts * 2 * (b.obstacle_dtype == FOT_F32 ? 4 : 8);
                if (bytes > L.max_dyn_bytes) L.max_dyn_bytes = bytes;
            }
        }
        const int64_t per_k = (int64_t)D.n_static + (D.dyn_mode != FOT_DYN_NONE ? (int64_t)D.S * D.P : 0);
        if (per_k > 0) {
            // chunk indices travel as 16 bits (strip_range): at most 65534 chunks of 8 entries per time step
            if (per_k > 65534 * 8) { err = "too many obstacle points in one instance"; return FOT_ERR_UNSUPPORTED; }
            D.ent_cap = (int32_t)((per_k + 15) & ~(int64_t)15);          // whole chunk pairs
            D.ent_off = L.n_entries;
            L.n_entries += (int64_t)D.ent_cap * P.n_total;
            L.any_obstacles = true;
        }
    }
    for (int i = b.n_inst - 1, run = 0; i >= 0; --i) {              // chain lengths, seen from each head
        L.desc[i].n_chained = run;
        run = L.desc[i].ego.has_prev_s == FOT_PREV_S_CHAINED ? run + 1 : 0;
    }
    if ((L.n_static > 0 && !b.static_xy) || (L.dyn_src_points > 0 && !b.dyn_xy)) {
        err = "obstacle offsets given without coordinates";
        return FOT_ERR_INVALID;
    }
    return FOT_OK;
}

}  // namespace fot
