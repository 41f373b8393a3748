// fot_math.hpp -- the planner arithmetic, float64, usable from gfx950 kernels and
// (for the CPU-side logic tests in tests/emu) from plain C++.
//
// What each block computes is defined by the reference
// (mnhrk15/integrated_path_planning); the line references say where.  The
// formulation is our own: trigonometry of the reference frame is carried as
// (cos, sin) pairs, so the per-sample Frenet->Cartesian transform needs no
// transcendental, and only the selected path gets an atan2 for its yaw.
#pragma once

#include <math.h>
#include "fot_types.h"

namespace fot {

// ---------------------------------------------------------------------------
// reciprocal and reciprocal square root: hardware seed + two Newton steps on the device (no scaling /
// fix-up sequence: denormal and overflow inputs do not occur here; 0, inf and NaN still propagate to
// non-finite results, which is all the callers rely on), plain IEEE expressions on the host
// ---------------------------------------------------------------------------

FOT_HD double fast_rcp(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rcp_f64 is good to ~2^-26; one Newton step squares that (a few units in the last place -- every consumer is a
    // product chain that rounds a few times anyway), a second one only buys the last bit
    double x = __builtin_amdgcn_rcp(a);
    x = fma(x, fma(-a, x, 1.0), x);
#ifdef FOT_NEWTON2
    x = fma(x, fma(-a, x, 1.0), x);
#endif
    return x;
#else
    return 1.0 / a;
#endif
}

FOT_HD double fast_rsqrt(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = fma(y, fma(-h * y, y, 0.5), y);
#ifdef FOT_NEWTON2
    y = fma(y, fma(-h * y, y, 0.5), y);
#endif
    return y;
#else
    return 1.0 / sqrt(a);
#endif
}

FOT_HD double sum_sq_unfused(double a, double b);

// a * b + c as ONE rounding on the device (v_fma_f64), spelled out where two kernels must produce the same bits from
// the same expression whatever the compiler's contraction heuristics make of the code around it; two roundings on the
// host (the CPU logic test compares with tolerances, and builds with -ffp-contract=off)
#if defined(__HIP_DEVICE_COMPILE__)
#define FOT_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#define FOT_FMA(a, b, c) ((a) * (b) + (c))
#endif

// the lateral quintic's value: q0 + t (q1 + t (q2 + t (q3 + t (q4 + t q5)))) (:688)
FOT_HD double quintic_value(const double *q, double t)
{
    return FOT_FMA(t, FOT_FMA(t, FOT_FMA(t, FOT_FMA(t, FOT_FMA(t, q[5], q[4]), q[3]), q[2]), q[1]), q[0]);
}

// ---------------------------------------------------------------------------
// Correctly rounded hypot and cube, for the nearest-point search.
//
// The search (coordinate_converter.py:202-308) DECIDES by comparing math.hypot(x - px, y - py) of neighbouring probes,
// with px = a + b h + c h**2.0 + d h**3.0 (cubic_spline.py:70-71): probes a few micrometres apart tie at rounding
// level, and whichever implementation's last bit differs picks another probe -- the arc length then ends a refinement
// step away (round 3's "nearest-point tie": 3e-6 m, once 3e-5 m when only the contraction flag of this file changed).
// Python's hypot, NumPy's power (glibc pow) and glibc's hypot all return the correctly rounded result in practice; so do
// these two (exact squares / products through FMA residuals, one correction step), on the device as on the host, and the
// probe position below follows the reference's own operation order with every operation rounded once.  The library's
// arc length is then the reference's bit for bit, whatever flags this file is compiled with.
// ---------------------------------------------------------------------------

FOT_HD double hypot_cr(double x, double y)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    x = fabs(x); y = fabs(y);
    if (isinf(x) || isinf(y)) return INFINITY;
    if (isnan(x) || isnan(y)) return NAN;
    if (x < y) { const double t = x; x = y; y = t; }
    if (y == 0.0) return x;
    double back = 1.0;                                            /* far from 1: scaled by a power of two (exact) */
    if (x > 0x1p500) { x *= 0x1p-600; y *= 0x1p-600; back = 0x1p600; }
    else if (x < 0x1p-500) { x *= 0x1p600; y *= 0x1p600; back = 0x1p-600; }
    const double xx = x * x, ex = __builtin_fma(x, x, -xx);       // x^2 = xx + ex exactly
    const double yy = y * y, ey = __builtin_fma(y, y, -yy);
    const double s = xx + yy, bb = s - xx;
    const double es = (xx - (s - bb)) + (yy - bb);                // xx + yy = s + es exactly
    const double lo = es + (ex + ey);
    const double h = sqrt(s);                                     // correctly rounded square root of the leading part
    const double r = __builtin_fma(-h, h, s) + lo;                // (x^2 + y^2) - h^2, to working precision
    return (h + r / (2.0 * h)) * back;
}

FOT_HD double cube_cr(double h)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double p = h * h, e = __builtin_fma(h, h, -p);          // h^2 = p + e exactly
    const double t = p * h, e2 = __builtin_fma(p, h, -t);         // p h = t + e2 exactly
    return t + __builtin_fma(e, h, e2);
}

// ---------------------------------------------------------------------------
// cubic spline (reference: src/planning/cubic_spline.py:47-166, 215-288)
// ---------------------------------------------------------------------------

struct SplinePt {
    double x, y, dx, dy, ddx, ddy, dddx, dddy;
};

// segment of v: number of knots <= v, minus one, clipped to [0, n-2]  (:154-166)
FOT_HD int spline_index(const SplineView &sp, double v)
{
    int lo = 0, hi = sp.n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sp.s[mid] <= v) lo = mid + 1; else hi = mid;
    }
    int idx = lo - 1;
    idx = idx < 0 ? 0 : idx;
    idx = idx > sp.n - 2 ? sp.n - 2 : idx;
    return idx;
}

// position and derivatives; everything NaN outside [s_0, s_end]  (:62, :91)
FOT_HD bool spline_point(const SplineView &sp, double s, SplinePt &o)
{
    if (!(s >= sp.s[0] && s <= sp.s[sp.n - 1])) {
        const double q = NAN;
        o.x = o.y = o.dx = o.dy = o.ddx = o.ddy = o.dddx = o.dddy = q;
        return false;
    }
    const int i = spline_index(sp, s);
    const double h = s - sp.s[i], h2 = h * h;
    const double bx = sp.bx[i], cx = sp.cx[i], dx = sp.dx[i];
    const double by = sp.by[i], cy = sp.cy[i], dy = sp.dy[i];
    o.x = sp.ax[i] + bx * h + cx * h2 + dx * (h2 * h);
    o.y = sp.ay[i] + by * h + cy * h2 + dy * (h2 * h);
    o.dx = bx + 2.0 * cx * h + 3.0 * dx * h2;
    o.dy = by + 2.0 * cy * h + 3.0 * dy * h2;
    o.ddx = 2.0 * cx + 6.0 * dx * h;
    o.ddy = 2.0 * cy + 6.0 * dy * h;
    o.dddx = 6.0 * dx;
    o.dddy = 6.0 * dy;
    return true;
}

// position alone, for the nearest-point search: the reference's expression a + b h + c h**2.0 + d h**3.0 evaluated left
// to right, every operation rounded once (no contraction), the cube correctly rounded as NumPy's power returns it
FOT_HD void spline_xy(const SplineView &sp, double s, double &x, double &y)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (!(s >= sp.s[0] && s <= sp.s[sp.n - 1])) { x = NAN; y = NAN; return; }
    const int i = spline_index(sp, s);
    const double h = s - sp.s[i], h2 = h * h, h3 = cube_cr(h);
    const double bxh = sp.bx[i] * h, cxh = sp.cx[i] * h2, dxh = sp.dx[i] * h3;
    const double byh = sp.by[i] * h, cyh = sp.cy[i] * h2, dyh = sp.dy[i] * h3;
    x = ((sp.ax[i] + bxh) + cxh) + dxh;
    y = ((sp.ay[i] + byh) + cyh) + dyh;
}

// tangent direction as (cos, sin), curvature (:246) and curvature rate (:265-273)
FOT_HD void spline_frame(const SplinePt &p, double &cos_r, double &sin_r, double &kappa, double &dkappa)
{
    const double d = p.dx * p.dx + p.dy * p.dy;
    const double rt = sqrt(d);
    cos_r = p.dx / rt;
    sin_r = p.dy / rt;
    const double d15 = d * rt;
    const double a = p.dx * p.ddy - p.dy * p.ddx;
    const double b = p.dx * p.dddy - p.dy * p.dddx;
    const double c = p.dx * p.ddx + p.dy * p.ddy;
    kappa = a / d15;
    dkappa = b / d15 - 3.0 * a * c / (d15 * d);
}

// ---------------------------------------------------------------------------
// nearest point on the path (reference: src/core/coordinate_converter.py:202-339)
// ---------------------------------------------------------------------------

// numpy.linspace(a, b, num)[i]
FOT_HD double linspace_at(double a, double b, int num, int i)
{
    // start + i * step with the product and the sum rounded separately, as NumPy forms them: a standing ego sits exactly
    // on its previous arc length, the 100-sample window is then symmetric about it and samples 49 and 50 TIE up to these
    // very roundings (a fused multiply-add here moved the result of a reference closed-loop call by 3e-5 m)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (num == 1) return a;
    if (i == num - 1) return b;
    const double step = (b - a) / (double)(num - 1);
    if (step == 0.0) return a + ((double)i / (double)(num - 1)) * (b - a);
    return a + (double)i * step;
}

struct ScanBest {
    double dist;
    int idx;
};

FOT_HD void scan_merge(ScanBest &a, const ScanBest &b)
{
    if (b.idx >= 0 && (a.idx < 0 || b.dist < a.dist || (b.dist == a.dist && b.idx < a.idx))) a = b;
}

// distance scan over linspace(s_lo, s_hi, num) restricted to samples lane, lane+nlanes, ...
// keeps the first strict minimum (:230-237); nan_first reproduces np.argmin (:336)
FOT_HD ScanBest scan_samples(const SplineView &sp, double x, double y, double s_lo, double s_hi, int num,
                             int lane, int nlanes, bool nan_first)
{
    ScanBest best = { INFINITY, -1 };
    for (int i = lane; i < num; i += nlanes) {
        const double s = linspace_at(s_lo, s_hi, num, i);
        double px, py;
        spline_xy(sp, s, px, py);
        double dist = hypot_cr(x - px, y - py);
        if (nan_first && isnan(dist)) dist = -INFINITY;
        if (dist < best.dist) { best.dist = dist; best.idx = i; }
    }
    return best;
}

FOT_HD int global_search_count(const SplineView &sp)
{
    const int num = (int)(sp.s[sp.n - 1] / 0.1);
    return num < 100 ? 100 : num;
}

// 20 rounds of {left, centre, right}, step 0.2 halving when the centre wins (:253-280)
FOT_HD double refine_nearest(const SplineView &sp, double x, double y, double best_s)
{
    const double s_end = sp.s[sp.n - 1];
    double ds = 0.2;
    for (int it = 0; it < 20; ++it) {
        const double s_left = fmax(0.0, best_s - ds);
        const double s_right = fmin(s_end, best_s + ds);
        double px, py;
        spline_xy(sp, s_left, px, py);
        const double dist_left = hypot_cr(x - px, y - py);
        spline_xy(sp, s_right, px, py);
        const double dist_right = hypot_cr(x - px, y - py);
        spline_xy(sp, best_s, px, py);
        const double dist_curr = hypot_cr(x - px, y - py);
        if (dist_left < dist_curr && dist_left < dist_right) best_s = s_left;
        else if (dist_right < dist_curr && dist_right < dist_left) best_s = s_right;
        else ds *= 0.5;
    }
    return best_s;
}

// reference point at rs + Cartesian->Frenet of the ego
// (coordinate_converter.py:285-308, :26-88; frenet_planner.py:362-371).  false = the reference raises.
FOT_HD bool frenet_state_at(const SplineView &sp, const fot_ego &ego, double rs, double *fr, double *ref)
{
    SplinePt p;
    spline_point(sp, rs, p);
    if (isnan(p.x) || isnan(p.y)) return false;
    double cos_r, sin_r, rkappa, rdkappa;
    spline_frame(p, cos_r, sin_r, rkappa, rdkappa);
    const double rtheta = atan2(p.dy, p.dx);
    if (isnan(rtheta) || isnan(rkappa) || isnan(rdkappa)) return false;
    ref[0] = rs; ref[1] = p.x; ref[2] = p.y; ref[3] = rtheta; ref[4] = rkappa; ref[5] = rdkappa;

    const double dx = ego.x - p.x, dy = ego.y - p.y;
    const double cr = cos(rtheta), sr = sin(rtheta);
    const double cross = cr * dy - sr * dx;
    const double d = copysign(hypot_cr(dx, dy), cross);           // (np.hypot, coordinate_converter.py:65)
    const double delta = ego.yaw - rtheta;
    const double tan_d = tan(delta), cos_d = cos(delta);
    const double omkd = 1.0 - rkappa * d;
    const double d_p = omkd * tan_d;
    const double krdp = rdkappa * d + rkappa * d_p;
    const double kappa = ego.last_kappa;
    const double d_pp = -krdp * tan_d + omkd / (cos_d * cos_d) * (kappa * omkd / cos_d - rkappa);
    const double s_d = ego.v * cos_d / omkd;
    const double dtp = omkd / cos_d * kappa - rkappa;
    const double s_dd = (ego.a * cos_d - s_d * s_d * (d_p * dtp - krdp)) / omkd;
    fr[0] = rs; fr[1] = s_d; fr[2] = s_dd;
    fr[3] = d;
    fr[4] = d_p * s_d;
    fr[5] = d_pp * (s_d * s_d) + d_p * s_dd;
    return true;
}

// FOT_EGO_IS_FRENET: the record holds s, s_d, s_dd, d, d_d, d_dd; the reference point is the spline frame at s
FOT_HD bool frenet_state_given(const SplineView &sp, const fot_ego &ego, double *fr, double *ref)
{
    fr[0] = ego.x; fr[1] = ego.y; fr[2] = ego.yaw; fr[3] = ego.v; fr[4] = ego.a; fr[5] = ego.last_kappa;
    SplinePt p;
    spline_point(sp, fr[0], p);
    if (isnan(p.x) || isnan(p.y)) return false;
    double cos_r, sin_r, rkappa, rdkappa;
    spline_frame(p, cos_r, sin_r, rkappa, rdkappa);
    ref[0] = fr[0]; ref[1] = p.x; ref[2] = p.y; ref[3] = atan2(p.dy, p.dx); ref[4] = rkappa; ref[5] = rdkappa;
    return !(isnan(ref[3]) || isnan(rkappa) || isnan(rdkappa));
}

// ---------------------------------------------------------------------------
// lattice polynomials (reference: frenet_planner.py:586-701)
// ---------------------------------------------------------------------------

// terminal-speed grid (:410-413): target, target - d_t_s, ..., and a final 0.0
FOT_HD double tv_value(const DevParams &P, const InstDesc &D, int itv)
{
    return itv <= D.n_down ? D.target_speed - (double)itv * P.d_t_s : 0.0;
}

// (The coefficients of both polynomials, like everything that enters a candidate's cost, are formed with every
// product and sum rounded on its own: whether the compiler contracts a particular a * b + c into a fused multiply-add
// depends on the code it is inlined into, and the records must not depend on which kernel evaluated a candidate.)
FOT_HD void lon_coeffs(const double *fr, double tv, const TimeInfo &ti, LonInfo &L)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    L.a0 = fr[0]; L.a1 = fr[1]; L.a2 = fr[2] / 2.0;
    const double b0 = tv - L.a1 - 2.0 * L.a2 * ti.T;
    const double b1 = -2.0 * L.a2;
    L.a3 = b0 * ti.qa[0] + b1 * ti.qa[1];
    L.a4 = b0 * ti.qa[2] + b1 * ti.qa[3];
    L.T = ti.T;
}

FOT_HD void lat_coeffs(const double *fr, double di, const TimeInfo &ti, double *q)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double T = ti.T;
    q[0] = fr[3]; q[1] = fr[4]; q[2] = fr[5] / 2.0;
    const double b0 = di - q[0] - q[1] * T - q[2] * T * T;
    const double b1 = -q[1] - 2.0 * q[2] * T;
    const double b2 = -2.0 * q[2];
    q[3] = b0 * ti.qi[0] + b1 * ti.qi[1] + b2 * ti.qi[2];
    q[4] = b0 * ti.qi[3] + b1 * ti.qi[4] + b2 * ti.qi[5];
    q[5] = b0 * ti.qi[6] + b1 * ti.qi[7] + b2 * ti.qi[8];
}

FOT_HD void lon_eval(const LonInfo &L, double t, double &s, double &sd, double &sdd, double &sddd)
{
    const double t2 = t * t, t3 = t2 * t, t4 = t2 * t2;
    s = L.a0 + L.a1 * t + L.a2 * t2 + L.a3 * t3 + L.a4 * t4;
    sd = L.a1 + 2.0 * L.a2 * t + 3.0 * L.a3 * t2 + 4.0 * L.a4 * t3;
    sdd = 2.0 * L.a2 + 6.0 * L.a3 * t + 12.0 * L.a4 * t2;
    sddd = 6.0 * L.a3 + 24.0 * L.a4 * t;
}

FOT_HD void lat_eval(const double *q, double t, double &d, double &dd, double &ddd, double &dddd)
{
    // Horner form of the quintic and its three derivatives (:688-691)
    d = quintic_value(q, t);
    dd = q[1] + t * (2.0 * q[2] + t * (3.0 * q[3] + t * (4.0 * q[4] + t * (5.0 * q[5]))));
    ddd = 2.0 * q[2] + t * (6.0 * q[3] + t * (12.0 * q[4] + t * (20.0 * q[5])));
    dddd = 6.0 * q[3] + t * (24.0 * q[4] + t * (60.0 * q[5]));
}

// longitudinal state of sample k of a profile, brake padding included (:483-500)
FOT_HD void lon_sample(const LonInfo &L, int k, double dt, double &s, double &sd, double &sdd, double &sddd)
{
    if (k < L.n_eval) {
        lon_eval(L, (double)k * dt, s, sd, sdd, sddd);
    } else {
        double u0, u1, u2;
        lon_eval(L, (double)(L.n_eval - 1) * dt, s, u0, u1, u2);
        sd = 0.0; sdd = 0.0; sddd = 0.0;
    }
}

// One longitudinal profile of an instance by its slot (Ti x tv grid entry, then the brake ladder): coefficients and
// sample counts; with_summary also the jerk sum over the samples and the final speed the cost reads (:703-734).
// Cheap and closed-form, so every kernel that needs a profile derives it on the spot -- no table of them in HBM.
FOT_HD LonInfo profile_info(const DevParams &P, const InstDesc &D, const double *fr, int slot, bool with_summary)
{
    LonInfo L;
    const int n_grid_lon = P.n_ti * D.n_tv;
    if (slot < n_grid_lon) {
        const int ti = slot / D.n_tv, itv = slot - ti * D.n_tv;
        lon_coeffs(fr, tv_value(P, D, itv), P.ti[ti], L);
        L.n_t = P.ti[ti].n_t;
        L.n_eval = L.n_t;
    } else {
        const TimeInfo &tb = P.brake[slot - n_grid_lon];
        lon_coeffs(fr, 0.0, tb, L);
        L.n_t = P.n_total;
        L.n_eval = tb.n_t;
    }
    L.Js = 0.0; L.sd_last = 0.0;
    if (with_summary) {
#if defined(__clang__)
#pragma clang fp contract(off)                           // part of the cost: see lateral_jerk_sum
#endif
        // sum over the polynomial samples k = 0..n-1 of jerk(t_k)^2 with jerk(t) = 6 a3 + 24 a4 t, t_k = k dt, in closed
        // form (the jerk is zero on the brake padding): n c0^2 + 2 c0 c1 sum(k) + c1^2 sum(k^2), c1 = 24 a4 dt
        const double n = (double)L.n_eval, c0 = 6.0 * L.a3, c1 = 24.0 * L.a4 * P.dt;
        const double sum_k = n * (n - 1.0) * 0.5, sum_k2 = (n - 1.0) * n * (2.0 * n - 1.0) / 6.0;
        L.Js = n * c0 * c0 + 2.0 * c0 * c1 * sum_k + c1 * c1 * sum_k2;
        // final speed: the quartic's derivative at the last sample, 0 on a brake profile's padding (lon_sample)
        if (L.n_t - 1 < L.n_eval) {
            const double t = (double)(L.n_t - 1) * P.dt, t2 = t * t, t3 = t2 * t;
            L.sd_last = L.a1 + 2.0 * L.a2 * t + 3.0 * L.a3 * t2 + 4.0 * L.a4 * t3;
        } else {
            L.sd_last = 0.0;
        }
    }
    return L;
}

// ---------------------------------------------------------------------------
// tiles: the unit of work of k_evaluate
// ---------------------------------------------------------------------------

// rows of LDS the profile in `slot` needs: one per sample; a brake-ladder profile holds its last state after n_eval
// samples, so n_eval rows plus ONE row for the hold (lanes clamp their row index)
FOT_HD int profile_rows(const DevParams &P, const InstDesc &D, int slot)
{
    const int n_grid_lon = P.n_ti * D.n_tv;
    if (slot < n_grid_lon) return P.ti[slot / D.n_tv].n_t;
    const int ne = P.brake[slot - n_grid_lon].n_t;
    return ne < P.n_total ? ne + 1 : ne;
}

// Two cuts of a lattice into tiles.
//  * per-wave rows: a tile's profiles must fit the rows ONE wave stages for itself -- three full-length profiles and a
//    little more (the seven brake-ladder entries of the default lattice then share one tile); 12 waves per CU x
//    (rows x 72 B + summaries) stay below the 160 KB of LDS: three waves per SIMD.
//  * groups: four consecutive tiles (one workgroup of k_evaluate) share ONE row table of GROUP_ROWS rows and
//    GROUP_MAX_PROFILES profiles, so a tile is 64 candidates wherever the lattice has them, and four such workgroups
//    share the CU's LDS: four waves per SIMD (k_evaluate then has to make do with 128 vector registers, which it
//    does).  A shape's tile list is padded with empty tiles (n = 0) so that group g is tiles [4g, 4g + 4).
// The handle picks the cut by the number of tiles either one makes of its lattice (build_tile_shapes).
#ifndef FOT_GROUP_TILES
#define FOT_GROUP_TILES 4
#endif
constexpr int GROUP_TILES = FOT_GROUP_TILES, GROUP_ROWS = 128 * FOT_GROUP_TILES, GROUP_MAX_PROFILES = 4 * FOT_GROUP_TILES;
FOT_HD int tile_row_budget(int n_total)
{
    const int want = 3 * n_total + 8, cap = 176;
    return want < cap ? want : (cap > n_total ? cap : n_total);
}

// Tile that starts at candidate `cand0` of an instance: n consecutive candidates (1..64, the candidates of the
// generation order Ti -> tv -> di, brake ladder last), spanning at most TILE_MAX_PROFILES longitudinal profiles whose
// rows fit `row_budget`.  Depends on the lattice shape only (planner constants + the instance's terminal-speed grid),
// so the host (tile counts), k_frenet_state (tile table) and the CPU logic test walk the same sequence.
FOT_HD int tile_extent(const DevParams &P, const InstDesc &D, int cand0, int row_budget)
{
    const int n_grid_lon = P.n_ti * D.n_tv;
    int n = 0, rows = 0, profs = 0, c = cand0;
    while (n < WAVE && c < D.n_cand_max && profs < TILE_MAX_PROFILES) {
        int slot, left;                                   // profile of candidate c, candidates of it from c on
        if (c < D.n_grid) { slot = c / P.n_di; left = (slot + 1) * P.n_di - c; }
        else { slot = n_grid_lon + (c - D.n_grid); left = 1; }
        const int r = profile_rows(P, D, slot);
        if (profs > 0 && rows + r > row_budget) break;    // (a single profile always fits: row_budget >= n_total)
        rows += r; ++profs;
        const int take = left < WAVE - n ? left : WAVE - n;
        n += take; c += take;
    }
    return n;
}

FOT_HD int count_tiles(const DevParams &P, const InstDesc &D, int row_budget)
{
    int t = 0;
    for (int c = 0; c < D.n_cand_max; ++t) c += tile_extent(P, D, c, row_budget);
    return t;
}

// ---------------------------------------------------------------------------
// Frenet -> Cartesian of one sample
// (reference: frenet_planner.py:792-799, coordinate_converter.py:128-158)
// ---------------------------------------------------------------------------

struct LonSample {
    double s, sd, sdd, rx, ry, cos_r, sin_r, kr, dkr;
    double inv_sd;                       // 1/sd, or 0 below EPS_S_DOT (d' = d'' = 0 there, frenet_planner.py:792-799)
};

struct CartSample {
    double x, y, cos_t, sin_t, kappa, v, a, omkd;
};

// One reciprocal and one reciprocal square root per sample: with h = sqrt(d'^2 + (1-kd)^2) the
// reference's atan2/cos/tan terms are cos = (1-kd)/h, sin = d'/h, tan = d'/(1-kd), (1-kd)/cos = h and
// v = sqrt((1-kd)^2 sd^2 + (d' sd)^2) = |sd| h.
FOT_HD void frenet_to_cart(const LonSample &L, double d, double d_d, double d_dd, CartSample &o)
{
    const double dp = d_d * L.inv_sd;
    const double dpp = (d_dd - dp * L.sdd) * (L.inv_sd * L.inv_sd);
    const double omkd = 1.0 - L.kr * d;
    const double inv_om = fast_rcp(omkd);
    const double hh = dp * dp + omkd * omkd;
    const double inv_h = fast_rsqrt(hh);
    const double h = hh * inv_h;
    const double cos_d = omkd * inv_h, sin_d = dp * inv_h;
    const double tan_d = dp * inv_om;
    const double inv_cos = h * inv_om;
    const double krdp = L.dkr * d + L.kr * dp;
    // kappa = ((d'' + krdp tan) cos^2 / (1-kd) + kr) cos / (1-kd); with cos = (1-kd)/h the factors (1-kd) cancel:
    //       = ((d'' + krdp tan) (1-kd) / h^2 + kr) / h
    const double kappa = ((dpp + krdp * tan_d) * omkd * (inv_h * inv_h) + L.kr) * inv_h;
    const double dtp = h * kappa - L.kr;
    o.x = L.rx - L.sin_r * d;
    o.y = L.ry + L.cos_r * d;
    o.cos_t = cos_d * L.cos_r - sin_d * L.sin_r;
    o.sin_t = sin_d * L.cos_r + cos_d * L.sin_r;
    o.kappa = kappa;
    o.v = fabs(L.sd) * h;
    o.a = L.sdd * h + L.sd * L.sd * inv_cos * (dp * dtp - krdp);
    o.omkd = omkd;
}

FOT_HD void load_lon_sample(const double *tab, int k, LonSample &L)
{
    L.s = tab[0 * FOT_MAX_NT + k];   L.sd = tab[1 * FOT_MAX_NT + k];    L.sdd = tab[2 * FOT_MAX_NT + k];
    L.rx = tab[3 * FOT_MAX_NT + k];  L.ry = tab[4 * FOT_MAX_NT + k];
    L.cos_r = tab[5 * FOT_MAX_NT + k]; L.sin_r = tab[6 * FOT_MAX_NT + k];
    L.kr = tab[7 * FOT_MAX_NT + k];  L.dkr = tab[8 * FOT_MAX_NT + k];
    L.inv_sd = tab[9 * FOT_MAX_NT + k];
}

// spline_frame with ONE reciprocal square root instead of a square root and five divisions: with r = (x'^2 + y'^2)^-1/2
// the tangent is (x', y') r, kappa = a r^3 and kappa' = (b - 3 a c r^2) r^3.  Every tile of k_evaluate rebuilds the rows of
// its profiles, so the frame costs as much as it is allowed to: the Newton-refined r is good to a few units in the
// last place, which is what the per-sample transform (fast_rcp / fast_rsqrt too) works with anyway.
FOT_HD void spline_frame_fast(const SplinePt &p, double &cos_r, double &sin_r, double &kappa, double &dkappa)
{
    const double d = p.dx * p.dx + p.dy * p.dy;
    const double r = fast_rsqrt(d);
    const double r2 = r * r, r3 = r2 * r;
    cos_r = p.dx * r;
    sin_r = p.dy * r;
    const double a = p.dx * p.ddy - p.dy * p.ddx;
    const double b = p.dx * p.dddy - p.dy * p.dddx;
    const double c = p.dx * p.ddx + p.dy * p.ddy;
    kappa = a * r3;
    dkappa = (b - 3.0 * a * c * r2) * r3;
}

// one entry of the longitudinal table: profile state + reference frame at s
FOT_HD void make_lon_sample(const SplineView &sp, const LonInfo &L, int k, double dt, LonSample &o, double &sddd)
{
    lon_sample(L, k, dt, o.s, o.sd, o.sdd, sddd);
    SplinePt p;
    spline_point(sp, o.s, p);
    o.rx = p.x; o.ry = p.y;
    spline_frame_fast(p, o.cos_r, o.sin_r, o.kr, o.dkr);
    o.inv_sd = fabs(o.sd) > 1e-3 ? fast_rcp(o.sd) : 0.0;        // EPS_S_DOT
}

// ---------------------------------------------------------------------------
// candidate evaluation: cost, truncation, kinematic checks
// (reference: frenet_planner.py:703-734, 826-887, 932-984, 995-1033)
// ---------------------------------------------------------------------------

struct CandResult {
    double cost, v_last, travel;
    int status, keep;
};

// Running state of the per-candidate checks of _check_paths (frenet_planner.py:932-984) over the
// kept prefix of a path.  One sample at a time, so the lattice kernel and the external-path entry
// (fot_check_paths) share the exact same tests.
struct PathSample {
    double x, y, cos_t, sin_t, kappa, v, a, d;
};

// What the per-sample loop reads of the planner and instance constants, by value: k_evaluate keeps these in
// registers instead of re-fetching them from the parameter blocks inside the loop.
struct LoopConst {
    double dt;
    double lim_speed, lim_accel, lim_curv, lim_lat;
    double road_lim;                     // max_road_width + 1e-9 (:982)
    int n_circ_fp;                       // footprint circles, 0: the single centre circle
};

FOT_HD LoopConst loop_const(const DevParams &P, const InstDesc &D)
{
    LoopConst c;
    c.dt = P.dt;
    c.lim_speed = D.lim_speed; c.lim_accel = D.lim_accel; c.lim_curv = D.lim_curv; c.lim_lat = D.lim_lat;
    c.road_lim = P.road_lim;
    c.n_circ_fp = P.has_footprint ? P.n_circ : 0;
    return c;
}

// The per-candidate flags live in ONE per-lane word (bit per flag) rather than in one wave mask each: eleven
// 64-bit lane masks would not fit the scalar register file next to the collision chunk buffers of k_evaluate.
enum : uint32_t {
    CK_NONFINITE = 1u, CK_NANSTEP = 2u, CK_SPEED = 4u, CK_ACCEL = 8u, CK_CURV = 16u, CK_LAT = 32u, CK_ROAD = 64u,
    CK_SINGULAR = 128u, CK_SEEN_NAN = 256u,
    CK_FAILED = CK_NONFINITE | CK_SPEED | CK_ACCEL | CK_CURV | CK_LAT | CK_ROAD | CK_SINGULAR
};

struct CheckAcc {
    uint32_t fl;                         // CK_* bits
    double max_step2;                    // largest squared step between consecutive samples
    PathSample prev;
};

FOT_HD void check_init(CheckAcc &c)
{
    c.fl = 0;
    c.max_step2 = -INFINITY;
    c.prev.x = c.prev.y = c.prev.kappa = c.prev.v = c.prev.a = c.prev.d = 0.0;
    c.prev.cos_t = 1.0; c.prev.sin_t = 0.0;
}

FOT_HD void check_flag(CheckAcc &c, bool cond, uint32_t bit) { c.fl |= cond ? bit : 0u; }

// |atan2(sn, cs)| > max(lim_curv * sqrt(step2), 0.1): the yaw-step cap of the low-speed rule (:1013-1022), for the few
// steps that turn by more than 0.09 rad.  A real function on the device, on purpose: inlined into k_evaluate's loop
// the arc tangent's and the square root's nineteen-odd constants are hoisted out of the loop into vector registers
// (or, at 128 registers, into scratch memory: 60 MB of spill traffic per launch) for a branch that almost never runs.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
bool yaw_step_over_cap(double sn, double cs, double lim_curv, double step2)
{
    return fabs(atan2(sn, cs)) > fmax(lim_curv * sqrt(step2), 0.1);
}

// k = index of the sample inside the path; has_geo / has_d: the low-speed rules and the road test
// only apply when the caller's path carries the arrays they read (:1013-1022, :982).  arc_step() returns
// |s_k - s_{k-1}|; it is only asked for in the low-speed branch.
template <class ArcStep>
FOT_HD void check_sample(const LoopConst &C, CheckAcc &c, int k, const PathSample &p, bool has_geo, bool has_d,
                         const ArcStep &arc_step)
{
    check_flag(c, !(isfinite(p.v) && isfinite(p.a) && isfinite(p.kappa)), CK_NONFINITE);     // :944-946
    if (k > 0) {
        const double sx = p.x - c.prev.x, sy = p.y - c.prev.y;                                // :953-956
        const double step2 = sum_sq_unfused(sx, sy);
        check_flag(c, isnan(step2), CK_NANSTEP);
        if (step2 > c.max_step2) c.max_step2 = step2;
        check_flag(c, p.v > C.lim_speed, CK_SPEED);                                           // :964
        check_flag(c, fabs(p.a) > C.lim_accel, CK_ACCEL);                                     // :966
        if (p.v > 0.5) {                                                                       // LOW_SPEED_CURVATURE_GATE
            check_flag(c, fabs(p.kappa) > C.lim_curv, CK_CURV);
        } else if (has_geo) {
            const double dd = fabs(p.d - c.prev.d);
            const double d_s = arc_step();
            check_flag(c, dd > fmax(1.5 * d_s, 0.02), CK_CURV);                               // lateral slip
            const double sn = p.sin_t * c.prev.cos_t - p.cos_t * c.prev.sin_t;                // sin/cos of the yaw step
            const double cs = p.cos_t * c.prev.cos_t + p.sin_t * c.prev.sin_t;
            // yaw-step cap: the cap is at least 0.1 rad and |atan2(sn, cs)| <= |sn| / cs for cs > 0, so a step with
            // |sn| <= 0.09 cs can never exceed it -- the arc tangent (and the square root) only for the others
            if (!(cs > 0.0 && fabs(sn) <= 0.09 * cs)) check_flag(c, yaw_step_over_cap(sn, cs, C.lim_curv, step2), CK_CURV);
        }
        check_flag(c, p.v * p.v * fabs(p.kappa) > C.lim_lat, CK_LAT);                         // :975
        check_flag(c, has_d && fabs(p.d) > C.road_lim, CK_ROAD);                 // :982
    }
    c.prev = p;
}

// first failing category in the reference's order; ST_PENDING = collision check outstanding
FOT_HD int check_status(const InstDesc &D, const CheckAcc &c, int keep)
{
    if (keep == 0 || (c.fl & CK_NONFINITE) || (!(c.fl & CK_NANSTEP) && sqrt(c.max_step2) > D.step_limit))
        return FOT_ST_DROPPED;
    if (c.fl & CK_SPEED) return FOT_ST_SPEED;
    if (c.fl & CK_ACCEL) return FOT_ST_ACCEL;
    if (c.fl & CK_CURV) return FOT_ST_CURVATURE;
    if (c.fl & CK_LAT) return FOT_ST_LAT_ACCEL;
    if (c.fl & CK_ROAD) return FOT_ST_ROAD;
    return ST_PENDING;
}

// lateral state of sample k: polynomial up to n_eval-1, then held (brake padding)
FOT_HD void lat_sample(const double *q, int k, int n_eval, double dt, double &d, double &dd, double &ddd, double &dddd)
{
    if (k < n_eval) {
        lat_eval(q, (double)k * dt, d, dd, ddd, dddd);
    } else {
        double u0, u1, u2;
        lat_eval(q, (double)(n_eval - 1) * dt, d, u0, u1, u2);
        dd = 0.0; ddd = 0.0; dddd = 0.0;
    }
}

// sum over the polynomial samples k = 0..n_eval-1 of the squared lateral jerk, jerk(t) = 6 q3 + 24 q4 t + 60 q5 t^2,
// t_k = k dt, in closed form (the jerk is zero on the brake padding): with c0 + c1 k + c2 k^2 it is
// c0^2 S0 + 2 c0 c1 S1 + (c1^2 + 2 c0 c2) S2 + 2 c1 c2 S3 + c2^2 S4, S_j = sum k^j (exact integers in double).
// The same value however the time range of a candidate is walked (one piece, or segments by different waves).
FOT_HD double lateral_jerk_sum(const double *q, int n_eval, double dt)
{
    // (every product and sum rounded on its own: whether the compiler contracts a particular a * b + c into a fused
    // multiply-add depends on the code around it, and the cost must not depend on which kernel evaluates it)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double n = (double)n_eval, c0 = 6.0 * q[3], c1 = 24.0 * q[4] * dt, c2 = 60.0 * q[5] * dt * dt;
    const double s1 = n * (n - 1.0) * 0.5, s2 = (n - 1.0) * n * (2.0 * n - 1.0) / 6.0, s3 = s1 * s1;
    const double s4 = (n - 1.0) * n * (2.0 * n - 1.0) * (3.0 * n * n - 3.0 * n - 1.0) / 30.0;
    return c0 * c0 * n + 2.0 * c0 * c1 * s1 + (c1 * c1 + 2.0 * c0 * c2) * s2 + 2.0 * c1 * c2 * s3 + c2 * c2 * s4;
}

// Sink protocol (every call site is reached with a wave-uniform k):
//   row_begin(k), row_end(k)      once per time step, by every lane of the wave (n_loop is wave-uniform, >= n_t)
//   put(k, circle, x, y, alive)   collision point of sample k of the kept prefix; alive == false when the candidate
//                                 has already failed a check, i.e. can no longer end as "collision check outstanding"
// and collided() tells whether the points handed over so far violate the (chance) constraint.
// Tab::load(k, LonSample&) returns row k of the candidate's longitudinal profile: state and reference frame at s(t_k).
// GlobalTab reads a [field][FOT_MAX_NT] table from memory, ComputeTab evaluates the row on the spot (rows past the
// profile's n_t are never used and come back unspecified).
struct GlobalTab {
    static constexpr bool LOCAL = false;                 // rows in the caller's frame (k_evaluate's rows: instance-local)
    const double *tab;
    FOT_HD void load(int k, LonSample &L) const { load_lon_sample(tab, k, L); }
    FOT_HD double s_at(int k) const { return tab[k]; }
};

struct ComputeTab {
    static constexpr bool LOCAL = false;
    SplineView sp;
    LonInfo L;
    double dt;
    FOT_HD void load(int k, LonSample &o) const
    {
        double sddd;
        if (k < L.n_t) make_lon_sample(sp, L, k, dt, o, sddd);
        else { o.s = o.sd = o.sdd = o.rx = o.ry = o.cos_r = o.sin_r = o.kr = o.dkr = o.inv_sd = 0.0; }
    }
    FOT_HD double s_at(int k) const
    {
        double s_, u0, u1, u2;
        lon_sample(L, k, dt, s_, u0, u1, u2);
        return s_;
    }
};

// What a candidate carries from one time step to the next, apart from the sink's collision state.  A candidate's
// time range may be cut into consecutive segments evaluated independently (k_evaluate does that for a handful of
// egos, where a tile alone on its SIMD is a chain of dependent steps): every field merges associatively
// (seg_merge), and a segment that starts at k0 > 0 first rebuilds sample k0 - 1 as its predecessor.
struct SegState {
    CheckAcc acc;
    double d_last, v_last;
    int first_nan, k_last;
};

FOT_HD void seg_init(SegState &g)
{
    check_init(g.acc);
    g.d_last = 0.0; g.v_last = 0.0;
    g.first_nan = -1; g.k_last = -1;
}

// time steps [k0, k1) of one candidate
template <class Tab, class Sink>
FOT_HD void evaluate_segment(const DevParams &P, const LoopConst &C, const LonInfo &L, const Tab &lon_tab,
                             const double *q, int k0, int k1, Sink &sink, SegState &g)
{
    const int n_t = L.n_t;
    CheckAcc &acc = g.acc;
    if (k0 > 0 && k0 - 1 < n_t) {                        // predecessor of the segment's first sample
        LonSample ls;
        lon_tab.load(k0 - 1, ls);
        double d, d_d, d_dd, d_ddd;
        lat_sample(q, k0 - 1, L.n_eval, C.dt, d, d_d, d_dd, d_ddd);
        CartSample c;
        frenet_to_cart(ls, d, d_d, d_dd, c);
        acc.prev.x = c.x; acc.prev.y = c.y; acc.prev.cos_t = c.cos_t; acc.prev.sin_t = c.sin_t;
        acc.prev.kappa = c.kappa; acc.prev.v = c.v; acc.prev.a = c.a; acc.prev.d = d;
    }
#if defined(__HIP_DEVICE_COMPILE__) && defined(FOT_EVAL_UNROLL)
#pragma unroll FOT_EVAL_UNROLL
#endif
    for (int k = k0; k < k1; ++k) {
      // table row first (every lane: rows past n_t exist and are ignored), then the sink's per-step prologue:
      // k_evaluate issues its scalar warm-up loads there, after the row has arrived
      LonSample ls;
      lon_tab.load(k, ls);
      sink.row_begin(k);
      if (k < n_t) {
        double d, d_d, d_dd, d_ddd;
        lat_sample(q, k, L.n_eval, C.dt, d, d_d, d_dd, d_ddd);
        g.d_last = d;
        CartSample c;
        frenet_to_cart(ls, d, d_d, d_dd, c);
        check_flag(acc, isfinite(c.omkd) && c.omkd <= 0.05, CK_SINGULAR);   // SINGULARITY_EPS, any sample
        if (!(acc.fl & CK_SEEN_NAN) && isnan(c.x)) { acc.fl |= CK_SEEN_NAN; g.first_nan = k; }
        if (!(acc.fl & CK_SEEN_NAN)) {
            PathSample ps;
            ps.x = c.x; ps.y = c.y; ps.cos_t = c.cos_t; ps.sin_t = c.sin_t; ps.kappa = c.kappa;
            ps.v = c.v; ps.a = c.a; ps.d = d;
            check_sample(C, acc, k, ps, true, true, [&] { return fabs(lon_tab.s_at(k) - lon_tab.s_at(k - 1)); });
            const bool alive = (acc.fl & CK_FAILED) == 0;
            if (C.n_circ_fp > 0) {
                for (int ci = 0; ci < C.n_circ_fp; ++ci)
                    sink.put(k, ci, c.x + P.circ_off[ci] * c.cos_t, c.y + P.circ_off[ci] * c.sin_t, alive);
            } else {
                sink.put(k, 0, c.x, c.y, alive);
            }
            g.v_last = c.v; g.k_last = k;
        }
      }
      sink.row_end(k);
    }
}

// g (segments up to some k) followed by n (the segment that starts there and holds a sample below n_t).  Returns
// false when g already ended the kept prefix (a NaN sample): n's checks and collision points then do not count.
FOT_HD bool seg_merge(SegState &g, const SegState &n)
{
    g.d_last = n.d_last;
    g.acc.fl |= n.acc.fl & CK_SINGULAR;
    if (g.acc.fl & CK_SEEN_NAN) return false;
    g.acc.fl |= n.acc.fl;
    if (n.acc.max_step2 > g.acc.max_step2) g.acc.max_step2 = n.acc.max_step2;
    if (n.acc.fl & CK_SEEN_NAN) g.first_nan = n.first_nan;
    if (n.k_last >= 0) { g.k_last = n.k_last; g.v_last = n.v_last; }
    return true;
}

template <class Tab>
FOT_HD void finish_candidate(const DevParams &P, const InstDesc &D, const LonInfo &L, const Tab &lon_tab,
                             const double *q, const SegState &g, bool collided, CandResult &out)
{
#if defined(__clang__)
#pragma clang fp contract(off)                           // the cost: the same bits from every kernel (lateral_jerk_sum)
#endif
    const int n_t = L.n_t;
    int keep = n_t;
    if (g.acc.fl & CK_SEEN_NAN) keep = g.first_nan >= 2 ? g.first_nan : 0;
    if (g.acc.fl & CK_SINGULAR) keep = 0;

    const double Jd = g.d_last * g.d_last;
    const double dv = D.target_speed - L.sd_last;
    const double Jt = (double)(n_t - 1) * P.dt;
    const double lat = P.k_j * lateral_jerk_sum(q, L.n_eval, P.dt) + P.k_t * Jt + P.k_d * Jd;
    const double lon = P.k_j * L.Js + P.k_t * Jt + P.k_s_dot * (dv * dv);
    out.cost = P.k_lat * lat + P.k_lon * lon;

    int st = check_status(D, g.acc, keep);
    if (st == ST_PENDING && collided) st = FOT_ST_COLLISION;              // only candidates that pass everything else
    out.status = st;
    out.keep = keep;
    out.v_last = g.v_last;
    out.travel = g.k_last >= 0 ? lon_tab.s_at(g.k_last) - lon_tab.s_at(0) : 0.0;   // NaN is sticky: sample 0 was valid
}

template <class Tab, class Sink>
FOT_HD void evaluate_candidate(const DevParams &P, const InstDesc &D, const LoopConst &C, const LonInfo &L,
                               const Tab &lon_tab, const double *q, int n_loop, Sink &sink, CandResult &out)
{
    SegState g;
    seg_init(g);
    evaluate_segment(P, C, L, lon_tab, q, 0, n_loop, sink, g);
    finish_candidate(P, D, L, lon_tab, q, g, sink.collided(), out);
}

// ---------------------------------------------------------------------------
// collision (reference: frenet_planner.py:1035-1233)
// ---------------------------------------------------------------------------

// obstacles of one instance in the caller's layout: static [n][2], dynamic [S][P][T][2]
struct ObstacleView {
    const void *stat;
    const void *dyn;
    int dtype;                         // FOT_F32 | FOT_F64
    const uint8_t *nan_track = nullptr;   // [S * P] 1: the pedestrian's track holds a NaN somewhere -- no obstacle at any
                                          // step (frenet_planner.py:1211-1219); nullptr: the tensor is already clean
    FOT_HD d2 at(const void *base, int64_t i) const
    {
        d2 o;
        if (dtype == FOT_F32) { o.x = (double)((const float *)base)[2 * i]; o.y = (double)((const float *)base)[2 * i + 1]; }
        else { o.x = ((const double *)base)[2 * i]; o.y = ((const double *)base)[2 * i + 1]; }
        return o;
    }
    FOT_HD d2 static_at(int j) const { return at(stat, j); }
    FOT_HD d2 dyn_at(int s, int p, int row, int P, int T) const { return at(dyn, ((int64_t)s * P + p) * T + row); }
};

// a*a + b*b with every operation rounded on its own -- NumPy's np.sum(diff ** 2, axis=2) squares, then adds
// (frenet_planner.py:1196-1197, 1231-1232); the device compiler would otherwise contract the sum into an FMA
FOT_HD double sum_sq_unfused(double a, double b)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double aa = a * a, bb = b * b;
    return aa + bb;
}

FOT_HD bool within(const d2 &o, double px, double py, double sq)       // (:1196-1198, :1231-1233)
{
    return sum_sq_unfused(px - o.x, py - o.y) <= sq;
}

// Exact per-candidate check, straight from the definition.  Source::get(k, circle, x, y) returns the
// collision points; Source::tindex(k) is round(t_k/dt) (== k for lattice candidates).
// Returns true when the candidate violates the (chance) constraint.
template <class Source>
FOT_HD bool collide_candidate(const DevParams &P, const InstDesc &D, const ObstacleView &obs, int keep, const Source &src)
{
    const int n_circ = P.has_footprint ? P.n_circ : 1;
    for (int k = 0; k < keep && D.n_static > 0; ++k)
        for (int ci = 0; ci < n_circ; ++ci) {
            double px, py;
            src.get(k, ci, px, py);
            for (int j = 0; j < D.n_static; ++j)
                if (within(obs.static_at(j), px, py, P.sq_r)) return true;
        }
    if (D.dyn_mode == FOT_DYN_NONE || D.P <= 0 || D.T <= 0) return false;
    const double sq = D.dyn_mode == FOT_DYN_SINGLE ? P.sq_r_dyn : P.sq_r;
    uint64_t hit_mask = 0;
    int viol = 0;
    for (int k = 0; k < keep; ++k) {
        int row = src.tindex(k);                                            // clip(round(t/dt), 0, T-1)
        row = row < 0 ? 0 : (row > D.T - 1 ? D.T - 1 : row);
        for (int ci = 0; ci < n_circ; ++ci) {
            double px, py;
            src.get(k, ci, px, py);
            for (int s = 0; s < D.S; ++s) {
                if ((hit_mask >> s) & 1) continue;
                bool hit = false;
                for (int p = 0; p < D.P; ++p) {
                    if (obs.nan_track && obs.nan_track[s * D.P + p]) continue;
                    if (within(obs.dyn_at(s, p, row, D.P, D.T), px, py, sq)) hit = true;
                }
                if (hit) {
                    hit_mask |= (uint64_t)1 << s;
                    if (++viol > D.max_viol) return true;
                }
            }
        }
    }
    return false;
}

// ---- broad phase ------------------------------------------------------------------------------
// Two conservative reductions in front of the exact test, neither of which can change a decision:
//  1. per (instance, time step k): the float32 bounding box of every candidate's collision points
//     at k.  Only obstacles (of the time row of k) inside that box grown by the collision radius can
//     touch any candidate at k; they are compacted into an entry list (k_cull).
//  2. per entry chunk: float32 squared distances in the instance-local frame against
//     filter_threshold(); only chunks that come within the threshold are re-checked in float64.

struct Box32 {
    float x0, y0, x1, y1;                                                   // empty when x0 > x1
};

FOT_HD Box32 box_empty()
{
    Box32 b; b.x0 = INFINITY; b.y0 = INFINITY; b.x1 = -INFINITY; b.y1 = -INFINITY; return b;
}

FOT_HD void box_add(Box32 &b, float x, float y)
{
    b.x0 = fminf(b.x0, x); b.y0 = fminf(b.y0, y); b.x1 = fmaxf(b.x1, x); b.y1 = fmaxf(b.y1, y);
}

FOT_HD void box_merge(Box32 &b, const Box32 &o)
{
    b.x0 = fminf(b.x0, o.x0); b.y0 = fminf(b.y0, o.y0); b.x1 = fmaxf(b.x1, o.x1); b.y1 = fmaxf(b.y1, o.y1);
}

// growth of the box: collision radius + float32 rounding of both points + slack
FOT_HD float cull_margin(double max_sq, const Box32 &b)
{
    return sqrtf((float)max_sq) * 1.000001f + 1e-3f
           + 2.4e-7f * (fabsf(b.x0) + fabsf(b.x1) + fabsf(b.y0) + fabsf(b.y1));
}

FOT_HD bool cull_inside(const Box32 &b, float m, float fx, float fy)
{
    return fx >= b.x0 - m && fx <= b.x1 + m && fy >= b.y0 - m && fy <= b.y1 + m;
}

// Bounding box (float32, instance-local frame) of the sample-k points of ALL lateral candidates of one
// longitudinal profile.  The quintic is affine in its target offset di (lat_coeffs), so at a fixed time the
// candidates' points lie on the segment of the path normal between the two extreme offsets; brake-ladder
// profiles have the single offset d0.  Footprint circle centres lie within max|circ_off| of these points, which
// box_footprint_slack() adds to the cull margin.  NaN points (beyond the path end) are never kept and are skipped.
// lateral offsets [d0, d1] the candidates of one horizon (d1 == d0 for a brake-ladder entry) span at sample k: the
// two extreme targets of the lateral grid.  Depends on the horizon only, not on the terminal speed: k_cull derives it
// once per (horizon, step) for all the profiles that share it.
FOT_HD void lateral_extent(const DevParams &P, const double *fr, bool brake, const TimeInfo &ti, int k, int n_eval,
                           double &d0, double &d1)
{
    double q[6], u0, u1, u2;
    lat_coeffs(fr, brake ? fr[3] : -(double)P.n_side * P.d_road_w, ti, q);
    lat_sample(q, k, n_eval, P.dt, d0, u0, u1, u2);
    d1 = d0;
    if (!brake) {
        lat_coeffs(fr, (double)(P.n_di - 1 - P.n_side) * P.d_road_w, ti, q);
        lat_sample(q, k, n_eval, P.dt, d1, u0, u1, u2);
    }
}

// the same from the two extreme targets' coefficients, solved once (they do not depend on the step)
// (q9: the low target's six coefficients, then the high target's q3..q5 -- q0..q2 are the start state, shared)
FOT_HD void lateral_extent_q(const double *q9, bool brake, int k, int n_eval, double dt, double &d0, double &d1)
{
    double u0, u1, u2;
    lat_sample(q9, k, n_eval, dt, d0, u0, u1, u2);
    d1 = d0;
    if (!brake) {
        const double q_hi[6] = { q9[0], q9[1], q9[2], q9[6], q9[7], q9[8] };
        lat_sample(q_hi, k, n_eval, dt, d1, u0, u1, u2);
    }
}

FOT_HD void lateral_extent_coeffs(const DevParams &P, const double *fr, bool brake, const TimeInfo &ti, double *q9)
{
    double q_hi[6];
    lat_coeffs(fr, brake ? fr[3] : -(double)P.n_side * P.d_road_w, ti, q9);
    lat_coeffs(fr, brake ? fr[3] : (double)(P.n_di - 1 - P.n_side) * P.d_road_w, ti, q_hi);
    q9[6] = q_hi[3]; q9[7] = q_hi[4]; q9[8] = q_hi[5];
}

// box of the two end points of the segment (rx, ry) + d * normal, d in {d0, d1}
FOT_HD Box32 segment_box(double rx, double ry, double cos_r, double sin_r, double d0, double d1, double ox, double oy)
{
    Box32 b = box_empty();
    const double x0 = rx - sin_r * d0, y0 = ry + cos_r * d0, x1 = rx - sin_r * d1, y1 = ry + cos_r * d1;
    if (x0 == x0 && y0 == y0) box_add(b, (float)(x0 - ox), (float)(y0 - oy));
    if (x1 == x1 && y1 == y1) box_add(b, (float)(x1 - ox), (float)(y1 - oy));
    return b;
}

FOT_HD Box32 profile_box(const DevParams &P, const double *fr, bool brake, const TimeInfo &ti, const LonSample &ls,
                         int k, int n_eval, double ox, double oy)
{
    double d0, d1;
    lateral_extent(P, fr, brake, ti, k, n_eval, d0, d1);
    return segment_box(ls.rx, ls.ry, ls.cos_r, ls.sin_r, d0, d1, ox, oy);
}

// The same box from a pre-computed lateral extent, with only what it needs of the reference frame: position and unit
// tangent at s(t_k) -- no curvature, and the tangent normalised with the fast reciprocal square root (the box is a
// conservative float32 hull with a margin of millimetres; 1e-15 of tangent direction is nothing to it).
FOT_HD Box32 profile_box_at(const DevParams &P, const InstDesc &D, const double *fr, const SplineView &sp, int slot,
                            int k, double d0, double d1)
{
    const LonInfo L = profile_info(P, D, fr, slot, false);
    if (k >= L.n_t) return box_empty();
    double s_, u0, u1, u2;
    lon_sample(L, k, P.dt, s_, u0, u1, u2);
    SplinePt p;
    spline_point(sp, s_, p);
    const double inv = fast_rsqrt(p.dx * p.dx + p.dy * p.dy);
    return segment_box(p.x, p.y, p.dx * inv, p.dy * inv, d0, d1, D.ego.x, D.ego.y);
}

// ... and from a profile whose quartic is already solved (k_cull keeps the instance's profiles in LDS: they do not
// depend on the step either)
struct LonQuartic {                      // what a box needs of a profile: 48 bytes instead of LonInfo's 72 (LDS)
    double a0, a1, a2, a3, a4;
    int32_t n_t, n_eval;
};

FOT_HD LonQuartic lon_quartic(const LonInfo &L)
{
    LonQuartic q;
    q.a0 = L.a0; q.a1 = L.a1; q.a2 = L.a2; q.a3 = L.a3; q.a4 = L.a4; q.n_t = L.n_t; q.n_eval = L.n_eval;
    return q;
}

FOT_HD Box32 profile_box_from(const LonQuartic &Q, const InstDesc &D, const SplineView &sp, int k, double dt, double d0,
                              double d1)
{
    if (k >= Q.n_t) return box_empty();
    LonInfo L;
    L.a0 = Q.a0; L.a1 = Q.a1; L.a2 = Q.a2; L.a3 = Q.a3; L.a4 = Q.a4; L.n_t = Q.n_t; L.n_eval = Q.n_eval;
    L.Js = 0.0; L.sd_last = 0.0; L.T = 0.0;
    double s_, u0, u1, u2;
    lon_sample(L, k, dt, s_, u0, u1, u2);
    SplinePt p;
    spline_point(sp, s_, p);
    const double inv = fast_rsqrt(p.dx * p.dx + p.dy * p.dy);
    return segment_box(p.x, p.y, p.dx * inv, p.dy * inv, d0, d1, D.ego.x, D.ego.y);
}

// index of a profile's lateral extent in a per-step table: horizons first, then the brake ladder
FOT_HD int extent_index(const DevParams &P, const InstDesc &D, int slot)
{
    const int n_grid_lon = P.n_ti * D.n_tv;
    return slot < n_grid_lon ? slot / D.n_tv : P.n_ti + (slot - n_grid_lon);
}

FOT_HD float box_footprint_slack(const DevParams &P)
{
    float m = 0.0f;
    if (P.has_footprint)
        for (int c = 0; c < P.n_circ; ++c) m = fmaxf(m, (float)fabs(P.circ_off[c]) * 1.000001f);
    return m;
}

// ---- strips: entry lists are ordered by bin along the longer side of the box, so that the 64 candidates of one
// wave (two or three neighbouring longitudinal profiles: a short stretch of road) only walk the chunks of the bins
// their own points can reach.  bin_of() is monotone in the coordinate, which is all the range argument needs.
constexpr int CULL_BINS = 32;
constexpr int ENT_CHUNK_ = 8;                                               // == ENT_CHUNK (declared below)

struct BinMap {
    int axis;                            // 0: bins along x, 1: along y
    float lo, inv_w;
};

FOT_HD BinMap bin_map(const Box32 &b, float margin)
{
    BinMap m;
    m.axis = (b.y1 - b.y0) > (b.x1 - b.x0) ? 1 : 0;
    const float lo = (m.axis ? b.y0 : b.x0) - margin, hi = (m.axis ? b.y1 : b.x1) + margin;
    m.lo = lo;
    m.inv_w = (float)CULL_BINS / fmaxf(hi - lo, 1e-6f);
    return m;
}

FOT_HD int bin_of_value(const BinMap &m, float v)
{
    const float t = (v - m.lo) * m.inv_w;
    return t >= (float)(CULL_BINS - 1) ? CULL_BINS - 1 : (t > 0.0f ? (int)t : 0);
}

FOT_HD int bin_of(const BinMap &m, float x, float y) { return bin_of_value(m, m.axis ? y : x); }

// Chunk range [c_lo, c_hi) of a list ordered by bin (bin_start[b] = first entry of bin b, bin_start[CULL_BINS] =
// entries) that holds every entry a point of box `wb` could touch, within the pair-padded list (k_evaluate loads
// chunks in pairs and may fetch -- not test -- one chunk past an odd range).  Packed as c_lo << 16 | c_hi; 0 = nothing.
template <class StartFn>
FOT_HD uint32_t strip_range(const BinMap &m, const Box32 &wb, float margin, const StartFn &bin_start)
{
    if (!(wb.x0 <= wb.x1)) return 0u;
    const float lo = (m.axis ? wb.y0 : wb.x0) - margin, hi = (m.axis ? wb.y1 : wb.x1) + margin;
    const int b_lo = bin_of_value(m, lo), b_hi = bin_of_value(m, hi);
    const int e_lo = bin_start(b_lo), e_hi = bin_start(b_hi + 1);
    if (e_hi <= e_lo) return 0u;
    const int c_lo = e_lo / ENT_CHUNK_, c_hi = (e_hi + ENT_CHUNK_ - 1) / ENT_CHUNK_;
    return ((uint32_t)c_lo << 16) | (uint32_t)c_hi;
}

// longitudinal-profile slots [first, last] covered by the candidates [idx0, idx1] of an instance
FOT_HD void wave_profile_span(const DevParams &P, const InstDesc &D, int n_tv_grid_lon, int idx0, int idx1,
                              int &first, int &last)
{
    const int per_prof = P.n_di;
    first = idx0 < D.n_grid ? idx0 / per_prof : n_tv_grid_lon + (idx0 - D.n_grid);
    last = idx1 < D.n_grid ? idx1 / per_prof : n_tv_grid_lon + (idx1 - D.n_grid);
}

constexpr int ENT_CHUNK = 8;
static_assert(ENT_CHUNK == 8, "strip_range assumes 8-entry chunks");                                                // entries per broad-phase chunk
constexpr int SID_STATIC = 255;                                             // entry is a static obstacle
// float32 entries are stored chunk-wise as structure of arrays, x[8] then y[8] (64 B): neighbouring
// obstacles sit in neighbouring registers, which is what the packed-float32 arithmetic of k_evaluate wants
struct alignas(64) f2x8 { float x[ENT_CHUNK]; float y[ENT_CHUNK]; };

FOT_HD void ent32_store(f2 *e32, int64_t pos, float x, float y)
{
    f2x8 *c = (f2x8 *)e32 + (pos >> 3);
    c->x[pos & 7] = x;
    c->y[pos & 7] = y;
}

// smallest float32 squared distance from (fx, fy) to the 8 entries of a chunk (FAR32 padded)
FOT_HD float min_sqdist32_8(const f2x8 &c, float fx, float fy)
{
    float t[ENT_CHUNK];
    for (int j = 0; j < ENT_CHUNK; ++j) {
        const float dx = fx - c.x[j], dy = fy - c.y[j];
        t[j] = fmaf(dy, dy, dx * dx);
    }
    float m = fminf(fminf(t[0], t[1]), t[2]);
    m = fminf(fminf(m, t[3]), t[4]);
    m = fminf(fminf(m, t[5]), t[6]);
    return fminf(m, t[7]);
}

// Upper bound of the float32 squared distance of any pair whose float64 squared distance is <= sq.
// Rounding the two points to float32 moves each coordinate difference by at most
// 2^-24 (2|p| + R + |d|); with |d| <= R = sqrt(sq) the squared distance moves by less than
// 4 (R+1) e, e = 2^-23 (|px| + |py| + 2R + 6); the three float32 roundings of the sum add 2^-22 relative.
struct FilterConst { float sq, r, sq_lo; }; // (float)sq, sqrt((float)sq) + 1, and the smaller squared radius rounded down

FOT_HD FilterConst filter_const(double sq, double sq_min)
{
    FilterConst f; f.sq = (float)sq; f.r = sqrtf((float)sq) + 1.0f;
    f.sq_lo = (float)sq_min * 0.9999999f;
    return f;
}

FOT_HD FilterConst filter_const(double sq) { return filter_const(sq, sq); }

// (e: 2^-21 -- room for a point that is itself computed in float32 instead of being the rounding of its float64 value;
//  the shipped kernels hand over rounded float64 points)
FOT_HD float filter_threshold(const FilterConst &f, float px, float py)
{
    const float e = (fabsf(px) + fabsf(py) + 2.0f * f.r + 12.0f) * 4.7683716e-7f;
    return (f.sq + 4.0f * f.r * e) * 1.000002f + 1e-30f;
}

FOT_HD float filter_threshold(double sq, float px, float py) { return filter_threshold(filter_const(sq), px, py); }

// The converse bound: a pair whose float32 squared distance is <= this value has a float64 squared distance <= the
// SMALLER of the two squared radii, whatever the obstacle's kind -- a certain hit that needs no float64 re-check.
// (A float32 value below sq means |d| < R + 1, so the same rounding bound applies; <= 0 when nothing is certain.)
FOT_HD float filter_threshold_sure(const FilterConst &f, float px, float py)
{
    const float e = (fabsf(px) + fabsf(py) + 2.0f * f.r + 12.0f) * 4.7683716e-7f;
    return (f.sq_lo - 4.0f * f.r * e) * 0.999998f - 1e-30f;
}

// Both thresholds for EVERY point of a float32 box (instance-local frame) grown by m: filter_threshold grows and
// filter_threshold_sure shrinks with |px| + |py|, so their values at the box's bound of |x| + |y| hold for each point
// inside.  k_cull evaluates this once per (tile, time step) for the box of the tile's profiles grown by the cull
// margin (collision radius + rounding slack + footprint offsets: the collision points of the tile's candidates lie
// inside); k_evaluate then compares against two wave-uniform values instead of deriving them per sample.
FOT_HD void box_thresholds(const FilterConst &f, const Box32 &b, float m, float &thr, float &thr_sure)
{
    const float bx = fmaxf(fabsf(b.x0), fabsf(b.x1)) + m, by = fmaxf(fabsf(b.y0), fabsf(b.y1)) + m;
    thr = filter_threshold(f, bx, by);
    thr_sure = filter_threshold_sure(f, bx, by);
}

// exact float64 test of one chunk (the reference's test); updates the per-sample hit state
FOT_HD void exact_chunk(const d2 *e64, const uint8_t *sid, double px, double py, double sq_static, double sq_dyn,
                        int max_viol, uint64_t &hit_mask, int &viol, bool &collided)
{
    for (int j = 0; j < ENT_CHUNK; ++j) {
        const int s = sid[j];
        if (s == SID_STATIC) {
            if (within(e64[j], px, py, sq_static)) collided = true;         // static obstacles are hard constraints
        } else if (!((hit_mask >> s) & 1) && within(e64[j], px, py, sq_dyn)) {
            hit_mask |= (uint64_t)1 << s;
            if (++viol > max_viol) collided = true;
        }
    }
}

// The same decision with the float64 coordinates touched only where float32 cannot tell: all float32 coordinates
// and sample ids of the chunk are fetched in one go (72 contiguous bytes), every entry is classified by its float32
// squared distance -- above `thr`: miss; at or below `thr_sure` (filter_threshold_sure: within the SMALLER radius for
// certain): hit of whatever kind; in between: the reference's float64 expression on that one entry.
FOT_HD void exact_chunk_f32first(const f2x8 &c32, const d2 *e64, const uint8_t *sid, float fx, float fy, float thr,
                                 float thr_sure, double px, double py, double sq_static, double sq_dyn, int max_viol,
                                 uint64_t &hit_mask, int &viol, bool &collided)
{
    uint8_t s8[ENT_CHUNK];
    for (int j = 0; j < ENT_CHUNK; ++j) s8[j] = sid[j];
    for (int j = 0; j < ENT_CHUNK; ++j) {
        const float dx = fx - c32.x[j], dy = fy - c32.y[j];
        const float d32 = fmaf(dy, dy, dx * dx);
        if (d32 > thr) continue;                                             // certainly outside either radius
        const int s = s8[j];
        const bool is_static = s == SID_STATIC;
        if (!is_static && ((hit_mask >> s) & 1)) continue;                   // this sample already counts
        const bool in = d32 <= thr_sure ? true : within(e64[j], px, py, is_static ? sq_static : sq_dyn);
        if (!in) continue;
        if (is_static) collided = true;                                      // static obstacles are hard constraints
        else {
            hit_mask |= (uint64_t)1 << s;
            if (++viol > max_viol) collided = true;
        }
    }
}

// Collision state of one candidate, fed sample by sample from evaluate_candidate and tested against the culled
// entry lists of its instance (same decision as collide_candidate).  cnt[k]: entries of time step k (multiple of 8);
// the entry arrays hold ent_cap slots per k.  This is the portable form; k_evaluate's sink is the same logic with
// the chunk walk on scalar loads.
struct EntryCollider {
    const uint32_t *rng;                 // [n_total] strip ranges of this candidate's wave, nullptr: no obstacles
    const float *thr_k = nullptr, *thr_sure_k = nullptr;   // [n_total] the tile's thresholds per step (box_thresholds),
                                                           // nullptr: derived from the point itself
    const f2 *e32; const d2 *e64; const uint8_t *sid;                       // of this instance
    int ent_cap, max_viol;
    double ox, oy, sq_static, sq_dyn, sq_max;
    uint64_t hit_mask;
    int viol;
    bool hit;
    FOT_HD void init(const DevParams &P, const InstDesc &D)
    {
        ent_cap = D.ent_cap; max_viol = D.max_viol;
        ox = D.ego.x; oy = D.ego.y;
        sq_static = P.sq_r;
        sq_dyn = D.dyn_mode == FOT_DYN_SINGLE ? P.sq_r_dyn : P.sq_r;
        sq_max = sq_dyn > P.sq_r ? sq_dyn : P.sq_r;
        hit_mask = 0; viol = 0; hit = false;
    }
    FOT_HD void row_begin(int) {}
    FOT_HD void row_end(int) {}
    FOT_HD void put(int k, int ci, double px, double py, bool alive)
    {
        put32(k, ci, (float)(px - ox), (float)(py - oy), alive, [&](double &ex, double &ey) { ex = px; ey = py; });
    }
    // the point in the instance-local float32 frame; get_exact(px, py): its float64 coordinates, where they are needed
    template <class GetExact>
    FOT_HD void put32(int k, int, float fx, float fy, bool alive, const GetExact &get_exact)
    {
        if (!rng || !alive || hit) return;
        const int c_lo = (int)(rng[k] >> 16), c_hi = (int)(rng[k] & 0xffffu);
        const int64_t base = (int64_t)k * ent_cap;
        const FilterConst fc = filter_const(sq_max, sq_dyn < sq_static ? sq_dyn : sq_static);
        const float thr = thr_k ? thr_k[k] : filter_threshold(fc, fx, fy);
        const float thr_sure = thr_sure_k ? thr_sure_k[k] : filter_threshold_sure(fc, fx, fy);
        for (int c = c_lo * ENT_CHUNK; c < c_hi * ENT_CHUNK && !hit; c += ENT_CHUNK) {
            const float m = min_sqdist32_8(*(const f2x8 *)(e32 + base + c), fx, fy);
            if (m > thr) continue;
            if (max_viol == 0 && m <= thr_sure) { hit = true; break; }      // certain hit: one violation is fatal
            double px, py;
            get_exact(px, py);
            exact_chunk_f32first(*(const f2x8 *)(e32 + base + c), e64 + base + c, sid + base + c, fx, fy, thr, thr_sure,
                                 px, py, sq_static, sq_dyn, max_viol, hit_mask, viol, hit);
        }
    }
    FOT_HD void restart() { hit_mask = 0; viol = 0; hit = false; }
    FOT_HD bool collided() const { return hit; }
};

// ---------------------------------------------------------------------------
// selection (reference: frenet_planner.py:307-324, 1235-1259)
// ---------------------------------------------------------------------------

FOT_HD int final_status(int status, double v_last, double travel, double max_stop)
{
    if (status == FOT_ST_OK && !isnan(max_stop)) {
        const bool stops = fabs(v_last) <= 0.15;                             // STOP_SPEED_EPS
        if (!(stops && travel <= max_stop + 1e-6)) return FOT_ST_STOP_DISTANCE;
    }
    return status;
}

// decode candidate index -> (longitudinal profile slot, lateral time info, lateral target)
struct CandDecode {
    int lon_slot;          // index into the instance's longitudinal profiles
    int brake;             // 1: brake-ladder entry
    int ti;                // horizon index or brake index
    double di;             // lateral target offset
};

FOT_HD CandDecode decode_candidate(const DevParams &P, const InstDesc &D, const double *fr, int idx)
{
    CandDecode c;
    if (idx < D.n_grid) {
        const int per_ti = D.n_tv * P.n_di;
        const int ti = idx / per_ti, rem = idx - ti * per_ti;
        const int itv = rem / P.n_di, idi = rem - itv * P.n_di;
        c.lon_slot = ti * D.n_tv + itv;
        c.brake = 0; c.ti = ti;
        c.di = (double)(idi - P.n_side) * P.d_road_w;
    } else {
        const int b = idx - D.n_grid;
        c.lon_slot = P.n_ti * D.n_tv + b;
        c.brake = 1; c.ti = b;
        c.di = fr[3];
    }
    return c;
}

// yaw of the selected path's samples.  A real function on the device, like yaw_step_over_cap and for the same reason:
// inlined behind the evaluation kernels' time-step loop (the selection runs there), the arc tangent's constants are
// hoisted in front of the loop and live across it.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
double yaw_from_sin_cos(double sn, double cs) { return atan2(sn, cs); }

// the 15 FrenetPath values of sample k of one candidate (data_structures.py:164-178 order)
template <class Tab>
FOT_HD void final_sample(const DevParams &P, const LonInfo &L, const Tab &lon_tab, const double *q, int k,
                         double *o /*[15]*/)
{
    double s, sd, sdd, sddd, d, d_d, d_dd, d_ddd;
    lon_sample(L, k, P.dt, s, sd, sdd, sddd);
    lat_sample(q, k, L.n_eval, P.dt, d, d_d, d_dd, d_ddd);
    LonSample ls;
    lon_tab.load(k, ls);
    CartSample c;
    frenet_to_cart(ls, d, d_d, d_dd, c);
    o[0] = (double)k * P.dt;
    o[1] = ls.s; o[2] = ls.sd; o[3] = ls.sdd; o[4] = sddd;
    o[5] = d; o[6] = d_d; o[7] = d_dd; o[8] = d_ddd;
    o[9] = c.x; o[10] = c.y; o[11] = yaw_from_sin_cos(c.sin_t, c.cos_t);
    o[12] = c.v; o[13] = c.a; o[14] = c.kappa;
}

// ---------------------------------------------------------------------------
// epsilon-band report (fot_debug_margins): how close each candidate came to any threshold
// ---------------------------------------------------------------------------

// Per candidate, per group of decisions, the smallest RELATIVE distance |value - threshold| / |threshold| over every
// comparison evaluate_candidate / the collision test / the stop filter makes on it.  A status that differs from the
// reference's while all its margins are >> float64 rounding would be a logic error; a tiny margin says the decision is
// within rounding of the threshold and may legitimately flip under a re-association.  Diagnostic only (one thread per
// candidate, rows recomputed on the spot); the entry lists of the last plan call supply the obstacles (everything
// outside them is more than the cull margin away from every point of the candidate).
enum { MG_SPEED = 0, MG_ACCEL, MG_CURV, MG_LAT, MG_ROAD, MG_COLLISION, MG_STOP, MG_STRUCT, MARGIN_CATS };
static_assert(MARGIN_CATS == FOT_MARGIN_GROUPS, "include/fot.h");

FOT_HD void margin_note(double *m, int cat, double value, double threshold)
{
    const double r = fabs(value - threshold) / fabs(threshold);
    if (r < m[cat]) m[cat] = r;                      // NaN never lowers a margin
}

struct MarginEntries {
    const int32_t *cnt;                  // [n_total] entries of each time step of this instance (nullptr: none)
    const d2 *e64; const uint8_t *sid;   // of this instance
    int ent_cap;
};

template <class Tab>
FOT_HD void candidate_margins(const DevParams &P, const InstDesc &D, const LonInfo &L, const Tab &lon_tab,
                              const double *q, const MarginEntries &ent, double *m /*[MARGIN_CATS]*/)
{
    for (int c = 0; c < MARGIN_CATS; ++c) m[c] = INFINITY;
    const LoopConst C = loop_const(P, D);
    const double sq_dyn = D.dyn_mode == FOT_DYN_SINGLE ? P.sq_r_dyn : P.sq_r;
    const int n_circ = P.has_footprint ? P.n_circ : 1;
    PathSample prev;
    prev.x = prev.y = prev.kappa = prev.v = prev.a = prev.d = 0.0; prev.cos_t = 1.0; prev.sin_t = 0.0;
    double v_last = 0.0;
    int k_last = -1;
    for (int k = 0; k < L.n_t; ++k) {
        LonSample ls;
        lon_tab.load(k, ls);
        double d, d_d, d_dd, d_ddd;
        lat_sample(q, k, L.n_eval, C.dt, d, d_d, d_dd, d_ddd);
        CartSample c;
        frenet_to_cart(ls, d, d_d, d_dd, c);
        margin_note(m, MG_STRUCT, c.omkd, 0.05);                                    // SINGULARITY_EPS
        margin_note(m, MG_STRUCT, fabs(ls.sd), 1e-3);                               // EPS_S_DOT
        if (isnan(c.x)) break;                                                       // truncation: no threshold involved
        if (k > 0) {
            const double sx = c.x - prev.x, sy = c.y - prev.y;
            const double step2 = sum_sq_unfused(sx, sy);
            margin_note(m, MG_STRUCT, sqrt(step2), D.step_limit);
            margin_note(m, MG_SPEED, c.v, C.lim_speed);
            margin_note(m, MG_ACCEL, fabs(c.a), C.lim_accel);
            margin_note(m, MG_CURV, c.v, 0.5);                                       // LOW_SPEED_CURVATURE_GATE
            if (c.v > 0.5) {
                margin_note(m, MG_CURV, fabs(c.kappa), C.lim_curv);
            } else {
                const double d_s = fabs(lon_tab.s_at(k) - lon_tab.s_at(k - 1));
                margin_note(m, MG_CURV, fabs(d - prev.d), fmax(1.5 * d_s, 0.02));
                const double sn = c.sin_t * prev.cos_t - c.cos_t * prev.sin_t;
                const double cs = c.cos_t * prev.cos_t + c.sin_t * prev.sin_t;
                margin_note(m, MG_CURV, fabs(atan2(sn, cs)), fmax(C.lim_curv * sqrt(step2), 0.1));
            }
            margin_note(m, MG_LAT, c.v * c.v * fabs(c.kappa), C.lim_lat);
            margin_note(m, MG_ROAD, fabs(d), C.road_lim);
        }
        if (ent.cnt) {
            const int n = ent.cnt[k];
            const int64_t base = (int64_t)k * ent.ent_cap;
            for (int ci = 0; ci < n_circ; ++ci) {
                const double off = P.has_footprint ? P.circ_off[ci] : 0.0;
                const double px = c.x + off * c.cos_t, py = c.y + off * c.sin_t;
                for (int j = 0; j < n; ++j) {
                    const d2 o = ent.e64[base + j];
                    if (!(fabs(o.x) < 1e300)) continue;                                // list padding
                    margin_note(m, MG_COLLISION, sum_sq_unfused(px - o.x, py - o.y),
                                ent.sid[base + j] == 255 ? P.sq_r : sq_dyn);
                }
            }
        }
        prev.x = c.x; prev.y = c.y; prev.cos_t = c.cos_t; prev.sin_t = c.sin_t; prev.kappa = c.kappa;
        prev.v = c.v; prev.a = c.a; prev.d = d;
        v_last = c.v; k_last = k;
    }
    if (!isnan(D.max_stop) && k_last >= 0) {
        margin_note(m, MG_STOP, fabs(v_last), 0.15);                                  // STOP_SPEED_EPS
        margin_note(m, MG_STOP, lon_tab.s_at(k_last) - lon_tab.s_at(0), D.max_stop + 1e-6);
    }
}

// ---------------------------------------------------------------------------
// prediction resampling (reference: src/prediction/trajectory_predictor.py:188-313)
// ---------------------------------------------------------------------------

// len(np.arange(sim_dt, max(plan_horizon, pred_len*sgan_dt) + 1e-9, sim_dt))
FOT_HD int resample_n_dense(double sgan_dt, double sim_dt, double plan_horizon, int pred_len)
{
    const double target = fmax(plan_horizon, (double)pred_len * sgan_dt);
    const double n = ceil((target + 1e-9 - sim_dt) / sim_dt);
    return n > 0.0 ? (int)n : 0;
}

// One coordinate axis of one pedestrian: n_src source values co[] at times ts(i) -> n_dense values.
// ts(i) = (i + first_k) * sgan_dt - staleness with first_k = 0 when an anchor point leads the sources, else 1.
struct ResampleAxis {
    double co[FOT_MAX_PRED_LEN + 1];
    int n_src, first_k;
    double sgan_dt, staleness;
    FOT_HD double ts(int i) const { return (double)(i + first_k) * sgan_dt - staleness; }
    // np.allclose(co, b): every |co_i - b| <= 1e-8 + 1e-5 |b|   (:292)
    FOT_HD bool all_close(double b) const
    {
        bool ok = true;
        for (int i = 0; i < n_src; ++i) ok = ok && (fabs(co[i] - b) <= 1e-8 + 1e-5 * fabs(b));
        return ok;
    }
    // np.interp at t, then the clamped-velocity tail beyond the last source (:297-311)
    FOT_HD double at(double t, bool constant, double v_tail) const
    {
        if (constant) return co[n_src - 1];
        const double t_last = ts(n_src - 1);
        if (n_src >= 2 && t > t_last) return co[n_src - 1] + v_tail * (t - t_last);
        if (t > t_last) return co[n_src - 1];
        if (t < ts(0)) return co[0];
        int j = 0;
        for (int i = 1; i < n_src; ++i) j = ts(i) <= t ? i : j;     // last source time <= t
        if (j == n_src - 1 || ts(j) == t) return co[j];
        const double slope = (co[j + 1] - co[j]) / (ts(j + 1) - ts(j));
        return slope * (t - ts(j)) + co[j];
    }
    FOT_HD double tail_velocity() const
    {
        if (n_src < 2) return 0.0;
        const int lookback = n_src < 3 ? n_src : 3;
        const double v = (co[n_src - 1] - co[n_src - lookback]) / ((double)(lookback - 1) * sgan_dt);
        return fmax(fmin(v, 2.5), -2.5);                              // MAX_WALKING_SPEED
    }
};

}  // namespace fot
