/*
 * fot_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see fot_oracle.h).
 *
 * float64 restatement of the reference's FrenetPlanner.plan() and the helpers it
 * calls.  Every function cites the reference lines it follows (paths relative
 * to the reference repository root).  Arithmetic is written in the reference's
 * operand order so that results agree with NumPy to rounding noise; where
 * NumPy calls LAPACK (inv/solve) or its own SIMD libm the match is ~1e-13, not
 * bitwise.  Build with -ffp-contract=off (see Makefile).
 */
#include "fot_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* spline: src/planning/cubic_spline.py                                       */
/* ------------------------------------------------------------------------- */

struct orc_spline {
    int n;
    double *s;                  /* knots [n] */
    double *ax, *bx, *cx, *dx;  /* a[n], b[n-1], c[n], d[n-1] */
    double *ay, *by, *cy, *dy;
};

static double *dalloc(int n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

void orc_spline_free(orc_spline *sp)
{
    if (!sp) return;
    free(sp->s);
    free(sp->ax); free(sp->bx); free(sp->cx); free(sp->dx);
    free(sp->ay); free(sp->by); free(sp->cy); free(sp->dy);
    free(sp);
}

static orc_spline *spline_alloc(int n)
{
    orc_spline *sp = (orc_spline *)calloc(1, sizeof(orc_spline));
    sp->n = n;
    sp->s = dalloc(n);
    sp->ax = dalloc(n); sp->bx = dalloc(n); sp->cx = dalloc(n); sp->dx = dalloc(n);
    sp->ay = dalloc(n); sp->by = dalloc(n); sp->cy = dalloc(n); sp->dy = dalloc(n);
    return sp;
}

/* dense LU with partial pivoting = what np.linalg.solve (LAPACK gesv) does;
 * cubic_spline.py:41 */
static void dense_solve(int n, double *A, double *b)
{
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(A[i * n + k]);
            if (v > best) { best = v; piv = i; }
        }
        if (piv != k) {
            for (int j = 0; j < n; ++j) {
                double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t;
            }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        double rp = 1.0 / A[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            double l = A[i * n + k] * rp;
            if (l == 0.0) continue;
            A[i * n + k] = l;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
            b[i] -= l * b[k];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double acc = b[i];
        for (int j = i + 1; j < n; ++j) acc -= A[i * n + j] * b[j];
        b[i] = acc / A[i * n + i];
    }
}

/* CubicSpline1D.__init__ cubic_spline.py:23-45, _calc_A :168-181, _calc_B :183-187 */
static void spline1d_fit(int n, const double *x, const double *y, double *a, double *b, double *c, double *d)
{
    double *h = dalloc(n);
    for (int i = 0; i < n - 1; ++i) h[i] = x[i + 1] - x[i];
    for (int i = 0; i < n; ++i) a[i] = y[i];

    double *A = dalloc(n * n);
    double *B = dalloc(n);
    A[0] = 1.0;
    for (int i = 0; i < n - 1; ++i) {
        if (i != n - 2) A[(i + 1) * n + (i + 1)] = 2.0 * (h[i] + h[i + 1]);
        A[(i + 1) * n + i] = h[i];
        A[i * n + (i + 1)] = h[i];
    }
    A[0 * n + 1] = 0.0;
    A[(n - 1) * n + (n - 2)] = 0.0;
    A[(n - 1) * n + (n - 1)] = 1.0;
    for (int i = 1; i < n - 1; ++i)
        B[i] = 3.0 * (a[i + 1] - a[i]) / h[i] - 3.0 * (a[i] - a[i - 1]) / h[i - 1];
    dense_solve(n, A, B);
    for (int i = 0; i < n; ++i) c[i] = B[i];
    for (int i = 0; i < n - 1; ++i) {
        d[i] = (c[i + 1] - c[i]) / (3.0 * h[i]);
        b[i] = (a[i + 1] - a[i]) / h[i] - h[i] * (2.0 * c[i] + c[i + 1]) / 3.0;
    }
    free(h); free(A); free(B);
}

/* CubicSpline2D.__init__/_calc_s cubic_spline.py:201-213 */
orc_spline *orc_spline_from_waypoints(int n, const double *wx, const double *wy)
{
    if (n < 2) return NULL;
    orc_spline *sp = spline_alloc(n);
    sp->s[0] = 0.0;
    double acc = 0.0;
    for (int i = 0; i < n - 1; ++i) {
        double ds = hypot(wx[i + 1] - wx[i], wy[i + 1] - wy[i]);
        acc += ds;                      /* np.cumsum */
        sp->s[i + 1] = acc;
    }
    for (int i = 0; i < n - 1; ++i)
        if (sp->s[i + 1] - sp->s[i] < 0) { orc_spline_free(sp); return NULL; }
    spline1d_fit(n, sp->s, wx, sp->ax, sp->bx, sp->cx, sp->dx);
    spline1d_fit(n, sp->s, wy, sp->ay, sp->by, sp->cy, sp->dy);
    return sp;
}

orc_spline *orc_spline_from_coeffs(int n, const double *s,
                                   const double *ax, const double *bx, const double *cx, const double *dx,
                                   const double *ay, const double *by, const double *cy, const double *dy)
{
    if (n < 2) return NULL;
    orc_spline *sp = spline_alloc(n);
    memcpy(sp->s, s, sizeof(double) * n);
    memcpy(sp->ax, ax, sizeof(double) * n);       memcpy(sp->ay, ay, sizeof(double) * n);
    memcpy(sp->bx, bx, sizeof(double) * (n - 1)); memcpy(sp->by, by, sizeof(double) * (n - 1));
    memcpy(sp->cx, cx, sizeof(double) * n);       memcpy(sp->cy, cy, sizeof(double) * n);
    memcpy(sp->dx, dx, sizeof(double) * (n - 1)); memcpy(sp->dy, dy, sizeof(double) * (n - 1));
    return sp;
}

int orc_spline_n(const orc_spline *sp) { return sp->n; }

void orc_spline_coeffs(const orc_spline *sp, double *s, double *ax, double *bx, double *cx, double *dx,
                       double *ay, double *by, double *cy, double *dy)
{
    int n = sp->n;
    memcpy(s, sp->s, sizeof(double) * n);
    memcpy(ax, sp->ax, sizeof(double) * n);       memcpy(ay, sp->ay, sizeof(double) * n);
    memcpy(bx, sp->bx, sizeof(double) * (n - 1)); memcpy(by, sp->by, sizeof(double) * (n - 1));
    memcpy(cx, sp->cx, sizeof(double) * n);       memcpy(cy, sp->cy, sizeof(double) * n);
    memcpy(dx, sp->dx, sizeof(double) * (n - 1)); memcpy(dy, sp->dy, sizeof(double) * (n - 1));
}

/* _search_index cubic_spline.py:154-166: searchsorted(side='right') - 1, clipped */
static int search_index(const double *x, int n, double v)
{
    int lo = 0, hi = n;                 /* first index with x[idx] > v */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (x[mid] <= v) lo = mid + 1; else hi = mid;
    }
    int idx = lo - 1;
    if (idx < 0) idx = 0;
    if (idx > n - 2) idx = n - 2;
    return idx;
}

typedef struct {
    double x, y, dx, dy, ddx, ddy, dddx, dddy;
    int valid;
} sp_point;

/* CubicSpline1D.calc_position/first/second/third cubic_spline.py:47-152 */
static void spline_point(const orc_spline *sp, double s, sp_point *o)
{
    int n = sp->n;
    if (!(s >= sp->s[0] && s <= sp->s[n - 1])) {
        o->valid = 0;
        o->x = o->y = o->dx = o->dy = o->ddx = o->ddy = o->dddx = o->dddy = NAN;
        return;
    }
    o->valid = 1;
    int i = search_index(sp->s, n, s);
    double h = s - sp->s[i];
    o->x = sp->ax[i] + sp->bx[i] * h + sp->cx[i] * (h * h) + sp->dx[i] * pow(h, 3.0);
    o->y = sp->ay[i] + sp->by[i] * h + sp->cy[i] * (h * h) + sp->dy[i] * pow(h, 3.0);
    o->dx = sp->bx[i] + 2.0 * sp->cx[i] * h + 3.0 * sp->dx[i] * (h * h);
    o->dy = sp->by[i] + 2.0 * sp->cy[i] * h + 3.0 * sp->dy[i] * (h * h);
    o->ddx = 2.0 * sp->cx[i] + 6.0 * sp->dx[i] * h;
    o->ddy = 2.0 * sp->cy[i] + 6.0 * sp->dy[i] * h;
    o->dddx = 6.0 * sp->dx[i];
    o->dddy = 6.0 * sp->dy[i];
}

/* calc_yaw :275-288, calc_curvature :228-247, calc_curvature_rate :249-273 */
static void spline_frame(const sp_point *p, double *yaw, double *kappa, double *dkappa)
{
    *yaw = atan2(p->dy, p->dx);
    *kappa = (p->ddy * p->dx - p->ddx * p->dy) / pow(p->dx * p->dx + p->dy * p->dy, 1.5);
    double a = p->dx * p->ddy - p->dy * p->ddx;
    double b = p->dx * p->dddy - p->dy * p->dddx;
    double c = p->dx * p->ddx + p->dy * p->ddy;
    double d = p->dx * p->dx + p->dy * p->dy;
    *dkappa = b / pow(d, 1.5) - 3.0 * a * c / pow(d, 2.5);
}

void orc_spline_eval(const orc_spline *sp, int n, const double *s, double *x, double *y,
                     double *yaw, double *kappa, double *dkappa)
{
    for (int i = 0; i < n; ++i) {
        sp_point p;
        spline_point(sp, s[i], &p);
        x[i] = p.x; y[i] = p.y;
        spline_frame(&p, &yaw[i], &kappa[i], &dkappa[i]);
    }
}

/* h**3.0 correctly rounded.  The reference raises a one-element float64 ARRAY to 3.0 (cubic_spline.py:70-71 behind
 * np.atleast_1d): NumPy's SIMD power loop, which on this container's CPU (AVX512) differs from the correctly rounded
 * cube in 5 % of the calls and from glibc's pow() in as many -- the reference's own last bit depends on the NumPy build.
 * The nearest-point search (below) decides on such last bits at a tie, so the oracle and the library both use the one
 * platform-independent value there: the correctly rounded one (exact products through fma() residuals). */
static double cube_cr(double h)
{
    double p = h * h, e = fma(h, h, -p);
    double t = p * h, e2 = fma(p, h, -t);
    return t + fma(e, h, e2);
}

/* calc_position for the nearest-point search: a + b h + c h**2.0 + d h**3.0, left to right (cubic_spline.py:70-71) */
static void spline_xy(const orc_spline *sp, double s, double *x, double *y)
{
    int n = sp->n;
    if (!(s >= sp->s[0] && s <= sp->s[n - 1])) { *x = NAN; *y = NAN; return; }
    int i = search_index(sp->s, n, s);
    double h = s - sp->s[i], h2 = h * h, h3 = cube_cr(h);
    *x = sp->ax[i] + sp->bx[i] * h + sp->cx[i] * h2 + sp->dx[i] * h3;
    *y = sp->ay[i] + sp->by[i] * h + sp->cy[i] * h2 + sp->dy[i] * h3;
}

/* ------------------------------------------------------------------------- */
/* nearest point: src/core/coordinate_converter.py:202-339                    */
/* ------------------------------------------------------------------------- */

/* np.linspace(start, stop, num)[i] (endpoint=True): start + i*step, last = stop */
static double linspace_at(double start, double stop, int num, int i)
{
    if (num == 1) return start;
    if (i == num - 1) return stop;
    double step = (stop - start) / (double)(num - 1);
    if (step == 0.0) return start + ((double)i / (double)(num - 1)) * (stop - start);
    return start + (double)i * step;
}

/* math.hypot as CPython >= 3.10 computes it: the correctly rounded hypotenuse.  glibc 2.35's hypot() differs from it in
 * the last place in about 0.6 % of the calls (tests/test_emu_logic.py measures both against exact arithmetic), and the
 * reference's nearest-point search DECIDES by comparing math.hypot values of probes micrometres apart
 * (coordinate_converter.py:230, 267-272): exact squares through fma() residuals, one correction step. */
static double py_hypot(double x, double y)
{
    x = fabs(x); y = fabs(y);
    if (isinf(x) || isinf(y)) return INFINITY;
    if (isnan(x) || isnan(y)) return NAN;
    if (x < y) { double t = x; x = y; y = t; }
    if (y == 0.0) return x;
    double back = 1.0;                                            /* far from 1: scaled by a power of two (exact) */
    if (x > 0x1p500) { x *= 0x1p-600; y *= 0x1p-600; back = 0x1p600; }
    else if (x < 0x1p-500) { x *= 0x1p600; y *= 0x1p600; back = 0x1p-600; }
    double xx = x * x, ex = fma(x, x, -xx);
    double yy = y * y, ey = fma(y, y, -yy);
    double s = xx + yy, bb = s - xx;
    double es = (xx - (s - bb)) + (yy - bb);
    double lo = es + (ex + ey);
    double h = sqrt(s);
    double r = fma(-h, h, s) + lo;
    return (h + r / (2.0 * h)) * back;
}

/* _global_search coordinate_converter.py:318-339 */
static double global_search(const orc_spline *sp, double x, double y)
{
    double L = sp->s[sp->n - 1];
    int num = (int)(L / 0.1);
    if (num < 100) num = 100;
    double best = INFINITY, best_s = 0.0;
    int have = 0;
    for (int i = 0; i < num; ++i) {
        double s = linspace_at(0.0, L, num, i);
        double px, py;
        spline_xy(sp, s, &px, &py);
        double dist = hypot(x - px, y - py);                         /* np.hypot (:333): the C library's */
        if (isnan(dist)) return s;                                       /* np.argmin: first NaN wins */
        if (!have || dist < best) { best = dist; best_s = s; have = 1; }
    }
    return best_s;
}

/* find_nearest_point_on_path coordinate_converter.py:202-308.  returns 0 ok, 1 raise */
static int nearest_point(const orc_spline *sp, double x, double y, int has_prev, double prev_s,
                         double *ref /*[6]*/, double *new_prev_s)
{
    double s_end = sp->s[sp->n - 1];
    double best_s = 0.0;
    if (has_prev) {
        double s_min = fmax(0.0, prev_s - 10.0);
        double s_max = fmin(s_end, prev_s + 10.0);
        double min_dist = INFINITY;
        for (int i = 0; i < 100; ++i) {
            double s = linspace_at(s_min, s_max, 100, i);
            double px, py;
            spline_xy(sp, s, &px, &py);
            double dist = py_hypot(x - px, y - py);
            if (dist < min_dist) { min_dist = dist; best_s = s; }
        }
        int at_lower = (fabs(best_s - s_min) < 1e-3) && (s_min > 0);
        int at_upper = (fabs(best_s - s_max) < 1e-3) && (s_max < s_end);
        if (at_lower || at_upper) best_s = global_search(sp, x, y);
    } else {
        best_s = global_search(sp, x, y);
    }

    double ds = 0.2;
    for (int it = 0; it < 20; ++it) {
        double s_left = fmax(0.0, best_s - ds);
        double s_right = fmin(s_end, best_s + ds);
        double pxl, pyl, pxr, pyr, px, py;
        spline_xy(sp, s_left, &pxl, &pyl);
        spline_xy(sp, s_right, &pxr, &pyr);
        double dist_left = py_hypot(x - pxl, y - pyl);
        double dist_right = py_hypot(x - pxr, y - pyr);
        spline_xy(sp, best_s, &px, &py);
        double dist_curr = py_hypot(x - px, y - py);
        if (dist_left < dist_curr && dist_left < dist_right) best_s = s_left;
        else if (dist_right < dist_curr && dist_right < dist_left) best_s = s_right;
        else ds *= 0.5;
    }
    *new_prev_s = best_s;

    double rs = best_s, rx, ry;
    spline_xy(sp, rs, &rx, &ry);
    if (isnan(rx) || isnan(ry)) {
        best_s = global_search(sp, x, y);
        rs = best_s;
        spline_xy(sp, rs, &rx, &ry);
        if (isnan(rx) || isnan(ry)) return 1;
    }
    sp_point p;
    spline_point(sp, rs, &p);
    double rtheta, rkappa, rdkappa;
    spline_frame(&p, &rtheta, &rkappa, &rdkappa);
    if (isnan(rtheta) || isnan(rkappa) || isnan(rdkappa)) return 1;
    ref[0] = rs; ref[1] = rx; ref[2] = ry; ref[3] = rtheta; ref[4] = rkappa; ref[5] = rdkappa;
    return 0;
}

/* cartesian_to_frenet coordinate_converter.py:26-88 + frenet_planner.py:334-374 */
int orc_cartesian_to_frenet_state(const orc_spline *sp, const orc_ego *ego,
                                  double *fr, double *ref, double *new_prev_s)
{
    *new_prev_s = ego->has_prev_s ? ego->prev_s : NAN;
    if (nearest_point(sp, ego->x, ego->y, ego->has_prev_s, ego->prev_s, ref, new_prev_s)) return 1;
    double rs = ref[0], rx = ref[1], ry = ref[2], rtheta = ref[3], rkappa = ref[4], rdkappa = ref[5];
    double dx = ego->x - rx, dy = ego->y - ry;
    double cos_r = cos(rtheta), sin_r = sin(rtheta);
    double cross = cos_r * dy - sin_r * dx;
    double d = copysign(hypot(dx, dy), cross);
    double delta = ego->yaw - rtheta;
    double tan_d = tan(delta), cos_d = cos(delta);
    double omkd = 1 - rkappa * d;
    double d_p = omkd * tan_d;
    double krdp = rdkappa * d + rkappa * d_p;
    double kappa = ego->last_kappa;
    double d_pp = (-krdp * tan_d + omkd / (cos_d * cos_d) * (kappa * omkd / cos_d - rkappa));
    double s_d = ego->v * cos_d / omkd;
    double dtp = omkd / cos_d * kappa - rkappa;
    double s_dd = (ego->a * cos_d - s_d * s_d * (d_p * dtp - krdp)) / omkd;
    fr[0] = rs; fr[1] = s_d; fr[2] = s_dd;
    fr[3] = d;
    fr[4] = d_p * s_d;                              /* frenet_planner.py:368 */
    fr[5] = d_pp * (s_d * s_d) + d_p * s_dd;        /* :369 */
    return 0;
}

/* ------------------------------------------------------------------------- */
/* lattice: frenet_planner.py:376-503, 586-734                                */
/* ------------------------------------------------------------------------- */

typedef struct {
    int n_t;                    /* samples */
    double T;
    double qa[2][2];            /* quartic_A_inv */
    double qi[3][3];            /* quintic_A_inv */
} time_cache;

/* _build_time_cache :586-617 (inverses in closed form instead of LAPACK) */
static void build_time_cache(double T, double dt, time_cache *tc)
{
    tc->T = T;
    tc->n_t = (int)nearbyint(T / dt) + 1;
    double a = 3.0 * pow(T, 2), b = 4.0 * pow(T, 3), c = 6.0 * T, d = 12.0 * pow(T, 2);
    double det = a * d - b * c;
    tc->qa[0][0] = d / det;  tc->qa[0][1] = -b / det;
    tc->qa[1][0] = -c / det; tc->qa[1][1] = a / det;
    double m[3][3] = {
        { pow(T, 3), pow(T, 4), pow(T, 5) },
        { 3.0 * pow(T, 2), 4.0 * pow(T, 3), 5.0 * pow(T, 4) },
        { 6.0 * T, 12.0 * pow(T, 2), 20.0 * pow(T, 3) } };
    double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    double c01 = m[1][0] * m[2][2] - m[1][2] * m[2][0];
    double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    double det3 = m[0][0] * c00 - m[0][1] * c01 + m[0][2] * c02;
    tc->qi[0][0] = c00 / det3;
    tc->qi[0][1] = -(m[0][1] * m[2][2] - m[0][2] * m[2][1]) / det3;
    tc->qi[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det3;
    tc->qi[1][0] = -c01 / det3;
    tc->qi[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det3;
    tc->qi[1][2] = -(m[0][0] * m[1][2] - m[0][2] * m[1][0]) / det3;
    tc->qi[2][0] = c02 / det3;
    tc->qi[2][1] = -(m[0][0] * m[2][1] - m[0][1] * m[2][0]) / det3;
    tc->qi[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det3;
}

typedef struct { double a0, a1, a2, a3, a4; } quartic;
typedef struct { double a0, a1, a2, a3, a4, a5; } quintic;

/* _build_longitudinal_profiles :619-658 */
static void lon_coeffs(const double *fr, double tv, const time_cache *tc, quartic *q)
{
    q->a0 = fr[0]; q->a1 = fr[1]; q->a2 = fr[2] / 2.0;
    double b0 = tv - q->a1 - 2.0 * q->a2 * tc->T;
    double b1 = -2.0 * q->a2;
    q->a3 = b0 * tc->qa[0][0] + b1 * tc->qa[0][1];
    q->a4 = b0 * tc->qa[1][0] + b1 * tc->qa[1][1];
}

/* _build_lateral_profiles :660-701 */
static void lat_coeffs(const double *fr, double di, const time_cache *tc, quintic *q)
{
    double T = tc->T;
    q->a0 = fr[3]; q->a1 = fr[4]; q->a2 = fr[5] / 2.0;
    double b0 = di - q->a0 - q->a1 * T - q->a2 * T * T;
    double b1 = -q->a1 - 2.0 * q->a2 * T;
    double b2 = -2.0 * q->a2;
    q->a3 = b0 * tc->qi[0][0] + b1 * tc->qi[0][1] + b2 * tc->qi[0][2];
    q->a4 = b0 * tc->qi[1][0] + b1 * tc->qi[1][1] + b2 * tc->qi[1][2];
    q->a5 = b0 * tc->qi[2][0] + b1 * tc->qi[2][1] + b2 * tc->qi[2][2];
}

typedef struct {
    int n_t;                    /* untruncated samples (after brake padding) */
    double t[ORC_MAX_NT];
    double s[ORC_MAX_NT], s_d[ORC_MAX_NT], s_dd[ORC_MAX_NT], s_ddd[ORC_MAX_NT];
    double d[ORC_MAX_NT], d_d[ORC_MAX_NT], d_dd[ORC_MAX_NT], d_ddd[ORC_MAX_NT];
    double x[ORC_MAX_NT], y[ORC_MAX_NT], yaw[ORC_MAX_NT], v[ORC_MAX_NT], a[ORC_MAX_NT], c[ORC_MAX_NT];
    double cost;
    int keep;
} cand_path;

static void eval_lon(const quartic *q, int n, double dt, cand_path *cp)
{
    for (int k = 0; k < n; ++k) {
        double t = (double)k * dt, t2 = t * t, t3 = t2 * t, t4 = t2 * t2;
        cp->t[k] = t;
        cp->s[k] = q->a0 + q->a1 * t + q->a2 * t2 + q->a3 * t3 + q->a4 * t4;
        cp->s_d[k] = q->a1 + 2.0 * q->a2 * t + 3.0 * q->a3 * t2 + 4.0 * q->a4 * t3;
        cp->s_dd[k] = 2.0 * q->a2 + 6.0 * q->a3 * t + 12.0 * q->a4 * t2;
        cp->s_ddd[k] = 6.0 * q->a3 + 24.0 * q->a4 * t;
    }
}

static void eval_lat(const quintic *q, int n, double dt, cand_path *cp)
{
    for (int k = 0; k < n; ++k) {
        double t = (double)k * dt, t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
        cp->d[k] = q->a0 + q->a1 * t + q->a2 * t2 + q->a3 * t3 + q->a4 * t4 + q->a5 * t5;
        cp->d_d[k] = q->a1 + 2.0 * q->a2 * t + 3.0 * q->a3 * t2 + 4.0 * q->a4 * t3 + 5.0 * q->a5 * t4;
        cp->d_dd[k] = 2.0 * q->a2 + 6.0 * q->a3 * t + 12.0 * q->a4 * t2 + 20.0 * q->a5 * t3;
        cp->d_ddd[k] = 6.0 * q->a3 + 24.0 * q->a4 * t + 60.0 * q->a5 * t2;
    }
}

/* np.sum(np.square(a)): NumPy's pairwise summation -- up to 128 elements 8 strided partial sums, combined pairwise,
 * remainder added serially; longer arrays are halved (the first half rounded down to a multiple of 8) and the two
 * halves' sums added */
static double np_sum_sq(const double *a, int n)
{
    if (n > 128) {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum_sq(a, n2) + np_sum_sq(a + n2, n - n2);
    }
    if (n < 8) {
        double r = 0.0;   /* NumPy starts from -0.0 semantics irrelevant here */
        for (int i = 0; i < n; ++i) r += a[i] * a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j] * a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j] * a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i] * a[i];
    return res;
}

/* _calculate_cost :703-734 */
static double path_cost(const orc_params *p, const cand_path *cp, double target_speed)
{
    int n = cp->n_t;
    double Jp = np_sum_sq(cp->d_ddd, n);
    double Jd = cp->d[n - 1] * cp->d[n - 1];
    double Js = np_sum_sq(cp->s_ddd, n);
    double dv = target_speed - cp->s_d[n - 1];
    double Jv = dv * dv;
    double Jt = cp->t[n - 1];
    double lat = p->k_j * Jp + p->k_t * Jt + p->k_d * Jd;
    double lon = p->k_j * Js + p->k_t * Jt + p->k_s_dot * Jv;
    return p->k_lat * lat + p->k_lon * lon;
}

/* lattice dimensions, frenet_planner.py:397-420, 469-475 */
typedef struct {
    int n_ti;                   /* number of horizons = n_ti_int + 1 */
    int n_tv, n_di, n_side;
    int n_brake_max;            /* ladder length before the s_d gate / n_pad skip */
    int n_total;                /* round(max_t/dt)+1 */
    double tv[256];
} lattice_dims;

static int lattice_setup(const orc_params *p, double target_speed, lattice_dims *L)
{
    L->n_ti = (int)((p->max_t - p->min_t) / p->dt + 1e-9) + 1;
    if (L->n_ti < 0) L->n_ti = 0;
    int n_down = (int)(target_speed / p->d_t_s + 1e-9);
    if (n_down + 1 <= 0 || n_down + 2 > 256) return -1;
    int n = 0;
    for (int k = 0; k <= n_down; ++k) L->tv[n++] = target_speed - (double)k * p->d_t_s;
    if (L->tv[n - 1] > 1e-9) L->tv[n++] = 0.0;
    L->n_tv = n;
    L->n_side = (int)(p->max_road_width / p->d_road_w + 1e-9);
    L->n_di = 2 * L->n_side + 1;
    if (L->n_side < 0) L->n_di = 0;
    /* np.arange(BRAKE_T_MIN, min_t - 1e-9, BRAKE_T_STEP) */
    double span = (p->min_t - 1e-9 - 0.5) / 0.5;
    int nb = (int)ceil(span);
    if (nb < 0) nb = 0;
    L->n_brake_max = nb;
    L->n_total = (int)nearbyint(p->max_t / p->dt) + 1;
    return 0;
}

static double horizon_T(const orc_params *p, int ti) { return p->min_t + (double)ti * p->dt; }

int orc_max_candidates(const orc_params *p, double target_speed)
{
    lattice_dims L;
    if (lattice_setup(p, target_speed, &L)) return -1;
    return L.n_ti * L.n_tv * L.n_di + L.n_brake_max;
}

/* Fill the Frenet arrays + cost of candidate `index` (generation order
 * Ti -> tv -> di, brake ladder last; frenet_planner.py:398-449).  Returns 0, or
 * -1 when index is past the last candidate. */
static int build_candidate(const orc_params *p, const double *fr, double target_speed,
                           const lattice_dims *L, int index, cand_path *cp)
{
    int n_grid = L->n_ti * L->n_tv * L->n_di;
    time_cache tc;
    if (index < n_grid) {
        int ti = index / (L->n_tv * L->n_di);
        int rem = index % (L->n_tv * L->n_di);
        int itv = rem / L->n_di, idi = rem % L->n_di;
        build_time_cache(horizon_T(p, ti), p->dt, &tc);
        if (tc.n_t > ORC_MAX_NT) return -2;
        quartic ql; quintic qd;
        lon_coeffs(fr, L->tv[itv], &tc, &ql);
        lat_coeffs(fr, (double)(idi - L->n_side) * p->d_road_w, &tc, &qd);
        cp->n_t = tc.n_t;
        eval_lon(&ql, tc.n_t, p->dt, cp);
        eval_lat(&qd, tc.n_t, p->dt, cp);
        cp->cost = path_cost(p, cp, target_speed);
        return 0;
    }
    /* brake ladder :453-503 */
    if (!(fr[1] > 0.1)) return -1;
    int b = index - n_grid, seen = 0;
    for (int j = 0; j < L->n_brake_max; ++j) {
        double Tb = 0.5 + (double)j * 0.5;
        build_time_cache(Tb, p->dt, &tc);
        int n_pad = L->n_total - tc.n_t;
        if (n_pad < 0) continue;
        if (seen++ != b) continue;
        if (L->n_total > ORC_MAX_NT) return -2;
        quartic ql; quintic qd;
        lon_coeffs(fr, 0.0, &tc, &ql);
        lat_coeffs(fr, fr[3], &tc, &qd);
        eval_lon(&ql, tc.n_t, p->dt, cp);
        eval_lat(&qd, tc.n_t, p->dt, cp);
        for (int k = tc.n_t; k < L->n_total; ++k) {
            cp->t[k] = (double)k * p->dt;
            cp->s[k] = cp->s[tc.n_t - 1]; cp->s_d[k] = 0.0; cp->s_dd[k] = 0.0; cp->s_ddd[k] = 0.0;
            cp->d[k] = cp->d[tc.n_t - 1]; cp->d_d[k] = 0.0; cp->d_dd[k] = 0.0; cp->d_ddd[k] = 0.0;
        }
        cp->n_t = L->n_total;
        cp->cost = path_cost(p, cp, target_speed);
        return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------------- */
/* Frenet -> Cartesian: frenet_planner.py:736-889, coordinate_converter.py:91-182 */
/* ------------------------------------------------------------------------- */

static void global_path(const orc_spline *sp, cand_path *cp)
{
    int n = cp->n_t, singular = 0, first_nan = -1;
    for (int k = 0; k < n; ++k) {
        sp_point pt;
        spline_point(sp, cp->s[k], &pt);
        double rtheta, rkappa, rdkappa;
        spline_frame(&pt, &rtheta, &rkappa, &rdkappa);
        double s_d = cp->s_d[k], s_dd = cp->s_dd[k], d = cp->d[k];
        int moving = fabs(s_d) > 1e-3;                               /* :792 EPS_S_DOT */
        double safe = moving ? s_d : 1.0;
        double dp = moving ? cp->d_d[k] / safe : 0.0;
        double dpp = moving ? (cp->d_dd[k] - dp * s_dd) / (safe * safe) : 0.0;

        double cos_r = cos(rtheta), sin_r = sin(rtheta);
        double x = pt.x - sin_r * d;
        double y = pt.y + cos_r * d;
        double omkd = 1 - rkappa * d;
        double tan_d = dp / omkd;
        double delta = atan2(dp, omkd);
        double cos_d = cos(delta);
        double th = delta + rtheta;
        double theta = atan2(sin(th), cos(th));                      /* np.angle(np.exp(1j*th)) */
        double krdp = rdkappa * d + rkappa * dp;
        double kappa = (((dpp + krdp * tan_d) * cos_d * cos_d) / omkd + rkappa) * cos_d / omkd;
        double d_dot = dp * s_d;
        double v = sqrt(omkd * omkd * s_d * s_d + d_dot * d_dot);
        double dtp = omkd / cos_d * kappa - rkappa;
        double a = s_dd * omkd / cos_d + s_d * s_d / cos_d * (dp * dtp - krdp);
        cp->x[k] = x; cp->y[k] = y; cp->yaw[k] = theta; cp->c[k] = kappa; cp->v[k] = v; cp->a[k] = a;
        if (isfinite(omkd) && omkd <= 0.05) singular = 1;            /* :826-827 */
        if (first_nan < 0 && isnan(x)) first_nan = k;
    }
    if (singular) { cp->keep = 0; return; }                          /* :828-833 -> x[0]=NaN -> keep 0 */
    if (first_nan >= 0) cp->keep = first_nan >= 2 ? first_nan : 0;   /* :866 */
    else cp->keep = n;
}

/* ------------------------------------------------------------------------- */
/* collision: frenet_planner.py:1035-1233                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    int n;                       /* expanded points */
    double px[ORC_MAX_NT * ORC_MAX_CIRCLES], py[ORC_MAX_NT * ORC_MAX_CIRCLES];
    int tidx[ORC_MAX_NT * ORC_MAX_CIRCLES];   /* round(t/dt) before clipping */
    double minx, miny, maxx, maxy;
    double sq_r, sq_r_dyn;
} coll_geom;

/* _path_collision_geometry :1126-1179 */
static int collision_geometry(const orc_params *p, int n, const double *x, const double *y,
                              const double *yaw, const double *t, double inflation, coll_geom *g)
{
    if (n <= 0) return 0;
    double ego_r;
    int m = 0;
    if (p->n_circles <= 0) {
        ego_r = p->robot_radius;
        for (int k = 0; k < n; ++k) {
            g->px[m] = x[k]; g->py[m] = y[k];
            g->tidx[m] = (int)nearbyint(t[k] / p->dt);
            ++m;
        }
    } else {
        ego_r = p->footprint_radius;
        for (int c = 0; c < p->n_circles; ++c)
            for (int k = 0; k < n; ++k) {
                g->px[m] = x[k] + p->footprint_offsets[c] * cos(yaw[k]);
                g->py[m] = y[k] + p->footprint_offsets[c] * sin(yaw[k]);
                g->tidx[m] = (int)nearbyint(t[k] / p->dt);
                ++m;
            }
    }
    g->n = m;
    double r = fmax(ego_r + p->obstacle_radius, 1e-6);
    double r_dyn = r * inflation;
    g->sq_r = r * r;
    g->sq_r_dyn = r_dyn * r_dyn;
    double aabb_r = fmax(r, r_dyn);
    double mnx = g->px[0], mxx = g->px[0], mny = g->py[0], mxy = g->py[0];
    for (int i = 1; i < m; ++i) {
        /* np.min/np.max propagate NaN; internal paths are finite here */
        if (g->px[i] < mnx) mnx = g->px[i];
        if (g->px[i] > mxx) mxx = g->px[i];
        if (g->py[i] < mny) mny = g->py[i];
        if (g->py[i] > mxy) mxy = g->py[i];
    }
    g->minx = mnx - aabb_r; g->miny = mny - aabb_r;
    g->maxx = mxx + aabb_r; g->maxy = mxy + aabb_r;
    return 1;
}

/* _hits_static :1181-1198 */
static int hits_static(const coll_geom *g, const double *sxy, int ns, double sq)
{
    for (int j = 0; j < ns; ++j) {
        double ox = sxy[2 * j], oy = sxy[2 * j + 1];
        if (!(ox >= g->minx && ox <= g->maxx && oy >= g->miny && oy <= g->maxy)) continue;
        for (int i = 0; i < g->n; ++i) {
            double dx = g->px[i] - ox, dy = g->py[i] - oy;
            if (dx * dx + dy * dy <= sq) return 1;
        }
    }
    return 0;
}

/* _hits_dynamic :1200-1233; dyn = [P][T][2] */
static int hits_dynamic(const coll_geom *g, const double *dyn, int P, int T, double sq)
{
    if (!dyn || P <= 0 || T <= 0) return 0;
    for (int o = 0; o < P; ++o) {
        const double *tr = dyn + (size_t)o * T * 2;
        double mnx = tr[0], mxx = tr[0], mny = tr[1], mxy = tr[1];
        for (int k = 1; k < T; ++k) {
            if (tr[2 * k] < mnx) mnx = tr[2 * k];
            if (tr[2 * k] > mxx) mxx = tr[2 * k];
            if (tr[2 * k + 1] < mny) mny = tr[2 * k + 1];
            if (tr[2 * k + 1] > mxy) mxy = tr[2 * k + 1];
            /* np.min / np.max propagate a NaN sample (:1211-1212): the track then fails the mask below */
            if (isnan(tr[2 * k])) mnx = mxx = NAN;
            if (isnan(tr[2 * k + 1])) mny = mxy = NAN;
        }
        if (!(mxx >= g->minx && mnx <= g->maxx && mxy >= g->miny && mny <= g->maxy)) continue;
        for (int i = 0; i < g->n; ++i) {
            int ti = g->tidx[i];
            if (ti < 0) ti = 0;
            if (ti > T - 1) ti = T - 1;
            double dx = g->px[i] - tr[2 * ti], dy = g->py[i] - tr[2 * ti + 1];
            if (dx * dx + dy * dy <= sq) return 1;
        }
    }
    return 0;
}

/* _path_is_collision_free :1035-1047, _check_collision :1049-1074,
 * _check_collision_distribution :1076-1124 */
int orc_path_collision_free(const orc_params *p, int n, const double *x, const double *y,
                            const double *yaw, const double *t, const orc_obstacles *obs)
{
    static const orc_obstacles none = { 0 };
    if (!obs) obs = &none;
    coll_geom g;
    int dist_mode = (obs->dyn_mode == 2 && obs->dyn && (size_t)obs->S * obs->P * obs->T > 0);
    if (!collision_geometry(p, n, x, y, yaw, t, dist_mode ? 1.0 : p->collision_margin_inflation, &g))
        return 1;
    if (obs->n_static > 0 && obs->static_xy && hits_static(&g, obs->static_xy, obs->n_static, g.sq_r))
        return 0;
    if (dist_mode) {
        int max_viol = (int)floor(p->chance_epsilon * (double)obs->S);
        int viol = 0;
        for (int k = 0; k < obs->S; ++k) {
            if (hits_dynamic(&g, obs->dyn + (size_t)k * obs->P * obs->T * 2, obs->P, obs->T, g.sq_r)) {
                if (++viol > max_viol) return 0;
            }
        }
        return 1;
    }
    if (obs->dyn_mode == 1 && obs->dyn && hits_dynamic(&g, obs->dyn, obs->P, obs->T, g.sq_r_dyn)) return 0;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* checks: frenet_planner.py:891-1033                                         */
/* ------------------------------------------------------------------------- */

/* _curvature_feasible :995-1033 */
static int curvature_feasible(const cand_path *cp, int n, double c_max_curv)
{
    for (int i = 1; i < n; ++i) {
        if (cp->v[i] > 0.5) {
            if (fabs(cp->c[i]) > c_max_curv) return 0;
        } else {
            double dd = fabs(cp->d[i] - cp->d[i - 1]);
            double d_s = fabs(cp->s[i] - cp->s[i - 1]);
            if (dd > fmax(1.5 * d_s, 0.02)) return 0;
            double dy_ = cp->yaw[i] - cp->yaw[i - 1];
            double dyaw = fabs(atan2(sin(dy_), cos(dy_)));
            double ds = hypot(cp->x[i] - cp->x[i - 1], cp->y[i] - cp->y[i - 1]);
            if (dyaw > fmax(c_max_curv * ds, 0.1)) return 0;
        }
    }
    return 1;
}

typedef struct { double max_speed, max_accel, max_curv, max_lat; } limits;

static void resolve_limits(const orc_params *p, const orc_overrides *ov, limits *l)
{
    l->max_speed = p->max_speed; l->max_accel = p->max_accel;
    l->max_curv = p->max_curvature; l->max_lat = p->max_lat_accel;
    if (ov) {
        if (!isnan(ov->max_speed)) l->max_speed = ov->max_speed;
        if (!isnan(ov->max_accel)) l->max_accel = ov->max_accel;
        if (!isnan(ov->max_curvature)) l->max_curv = ov->max_curvature;
        if (!isnan(ov->max_lat_accel)) l->max_lat = ov->max_lat_accel;
    }
}

/* _check_paths :932-991 for one candidate */
static int check_candidate(const orc_params *p, const limits *l, const cand_path *cp, const orc_obstacles *obs)
{
    int n = cp->keep;
    if (n == 0) return ORC_ST_DROPPED;
    for (int k = 0; k < n; ++k)
        if (!(isfinite(cp->v[k]) && isfinite(cp->a[k]) && isfinite(cp->c[k]))) return ORC_ST_DROPPED;
    if (n >= 2) {
        double mx = -INFINITY;
        int nan_seen = 0;
        for (int k = 1; k < n; ++k) {
            double st = hypot(cp->x[k] - cp->x[k - 1], cp->y[k] - cp->y[k - 1]);
            if (isnan(st)) nan_seen = 1;
            if (st > mx) mx = st;
        }
        if (!nan_seen && mx > fmax(l->max_speed, p->max_speed) * p->dt * 3.0) return ORC_ST_DROPPED;
    }
    for (int k = 1; k < n; ++k) if (cp->v[k] > l->max_speed) return ORC_ST_SPEED;
    for (int k = 1; k < n; ++k) if (fabs(cp->a[k]) > l->max_accel) return ORC_ST_ACCEL;
    if (!curvature_feasible(cp, n, l->max_curv)) return ORC_ST_CURV;
    for (int k = 1; k < n; ++k) if (cp->v[k] * cp->v[k] * fabs(cp->c[k]) > l->max_lat) return ORC_ST_LATACC;
    for (int k = 1; k < n; ++k) if (fabs(cp->d[k]) > p->max_road_width + 1e-9) return ORC_ST_ROAD;
    if (!orc_path_collision_free(p, n, cp->x, cp->y, cp->yaw, cp->t, obs)) return ORC_ST_COLLISION;
    return ORC_ST_OK;
}

/* ------------------------------------------------------------------------- */
/* plan(): frenet_planner.py:227-324, 1235-1259                               */
/* ------------------------------------------------------------------------- */

static void copy_path(const cand_path *cp, orc_result *out)
{
    int n = cp->keep;
    out->n_keep = n;
    size_t b = sizeof(double) * (size_t)n;
    memcpy(out->t, cp->t, b); memcpy(out->s, cp->s, b); memcpy(out->s_d, cp->s_d, b);
    memcpy(out->s_dd, cp->s_dd, b); memcpy(out->s_ddd, cp->s_ddd, b);
    memcpy(out->d, cp->d, b); memcpy(out->d_d, cp->d_d, b); memcpy(out->d_dd, cp->d_dd, b);
    memcpy(out->d_ddd, cp->d_ddd, b);
    memcpy(out->x, cp->x, b); memcpy(out->y, cp->y, b); memcpy(out->yaw, cp->yaw, b);
    memcpy(out->v, cp->v, b); memcpy(out->a, cp->a, b); memcpy(out->c, cp->c, b);
}

int orc_plan(const orc_params *p, const orc_spline *sp, const orc_ego *ego,
             double target_speed, const orc_overrides *ov, double max_stop_distance,
             const orc_obstacles *obs, orc_result *out, orc_cand_table *table)
{
    memset(out, 0, sizeof(*out));
    out->best_index = -1;
    out->cost = INFINITY;
    out->new_last_kappa = ego->last_kappa;
    out->new_prev_s = ego->has_prev_s ? ego->prev_s : NAN;
    if (orc_cartesian_to_frenet_state(sp, ego, out->frenet0, out->ref0, &out->new_prev_s)) {
        out->status = ORC_PLAN_C2F_FAILED;
        return 0;
    }
    lattice_dims L;
    int rc = lattice_setup(p, target_speed, &L);
    if (rc) return rc;
    limits lim;
    resolve_limits(p, ov, &lim);

    cand_path *cp = (cand_path *)malloc(sizeof(cand_path));
    cand_path *best = (cand_path *)malloc(sizeof(cand_path));
    double min_cost = INFINITY;
    int n_cand = 0;
    out->stats_valid = 1;
    for (int idx = 0;; ++idx) {
        int r = build_candidate(p, out->frenet0, target_speed, &L, idx, cp);
        if (r == -2) { free(cp); free(best); return -2; }
        if (r) break;
        ++n_cand;
        global_path(sp, cp);
        int st = check_candidate(p, &lim, cp, obs);
        if (st == ORC_ST_OK && !isnan(max_stop_distance)) {          /* :307-324 */
            int n = cp->keep;
            int stops = fabs(cp->v[n - 1]) <= 0.15;
            double travel = cp->s[n - 1] - cp->s[0];
            if (!(stops && travel <= max_stop_distance + 1e-6)) st = ORC_ST_STOPDIST;
        }
        if (st < 8) out->stats[st]++;
        if (table) {
            if (table->cost) table->cost[idx] = cp->cost;
            if (table->status) table->status[idx] = st;
            if (table->keep) table->keep[idx] = cp->keep;
            if (table->n_t) table->n_t[idx] = cp->n_t;
        }
        if (st == ORC_ST_OK && cp->cost < min_cost) {                /* :1254-1257 strict < */
            min_cost = cp->cost;
            out->best_index = idx;
            cand_path *tmp = best; best = cp; cp = tmp;
        }
    }
    out->n_cand = n_cand;
    if (out->best_index >= 0) {
        out->status = ORC_PLAN_OK;
        out->cost = min_cost;
        copy_path(best, out);
        if (best->keep > 1) out->new_last_kappa = best->c[1];        /* :301-302 */
    } else {
        out->status = ORC_PLAN_NO_PATH;
    }
    free(cp); free(best);
    return 0;
}

int orc_candidate_path(const orc_params *p, const orc_spline *sp, const double *frenet0,
                       double target_speed, int index, double *arrays, double *cost)
{
    lattice_dims L;
    if (lattice_setup(p, target_speed, &L)) return -1;
    cand_path *cp = (cand_path *)calloc(1, sizeof(cand_path));
    if (build_candidate(p, frenet0, target_speed, &L, index, cp)) { free(cp); return -1; }
    global_path(sp, cp);
    const double *src[15] = { cp->t, cp->s, cp->s_d, cp->s_dd, cp->s_ddd, cp->d, cp->d_d, cp->d_dd,
                              cp->d_ddd, cp->x, cp->y, cp->yaw, cp->v, cp->a, cp->c };
    /* arrays hold the UNTRUNCATED n_t samples; the return value is keep */
    for (int f = 0; f < 15; ++f) memcpy(arrays + f * ORC_MAX_NT, src[f], sizeof(double) * ORC_MAX_NT);
    if (cost) *cost = cp->cost;
    int keep = cp->keep;
    free(cp);
    return keep;
}

int orc_plan_batch(const orc_params *p, const orc_spline *sp, int n_inst, const orc_ego *ego,
                   const double *target_speed, const orc_overrides *ov, const double *max_stop,
                   const orc_obstacles *obs, orc_result *out)
{
    for (int i = 0; i < n_inst; ++i) {
        int rc = orc_plan(p, sp, &ego[i], target_speed[i], ov ? &ov[i] : NULL,
                          max_stop ? max_stop[i] : NAN, obs ? &obs[i] : NULL, &out[i], NULL);
        if (rc) return rc;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* prediction resampling, src/prediction/trajectory_predictor.py */
/* ------------------------------------------------------------------------- */

/* len(np.arange(start, stop, step)) = ceil((stop - start) / step) */
int orc_n_dense(double sgan_dt, double sim_dt, double plan_horizon, int pred_len)
{
    double target = fmax(plan_horizon, (double)pred_len * sgan_dt);
    double n = ceil((target + 1e-9 - sim_dt) / sim_dt);
    return n > 0 ? (int)n : 0;
}

/* np.allclose(a, b): all |a_i - b| <= 1e-8 + 1e-5 |b| */
static int all_close(const double *a, int n, double b)
{
    for (int i = 0; i < n; ++i)
        if (!(fabs(a[i] - b) <= 1e-8 + 1e-5 * fabs(b))) return 0;
    return 1;
}

/* np.interp(x, xp, fp) for one x (xp increasing) */
static double np_interp(double x, const double *xp, const double *fp, int n)
{
    if (isnan(x)) return x;
    if (x > xp[n - 1]) return fp[n - 1];
    if (x < xp[0]) return fp[0];
    int lo = 0, hi = n;                     /* j with xp[j] <= x < xp[j+1] */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid + 1; else hi = mid; }
    int j = lo - 1;
    if (j == n - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double r = slope * (x - xp[j]) + fp[j];
    if (isnan(r)) {
        r = slope * (x - xp[j + 1]) + fp[j + 1];
        if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}

int orc_process_prediction(double sgan_dt, double sim_dt, double plan_horizon, int pred_len, int P,
                           const double *pred, const double *anchor, double staleness, double *out)
{
    int n_dense = orc_n_dense(sgan_dt, sim_dt, plan_horizon, pred_len);
    int n_src = pred_len + (anchor ? 1 : 0);
    double *ts = dalloc(n_src), *co = dalloc(n_src);
    int o = 0;
    if (anchor) ts[o++] = -staleness;                                   /* :268-270 */
    for (int k = 1; k <= pred_len; ++k) ts[o++] = (double)k * sgan_dt - staleness;
    for (int p = 0; p < P; ++p)
        for (int ax = 0; ax < 2; ++ax) {
            o = 0;
            if (anchor) co[o++] = anchor[2 * p + ax];
            for (int k = 0; k < pred_len; ++k) co[o++] = pred[((size_t)k * P + p) * 2 + ax];
            double *dst = out + ((size_t)p * n_dense) * 2 + ax;
            if (all_close(co, n_src, co[0]) || all_close(co, n_src, 0.0)) {   /* :292-294 */
                for (int i = 0; i < n_dense; ++i) dst[2 * i] = co[n_src - 1];
                continue;
            }
            int lookback = n_src < 3 ? n_src : 3;
            double v_tail = 0.0;
            if (n_src >= 2) {
                v_tail = (co[n_src - 1] - co[n_src - lookback]) / ((double)(lookback - 1) * sgan_dt);
                v_tail = fmax(fmin(v_tail, 2.5), -2.5);                  /* MAX_WALKING_SPEED */
            }
            for (int i = 0; i < n_dense; ++i) {
                double t = sim_dt + (double)i * sim_dt;                   /* np.arange(sim_dt, ..., sim_dt)[i] */
                double v = np_interp(t, ts, co, n_src);
                if (n_src >= 2 && t > ts[n_src - 1]) v = co[n_src - 1] + v_tail * (t - ts[n_src - 1]);
                dst[2 * i] = v;
            }
        }
    free(ts); free(co);
    return n_dense;
}

int orc_predict_cv(double sgan_dt, double sim_dt, double plan_horizon, int pred_len, int P,
                   const double *obs_last, const double *obs_prev, double staleness, double *out)
{
    return orc_predict_cv_obs(sgan_dt, sim_dt, plan_horizon, pred_len, P, obs_last, obs_prev, 0, staleness, out);
}

/* obs_f32: the observations were float32 tensors (observer.py:134) -- positions rounded to float32 and the velocity
 * (p_curr - p_prev) / sgan_dt evaluated in float32 (NumPy: float32 array / Python float), :203-217 */
int orc_predict_cv_obs(double sgan_dt, double sim_dt, double plan_horizon, int pred_len, int P,
                       const double *obs_last, const double *obs_prev, int obs_f32, double staleness, double *out)
{
    int n_dense = orc_n_dense(sgan_dt, sim_dt, plan_horizon, pred_len);
    for (int p = 0; p < P; ++p)
        for (int ax = 0; ax < 2; ++ax) {
            double cur = obs_last[2 * p + ax];
            double vel = obs_prev ? (cur - obs_prev[2 * p + ax]) / sgan_dt : 0.0;
            if (obs_f32) {
                volatile float c32 = (float)cur, p32 = obs_prev ? (float)obs_prev[2 * p + ax] : 0.0f;
                volatile float d32 = c32 - p32;
                volatile float v32 = d32 / (float)sgan_dt;
                cur = (double)c32;
                vel = obs_prev ? (double)v32 : 0.0;
            }
            for (int i = 0; i < n_dense; ++i) {
                double t = (sim_dt + (double)i * sim_dt) + staleness;
                out[((size_t)p * n_dense + i) * 2 + ax] = cur + vel * t;
            }
        }
    return n_dense;
}

int orc_best_sample(int S, int P, int T, const double *samples, double *dist_out)
{
    size_t n = (size_t)P * T;
    int best = 0;
    double best_d = INFINITY;
    for (int s = 0; s < S; ++s) {
        double acc = 0.0;
        for (size_t i = 0; i < n; ++i) {
            double mx = 0.0, my = 0.0;
            for (int q = 0; q < S; ++q) { mx += samples[((size_t)q * n + i) * 2]; my += samples[((size_t)q * n + i) * 2 + 1]; }
            mx /= (double)S; my /= (double)S;
            double dx = samples[((size_t)s * n + i) * 2] - mx, dy = samples[((size_t)s * n + i) * 2 + 1] - my;
            acc += sqrt(dx * dx + dy * dy);
        }
        if (dist_out) dist_out[s] = acc;
        if (acc < best_d) { best_d = acc; best = s; }
    }
    return best;
}

/* ------------------------------------------------------------------------- */
/* safety metrics, src/core/data_structures.py:301-388            */
/* ------------------------------------------------------------------------- */

void orc_safety_metrics(const orc_params *p, double ego_radius, double ped_radius, const double *ego, int P,
                        const double *ped_pos, const double *ped_vel, double *out)
{
    double cx[ORC_MAX_CIRCLES], cy[ORC_MAX_CIRCLES];
    int nc = 1;
    double combined = ego_radius + ped_radius;
    if (p->n_circles > 0) {                                       /* footprint.circle_centers, footprint.py:42-45 */
        nc = p->n_circles;
        for (int c = 0; c < nc; ++c) {
            cx[c] = ego[0] + p->footprint_offsets[c] * cos(ego[2]);
            cy[c] = ego[1] + p->footprint_offsets[c] * sin(ego[2]);
        }
        combined = p->footprint_radius + ped_radius;
    } else {
        cx[0] = ego[0]; cy[0] = ego[1];
    }
    double min_d = INFINITY, ttc = INFINITY, ahead_min = INFINITY;
    const double evx = ego[3] * cos(ego[2]), evy = ego[3] * sin(ego[2]);
    const double hx = cos(ego[2]), hy = sin(ego[2]);
    for (int c = 0; c < nc; ++c)
        for (int i = 0; i < P; ++i) {
            const double rx = ped_pos[2 * i] - cx[c], ry = ped_pos[2 * i + 1] - cy[c];
            const double dist = sqrt(rx * rx + ry * ry);
            if (dist < min_d) min_d = dist;
            const double vx = ped_vel[2 * i] - evx, vy = ped_vel[2 * i + 1] - evy;
            const double along = -(rx * vx + ry * vy) / (sqrt(rx * rx + ry * ry) + 1e-8);
            if (along > 1e-5) {
                const double t = (dist - combined) / along;
                if (t >= 0 && t < ttc) ttc = t;
            }
            const double ex = ped_pos[2 * i] - ego[0], ey = ped_pos[2 * i + 1] - ego[1];
            if (ex * hx + ey * hy > 0.0 && dist < ahead_min) ahead_min = dist;
        }
    out[0] = min_d;
    out[1] = min_d < combined ? 1.0 : 0.0;
    out[2] = ttc;
    out[3] = min_d - combined;
    out[4] = isinf(ahead_min) ? INFINITY : ahead_min - combined;
}
