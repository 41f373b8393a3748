// fot_kernels.hip -- gfx950 kernels of the Frenet optimal-trajectory planner.
//
// Pipeline of one fot_plan_batch (one launch each over the whole batch; every decision in float64):
//   k_frenet_state : 4 waves / instance: pulls the instance's descriptor out of the pinned staging block, nearest point
//                    (sample scan over the workgroup, three refinement rounds per evaluation) + Cartesian->Frenet
//   k_cull         : 8 waves / (instance, 8 consecutive time steps): derives and merges the boxes of the instance's
//                    longitudinal profiles, sorts the obstacles of those time rows that lie inside the grown boxes
//                    into per-step entry lists (strips, LDS atomics), writes each candidate wave's chunk range
//   k_evaluate     : 1 wave / TILE (up to 64 consecutive candidates of one instance, at most three full-length
//                    profiles), 1 lane / candidate, tiles dealt out longest first: the wave builds the rows of its
//                    tile's profiles in its own slice of LDS, then: quintic, Frenet->Cartesian, cost,
//                    truncation, kinematic checks, and -- while the candidate can still pass -- the collision test of
//                    each sample against the entry list of its time step (wave-uniform chunk walk on scalar loads,
//                    float32 bounds, float64 only between them).  Candidate points never leave registers.
//   (selection)    : the wave that finishes an instance's last tile: stop-distance filter, rejection histogram,
//                    first-minimum argmin (lowest index wins ties), selected path rebuilt from its profile and written
//                    out -- inside the evaluation launch (select_instance_wave)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fot_math.hpp"
#include "fot_kernels.h"

namespace fot {

// ---------------------------------------------------------------------------
// wave-level reductions on the DPP data path (no LDS traffic)
// ---------------------------------------------------------------------------

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v)
{
    // lanes without a valid source (or masked rows) keep their own value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// butterfly inside quads / half rows / rows, then row_bcast15 and row_bcast31: lane 63 ends with the reduction
#define FOT_WAVE_REDUCE_F32(NAME, OP)                                                        \
    __device__ __forceinline__ float NAME(float v)                                           \
    {                                                                                        \
        v = OP(v, dpp_f32<0xB1, 0xf>(v));  /* quad_perm [1,0,3,2] */                         \
        v = OP(v, dpp_f32<0x4E, 0xf>(v));  /* quad_perm [2,3,0,1] */                         \
        v = OP(v, dpp_f32<0x141, 0xf>(v)); /* row_half_mirror */                             \
        v = OP(v, dpp_f32<0x140, 0xf>(v)); /* row_mirror */                                  \
        v = OP(v, dpp_f32<0x142, 0xa>(v)); /* row_bcast15 -> rows 1, 3 */                    \
        v = OP(v, dpp_f32<0x143, 0xc>(v)); /* row_bcast31 -> rows 2, 3 */                    \
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));             \
    }
FOT_WAVE_REDUCE_F32(wave_min_f32, fminf)
FOT_WAVE_REDUCE_F32(wave_max_f32, fmaxf)

// ---------------------------------------------------------------------------
// ego -> Frenet state
// ---------------------------------------------------------------------------

__device__ __forceinline__ ScanBest wave_argmin(ScanBest b)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ScanBest o;
        o.dist = __shfl_xor(b.dist, off, WAVE);
        o.idx = __shfl_xor(b.idx, off, WAVE);
        scan_merge(b, o);
    }
    return b;
}

// The reference spline (9 coefficient arrays of n knots) copied into LDS: the nearest-point search and the segment
// lookups are chains of dependent reads of these arrays, each an L2 round trip otherwise.  Paths with more knots than
// the launch reserved LDS for (lds_knots) stay in HBM.  Dynamic LDS: 9 * lds_knots doubles.
extern __shared__ double s_spl[];
constexpr int SPLINE_LDS_KNOTS = 512;

__device__ __forceinline__ SplineView stage_spline(const SplineView &g, int lds_knots, double *s_spl = fot::s_spl)
{
    if (g.n > lds_knots) return g;
    const double *src[9] = { g.s, g.ax, g.bx, g.cx, g.dx, g.ay, g.by, g.cy, g.dy };
    for (int i = threadIdx.x; i < 9 * g.n; i += blockDim.x) {
        const int a = i / g.n, j = i - a * g.n;
        s_spl[i] = src[a][j];
    }
    __syncthreads();
    SplineView l = g;
    l.s = s_spl; l.ax = s_spl + g.n; l.bx = s_spl + 2 * g.n; l.cx = s_spl + 3 * g.n; l.dx = s_spl + 4 * g.n;
    l.ay = s_spl + 5 * g.n; l.by = s_spl + 6 * g.n; l.cy = s_spl + 7 * g.n; l.dy = s_spl + 8 * g.n;
    return l;
}

// refine_nearest (fot_math.hpp), three rounds per evaluation.  A round compares the distances at s - ds, s, s + ds and
// either moves left, moves right or halves ds; the next round's probes depend on that outcome.  39 lanes evaluate, in
// one go, the probes of round r (3 lanes), of round r+1 under each of the 3 outcomes of r (9 lanes) and of round r+2
// under each of the 9 outcome pairs (27 lanes); the rounds are then resolved from the shuffled distances.  Every lane
// derives its (s, ds) with the very expressions of the sequential loop, so the result is bit-identical to it -- a
// third of the dependent spline evaluations.  Uniform across the wave.
__device__ __forceinline__ void refine_apply(int outcome, double s_end, double &s, double &ds)
{
    if (outcome == 0) s = fmax(0.0, s - ds);                      // left probe won
    else if (outcome == 1) s = fmin(s_end, s + ds);               // right probe won
    else ds *= 0.5;                                               // centre: halve the step
}

__device__ __forceinline__ double refine_nearest_wave(const SplineView &sp, double x, double y, double best_s, int lane)
{
    const double s_end = sp.s[sp.n - 1];
    // role of this lane: depth 0 (lanes 0-2), depth 1 (3-11: outcome o1), depth 2 (12-38: outcomes o1, o2)
    int depth = 0, o1 = 0, o2 = 0, pos = lane;
    if (lane >= 12) { const int i = lane - 12; depth = 2; o1 = i / 9; o2 = (i / 3) % 3; pos = i % 3; }
    else if (lane >= 3) { const int i = lane - 3; depth = 1; o1 = i / 3; pos = i % 3; }
    if (lane >= 39) { depth = 0; pos = 1; }                       // idle lanes probe the centre (values unused)
    double ds = 0.2;
    for (int it = 0; it < 20; it += 3) {
        double ms = best_s, mds = ds;
        if (depth >= 1) refine_apply(o1, s_end, ms, mds);
        if (depth >= 2) refine_apply(o2, s_end, ms, mds);
        const double probe = pos == 0 ? fmax(0.0, ms - mds) : (pos == 1 ? ms : fmin(s_end, ms + mds));
        double px, py;
        spline_xy(sp, probe, px, py);
        const double dist = hypot_cr(x - px, y - py);
        const int rounds = 20 - it < 3 ? 20 - it : 3;
        int base = 0, r1 = 0;
        for (int r = 0; r < rounds; ++r) {
            const double dist_left = __shfl(dist, base, WAVE), dist_curr = __shfl(dist, base + 1, WAVE),
                         dist_right = __shfl(dist, base + 2, WAVE);
            int o = 2;
            if (dist_left < dist_curr && dist_left < dist_right) o = 0;
            else if (dist_right < dist_curr && dist_right < dist_left) o = 1;
            refine_apply(o, s_end, best_s, ds);
            if (r == 0) { r1 = o; base = 3 + 3 * o; } else base = 12 + 9 * r1 + 3 * o;
        }
    }
    return best_s;
}

constexpr int FRENET_WG = 256;

// NanScan (fot_kernels.h): block `part` of `parts` of instance D flags the tracks [j0, j1) it owns.  A track is a
// (sample, pedestrian) pair, T points; the flags of the block's own tracks are cleared, then every NaN coordinate sets
// its track's flag (plain byte stores of the same value: no ordering between them is needed).
template <typename T>
__device__ __forceinline__ void scan_nan_tracks(const InstDesc &D, const T *__restrict__ dyn_xy, uint8_t *__restrict__ flag,
                                                int part, int parts, T *__restrict__ stage)
{
    if (D.dyn_mode == FOT_DYN_NONE) return;
    const int n_tracks = D.S * D.P;
    const int per = (n_tracks + parts - 1) / parts;
    const int j0 = part * per, j1 = j0 + per < n_tracks ? j0 + per : n_tracks;
    if (j0 >= j1) return;
    uint8_t *f = flag + D.nan_off;
    for (int j = j0 + (int)threadIdx.x; j < j1; j += FRENET_WG) f[j] = 0;
    __syncthreads();
    struct Pt { T x, y; };
    const Pt *base = (const Pt *)dyn_xy + D.dyn_off;
    Pt *out = stage ? (Pt *)stage + D.dyn_off : nullptr;          // (NanScan::stage: what is read is also copied into HBM)
    if (D.dyn_tmajor) {                                          // [T][S][P]: thread = track, rows n_tracks apart
        constexpr int TU = 8;                                     // rows in flight per thread (neighbouring threads read
        for (int j = j0 + (int)threadIdx.x; j < j1; j += FRENET_WG) {   // neighbouring points of a row)
            bool bad = false;
            for (int t0 = 0; t0 < D.T; t0 += TU) {
                Pt p[TU];
#pragma unroll
                for (int u = 0; u < TU; ++u) p[u] = base[(int64_t)(t0 + u < D.T ? t0 + u : D.T - 1) * n_tracks + j];
#pragma unroll
                for (int u = 0; u < TU; ++u) bad |= (p[u].x != p[u].x) | (p[u].y != p[u].y);
                if (out) {
#pragma unroll
                    for (int u = 0; u < TU; ++u) if (t0 + u < D.T) out[(int64_t)(t0 + u) * n_tracks + j] = p[u];
                }
            }
            if (bad) f[j] = 1;
        }
    } else {                                                     // [S][P][T]: the block's tracks are one contiguous run
        // 16-byte loads, sixteen in flight per thread (64 KB per block and round): the run is read once, at memory speed,
        // while the nearest-point blocks of the same launch wait on their chains of dependent spline evaluations
        typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int PER = (int)(16 / sizeof(T)) / 2;            // points per load: 2 (float) or 1 (double)
        constexpr int UNROLL = 16;
        const int64_t i0 = (int64_t)j0 * D.T, i1 = (int64_t)j1 * D.T;            // points [i0, i1)
        // head: points in front of the first 16-byte boundary (at most one, float only)
        int64_t a0 = i0;
        if (PER == 2 && ((uintptr_t)(base + i0) & 15)) {             // (global loads need dword alignment only; speed)
            if (threadIdx.x == 0) {
                const Pt p = base[i0];
                if ((p.x != p.x) | (p.y != p.y)) f[(int)(i0 / D.T)] = 1;
                if (out) out[i0] = p;
            }
            a0 = i0 + 1;
        }
        const int64_t n_vec = (i1 - a0) / PER;
        const V *vb = (const V *)(base + a0);
        V *vo = out ? (V *)(out + a0) : nullptr;
        for (int64_t vbase = threadIdx.x; vbase < n_vec; vbase += (int64_t)UNROLL * FRENET_WG) {
            V v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t vi = vbase + (int64_t)u * FRENET_WG;
                v[u] = vb[vi < n_vec ? vi : n_vec - 1];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t vi = vbase + (int64_t)u * FRENET_WG;
                if (vi >= n_vec) continue;
                if (vo) vo[vi] = v[u];
                bool b0 = (v[u][0] != v[u][0]) | (v[u][1] != v[u][1]);
                if (PER == 2) {
                    const bool b1 = (v[u][2 % (2 * PER)] != v[u][2 % (2 * PER)]) | (v[u][3 % (2 * PER)] != v[u][3 % (2 * PER)]);
                    if (b1) f[(int)((a0 + vi * PER + 1) / D.T)] = 1;
                }
                if (b0) f[(int)((a0 + vi * PER) / D.T)] = 1;
            }
        }
        // tail: the last point when an odd number is left (float only)
        if (PER == 2 && threadIdx.x == 0 && a0 + n_vec * PER < i1) {
            const Pt p = base[i1 - 1];
            if ((p.x != p.x) | (p.y != p.y)) f[(int)((i1 - 1) / D.T)] = 1;
            if (out) out[i1 - 1] = p;
        }
    }
}

// arg-min of a sample scan over the whole workgroup (lowest index wins ties); every thread gets the result
__device__ __forceinline__ ScanBest block_argmin(ScanBest b, ScanBest *s_best)
{
    b = wave_argmin(b);
    __syncthreads();                                              // s_best free again
    if ((threadIdx.x & (WAVE - 1)) == 0) s_best[threadIdx.x / WAVE] = b;
    __syncthreads();
    ScanBest r = s_best[0];
    for (int w = 1; w < FRENET_WG / WAVE; ++w) scan_merge(r, s_best[w]);
    return r;
}

// One workgroup (4 waves) per instance: the sample scans are spread over all threads, the refinement and the Frenet
// state are uniform values every wave computes alike.  `block` is the workgroup's role: an instance, or (behind the
// instances) one part of the NaN scan of an instance's tensor.  `sp`: the reference path, staged by the caller.
__device__ __forceinline__ void frenet_state_block(const DevParams *__restrict__ Pp, const SplineView &sp, const InstDesc *desc,
                                                   InstState *__restrict__ state, int n_inst, const MetaImport &imp,
                                                   const NanScan &scan, int32_t *__restrict__ inst_done, int block)
{
    __shared__ ScanBest s_best[FRENET_WG / WAVE];
    __shared__ InstDesc s_desc;                                   // the descriptor being worked on, read once
    int inst = block;
    if (inst >= n_inst) {
        // the blocks behind the nearest-point blocks: NaN scan of the dynamic tensors (memory-bound, on CUs whose
        // nearest-point block is a chain of dependent spline evaluations)
        const int b = inst - n_inst, si = b / scan.blocks_per_inst, part = b - si * scan.blocks_per_inst;
        if (si >= n_inst) return;
        const InstDesc &D = (imp.h_desc ? imp.h_desc : desc)[si];
        if (!D.dyn_tmajor && !scan.stage && !scan.eager) return;                // ([S][P][T] tensors in HBM: k_cull looks for NaNs itself)
        if (scan.dtype == FOT_F32)
            scan_nan_tracks(D, (const float *)scan.dyn_xy, scan.flag, part, scan.blocks_per_inst, (float *)scan.stage);
        else
            scan_nan_tracks(D, (const double *)scan.dyn_xy, scan.flag, part, scan.blocks_per_inst, (double *)scan.stage);
        return;
    }
    constexpr int DESC_WORDS = (int)(sizeof(InstDesc) / sizeof(unsigned long long));
    static_assert(sizeof(InstDesc) % sizeof(unsigned long long) == 0, "InstDesc is copied in 8-byte words");
    if (imp.h_desc) {
        // First kernel of a plan call: the descriptors still sit in the caller-side pinned staging block.  Every
        // workgroup moves its own instance's descriptor into HBM for the kernels that follow
        // (instead of a separate H2D copy in front of the launch sequence) and works from the staging copy itself.
        desc = imp.h_desc;
        const unsigned long long *src = (const unsigned long long *)&desc[inst];
        if (threadIdx.x < DESC_WORDS) {
            const unsigned long long v = src[threadIdx.x];
            ((unsigned long long *)&imp.d_desc[inst])[threadIdx.x] = v;
            ((unsigned long long *)&s_desc)[threadIdx.x] = v;
        }
        __syncthreads();
    } else {
        if (threadIdx.x < DESC_WORDS)
            ((unsigned long long *)&s_desc)[threadIdx.x] = ((const unsigned long long *)&desc[inst])[threadIdx.x];
        __syncthreads();
    }
    if (s_desc.ego.has_prev_s == FOT_PREV_S_CHAINED) return;      // handled by the head of its chain
    const DevParams &P = *Pp;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const double s_end = sp.s[sp.n - 1];
    const int n_glob = global_search_count(sp);
    double carry_prev_s = 0.0;                                 // new_prev_s of the previous member of the chain
    bool chained = false;
    const int n_members = 1 + s_desc.n_chained;
    for (int m = 0; m < n_members; ++m, ++inst) {
        if (m > 0) {                                           // next member of the chain: its descriptor
            __syncthreads();
            if (threadIdx.x < DESC_WORDS)
                ((unsigned long long *)&s_desc)[threadIdx.x] = ((const unsigned long long *)&desc[inst])[threadIdx.x];
            __syncthreads();
        }
        const InstDesc &D = s_desc;
        const double x = D.ego.x, y = D.ego.y;
        const bool has_prev = chained ? true : D.ego.has_prev_s != 0;
        const double prev_s = chained ? carry_prev_s : D.ego.prev_s;

        double best_s = 0.0;
        bool need_global = true;
        const bool given = D.ego.has_prev_s == FOT_EGO_IS_FRENET;  // the Frenet state itself: nothing to search
        if (given) need_global = false;
        if (has_prev && !given) {                              // cached window +-10 m, 100 samples
            const double s_min = fmax(0.0, prev_s - 10.0);
            const double s_max = fmin(s_end, prev_s + 10.0);
            ScanBest b = block_argmin(scan_samples(sp, x, y, s_min, s_max, 100, tid, FRENET_WG, false), s_best);
            best_s = b.idx >= 0 ? linspace_at(s_min, s_max, 100, b.idx) : 0.0;
            const bool at_lower = fabs(best_s - s_min) < 1e-3 && s_min > 0.0;
            const bool at_upper = fabs(best_s - s_max) < 1e-3 && s_max < s_end;
            need_global = at_lower || at_upper;
        }
        if (need_global) {
            ScanBest b = block_argmin(scan_samples(sp, x, y, 0.0, s_end, n_glob, tid, FRENET_WG, true), s_best);
            best_s = linspace_at(0.0, s_end, n_glob, b.idx >= 0 ? b.idx : 0);
        }
        if (!given) best_s = refine_nearest_wave(sp, x, y, best_s, lane);
        const double new_prev_s = given ? NAN : best_s;

        double fr[6], ref[6];
        bool ok = given ? frenet_state_given(sp, D.ego, fr, ref) : frenet_state_at(sp, D.ego, best_s, fr, ref);
        if (!ok && !given) {
            double px, py;
            spline_xy(sp, best_s, px, py);
            if (isnan(px) || isnan(py)) {                     // coordinate_converter.py:289-295
                ScanBest b = block_argmin(scan_samples(sp, x, y, 0.0, s_end, n_glob, tid, FRENET_WG, true), s_best);
                best_s = linspace_at(0.0, s_end, n_glob, b.idx >= 0 ? b.idx : 0);
                ok = frenet_state_at(sp, D.ego, best_s, fr, ref);
            }
        }
        if (tid == 0) {
            if (inst_done) inst_done[inst] = 0;                   // tiles of this instance evaluated so far (tile_done)
            InstState &S = state[inst];
            for (int i = 0; i < 6; ++i) { S.frenet0[i] = ok ? fr[i] : NAN; S.ref0[i] = ok ? ref[i] : NAN; }
            S.new_prev_s = new_prev_s;
            S.c2f_ok = ok ? 1 : 0;
            const int nb = (ok && fr[1] > 0.1) ? P.n_brake : 0;   // BRAKE_MIN_SPEED gate
            S.n_brake = nb;
            S.n_cand = ok ? D.n_grid + nb : 0;
        }
        carry_prev_s = new_prev_s;
        chained = true;
    }
}

__global__ void __launch_bounds__(FRENET_WG)
k_frenet_state(const DevParams *__restrict__ Pp, SplineView sp_hbm, int lds_knots, const InstDesc *desc,
               InstState *__restrict__ state, int n_inst, MetaImport imp, NanScan scan, int32_t *__restrict__ inst_done)
{
    SplineView sp = sp_hbm;
    if ((int)blockIdx.x < n_inst) sp = stage_spline(sp_hbm, lds_knots);       // (the scan blocks never look at the path)
    frenet_state_block(Pp, sp, desc, state, n_inst, imp, scan, inst_done, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------
// candidate evaluation
// ---------------------------------------------------------------------------

// a wave-uniform double as an opaque scalar-register value
__device__ __forceinline__ double uniform_f64(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    asm volatile("" : "+s"(lo), "+s"(hi));                       // opaque: a register value, not a re-loadable address
    return __hiloint2double(hi, lo);
}

typedef float f16 __attribute__((ext_vector_type(16)));           // one chunk = 8 (x, y) pairs in 16 SGPRs

// s_load_dwordx16 of the chunk OFF bytes behind src, NOT waited for (see FusedSink::put)
template <int OFF>
__device__ __forceinline__ void sload_chunk(f16 &dst, const f2x8 *src)
{
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(dst) : "s"(src), "n"(OFF) : "memory");
}

// the same, issued before anything that reads `ahead` afterwards: the compiler otherwise sinks the load below the
// distance arithmetic of the chunk in use (which does not depend on it) and the wait then follows it at once
template <int OFF>
__device__ __forceinline__ void sload_chunk_ahead(f16 &dst, const f2x8 *src, float &ahead)
{
    asm volatile("s_load_dwordx16 %0, %2, %3" : "=s"(dst), "+v"(ahead) : "s"(src), "n"(OFF) : "memory");
}

// waits for every outstanding scalar load; `c` is tied in so that its uses stay behind the wait
__device__ __forceinline__ void swait_chunk(f16 &c)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c) : : "memory");
}

// smallest float32 squared distance from (fx, fy) to the 8 entries of a chunk held in SGPRs as x[8], y[8].
// Two obstacles per instruction: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on (x_j, x_j+1) and (y_j, y_j+1).
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float min_sqdist32_f16(const f16 &c, float fx, float fy)
{
    const v2f px = { fx, fx }, py = { fy, fy };
    v2f t[ENT_CHUNK / 2];
#pragma unroll
    for (int j = 0; j < ENT_CHUNK / 2; ++j) {
        const v2f ox = { c[2 * j], c[2 * j + 1] }, oy = { c[8 + 2 * j], c[8 + 2 * j + 1] };
        const v2f dx = px - ox, dy = py - oy;
        t[j] = __builtin_elementwise_fma(dy, dy, dx * dx);
    }
    float m = fminf(fminf(t[0].x, t[0].y), t[1].x);
    m = fminf(fminf(m, t[1].y), t[2].x);
    m = fminf(fminf(m, t[2].y), t[3].x);
    return fminf(m, t[3].y);
}

// EntryCollider (fot_math.hpp) with the chunk walk on the scalar unit.  Everything that addresses the entry lists is
// wave-uniform (instance, time step), so one chunk (8 obstacles, 64 B) is one s_load_dwordx16 shared by the 64
// candidates of the wave; each lane keeps the float32 minimum squared distance of the chunk and only chunks that
// come within the conservative threshold are re-checked in float64, so the decision is the reference's.
// Scalar registers are the scarce resource of k_evaluate (two 16-register chunk buffers): the sink keeps only
// what the per-chunk loop needs, the exact re-check fetches its constants where it runs.
// k_evaluate's only parameter, i.e. the layout of its argument segment.  What only rare branches or the epilogue
// read -- the spline view, the float64 / sample-id entry arrays, the output arrays -- is fetched from the segment at
// the point of use instead of occupying scalar registers across the time-step loop.
// Agent-coherent accesses to the candidate arrays: written by one wave, read by the wave that selects for the
// instance -- possibly on another XCD, whose L2 is a different one.  Relaxed atomic accesses at agent scope are plain
// loads / stores with the sc1 bit (write-through / read-through the XCD's L2): no L2 writeback or invalidation, which
// is what an agent-scope fence costs on this multi-XCD part (every wave paying one made the launch 45 % longer).
template <typename T>
__device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct EvalKernArgs {
    const DevParams *Pp; SplineView sp; const InstDesc *desc; const InstState *state;
    int row_budget, lds_knots, ablate, n_inst, max_tiles;
    const int32_t *tile_cand0, *tile_n;
    const TileStep *wave_rng; const f2 *ent32; const d2 *ent64; const uint8_t *ent_sid;
    double *cand_cost; uint8_t *cand_status; uint16_t *cand_keep;   // per candidate: for fot_debug_candidates only
    TilePart *parts;                                 // per tile: what its wave found (tile_done)
    fot_result *out; int32_t *inst_done;             // selection by the wave that finishes an instance's last tile
    int32_t *done_flag; int32_t done_seq;            // host-visible flag per record (CandArrays::done_flag)
};
// k_evaluate's argument segment: EVAL_LEAD_PTRS read-only pointers (passed on their own so that they carry
// `__restrict__`: only no-alias inputs are certain to keep their loads on the scalar unit), then this struct
constexpr int EVAL_LEAD_PTRS = 7;
__device__ __forceinline__ const EvalKernArgs &eval_kernargs()
{
    return *(const EvalKernArgs *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + EVAL_LEAD_PTRS * sizeof(void *));
}

struct FusedSink {
    const DevParams *Pp;
    const InstDesc *Dp;
    // The per-step values of the tile (TileStep) sit in lanes: lane l holds time step step_base + l, 64 steps at a time;
    // a walk that crosses a multiple of 64 (more than 64 samples per candidate) reloads them there (steps_reload).
    uint32_t my_rng;                     // lane l: strip range of time step step_base + l (0: nothing to test)
    float my_thr, my_thr_sure;           // lane l: the float32 thresholds of that time step (TileStep)
    float my_thr_fatal;                  // lane l: my_thr_sure where a certain hit settles the candidate (no chance
                                         // budget), else -1
    int step_base;                       // multiple of 64 (wave-uniform)
    int step_row;                        // number of this tile in the batch: its TileSteps start at step_row * n_total
    int lane_id;
    float thr, thr_fatal;                // of the current time step (wave-uniform)
    const f2x8 *chunks;                  // float32 entries of this instance, ent_cap / 8 chunks per time step
    int chunks_per_k;                    // ent_cap / 8
    uint64_t hit_mask;
    int viol;
    bool hit;
    int c_lo, n_chunks;                  // chunk range of the current time step
    uint32_t pf;                         // destination of the warm-up loads below, reserved until they have landed
    bool no_warm;

    // Reads the range of time step k and touches the first cache lines of its chunks, so that they are on their
    // way into the scalar cache while the sample arithmetic runs.  The loads are hand-issued and their (unused)
    // destination register stays tied to the sink until pf_wait(): a scalar load retires at any later time, and
    // must not land in a register the compiler has meanwhile given to something else.
    // lane l <- TileStep of time step base + l of this tile (every lane of the wave is active here: the region that
    // calls row_begin holds lanes 0 .. min(n_total, 64) - 1 at least, all 64 once a second block of steps exists)
    __device__ __forceinline__ void steps_load(int base)
    {
        const DevParams &P = *Pp;
        const EvalKernArgs &KA = eval_kernargs();
        TileStep st = { 0u, 0.0f, 0.0f, 0u };
        if (Dp->ent_cap != 0 && base + lane_id < P.n_total && !(KA.ablate & 1))
            st = KA.wave_rng[(int64_t)step_row * P.n_total + base + lane_id];
        my_rng = st.rng; my_thr = st.thr; my_thr_sure = st.thr_sure;
        my_thr_fatal = Dp->max_viol == 0 ? st.thr_sure : -1.0f;
        step_base = base;
    }

    __device__ __forceinline__ void row_begin(int k)
    {
        if ((k & ~(WAVE - 1)) != step_base) steps_load(k & ~(WAVE - 1));      // wave-uniform; never taken up to 64 samples
        const int kl = k & (WAVE - 1);
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)my_rng, kl);
        c_lo = (int)(r >> 16);
        n_chunks = (int)(r & 0xffffu) - c_lo;
        thr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_thr), kl));
        thr_fatal = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_thr_fatal), kl));
        if (n_chunks > 0 && !no_warm) {
            const f2x8 *row = chunks + (int64_t)k * chunks_per_k + c_lo;
            asm volatile("s_load_dword %0, %1, 0x0\n\t"
                         "s_load_dword %0, %1, 0x40\n\t"
                         "s_load_dword %0, %1, 0x80\n\t"
                         "s_load_dword %0, %1, 0xc0"
                         : "=&s"(pf) : "s"(row) : "memory");          // (early clobber: four loads read `row`)
        } else {
            pf = 0;
        }
    }

    // end of the time step, reached by every lane: the warm-up loads have landed, pf may be reused
    __device__ __forceinline__ void row_end(int) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pf) : : "memory"); }

    __device__ __forceinline__ void put(int k, int ci, double px, double py, bool alive)
    {
        // (the rows are instance-local: evaluate_tile) -- the point itself is its own exact form
        put32(k, ci, (float)px, (float)py, alive, [&](double &ex, double &ey) { ex = px; ey = py; });
    }

    // the point in the instance-local float32 frame; get_exact(px, py) yields its float64 coordinates, asked for only by
    // the lanes whose float32 distance to some entry lies between the two thresholds
    template <class GetExact>
    __device__ __forceinline__ void put32(int k, int, float fx_in, float fy, bool alive, const GetExact &get_exact)
    {
        if (n_chunks == 0) return;                                // wave-uniform
        if (!alive || hit) return;                                // lanes whose collision outcome is already settled
        float fx = fx_in;                                         // (tied into the hand-issued loads below)
        bool sure = false;                                        // some obstacle is certainly within its radius
        const f2x8 *row = chunks + (int64_t)k * chunks_per_k + c_lo;
        for (int c0 = 0; c0 < n_chunks; c0 += 32) {               // 32 chunks per pass: one bit per chunk and lane
            const int nb = n_chunks - c0 < 32 ? n_chunks - c0 : 32;   // (may be odd: the last pair then tests one chunk)
            const f2x8 *cp = row + c0;
            uint32_t near_bits = 0;                               // chunk c0+i within the threshold -> bit nb-1-i
            // Branch-free loop over two chunk buffers filled by hand-issued scalar loads.  SMEM returns out of
            // order, so a compiler-placed wait for the chunk in use would also wait for the prefetch behind it;
            // the loads are therefore inline asm (invisible to the waitcnt pass) and each buffer is waited for
            // right before its own use, one chunk of arithmetic after its load was issued.  The last pair
            // prefetches one chunk past the list (allocated slack, never used).
            f16 ca, cb;
            sload_chunk<0>(ca, cp);
            swait_chunk(ca);
            for (int c = 0; c < nb; c += 2) {
                sload_chunk_ahead<64>(cb, cp, fx);
                const float ma = min_sqdist32_f16(ca, fx, fy);
                near_bits = (near_bits << 1) | (uint32_t)(ma <= thr);
                sure |= ma <= thr_fatal;
                swait_chunk(cb);
                sload_chunk_ahead<128>(ca, cp, fx);
                if (c + 1 < nb) {                                     // wave-uniform
                    const float mb = min_sqdist32_f16(cb, fx, fy);
                    near_bits = (near_bits << 1) | (uint32_t)(mb <= thr);
                    sure |= mb <= thr_fatal;
                }
                swait_chunk(ca);
                cp += 2;
            }
            // a single violation is fatal (no chance constraint budget): a certain float32 hit settles the candidate
            if (sure) { hit = true; return; }
            if (near_bits != 0) {
                double px, py;
                get_exact(px, py);
                exact(k, c0, nb, near_bits, px, py, fx, fy);
            }
        }
    }

    // rare: exact float64 re-check of the chunks whose float32 distance came within the threshold
    __device__ __forceinline__ void exact(int k, int c0, int nb, uint32_t near_bits, double px, double py, float fx,
                                          float fy)
    {
        const DevParams &P = *Pp;
        const InstDesc &D = *Dp;
        px += D.ego.x; py += D.ego.y;                             // the exact entries are in the caller's frame
        const double sq_dyn = D.dyn_mode == FOT_DYN_SINGLE ? P.sq_r_dyn : P.sq_r;
        const EvalKernArgs &KA = eval_kernargs();
        const float thr_sure = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_thr_sure), k & (WAVE - 1)));
        const d2 *e64 = KA.ent64 + D.ent_off;
        const uint8_t *sid = KA.ent_sid + D.ent_off;
        const int64_t base = ((int64_t)k * chunks_per_k + c_lo) * ENT_CHUNK;
        while (near_bits != 0 && !hit) {
            const int hb = 31 - __clz((int)near_bits);             // highest bit = earliest chunk
            near_bits &= ~(1u << hb);
            const int64_t e = base + (int64_t)(c0 + nb - 1 - hb) * ENT_CHUNK;
            exact_chunk_f32first(chunks[e / ENT_CHUNK], e64 + e, sid + e, fx, fy, thr, thr_sure, px, py, P.sq_r, sq_dyn,
                                 D.max_viol, hit_mask, viol, hit);
        }
    }

    __device__ __forceinline__ bool collided() const { return hit; }
};

// Longitudinal rows of a tile's profiles in the wave's own slice of LDS: the 64 candidates of a tile share two or
// three longitudinal profiles (seven short ones in the brake-ladder tile).  Per profile a run of rows [k][field],
// 9 fields = 72 contiguous bytes per row (the arc length s is not in the row: only the low-speed rule and the
// travelled distance read it, and they rebuild it from the profile's polynomial); a profile's run holds
// profile_rows() rows and lanes clamp their row index to it (the last row of a brake-ladder profile is its hold).
constexpr int EVAL_WG = WAVES_PER_GROUP * WAVE;
extern __shared__ double s_lon[];

constexpr int ROW_FIELDS = 9;
constexpr int LONINFO_DOUBLES = (int)(sizeof(LonInfo) / sizeof(double));
static_assert(sizeof(LonInfo) % sizeof(double) == 0, "LonInfo sits in the double-typed LDS array");
// doubles of LDS per wave: rows | profile summaries | row offsets of the profiles (ints)
__host__ __device__ constexpr int eval_wave_doubles(int row_budget)
{
    return row_budget * ROW_FIELDS + TILE_MAX_PROFILES * LONINFO_DOUBLES + TILE_MAX_PROFILES / 2;
}
// doubles of LDS of a group's shared table (k_evaluate_group)
__host__ __device__ constexpr int eval_group_doubles()
{
    return GROUP_ROWS * ROW_FIELDS + GROUP_MAX_PROFILES * LONINFO_DOUBLES + GROUP_MAX_PROFILES / 2;
}
// A tile cut into time segments (k_evaluate_split): what a later segment hands to the wave of segment 0, per lane
constexpr int SEG_MAX = 4;
constexpr int SEG_F64 = 4, SEG_I32 = 4;  // d_last, v_last, max_step2, hit_mask | flags, first_nan, k_last, hit
constexpr int SEG_DOUBLES = (SEG_F64 + SEG_I32 / 2) * WAVE;

struct StagedTab {
    static constexpr bool LOCAL = true;  // rx, ry relative to the ego position (evaluate_tile builds them so)
    int lds_row0;                        // index of this lane's profile row 0 in s_lon
    int k_max;                           // last row of the run (grid: n_t - 1; brake ladder: the hold row n_eval)
    double dt;
#ifdef FOT_TIMELINE
    mutable uint64_t t_rows = 0;         // shader cycles between issuing the row reads and having them (diagnostic)
#endif
    __device__ __forceinline__ void load(int k, LonSample &L) const
    {
#ifdef FOT_TIMELINE
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
#endif
        const int kk = k < k_max ? k : k_max;
        const double *r = s_lon + lds_row0 + kk * ROW_FIELDS;
        L.s = 0.0; L.sd = r[0]; L.sdd = r[1]; L.rx = r[2]; L.ry = r[3];
        L.cos_r = r[4]; L.sin_r = r[5]; L.kr = r[6]; L.dkr = r[7]; L.inv_sd = r[8];
        // the row is in registers from here on: what follows (the sink's scalar warm-up loads) must not sit
        // between these reads and the wait for them
        asm volatile("" : "+v"(L.sd), "+v"(L.sdd), "+v"(L.rx), "+v"(L.ry), "+v"(L.cos_r), "+v"(L.sin_r),
                          "+v"(L.kr), "+v"(L.dkr), "+v"(L.inv_sd));
#ifdef FOT_TIMELINE
        t_rows += __builtin_amdgcn_s_memtime() - t0;
#endif
    }
    // arc length s(t_k): asked for by the low-speed rules only (a few percent of the samples, but every eighth time
    // step of a wave has such a lane) -- lanes re-read their profile's coefficients from the tile's summaries
    const LonInfo *info;                 // in LDS
    __device__ __forceinline__ double s_at(int k) const
    {
        double s_, u0, u1, u2;
        lon_sample(*info, k, dt, s_, u0, u1, u2);
        return s_;
    }
};

#ifdef FOT_TIMELINE
// diagnostic build (scripts/timeline.sh): per tile of the last k_evaluate launch -- tile start, start of the sample
// loop (after the LDS rows are built), tile end on the 100 MHz clock, and the chunks its strip ranges cover
__device__ uint64_t g_timeline[4 * 16384];
__device__ uint64_t g_shader_clock[2 * 16384];     // s_memtime at loop start / end: shader cycles (in-kernel clock)
__device__ uint64_t g_row_wait[16384];             // shader cycles lane 0 spent between issuing and having its LDS rows
__device__ uint64_t g_phase[4 * 16384];            // kernel entry, spline staged, profile summaries ready, loop done
__shared__ uint64_t s_tl_entry[2];                 // kernel entry / spline staged, of this workgroup
extern "C" int fot_timeline_read_phase(uint64_t *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(uint64_t) * (size_t)n_words);
}
extern "C" int fot_timeline_read_rows(uint64_t *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_row_wait), sizeof(uint64_t) * (size_t)n_words);
}
extern "C" int fot_timeline_read(uint64_t *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(uint64_t) * (size_t)n_words);
}
extern "C" int fot_timeline_read_clock(uint64_t *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_shader_clock), sizeof(uint64_t) * (size_t)n_words);
}
#endif

// this wave's LDS writes (other lanes') before the reads that follow, and the reads before the next writes
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One tile: rows of its profiles into LDS, then one candidate per lane.
//   TILE_WAVE   the wave owns the tile and its slice of LDS (rows_base).
//   TILE_SPLIT  the workgroup's n_sub waves share the tile -- they build the rows together, wave `sub` walks the time
//               steps [sub, sub + 1) * ceil(n_loop / n_sub) of every candidate, and wave 0 merges the segments
//               (seg_merge) through `s_part`.
//   TILE_GROUP  the workgroup's GROUP_TILES waves take the tiles grp_tile0 + sub of one group (fot_math.hpp): they
//               build the rows of ALL the group's profiles together, each walks its own tile.
enum { TILE_WAVE = 0, TILE_SPLIT = 1, TILE_GROUP = 2 };
template <int MODE>
__device__ __forceinline__ void evaluate_tile(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc,
                                              const InstState *__restrict__ state,
                                              const int32_t *__restrict__ tile_cand0, const int32_t *__restrict__ tile_n,
                                              const TileStep *__restrict__ wave_rng, const f2 *__restrict__ ent32,
                                              const EvalKernArgs &a, const SplineView &sp_lds,
                                              double *my_rows, int inst, int tile, int lane, int tl_tag, TilePart &tp,
                                              int sub = 0, int n_sub = 1, double *s_part = nullptr, int grp_tile0 = 0)
{
    constexpr bool SPLIT = MODE == TILE_SPLIT, GROUP = MODE == TILE_GROUP;
    const int seg = SPLIT ? sub : 0, n_seg = SPLIT ? n_sub : 1;
    const DevParams &P = *Pp;
    const InstDesc &D = desc[inst];
    const InstState &S = state[inst];
    const int n_total = P.n_total;
    const int cand0 = tile_cand0[D.shape_off + tile];
    int n = tile_n[D.shape_off + tile];
    // candidates whose profiles are staged in this slice: the tile's own, or the whole group's
    int stage_c0 = cand0, stage_c1 = cand0 + n;                  // [first, one past the last)
    if constexpr (GROUP) {
        stage_c0 = tile_cand0[D.shape_off + grp_tile0];
        stage_c1 = tile_cand0[D.shape_off + grp_tile0 + GROUP_TILES - 1] + tile_n[D.shape_off + grp_tile0 + GROUP_TILES - 1];
    }
    if constexpr (!GROUP) { if (n <= 0) return; }               // (a padding tile of the grouped cut)
    if (!S.c2f_ok || stage_c0 >= S.n_cand) return;              // (the brake ladder of a standing ego) -- the whole workgroup
    if (stage_c1 > S.n_cand) stage_c1 = S.n_cand;
    if (cand0 + n > S.n_cand) n = S.n_cand - cand0 > 0 ? S.n_cand - cand0 : 0;
#ifdef FOT_TIMELINE
    const uint64_t t_blk = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int PROF_CAP = GROUP ? GROUP_MAX_PROFILES : TILE_MAX_PROFILES;
    LonInfo *s_info = (LonInfo *)(my_rows + (GROUP ? GROUP_ROWS : a.row_budget) * ROW_FIELDS);
    int *s_row0 = (int *)(s_info + PROF_CAP);
    // staged profiles: of the first .. last staged candidate (slots grow with the candidate index)
    const int slot_lo = decode_candidate(P, D, S.frenet0, stage_c0).lon_slot;
    const int n_stage = decode_candidate(P, D, S.frenet0, stage_c1 - 1).lon_slot - slot_lo + 1;
    const auto lds_fence = [] { if constexpr (MODE != TILE_WAVE) __syncthreads(); else wave_lds_fence(); };
    if constexpr (MODE == TILE_WAVE) wave_lds_fence();           // the previous tile's rows are no longer read
    // who builds: the wave itself, or all waves of the workgroup together
    const int bld_id = (MODE == TILE_WAVE ? 0 : sub * WAVE) + lane, bld_n = (MODE == TILE_WAVE ? 1 : n_sub) * WAVE;
    int total_rows = 0, n_loop = 0;
    for (int p = 0; p < n_stage; ++p) {                         // wave-uniform: a handful of scalar operations
        const int r = profile_rows(P, D, slot_lo + p);
        if (bld_id == p) s_row0[p] = total_rows;
        total_rows += r;
    }
    if (bld_id < n_stage) s_info[bld_id] = profile_info(P, D, S.frenet0, slot_lo + bld_id, true);
    lds_fence();
#ifdef FOT_TIMELINE
    const uint64_t t_info = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = bld_id; i < total_rows; i += bld_n) {
        int p = 0;
        for (int pp = 1; pp < n_stage; ++pp) p = i >= s_row0[pp] ? pp : p;
        const int k = i - s_row0[p];
        const LonInfo Lp = s_info[p];
        LonSample ls;
        double sddd;
        make_lon_sample(sp_lds, Lp, k, P.dt, ls, sddd);         // (k == n_eval of a brake profile: its hold state)
        double *r = my_rows + (int64_t)i * ROW_FIELDS;
        // the reference point in the instance-local frame (origin: the ego position, like the entry lists): the walk then
        // hands its points to the collision test as they are, and nothing else of it depends on the frame
        r[0] = ls.sd; r[1] = ls.sdd; r[2] = ls.rx - D.ego.x; r[3] = ls.ry - D.ego.y;
        r[4] = ls.cos_r; r[5] = ls.sin_r; r[6] = ls.kr; r[7] = ls.dkr; r[8] = ls.inv_sd;
    }
    lds_fence();
    if constexpr (GROUP) { if (n <= 0) return; }                // (a padding tile, or one past a standing ego's lattice)
    // the time steps this tile needs: the longest of its OWN profiles
    {
        const int own_lo = decode_candidate(P, D, S.frenet0, cand0).lon_slot - slot_lo;
        const int own_hi = decode_candidate(P, D, S.frenet0, cand0 + n - 1).lon_slot - slot_lo;
        for (int p = own_lo; p <= own_hi; ++p) { const int nt = s_info[p].n_t; n_loop = nt > n_loop ? nt : n_loop; }
        n_loop = __builtin_amdgcn_readfirstlane(n_loop);
    }
    // this wave's time steps
    const int seg_len = (n_loop + n_seg - 1) / n_seg;
    const int k0 = SPLIT ? seg * seg_len : 0;
    const int k1 = SPLIT ? (k0 + seg_len < n_loop ? k0 + seg_len : n_loop) : n_loop;

    // lane l holds the strip range of time step step_base + l (read back with v_readlane), 64 steps at a time: loaded
    // while every lane of the wave is still active, lanes without a candidate included
    const int step_base0 = k0 & ~(WAVE - 1);
    TileStep my_step = { 0u, 0.0f, 0.0f, 0u };
    if (D.ent_cap != 0 && step_base0 + lane < n_total && !(a.ablate & 1))
        my_step = wave_rng[(int64_t)(D.tile0 + tile) * n_total + step_base0 + lane];
    const uint32_t my_rng = my_step.rng;
    // (while every lane is active: the time-step loop reads lane k of these, candidate or not)
    const float my_thr_fatal = D.max_viol == 0 ? my_step.thr_sure : -1.0f;
#ifdef FOT_TIMELINE
    const uint64_t t_wave = __builtin_amdgcn_s_memrealtime();
    const uint64_t c_wave = __builtin_amdgcn_s_memtime();
    const int tl_wave = D.tile0 + tile;
    uint64_t tl_rows = 0;
    int tl_chunks = (int)(my_rng & 0xffffu) - (int)(my_rng >> 16);
    for (int o = 32; o > 0; o >>= 1) tl_chunks += __shfl_xor(tl_chunks, o);
#endif
    // per-lane state that outlives the segment loop in the split form
    SegState g;
    seg_init(g);
    LonInfo L;
    StagedTab tab;
    uint64_t hit_mask = 0;
    bool hit = false;
    int64_t slot = 0;
    double q[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };              // lateral quintic of the lane's candidate
    // Lanes 0 .. n_total-1 hold the per-step values of the tile (my_step) that the time-step loop reads ACROSS lanes
    // (v_readlane of lane k): they must be active wherever the compiler may place a copy of those registers, so they
    // all enter the region below -- a lane without a candidate walks an empty path (n_t = 0).
    const bool has_cand = lane < n;
    if (has_cand || lane < n_total) {                            // (n_total > 64: every lane)
        const int idx = cand0 + (has_cand ? lane : 0);           // candidate index inside the instance
        slot = (int64_t)D.cand_off + idx;
        const CandDecode cd = decode_candidate(P, D, S.frenet0, idx);
        const int p = cd.lon_slot - slot_lo;
        L = s_info[p];
        if (!has_cand) L.n_t = 0;
        tab.lds_row0 = (int)(my_rows - s_lon) + s_row0[p] * ROW_FIELDS;
        tab.k_max = profile_rows(P, D, cd.lon_slot) - 1;
        tab.info = s_info + p;
        tab.dt = P.dt;
        lat_coeffs(S.frenet0, cd.di, cd.brake ? P.brake[cd.ti] : P.ti[cd.ti], q);
        // q[0..2] are the instance's lateral state (wave-uniform): as scalar values they end up in spilled SGPRs and
        // come back through v_readlane in every time step -- three vector registers are cheaper
        asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]));

        FusedSink sink;
        sink.Pp = Pp; sink.Dp = &D;
        sink.my_rng = my_rng;
        sink.my_thr = my_step.thr; sink.my_thr_sure = my_step.thr_sure; sink.thr = 0.0f; sink.thr_fatal = -1.0f;
        sink.my_thr_fatal = my_thr_fatal;
        sink.step_base = step_base0; sink.step_row = D.tile0 + tile; sink.lane_id = lane;
        sink.chunks = (const f2x8 *)(ent32 + D.ent_off);
        sink.chunks_per_k = D.ent_cap / ENT_CHUNK;
        sink.hit_mask = 0; sink.viol = 0; sink.hit = false; sink.n_chunks = 0; sink.c_lo = 0; sink.pf = 0;
        sink.no_warm = (a.ablate & 2) != 0;
        // loop constants as opaque register values: the compiler then keeps them instead of re-fetching each one from
        // the parameter blocks, behind a scalar-memory wait, in every time step
        LoopConst lc = loop_const(P, D);
        // The limits live in VECTOR registers although they are wave-uniform: as scalars they do not fit next to the two
        // chunk buffers, get lane-spilled and come back through ~27 v_readlane per time step -- vector issue slots of a
        // kernel that is bound by exactly those (r02: 0.257 -> 0.242 ms).  12 VGPRs; the kernel still fits 168.
        asm volatile("" : "+v"(lc.dt), "+v"(lc.lim_speed), "+v"(lc.lim_accel), "+v"(lc.lim_curv), "+v"(lc.lim_lat),
                          "+v"(lc.road_lim));
        lc.n_circ_fp = __builtin_amdgcn_readfirstlane(lc.n_circ_fp);
        asm volatile("" : "+s"(lc.n_circ_fp));
        evaluate_segment(P, lc, L, tab, q, k0, k1, sink, g);   // (SPLIT: a quarter of the steps -- latency, not issue, bound)
        hit_mask = sink.hit_mask; hit = sink.hit;
#ifdef FOT_TIMELINE
        tl_rows = tab.t_rows;
#endif
    }
#ifdef FOT_TIMELINE
    const uint64_t t_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (SPLIT) {
        // segments 1.. leave their state in LDS, the wave of segment 0 folds them in, in time order
        double *pf = s_part + (seg - 1) * SEG_DOUBLES;
        int *pi = (int *)(pf + SEG_F64 * WAVE);
        if (seg > 0 && has_cand) {
            pf[0 * WAVE + lane] = g.d_last; pf[1 * WAVE + lane] = g.v_last;
            pf[2 * WAVE + lane] = g.acc.max_step2; ((uint64_t *)pf)[3 * WAVE + lane] = hit_mask;
            pi[0 * WAVE + lane] = (int)g.acc.fl; pi[1 * WAVE + lane] = g.first_nan;
            pi[2 * WAVE + lane] = g.k_last; pi[3 * WAVE + lane] = hit ? 1 : 0;
        }
        __syncthreads();
        if (seg != 0) return;
        if (has_cand) {
            for (int sg = 1; sg < n_seg; ++sg) {
                if (sg * seg_len >= L.n_t) break;                // that segment held no sample of this candidate
                const double *qf = s_part + (sg - 1) * SEG_DOUBLES;
                const int *qi = (const int *)(qf + SEG_F64 * WAVE);
                SegState nx;
                nx.d_last = qf[0 * WAVE + lane]; nx.v_last = qf[1 * WAVE + lane];
                nx.acc.max_step2 = qf[2 * WAVE + lane];
                nx.acc.fl = (uint32_t)qi[0 * WAVE + lane]; nx.first_nan = qi[1 * WAVE + lane];
                nx.k_last = qi[2 * WAVE + lane];
                if (seg_merge(g, nx)) {
                    hit_mask |= ((const uint64_t *)qf)[3 * WAVE + lane];
                    hit |= qi[3 * WAVE + lane] != 0;
                }
            }
            hit |= __popcll(hit_mask) > D.max_viol;
        }
    }
    // The candidate's final status (stop-distance filter included, frenet_planner.py:307-324) and, per tile, what the
    // selection needs: the histogram of the statuses and the cheapest 'ok' candidate (lowest index among equals)
    int st_final = FOT_ST_DROPPED;
    ScanBest mine = { INFINITY, -1 };
    int my_keep = 0;
    if (has_cand) {
        CandResult r;
        {
            LonInfo Lf = *tab.info;                              // (read again here: not carried through the walk)
            Lf.n_t = L.n_t;
            finish_candidate(P, D, Lf, tab, q, g, hit, r);
        }
        st_final = final_status(r.status, r.v_last, r.travel, D.max_stop);
        const EvalKernArgs &KA = eval_kernargs();
        KA.cand_cost[slot] = r.cost;
        KA.cand_status[slot] = (uint8_t)st_final;
        KA.cand_keep[slot] = (uint16_t)r.keep;
        my_keep = r.keep;
        if (st_final == FOT_ST_OK) { mine.dist = r.cost; mine.idx = cand0 + lane; }
    }
    {
        const ScanBest best = wave_argmin(mine);                 // every lane of the wave is here
        tp.cost = best.dist; tp.idx = best.idx;
        tp.keep = __builtin_amdgcn_readfirstlane(__shfl(my_keep, best.idx >= 0 ? best.idx - cand0 : 0, WAVE));
#pragma unroll
        for (int c = 0; c < 8; ++c) tp.cnt[c] = __popcll(__ballot(st_final == c));
    }
#ifdef FOT_TIMELINE
    tl_rows = (uint64_t)__shfl((unsigned long long)tl_rows, 0);
    if (tl_wave < 16384 && lane == 0) {
        g_timeline[tl_wave * 4 + 0] = t_blk; g_timeline[tl_wave * 4 + 1] = t_wave;
        g_timeline[tl_wave * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        g_timeline[tl_wave * 4 + 3] = (uint64_t)(uint32_t)tl_chunks | ((uint64_t)(uint32_t)tl_tag << 32);
        g_shader_clock[tl_wave * 2 + 0] = c_wave; g_shader_clock[tl_wave * 2 + 1] = __builtin_amdgcn_s_memtime();
        g_row_wait[tl_wave] = tl_rows;
        g_phase[tl_wave * 4 + 0] = s_tl_entry[0]; g_phase[tl_wave * 4 + 1] = s_tl_entry[1];
        g_phase[tl_wave * 4 + 2] = t_info; g_phase[tl_wave * 4 + 3] = t_loop_end;
    }
#endif
}

// ---------------------------------------------------------------------------
// selection + output (reference: frenet_planner.py:294-324, 1235-1259)
// ---------------------------------------------------------------------------

// One wave selects for one instance: stop-distance filter, rejection histogram, first-minimum arg-min (lowest index
// wins ties), the selected path rebuilt from its profile, the record written.  Run by the wave that finishes the
// instance's LAST tile (tile_done below) -- no separate launch behind the evaluation, and the selections of the
// instances that finish early overlap with the evaluation of the others.  A real function: its registers (final_sample
// holds an arc tangent) must not count against the time-step loop's.
__device__ __forceinline__ TilePart tile_part_empty()
{
    TilePart t;
    t.cost = INFINITY; t.idx = -1; t.keep = 0;
    for (int c = 0; c < 8; ++c) t.cnt[c] = 0;
    return t;
}

__device__ __forceinline__ void select_instance_wave(int inst, int lane)
{
    const EvalKernArgs &KA = eval_kernargs();
    const DevParams &P = *KA.Pp;
    const InstDesc &D = KA.desc[inst];
    const InstState &S = KA.state[inst];
    fot_result &R = KA.out[inst];
    // The record's header starts out all zero, and so do the first n_total entries of its 15 path arrays wherever no
    // path sample lands (this wave alone writes the record: program order is enough).  Entries from n_total on are never
    // touched: the arrays' stride is FOT_MAX_NT whatever the planner's horizon, the work here is not.
    constexpr int HEADER_WORDS = (int)(offsetof(fot_result, t) / sizeof(unsigned long long));
    static_assert(offsetof(fot_result, t) % sizeof(unsigned long long) == 0, "the header is zeroed in 8-byte words");
    if (lane < HEADER_WORDS) ((unsigned long long *)&R)[lane] = 0ull;
    static_assert(HEADER_WORDS <= WAVE, "one word per lane");
    const auto zero_samples = [&](int k0) {                       // entries [k0, n_total) of every array
        for (int k = k0 + lane; k < P.n_total; k += WAVE)
#pragma unroll
            for (int f = 0; f < 15; ++f) R.t[f * FOT_MAX_NT + k] = 0.0;
    };
    if (!S.c2f_ok) {
        zero_samples(0);
        if (lane == 0) {
            R.status = FOT_PLAN_C2F_FAILED; R.best_index = -1; R.n_cand = 0; R.n_keep = 0;
            R.cost = INFINITY; R.stats_valid = 0;
            R.new_last_kappa = D.ego.last_kappa; R.new_prev_s = S.new_prev_s;
            for (int i = 0; i < 6; ++i) { R.frenet0[i] = NAN; R.ref0[i] = NAN; }
        }
        return;
    }
    // over the instance's tiles (lane = tile): histogram summed, first-minimum arg-min (lowest index wins ties)
    int cnt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    ScanBest best = { INFINITY, -1 };
    int keep_l = 0;
    for (int t = lane; t < D.n_tiles; t += WAVE) {
        const TilePart *tpp = KA.parts + D.tile0 + t;
        const double c_t = ld_agent(&tpp->cost);
        const int i_t = ld_agent(&tpp->idx), k_t = ld_agent(&tpp->keep);
#pragma unroll
        for (int c = 0; c < 8; ++c) cnt[c] += ld_agent(&tpp->cnt[c]);
        const ScanBest o = { c_t, i_t };
        const int before = best.idx;
        scan_merge(best, o);
        if (best.idx != before) keep_l = k_t;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
        for (int off = 32; off >= 1; off >>= 1) cnt[c] += __shfl_xor(cnt[c], off, WAVE);
    {
        const int mine_idx = best.idx;
        best = wave_argmin(best);
        // the keep of the winning tile: held by the lane whose own best it is
        const unsigned long long who = __ballot(mine_idx == best.idx && best.idx >= 0);
        keep_l = who ? __shfl(keep_l, __ffsll((long long)who) - 1, WAVE) : 0;
    }
    if (lane == 0) {
        R.status = best.idx >= 0 ? FOT_PLAN_OK : FOT_PLAN_NO_PATH;
        R.best_index = best.idx;
        R.n_cand = S.n_cand;
        R.cost = best.idx >= 0 ? best.dist : INFINITY;
        for (int c = 0; c < 8; ++c) R.stats[c] = cnt[c];
        R.stats_valid = 1;
        R.new_prev_s = S.new_prev_s;
        for (int i = 0; i < 6; ++i) { R.frenet0[i] = S.frenet0[i]; R.ref0[i] = S.ref0[i]; }
    }
    if (best.idx < 0) {
        zero_samples(0);
        if (lane == 0) { R.n_keep = 0; R.new_last_kappa = D.ego.last_kappa; }
        return;
    }
    const int keep = keep_l;
    const CandDecode cd = decode_candidate(P, D, S.frenet0, best.idx);
    const LonInfo L = profile_info(P, D, S.frenet0, cd.lon_slot, false);
    ComputeTab tab;
    tab.sp = KA.sp; tab.L = L; tab.dt = P.dt;
    double q[6];
    lat_coeffs(S.frenet0, cd.di, cd.brake ? P.brake[cd.ti] : P.ti[cd.ti], q);
#ifndef FOT_SEL_NO_PATH
    for (int k = lane; k < keep; k += WAVE) {                              // (more than 64 samples: two rounds)
        double o[15];
        final_sample(P, L, tab, q, k, o);
        R.t[k] = o[0]; R.s[k] = o[1]; R.s_d[k] = o[2]; R.s_dd[k] = o[3]; R.s_ddd[k] = o[4];
        R.d[k] = o[5]; R.d_d[k] = o[6]; R.d_dd[k] = o[7]; R.d_ddd[k] = o[8];
        R.x[k] = o[9]; R.y[k] = o[10]; R.yaw[k] = o[11]; R.v[k] = o[12]; R.a[k] = o[13];
        R.c[k] = o[14];
        if (k == 1) R.new_last_kappa = o[14];                              // frenet_planner.py:301-302
    }
#endif
    zero_samples(keep);
    if (lane == 0) {
        R.n_keep = keep;
        if (keep <= 1) R.new_last_kappa = D.ego.last_kappa;
    }
}

// The record of `inst` is complete: tell a host that polls for it (a synchronous small call, fot_host.cpp wait_records).
// The release is at system scope -- the record lies in pinned host memory --, the flag follows it.
__device__ __forceinline__ void record_written(int inst, int lane)
{
    const EvalKernArgs &KA = eval_kernargs();
    if (!KA.done_flag) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(KA.done_flag + inst, KA.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A wave is done with tile `tile` of instance `inst`: it leaves what it found (tp; empty for a tile without candidates)
// and counts itself.  Every tile of the instance arrives exactly once; the wave that completes the count selects.
// k_frenet_state zeroes the counters.  The partial results travel as agent-coherent stores / loads (st_agent,
// ld_agent): once this wave's stores are acknowledged (vmcnt) they are visible to every CU, so the count needs no fence.
__device__ __forceinline__ void tile_done(int inst, int tile, int lane, const TilePart &tp)
{
    const EvalKernArgs &KA = eval_kernargs();
    if (lane == 0) {
        TilePart *dst = KA.parts + KA.desc[inst].tile0 + tile;
        st_agent(&dst->cost, tp.cost); st_agent(&dst->idx, tp.idx); st_agent(&dst->keep, tp.keep);
#pragma unroll
        for (int c = 0; c < 8; ++c) st_agent(&dst->cnt[c], tp.cnt[c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores, before the count
    int last = 0;
    // (-DFOT_SAFE_SELECT: the textbook form -- an acquire-release count at agent scope, i.e. an L2 write-back before and
    //  an invalidate behind it in EVERY wave: +45 % on the launch when measured; the shipped form relies on the sc1
    //  stores / loads above and below, which scripts/isa_check_async.py verifies in the ISA of every build)
#ifdef FOT_SAFE_SELECT
    constexpr int COUNT_ORDER = __ATOMIC_ACQ_REL;
#else
    constexpr int COUNT_ORDER = __ATOMIC_RELAXED;
#endif
    if (lane == 0)
        last = __hip_atomic_fetch_add(&KA.inst_done[inst], 1, COUNT_ORDER, __HIP_MEMORY_SCOPE_AGENT)
               == KA.desc[inst].n_tiles - 1 ? 1 : 0;
    last = __builtin_amdgcn_readfirstlane(last);
    asm volatile("" ::: "memory");                                // no load of the partials may be hoisted above the count
    if (!last) return;
    select_instance_wave(inst, lane);
    record_written(inst, lane);
}

// the degenerate launch: a batch without a single tile (no horizons, no brake ladder) still gets its records
__global__ void __launch_bounds__(WAVE)
k_select_only(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc, const InstState *__restrict__ state,
              const int32_t *__restrict__ tile_cand0, const int32_t *__restrict__ tile_n,
              const TileStep *__restrict__ wave_rng, const f2 *__restrict__ ent32, const EvalKernArgs a)
{
    if ((int)blockIdx.x < a.n_inst) {
        select_instance_wave((int)blockIdx.x, (int)threadIdx.x);
        record_written((int)blockIdx.x, (int)threadIdx.x);
    }
}

// One wave per tile.  The grid deals the tiles out position-major and XCD-aligned: workgroup b serves the instances
// x, x + 8, ... with x = b mod 8 -- the XCD that, under round-robin placement, also ran k_cull's workgroups for them,
// so their lists sit in its L2 (speed only) -- and an instance's LAST tile comes first (late horizons and the brake
// ladder run longest), so the long tiles start early and the short ones fill the end of the launch.  The waves of a
// workgroup share nothing but the staged spline: each has its own slice of LDS.
__global__ void __launch_bounds__(EVAL_WG) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_evaluate(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc, const InstState *__restrict__ state,
           const int32_t *__restrict__ tile_cand0, const int32_t *__restrict__ tile_n,
           const TileStep *__restrict__ wave_rng, const f2 *__restrict__ ent32, const EvalKernArgs a)
{
    // (eval_kernargs() addresses the struct's fields in the argument segment, behind the EVAL_LEAD_PTRS pointers)
    const int waves_per_wg = (int)blockDim.x / WAVE;
    const int wave_doubles = eval_wave_doubles(a.row_budget);
#ifdef FOT_TIMELINE
    if (threadIdx.x == 0) s_tl_entry[0] = __builtin_amdgcn_s_memrealtime();
#endif
    // LDS: per wave [rows | summaries | row offsets], then the spline (shared by the workgroup's waves)
    const SplineView sp_lds = stage_spline(a.sp, a.lds_knots, s_lon + waves_per_wg * wave_doubles);
#ifdef FOT_TIMELINE
    if (threadIdx.x == 0) s_tl_entry[1] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
#endif
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const int lane = threadIdx.x & (WAVE - 1);
    double *my_rows = s_lon + wv * wave_doubles;
    const int x = (int)blockIdx.x & (N_XCD - 1), q = ((int)blockIdx.x >> 3) * waves_per_wg + wv;
    const int m_x = (a.n_inst - x + N_XCD - 1) / N_XCD;            // instances x, x + 8, ...
    if (m_x <= 0 || q >= m_x * a.max_tiles) return;
    const int pos = q / m_x, j = q - pos * m_x;
    const int inst = x + N_XCD * j;
    const int n_tiles = desc[inst].n_tiles;
    if (pos >= n_tiles) return;                                  // a shorter lattice than the batch's longest
    TilePart tp = tile_part_empty();
    evaluate_tile<TILE_WAVE>(Pp, desc, state, tile_cand0, tile_n, wave_rng, ent32, a, sp_lds, my_rows, inst,
                         n_tiles - 1 - pos, lane, x, tp);
    tile_done(inst, n_tiles - 1 - pos, lane, tp);
}

// The same for a handful of egos (fewer tiles than the GPU has SIMDs): a tile alone on its SIMD is a chain of
// ~50 dependent time steps of ~1.1 us, so the workgroup's waves (blockDim.x / 64 <= SEG_MAX) take a time segment
// each of ONE tile and its first wave merges them.  One workgroup per tile, same tile order.
__global__ void __launch_bounds__(SEG_MAX * WAVE)
k_evaluate_split(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc,
                 const InstState *__restrict__ state, const int32_t *__restrict__ tile_cand0,
                 const int32_t *__restrict__ tile_n, const TileStep *__restrict__ wave_rng,
                 const f2 *__restrict__ ent32, const EvalKernArgs a)
{
    const int n_seg = (int)blockDim.x / WAVE;
    const int wave_doubles = eval_wave_doubles(a.row_budget);
    // LDS: [rows | summaries | row offsets] of the tile, the segments' hand-over, then the spline
    double *s_part = s_lon + wave_doubles;
    const SplineView sp_lds = stage_spline(a.sp, a.lds_knots, s_part + (SEG_MAX - 1) * SEG_DOUBLES);
    const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const int lane = threadIdx.x & (WAVE - 1);
    const int x = (int)blockIdx.x & (N_XCD - 1), q = (int)blockIdx.x >> 3;
    const int m_x = (a.n_inst - x + N_XCD - 1) / N_XCD;
    if (m_x <= 0) return;
    if (q >= m_x * a.max_tiles) return;
    const int pos = q / m_x, j = q - pos * m_x;
    const int inst = x + N_XCD * j;
    const int n_tiles = desc[inst].n_tiles;
    if (pos >= n_tiles) return;
    const int tile = n_tiles - 1 - pos;
    TilePart tp = tile_part_empty();
    evaluate_tile<TILE_SPLIT>(Pp, desc, state, tile_cand0, tile_n, wave_rng, ent32, a, sp_lds, s_lon, inst, tile, lane, x,
                              tp, seg, n_seg, s_part);
    if (seg == 0) tile_done(inst, tile, lane, tp);               // (the wave that merged the segments and holds the results)
}

// The grouped cut (fot_math.hpp): one workgroup per group of GROUP_TILES tiles, one shared row table, four such
// workgroups per CU -- four waves per SIMD.  Same order as above with groups in the place of tiles: queue x holds the
// groups of the instances x, x + 8, ... position-major, an instance's last group first.
#ifndef FOT_GROUP_WAVES
#define FOT_GROUP_WAVES 4
#endif
__global__ void __launch_bounds__(GROUP_TILES * WAVE) __attribute__((amdgpu_waves_per_eu(FOT_GROUP_WAVES, FOT_GROUP_WAVES)))
k_evaluate_group(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc,
                 const InstState *__restrict__ state, const int32_t *__restrict__ tile_cand0,
                 const int32_t *__restrict__ tile_n, const TileStep *__restrict__ wave_rng,
                 const f2 *__restrict__ ent32, const EvalKernArgs a)
{
    // LDS: [rows | summaries | row offsets] of the group, then the spline
    const SplineView sp_lds = stage_spline(a.sp, a.lds_knots, s_lon + eval_group_doubles());
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const int lane = threadIdx.x & (WAVE - 1);
    const int x = (int)blockIdx.x & (N_XCD - 1), q = (int)blockIdx.x >> 3;
    const int m_x = (a.n_inst - x + N_XCD - 1) / N_XCD;
    const int n_entries = m_x * (a.max_tiles / GROUP_TILES);      // groups in this queue
    if (m_x <= 0 || q >= n_entries) return;
    const int pos = q / m_x, j = q - pos * m_x;
    const int inst = x + N_XCD * j;
    const int n_groups = desc[inst].n_tiles / GROUP_TILES;
    if (pos >= n_groups) return;                                 // a shorter lattice than the batch's longest
    const int tile0 = (n_groups - 1 - pos) * GROUP_TILES;
    TilePart tp = tile_part_empty();
    evaluate_tile<TILE_GROUP>(Pp, desc, state, tile_cand0, tile_n, wave_rng, ent32, a, sp_lds, s_lon, inst, tile0 + wv,
                              lane, x, tp, wv, GROUP_TILES, nullptr, tile0);
    tile_done(inst, tile0 + wv, lane, tp);
}

// ---------------------------------------------------------------------------
// collision broad phase: entry lists
// ---------------------------------------------------------------------------

// One workgroup of CULL_KG waves per (instance, group of CULL_KG consecutive time steps); wave w owns step k0 + w.  The caller's prediction tensor is
// [S][P][T][2]: the T samples of one pedestrian are contiguous, so a lane that takes obstacle (s, p) reads its
// CULL_KG consecutive samples as one contiguous run -- every fetched sector is used -- and classifies it against the
// boxes of the group's time steps.  Entries of step k: the static obstacles and the dynamic obstacles of time row
// min(k, T-1) that lie inside the candidates' bounding box of k (merged over the instance's longitudinal profiles)
// grown by the collision radius, ordered by bin along the longer side of the box (counting sort through LDS
// atomics), FAR32-padded to chunk pairs.  Then, per tile of the instance (k_evaluate's unit of work), the chunk range
// its own profiles' boxes can reach (strip_range) -- the only thing k_evaluate reads per time step.
// Groups of FOUR steps: 62.4 us solo against 61.4 with eight (the step-independent solves are repeated by twice as many
// workgroups), but a 256-thread workgroup with 16 KB of LDS finds room beside the evaluation's workgroups where a
// 512-thread one with 29 KB does not -- the step with four plan calls in flight is 1.5 % shorter (scripts/ab_kg.sh;
// sixteen steps: 79 us, two: 77 us).
#ifndef FOT_CULL_KG
#define FOT_CULL_KG 4
#endif
constexpr int CULL_KG = FOT_CULL_KG;    // time steps (= waves) of a group in k_cull
constexpr int CULL_LIST = 256;          // kept obstacles per time step remembered between the two passes
constexpr int CULL_PBOX = 96;           // profiles per instance whose boxes are kept in LDS (more: recomputed)
constexpr uint32_t CULL_IDX_MASK = 0xFFFFFu;   // obstacle index (< 2^20, fot_setup.hpp) | bin << 20
constexpr int CULL_VER = 256;           // pedestrian tracks a group checks for NaN (the ones inside its boxes), [S][P][T] layout
constexpr int CULL_BAD = 32;            // ... and tracks found to hold one, remembered by the group
constexpr uint32_t CULL_DEAD = 0xFFFFFFFFu;    // a remembered obstacle that turned out to be such a track

// One group: KG waves, the KG consecutive time steps from k0 of instance `inst`.  `sp`: the reference path as the caller
// staged it; `dyn_lds`: (n_ti + n_brake) * (KG * 2 + 9) doubles of dynamic LDS (launch_cull).
template <typename T, int KG>
__device__ __forceinline__ void
cull_group(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc, const InstState *__restrict__ state,
           const SplineView &sp, double *dyn_lds, const T *__restrict__ static_xy, const T *__restrict__ dyn_xy,
           int32_t *__restrict__ ent_cnt, f2 *__restrict__ ent32, d2 *__restrict__ ent64, uint8_t *__restrict__ ent_sid,
           TileStep *__restrict__ wave_rng, const int32_t *__restrict__ tile_cand0, const int32_t *__restrict__ tile_n,
           const int32_t *__restrict__ tile_span, const uint8_t *__restrict__ nan_flag, int ablate, int inst, int k0)
{
    __shared__ int s_cnt[KG][CULL_BINS + 1];                // pass 1: entries per bin; pass 2: write cursors
    __shared__ int s_start[KG][CULL_BINS + 1];
    __shared__ int s_nin[KG];
    __shared__ uint32_t s_list[KG][CULL_LIST];              // kept obstacles of each step: index | bin << 20
    __shared__ uint32_t s_ver[CULL_VER];                    // (lazy NaN check) tracks inside some box of the group
    __shared__ uint32_t s_bad[CULL_BAD];                    // ... those that hold a NaN
    __shared__ int s_nver, s_nbad;
    __shared__ Box32 s_box[KG];                             // per-step constants
    __shared__ BinMap s_bm[KG];
    __shared__ float s_margin[KG];
    __shared__ Box32 s_pbox[KG][CULL_PBOX];                 // boxes of the instance's profiles at the group's steps
    // what the boxes need of the instance that does not depend on the step, solved once per workgroup: the quartic of
    // every profile (and its horizon's index), the quintics of the two extreme lateral targets of every horizon
    __shared__ LonQuartic s_linfo[CULL_PBOX];
    __shared__ uint8_t s_lext[CULL_PBOX];
    const DevParams &P = *Pp;
    // dynamic LDS, sized by the planner's horizons (launch_cull): the lateral extent per horizon / brake-ladder entry
    // and step, and the two extreme lateral quintics of every horizon
    const int n_ext_cap = P.n_ti + P.n_brake;
    double *s_ext = dyn_lds;                                     // [KG][n_ext_cap][2]
    double *s_latq = s_ext + KG * n_ext_cap * 2;                 // [n_ext_cap][9]
    const InstDesc &D = desc[inst];
    const InstState &S = state[inst];
    if (D.ent_cap == 0) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);   // this wave's time step inside the group
    const int nk = P.n_total - k0 < KG ? P.n_total - k0 : KG;
    const int n_grid_lon = P.n_ti * D.n_tv;
    const int n_prof = S.c2f_ok ? n_grid_lon + S.n_brake : 0;
    const double sq_dyn = D.dyn_mode == FOT_DYN_SINGLE ? P.sq_r_dyn : P.sq_r;
    const double sq_max = sq_dyn > P.sq_r ? sq_dyn : P.sq_r;
    const FilterConst fc = filter_const(sq_max, sq_dyn < P.sq_r ? sq_dyn : P.sq_r);
    const float slack = box_footprint_slack(P);
    const int n_ext = P.n_ti + (S.c2f_ok ? S.n_brake : 0);
    for (int w = tid; w < n_prof && w < CULL_PBOX; w += KG * WAVE) {
        s_linfo[w] = lon_quartic(profile_info(P, D, S.frenet0, w, false));
        s_lext[w] = (uint8_t)extent_index(P, D, w);
    }
    if (S.c2f_ok && tid < n_ext) {                               // (n_ext <= FOT_MAX_TI + FOT_MAX_BRAKE < the workgroup)
        const bool brake = tid >= P.n_ti;
        lateral_extent_coeffs(P, S.frenet0, brake, brake ? P.brake[tid - P.n_ti] : P.ti[tid], s_latq + tid * 9);
    }
    __syncthreads();
    if (ablate & 16) return;                                     // (timing diagnostics: launch + per-instance constants)

    struct Raw { T x, y; };                                       // as stored: widened only when it is classified
    const bool dyn_on = D.dyn_mode != FOT_DYN_NONE;
    const int n_dyn = dyn_on ? D.S * D.P : 0;
    const int total = D.n_static + n_dyn;
    const bool tmajor = D.dyn_tmajor != 0;                       // [T][S][P][2] instead of the caller's [S][P][T][2]

    // Pass-1 lane roles.  Caller layout: lane = (obstacle slot, step) -- the KG lanes of one obstacle read its KG
    // consecutive samples, a contiguous run of the [S][P][T][2] tensor (32 bytes at KG = 4, float32), so a wave-wide load touches 64 / KG such runs
    // instead of 64 scattered cache lines; the group's 8 waves share the obstacles.  T-major layout: the obstacles of
    // one time row are contiguous, so wave w takes step k0 + w alone and its lanes read 64 consecutive obstacles.
    const int kl = tmajor ? wv : (lane & (KG - 1));          // this lane's step inside the group
    const int i_first = tmajor ? lane : wv * (WAVE / KG) + lane / KG;
    constexpr int STRIDE = WAVE;                                  // obstacles between two of a lane's loads
    const int row_l = k0 + kl < D.T - 1 ? k0 + kl : D.T - 1;
    // Branch-free: an index past the end is clamped (classify() ignores its value) and static / dynamic is a select
    // between two addresses, so the UNROLL loads of a lane are issued back to back -- behind a branch each one would
    // wait for the one before it, one gather in flight per lane, and the pass would crawl along at memory latency.
    auto fetch = [&](int i, Raw &o) {
        const int ic = i < total ? i : total - 1;                 // (total > 0 inside the loop)
        const int j = ic - D.n_static;                            // = s * P + p when dynamic
        const T *ps = static_xy + 2 * ((int64_t)D.static_off + ic);
        const T *pd = dyn_xy + 2 * (D.dyn_off + (tmajor ? (int64_t)row_l * n_dyn + j : (int64_t)j * D.T + row_l));
        const Raw *src = (const Raw *)(ic < D.n_static ? ps : pd);
        o = *src;
    };
    // Boxes of the group's time steps over all longitudinal profiles of the instance.  The (step, horizon) lateral
    // extents and then the (step, profile) boxes are spread over ALL threads of the workgroup -- full waves of float64
    // work instead of waves with a third of their lanes busy --, then wave wv merges the boxes of step k0 + wv.
    if (!(ablate & 8)) {
        for (int i = tid; i < nk * n_ext; i += KG * WAVE) {
            const int ks = i / n_ext, e = i - ks * n_ext;
            const bool brake = e >= P.n_ti;
            const int n_eval = brake ? P.brake[e - P.n_ti].n_t : P.ti[e].n_t;
            double *ext = s_ext + (ks * n_ext_cap + e) * 2;
            lateral_extent_q(s_latq + e * 9, brake, k0 + ks, n_eval, P.dt, ext[0], ext[1]);
        }
        __syncthreads();
        const int n_tab = n_prof < CULL_PBOX ? n_prof : CULL_PBOX;
        for (int i = tid; i < nk * n_tab; i += KG * WAVE) {
            const int ks = i / n_tab, w = i - ks * n_tab;
            const double *ext = s_ext + (ks * n_ext_cap + s_lext[w]) * 2;
            s_pbox[ks][w] = profile_box_from(s_linfo[w], D, sp, k0 + ks, P.dt, ext[0], ext[1]);   // read again below, per tile
        }
        __syncthreads();
    }
    {
        Box32 bw = box_empty();
        if (wv < nk && !(ablate & 8)) {
            for (int w = lane; w < n_prof; w += WAVE) {
                if (w < CULL_PBOX) {
                    box_merge(bw, s_pbox[wv][w]);
                } else {                                            // (more profiles than the table holds: on the spot)
                    const double *ext = s_ext + (wv * n_ext_cap + extent_index(P, D, w)) * 2;
                    box_merge(bw, profile_box_at(P, D, S.frenet0, sp, w, k0 + wv, ext[0], ext[1]));
                }
            }
        }
        bw.x0 = wave_min_f32(bw.x0); bw.y0 = wave_min_f32(bw.y0);
        bw.x1 = wave_max_f32(bw.x1); bw.y1 = wave_max_f32(bw.y1);
        const float mw = cull_margin(sq_max, bw) + slack;
        if (lane == 0) { s_box[wv] = bw; s_margin[wv] = mw; s_bm[wv] = bin_map(bw, mw); s_nin[wv] = 0; }
        if (lane <= CULL_BINS) s_cnt[wv][lane] = 0;
        if (tid == 0) { s_nver = 0; s_nbad = 0; }
    }
    __syncthreads();

    // Which pedestrian tracks hold a NaN (NanScan, fot_kernels.h).  Time-major tensors: the flags k_frenet_state's scan
    // blocks left.  The caller's [S][P][T][2] layout: LAZILY, here -- only a track that lies inside some box of the group
    // can become an entry, its T samples are one contiguous run (408 bytes at T = 51: one wave-wide load) and the group's
    // first pass has the KG lanes of a track side by side, so the group collects the few dozen tracks its boxes touch
    // (s_ver), its waves share them out and look through them, and nobody reads the tensor a second time just to find
    // NaNs: the whole tensor crosses HBM once per plan call instead of twice.  (Every wave looking through its OWN
    // tracks right behind its gathers, one barrier less, was 8 us slower: the slowest wave sets the pace.)
    const bool lazy = dyn_on && !tmajor && !(ablate & 128);       // (128: NanScan::eager, the flags are there)
    const Raw *tracks = (const Raw *)(dyn_xy + 2 * D.dyn_off);   // [n_dyn][T] in the caller's layout
    auto track_has_nan = [&](int j) {                            // one lane on its own (the paths that did not fit)
        bool bad = false;
        for (int t = 0; t < D.T; ++t) { const Raw p = tracks[(int64_t)j * D.T + t]; bad |= (p.x != p.x) | (p.y != p.y); }
        return bad;
    };
    auto track_bad = [&](int i) {                                // obstacle i, by whatever this instance's layout offers
        if (i < D.n_static) return false;
        return lazy ? track_has_nan(i - D.n_static) : nan_flag[D.nan_off + (i - D.n_static)] != 0;
    };

    // pass 1: histogram per step, kept obstacles remembered
    {
        const Box32 bl = s_box[kl];
        const float ml = s_margin[kl];
        const BinMap bml = s_bm[kl];
        const bool live_l = kl < nk && bl.x0 <= bl.x1 && !(ablate & 4);
        auto classify = [&](int i, const Raw &o) {
            const float fx = (float)((double)o.x - D.ego.x), fy = (float)((double)o.y - D.ego.y);
            const bool in = i < total && live_l && cull_inside(bl, ml, fx, fy);
            if (lazy) {
                // remembered, not yet counted: its track is looked through first.  The KG lanes of an obstacle are
                // neighbours (lane & (KG - 1) = step): the first of them enters the track once for the whole group
                const uint64_t m = __ballot(in && i >= D.n_static);
                if ((lane & (KG - 1)) == 0 && ((m >> lane) & ((1ull << KG) - 1ull)) != 0ull) {
                    const int v = atomicAdd(&s_nver, 1);
                    if (v < CULL_VER) s_ver[v] = (uint32_t)i;
                }
                if (in) {
                    const int j = atomicAdd(&s_nin[kl], 1);
                    if (j < CULL_LIST) s_list[kl][j] = (uint32_t)i | ((uint32_t)bin_of(bml, fx, fy) << 20);
                }
                return;
            }
            if (in) {
                // a pedestrian whose track holds a NaN anywhere is no obstacle at any step (NanScan; asked for the few
                // obstacles inside the box only)
                if (i >= D.n_static && nan_flag[D.nan_off + (i - D.n_static)]) return;
                const int bin = bin_of(bml, fx, fy);
                atomicAdd(&s_cnt[kl][bin], 1);
                const int j = atomicAdd(&s_nin[kl], 1);
                if (j < CULL_LIST) s_list[kl][j] = (uint32_t)i | ((uint32_t)bin << 20);
            }
        };
#ifndef FOT_CULL_UNROLL
#define FOT_CULL_UNROLL 4
#endif
        constexpr int UNROLL = FOT_CULL_UNROLL;                   // gathers in flight per lane
        if (ablate & 64) {                                        // (timing diagnostics: no gathers)
        } else if (D.n_static == 0) {
            // only the prediction tensor (the usual case): a lane's obstacles are a fixed number of bytes apart, so the
            // address is a running pointer instead of two 64-bit multiply-adds and a select per gather
            const char *p0 = (const char *)(dyn_xy + 2 * (D.dyn_off + (tmajor ? (int64_t)row_l * n_dyn + i_first
                                                                              : (int64_t)i_first * D.T + row_l)));
            const int64_t step = (int64_t)(tmajor ? STRIDE : STRIDE * D.T) * (int64_t)(2 * sizeof(T));
            const char *p = p0;
            for (int ib = i_first; ib < total; ib += UNROLL * STRIDE) {
                Raw o[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {                // (past the end: the lane's first obstacle again, ignored)
                    o[u] = *(const Raw *)(ib + u * STRIDE < total ? p : p0);
                    p += step;
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) classify(ib + u * STRIDE, o[u]);
            }
        } else {
            for (int ib = i_first; ib < total; ib += UNROLL * STRIDE) {
                Raw o[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) fetch(ib + u * STRIDE, o[u]);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) classify(ib + u * STRIDE, o[u]);
            }
        }
    }
    __syncthreads();
    if (lazy) {
        // the tracks inside the group's boxes, one wave-wide load each (lane = sample), VU of them in flight per wave
#ifndef FOT_CULL_VU
#define FOT_CULL_VU 4
#endif
        constexpr int VU = FOT_CULL_VU;
        const int n_ver = s_nver < CULL_VER ? s_nver : CULL_VER;
        for (int v0 = wv; v0 < n_ver; v0 += VU * KG) {
            Raw p[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int v = v0 + u * KG < n_ver ? v0 + u * KG : v0;
                p[u] = tracks[(int64_t)((int)s_ver[v] - D.n_static) * D.T + (lane < D.T ? lane : D.T - 1)];
            }
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int v = v0 + u * KG;
                if (v >= n_ver) break;
                bool bad = (p[u].x != p[u].x) | (p[u].y != p[u].y);
                for (int t = WAVE + lane; t < D.T; t += WAVE) {   // (more than 64 samples per track)
                    const Raw q = tracks[(int64_t)((int)s_ver[v] - D.n_static) * D.T + t];
                    bad |= (q.x != q.x) | (q.y != q.y);
                }
                if (__ballot(bad) != 0ull && lane == 0) {
                    const int nb = atomicAdd(&s_nbad, 1);
                    if (nb < CULL_BAD) s_bad[nb] = s_ver[v];
                }
            }
        }
        __syncthreads();
    }
    // wave wv: everything else of time step k0 + wv
    if (wv >= nk) return;
    const int kk = wv, k = k0 + wv;
    // lazy: did everything fit (the remembered obstacles of this step, the group's tracks, the NaN tracks among them)?
    const bool listed = !lazy || (s_nin[kk] <= CULL_LIST && s_nver <= CULL_VER && s_nbad <= CULL_BAD);
    const Box32 bk = s_box[kk];
    const BinMap bmk = s_bm[kk];
    const float mk = s_margin[kk];
    const bool live_k = bk.x0 <= bk.x1;
    const int row = k < D.T - 1 ? k : D.T - 1;
    auto point = [&](int i, d2 &o, int &sid) {                                      // obstacle i at step k
        sid = SID_STATIC;
        if (i < D.n_static) {
            const int64_t in = D.static_off + i;
            o.x = (double)static_xy[2 * in]; o.y = (double)static_xy[2 * in + 1];
        } else {
            const int j = i - D.n_static;
            const int64_t in = D.dyn_off + (tmajor ? (int64_t)row * n_dyn + j : (int64_t)j * D.T + row);
            o.x = (double)dyn_xy[2 * in]; o.y = (double)dyn_xy[2 * in + 1];
            sid = j / D.P;
        }
    };
    if (lazy) {
        // the histogram of this step, now that the NaN tracks are known: over the remembered obstacles, or -- something
        // did not fit -- over the whole row again, every kept track looked through by its own lane
        if (listed) {
            const int n_bad = s_nbad;
            for (int j = lane; j < s_nin[kk]; j += WAVE) {
                const uint32_t e = s_list[kk][j];
                bool bad = false;
                for (int b = 0; b < n_bad; ++b) bad |= s_bad[b] == (e & CULL_IDX_MASK);
                if (bad) s_list[kk][j] = CULL_DEAD;
                else atomicAdd(&s_cnt[kk][e >> 20], 1);
            }
        } else {
            for (int i0 = 0; i0 < total; i0 += WAVE) {
                const int i = i0 + lane;
                if (i >= total) continue;
                d2 o; int sid;
                point(i, o, sid);
                const float fx = (float)(o.x - D.ego.x), fy = (float)(o.y - D.ego.y);
                if (live_k && cull_inside(bk, mk, fx, fy) && !track_bad(i)) atomicAdd(&s_cnt[kk][bin_of(bmk, fx, fy)], 1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    {   // exclusive prefix over the bins (lane = bin; entry CULL_BINS ends with the total); cursors = starts
        const int c = lane < CULL_BINS ? s_cnt[kk][lane] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int t = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += t;
        }
        if (lane <= CULL_BINS) { s_start[kk][lane] = incl - c; s_cnt[kk][lane] = incl - c; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // this wave's LDS writes (other lanes') before the reads below
    __builtin_amdgcn_wave_barrier();
    const int64_t base = D.ent_off + (int64_t)k * D.ent_cap;
    TileStep *rng = wave_rng + (int64_t)D.tile0 * P.n_total + k;                   // + tile * n_total
    const int count = s_start[kk][CULL_BINS];
    // pass 2: scatter into the list (order inside a bin is irrelevant)
    if (lazy ? listed : count <= CULL_LIST) {                    // straight from the remembered indices
        for (int j = lane; j < ((ablate & 2) ? 0 : (lazy ? s_nin[kk] : count)); j += WAVE) {
            const uint32_t e = s_list[kk][j];
            if (e == CULL_DEAD) continue;
            const int i = (int)(e & CULL_IDX_MASK), bin = (int)(e >> 20);
            d2 o; int sid;
            point(i, o, sid);
            const int pos = atomicAdd(&s_cnt[kk][bin], 1);
            ent32_store(ent32, base + pos, (float)(o.x - D.ego.x), (float)(o.y - D.ego.y));
            ent64[base + pos] = o;
            ent_sid[base + pos] = (uint8_t)sid;
        }
    } else {                                                     // more kept than remembered: classify the row again
        for (int i0 = 0; i0 < total; i0 += WAVE) {
            const int i = i0 + lane;
            if (i >= total) continue;
            d2 o; int sid;
            point(i, o, sid);
            const float fx = (float)(o.x - D.ego.x), fy = (float)(o.y - D.ego.y);
            if (live_k && cull_inside(bk, mk, fx, fy) && !track_bad(i)) {
                const int bin = bin_of(bmk, fx, fy);
                const int pos = atomicAdd(&s_cnt[kk][bin], 1);
                ent32_store(ent32, base + pos, fx, fy);
                ent64[base + pos] = o;
                ent_sid[base + pos] = (uint8_t)sid;
            }
        }
    }
    const int padded = (count + 2 * ENT_CHUNK - 1) & ~(2 * ENT_CHUNK - 1);        // whole chunk pairs (k_evaluate)
    if (lane < padded - count) {
        d2 o; o.x = INFINITY; o.y = INFINITY;
        ent32_store(ent32, base + count + lane, FAR32, FAR32);
        ent64[base + count + lane] = o;
        ent_sid[base + count + lane] = SID_STATIC;
    }
    if (lane == 0) ent_cnt[(int64_t)inst * P.n_total + k] = padded;
    // chunk range of every tile of the instance
    for (int w = lane; w < D.n_tiles; w += WAVE) {
        TileStep r = { 0u, 0.0f, 0.0f, 0u };
        const int idx0 = tile_cand0[D.shape_off + w];
        if (live_k && idx0 < S.n_cand && tile_n[D.shape_off + w] > 0 && !(ablate & 1)) {   // (not a padding tile)
            // its profiles, from the table (a standing ego has no brake ladder: the span ends with its last profile)
            const uint32_t sp01 = (uint32_t)tile_span[D.shape_off + w];
            const int s0 = (int)(sp01 >> 16), s1 = (int)(sp01 & 0xffffu) < n_prof - 1 ? (int)(sp01 & 0xffffu) : n_prof - 1;
            Box32 wb = box_empty();
            for (int sl = s0; sl <= s1; ++sl) {
                if (sl < CULL_PBOX) { box_merge(wb, s_pbox[kk][sl]); continue; }
                const double *ext = s_ext + (kk * n_ext_cap + extent_index(P, D, sl)) * 2;
                box_merge(wb, profile_box_at(P, D, S.frenet0, sp, sl, k, ext[0], ext[1]));
            }
            const float wm = cull_margin(sq_max, wb) + slack;
            r.rng = strip_range(bmk, wb, wm, [&](int bb) { return s_start[kk][bb]; });
            box_thresholds(fc, wb, wm, r.thr, r.thr_sure);
        }
        rng[(int64_t)w * P.n_total] = r;
    }
}

// One workgroup of CULL_KG waves per (instance, group of CULL_KG consecutive time steps).
template <typename T>
__global__ void __launch_bounds__(CULL_KG * WAVE)
k_cull(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc, const InstState *__restrict__ state,
       int n_inst, SplineView sp_hbm, int lds_knots, const T *__restrict__ static_xy, const T *__restrict__ dyn_xy,
       int32_t *__restrict__ ent_cnt, f2 *__restrict__ ent32, d2 *__restrict__ ent64, uint8_t *__restrict__ ent_sid,
       TileStep *__restrict__ wave_rng, const int32_t *__restrict__ tile_cand0, const int32_t *__restrict__ tile_n,
       const int32_t *__restrict__ tile_span, const uint8_t *__restrict__ nan_flag,
       int ablate)
{
    if (ablate & 32) return;                                     // (timing diagnostics: the launch alone)
    const SplineView sp = stage_spline(sp_hbm, lds_knots);      // every wave, before any of them leaves
    const int groups = (Pp->n_total + CULL_KG - 1) / CULL_KG;
    // workgroups go round-robin over the 8 XCDs: all groups of instance i run back to back on XCD i mod 8, so the
    // runs that neighbouring groups cut out of the same cache lines of the prediction tensor meet in one L2
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int inst = xcd + 8 * (seq / groups), k0 = (seq % groups) * CULL_KG;
    if (inst >= n_inst) return;
    cull_group<T, CULL_KG>(Pp, desc, state, sp, s_spl + 9 * lds_knots, static_xy, dyn_xy, ent_cnt, ent32, ent64, ent_sid,
                           wave_rng, tile_cand0, tile_n, tile_span, nan_flag, ablate, inst, k0);
}

// ---------------------------------------------------------------------------
// one candidate's full (untruncated) arrays, for fot_debug_candidate_path
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(WAVE)
k_debug_path(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc, const InstState *__restrict__ state,
             SplineView sp, int inst, int idx,
             double *__restrict__ out /* [15][FOT_MAX_NT] */, int32_t *__restrict__ meta /* n_t, valid */)
{
    const DevParams &P = *Pp;
    const InstDesc &D = desc[inst];
    const InstState &S = state[inst];
    const int lane = threadIdx.x;
    if (!S.c2f_ok || idx < 0 || idx >= S.n_cand) { if (lane == 0) { meta[0] = 0; meta[1] = 0; } return; }
    const CandDecode cd = decode_candidate(P, D, S.frenet0, idx);
    const LonInfo L = profile_info(P, D, S.frenet0, cd.lon_slot, false);
    ComputeTab tab;
    tab.sp = sp; tab.L = L; tab.dt = P.dt;
    double q[6];
    lat_coeffs(S.frenet0, cd.di, cd.brake ? P.brake[cd.ti] : P.ti[cd.ti], q);
    for (int k = lane; k < L.n_t; k += WAVE) {
        double o[15];
        final_sample(P, L, tab, q, k, o);
        for (int f = 0; f < 15; ++f) out[f * FOT_MAX_NT + k] = o[f];
    }
    if (lane == 0) { meta[0] = L.n_t; meta[1] = 1; }
}

// ---------------------------------------------------------------------------
// epsilon-band report of one instance of the last plan call (fot_debug_margins): one thread per candidate
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(WAVE)
k_debug_margins(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc, const InstState *__restrict__ state,
                SplineView sp, int inst, const int32_t *__restrict__ ent_cnt, const d2 *__restrict__ ent64,
                const uint8_t *__restrict__ ent_sid, int cap, double *__restrict__ out /* [cap][MARGIN_CATS] */)
{
    const DevParams &P = *Pp;
    const InstDesc &D = desc[inst];
    const InstState &S = state[inst];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (!S.c2f_ok || idx >= S.n_cand || idx >= cap) return;
    const CandDecode cd = decode_candidate(P, D, S.frenet0, idx);
    const LonInfo L = profile_info(P, D, S.frenet0, cd.lon_slot, false);
    ComputeTab tab;
    tab.sp = sp; tab.L = L; tab.dt = P.dt;
    double q[6];
    lat_coeffs(S.frenet0, cd.di, cd.brake ? P.brake[cd.ti] : P.ti[cd.ti], q);
    MarginEntries ent;
    ent.cnt = D.ent_cap != 0 ? ent_cnt + (int64_t)inst * P.n_total : nullptr;
    ent.e64 = ent64 + D.ent_off; ent.sid = ent_sid + D.ent_off; ent.ent_cap = D.ent_cap;
    double m[MARGIN_CATS];
    candidate_margins(P, D, L, tab, q, ent, m);
    for (int c = 0; c < MARGIN_CATS; ++c) out[(int64_t)idx * MARGIN_CATS + c] = m[c];
}

// ---------------------------------------------------------------------------
// wire form of the records (fot_pack_records_device): one wave per record
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(WAVE)
k_pack_wire(int n, int n_total, int stride, const fot_result *__restrict__ src, unsigned char *__restrict__ dst)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const fot_result &R = src[i];
    unsigned char *w = dst + (int64_t)i * stride;
    fot_wire_header *H = (fot_wire_header *)w;
    if (lane == 0) {
        H->status = R.status; H->best_index = R.best_index; H->n_cand = R.n_cand; H->n_keep = R.n_keep;
        H->cost = R.cost; H->stats_valid = R.stats_valid; H->n_total = n_total;
        H->new_last_kappa = R.new_last_kappa; H->new_prev_s = R.new_prev_s;
    }
    if (lane < 8) H->stats[lane] = R.stats[lane];
    if (lane < 6) { H->frenet0[lane] = R.frenet0[lane]; H->ref0[lane] = R.ref0[lane]; }
    float *path = (float *)(w + sizeof(fot_wire_header));
    const double *arr = R.t;                                  // the 15 arrays are contiguous, FOT_MAX_NT doubles each
    // s, x and y travel as offsets from the record's own start state (s0, reference point): float32 then resolves the
    // path to 2^-24 of its LENGTH (4e-6 m at 70 m), wherever in the world it lies; samples past n_keep are zero
    const int keep = R.n_keep;
    for (int f = 0; f < 15; ++f) {
        const double base = f == 1 ? R.frenet0[0] : f == 9 ? R.ref0[1] : f == 10 ? R.ref0[2] : 0.0;
        for (int k = lane; k < n_total; k += WAVE)
            path[f * n_total + k] = k < keep ? (float)(arr[f * FOT_MAX_NT + k] - base) : 0.0f;
    }
    // padding bytes stay as they are (never read)
}

// ---------------------------------------------------------------------------
// spline evaluation (fot_spline_eval)
// ---------------------------------------------------------------------------

__global__ void k_spline_eval(SplineView sp, int n, const double *__restrict__ s, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SplinePt p;
    spline_point(sp, s[i], p);
    double cr, sr, kappa, dkappa;
    spline_frame(p, cr, sr, kappa, dkappa);
    out[0 * (int64_t)n + i] = p.x;
    out[1 * (int64_t)n + i] = p.y;
    out[2 * (int64_t)n + i] = atan2(p.dy, p.dx);
    out[3 * (int64_t)n + i] = kappa;
    out[4 * (int64_t)n + i] = dkappa;
}

// external paths (fot_check_collision_paths / fot_check_paths): one lane per path, arrays [n_paths][FOT_MAX_NT]
// in the caller's row-major layout, obstacles straight from the caller's layout (exact float64 test)
struct PathArraySource {
    const double *x, *y, *yaw, *t;      // this path's rows
    double dt;
    const double *circ_off;
    int has_footprint;
    __device__ __forceinline__ int tindex(int k) const { return (int)nearbyint(t[k] / dt); }
    __device__ __forceinline__ void get(int k, int ci, double &px, double &py) const
    {
        if (has_footprint) {                 // frenet_planner.py:1162-1167
            px = x[k] + circ_off[ci] * cos(yaw[k]);
            py = y[k] + circ_off[ci] * sin(yaw[k]);
        } else {
            px = x[k]; py = y[k];
        }
    }
};

// mode 0: collision only (status_out = 1 free / 0 hit); mode 1: _check_paths categories (FOT_ST_*)
__global__ void k_check_ext(const DevParams *__restrict__ Pp, const InstDesc *__restrict__ desc, int n_paths, int mode,
                            const int32_t *__restrict__ len, const int32_t *__restrict__ flags,
                            const double *__restrict__ arrays /* [9][n_paths][FOT_MAX_NT]: x y yaw v a c d s t */,
                            const double *__restrict__ static_xy, const double *__restrict__ dyn_xy,
                            int32_t *__restrict__ status_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_paths) return;
    const DevParams &P = *Pp;
    const InstDesc &D = desc[0];
    const int64_t plane = (int64_t)n_paths * FOT_MAX_NT;
    const double *row = arrays + (int64_t)i * FOT_MAX_NT;
    const double *ax = row, *ay = row + plane, *ayaw = row + 2 * plane, *av = row + 3 * plane, *aa = row + 4 * plane,
                 *ac = row + 5 * plane, *ad = row + 6 * plane, *as = row + 7 * plane, *at = row + 8 * plane;
    const int n = len[i];
    PathArraySource src;
    src.x = ax; src.y = ay; src.yaw = ayaw; src.t = at; src.dt = P.dt;
    src.circ_off = P.circ_off; src.has_footprint = P.has_footprint;
    ObstacleView obs;
    obs.stat = static_xy; obs.dyn = dyn_xy; obs.dtype = FOT_F64;
    if (mode == 0) {
        const bool hit = n > 0 && collide_candidate(P, D, obs, n, src);
        status_out[i] = hit ? 0 : 1;
        return;
    }
    const bool has_geo = flags[i] & 1, has_d = flags[i] & 2;
    CheckAcc acc;
    check_init(acc);
    const LoopConst lc = loop_const(P, D);
    for (int k = 0; k < n; ++k) {
        PathSample ps;
        ps.x = ax[k]; ps.y = ay[k]; ps.cos_t = cos(ayaw[k]); ps.sin_t = sin(ayaw[k]);
        ps.kappa = ac[k]; ps.v = av[k]; ps.a = aa[k]; ps.d = ad[k];
        check_sample(lc, acc, k, ps, has_geo, has_d, [&] { return fabs(as[k] - as[k - 1]); });
    }
    int st = check_status(D, acc, n);
    if (st == ST_PENDING && n > 0 && collide_candidate(P, D, obs, n, src)) st = FOT_ST_COLLISION;
    if (n > 0) st = final_status(st, av[n - 1], as[n - 1] - as[0], D.max_stop);
    status_out[i] = st;
}

// ---------------------------------------------------------------------------
// prediction resampling -> obstacle tensor
// ---------------------------------------------------------------------------

struct ResampleArgs {
    double sgan_dt, sim_dt, staleness;
    int S, pred_len, P, n_dense, T;          // T = n_dense + prepend
    int has_anchor, prepend, cv;             // cv: sources are (obs_prev, obs_last) -> constant velocity; 2: float32 obs
    int tmajor;                              // out[k][s][p][axis] (FOT_OUT_TMAJOR) instead of out[s][p][k][axis]
    // per-episode blocks (the closed loop's distribution frame): pedestrian p belongs to episode ped_ep[p], whose
    // pedestrians are [ep_ped0[e], ep_ped0[e + 1]) and whose [S][P_e][T][2] block starts at point ep_blk[e]; nullptr: one
    // [S][P][T][2] tensor
    const int32_t *ped_ep; const int32_t *ep_ped0; const int64_t *ep_blk;
};

// one thread per (sample, pedestrian, axis); out[s][p][k][axis]
template <typename TI, typename TO>
__global__ void k_resample(ResampleArgs A, const TI *__restrict__ pred, const double *__restrict__ anchor,
                           const double *__restrict__ current, TO *__restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= A.S * A.P * 2) return;
    const int ax = idx & 1, sp = idx >> 1, p = sp % A.P, smp = sp / A.P;
    // element (row k) of this thread's (sample, pedestrian, axis): rows are T apart in the reference's layout and a whole
    // S x P plane apart in the T-major one (where neighbouring threads then write neighbouring addresses)
    const int64_t k_stride = A.tmajor ? (int64_t)A.S * A.P * 2 : 2;
    TO *dst = out + (A.tmajor ? (int64_t)sp * 2 : (int64_t)sp * A.T * 2) + ax;
    if (A.ped_ep) {
        const int e = A.ped_ep[p], p0 = A.ep_ped0[e], P_e = A.ep_ped0[e + 1] - p0;
        dst = out + 2 * (A.ep_blk[e] + ((int64_t)smp * P_e + (p - p0)) * A.T) + ax;
    }
    if (A.prepend) dst[0] = (TO)current[2 * p + ax];
    dst += k_stride * A.prepend;
    if (A.cv) {                                                     // predict_cv (:188-231)
        const double cur = anchor[2 * p + ax];                      // obs_last
        double vel = 0.0;
        if (pred && A.cv == 2)                                      // float32 observations: float32 velocity (:216)
            vel = (double)(((float)cur - (float)pred[2 * p + ax]) / (float)A.sgan_dt);
        else if (pred)
            vel = (cur - (double)pred[2 * p + ax]) / A.sgan_dt;     // obs_prev
        for (int i = 0; i < A.n_dense; ++i) {
            const double t = (A.sim_dt + (double)i * A.sim_dt) + A.staleness;
            dst[k_stride * i] = (TO)(cur + vel * t);
        }
        return;
    }
    ResampleAxis R;
    R.sgan_dt = A.sgan_dt; R.staleness = A.staleness;
    R.first_k = A.has_anchor ? 0 : 1;
    int n = 0;
    if (A.has_anchor) R.co[n++] = anchor[2 * p + ax];
    for (int k = 0; k < A.pred_len; ++k) R.co[n++] = (double)pred[(((int64_t)smp * A.pred_len + k) * A.P + p) * 2 + ax];
    R.n_src = n;
    const bool constant = R.all_close(R.co[0]) || R.all_close(0.0);
    const double v_tail = R.tail_velocity();
    for (int i = 0; i < A.n_dense; ++i)
        dst[k_stride * i] = (TO)R.at(A.sim_dt + (double)i * A.sim_dt, constant, v_tail);
}

// block = sample: sum over (p, k) of the distance to the sample mean (predict_single_best :346-350)
template <typename TO>
__global__ void __launch_bounds__(256)
k_sample_dist(int S, int P, int T, int skip, int tmajor, const TO *__restrict__ out, double *__restrict__ dist)
{
    auto at = [&](int q, int p, int k) {
        return out + (tmajor ? ((int64_t)k * S + q) * P + p : ((int64_t)q * P + p) * T + k) * 2;
    };
    const int smp = blockIdx.x;
    const int n = P * (T - skip);
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int p = i / (T - skip), k = i - p * (T - skip) + skip;  // the prepended current position is not part of it
        double mx = 0.0, my = 0.0;
        for (int q = 0; q < S; ++q) {
            const TO *e = at(q, p, k);
            mx += (double)e[0]; my += (double)e[1];
        }
        mx /= (double)S; my /= (double)S;
        const TO *e = at(smp, p, k);
        const double dx = (double)e[0] - mx, dy = (double)e[1] - my;
        acc += sqrt(dx * dx + dy * dy);
    }
    __shared__ double part[256 / WAVE];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) part[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x / WAVE); ++w) t += part[w];
        dist[smp] = t;
    }
}

// ---------------------------------------------------------------------------
// safety metrics, one wave per ego, lanes over (footprint circle, pedestrian) pairs
// ---------------------------------------------------------------------------

__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off, WAVE));
    return v;
}

__global__ void __launch_bounds__(WAVE)
k_safety(const DevParams *__restrict__ Pp, int n, const double *__restrict__ ego, const int32_t *__restrict__ ped_off,
         const double *__restrict__ ped_pos, const double *__restrict__ ped_vel, double ego_radius, double ped_radius,
         double footprint_radius, int use_fp, fot_safety *__restrict__ out)
{
    const int e = blockIdx.x;
    if (e >= n) return;
    const DevParams &P = *Pp;
    const double x = ego[4 * e], y = ego[4 * e + 1], yaw = ego[4 * e + 2], v = ego[4 * e + 3];
    const double hx = cos(yaw), hy = sin(yaw);
    const bool fp = use_fp && P.has_footprint;
    const int nc = fp ? P.n_circ : 1;
    const double combined = (fp ? footprint_radius : ego_radius) + ped_radius;
    const int p0 = ped_off[e], np_ = ped_off[e + 1] - p0;
    double min_d = INFINITY, ttc = INFINITY, ahead = INFINITY;
    for (int i = threadIdx.x; i < nc * np_; i += WAVE) {
        const int c = i / np_, p = i - c * np_;
        const double off = fp ? P.circ_off[c] : 0.0;
        const double cx = x + off * hx, cy = y + off * hy;                      // footprint.py:42-45
        const double px = ped_pos[2 * (p0 + p)], py = ped_pos[2 * (p0 + p) + 1];
        const double rx = px - cx, ry = py - cy;
        const double dist = sqrt(rx * rx + ry * ry);
        min_d = fmin(min_d, dist);
        const double vx = ped_vel[2 * (p0 + p)] - v * hx, vy = ped_vel[2 * (p0 + p) + 1] - v * hy;
        const double along = -(rx * vx + ry * vy) / (dist + 1e-8);
        if (along > 1e-5) {
            const double t = (dist - combined) / along;
            if (t >= 0.0) ttc = fmin(ttc, t);
        }
        if ((px - x) * hx + (py - y) * hy > 0.0) ahead = fmin(ahead, dist);      // strictly ahead of the vehicle centre
    }
    min_d = wave_min_f64(min_d); ttc = wave_min_f64(ttc); ahead = wave_min_f64(ahead);
    if (threadIdx.x == 0) {
        fot_safety r;
        r.min_distance = min_d; r.ttc = ttc; r.clearance = min_d - combined;
        r.clearance_ahead = isinf(ahead) ? INFINITY : ahead - combined;
        r.collision = min_d < combined ? 1 : 0; r._pad = 0;
        out[e] = r;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

#define FOT_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

int launch_frenet_state(const DevParams *P, SplineView sp, const InstDesc *desc, InstState *state, int n_inst,
                        MetaImport imp, NanScan scan, int32_t *inst_done, hipStream_t st)
{
    if (n_inst <= 0) return 0;
    const int lds_knots = sp.n <= SPLINE_LDS_KNOTS ? sp.n : 0;
    const int64_t grid = (int64_t)n_inst * (1 + (scan.flag ? scan.blocks_per_inst : 0));
    if (grid > 0x7fffffffLL) return (int)hipErrorInvalidConfiguration;
    k_frenet_state<<<(unsigned)grid, FRENET_WG, sizeof(double) * 9 * (size_t)lds_knots, st>>>(P, sp, lds_knots, desc, state,
                                                                                         n_inst, imp, scan, inst_done);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_cull(const DevParams *P, const InstDesc *desc, const InstState *state, int n_inst, int n_total, int n_ext,
                SplineView sp, const void *static_xy, const void *dyn_xy, int dtype, EntryArrays e, TileTable tiles,
                hipStream_t st)
{
    if (n_inst <= 0 || n_total <= 0) return 0;
    const unsigned grid = (unsigned)((int64_t)((n_inst + 7) / 8 * 8) * ((n_total + CULL_KG - 1) / CULL_KG));
    static const int ablate_env = getenv("FOT_CULL_ABLATE") ? atoi(getenv("FOT_CULL_ABLATE")) : 0;   // timing diagnostics
    const int ablate = (ablate_env & ~128) | (e.eager_nan ? 128 : 0);
    const int lds_knots = sp.n <= 64 ? sp.n : 0;                           // a short spline rides along in LDS
    // + per horizon / brake-ladder entry: lateral extents of the group's steps, the two extreme lateral quintics
    const size_t lds = sizeof(double) * (9 * (size_t)lds_knots + (size_t)n_ext * (CULL_KG * 2 + 9));
    if (dtype == FOT_F32)
        k_cull<float><<<grid, CULL_KG * WAVE, lds, st>>>(P, desc, state, n_inst, sp, lds_knots, (const float *)static_xy,
                                             (const float *)dyn_xy, e.cnt, e.e32, e.e64, e.sid, e.rng, tiles.cand0, tiles.n, tiles.span,
                                             e.nan_flag, ablate);
    else
        k_cull<double><<<grid, CULL_KG * WAVE, lds, st>>>(P, desc, state, n_inst, sp, lds_knots, (const double *)static_xy,
                                              (const double *)dyn_xy, e.cnt, e.e32, e.e64, e.sid, e.rng, tiles.cand0,
                                              tiles.n, tiles.span, e.nan_flag, ablate);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_evaluate(const DevParams *P, SplineView sp, const InstDesc *desc, const InstState *state, int n_total,
                    int n_inst, TileTable tiles, EntryArrays e, CandArrays c, fot_result *out, int32_t *inst_done,
                    hipStream_t st)
{
    if (n_inst <= 0) return 0;
    static const int ablate = getenv("FOT_EVAL_ABLATE") ? atoi(getenv("FOT_EVAL_ABLATE")) : 0;
    const int lds_knots = sp.n <= 28 ? sp.n : 0;                           // a short spline rides along (2 KB at most)
    // Three launch shapes, 8 queues each (workgroup b serves the instances b mod 8):
    //  * a handful of egos: every tile one workgroup, cut into time segments (k_evaluate_split);
    //  * the grouped cut: one workgroup per group of four tiles (k_evaluate_group, four waves per SIMD);
    //  * the per-wave cut: four independent tiles per workgroup (k_evaluate, three waves per SIMD).
    int n_seg = tiles.n_tiles <= 2304 ? SEG_MAX : 1;       // (scripts/segment_sweep.py: four segments win up to ~64 egos)
    if (tiles.eval_segments >= 1 && tiles.eval_segments <= SEG_MAX) n_seg = tiles.eval_segments;
    static const int force_wpw = getenv("FOT_EVAL_WPW") ? atoi(getenv("FOT_EVAL_WPW")) : 0;      // diagnostics
    int wpw = n_seg > 1 ? 1 : tiles.n_tiles >= 1024 ? EVAL_WG / WAVE : 1;
    if (n_seg == 1 && force_wpw >= 1 && force_wpw <= EVAL_WG / WAVE) wpw = force_wpw;
    const int64_t per_queue = (int64_t)((n_inst + N_XCD - 1) / N_XCD) * tiles.max_tiles;
    const int64_t n_blocks = (per_queue + wpw - 1) / wpw * N_XCD;
    if (n_blocks > 0x7fffffffLL) return (int)hipErrorInvalidConfiguration;
    const size_t lds = sizeof(double) * ((size_t)eval_wave_doubles(tiles.row_budget) * wpw + 9 * (size_t)lds_knots
                                         + (n_seg > 1 ? (size_t)(SEG_MAX - 1) * SEG_DOUBLES : 0));
    EvalKernArgs a;
    a.Pp = P; a.sp = sp; a.desc = desc; a.state = state;
    a.row_budget = tiles.row_budget; a.lds_knots = lds_knots; a.ablate = ablate;
    a.n_inst = n_inst; a.max_tiles = tiles.max_tiles;
    a.tile_cand0 = tiles.cand0; a.tile_n = tiles.n;
    a.wave_rng = e.rng; a.ent32 = e.e32; a.ent64 = e.e64; a.ent_sid = e.sid;
    a.cand_cost = c.cost; a.cand_status = c.status; a.cand_keep = c.keep; a.parts = c.parts;
    a.out = out; a.inst_done = inst_done; a.done_flag = c.done_flag; a.done_seq = c.done_seq;
    if (tiles.n_tiles <= 0) {
        k_select_only<<<(unsigned)n_inst, WAVE, 0, st>>>(P, desc, state, tiles.cand0, tiles.n, e.rng, e.e32, a);
    } else if (n_seg > 1) {
        k_evaluate_split<<<(unsigned)n_blocks, n_seg * WAVE, lds, st>>>(P, desc, state, tiles.cand0, tiles.n, e.rng,
                                                                        e.e32, a);
    } else if (tiles.grouped) {                                            // one workgroup per group of tiles
        const int per_q = (n_inst + N_XCD - 1) / N_XCD * (tiles.max_tiles / GROUP_TILES);
        const size_t g_lds = sizeof(double) * ((size_t)eval_group_doubles() + 9 * (size_t)lds_knots);
        k_evaluate_group<<<(unsigned)(per_q * N_XCD), GROUP_TILES * WAVE, g_lds, st>>>(P, desc, state, tiles.cand0,
                                                                                      tiles.n, e.rng, e.e32, a);
    } else {
        k_evaluate<<<(unsigned)n_blocks, wpw * WAVE, lds, st>>>(P, desc, state, tiles.cand0, tiles.n, e.rng, e.e32, a);
    }
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_debug_path(const DevParams *P, const InstDesc *desc, const InstState *state,
                      SplineView sp, int inst, int idx, double *out, int32_t *meta, hipStream_t st)
{
    k_debug_path<<<1, WAVE, 0, st>>>(P, desc, state, sp, inst, idx, out, meta);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_debug_margins(const DevParams *P, const InstDesc *desc, const InstState *state, SplineView sp, int inst,
                         EntryArrays e, int cap, double *out, hipStream_t st)
{
    if (cap <= 0) return 0;
    k_debug_margins<<<(cap + WAVE - 1) / WAVE, WAVE, 0, st>>>(P, desc, state, sp, inst, e.cnt, e.e64, e.sid, cap, out);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_pack_wire(int n, int n_total, int stride, const fot_result *src, unsigned char *dst, hipStream_t st)
{
    if (n <= 0) return 0;
    k_pack_wire<<<n, WAVE, 0, st>>>(n, n_total, stride, src, dst);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_spline_eval(SplineView sp, int n, const double *s, double *out, hipStream_t st)
{
    if (n <= 0) return 0;
    k_spline_eval<<<(n + 255) / 256, 256, 0, st>>>(sp, n, s, out);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_resample(double sgan_dt, double sim_dt, double staleness, int S, int pred_len, int P, int n_dense,
                    int has_anchor, int prepend, int cv, const void *pred, int pred_dtype, const double *anchor,
                    const double *current, void *out, int out_dtype, int tmajor, hipStream_t st,
                    const int32_t *ped_ep, const int32_t *ep_ped0, const int64_t *ep_blk)
{
    const int total = S * P * 2;
    if (total <= 0) return 0;
    ResampleArgs A;
    A.sgan_dt = sgan_dt; A.sim_dt = sim_dt; A.staleness = staleness;
    A.S = S; A.pred_len = pred_len; A.P = P; A.n_dense = n_dense; A.T = n_dense + (prepend ? 1 : 0);
    A.has_anchor = has_anchor; A.prepend = prepend; A.cv = cv; A.tmajor = tmajor;
    A.ped_ep = ped_ep; A.ep_ped0 = ep_ped0; A.ep_blk = ep_blk;
    if (ped_ep && tmajor) return (int)hipErrorInvalidValue;
    const int bs = 128, grid = (total + bs - 1) / bs;
    if (pred_dtype == FOT_F32 && out_dtype == FOT_F32)
        k_resample<float, float><<<grid, bs, 0, st>>>(A, (const float *)pred, anchor, current, (float *)out);
    else if (pred_dtype == FOT_F32)
        k_resample<float, double><<<grid, bs, 0, st>>>(A, (const float *)pred, anchor, current, (double *)out);
    else if (out_dtype == FOT_F32)
        k_resample<double, float><<<grid, bs, 0, st>>>(A, (const double *)pred, anchor, current, (float *)out);
    else
        k_resample<double, double><<<grid, bs, 0, st>>>(A, (const double *)pred, anchor, current, (double *)out);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_sample_dist(int S, int P, int T, int skip, const void *out, int out_dtype, int tmajor, double *dist,
                       hipStream_t st)
{
    if (S <= 0) return 0;
    if (out_dtype == FOT_F32) k_sample_dist<float><<<S, 256, 0, st>>>(S, P, T, skip, tmajor, (const float *)out, dist);
    else k_sample_dist<double><<<S, 256, 0, st>>>(S, P, T, skip, tmajor, (const double *)out, dist);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_safety(const DevParams *P, int n, const double *ego, const int32_t *ped_off, const double *ped_pos,
                  const double *ped_vel, double ego_radius, double ped_radius, double footprint_radius, int use_fp,
                  fot_safety *out, hipStream_t st)
{
    if (n <= 0) return 0;
    k_safety<<<n, WAVE, 0, st>>>(P, n, ego, ped_off, ped_pos, ped_vel, ego_radius, ped_radius, footprint_radius, use_fp, out);
    FOT_LAUNCH_CHECK();
    return 0;
}

int launch_check_ext(const DevParams *P, const InstDesc *desc, int n_paths, int mode, const int32_t *len,
                     const int32_t *flags, const double *arrays, const double *static_xy, const double *dyn_xy,
                     int32_t *status_out, hipStream_t st)
{
    if (n_paths <= 0) return 0;
    k_check_ext<<<(n_paths + 63) / 64, 64, 0, st>>>(P, desc, n_paths, mode, len, flags, arrays, static_xy, dyn_xy,
                                                    status_out);
    FOT_LAUNCH_CHECK();
    return 0;
}

}  // namespace fot
