// fot_emu.cpp -- TEST-ONLY host emulation of the kernel pipeline.
//
// Runs the SAME arithmetic headers the gfx950 kernels are built from
// (csrc/fot_math.hpp, csrc/fot_setup.hpp) with plain loops in place of the
// launch grid, so that the planner logic can be checked against the oracle in
// the build container, where there is no GPU.  It is compiled by
// tests/test_emu_logic.py into tests/emu/_build/ and is never loaded by the
// product (integrated_path_planning_amd/), which only ever drives libfot.so.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../integrated_path_planning_amd/csrc/fot_math.hpp"
#include "../../integrated_path_planning_amd/csrc/fot_setup.hpp"

using namespace fot;

namespace {
// records the collision points of a candidate; never reports a collision (kinematic status only)
struct VecSink {
    std::vector<d2> *pts; int n_total;
    void row_begin(int) {}
    void row_end(int) {}
    void put(int k, int ci, double x, double y, bool) { d2 v; v.x = x; v.y = y; (*pts)[(size_t)ci * n_total + k] = v; }
    bool collided() const { return false; }
};
struct VecSource {
    const d2 *pts; int n_total;
    void get(int k, int ci, double &x, double &y) const { const d2 &v = pts[(size_t)ci * n_total + k]; x = v.x; y = v.y; }
    int tindex(int k) const { return k; }
};
}  // namespace

extern "C" int emu_plan_batch(const fot_params *params, int n_knots, const double *wx, const double *wy,
                              const fot_batch *b, fot_result *out,
                              int cand_cap, double *cand_cost, int32_t *cand_status, int32_t *cand_keep, char *errbuf)
{
    std::string err;
    DevParams P;
    int rc = build_dev_params(*params, P, err);
    HostSpline hs;
    if (rc == FOT_OK) rc = build_spline(n_knots, wx, wy, hs, err);
    BatchLayout L;
    TileShapes shapes;
    if (rc == FOT_OK) build_tile_shapes(P, shapes);
    if (rc == FOT_OK) rc = build_batch_layout(*params, P, shapes, *b, L, err);
    if (rc != FOT_OK) { if (errbuf) std::strncpy(errbuf, err.c_str(), 255); return rc; }
    SplineView sp;
    sp.n = hs.n; sp.s = hs.s.data();
    sp.ax = hs.ax.data(); sp.bx = hs.bx.data(); sp.cx = hs.cx.data(); sp.dx = hs.dx.data();
    sp.ay = hs.ay.data(); sp.by = hs.by.data(); sp.cy = hs.cy.data(); sp.dy = hs.dy.data();

    std::memset(out, 0, sizeof(fot_result) * (size_t)L.n_inst);
    std::vector<LonInfo> lon_info((size_t)L.n_lon + 1);
    std::vector<double> lon_tab((size_t)(L.n_lon + 1) * LON_FIELDS * FOT_MAX_NT, 0.0);
    for (int inst = 0; inst < L.n_inst; ++inst) {
        const InstDesc &D = L.desc[inst];
        fot_result &R = out[inst];
        // --- k_frenet_state (nlanes = 1)
        InstState S;
        const double x = D.ego.x, y = D.ego.y, s_end = sp.s[sp.n - 1];
        double best_s = 0.0;
        bool need_global = true;
        const bool chained = D.ego.has_prev_s == FOT_PREV_S_CHAINED;
        const double prev_s_in = chained ? out[inst - 1].new_prev_s : D.ego.prev_s;
        const bool given = D.ego.has_prev_s == FOT_EGO_IS_FRENET;
        if (given) need_global = false;
        if (D.ego.has_prev_s && !given) {
            const double s_min = fmax(0.0, prev_s_in - 10.0), s_max = fmin(s_end, prev_s_in + 10.0);
            ScanBest bb = scan_samples(sp, x, y, s_min, s_max, 100, 0, 1, false);
            best_s = bb.idx >= 0 ? linspace_at(s_min, s_max, 100, bb.idx) : 0.0;
            need_global = (fabs(best_s - s_min) < 1e-3 && s_min > 0.0) || (fabs(best_s - s_max) < 1e-3 && s_max < s_end);
        }
        const int n_glob = global_search_count(sp);
        if (need_global) {
            ScanBest bb = scan_samples(sp, x, y, 0.0, s_end, n_glob, 0, 1, true);
            best_s = linspace_at(0.0, s_end, n_glob, bb.idx >= 0 ? bb.idx : 0);
        }
        if (!given) best_s = refine_nearest(sp, x, y, best_s);
        S.new_prev_s = given ? NAN : best_s;
        bool ok = given ? frenet_state_given(sp, D.ego, S.frenet0, S.ref0) : frenet_state_at(sp, D.ego, best_s, S.frenet0, S.ref0);
        S.c2f_ok = ok;
        S.n_brake = (ok && S.frenet0[1] > 0.1) ? P.n_brake : 0;
        S.n_cand = ok ? D.n_grid + S.n_brake : 0;
        if (!ok) {
            R.status = FOT_PLAN_C2F_FAILED; R.best_index = -1; R.cost = INFINITY;
            R.new_last_kappa = D.ego.last_kappa; R.new_prev_s = S.new_prev_s;
            continue;
        }
        // --- k_lon_table
        const int n_grid_lon = P.n_ti * D.n_tv;
        std::vector<Box32> boxes((size_t)P.n_total, box_empty());           // k_lon_table: per profile, merged by k_cull
        std::vector<Box32> pbox((size_t)(n_grid_lon + S.n_brake + 1) * P.n_total, box_empty());
        for (int slot = 0; slot < n_grid_lon + S.n_brake; ++slot) {
            LonInfo Li;
            if (slot < n_grid_lon) {
                const int ti = slot / D.n_tv, itv = slot - ti * D.n_tv;
                lon_coeffs(S.frenet0, tv_value(P, D, itv), P.ti[ti], Li);
                Li.n_t = P.ti[ti].n_t; Li.n_eval = Li.n_t;
            } else {
                const TimeInfo &tb = P.brake[slot - n_grid_lon];
                lon_coeffs(S.frenet0, 0.0, tb, Li);
                Li.n_t = P.n_total; Li.n_eval = tb.n_t;
            }
            double *tab = lon_tab.data() + (size_t)(D.lon_off + slot) * (LON_FIELDS * FOT_MAX_NT);
            double js = 0.0, sd_last = 0.0;
            const bool brake = slot >= n_grid_lon;
            const TimeInfo &lat_ti = brake ? P.brake[slot - n_grid_lon] : P.ti[slot / D.n_tv];
            for (int k = 0; k < Li.n_t; ++k) {
                LonSample ls; double sddd;
                make_lon_sample(sp, Li, k, P.dt, ls, sddd);
                {   // the box k_cull derives (extent per horizon + position / tangent only) ...
                    double d0, d1;
                    lateral_extent(P, S.frenet0, brake, lat_ti, k, Li.n_eval, d0, d1);
                    pbox[(size_t)slot * P.n_total + k] = profile_box_at(P, D, S.frenet0, sp, slot, k, d0, d1);
                    {   // k_cull's form: quartic and extreme lateral quintics solved once per instance, kept in LDS
                        double q9[9], e0, e1;
                        lateral_extent_coeffs(P, S.frenet0, brake, lat_ti, q9);
                        lateral_extent_q(q9, brake, k, Li.n_eval, P.dt, e0, e1);
                        const Box32 viaq = profile_box_from(lon_quartic(Li), D, sp, k, P.dt, e0, e1);
                        if (std::memcmp(&e0, &d0, sizeof(double)) || std::memcmp(&e1, &d1, sizeof(double)) ||
                            std::memcmp(&viaq, &pbox[(size_t)slot * P.n_total + k], sizeof(Box32))) return -114;
                    }
                    // ... agrees with the one built from the full row
                    const Box32 ref = profile_box(P, S.frenet0, brake, lat_ti, ls, k, Li.n_eval, D.ego.x, D.ego.y);
                    const Box32 &got = pbox[(size_t)slot * P.n_total + k];
                    const bool both_empty = !(ref.x0 <= ref.x1) && !(got.x0 <= got.x1);
                    if (!both_empty && (fabsf(ref.x0 - got.x0) > 1e-4f || fabsf(ref.x1 - got.x1) > 1e-4f ||
                                        fabsf(ref.y0 - got.y0) > 1e-4f || fabsf(ref.y1 - got.y1) > 1e-4f)) return -107;
                }
                box_merge(boxes[k], pbox[(size_t)slot * P.n_total + k]);
                tab[0 * FOT_MAX_NT + k] = ls.s; tab[1 * FOT_MAX_NT + k] = ls.sd; tab[2 * FOT_MAX_NT + k] = ls.sdd;
                tab[3 * FOT_MAX_NT + k] = ls.rx; tab[4 * FOT_MAX_NT + k] = ls.ry;
                tab[5 * FOT_MAX_NT + k] = ls.cos_r; tab[6 * FOT_MAX_NT + k] = ls.sin_r;
                tab[7 * FOT_MAX_NT + k] = ls.kr; tab[8 * FOT_MAX_NT + k] = ls.dkr; tab[9 * FOT_MAX_NT + k] = ls.inv_sd;
                js += sddd * sddd; sd_last = ls.sd;
            }
            Li.Js = js; Li.sd_last = sd_last;
            lon_info[D.lon_off + slot] = Li;
        }
        // --- k_cull: entry lists per time step
        ObstacleView obs;
        obs.dtype = b->obstacle_dtype;
        const size_t esz = b->obstacle_dtype == FOT_F32 ? sizeof(float) : sizeof(double);
        obs.stat = b->static_xy ? (const char *)b->static_xy + 2 * esz * (size_t)D.static_off : nullptr;
        obs.dyn = b->dyn_xy ? (const char *)b->dyn_xy + 2 * esz * (size_t)D.dyn_off : nullptr;
        std::vector<int32_t> cnt((size_t)P.n_total, 0);
        const size_t cap = (size_t)D.ent_cap;
        f2 farq; farq.x = FAR32; farq.y = FAR32;
        d2 infq; infq.x = INFINITY; infq.y = INFINITY;
        std::vector<f2> e32(cap * P.n_total + 16, farq);
        std::vector<d2> e64(cap * P.n_total + 16, infq);
        std::vector<uint8_t> sid(cap * P.n_total + 16, SID_STATIC);
        // tiles of the instance (k_evaluate's units of work): the walk the host and k_frenet_state do
        std::vector<int> tile_c0, tile_n, tile_of((size_t)std::max(D.n_cand_max, 1), -1);
        // the handle's table (either cut, padding tiles of the grouped cut included): consecutive, complete, within
        // the limits the kernels rely on
        for (int t = 0, c = 0; t < shapes.tiles_of(D.n_tv); ++t) {
            const int c0 = shapes.cand0[(size_t)D.shape_off + t], n = shapes.n[(size_t)D.shape_off + t];
            if (c0 != c || n < 0 || n > WAVE || (n == 0 && !shapes.grouped)) return -102;
            if (n > 0) {
                int s0, s1, rows = 0;
                wave_profile_span(P, D, n_grid_lon, c, c + n - 1, s0, s1);
                if (s1 - s0 + 1 > TILE_MAX_PROFILES) return -103;
                if ((uint32_t)shapes.span[(size_t)D.shape_off + t] != (((uint32_t)s0 << 16) | (uint32_t)s1)) return -115;
                for (int sl = s0; sl <= s1; ++sl) rows += profile_rows(P, D, sl);
                if (rows > L.row_budget && s1 > s0) return -104;          // what a wave staging the tile alone needs
                if (!shapes.grouped && tile_extent(P, D, c, L.row_budget) != n) return -106;
                for (int i = c; i < c + n; ++i) tile_of[(size_t)i] = t;
            }
            tile_c0.push_back(c); tile_n.push_back(n);
            c += n;
            if (t == shapes.tiles_of(D.n_tv) - 1 && c != D.n_cand_max) return -108;    // every candidate in some tile
        }
        if (shapes.grouped) {                                             // a group's profiles fit its shared table
            if (shapes.tiles_of(D.n_tv) % GROUP_TILES) return -109;
            for (int g0 = 0; g0 < (int)tile_c0.size(); g0 += GROUP_TILES) {
                const int c0 = tile_c0[(size_t)g0], c1 = tile_c0[(size_t)g0 + GROUP_TILES - 1] + tile_n[(size_t)g0 + GROUP_TILES - 1];
                if (c1 <= c0) return -109;
                int s0, s1, rows = 0;
                wave_profile_span(P, D, n_grid_lon, c0, c1 - 1, s0, s1);
                for (int sl = s0; sl <= s1; ++sl) rows += profile_rows(P, D, sl);
                if (s1 - s0 + 1 > GROUP_MAX_PROFILES || (rows > GROUP_ROWS && s1 > s0)) return -109;
            }
        }
        if ((int)tile_c0.size() != D.n_tiles) return -105;                // what build_batch_layout counted
        const int n_tiles_inst = (int)tile_c0.size();
        std::vector<uint32_t> rng((size_t)std::max(n_tiles_inst, 1) * P.n_total, 0u);
        // k_cull's per (tile, step) thresholds (TileStep) and the bound of |x| + |y| they were taken at
        std::vector<float> thr_t(rng.size(), 0.0f), thr_sure_t(rng.size(), 0.0f), bound_t(rng.size(), -1.0f);
        // NanScan (fot_kernels.h): a track with a NaN coordinate at any step is no obstacle at any step
        std::vector<uint8_t> nan_track((size_t)(D.dyn_mode != FOT_DYN_NONE ? D.S * D.P : 0) + 1, 0);
        if (D.ent_cap > 0) {
            const double sq_dyn = D.dyn_mode == FOT_DYN_SINGLE ? P.sq_r_dyn : P.sq_r;
            const double sq_max = sq_dyn > P.sq_r ? sq_dyn : P.sq_r;
            for (size_t j = 0; j + 1 < nan_track.size(); ++j)
                for (int t = 0; t < D.T; ++t) {
                    const d2 o = obs.dyn_at((int)j / D.P, (int)j % D.P, t, D.P, D.T);
                    if (o.x != o.x || o.y != o.y) nan_track[j] = 1;
                }
            obs.nan_track = nan_track.data();                                // (the definition applies the same rule)
            for (int k = 0; k < P.n_total; ++k) {
                const Box32 &bx = boxes[k];
                if (!(bx.x0 <= bx.x1)) continue;
                const float margin = cull_margin(sq_max, bx) + box_footprint_slack(P);
                const BinMap bm = bin_map(bx, margin);
                const int n_dyn = D.dyn_mode != FOT_DYN_NONE ? D.S * D.P : 0;
                const int row = k < D.T - 1 ? k : D.T - 1;
                struct Ent { d2 o; float fx, fy; int sid, bin; };
                std::vector<Ent> ents;
                for (int i = 0; i < D.n_static + n_dyn; ++i) {
                    Ent e; e.sid = SID_STATIC;
                    if (i < D.n_static) e.o = obs.static_at(i);
                    else { const int j = i - D.n_static; e.o = obs.dyn_at(j / D.P, j % D.P, row, D.P, D.T); e.sid = j / D.P; }
                    e.fx = (float)(e.o.x - D.ego.x); e.fy = (float)(e.o.y - D.ego.y);
                    if (!cull_inside(bx, margin, e.fx, e.fy)) continue;
                    if (i >= D.n_static && nan_track[(size_t)(i - D.n_static)]) continue;
                    e.bin = bin_of(bm, e.fx, e.fy);
                    ents.push_back(e);
                }
                std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.bin < b.bin; });
                int bin_start[CULL_BINS + 1];
                for (int b = 0, i = 0; b <= CULL_BINS; ++b) {
                    while (i < (int)ents.size() && ents[i].bin < b) ++i;
                    bin_start[b] = i;
                }
                for (size_t i = 0; i < ents.size(); ++i) {
                    ent32_store(e32.data(), (int64_t)cap * k + (int64_t)i, ents[i].fx, ents[i].fy);
                    e64[cap * k + i] = ents[i].o; sid[cap * k + i] = (uint8_t)ents[i].sid;
                }
                const int count = (int)ents.size();
                cnt[k] = (count + 2 * ENT_CHUNK - 1) & ~(2 * ENT_CHUNK - 1);
                for (int w = 0; w < n_tiles_inst; ++w) {
                    if (tile_c0[w] >= S.n_cand || tile_n[w] == 0) continue;
                    int s0, s1;
                    wave_profile_span(P, D, n_grid_lon, tile_c0[w], std::min(tile_c0[w] + tile_n[w] - 1, S.n_cand - 1), s0, s1);
                    Box32 wb = box_empty();
                    for (int sl = s0; sl <= s1; ++sl) box_merge(wb, pbox[(size_t)sl * P.n_total + k]);
                    const float wm = cull_margin(sq_max, wb) + box_footprint_slack(P);
                    const uint32_t r = strip_range(bm, wb, wm, [&](int b) { return bin_start[b]; });
                    if ((int)(r & 0xffffu) * ENT_CHUNK > cnt[k]) return -101;          // range must stay inside the padded list
                    rng[(size_t)w * P.n_total + k] = r;
                    const double sq_dyn_ = D.dyn_mode == FOT_DYN_SINGLE ? P.sq_r_dyn : P.sq_r;
                    const FilterConst fc = filter_const(sq_max, sq_dyn_ < P.sq_r ? sq_dyn_ : P.sq_r);
                    const size_t at = (size_t)w * P.n_total + k;
                    box_thresholds(fc, wb, wm, thr_t[at], thr_sure_t[at]);
                    bound_t[at] = std::fmax(std::fabs(wb.x0), std::fabs(wb.x1)) + std::fmax(std::fabs(wb.y0), std::fabs(wb.y1));
                }
            }
        }
        // --- k_evaluate (collision test inside, against the entry lists) + k_select
        const size_t per_cand = (size_t)P.n_circ * P.n_total;
        std::vector<d2> pts(per_cand);
        int cnt_st[8] = { 0 };
        ScanBest best = { INFINITY, -1 };
        int best_keep = 0;
        for (int idx = 0; idx < S.n_cand; ++idx) {
            const CandDecode cd = decode_candidate(P, D, S.frenet0, idx);
            const LonInfo &Li = lon_info[D.lon_off + cd.lon_slot];
            const double *tab = lon_tab.data() + (size_t)(D.lon_off + cd.lon_slot) * (LON_FIELDS * FOT_MAX_NT);
            double q[6];
            lat_coeffs(S.frenet0, cd.di, cd.brake ? P.brake[cd.ti] : P.ti[cd.ti], q);
            EntryCollider ec;
            ec.init(P, D);
            ec.rng = D.ent_cap > 0 ? rng.data() + (size_t)tile_of[(size_t)idx] * P.n_total : nullptr;
            ec.e32 = e32.data(); ec.e64 = e64.data(); ec.sid = sid.data();
            if (D.ent_cap > 0) {
                ec.thr_k = thr_t.data() + (size_t)tile_of[(size_t)idx] * P.n_total;
                ec.thr_sure_k = thr_sure_t.data() + (size_t)tile_of[(size_t)idx] * P.n_total;
            }
            CandResult r;
            evaluate_candidate(P, D, loop_const(P, D), Li, GlobalTab{ tab }, q, P.n_total, ec, r);
            {   // broad phase + in-loop test vs the definition on the recorded points
                VecSink vs = { &pts, P.n_total };
                CandResult rk;
                evaluate_candidate(P, D, loop_const(P, D), Li, GlobalTab{ tab }, q, P.n_total, vs, rk);
                int want = rk.status;
                if (want == ST_PENDING && D.ent_cap > 0) {
                    VecSource src = { pts.data(), P.n_total };
                    if (collide_candidate(P, D, obs, rk.keep, src)) want = FOT_ST_COLLISION;
                }
                if (want != r.status || rk.keep != r.keep) return -100;
                // the tile's box bounds |x| + |y| of every collision point of its candidates (box_thresholds)
                const int n_circ = P.has_footprint ? P.n_circ : 1;
                for (int k = 0; k < rk.keep && D.ent_cap > 0; ++k) {
                    const float bound = bound_t[(size_t)tile_of[(size_t)idx] * P.n_total + k];
                    if (bound < 0.0f) continue;                              // (no live box at this step)
                    for (int ci = 0; ci < n_circ; ++ci) {
                        const d2 &v = pts[(size_t)ci * P.n_total + k];
                        const float fx = (float)(v.x - D.ego.x), fy = (float)(v.y - D.ego.y);
                        if (!(std::fabs(fx) + std::fabs(fy) <= bound + 2.0f * box_footprint_slack(P) + 2e-3f)) return -113;
                    }
                }
            }
            for (int n_seg = 2; n_seg <= 4; ++n_seg) {   // k_evaluate_split: time segments merged == the single walk
                const int n_loop = P.n_total, seg_len = (n_loop + n_seg - 1) / n_seg;
                const LoopConst lc = loop_const(P, D);
                SegState g;
                uint64_t hit_mask = 0;
                bool hit = false;
                for (int sg = 0; sg < n_seg; ++sg) {
                    if (sg > 0 && sg * seg_len >= Li.n_t) break;
                    EntryCollider es;
                    es.init(P, D);
                    es.rng = ec.rng; es.e32 = ec.e32; es.e64 = ec.e64; es.sid = ec.sid;
                    SegState part;
                    seg_init(part);
                    evaluate_segment(P, lc, Li, GlobalTab{ tab }, q, sg * seg_len, std::min(n_loop, (sg + 1) * seg_len),
                                     es, part);
                    bool counts = true;
                    if (sg == 0) g = part; else counts = seg_merge(g, part);
                    if (counts) { hit_mask |= es.hit_mask; hit |= es.hit; }
                }
                int pc = 0;
                for (uint64_t m = hit_mask; m; m &= m - 1) ++pc;
                hit |= pc > D.max_viol;
                CandResult rs;
                finish_candidate(P, D, Li, GlobalTab{ tab }, q, g, hit, rs);
                if (rs.status != r.status || rs.keep != r.keep) return -110;
                if (std::memcmp(&rs.v_last, &r.v_last, sizeof(double)) || std::memcmp(&rs.travel, &r.travel, sizeof(double)))
                    return -111;
                if (std::memcmp(&rs.cost, &r.cost, sizeof(double))) return -112;     // (the jerk sums are closed forms)
            }
            int st = r.status;
            st = final_status(st, r.v_last, r.travel, D.max_stop);
            if (st < 8) cnt_st[st]++;
            if (idx < cand_cap && inst == 0) {
                if (cand_cost) cand_cost[idx] = r.cost;
                if (cand_status) cand_status[idx] = st;
                if (cand_keep) cand_keep[idx] = r.keep;
            }
            if (st == FOT_ST_OK && r.cost < best.dist) { best.dist = r.cost; best.idx = idx; best_keep = r.keep; }
        }
        R.status = best.idx >= 0 ? FOT_PLAN_OK : FOT_PLAN_NO_PATH;
        R.best_index = best.idx; R.n_cand = S.n_cand; R.cost = best.idx >= 0 ? best.dist : INFINITY;
        for (int c = 0; c < 8; ++c) R.stats[c] = cnt_st[c];
        R.stats_valid = 1; R.new_prev_s = S.new_prev_s; R.new_last_kappa = D.ego.last_kappa;
        std::memcpy(R.frenet0, S.frenet0, sizeof(double) * 6);
        std::memcpy(R.ref0, S.ref0, sizeof(double) * 6);
        if (best.idx >= 0) {
            const CandDecode cd = decode_candidate(P, D, S.frenet0, best.idx);
            const LonInfo &Li = lon_info[D.lon_off + cd.lon_slot];
            const double *tab = lon_tab.data() + (size_t)(D.lon_off + cd.lon_slot) * (LON_FIELDS * FOT_MAX_NT);
            double q[6];
            lat_coeffs(S.frenet0, cd.di, cd.brake ? P.brake[cd.ti] : P.ti[cd.ti], q);
            R.n_keep = best_keep;
            double *dst[15] = { R.t, R.s, R.s_d, R.s_dd, R.s_ddd, R.d, R.d_d, R.d_dd, R.d_ddd, R.x, R.y, R.yaw, R.v, R.a, R.c };
            for (int k = 0; k < best_keep; ++k) {
                double o[15];
                final_sample(P, Li, GlobalTab{ tab }, q, k, o);
                for (int f = 0; f < 15; ++f) dst[f][k] = o[f];
            }
            if (best_keep > 1) R.new_last_kappa = R.c[1];
        }
    }
    return FOT_OK;
}

extern "C" int emu_spline(int n, const double *wx, const double *wy, double *out9n)
{
    HostSpline hs; std::string err;
    int rc = build_spline(n, wx, wy, hs, err);
    if (rc) return rc;
    const std::vector<double> *src[9] = { &hs.s, &hs.ax, &hs.bx, &hs.cx, &hs.dx, &hs.ay, &hs.by, &hs.cy, &hs.dy };
    for (int f = 0; f < 9; ++f) for (int i = 0; i < n; ++i) out9n[f * n + i] = (*src[f])[i];
    return 0;
}

// Both cuts of a planner's lattice, every terminal-speed grid size: the invariants the kernels rely on.
// 0 = fine; otherwise -(1000 * cut + 10 * check + ...) style codes: -1xx wave cut, -2xx grouped cut.
extern "C" int emu_check_tile_tables(const fot_params *params, int32_t *n_tiles_wave, int32_t *n_tiles_grouped, int32_t *chosen)
{
    std::string err;
    DevParams P;
    if (build_dev_params(*params, P, err) != FOT_OK) return -1;
    TileShapes cuts[2], picked;
    build_tile_shapes_wave(P, cuts[0]);
    build_tile_shapes_grouped(P, cuts[1]);
    build_tile_shapes(P, picked);
    *n_tiles_wave = (int32_t)cuts[0].n_real; *n_tiles_grouped = (int32_t)cuts[1].n_real; *chosen = picked.grouped;
    if (picked.cand0 != cuts[picked.grouped].cand0 || picked.n != cuts[picked.grouped].n) return -2;
    if (picked.span.size() != picked.cand0.size()) return -3;
    for (int c = 0; c < 2; ++c) {
        TileShapes &T = cuts[c];
        fill_tile_spans(P, T);
        const int base = -100 * (c + 1);
        for (int n_tv = 1; n_tv <= FOT_MAX_TV; ++n_tv) {
            const InstDesc D = shape_desc(P, n_tv);
            const int n_grid_lon = P.n_ti * n_tv, nt = T.tiles_of(n_tv);
            if (nt <= 0) return base - 1;
            if (T.grouped && nt % GROUP_TILES) return base - 2;
            int cnext = 0;
            for (int t = 0; t < nt; ++t) {
                const int c0 = T.cand0[(size_t)T.off[n_tv] + t], n = T.n[(size_t)T.off[n_tv] + t];
                if (c0 != cnext || n < 0 || n > WAVE || (n == 0 && !T.grouped)) return base - 3;
                cnext += n;
                if (n == 0) { if (T.span[(size_t)T.off[n_tv] + t] != 0) return base - 4; continue; }
                int s0, s1, rows = 0;
                wave_profile_span(P, D, n_grid_lon, c0, c0 + n - 1, s0, s1);
                if (s1 - s0 + 1 > TILE_MAX_PROFILES) return base - 5;
                for (int sl = s0; sl <= s1; ++sl) rows += profile_rows(P, D, sl);
                if (rows > T.row_budget && s1 > s0) return base - 6;
                if ((uint32_t)T.span[(size_t)T.off[n_tv] + t] != (((uint32_t)s0 << 16) | (uint32_t)s1)) return base - 7;
            }
            if (cnext != D.n_cand_max) return base - 8;                     // every candidate in exactly one tile
            if (T.grouped) {
                for (int g0 = 0; g0 < nt; g0 += GROUP_TILES) {
                    const size_t o = (size_t)T.off[n_tv] + g0;
                    const int c0 = T.cand0[o], c1 = T.cand0[o + GROUP_TILES - 1] + T.n[o + GROUP_TILES - 1];
                    if (c1 <= c0 || T.n[o] == 0) return base - 9;           // no empty group, padding only at a group's end
                    int s0, s1, rows = 0;
                    wave_profile_span(P, D, n_grid_lon, c0, c1 - 1, s0, s1);
                    for (int sl = s0; sl <= s1; ++sl) rows += profile_rows(P, D, sl);
                    if (s1 - s0 + 1 > GROUP_MAX_PROFILES || (rows > GROUP_ROWS && s1 > s0)) return base - 10;
                    for (int t = 1; t < GROUP_TILES; ++t)
                        if (T.n[o + t] > 0 && T.n[o + t - 1] == 0) return base - 11;
                }
            }
        }
    }
    return 0;
}

// hypot_cr / cube_cr (fot_math.hpp) on arrays, for the test that holds them to Python's math.hypot and NumPy's power
extern "C" void emu_hypot_cr(int n, const double *x, const double *y, double *out)
{
    for (int i = 0; i < n; ++i) out[i] = hypot_cr(x[i], y[i]);
}

extern "C" void emu_cube_cr(int n, const double *h, double *out)
{
    for (int i = 0; i < n; ++i) out[i] = cube_cr(h[i]);
}

extern "C" void emu_spline_xy(int n_knots, const double *coef9n, int n, const double *s, double *x, double *y)
{
    SplineView v;
    v.s = coef9n; v.ax = coef9n + n_knots; v.bx = coef9n + 2 * n_knots; v.cx = coef9n + 3 * n_knots; v.dx = coef9n + 4 * n_knots;
    v.ay = coef9n + 5 * n_knots; v.by = coef9n + 6 * n_knots; v.cy = coef9n + 7 * n_knots; v.dy = coef9n + 8 * n_knots;
    v.n = n_knots; v._pad = 0;
    for (int i = 0; i < n; ++i) spline_xy(v, s[i], x[i], y[i]);
}
