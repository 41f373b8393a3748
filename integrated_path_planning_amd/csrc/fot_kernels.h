// fot_kernels.h -- host-callable launchers of the gfx950 kernels (fot_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include "fot_types.h"

namespace fot {

// per-candidate arrays in HBM, one slot per candidate (instances padded to multiples of 64)
// what one tile's wave leaves for the selection: its cheapest 'ok' candidate (index inside the instance, kept samples)
// and the histogram of its candidates' final statuses (FOT_ST_* 0..7)
struct TilePart {
    double cost;
    int32_t idx, keep;
    int32_t cnt[8];
};

struct CandArrays {
    double *cost;                       // per candidate, for fot_debug_candidates: cost, final status, kept samples
    uint8_t *status;
    uint16_t *keep;                     // (up to FOT_MAX_NT = 256 kept samples)
    TilePart *parts;                    // [n_tiles of the batch]
    int32_t *done_flag = nullptr;       // pinned, one per instance, or nullptr: raised to done_seq behind the instance's record
    int32_t done_seq = 0;
};

// What k_cull leaves per (tile, time step) for k_evaluate: the chunk range the tile's own profiles can reach
// (c_lo << 16 | c_hi) and the float32 filter thresholds valid for every collision point of the tile at that step
// (filter_threshold / filter_threshold_sure at the bound of |x| + |y| over the tile's box, fot_math.hpp box_thresholds).
struct TileStep {
    uint32_t rng;
    float thr, thr_sure;
    uint32_t pad;
};

// broad-phase entry lists in HBM: per instance n_total * ent_cap slots
struct EntryArrays {
    int32_t *cnt;      // [n_inst][n_total] entries of each time step (multiple of 8)
    f2 *e32;           // instance-local float32 coordinates
    d2 *e64;           // exact coordinates
    uint8_t *sid;      // prediction sample of the entry, SID_STATIC for static obstacles
    TileStep *rng;     // [n_tiles][n_total] what a tile needs of time step k: chunk range + float32 thresholds
    const uint8_t *nan_flag = nullptr;   // [n_tracks] NanScan::flag
    int eager_nan = 0;                   // the flags are there for every layout (NanScan::eager): k_cull looks nothing up itself
};

// The reference ignores a pedestrian whose track holds a NaN coordinate at ANY time step, at EVERY time step (its
// pre-filter takes np.min / np.max over the whole track, frenet_planner.py:1211-1219).  The blocks behind the
// n_inst nearest-point blocks of k_frenet_state's launch scan the caller's dynamic tensors once -- whatever produced
// them: the host packer, fot_resample_predictions, a PyTorch tensor handed to fot_plan_batch_device -- and leave one
// flag per (sample, pedestrian) track; k_cull drops flagged tracks.  blocks_per_inst = 0: no dynamic obstacles.
struct NanScan {
    const void *dyn_xy = nullptr;
    int dtype = 0;
    uint8_t *flag = nullptr;            // [n_tracks] 1: the track holds a NaN
    int blocks_per_inst = 0;
    int eager = 0;                      // scan [S][P][T] tensors too (FOT_NAN_SCAN=eager; default: k_cull finds their NaNs)
    void *stage = nullptr;              // HBM copy of the tensors, written as they are scanned (same offsets), or nullptr:
                                        // a small call's tensors lie in pinned host memory -- one pass over PCIe instead
                                        // of three (scan, classification, scatter)
};

// descriptors still in pinned host memory, to be moved into HBM by k_frenet_state (h_desc == nullptr: already there)
struct MetaImport {
    const InstDesc *h_desc = nullptr;
    InstDesc *d_desc = nullptr;
};

// The handle's tile table in HBM (built once per handle, one run per terminal-speed grid size = lattice shape):
// tile t of an instance = candidates [cand0[shape_off + t], + n[shape_off + t]).  n_tiles / max_tiles: of the batch.
struct TileTable {
    const int32_t *cand0 = nullptr, *n = nullptr;
    const int32_t *span = nullptr;      // first << 16 | last profile of each tile
    int n_tiles = 0, max_tiles = 0, row_budget = 0;
    int grouped = 0;                    // groups of GROUP_TILES tiles share a row table (TileShapes::grouped)
    int eval_segments = 0;              // time segments per tile in k_evaluate: 0 = by batch size, 1..4 forced (tests)
};

// every launcher returns 0 or the hipError_t of the launch
int launch_frenet_state(const DevParams *P, SplineView sp, const InstDesc *desc, InstState *state, int n_inst,
                        MetaImport imp, NanScan scan, int32_t *inst_done, hipStream_t st);
// n_ext: horizons + brake-ladder entries of the planner (sizes k_cull's per-horizon tables in LDS)
int launch_cull(const DevParams *P, const InstDesc *desc, const InstState *state, int n_inst, int n_total, int n_ext,
                SplineView sp, const void *static_xy, const void *dyn_xy, int dtype, EntryArrays e, TileTable tiles,
                hipStream_t st);
// evaluation + selection: the records land in `out`; inst_done: one counter per instance (zeroed by k_frenet_state)
int launch_evaluate(const DevParams *P, SplineView sp, const InstDesc *desc, const InstState *state, int n_total,
                    int n_inst, TileTable tiles, EntryArrays e, CandArrays c, fot_result *out, int32_t *inst_done,
                    hipStream_t st);
int launch_debug_path(const DevParams *P, const InstDesc *desc, const InstState *state,
                      SplineView sp, int inst, int idx, double *out, int32_t *meta, hipStream_t st);
int launch_debug_margins(const DevParams *P, const InstDesc *desc, const InstState *state, SplineView sp, int inst,
                         EntryArrays e, int cap, double *out, hipStream_t st);
int launch_pack_wire(int n, int n_total, int stride, const fot_result *src, unsigned char *dst, hipStream_t st);
int launch_spline_eval(SplineView sp, int n, const double *s, double *out, hipStream_t st);
int launch_resample(double sgan_dt, double sim_dt, double staleness, int S, int pred_len, int P, int n_dense,
                    int has_anchor, int prepend, int cv, const void *pred, int pred_dtype, const double *anchor,
                    const double *current, void *out, int out_dtype, int tmajor, hipStream_t st,
                    const int32_t *ped_ep = nullptr, const int32_t *ep_ped0 = nullptr, const int64_t *ep_blk = nullptr);
int launch_sample_dist(int S, int P, int T, int skip, const void *out, int out_dtype, int tmajor, double *dist,
                       hipStream_t st);
int launch_safety(const DevParams *P, int n, const double *ego, const int32_t *ped_off, const double *ped_pos,
                  const double *ped_vel, double ego_radius, double ped_radius, double footprint_radius, int use_fp,
                  fot_safety *out, hipStream_t st);
int launch_check_ext(const DevParams *P, const InstDesc *desc, int n_paths, int mode, const int32_t *len,
                     const int32_t *flags, const double *arrays, const double *static_xy, const double *dyn_xy,
                     int32_t *status_out, hipStream_t st);

}  // namespace fot
