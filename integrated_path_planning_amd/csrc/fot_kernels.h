// fot_kernels.h -- host-callable launchers of the gfx950 kernels (fot_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include "fot_types.h"

namespace fot {

// per-candidate arrays in HBM, one slot per candidate (instances padded to multiples of 64)
struct CandArrays {
    double *cost, *v_last, *travel;
    uint8_t *status, *keep;
};

// every launcher returns 0 or the hipError_t of the launch
int launch_prep_static(const InstDesc *desc, int n_inst, int max_static4, const void *src, int dtype, d2 *stat,
                       f2 *stat32, hipStream_t st);
int launch_prep_dyn(const InstDesc *desc, int n_inst, int64_t max_rows32, const void *src, int dtype, d2 *rows,
                    f2 *rows32, hipStream_t st);
int launch_frenet_state(const DevParams *P, SplineView sp, const InstDesc *desc, InstState *state, int n_inst,
                        hipStream_t st);
int launch_lon_table(const DevParams *P, SplineView sp, const InstDesc *desc, const InstState *state,
                     LonInfo *lon_info, double *lon_tab, int n_inst, int max_lon, hipStream_t st);
int launch_evaluate(const DevParams *P, const InstDesc *desc, const InstState *state, const LonInfo *lon_info,
                    const double *lon_tab, const int32_t *wave_inst, const int32_t *wave_base, int n_waves,
                    CandArrays c, d2 *pts, hipStream_t st);
int launch_collide(const DevParams *P, const InstDesc *desc, const int32_t *wave_inst, const int32_t *wave_base,
                   int n_waves, const d2 *stat, const f2 *stat32, const d2 *rows, const f2 *rows32, const d2 *pts,
                   CandArrays c, hipStream_t st);
int launch_select(const DevParams *P, const InstDesc *desc, const InstState *state, const LonInfo *lon_info,
                  const double *lon_tab, CandArrays c, fot_result *out, int n_inst, hipStream_t st);
int launch_spline_eval(SplineView sp, int n, const double *s, double *out, hipStream_t st);
int launch_collide_ext(const DevParams *P, const InstDesc *desc, int n_paths, const int32_t *len, const d2 *pts,
                       const int32_t *tidx, const d2 *stat, const d2 *rows, int32_t *free_out, hipStream_t st);

}  // namespace fot
