// fot_host.cpp -- the C ABI of libfot.so (include/fot.h): handle, device workspace, batch staging.
// No torch, no oracle, no CPU fallback: every entry point drives the gfx950 kernels or fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "fot_kernels.h"
#include "fot_math.hpp"
#include "fot_setup.hpp"

using namespace fot;

namespace {

thread_local std::string g_create_error;

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    // the same, and a block that grew starts out all zero (record blocks: entries past n_total are never written)
    hipError_t ensure_zeroed(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        hipError_t e = ensure(bytes);
        return e != hipSuccess ? e : hipMemset(p, 0, cap);
    }
    template <class T> T *as() const { return (T *)p; }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        // fine-grained (coherent) whatever HIP_HOST_COHERENT says: kernels write records and their flags into these
        // blocks while the host polls them (wait_records)
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocCoherent | hipHostMallocMapped);
        if (e != hipSuccess) { p = nullptr; return e; }
        std::memset(p, 0, want);                                  // (record blocks: entries past n_total are never written)
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// Per-lane workspace.  A large batch is split into FOT_LANES contiguous sub-batches that run on their own
// streams: the short, latency-bound kernels of one half (nearest-point search, tables, cull, select) overlap
// with the wide kernels of the other half instead of leaving the GPU idle between them.
struct Workspace {
    hipStream_t stream = nullptr;            // internal stream of this lane
    // The descriptors of a call are staged in a pinned block that the call's first kernel reads.  Two blocks take turns,
    // each with the event recorded BEHIND the last kernel of the call that used it: an event between two kernels costs
    // the second one about 6 us (profiles/r03_latency_anatomy.json: the gap behind k_frenet_state), and with two blocks
    // the host only ever waits for the call before the previous one.
    hipEvent_t staging_done[2] = { nullptr, nullptr };
    hipEvent_t done = nullptr;               // this lane's part of the current call has been enqueued up to here
    bool staging_pending[2] = { false, false };
    int staging_slot = 0;
    PinnedBuf staging[2];
    DevBuf dMeta;                            // InstDesc[]
    DevBuf dState;
    DevBuf dCost, dStatus, dKeep, dParts;
    DevBuf dEntCnt, dEnt32, dEnt64, dEntSid, dWaveRng;   // broad phase: culled entry lists + per-wave chunk ranges
    DevBuf dNanFlag;                         // one flag per pedestrian track (NanScan)
    DevBuf dDone;                            // tiles evaluated so far, per instance (tile_done)
    BatchLayout last;                        // layout of this lane's part of the most recent plan call
    int first_inst = 0;                      // global index of this lane's first instance
    void release()
    {
        DevBuf *bufs[] = { &dMeta, &dState, &dCost, &dParts, &dStatus, &dKeep,
                           &dWaveRng, &dEntCnt, &dEnt32, &dEnt64, &dEntSid, &dNanFlag, &dDone };
        for (DevBuf *b : bufs) b->release();
        for (int i = 0; i < 2; ++i) {
            staging[i].release();
            if (staging_done[i]) (void)hipEventDestroy(staging_done[i]);
        }
        if (done) (void)hipEventDestroy(done);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// fot_loop_begin / fot_loop_step: the episodes' own state (what IntegratedSimulator, its planner and its
// FailSafeStateMachine carry from step to step), one slot per episode
struct LoopEpisodes {
    int n = 0;
    fot_loop_config cfg = fot_loop_config();
    std::vector<double> ego;                 // [n][5] x, y, yaw, v, a
    std::vector<double> prev_s, last_kappa;  // planner.converter._prev_s (NaN: none yet), planner._last_kappa
    std::vector<double> goal_prev_s;         // the simulator's own converter (integrated_simulator.py:873)
    std::vector<double> last_clearance;      // clearance_ahead of the step's own metrics (emergency stop)
    std::vector<double> clear, clear_ahead;  // the state machine's _last_clearance / _last_clearance_ahead
    std::vector<int32_t> state, fails;       // 0 / 1 / 2 = NORMAL / CAUTION / EMERGENCY; consecutive failures
    std::vector<int32_t> stats;              // [n][8] last_check_stats, a row of -1: None
};

// fot_loop_*: what one closed-loop step leaves behind for the step's later calls
struct LoopState {
    LoopEpisodes ep;
    PinnedBuf hFrame, hObserve, hOut, hRec;  // frame inputs | fot_loop_observe's inputs | small outputs | records
    DevBuf dDyn, dStatic;                    // the prediction tensor; the static points, one copy per request
    std::vector<double> static_xy;           // host copy of the static points
    int static_tiles = 0;                    // copies resident in dStatic
    std::vector<int32_t> ped_off;            // of the frame
    std::vector<int64_t> blk_off;            // first point of each episode's block in the tensor
    std::vector<int32_t> t_len;              // samples per track of each episode's block
    int dist_S = 0;                          // > 0: the blocks are [dist_S][P_e][t_len][2] distributions
    bool have_frame = false;
    const void *dyn_ptr = nullptr;           // the tensor: dDyn, or the pinned current positions (predictor not ready)
    const int32_t *p_off = nullptr;          // the frame's pedestrians in hFrame
    const double *p_pos = nullptr, *p_vel = nullptr;
    double ego_radius = 0.0, ped_radius = 0.0;
    int use_footprint = 0;
    int observe_n = -1;                      // egos of a fot_loop_observe_begin not collected yet
    void release()
    {
        hFrame.release(); hObserve.release(); hOut.release(); hRec.release();
        dDyn.release(); dStatic.release();
    }
};

constexpr int FOT_LANES = 4;                  // lanes available; lanes_cfg of them are used (FOT_LANES env, default 1)
constexpr int FOT_SPLIT_MIN_INSTANCES = 32;  // smaller batches run as one piece on the caller's stream

struct fot_handle {
    int device = 0;
    hipStream_t stream = nullptr;            // the handle's own stream (host-pointer entry points, helpers)
    hipEvent_t fork = nullptr;               // caller's stream -> lanes
    // One handle = ONE workspace and one set of scratch buffers, so its enqueues are ordered whatever streams the
    // caller passes: an enqueue on a stream other than the previous one first waits (device side) for the event
    // recorded behind the previous enqueue.  Work of different handles still overlaps freely.
    hipEvent_t order_done = nullptr;
    hipStream_t order_stream = nullptr;
    bool order_valid = false;
    int eval_segments = 0;               // fot_debug_set_eval_segments
    // Completion of a synchronous small call without a stream synchronisation: the wave that writes a record (into
    // pinned memory) raises that record's flag to the call's sequence number behind a system-scope release; the host
    // polls the flags.  hipStreamSynchronize returns some 10 us after the last kernel ended on this platform -- a sixth
    // of a one-ego plan call.
    PinnedBuf hDone;
    int32_t done_seq = 0;
    bool done_seq_armed = false;             // the call being enqueued wants the flags
    int tile_cut = 0;                    // fot_debug_set_tile_cut (TILE_CUT_*)
    fot_params params;
    DevParams P;
    DevBuf dP;
    HostSpline spline;
    DevBuf dSpline;
    TileShapes shapes;                       // tile table (host copy) ...
    DevBuf dShapes;                          // ... and in HBM: cand0[] | n[] | span[]
    bool has_path = false;
    Workspace ws[FOT_LANES];
    int lanes_cfg = 1;                       // sub-batches a large batch is split into
    int lanes_used = 0;                      // lanes of the most recent plan call
    DevBuf dUserStatic, dUserDyn, dOut;      // device copies for the host-pointer entry point
    PinnedBuf hSmallIn, hSmallOut;           // ... and, for small calls, pinned host blocks the kernels use directly
    PinnedBuf hRecOut;                       // ... the records of a small plan call (a block of its own: zero past n_total)
    DevBuf dTmpA, dTmpB, dTmpC, dTmpD;
    LoopState loop;
    bool last_valid = false;
    // profiling: event pairs around kernel launches
    bool prof_on = false;
    std::vector<hipEvent_t> prof_pool;       // all events ever created
    size_t prof_used = 0;                    // events handed out since the last read
    std::vector<int> prof_kernel;            // kernel id of pair i (events 2i, 2i+1)
    int32_t prof_launches[FOT_PROFILE_KERNELS] = { 0 };
    double prof_ms[FOT_PROFILE_KERNELS] = { 0 };
    std::string err;
};

namespace {

// Every handle fot_create returned and fot_destroy has not taken back: fot_destroy of anything else (a second destroy,
// a stale pointer) is a no-op instead of a double free, and a binding can ask how many it still owns (fot_live_handles).
std::mutex g_live_mu;
std::unordered_set<fot_handle *> &live_handles()
{
    static std::unordered_set<fot_handle *> *s = new std::unordered_set<fot_handle *>();   // (never destroyed: no static
    return *s;                                                                            //  destructor order to lose)
}

// Teardown never blocks for good.  Work the handle enqueued may sit on a CALLER's stream (fot_plan_batch_device on a
// PyTorch stream) whose owner is free to have destroyed it, and a destroy may run while the process is exiting: the
// handle's streams and its ordering event are POLLED (hipStreamQuery / hipEventQuery) for a bounded time; if they do
// not drain, or the runtime answers with an error, the device and pinned memory is left to the process teardown
// instead of being freed under work that may still touch it.
double destroy_timeout_s()
{
    if (const char *ev = std::getenv("FOT_DESTROY_TIMEOUT_MS")) { const double v = std::atof(ev); if (v >= 0.0) return v * 1e-3; }
    return 5.0;
}

void destroy_handle(fot_handle *h);           // (below fot_create: frees what a handle owns, registered or not)

template <class Query>
bool drained(const Query &query, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = query();
        if (e == hipSuccess) return true;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); return false; }   // stream / context gone: nothing to wait for, nothing to free
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

int fail(fot_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(fot_handle *h, hipError_t e, const char *what)
{
    return fail(h, FOT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail((h), e_, #expr); } while (0)
#define LAUNCH_TRY(h, expr) do { int r_ = (expr); if (r_ != 0) return hip_fail((h), (hipError_t)r_, #expr); } while (0)

// see fot_handle::order_done
int order_begin(fot_handle *h, hipStream_t st)
{
    if (h->order_valid && st != h->order_stream) HIP_TRY(h, hipStreamWaitEvent(st, h->order_done, 0));
    return FOT_OK;
}

int order_end(fot_handle *h, hipStream_t st)
{
    HIP_TRY(h, hipEventRecord(h->order_done, st));
    h->order_stream = st;
    h->order_valid = true;
    return FOT_OK;
}

SplineView spline_view(const fot_handle *h)
{
    SplineView v;
    const double *b = h->dSpline.as<double>();
    const int n = h->spline.n;
    v.s = b; v.ax = b + n; v.bx = b + 2 * n; v.cx = b + 3 * n; v.dx = b + 4 * n;
    v.ay = b + 5 * n; v.by = b + 6 * n; v.cy = b + 7 * n; v.dy = b + 8 * n;
    v.n = n; v._pad = 0;
    return v;
}

int upload_spline(fot_handle *h)
{
    const int n = h->spline.n;
    std::vector<double> flat((size_t)9 * n, 0.0);
    const std::vector<double> *src[9] = { &h->spline.s, &h->spline.ax, &h->spline.bx, &h->spline.cx, &h->spline.dx,
                                          &h->spline.ay, &h->spline.by, &h->spline.cy, &h->spline.dy };
    for (int f = 0; f < 9; ++f)
        std::memcpy(flat.data() + (size_t)f * n, src[f]->data(), sizeof(double) * std::min((size_t)n, src[f]->size()));
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());                          // nothing may still be reading the old spline
    HIP_TRY(h, h->dSpline.ensure(flat.size() * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->dSpline.p, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    h->has_path = true;
    h->last_valid = false;
    return FOT_OK;
}

// (re)builds the handle's tile table for the given cut and puts it into HBM: cand0[] | n[] | span[]
int upload_tile_shapes(fot_handle *h, int cut)
{
    build_tile_shapes(h->P, h->shapes, cut);
    h->tile_cut = cut;
    const size_t nt = h->shapes.cand0.size();
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());                          // nothing may still be reading the old table
    HIP_TRY(h, h->dShapes.ensure(sizeof(int32_t) * 3 * std::max<size_t>(nt, 1)));
    if (nt) {
        HIP_TRY(h, hipMemcpy(h->dShapes.p, h->shapes.cand0.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->dShapes.as<int32_t>() + nt, h->shapes.n.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->dShapes.as<int32_t>() + 2 * nt, h->shapes.span.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
    }
    h->last_valid = false;
    return FOT_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Calls whose inputs and outputs stay below this size skip the copy operations: the kernels read / write pinned host
// memory directly (fot_plan_batch, fot_safety_metrics_batch, fot_frenet_state_batch, the host path of the resampler).
constexpr size_t SMALL_CALL_BYTES = (size_t)1 << 20;

const char *const kKernelNames[FOT_PROFILE_KERNELS] = { "k_frenet_state", "k_cull", "k_evaluate" };

// accumulate finished event pairs into the per-kernel totals (waits for them)
int prof_drain(fot_handle *h)
{
    for (size_t i = 0; i < h->prof_kernel.size(); ++i) {
        hipEvent_t a = h->prof_pool[2 * i], b = h->prof_pool[2 * i + 1];
        HIP_TRY(h, hipEventSynchronize(b));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, a, b));
        h->prof_launches[h->prof_kernel[i]] += 1;
        h->prof_ms[h->prof_kernel[i]] += (double)ms;
    }
    h->prof_kernel.clear();
    h->prof_used = 0;
    return FOT_OK;
}

// brackets one launch with events when profiling is on
struct ProfScope {
    fot_handle *h; hipStream_t st; hipEvent_t stop = nullptr; bool active = false;
    ProfScope(fot_handle *h_, int kernel, hipStream_t st_) : h(h_), st(st_)
    {
        if (!h->prof_on) return;
        while (h->prof_pool.size() < h->prof_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            h->prof_pool.push_back(e);
        }
        hipEvent_t start = h->prof_pool[h->prof_used];
        stop = h->prof_pool[h->prof_used + 1];
        if (hipEventRecord(start, st) != hipSuccess) return;
        h->prof_used += 2;
        h->prof_kernel.push_back(kernel);
        active = true;
    }
    ~ProfScope() { if (active) (void)hipEventRecord(stop, st); }
};

// Stage descriptors, size the lane's workspace and enqueue the whole pipeline for the sub-batch `b` on `st`.
// d_static / d_dyn are device pointers to the caller's obstacle coordinates (offsets in b are absolute),
// d_out the device fot_result slot of the sub-batch's first instance.
// sync_caller: see enqueue_plan.
// d_dyn_stage: HBM block the NaN-scan blocks copy the dynamic tensors into as they read them (a small call's tensors
// lie in pinned host memory: the later kernels then read the copy instead of crossing PCIe again), or nullptr.
int enqueue_lane(fot_handle *h, Workspace &w, const fot_batch &b, const void *d_static, const void *d_dyn,
                 fot_result *d_out, hipStream_t st, bool sync_caller = false, void *d_dyn_stage = nullptr)
{
    BatchLayout &L = w.last;
    std::string err;
    int rc = build_batch_layout(h->params, h->P, h->shapes, b, L, err);
    if (rc != FOT_OK) return fail(h, rc, err);
    if (L.n_inst == 0) return FOT_OK;

    // --- staging: the descriptors in a pinned block (k_frenet_state pulls them into HBM)
    const size_t desc_bytes = align256(sizeof(InstDesc) * (size_t)L.n_inst);
    const size_t meta_bytes = desc_bytes;
    const int slot = (w.staging_slot ^= 1);
    if (w.staging_pending[slot]) { HIP_TRY(h, hipEventSynchronize(w.staging_done[slot])); w.staging_pending[slot] = false; }
    HIP_TRY(h, w.staging[slot].ensure(desc_bytes));
    char *stg = (char *)w.staging[slot].p;
    std::memcpy(stg, L.desc.data(), sizeof(InstDesc) * (size_t)L.n_inst);

    // --- workspace (grow-only; a growing hipFree/hipMalloc synchronises, steady state does not)
    const DevParams &P = h->P;
    HIP_TRY(h, w.dMeta.ensure(meta_bytes));
    HIP_TRY(h, w.dState.ensure(sizeof(InstState) * (size_t)L.n_inst));
    const size_t slots = (size_t)std::max<int64_t>(L.n_slots, 1);
    HIP_TRY(h, w.dCost.ensure(sizeof(double) * slots));
    HIP_TRY(h, w.dParts.ensure(sizeof(TilePart) * (size_t)std::max(L.n_tiles, 1)));
    HIP_TRY(h, w.dStatus.ensure(slots));
    HIP_TRY(h, w.dKeep.ensure(sizeof(uint16_t) * slots));
    const size_t n_ent = (size_t)L.n_entries + 64;              // slack: the scalar prefetch reads one chunk ahead
    HIP_TRY(h, w.dEntCnt.ensure(sizeof(int32_t) * (size_t)P.n_total * (size_t)L.n_inst));
    HIP_TRY(h, w.dEnt32.ensure(sizeof(f2) * n_ent));
    HIP_TRY(h, w.dEnt64.ensure(sizeof(d2) * n_ent));
    HIP_TRY(h, w.dEntSid.ensure(n_ent));
    HIP_TRY(h, w.dWaveRng.ensure(sizeof(TileStep) * (size_t)P.n_total * (size_t)std::max(L.n_tiles, 1)));
    HIP_TRY(h, w.dNanFlag.ensure((size_t)std::max<int64_t>(L.n_tracks, 16)));
    HIP_TRY(h, w.dDone.ensure(sizeof(int32_t) * (size_t)L.n_inst));

    // no H2D copy in front of the kernels: k_frenet_state pulls the staging block into HBM (one dependent hop less)
    w.staging_pending[slot] = true;

    const InstDesc *d_desc = (const InstDesc *)w.dMeta.p;
    TileTable tt;
    tt.cand0 = h->dShapes.as<int32_t>();
    tt.n = h->dShapes.as<int32_t>() + h->shapes.cand0.size();
    tt.span = h->dShapes.as<int32_t>() + 2 * h->shapes.cand0.size();
    tt.n_tiles = L.n_tiles; tt.max_tiles = L.max_tiles; tt.row_budget = L.row_budget;
    tt.eval_segments = h->eval_segments;
    tt.grouped = L.grouped;
    const DevParams *dP = h->dP.as<DevParams>();
    const SplineView sv = spline_view(h);
    CandArrays ca;
    ca.cost = w.dCost.as<double>(); ca.parts = w.dParts.as<TilePart>();
    ca.status = w.dStatus.as<uint8_t>(); ca.keep = w.dKeep.as<uint16_t>();
    if (sync_caller && h->done_seq_armed) { ca.done_flag = (int32_t *)h->hDone.p; ca.done_seq = h->done_seq; }

    EntryArrays ea;
    ea.cnt = w.dEntCnt.as<int32_t>(); ea.e32 = w.dEnt32.as<f2>(); ea.e64 = w.dEnt64.as<d2>();
    ea.sid = w.dEntSid.as<uint8_t>(); ea.rng = w.dWaveRng.as<TileStep>();
    ea.nan_flag = w.dNanFlag.as<uint8_t>();
    MetaImport imp;
    imp.h_desc = (const InstDesc *)stg;
    imp.d_desc = (InstDesc *)w.dMeta.p;
    // one scan block per 256 KB of an instance's tensor (few, fat blocks: each first reads its descriptor out of the
    // pinned staging block, a PCIe round trip), at most 64 per instance
    // Only time-major tensors are scanned (k_cull finds the NaN tracks of the caller's [S][P][T] layout itself, among the
    // few its boxes touch); a small call's tensors in pinned memory take the same blocks for their one pass over PCIe.
    // (FOT_NAN_SCAN=eager: the scan blocks for every layout, as before round 4 -- the scan then doubles as a prefetch of
    // the tensor into the memory-side cache: 4 us less on the serial step of config 4, the same step with three calls in
    // flight, the tensor read twice.  A small call's staging pass leaves the flags behind anyway: k_cull uses them.)
    static const bool eager_env = getenv("FOT_NAN_SCAN") && !std::strcmp(getenv("FOT_NAN_SCAN"), "eager");
    const bool eager_nan = eager_env || d_dyn_stage != nullptr;
    ea.eager_nan = eager_nan ? 1 : 0;
    NanScan scan;
    scan.eager = ea.eager_nan;
    if (L.n_tracks > 0 && (L.any_tmajor || eager_nan)) {
        scan.dyn_xy = d_dyn; scan.dtype = b.obstacle_dtype; scan.flag = w.dNanFlag.as<uint8_t>();
        scan.blocks_per_inst = (int)std::min<int64_t>(64, std::max<int64_t>(1, (L.max_dyn_bytes + 262143) / 262144));
        if (d_dyn_stage) {                                       // (cull and the rest read the copy)
            scan.stage = d_dyn_stage; d_dyn = d_dyn_stage;
            // across PCIe a block moves 64 KB per round trip: one block per 64 KB, so that the copy is done well inside
            // the nearest-point chain it runs beside
            scan.blocks_per_inst = (int)std::min<int64_t>(64, std::max<int64_t>(1, (L.max_dyn_bytes + 65535) / 65536));
        }
    }
    {
        ProfScope ps(h, 0, st);
        LAUNCH_TRY(h, launch_frenet_state(dP, sv, d_desc, w.dState.as<InstState>(), L.n_inst, imp, scan, w.dDone.as<int32_t>(), st));
    }
    if (L.any_obstacles) {
        ProfScope ps(h, 1, st);
        LAUNCH_TRY(h, launch_cull(dP, d_desc, w.dState.as<InstState>(), L.n_inst, P.n_total, P.n_ti + P.n_brake, sv,
                                  d_static, d_dyn, b.obstacle_dtype, ea, tt, st));
    }
    {
        ProfScope ps(h, 2, st);
        LAUNCH_TRY(h, launch_evaluate(dP, sv, d_desc, w.dState.as<InstState>(), P.n_total, L.n_inst, tt, ea, ca, d_out,
                                      w.dDone.as<int32_t>(), st));
    }
    HIP_TRY(h, hipEventRecord(w.staging_done[slot], st));        // (behind the call: see Workspace::staging_done)
    return FOT_OK;
}

// sub-batch [i0, i0+n) of b: per-instance arrays shifted, obstacle offsets stay absolute
fot_batch sub_batch(const fot_batch &b, int i0, int n)
{
    fot_batch s = b;
    s.n_inst = n;
    s.ego = b.ego + i0;
    s.target_speed = b.target_speed + i0;
    s.overrides = b.overrides ? b.overrides + i0 : nullptr;
    s.max_stop_distance = b.max_stop_distance ? b.max_stop_distance + i0 : nullptr;
    s.static_off = b.static_off ? b.static_off + i0 : nullptr;
    s.dyn_off = b.dyn_off ? b.dyn_off + i0 : nullptr;
    s.dyn_dims = b.dyn_dims ? b.dyn_dims + 4 * (size_t)i0 : nullptr;
    return s;
}

// Enqueue one plan call behind everything already on `user`; small batches run on `user` itself, large ones
// fork into the lanes' streams and join `user` again.
// sync_caller: the caller waits for the records right behind this call (the synchronous entry points): the selecting
// waves then raise a flag per record in pinned memory (wait_records).
int enqueue_plan(fot_handle *h, const fot_batch &b, const void *d_static, const void *d_dyn, fot_result *d_out,
                 hipStream_t user, bool sync_caller = false, void *d_dyn_stage = nullptr)
{
    if (!h->has_path) return fail(h, FOT_ERR_NO_PATH_SET, "fot_set_path_* has not been called");
    h->last_valid = false;
    if (b.n_inst < 0) return fail(h, FOT_ERR_INVALID, "n_inst < 0");
    if (b.n_inst == 0) { h->lanes_used = 0; h->last_valid = true; return FOT_OK; }
    if (!d_out) return fail(h, FOT_ERR_INVALID, "out is NULL");
    if (!b.ego || !b.target_speed) return fail(h, FOT_ERR_INVALID, "ego / target_speed missing");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->prof_on && h->prof_kernel.size() > 16384) { int r = prof_drain(h); if (r != FOT_OK) return r; }
    { int r = order_begin(h, user); if (r != FOT_OK) return r; }

    if (b.n_inst < FOT_SPLIT_MIN_INSTANCES * h->lanes_cfg / 2 || h->lanes_cfg <= 1) {
        h->ws[0].first_inst = 0;
        int rc = enqueue_lane(h, h->ws[0], b, d_static, d_dyn, d_out, user, sync_caller, d_dyn_stage);
        if (rc != FOT_OK) return rc;
        h->lanes_used = 1;
        h->last_valid = true;
        return order_end(h, user);
    }
    HIP_TRY(h, hipEventRecord(h->fork, user));
    int i0 = 0;
    const int lanes = h->lanes_cfg;
    for (int l = 0; l < lanes; ++l) {
        Workspace &w = h->ws[l];
        int n = b.n_inst / lanes + (l < b.n_inst % lanes ? 1 : 0);
        if (l == lanes - 1) n = b.n_inst - i0;
        while (i0 + n < b.n_inst && b.ego[i0 + n].has_prev_s == FOT_PREV_S_CHAINED) ++n;   // never cut a chain
        if (n > b.n_inst - i0) n = b.n_inst - i0;
        if (n <= 0) { w.last = BatchLayout(); w.first_inst = i0; continue; }
        HIP_TRY(h, hipStreamWaitEvent(w.stream, h->fork, 0));
        w.first_inst = i0;
        int rc = enqueue_lane(h, w, sub_batch(b, i0, n), d_static, d_dyn, d_out + i0, w.stream);
        if (rc != FOT_OK) return rc;
        HIP_TRY(h, hipEventRecord(w.done, w.stream));
        HIP_TRY(h, hipStreamWaitEvent(user, w.done, 0));
        i0 += n;
    }
    h->lanes_used = lanes;
    h->last_valid = true;
    return order_end(h, user);
}

// Arms the record flags for the next synchronous call of n records (false: this call waits on the stream instead).
bool arm_records(fot_handle *h, int n)
{
    h->done_seq_armed = false;
    static const bool off = std::getenv("FOT_NO_RECORD_FLAGS") != nullptr;       // diagnostics scripts
    if (off || h->prof_on || n <= 0 || n > 64) return false;
    // only the single-lane path hands the flags to its kernels (the predicate of enqueue_plan): a call the lanes split
    // would raise none, and wait_records would spin out its 20 ms before falling back to the stream
    if (h->lanes_cfg > 1 && n >= FOT_SPLIT_MIN_INSTANCES * h->lanes_cfg / 2) return false;
    if (h->hDone.ensure(sizeof(int32_t) * 64) != hipSuccess) return false;
    if (++h->done_seq == 0) {                                    // (wrapped: no stale flag may equal the new number)
        std::memset(h->hDone.p, 0, sizeof(int32_t) * 64);
        h->done_seq = 1;
    }
    h->done_seq_armed = true;
    return true;
}

// Waits until the n records of the call armed above are in pinned memory; after 20 ms without them (a kernel that
// faulted, a path that raises no flags) the stream is synchronised instead -- never a hang here that the stream
// synchronisation would not have been.
int wait_records(fot_handle *h, int n, hipStream_t st)
{
    const bool armed = h->done_seq_armed;
    h->done_seq_armed = false;
    if (armed) {
        volatile const int32_t *flag = (volatile const int32_t *)h->hDone.p;
        const auto t0 = std::chrono::steady_clock::now();
        int i = 0, spins = 0;
        while (i < n) {
            if (flag[i] == h->done_seq) { ++i; continue; }
            __builtin_ia32_pause();
            if ((++spins & 1023) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.02) break;
        }
        if (i == n) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return FOT_OK;
        }
    }
    HIP_TRY(h, hipStreamSynchronize(st));
    return FOT_OK;
}

// lane and local index of global instance `inst` of the most recent plan call
Workspace *lane_of(fot_handle *h, int inst, int *local)
{
    for (int l = h->lanes_used - 1; l >= 0; --l)
        if (inst >= h->ws[l].first_inst && inst < h->ws[l].first_inst + h->ws[l].last.n_inst) {
            *local = inst - h->ws[l].first_inst;
            return &h->ws[l];
        }
    return nullptr;
}

}  // namespace

extern "C" {

const char *fot_version(void) { return "libfot 0.2 (gfx950, float64 lattice)"; }

int32_t fot_abi_info(int32_t cap, int32_t *out)
{
    const int32_t v[FOT_ABI_INFO_WORDS] = {
        FOT_ABI_VERSION,
        (int32_t)sizeof(fot_params), (int32_t)sizeof(fot_ego), (int32_t)sizeof(fot_overrides), (int32_t)sizeof(fot_result),
        (int32_t)sizeof(fot_batch), (int32_t)sizeof(fot_resample_params), (int32_t)sizeof(fot_safety),
        (int32_t)sizeof(fot_loop_frame), (int32_t)sizeof(fot_loop_request), (int32_t)sizeof(fot_wire_header),
        FOT_MAX_NT, FOT_MAX_CIRCLES, FOT_MAX_TI, FOT_MAX_TV, FOT_MAX_BRAKE, FOT_MAX_SAMPLES, FOT_MAX_PRED_LEN,
        FOT_PROFILE_KERNELS, FOT_MARGIN_GROUPS,
        (int32_t)sizeof(fot_loop_config), (int32_t)sizeof(fot_loop_step_out),
    };
    for (int i = 0; i < FOT_ABI_INFO_WORDS && i < cap && out; ++i) out[i] = v[i];
    return FOT_ABI_INFO_WORDS;
}

const char *fot_last_error(const fot_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fot_create(const fot_params *params, int device, fot_handle **out)
{
    if (!params || !out) return fail(nullptr, FOT_ERR_INVALID, "params/out is NULL");
    *out = nullptr;
    DevParams P;
    std::string err;
    int rc = build_dev_params(*params, P, err);
    if (rc != FOT_OK) return fail(nullptr, rc, err);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, FOT_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0) { e = hipGetDevice(&device); if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDevice"); }
    if (device >= n_dev) return fail(nullptr, FOT_ERR_INVALID, "device index out of range");
    fot_handle *h = new (std::nothrow) fot_handle();
    if (!h) return fail(nullptr, FOT_ERR_HIP, "out of host memory");
    h->device = device;
    if (const char *ev = std::getenv("FOT_LANES")) {             // tuning knob: 1 disables the split
        const int v = std::atoi(ev);
        if (v >= 1 && v <= FOT_LANES) h->lanes_cfg = v;
    }
    h->params = *params;
    h->P = P;
    auto bail = [&](hipError_t ee, const char *what) {
        int r = hip_fail(nullptr, ee, what);
        destroy_handle(h);
        return r;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipEventCreateWithFlags(&h->fork, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&h->order_done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    for (int l = 0; l < FOT_LANES; ++l) {
        Workspace &w = h->ws[l];
        // lane streams only when the handle splits batches; lane 0 of a single-lane handle runs on the caller's stream
        if (h->lanes_cfg > 1 && l < h->lanes_cfg &&
            (e = hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
        if (l >= h->lanes_cfg) continue;
        for (int i = 0; i < 2; ++i)
            if ((e = hipEventCreateWithFlags(&w.staging_done[i], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&w.done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    if ((e = h->dP.ensure(sizeof(DevParams))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemcpy(h->dP.p, &h->P, sizeof(DevParams), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    {
        int cut = TILE_CUT_AUTO;                                 // FOT_TILE_CUT=wave|group: diagnostics scripts
        if (const char *ev = std::getenv("FOT_TILE_CUT")) cut = ev[0] == 'g' ? TILE_CUT_GROUP : ev[0] == 'w' ? TILE_CUT_WAVE : TILE_CUT_AUTO;
        if (upload_tile_shapes(h, cut) != FOT_OK) { std::string m = h->err; destroy_handle(h); return fail(nullptr, FOT_ERR_HIP, m); }
    }
    { std::lock_guard<std::mutex> lk(g_live_mu); live_handles().insert(h); }
    *out = h;
    return FOT_OK;
}

int32_t fot_live_handles(void)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    return (int32_t)live_handles().size();
}

void fot_destroy(fot_handle *h)
{
    if (!h) return;
    {   // idempotent: only a handle fot_create handed out and nobody has destroyed yet (the pointer is not even read else)
        std::lock_guard<std::mutex> lk(g_live_mu);
        auto &live = live_handles();
        auto it = live.find(h);
        if (it == live.end()) return;
        live.erase(it);
    }
    destroy_handle(h);
}

}  // extern "C"

namespace {

void destroy_handle(fot_handle *h)
{
    const double budget = destroy_timeout_s();
    bool quiet = hipSetDevice(h->device) == hipSuccess;
    // the handle's own stream, the lanes' streams, and the event behind its last enqueue (which may be on a caller's
    // stream): polled, never waited on
    if (quiet && h->stream) quiet = drained([&] { return hipStreamQuery(h->stream); }, budget);
    if (quiet && h->order_valid) quiet = drained([&] { return hipEventQuery(h->order_done); }, budget);
    for (Workspace &w : h->ws) if (quiet && w.stream) quiet = drained([&] { return hipStreamQuery(w.stream); }, budget);
    if (!quiet) {
        // something is still running (or the runtime is already gone): the host-side struct goes, everything the device
        // may still touch stays until the process ends
        (void)hipGetLastError();
        delete h;
        return;
    }
    DevBuf *bufs[] = { &h->dP, &h->dSpline, &h->dShapes, &h->dUserStatic, &h->dUserDyn, &h->dOut, &h->dTmpA, &h->dTmpB, &h->dTmpC, &h->dTmpD
                     };
    for (DevBuf *b : bufs) b->release();
    for (Workspace &w : h->ws) w.release();
    h->hSmallIn.release(); h->hSmallOut.release(); h->hRecOut.release(); h->hDone.release();
    h->loop.release();
    for (hipEvent_t e : h->prof_pool) (void)hipEventDestroy(e);
    if (h->fork) (void)hipEventDestroy(h->fork);
    if (h->order_done) (void)hipEventDestroy(h->order_done);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

}  // namespace

extern "C" {

int fot_set_path_waypoints(fot_handle *h, int32_t n, const double *wx, const double *wy)
{
    if (!h) return FOT_ERR_INVALID;
    std::string err;
    HostSpline sp;
    int rc = build_spline(n, wx, wy, sp, err);
    if (rc != FOT_OK) return fail(h, rc, err);
    h->spline = sp;
    return upload_spline(h);
}

int fot_set_path_coeffs(fot_handle *h, int32_t n, const double *s,
                        const double *ax, const double *bx, const double *cx, const double *dx,
                        const double *ay, const double *by, const double *cy, const double *dy)
{
    if (!h) return FOT_ERR_INVALID;
    if (n < 2 || !s || !ax || !bx || !cx || !dx || !ay || !by || !cy || !dy)
        return fail(h, FOT_ERR_INVALID, "spline needs >= 2 knots and all nine coefficient arrays");
    HostSpline &sp = h->spline;
    sp.n = n;
    sp.s.assign(s, s + n);
    sp.ax.assign(ax, ax + n); sp.bx.assign(bx, bx + n - 1); sp.cx.assign(cx, cx + n); sp.dx.assign(dx, dx + n - 1);
    sp.ay.assign(ay, ay + n); sp.by.assign(by, by + n - 1); sp.cy.assign(cy, cy + n); sp.dy.assign(dy, dy + n - 1);
    sp.bx.resize(n, 0.0); sp.dx.resize(n, 0.0); sp.by.resize(n, 0.0); sp.dy.resize(n, 0.0);
    return upload_spline(h);
}

int fot_get_path_coeffs(const fot_handle *h, int32_t *n_out, double *s,
                        double *ax, double *bx, double *cx, double *dx,
                        double *ay, double *by, double *cy, double *dy)
{
    if (!h || !h->has_path) return FOT_ERR_NO_PATH_SET;
    const HostSpline &sp = h->spline;
    const int n = sp.n;
    if (n_out) *n_out = n;
    auto cp = [](double *dst, const std::vector<double> &src, int cnt) {
        if (dst) std::memcpy(dst, src.data(), sizeof(double) * (size_t)cnt);
    };
    cp(s, sp.s, n);
    cp(ax, sp.ax, n); cp(bx, sp.bx, n - 1); cp(cx, sp.cx, n); cp(dx, sp.dx, n - 1);
    cp(ay, sp.ay, n); cp(by, sp.by, n - 1); cp(cy, sp.cy, n); cp(dy, sp.dy, n - 1);
    return FOT_OK;
}

int fot_spline_eval(fot_handle *h, int32_t n, const double *s, double *x, double *y,
                    double *yaw, double *kappa, double *dkappa)
{
    if (!h) return FOT_ERR_INVALID;
    if (!h->has_path) return fail(h, FOT_ERR_NO_PATH_SET, "fot_set_path_* has not been called");
    if (n <= 0) return FOT_OK;
    if (!s) return fail(h, FOT_ERR_INVALID, "s is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    { int r = order_begin(h, h->stream); if (r != FOT_OK) return r; }
    HIP_TRY(h, h->dTmpA.ensure(sizeof(double) * (size_t)n));
    HIP_TRY(h, h->dTmpB.ensure(sizeof(double) * 5 * (size_t)n));
    HIP_TRY(h, hipMemcpyAsync(h->dTmpA.p, s, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    LAUNCH_TRY(h, launch_spline_eval(spline_view(h), n, h->dTmpA.as<double>(), h->dTmpB.as<double>(), h->stream));
    std::vector<double> outv((size_t)5 * n);
    HIP_TRY(h, hipMemcpyAsync(outv.data(), h->dTmpB.p, sizeof(double) * 5 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double *dst[5] = { x, y, yaw, kappa, dkappa };
    for (int f = 0; f < 5; ++f)
        if (dst[f]) std::memcpy(dst[f], outv.data() + (size_t)f * n, sizeof(double) * (size_t)n);
    return FOT_OK;
}

int fot_plan_batch_device(fot_handle *h, const fot_batch *batch, fot_result *out_dev, void *stream)
{
    if (!h) return FOT_ERR_INVALID;
    if (!batch) return fail(h, FOT_ERR_INVALID, "batch is NULL");
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    return enqueue_plan(h, *batch, batch->static_xy, batch->dyn_xy, out_dev, st);
}

int fot_resample_n_dense(const fot_resample_params *rp, int32_t pred_len)
{
    if (!rp || !(rp->sim_dt > 0.0) || !(rp->sgan_dt > 0.0) || pred_len < 1) return FOT_ERR_INVALID;
    return resample_n_dense(rp->sgan_dt, rp->sim_dt, rp->plan_horizon, pred_len);
}

namespace {

// shared body of fot_resample_predictions (cv = 0) and fot_predict_cv (cv = 1: anchor = obs_last, pred = obs_prev)
int resample_common(fot_handle *h, const fot_resample_params *rp, int cv, int32_t S, int32_t pred_len, int32_t P,
                    const void *pred, int32_t pred_dtype, const double *anchor, const double *current,
                    double staleness, void *out, int32_t out_dtype, int32_t on_device, int32_t *T_out,
                    double *sample_dist, void *stream)
{
    if (!h) return FOT_ERR_INVALID;
    if (!rp || !(rp->sim_dt > 0.0) || !(rp->sgan_dt > 0.0)) return fail(h, FOT_ERR_INVALID, "sgan_dt and sim_dt must be positive");
    if (S < 0 || P < 0 || pred_len < 1) return fail(h, FOT_ERR_INVALID, "bad S / P / pred_len");
    if (pred_len > FOT_MAX_PRED_LEN) return fail(h, FOT_ERR_UNSUPPORTED, "pred_len > FOT_MAX_PRED_LEN");
    if ((pred_dtype != FOT_F32 && pred_dtype != FOT_F64) || (out_dtype != FOT_F32 && out_dtype != FOT_F64))
        return fail(h, FOT_ERR_INVALID, "dtype");
    const int n_dense = resample_n_dense(rp->sgan_dt, rp->sim_dt, rp->plan_horizon, pred_len);
    const int T = n_dense + (current ? 1 : 0);
    if (T > FOT_MAX_NT) return fail(h, FOT_ERR_UNSUPPORTED, "more than FOT_MAX_NT time steps");
    if (T_out) *T_out = T;
    const int tmajor = (on_device & FOT_OUT_TMAJOR) ? 1 : 0;
    on_device &= FOT_OUT_DEVICE;
    if (S == 0 || P == 0 || T == 0) return FOT_OK;
    if (!out || (!cv && !pred) || (cv && !anchor)) return fail(h, FOT_ERR_INVALID, "NULL tensor");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    // small per-pedestrian inputs: anchor | current  (and obs_prev for cv) through the handle's scratch
    const size_t row = sizeof(double) * 2 * (size_t)P;
    { int r = order_begin(h, st); if (r != FOT_OK) return r; }  // the scratch below is shared by every entry point
    {
        const size_t in_elem_ = pred_dtype == FOT_F32 ? 4 : 8, out_elem_ = out_dtype == FOT_F32 ? 4 : 8;
        const size_t in_b = cv ? 0 : in_elem_ * 2 * (size_t)P * (size_t)S * pred_len;
        const size_t out_b = out_elem_ * 2 * (size_t)S * P * T;
        if (!on_device && 3 * align256(row) + align256(in_b) <= SMALL_CALL_BYTES &&
            out_b + sizeof(double) * (size_t)S <= SMALL_CALL_BYTES) {
            // small host call: everything through pinned memory the kernels access directly
            HIP_TRY(h, h->hSmallIn.ensure(3 * align256(row) + align256(in_b)));
            HIP_TRY(h, h->hSmallOut.ensure(align256(out_b) + sizeof(double) * (size_t)S));
            char *p = (char *)h->hSmallIn.p;
            const double *pa = nullptr, *pc = nullptr;
            const void *pp = nullptr;
            if (anchor) { std::memcpy(p, anchor, row); pa = (const double *)p; }
            if (current) { std::memcpy(p + align256(row), current, row); pc = (const double *)(p + align256(row)); }
            if (cv) { if (pred) { std::memcpy(p + 2 * align256(row), pred, row); pp = p + 2 * align256(row); } }
            else { std::memcpy(p + 3 * align256(row), pred, in_b); pp = p + 3 * align256(row); }
            char *po = (char *)h->hSmallOut.p;
            LAUNCH_TRY(h, launch_resample(rp->sgan_dt, rp->sim_dt, staleness, S, pred_len, P, n_dense, anchor ? 1 : 0,
                                          current ? 1 : 0, cv, pp, cv ? FOT_F64 : pred_dtype, pa, pc, po, out_dtype,
                                          tmajor, st));
            if (sample_dist)
                LAUNCH_TRY(h, launch_sample_dist(S, P, T, current ? 1 : 0, po, out_dtype, tmajor,
                                                 (double *)(po + align256(out_b)), st));
            { int r = order_end(h, st); if (r != FOT_OK) return r; }
            HIP_TRY(h, hipStreamSynchronize(st));
            std::memcpy(out, po, out_b);
            if (sample_dist) std::memcpy(sample_dist, po + align256(out_b), sizeof(double) * (size_t)S);
            return FOT_OK;
        }
    }
    HIP_TRY(h, h->dTmpA.ensure(3 * row + 64));
    char *scr = (char *)h->dTmpA.p;
    const double *d_anchor = nullptr, *d_current = nullptr;
    const void *d_pred = pred;
    if (anchor) { HIP_TRY(h, hipMemcpyAsync(scr, anchor, row, hipMemcpyHostToDevice, st)); d_anchor = (const double *)scr; }
    if (current) { HIP_TRY(h, hipMemcpyAsync(scr + row, current, row, hipMemcpyHostToDevice, st)); d_current = (const double *)(scr + row); }
    void *d_out = out;
    const size_t in_elem = pred_dtype == FOT_F32 ? 4 : 8, out_elem = out_dtype == FOT_F32 ? 4 : 8;
    const size_t in_bytes = in_elem * 2 * (size_t)P * (cv ? 1 : (size_t)S * pred_len);
    const size_t out_bytes = out_elem * 2 * (size_t)S * P * T;
    if (cv) {                                                      // obs_prev is a host array of doubles
        d_pred = nullptr;
        if (pred) { HIP_TRY(h, hipMemcpyAsync(scr + 2 * row, pred, row, hipMemcpyHostToDevice, st)); d_pred = scr + 2 * row; }
    } else if (!on_device) {
        HIP_TRY(h, h->dTmpB.ensure(in_bytes));
        HIP_TRY(h, hipMemcpyAsync(h->dTmpB.p, pred, in_bytes, hipMemcpyHostToDevice, st));
        d_pred = h->dTmpB.p;
    }
    if (!on_device) { HIP_TRY(h, h->dTmpC.ensure(out_bytes)); d_out = h->dTmpC.p; }
    LAUNCH_TRY(h, launch_resample(rp->sgan_dt, rp->sim_dt, staleness, S, pred_len, P, n_dense, anchor ? 1 : 0,
                                  current ? 1 : 0, cv, d_pred, cv ? FOT_F64 : pred_dtype, d_anchor, d_current, d_out,
                                  out_dtype, tmajor, st));
    if (sample_dist) {
        HIP_TRY(h, h->dTmpD.ensure(sizeof(double) * (size_t)S));
        LAUNCH_TRY(h, launch_sample_dist(S, P, T, current ? 1 : 0, d_out, out_dtype, tmajor, h->dTmpD.as<double>(), st));
        HIP_TRY(h, hipMemcpyAsync(sample_dist, h->dTmpD.p, sizeof(double) * (size_t)S, hipMemcpyDeviceToHost, st));
    }
    if (!on_device) HIP_TRY(h, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    { int r = order_end(h, st); if (r != FOT_OK) return r; }
    if (!on_device || sample_dist) HIP_TRY(h, hipStreamSynchronize(st));
    return FOT_OK;
}

}  // namespace

int fot_resample_predictions(fot_handle *h, const fot_resample_params *rp, int32_t S, int32_t pred_len, int32_t P,
                             const void *pred, int32_t pred_dtype, const double *anchor, const double *current,
                             double staleness, void *out, int32_t out_dtype, int32_t on_device, int32_t *T_out,
                             double *sample_dist, void *stream)
{
    return resample_common(h, rp, 0, S, pred_len, P, pred, pred_dtype, anchor, current, staleness, out, out_dtype,
                           on_device, T_out, sample_dist, stream);
}

int fot_predict_cv(fot_handle *h, const fot_resample_params *rp, int32_t pred_len, int32_t P,
                   const void *obs_last, const void *obs_prev, int32_t obs_dtype, const double *current,
                   double staleness, void *out, int32_t out_dtype, int32_t on_device, int32_t *T_out, void *stream)
{
    if (!h) return FOT_ERR_INVALID;
    if (obs_dtype == FOT_F64)
        return resample_common(h, rp, 1, 1, pred_len, P, obs_prev, FOT_F64, (const double *)obs_last, current, staleness,
                               out, out_dtype, on_device, T_out, nullptr, stream);
    if (obs_dtype != FOT_F32) return fail(h, FOT_ERR_INVALID, "obs_dtype");
    // float32 observations travel widened (exact); cv mode 2 forms the velocity in float32
    const size_t n = 2 * (size_t)(P > 0 ? P : 0);
    std::vector<double> last(n), prev(n);
    for (size_t i = 0; i < n && obs_last; ++i) last[i] = (double)((const float *)obs_last)[i];
    for (size_t i = 0; i < n && obs_prev; ++i) prev[i] = (double)((const float *)obs_prev)[i];
    return resample_common(h, rp, 2, 1, pred_len, P, obs_prev ? prev.data() : nullptr, FOT_F64,
                           obs_last ? last.data() : nullptr, current, staleness, out, out_dtype, on_device, T_out,
                           nullptr, stream);
}

int fot_safety_metrics_batch(fot_handle *h, int32_t n, const double *ego, const int32_t *ped_off,
                             const double *ped_pos, const double *ped_vel, double ego_radius, double ped_radius,
                             int32_t use_footprint, fot_safety *out)
{
    if (!h) return FOT_ERR_INVALID;
    if (n <= 0) return n == 0 ? FOT_OK : fail(h, FOT_ERR_INVALID, "n < 0");
    if (!ego || !ped_off || !out) return fail(h, FOT_ERR_INVALID, "NULL array");
    for (int i = 0; i < n; ++i)
        if (ped_off[i + 1] < ped_off[i] || ped_off[0] < 0) return fail(h, FOT_ERR_INVALID, "ped_off must be non-decreasing");
    const size_t n_ped = (size_t)ped_off[n];
    if (n_ped > 0 && (!ped_pos || !ped_vel)) return fail(h, FOT_ERR_INVALID, "NULL pedestrian array");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    { int r = order_begin(h, st); if (r != FOT_OK) return r; }
    const size_t ego_b = sizeof(double) * 4 * (size_t)n, off_b = align256(sizeof(int32_t) * ((size_t)n + 1));
    const size_t ped_b = sizeof(double) * 2 * std::max<size_t>(n_ped, 1);
    if (align256(ego_b) + off_b + 2 * align256(ped_b) <= SMALL_CALL_BYTES && sizeof(fot_safety) * (size_t)n <= SMALL_CALL_BYTES) {
        // small call: the kernel reads its inputs from, and writes its results to, pinned host memory (no copy operations)
        HIP_TRY(h, h->hSmallIn.ensure(align256(ego_b) + off_b + 2 * align256(ped_b)));
        HIP_TRY(h, h->hSmallOut.ensure(sizeof(fot_safety) * (size_t)n));
        char *p = (char *)h->hSmallIn.p;
        char *p_off = p + align256(ego_b), *p_pos = p_off + off_b, *p_vel = p_pos + align256(ped_b);
        std::memcpy(p, ego, ego_b);
        std::memcpy(p_off, ped_off, sizeof(int32_t) * ((size_t)n + 1));
        if (n_ped) { std::memcpy(p_pos, ped_pos, sizeof(double) * 2 * n_ped); std::memcpy(p_vel, ped_vel, sizeof(double) * 2 * n_ped); }
        LAUNCH_TRY(h, launch_safety(h->dP.as<DevParams>(), n, (const double *)p, (const int32_t *)p_off, (const double *)p_pos,
                                    (const double *)p_vel, ego_radius, ped_radius, h->params.footprint_radius,
                                    use_footprint, (fot_safety *)h->hSmallOut.p, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        std::memcpy(out, h->hSmallOut.p, sizeof(fot_safety) * (size_t)n);
        return FOT_OK;
    }
    HIP_TRY(h, h->dTmpA.ensure(align256(ego_b) + off_b));
    HIP_TRY(h, h->dTmpB.ensure(2 * align256(ped_b)));
    HIP_TRY(h, h->dTmpC.ensure(sizeof(fot_safety) * (size_t)n));
    char *a = (char *)h->dTmpA.p, *b = (char *)h->dTmpB.p;
    HIP_TRY(h, hipMemcpyAsync(a, ego, ego_b, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(a + align256(ego_b), ped_off, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, st));
    if (n_ped) {
        HIP_TRY(h, hipMemcpyAsync(b, ped_pos, sizeof(double) * 2 * n_ped, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(b + align256(ped_b), ped_vel, sizeof(double) * 2 * n_ped, hipMemcpyHostToDevice, st));
    }
    LAUNCH_TRY(h, launch_safety(h->dP.as<DevParams>(), n, (const double *)a, (const int32_t *)(a + align256(ego_b)),
                                (const double *)b, (const double *)(b + align256(ped_b)), ego_radius, ped_radius,
                                h->params.footprint_radius, use_footprint, h->dTmpC.as<fot_safety>(), st));
    HIP_TRY(h, hipMemcpyAsync(out, h->dTmpC.p, sizeof(fot_safety) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return FOT_OK;
}

int fot_loop_set_static(fot_handle *h, int32_t n_points, const double *xy)
{
    if (!h) return FOT_ERR_INVALID;
    if (n_points < 0 || (n_points > 0 && !xy)) return fail(h, FOT_ERR_INVALID, "static points");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));                 // nothing may still be reading the old copies
    h->loop.static_xy.assign(xy, xy + 2 * (size_t)n_points);
    h->loop.static_tiles = 0;
    return FOT_OK;
}

}  // extern "C"

namespace {

// fot_loop_plan; rec_first: the request's records start at record rec_first of the handle's pinned block (the escalation
// levels of a step land behind its level-0 records, which stay where they are)
int loop_plan_impl(fot_handle *h, const fot_loop_frame *frame, int32_t n_req, const fot_loop_request *req,
                   fot_safety *safety_out, const fot_result **records, int32_t rec_first)
{
    if (!h) return FOT_ERR_INVALID;
    if (!h->has_path) return fail(h, FOT_ERR_NO_PATH_SET, "fot_set_path_* has not been called");
    if (n_req < 0 || (n_req > 0 && (!req || !records))) return fail(h, FOT_ERR_INVALID, "requests");
    LoopState &L = h->loop;
    if (!frame && !L.have_frame) return fail(h, FOT_ERR_INVALID, "no frame: the first fot_loop_plan of a step carries one");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    if (L.observe_n > 0) {                                       // a fot_loop_observe_begin nobody collected: its launches
        HIP_TRY(h, hipStreamSynchronize(st));                    // still read the frame's pinned block
        L.observe_n = -1;
    }
    { int r = order_begin(h, st); if (r != FOT_OK) return r; }
    bool metrics = false;
    if (frame) {
        const int n = frame->n_episodes;
        if (n < 0 || !frame->ped_off) return fail(h, FOT_ERR_INVALID, "frame: n_episodes / ped_off (one offset even for no episode)");
        L.have_frame = false;
        for (int i = 0; i < n; ++i)
            if (frame->ped_off[i + 1] < frame->ped_off[i] || frame->ped_off[0] != 0)
                return fail(h, FOT_ERR_INVALID, "ped_off must start at 0 and be non-decreasing");
        const size_t n_ped = n > 0 ? (size_t)frame->ped_off[n] : 0;
        if (n_ped > 0 && (!frame->ped_pos || !frame->ped_vel)) return fail(h, FOT_ERR_INVALID, "frame: NULL pedestrian array");
        const bool ready = frame->obs_last != nullptr;
        const bool dist = ready && frame->dist_raw != nullptr;
        if (dist && (frame->dist_S < 1 || frame->dist_S > FOT_MAX_SAMPLES || (frame->dist_dtype != FOT_F32 && frame->dist_dtype != FOT_F64)))
            return fail(h, dist && frame->dist_S > FOT_MAX_SAMPLES ? FOT_ERR_UNSUPPORTED : FOT_ERR_INVALID, "frame: dist_S / dist_dtype");
        if (ready && !dist && n > 0 && !frame->prepend) return fail(h, FOT_ERR_INVALID, "frame: prepend missing");
        if (ready && (!(frame->rp.sim_dt > 0.0) || !(frame->rp.sgan_dt > 0.0) || frame->pred_len < 1 ||
                      frame->pred_len > FOT_MAX_PRED_LEN))
            return fail(h, FOT_ERR_INVALID, "frame: predictor parameters");
        const int n_dense = ready ? resample_n_dense(frame->rp.sgan_dt, frame->rp.sim_dt, frame->rp.plan_horizon, frame->pred_len) : 1;
        if (n_dense + 1 > FOT_MAX_NT) return fail(h, FOT_ERR_UNSUPPORTED, "more than FOT_MAX_NT time steps");
        // pinned block: ego[n][4] | ped_off | pos | vel | obs_last (widened) | obs_prev (widened) | block offsets | ped -> episode
        const size_t ego_b = align256(sizeof(double) * 4 * (size_t)std::max(n, 1));
        const size_t off_b = align256(sizeof(int32_t) * ((size_t)n + 1));
        const size_t ped_b = align256(sizeof(double) * 2 * std::max<size_t>(n_ped, 1));
        const size_t blk_b = align256(sizeof(int64_t) * ((size_t)n + 1)), pe_b = align256(sizeof(int32_t) * std::max<size_t>(n_ped, 1));
        HIP_TRY(h, L.hFrame.ensure(ego_b + off_b + 4 * ped_b + blk_b + pe_b));
        char *p = (char *)L.hFrame.p;
        double *p_ego = (double *)p;
        int32_t *p_off = (int32_t *)(p + ego_b);
        double *p_pos = (double *)(p + ego_b + off_b), *p_vel = (double *)((char *)p_pos + ped_b);
        double *p_last = (double *)((char *)p_vel + ped_b), *p_prev = (double *)((char *)p_last + ped_b);
        if (frame->ego) std::memcpy(p_ego, frame->ego, sizeof(double) * 4 * (size_t)n);
        std::memcpy(p_off, frame->ped_off, sizeof(int32_t) * ((size_t)n + 1));
        if (n_ped) {
            std::memcpy(p_pos, frame->ped_pos, sizeof(double) * 2 * n_ped);
            std::memcpy(p_vel, frame->ped_vel, sizeof(double) * 2 * n_ped);
            if (ready) for (size_t i = 0; i < 2 * n_ped; ++i) p_last[i] = (double)frame->obs_last[i];   // (exact)
            if (ready && frame->obs_prev) for (size_t i = 0; i < 2 * n_ped; ++i) p_prev[i] = (double)frame->obs_prev[i];
        }
        L.ped_off.assign(frame->ped_off, frame->ped_off + n + 1);
        L.blk_off.assign((size_t)n + 1, 0);
        L.t_len.assign((size_t)std::max(n, 1), 1);
        L.dist_S = dist ? frame->dist_S : 0;
        for (int e = 0; e < n; ++e) {
            L.t_len[e] = dist ? n_dense + 1 : ready ? n_dense + (frame->prepend[e] ? 1 : 0) : 1;
            L.blk_off[e + 1] = L.blk_off[e] + (int64_t)(dist ? frame->dist_S : 1) * (L.ped_off[e + 1] - L.ped_off[e]) * L.t_len[e];
        }
        L.p_off = p_off; L.p_pos = p_pos; L.p_vel = p_vel;
        L.ego_radius = frame->ego_radius; L.ped_radius = frame->ped_radius; L.use_footprint = frame->use_footprint;
        L.dyn_ptr = p_pos;                                         // not ready: the current positions, read in place
        if (dist && n_ped > 0) {
            // every sample of every pedestrian in one launch, each episode's samples into its own [S][P_e][T][2] block
            int64_t *p_blk = (int64_t *)((char *)p_prev + ped_b);
            int32_t *p_pe = (int32_t *)((char *)p_blk + blk_b);
            for (int e = 0; e <= n; ++e) p_blk[e] = L.blk_off[e];
            for (int e = 0; e < n; ++e) for (int q = L.ped_off[e]; q < L.ped_off[e + 1]; ++q) p_pe[q] = e;
            HIP_TRY(h, L.dDyn.ensure(sizeof(double) * 2 * (size_t)L.blk_off[n]));
            L.dyn_ptr = L.dDyn.p;
            LAUNCH_TRY(h, launch_resample(frame->rp.sgan_dt, frame->rp.sim_dt, frame->staleness, frame->dist_S, frame->pred_len,
                                          (int)n_ped, n_dense, 1, 1, 0, frame->dist_raw, frame->dist_dtype, p_last, p_pos,
                                          L.dDyn.p, FOT_F64, 0, st, p_pe, p_off, p_blk));
        } else if (ready && n_ped > 0) {
            HIP_TRY(h, L.dDyn.ensure(sizeof(double) * 2 * (size_t)L.blk_off[n]));
            L.dyn_ptr = L.dDyn.p;
            // one launch per run of episodes that agree on the prepend (normally one run: the whole frame)
            for (int e0 = 0; e0 < n;) {
                int e1 = e0 + 1;
                while (e1 < n && (frame->prepend[e1] != 0) == (frame->prepend[e0] != 0)) ++e1;
                const int r0 = L.ped_off[e0], cnt = L.ped_off[e1] - r0, pre = frame->prepend[e0] ? 1 : 0;
                if (cnt > 0)
                    LAUNCH_TRY(h, launch_resample(frame->rp.sgan_dt, frame->rp.sim_dt, frame->staleness, 1, frame->pred_len,
                                                  cnt, n_dense, 1, pre, 2, frame->obs_prev ? p_prev + 2 * (size_t)r0 : nullptr,
                                                  FOT_F64, p_last + 2 * (size_t)r0, pre ? p_pos + 2 * (size_t)r0 : nullptr,
                                                  L.dDyn.as<double>() + 2 * L.blk_off[e0], FOT_F64, 0, st));
                e0 = e1;
            }
        }
        if (frame->ego && safety_out && n > 0) {
            HIP_TRY(h, L.hOut.ensure(sizeof(fot_safety) * (size_t)n));
            LAUNCH_TRY(h, launch_safety(h->dP.as<DevParams>(), n, p_ego, p_off, p_pos, p_vel, L.ego_radius, L.ped_radius,
                                        h->params.footprint_radius, L.use_footprint, (fot_safety *)L.hOut.p, st));
            metrics = true;
        }
        L.have_frame = true;
    }
    const int n_ep = (int)L.ped_off.size() - 1;
    if (n_req > 0) {
        const int n_static = (int)(L.static_xy.size() / 2);
        if (n_static > 0 && L.static_tiles < n_req) {              // (grows a few times in the life of a loop)
            const int tiles = std::max(n_req, 2 * L.static_tiles);
            std::vector<double> rep((size_t)tiles * L.static_xy.size());
            for (int t = 0; t < tiles; ++t)
                std::memcpy(rep.data() + (size_t)t * L.static_xy.size(), L.static_xy.data(), sizeof(double) * L.static_xy.size());
            HIP_TRY(h, hipStreamSynchronize(st));
            HIP_TRY(h, L.dStatic.ensure(sizeof(double) * rep.size()));
            HIP_TRY(h, hipMemcpy(L.dStatic.p, rep.data(), sizeof(double) * rep.size(), hipMemcpyHostToDevice));
            L.static_tiles = tiles;
        }
        std::vector<fot_ego> ego((size_t)n_req);
        std::vector<fot_overrides> ov((size_t)n_req);
        std::vector<double> tgt((size_t)n_req), stop((size_t)n_req);
        std::vector<int32_t> s_off((size_t)n_req + 1), dims(4 * (size_t)n_req);
        std::vector<int64_t> d_off((size_t)n_req);
        bool any_dyn = false;
        for (int j = 0; j < n_req; ++j) {
            const int e = req[j].episode;
            if (e < 0 || e >= n_ep) return fail(h, FOT_ERR_INVALID, "request: episode out of range");
            ego[j] = req[j].ego; ov[j] = req[j].overrides; tgt[j] = req[j].target_speed; stop[j] = req[j].max_stop_distance;
            s_off[j] = j * n_static;
            const int P_e = L.ped_off[e + 1] - L.ped_off[e];
            d_off[j] = L.blk_off[e];
            dims[4 * j] = P_e > 0 ? (L.dist_S > 0 ? FOT_DYN_DISTRIBUTION : FOT_DYN_SINGLE) : FOT_DYN_NONE;
            dims[4 * j + 1] = L.dist_S > 0 ? L.dist_S : 1; dims[4 * j + 2] = P_e; dims[4 * j + 3] = L.t_len[e];
            any_dyn = any_dyn || P_e > 0;
        }
        s_off[n_req] = n_req * n_static;
        fot_batch b = fot_batch();
        b.n_inst = n_req; b.obstacle_dtype = FOT_F64;
        b.ego = ego.data(); b.target_speed = tgt.data(); b.overrides = ov.data(); b.max_stop_distance = stop.data();
        if (n_static > 0) { b.static_xy = L.dStatic.p; b.static_off = s_off.data(); }
        if (any_dyn) { b.dyn_xy = L.dyn_ptr; b.dyn_off = d_off.data(); b.dyn_dims = dims.data(); }
        if (rec_first > 0 && sizeof(fot_result) * ((size_t)rec_first + (size_t)n_req) > L.hRec.cap)
            return fail(h, FOT_ERR_INVALID, "internal: record block too small for the escalation levels");
        HIP_TRY(h, L.hRec.ensure(sizeof(fot_result) * ((size_t)rec_first + (size_t)n_req)));
        fot_result *rec_out = (fot_result *)L.hRec.p + rec_first;
        {
            // the records' flags instead of the stream (wait_records): the metrics' kernel ran ahead of the plan kernels on
            // this stream and wrote host memory directly, so its results are there once a record behind it is
            arm_records(h, n_req);
            int rc = enqueue_plan(h, b, b.static_xy, b.dyn_xy, rec_out, st, true);
            if (rc != FOT_OK) { h->done_seq_armed = false; return rc; }
            rc = wait_records(h, n_req, st);
            if (rc != FOT_OK) return rc;
        }
    } else {
        int r = order_end(h, st); if (r != FOT_OK) return r;
        HIP_TRY(h, hipStreamSynchronize(st));
    }
    if (metrics) std::memcpy(safety_out, L.hOut.p, sizeof(fot_safety) * (size_t)n_ep);
    if (records) *records = n_req > 0 ? (const fot_result *)L.hRec.p + rec_first : nullptr;
    return FOT_OK;
}

}  // namespace

extern "C" {

int fot_loop_plan(fot_handle *h, const fot_loop_frame *frame, int32_t n_req, const fot_loop_request *req,
                  fot_safety *safety_out, const fot_result **records)
{
    return loop_plan_impl(h, frame, n_req, req, safety_out, records, 0);
}

int fot_loop_observe_begin(fot_handle *h, int32_t n, const double *ego5, const double *prev_s)
{
    if (!h) return FOT_ERR_INVALID;
    if (!h->has_path) return fail(h, FOT_ERR_NO_PATH_SET, "fot_set_path_* has not been called");
    LoopState &L = h->loop;
    L.observe_n = -1;
    if (!L.have_frame) return fail(h, FOT_ERR_INVALID, "no frame: fot_loop_plan of this step comes first");
    if (n != (int)L.ped_off.size() - 1) return fail(h, FOT_ERR_INVALID, "one ego per episode of the frame");
    if (n <= 0) { L.observe_n = 0; return FOT_OK; }
    if (!ego5) return fail(h, FOT_ERR_INVALID, "ego5 is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    { int r = order_begin(h, st); if (r != FOT_OK) return r; }
    const size_t ego_b = align256(sizeof(double) * 4 * (size_t)n), desc_b = align256(sizeof(InstDesc) * (size_t)n);
    const size_t saf_b = align256(sizeof(fot_safety) * (size_t)n);
    HIP_TRY(h, L.hObserve.ensure(ego_b + desc_b));
    HIP_TRY(h, L.hOut.ensure(saf_b + sizeof(InstState) * (size_t)n));
    double *p_ego = (double *)L.hObserve.p;
    InstDesc *p_desc = (InstDesc *)((char *)L.hObserve.p + ego_b);
    for (int i = 0; i < n; ++i) {
        const double *e = ego5 + 5 * (size_t)i;
        p_ego[4 * i] = e[0]; p_ego[4 * i + 1] = e[1]; p_ego[4 * i + 2] = e[2]; p_ego[4 * i + 3] = e[3];
        InstDesc d = InstDesc();
        d.ego.x = e[0]; d.ego.y = e[1]; d.ego.yaw = e[2]; d.ego.v = e[3]; d.ego.a = e[4];
        const bool cached = prev_s && !std::isnan(prev_s[i]);
        d.ego.has_prev_s = cached ? 1 : 0;
        d.ego.prev_s = cached ? prev_s[i] : 0.0;
        p_desc[i] = d;
    }
    LAUNCH_TRY(h, launch_safety(h->dP.as<DevParams>(), n, p_ego, L.p_off, L.p_pos, L.p_vel, L.ego_radius, L.ped_radius,
                                h->params.footprint_radius, L.use_footprint, (fot_safety *)L.hOut.p, st));
    InstState *p_state = (InstState *)((char *)L.hOut.p + saf_b);
    LAUNCH_TRY(h, launch_frenet_state(h->dP.as<DevParams>(), spline_view(h), p_desc, p_state, n, MetaImport(), NanScan(),
                                      nullptr, st));
    { int r = order_end(h, st); if (r != FOT_OK) return r; }
    L.observe_n = n;
    return FOT_OK;
}

int fot_loop_observe_end(fot_handle *h, fot_safety *safety_out, double *new_prev_s)
{
    if (!h) return FOT_ERR_INVALID;
    LoopState &L = h->loop;
    const int n = L.observe_n;
    if (n < 0) return fail(h, FOT_ERR_INVALID, "no fot_loop_observe_begin to collect");
    L.observe_n = -1;
    if (n == 0) return FOT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (safety_out) std::memcpy(safety_out, L.hOut.p, sizeof(fot_safety) * (size_t)n);
    const InstState *p_state = (const InstState *)((const char *)L.hOut.p + align256(sizeof(fot_safety) * (size_t)n));
    if (new_prev_s) for (int i = 0; i < n; ++i) new_prev_s[i] = p_state[i].new_prev_s;
    return FOT_OK;
}

int fot_loop_observe(fot_handle *h, int32_t n, const double *ego5, const double *prev_s,
                     fot_safety *safety_out, double *new_prev_s)
{
    int rc = fot_loop_observe_begin(h, n, ego5, prev_s);
    return rc != FOT_OK ? rc : fot_loop_observe_end(h, safety_out, new_prev_s);
}

}  // extern "C"

// ---- the whole lock step of n episodes behind ONE call (SURVEY 8 f2 + f4) -------------------------------------------
namespace {

// FailSafeStateMachine._get_planner_config (state_machine.py:181-247): the planner configuration of `state` on the
// clearance ahead the machine last observed -> target speed, constraint overrides (NaN = absent), stop room (NaN = None)
void sm_config(const fot_loop_config &c, int state, double clear_ahead, double *target, fot_overrides *ov, double *stop)
{
    const bool fin = std::isfinite(clear_ahead), has_env = fin && c.envelope_decel > 0.0;
    const double v_env = std::sqrt(2.0 * c.envelope_decel * std::fmax((fin ? clear_ahead : 0.0) - c.envelope_standoff, 0.0));
    const double stop_room = fin ? std::fmax(clear_ahead - 0.2, 0.05) : NAN;
    *target = c.target_speed;
    ov->max_speed = ov->max_accel = ov->max_curvature = ov->max_lat_accel = NAN;
    *stop = NAN;
    if (state == 0) {
        if (has_env && v_env < c.target_speed) *target = v_env;
    } else if (state == 1) {
        const double t_ca = c.target_speed * c.caution_speed_mult;
        *target = has_env ? std::fmin(t_ca, v_env) : t_ca;
        if (has_env && v_env <= 0.0) *stop = stop_room;
        ov->max_accel = c.caution_accel; ov->max_speed = c.caution_speed;
    } else {
        *target = 0.0;
        ov->max_accel = c.emergency_accel; ov->max_lat_accel = c.emergency_lat_accel;
        if (c.envelope_decel > 0.0) *stop = stop_room;
    }
}

// FailSafeStateMachine.update (state_machine.py:116-179) of episode e: observe the metrics, then the transition
void sm_update(LoopEpisodes &E, int e, bool found, double clearance, double clearance_ahead, double speed)
{
    const fot_loop_config &c = E.cfg;
    E.clear[e] = clearance; E.clear_ahead[e] = clearance_ahead;
    const int st = E.state[e], fl = E.fails[e];
    const double trigger = c.trigger_clearance_caution + c.trigger_time_headway * std::fmax(speed, 0.0);
    if (st == 0) {
        if (!found) { E.state[e] = 1; E.fails[e] = fl + 1; }
        else if (trigger > 0.0 && clearance < trigger) { E.state[e] = 1; E.fails[e] = 0; }   // preventive escalation
        else E.fails[e] = 0;
    } else if (st == 1) {
        if (found && fl == 0) { if (clearance > std::fmax(c.clearance_caution, trigger)) E.state[e] = 0; }
        else if (!found) { E.state[e] = 2; E.fails[e] = fl + 1; }
        else E.fails[e] = 0;
    } else {
        if (found && clearance > c.clearance_emergency) E.state[e] = 1;
    }
}

}  // namespace

extern "C" {

int fot_loop_begin(fot_handle *h, int32_t n_episodes, const fot_loop_config *cfg, const double *ego5)
{
    if (!h) return FOT_ERR_INVALID;
    if (n_episodes < 0 || !cfg || (n_episodes > 0 && !ego5)) return fail(h, FOT_ERR_INVALID, "fot_loop_begin: arguments");
    if (!(cfg->dt > 0.0) || cfg->max_replan < 0 || cfg->max_replan > 8) return fail(h, FOT_ERR_INVALID, "fot_loop_begin: dt / max_replan");
    LoopEpisodes &E = h->loop.ep;
    const size_t n = (size_t)n_episodes;
    E.n = n_episodes; E.cfg = *cfg;
    E.ego.assign(ego5, ego5 + 5 * n);
    E.prev_s.assign(n, NAN); E.last_kappa.assign(n, 0.0); E.goal_prev_s.assign(n, NAN);
    E.last_clearance.assign(n, INFINITY); E.clear.assign(n, INFINITY); E.clear_ahead.assign(n, INFINITY);
    E.state.assign(n, 0); E.fails.assign(n, 0); E.stats.assign(8 * n, -1);
    return FOT_OK;
}

int fot_loop_step(fot_handle *h, const fot_loop_frame *frame, const int32_t *episode, fot_loop_step_out *out)
{
    if (!h) return FOT_ERR_INVALID;
    if (!frame || !out) return fail(h, FOT_ERR_INVALID, "fot_loop_step: frame / out");
    LoopState &L = h->loop;
    LoopEpisodes &E = L.ep;
    const fot_loop_config &c = E.cfg;
    const int n = frame->n_episodes;
    if (n < 0 || (n > 0 && !episode)) return fail(h, FOT_ERR_INVALID, "fot_loop_step: episode");
    out->records = nullptr; out->n_records = 0;
    if (n == 0) return FOT_OK;
    std::vector<uint8_t> seen((size_t)std::max(E.n, 1), 0);
    for (int i = 0; i < n; ++i) {
        if (episode[i] < 0 || episode[i] >= E.n || seen[(size_t)episode[i]]) return fail(h, FOT_ERR_INVALID, "fot_loop_step: episode slots must be distinct and below fot_loop_begin's count");
        seen[(size_t)episode[i]] = 1;
    }
    const int max_lvl = 1 + c.max_replan < 3 ? 1 + c.max_replan : 3;     // NORMAL -> CAUTION -> EMERGENCY, then no change
    // --- level 0 of every episode: the configuration of its current state (issued on LAST step's clearance), with the
    //     frame's prediction and the metrics of the current ego states
    std::vector<int> st0((size_t)n), n_lvl((size_t)n);
    std::vector<double> speed((size_t)n), ego4(4 * (size_t)n);
    std::vector<fot_loop_request> req((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int e = episode[i];
        st0[i] = E.state[e];
        n_lvl[i] = std::min(3 - st0[i], max_lvl);
        const double *g = &E.ego[5 * (size_t)e];
        speed[i] = g[3];
        for (int k = 0; k < 4; ++k) ego4[4 * (size_t)i + k] = g[k];
        fot_loop_request &r = req[i];
        r = fot_loop_request();
        r.ego.x = g[0]; r.ego.y = g[1]; r.ego.yaw = g[2]; r.ego.v = g[3]; r.ego.a = g[4];
        r.ego.last_kappa = E.last_kappa[e];
        r.ego.has_prev_s = std::isnan(E.prev_s[e]) ? 0 : 1;
        r.ego.prev_s = std::isnan(E.prev_s[e]) ? 0.0 : E.prev_s[e];
        sm_config(c, st0[i], E.clear_ahead[e], &r.target_speed, &r.overrides, &r.max_stop_distance);
        r.episode = i;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    if (L.observe_n > 0) { HIP_TRY(h, hipStreamSynchronize(h->stream)); L.observe_n = -1; }
    // (all records of the step in one pinned block that must not move between the two plan calls)
    HIP_TRY(h, L.hRec.ensure(sizeof(fot_result) * (size_t)n * (size_t)max_lvl));
    fot_loop_frame fr = *frame;
    fr.ego = ego4.data();
    std::vector<fot_safety> m((size_t)n);
    const fot_result *rec = nullptr;
    { int rc = loop_plan_impl(h, &fr, n, req.data(), m.data(), &rec, 0); if (rc != FOT_OK) return rc; }
    rec = (const fot_result *)L.hRec.p;
    for (int i = 0; i < n; ++i) E.last_clearance[episode[i]] = m[i].clearance_ahead;
    // --- episodes whose first attempt failed: every further escalation level they can reach in ONE more launch (the
    //     configurations update(False, ...) would issue on THIS step's metrics, nearest-point cache chained)
    std::vector<int> next_rec((size_t)n, -1);
    std::vector<fot_loop_request> more;
    for (int i = 0; i < n; ++i) {
        if (rec[i].status == FOT_PLAN_OK || n_lvl[i] <= 1) continue;
        const int e = episode[i];
        next_rec[i] = n + (int)more.size();
        const double nps0 = rec[i].new_prev_s, p = std::isnan(nps0) ? E.prev_s[e] : nps0;
        for (int lvl = 1; lvl < n_lvl[i]; ++lvl) {
            fot_loop_request r = req[i];
            const bool chain = lvl > 1;
            r.ego.has_prev_s = chain ? FOT_PREV_S_CHAINED : (std::isnan(p) ? 0 : 1);
            r.ego.prev_s = (chain || std::isnan(p)) ? 0.0 : p;
            sm_config(c, st0[i] + lvl, m[i].clearance_ahead, &r.target_speed, &r.overrides, &r.max_stop_distance);
            more.push_back(r);
        }
    }
    int n_rec = n;
    if (!more.empty()) {
        const fot_result *unused = nullptr;
        int rc = loop_plan_impl(h, nullptr, (int)more.size(), more.data(), nullptr, &unused, n);
        if (rc != FOT_OK) return rc;
        rec = (const fot_result *)L.hRec.p;
        n_rec += (int)more.size();
    }
    // --- replay of the retry loop (integrated_simulator.py:576-653), episode by episode
    std::vector<int> path_rec((size_t)n, -1);
    auto adopt = [&](int i, int r) {                             // planner state after a plan() call
        const int e = episode[i];
        if (!std::isnan(rec[r].new_prev_s)) E.prev_s[e] = rec[r].new_prev_s;
        for (int k = 0; k < 8; ++k) E.stats[8 * (size_t)e + k] = rec[r].stats_valid ? rec[r].stats[k] : -1;
        if (rec[r].status == FOT_PLAN_OK) { E.last_kappa[e] = rec[r].new_last_kappa; path_rec[i] = r; }
    };
    for (int i = 0; i < n; ++i) {
        const int e = episode[i];
        int cur = i, retries = 0;
        adopt(i, cur);
        bool found = rec[cur].status == FOT_PLAN_OK;
        int issued = st0[i];                                      // state of the configuration the attempt ran under
        sm_update(E, e, found, m[i].clearance, m[i].clearance_ahead, speed[i]);
        while (!found && E.state[e] != issued && retries < c.max_replan && retries + 1 < n_lvl[i]) {
            cur = retries == 0 ? next_rec[i] : cur + 1;
            ++retries;
            adopt(i, cur);
            const bool ok = rec[cur].status == FOT_PLAN_OK;
            found = found || ok;
            issued = E.state[e];
            if (!ok) sm_update(E, e, false, m[i].clearance, m[i].clearance_ahead, speed[i]);
        }
    }
    // --- ego update (:655-676) or emergency stop (:749-802)
    std::vector<double> ego5n(5 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const int e = episode[i];
        double *g = &E.ego[5 * (size_t)e];
        const double old_a = g[4];
        const int r = path_rec[i];
        const int keep = r >= 0 ? rec[r].n_keep : 0;
        double jerk;
        if (keep >= 2) {
            g[0] = rec[r].x[1]; g[1] = rec[r].y[1]; g[2] = rec[r].yaw[1]; g[3] = rec[r].v[1]; g[4] = rec[r].a[1];
            jerk = (g[4] - old_a) / c.dt;
        } else {
            // the position integrates along the heading at the OLD speed; the deceleration is what stopping 0.2 m short
            // of the nearest pedestrian ahead needs, bounded to [max_accel, emergency_decel]
            const double v = g[3], clr = E.last_clearance[e];
            const double cap = std::isnan(c.emergency_decel) ? c.max_accel * 2.0 : c.emergency_decel;
            const double required = std::isfinite(clr) ? v * v / (2.0 * std::fmax(clr - 0.2, 0.05)) : cap;
            const double max_dec = std::fmin(std::fmax(required, c.max_accel), cap);
            const double nv = std::fmax(0.0, v - max_dec * c.dt), na = nv > 0.0 ? -max_dec : 0.0;
            g[0] = g[0] + v * std::cos(g[2]) * c.dt; g[1] = g[1] + v * std::sin(g[2]) * c.dt;
            g[3] = nv; g[4] = na;
            jerk = (na - old_a) / c.dt;
            E.last_kappa[e] = 0.0;                                // planner.reset_ego_curvature()
        }
        for (int k = 0; k < 5; ++k) ego5n[5 * (size_t)i + k] = g[k];
        if (out->ego) for (int k = 0; k < 5; ++k) out->ego[5 * (size_t)i + k] = g[k];
        if (out->jerk) out->jerk[i] = jerk;
        if (out->record) out->record[i] = r;
        if (out->keep) out->keep[i] = keep;
        if (out->cost) out->cost[i] = rec[r >= 0 ? r : 0].cost;
    }
    // --- result metrics on the new ego states and the goal test's nearest point (:864-883): enqueued, the rest of the
    //     outputs filled while they run
    std::vector<double> gps((size_t)n);
    for (int i = 0; i < n; ++i) gps[i] = E.goal_prev_s[episode[i]];
    { int rc = fot_loop_observe_begin(h, n, ego5n.data(), gps.data()); if (rc != FOT_OK) return rc; }
    for (int i = 0; i < n; ++i) {
        const int e = episode[i];
        if (out->state) out->state[i] = E.state[e];
        if (out->stats) for (int k = 0; k < 8; ++k) out->stats[8 * (size_t)i + k] = E.stats[8 * (size_t)e + k];
        if (out->before) out->before[i] = m[i];
    }
    std::vector<fot_safety> after((size_t)n);
    { int rc = fot_loop_observe_end(h, after.data(), gps.data()); if (rc != FOT_OK) return rc; }
    for (int i = 0; i < n; ++i) {
        E.goal_prev_s[episode[i]] = gps[i];
        if (out->after) out->after[i] = after[i];
        if (out->s_now) out->s_now[i] = gps[i];
    }
    out->records = rec;
    out->n_records = n_rec;
    return FOT_OK;
}

}  // extern "C"

extern "C" {

int fot_gather_paths(const fot_result *records, int32_t n, const int32_t *index, int32_t kmax, double *out)
{
    if (n < 0 || kmax < 0 || kmax > FOT_MAX_NT) return FOT_ERR_INVALID;
    if (n == 0 || kmax == 0) return FOT_OK;
    if (!records || !index || !out) return FOT_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (index[i] < 0) return FOT_ERR_INVALID;
        const double *src = records[index[i]].t;                 // the 15 arrays lie back to back (static_assert below)
        for (int f = 0; f < 15; ++f)
            std::memcpy(out + ((size_t)f * n + i) * kmax, src + (size_t)f * FOT_MAX_NT, sizeof(double) * (size_t)kmax);
    }
    return FOT_OK;
}

int fot_profile_enable(fot_handle *h, int on)
{
    if (!h) return FOT_ERR_INVALID;
    if (!on && h->prof_on) { int r = prof_drain(h); if (r != FOT_OK) return r; }
    h->prof_on = on != 0;
    return FOT_OK;
}

int fot_profile_read(fot_handle *h, int reset, int32_t cap, int32_t *launches, double *total_ms)
{
    if (!h) return FOT_ERR_INVALID;
    if (cap < 0) return fail(h, FOT_ERR_INVALID, "fot_profile_read: cap < 0");
    HIP_TRY(h, hipSetDevice(h->device));
    int r = prof_drain(h);
    if (r != FOT_OK) return r;
    for (int k = 0; k < FOT_PROFILE_KERNELS; ++k) {
        if (k < cap) {                                           // (never past the caller's arrays, whatever header it was built with)
            if (launches) launches[k] = h->prof_launches[k];
            if (total_ms) total_ms[k] = h->prof_ms[k];
        }
        if (reset) { h->prof_launches[k] = 0; h->prof_ms[k] = 0.0; }
    }
    return FOT_PROFILE_KERNELS;
}

const char *fot_profile_kernel_name(int index)
{
    return index >= 0 && index < FOT_PROFILE_KERNELS ? kKernelNames[index] : "";
}

int fot_synchronize(fot_handle *h)
{
    if (!h) return FOT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FOT_OK;
}

int fot_plan_batch(fot_handle *h, const fot_batch *batch, fot_result *out)
{
    if (!h) return FOT_ERR_INVALID;
    if (!batch) return fail(h, FOT_ERR_INVALID, "batch is NULL");
    if (batch->n_inst > 0 && !out) return fail(h, FOT_ERR_INVALID, "out is NULL");
    if (batch->n_inst <= 0) return batch->n_inst == 0 ? FOT_OK : fail(h, FOT_ERR_INVALID, "n_inst < 0");
    if (!h->has_path) return fail(h, FOT_ERR_NO_PATH_SET, "fot_set_path_* has not been called");
    // extents of the caller's obstacle arrays
    BatchLayout probe;
    std::string err;
    int rc = build_batch_layout(h->params, h->P, h->shapes, *batch, probe, err);
    if (rc != FOT_OK) return fail(h, rc, err);
    const size_t elem = batch->obstacle_dtype == FOT_F32 ? sizeof(float) : sizeof(double);
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t st_bytes = (size_t)probe.n_static * 2 * elem, dy_bytes = (size_t)probe.dyn_src_points * 2 * elem;
    const size_t out_bytes = sizeof(fot_result) * (size_t)batch->n_inst;
    // A plan step for one or a few egos is latency, not bandwidth: its obstacle points and its records (written once,
    // by the selecting wave) then travel straight between the kernels and pinned host memory -- two copy operations and
    // their synchronisation less per call.  The dynamic tensors cross the bus ONCE: the NaN-scan blocks of the first
    // launch read them out of the pinned block and leave a copy in HBM for k_cull (NanScan::stage); the few static points
    // are read in place.  Up to the 2 MiB below that beats a staging copy plus its wait (scripts/size_sweep.py), beyond
    // it the tensors are staged in HBM with copy operations.
    if (st_bytes + dy_bytes <= 2 * SMALL_CALL_BYTES && out_bytes <= 8 * SMALL_CALL_BYTES) {   // (records stream out as instances finish)
        const size_t dy_off = align256(st_bytes);
        HIP_TRY(h, h->hSmallIn.ensure(dy_off + dy_bytes + 256));
        HIP_TRY(h, h->hRecOut.ensure(out_bytes));
        char *in = (char *)h->hSmallIn.p;
        if (st_bytes) std::memcpy(in, batch->static_xy, st_bytes);
        if (dy_bytes) std::memcpy(in + dy_off, batch->dyn_xy, dy_bytes);
        // the dynamic tensors cross PCIe once: the scan blocks of the first launch leave a copy in HBM for the others
        static const bool no_stage = std::getenv("FOT_NO_SCAN_STAGE") != nullptr;        // diagnostics scripts
        void *stage = nullptr;
        if (dy_bytes && !no_stage) { HIP_TRY(h, h->dUserDyn.ensure(dy_bytes + 256)); stage = h->dUserDyn.p; }
        arm_records(h, batch->n_inst);
        rc = enqueue_plan(h, *batch, in, in + dy_off, (fot_result *)h->hRecOut.p, h->stream, true, stage);
        if (rc != FOT_OK) { h->done_seq_armed = false; return rc; }
        rc = wait_records(h, batch->n_inst, h->stream);
        if (rc != FOT_OK) return rc;
        // what the device wrote of each record: the header and the first n_total entries of the 15 path arrays (a fifth of
        // the record at 51 samples -- the whole-record copy cost a one-ego call 2 us); the caller's entries past n_total stay
        // as they are
        const size_t head = offsetof(fot_result, t), used = sizeof(double) * (size_t)h->P.n_total;
        for (int i = 0; i < batch->n_inst; ++i) {
            const fot_result *src = (const fot_result *)h->hRecOut.p + i;
            std::memcpy(&out[i], src, head);
            for (int f = 0; f < 15; ++f) std::memcpy(out[i].t + (size_t)f * FOT_MAX_NT, src->t + (size_t)f * FOT_MAX_NT, used);
        }
        return FOT_OK;
    }
    HIP_TRY(h, h->dUserStatic.ensure(std::max<size_t>(st_bytes, 16)));
    HIP_TRY(h, h->dUserDyn.ensure(std::max<size_t>(dy_bytes, 16)));
    HIP_TRY(h, h->dOut.ensure_zeroed(sizeof(fot_result) * (size_t)batch->n_inst));
    if (st_bytes) HIP_TRY(h, hipMemcpyAsync(h->dUserStatic.p, batch->static_xy, st_bytes, hipMemcpyHostToDevice, h->stream));
    if (dy_bytes) HIP_TRY(h, hipMemcpyAsync(h->dUserDyn.p, batch->dyn_xy, dy_bytes, hipMemcpyHostToDevice, h->stream));
    rc = enqueue_plan(h, *batch, h->dUserStatic.p, h->dUserDyn.p, h->dOut.as<fot_result>(), h->stream);
    if (rc != FOT_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(out, h->dOut.p, sizeof(fot_result) * (size_t)batch->n_inst, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FOT_OK;
}

int fot_frenet_state_batch(fot_handle *h, int32_t n, const fot_ego *ego,
                           double *frenet, double *ref, double *new_prev_s, int32_t *ok)
{
    if (!h) return FOT_ERR_INVALID;
    if (!h->has_path) return fail(h, FOT_ERR_NO_PATH_SET, "fot_set_path_* has not been called");
    if (n <= 0) return FOT_OK;
    if (!ego) return fail(h, FOT_ERR_INVALID, "ego is NULL");
    std::vector<InstDesc> desc((size_t)n);
    for (int i = 0; i < n; ++i) { desc[i] = InstDesc(); desc[i].ego = ego[i]; }
    for (int i = n - 1, run = 0; i >= 0; --i) {                 // chain lengths, as build_batch_layout sets them
        desc[i].n_chained = run;
        run = desc[i].ego.has_prev_s == FOT_PREV_S_CHAINED ? run + 1 : 0;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    { int r = order_begin(h, h->stream); if (r != FOT_OK) return r; }
    if (sizeof(InstDesc) * (size_t)n <= SMALL_CALL_BYTES) {      // small call: straight from / into pinned host memory
        HIP_TRY(h, h->hSmallIn.ensure(sizeof(InstDesc) * (size_t)n));
        HIP_TRY(h, h->hSmallOut.ensure(sizeof(InstState) * (size_t)n));
        std::memcpy(h->hSmallIn.p, desc.data(), sizeof(InstDesc) * (size_t)n);
        LAUNCH_TRY(h, launch_frenet_state(h->dP.as<DevParams>(), spline_view(h), (const InstDesc *)h->hSmallIn.p,
                                          (InstState *)h->hSmallOut.p, n, MetaImport(), NanScan(), nullptr, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        const InstState *stp = (const InstState *)h->hSmallOut.p;
        for (int i = 0; i < n; ++i) {
            if (frenet) std::memcpy(frenet + 6 * (size_t)i, stp[i].frenet0, sizeof(double) * 6);
            if (ref) std::memcpy(ref + 6 * (size_t)i, stp[i].ref0, sizeof(double) * 6);
            if (new_prev_s) new_prev_s[i] = stp[i].new_prev_s;
            if (ok) ok[i] = stp[i].c2f_ok;
        }
        return FOT_OK;
    }
    HIP_TRY(h, h->dTmpA.ensure(sizeof(InstDesc) * (size_t)n));
    HIP_TRY(h, h->dTmpB.ensure(sizeof(InstState) * (size_t)n));
    HIP_TRY(h, hipMemcpyAsync(h->dTmpA.p, desc.data(), sizeof(InstDesc) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    LAUNCH_TRY(h, launch_frenet_state(h->dP.as<DevParams>(), spline_view(h), h->dTmpA.as<InstDesc>(),
                                      h->dTmpB.as<InstState>(), n, MetaImport(), NanScan(), nullptr, h->stream));
    std::vector<InstState> st((size_t)n);
    HIP_TRY(h, hipMemcpyAsync(st.data(), h->dTmpB.p, sizeof(InstState) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; ++i) {
        if (frenet) std::memcpy(frenet + 6 * (size_t)i, st[i].frenet0, sizeof(double) * 6);
        if (ref) std::memcpy(ref + 6 * (size_t)i, st[i].ref0, sizeof(double) * 6);
        if (new_prev_s) new_prev_s[i] = st[i].new_prev_s;
        if (ok) ok[i] = st[i].c2f_ok;
    }
    return FOT_OK;
}

int fot_debug_candidates(fot_handle *h, int32_t inst, int32_t cap, double *cost,
                         int32_t *status, int32_t *keep, int32_t *n_t)
{
    if (!h) return FOT_ERR_INVALID;
    if (!h->last_valid) return fail(h, FOT_ERR_INVALID, "no completed plan call on this handle");
    int local = 0;
    Workspace *w = lane_of(h, inst, &local);
    if (!w) return fail(h, FOT_ERR_INVALID, "instance index out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());                          // diagnostic entry: whatever stream the plan ran on
    const InstDesc &D = w->last.desc[local];
    InstState S;
    HIP_TRY(h, hipMemcpy(&S, w->dState.as<InstState>() + local, sizeof(InstState), hipMemcpyDeviceToHost));
    const int n = S.n_cand;
    const int m = n < cap ? n : cap;
    if (m <= 0) return n;
    std::vector<uint8_t> st8((size_t)m);
    std::vector<uint16_t> kp16((size_t)m);
    if (cost) HIP_TRY(h, hipMemcpy(cost, w->dCost.as<double>() + D.cand_off, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(st8.data(), w->dStatus.as<uint8_t>() + D.cand_off, (size_t)m, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(kp16.data(), w->dKeep.as<uint16_t>() + D.cand_off, sizeof(uint16_t) * (size_t)m, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; ++i) {
        if (status) status[i] = st8[i];
        if (keep) keep[i] = kp16[i];
        if (n_t) {
            if (i < D.n_grid) n_t[i] = h->P.ti[i / (D.n_tv * h->P.n_di)].n_t;
            else n_t[i] = h->P.n_total;
        }
    }
    return n;
}

int fot_debug_candidate_path(fot_handle *h, int32_t inst, int32_t index, double *arrays, int32_t *n_t)
{
    if (!h) return FOT_ERR_INVALID;
    if (!h->last_valid) return fail(h, FOT_ERR_INVALID, "no completed plan call on this handle");
    int local = 0;
    Workspace *w = lane_of(h, inst, &local);
    if (!w) return fail(h, FOT_ERR_INVALID, "instance index out of range");
    if (!arrays) return fail(h, FOT_ERR_INVALID, "arrays is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, h->dTmpA.ensure(sizeof(double) * 15 * FOT_MAX_NT));
    HIP_TRY(h, h->dTmpD.ensure(sizeof(int32_t) * 2));
    HIP_TRY(h, hipMemsetAsync(h->dTmpA.p, 0, sizeof(double) * 15 * FOT_MAX_NT, h->stream));
    LAUNCH_TRY(h, launch_debug_path(h->dP.as<DevParams>(), (const InstDesc *)w->dMeta.p, w->dState.as<InstState>(),
                                    spline_view(h), local, index,
                                    h->dTmpA.as<double>(), h->dTmpD.as<int32_t>(), h->stream));
    int32_t meta[2] = { 0, 0 };
    HIP_TRY(h, hipMemcpyAsync(arrays, h->dTmpA.p, sizeof(double) * 15 * FOT_MAX_NT, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(meta, h->dTmpD.p, sizeof(meta), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!meta[1]) return fail(h, FOT_ERR_INVALID, "candidate index out of range");
    if (n_t) *n_t = meta[0];
    return FOT_OK;
}

int32_t fot_wire_n_total(const fot_handle *h) { return h ? h->P.n_total : FOT_ERR_INVALID; }

int32_t fot_wire_record_bytes(int32_t n_total)
{
    if (n_total < 1 || n_total > FOT_MAX_NT) return FOT_ERR_INVALID;
    return (int32_t)align256(sizeof(fot_wire_header) + sizeof(float) * 15 * (size_t)n_total);
}

int fot_pack_records_device(fot_handle *h, int32_t n, const fot_result *records_dev, void *wire_dev, void *stream)
{
    if (!h) return FOT_ERR_INVALID;
    if (n <= 0) return n == 0 ? FOT_OK : fail(h, FOT_ERR_INVALID, "n < 0");
    if (!records_dev || !wire_dev) return fail(h, FOT_ERR_INVALID, "NULL buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    // ordered behind the handle's previous enqueue whatever stream that ran on: the records are usually the output of
    // the plan call just made, and NULL (= the handle's own stream) is also torch's default-stream handle
    { int r = order_begin(h, st); if (r != FOT_OK) return r; }
    LAUNCH_TRY(h, launch_pack_wire(n, h->P.n_total, fot_wire_record_bytes(h->P.n_total), records_dev,
                                   (unsigned char *)wire_dev, st));
    return order_end(h, st);
}

static_assert(offsetof(fot_result, s) - offsetof(fot_result, t) == sizeof(double) * FOT_MAX_NT &&
              offsetof(fot_result, c) - offsetof(fot_result, t) == sizeof(double) * FOT_MAX_NT * 14,
              "the 15 path arrays of fot_result are contiguous");

int fot_pack_records_host(int32_t n_total, int32_t n, const fot_result *records, void *wire)
{
    const int32_t stride = fot_wire_record_bytes(n_total);
    if (stride < 0 || n < 0 || (n > 0 && (!records || !wire))) return FOT_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        const fot_result &R = records[i];
        unsigned char *w = (unsigned char *)wire + (size_t)i * stride;
        std::memset(w, 0, (size_t)stride);
        fot_wire_header H;
        std::memset(&H, 0, sizeof(H));
        H.status = R.status; H.best_index = R.best_index; H.n_cand = R.n_cand; H.n_keep = R.n_keep;
        H.cost = R.cost; H.stats_valid = R.stats_valid; H.n_total = n_total;
        H.new_last_kappa = R.new_last_kappa; H.new_prev_s = R.new_prev_s;
        std::memcpy(H.stats, R.stats, sizeof(H.stats));
        std::memcpy(H.frenet0, R.frenet0, sizeof(H.frenet0));
        std::memcpy(H.ref0, R.ref0, sizeof(H.ref0));
        std::memcpy(w, &H, sizeof(H));
        float *path = (float *)(w + sizeof(H));
        const double *arr = R.t;
        for (int f = 0; f < 15; ++f) {                           // (k_pack_wire: offsets for s, x, y; zeros past n_keep)
            const double base = f == 1 ? R.frenet0[0] : f == 9 ? R.ref0[1] : f == 10 ? R.ref0[2] : 0.0;
            for (int k = 0; k < n_total; ++k)
                path[f * n_total + k] = k < R.n_keep ? (float)(arr[f * FOT_MAX_NT + k] - base) : 0.0f;
        }
    }
    return FOT_OK;
}

int fot_unpack_records(int32_t n_total, int32_t n, const void *wire, fot_result *records)
{
    const int32_t stride = fot_wire_record_bytes(n_total);
    if (stride < 0 || n < 0 || (n > 0 && (!records || !wire))) return FOT_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        const unsigned char *w = (const unsigned char *)wire + (size_t)i * stride;
        fot_wire_header H;
        std::memcpy(&H, w, sizeof(H));
        if (H.n_total != n_total) return FOT_ERR_INVALID;
        if (H.n_keep < 0 || H.n_keep > n_total || H.n_cand < 0 || H.best_index < -1 || H.best_index >= (H.n_cand > 0 ? H.n_cand : 1))
            return FOT_ERR_INVALID;                              // (a corrupt or foreign record)
        fot_result &R = records[i];
        std::memset(&R, 0, sizeof(R));
        R.status = H.status; R.best_index = H.best_index; R.n_cand = H.n_cand; R.n_keep = H.n_keep;
        R.cost = H.cost; R.stats_valid = H.stats_valid;
        R.new_last_kappa = H.new_last_kappa; R.new_prev_s = H.new_prev_s;
        std::memcpy(R.stats, H.stats, sizeof(H.stats));
        std::memcpy(R.frenet0, H.frenet0, sizeof(H.frenet0));
        std::memcpy(R.ref0, H.ref0, sizeof(H.ref0));
        const float *path = (const float *)(w + sizeof(H));
        double *arr = R.t;
        const int keep = H.n_keep < n_total ? H.n_keep : n_total;
        for (int f = 0; f < 15; ++f) {
            const double base = f == 1 ? R.frenet0[0] : f == 9 ? R.ref0[1] : f == 10 ? R.ref0[2] : 0.0;
            for (int k = 0; k < keep; ++k) arr[f * FOT_MAX_NT + k] = base + (double)path[f * n_total + k];
        }
    }
    return FOT_OK;
}

int fot_debug_set_eval_segments(fot_handle *h, int32_t n_seg)
{
    if (!h) return FOT_ERR_INVALID;
    if (n_seg < 0 || n_seg > 4) return fail(h, FOT_ERR_INVALID, "fot_debug_set_eval_segments: 0 (automatic) .. 4");
    h->eval_segments = n_seg;
    return FOT_OK;
}

int fot_debug_set_tile_cut(fot_handle *h, int32_t cut)
{
    if (!h) return FOT_ERR_INVALID;
    if (cut != TILE_CUT_AUTO && cut != TILE_CUT_WAVE && cut != TILE_CUT_GROUP)
        return fail(h, FOT_ERR_INVALID, "fot_debug_set_tile_cut: 0 (automatic), 1 (per-wave rows), 2 (groups)");
    if (cut == h->tile_cut) return FOT_OK;
    return upload_tile_shapes(h, cut);
}

int fot_debug_time_info(const fot_handle *h, double time, int32_t *n_t, double *quartic_inv4, double *quintic_inv9)
{
    if (!h || !(time > 0.0)) return FOT_ERR_INVALID;
    TimeInfo ti;
    if (!time_info(time, h->params.dt, ti)) return FOT_ERR_UNSUPPORTED;      // more than FOT_MAX_NT samples
    if (n_t) *n_t = ti.n_t;
    if (quartic_inv4) std::memcpy(quartic_inv4, ti.qa, sizeof(ti.qa));
    if (quintic_inv9) std::memcpy(quintic_inv9, ti.qi, sizeof(ti.qi));
    return FOT_OK;
}

int fot_debug_margins(fot_handle *h, int32_t inst, int32_t cap, double *margins)
{
    if (!h) return FOT_ERR_INVALID;
    if (!h->last_valid) return fail(h, FOT_ERR_INVALID, "no completed plan call on this handle");
    int local = 0;
    Workspace *w = lane_of(h, inst, &local);
    if (!w) return fail(h, FOT_ERR_INVALID, "instance index out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());                          // diagnostic entry: whatever stream the plan ran on
    InstState S;
    HIP_TRY(h, hipMemcpy(&S, w->dState.as<InstState>() + local, sizeof(InstState), hipMemcpyDeviceToHost));
    const int n = S.n_cand;
    const int m = n < cap ? n : cap;
    if (m <= 0 || !margins) return n;
    const size_t bytes = sizeof(double) * FOT_MARGIN_GROUPS * (size_t)m;
    HIP_TRY(h, h->dTmpB.ensure(bytes));
    EntryArrays ea;
    ea.cnt = w->dEntCnt.as<int32_t>(); ea.e32 = w->dEnt32.as<f2>(); ea.e64 = w->dEnt64.as<d2>();
    ea.sid = w->dEntSid.as<uint8_t>(); ea.rng = w->dWaveRng.as<TileStep>();
    LAUNCH_TRY(h, launch_debug_margins(h->dP.as<DevParams>(), (const InstDesc *)w->dMeta.p, w->dState.as<InstState>(),
                                       spline_view(h), local, ea, m, h->dTmpB.as<double>(), h->stream));
    HIP_TRY(h, hipMemcpyAsync(margins, h->dTmpB.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return n;
}

}  // extern "C"

namespace {

// shared body of fot_check_collision_paths (mode 0) and fot_check_paths (mode 1)
int check_ext(fot_handle *h, int mode, int32_t n_paths, const int32_t *len, const int32_t *flags,
              const double *const arr[9], const fot_overrides *ov, double max_stop,
              int32_t n_static, const double *static_xy, int32_t dmode, int32_t S, int32_t Pn, int32_t T,
              const double *dyn, int32_t *status_out)
{
    if (!h) return FOT_ERR_INVALID;
    if (n_paths <= 0) return FOT_OK;
    if (!len || !status_out) return fail(h, FOT_ERR_INVALID, "NULL path array");
    const DevParams &P = h->P;
    // one single-instance batch carries limits + obstacle set
    fot_ego ego = {};
    double target = 0.0;
    int32_t soff[2] = { 0, n_static > 0 ? n_static : 0 };
    int64_t doff[1] = { 0 };
    int32_t dims[4] = { dmode, S, Pn, T };
    fot_batch b = {};
    b.n_inst = 1; b.obstacle_dtype = FOT_F64; b.ego = &ego; b.target_speed = &target;
    b.overrides = ov; b.max_stop_distance = &max_stop;
    b.static_xy = static_xy; b.static_off = soff; b.dyn_xy = dyn; b.dyn_off = doff; b.dyn_dims = dims;
    BatchLayout L;
    std::string err;
    int rc = build_batch_layout(h->params, P, h->shapes, b, L, err);
    if (rc != FOT_OK) return fail(h, rc, err);
    h->last_valid = false;

    const size_t np = (size_t)n_paths, plane = np * FOT_MAX_NT;
    std::vector<double> flat(9 * plane, 0.0);
    for (int i = 0; i < n_paths; ++i)
        if (len[i] < 0 || len[i] > FOT_MAX_NT) return fail(h, FOT_ERR_INVALID, "path length out of range");
    for (int f = 0; f < 9; ++f)
        if (arr[f]) std::memcpy(flat.data() + f * plane, arr[f], sizeof(double) * plane);
    std::vector<int32_t> meta(2 * np, 3);
    std::memcpy(meta.data(), len, sizeof(int32_t) * np);
    if (flags) std::memcpy(meta.data() + np, flags, sizeof(int32_t) * np);

    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    { int r = order_begin(h, st); if (r != FOT_OK) return r; }
    const size_t st_bytes = sizeof(double) * 2 * (size_t)L.n_static, dy_bytes = sizeof(double) * 2 * (size_t)L.dyn_src_points;
    HIP_TRY(h, h->dUserStatic.ensure(std::max<size_t>(st_bytes, 16)));
    HIP_TRY(h, h->dUserDyn.ensure(std::max<size_t>(dy_bytes, 16)));
    HIP_TRY(h, h->dTmpA.ensure(sizeof(InstDesc)));
    HIP_TRY(h, h->dTmpB.ensure(sizeof(double) * flat.size()));
    HIP_TRY(h, h->dTmpC.ensure(sizeof(int32_t) * meta.size()));
    HIP_TRY(h, h->dTmpD.ensure(sizeof(int32_t) * np));
    if (st_bytes) HIP_TRY(h, hipMemcpyAsync(h->dUserStatic.p, static_xy, st_bytes, hipMemcpyHostToDevice, st));
    // a pedestrian whose track holds a NaN coordinate anywhere is no obstacle at any step (the reference's pre-filter
    // takes np.min / np.max over the whole track, frenet_planner.py:1211-1219): k_check_ext tests step by step, where an
    // all-NaN track says the same
    std::vector<double> dyn_clean;
    if (dy_bytes && L.desc[0].dyn_mode != FOT_DYN_NONE) {
        const InstDesc &D0 = L.desc[0];
        const size_t row = 2 * (size_t)D0.T;
        for (size_t j = 0; j < (size_t)D0.S * D0.P; ++j) {
            bool bad = false;
            for (size_t e = 0; e < row; ++e) bad |= std::isnan(dyn[j * row + e]);
            if (!bad) continue;
            if (dyn_clean.empty()) dyn_clean.assign(dyn, dyn + 2 * (size_t)L.dyn_src_points);
            for (size_t e = 0; e < row; ++e) dyn_clean[j * row + e] = NAN;
        }
    }
    if (dy_bytes) HIP_TRY(h, hipMemcpyAsync(h->dUserDyn.p, dyn_clean.empty() ? dyn : dyn_clean.data(), dy_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->dTmpA.p, L.desc.data(), sizeof(InstDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->dTmpB.p, flat.data(), sizeof(double) * flat.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->dTmpC.p, meta.data(), sizeof(int32_t) * meta.size(), hipMemcpyHostToDevice, st));
    LAUNCH_TRY(h, launch_check_ext(h->dP.as<DevParams>(), h->dTmpA.as<InstDesc>(), n_paths, mode, h->dTmpC.as<int32_t>(),
                                   h->dTmpC.as<int32_t>() + np, h->dTmpB.as<double>(), h->dUserStatic.as<double>(),
                                   h->dUserDyn.as<double>(), h->dTmpD.as<int32_t>(), st));
    HIP_TRY(h, hipMemcpyAsync(status_out, h->dTmpD.p, sizeof(int32_t) * np, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return FOT_OK;
}

}  // namespace

extern "C" {

int fot_check_collision_paths(fot_handle *h, int32_t n_paths, const int32_t *len,
                              const double *x, const double *y, const double *yaw, const double *t,
                              int32_t n_static, const double *static_xy,
                              int32_t mode, int32_t S, int32_t Pn, int32_t T, const double *dyn,
                              int32_t *free_out)
{
    if (!h) return FOT_ERR_INVALID;
    if (n_paths > 0 && (!x || !y || !t)) return fail(h, FOT_ERR_INVALID, "NULL path array");
    if (n_paths > 0 && h->P.has_footprint && !yaw)
        return fail(h, FOT_ERR_INVALID, "yaw is required with a multi-circle footprint");
    const double *arr[9] = { x, y, yaw, nullptr, nullptr, nullptr, nullptr, nullptr, t };
    return check_ext(h, 0, n_paths, len, nullptr, arr, nullptr, NAN, n_static, static_xy, mode, S, Pn, T, dyn, free_out);
}

int fot_check_paths(fot_handle *h, int32_t n_paths, const int32_t *len, const int32_t *flags,
                    const double *x, const double *y, const double *yaw, const double *v, const double *a,
                    const double *c, const double *d, const double *s, const double *t,
                    const fot_overrides *overrides, double max_stop_distance,
                    int32_t n_static, const double *static_xy,
                    int32_t mode, int32_t S, int32_t Pn, int32_t T, const double *dyn, int32_t *status_out)
{
    if (!h) return FOT_ERR_INVALID;
    if (n_paths > 0 && (!x || !y || !v || !a || !c || !t)) return fail(h, FOT_ERR_INVALID, "NULL path array");
    const double *arr[9] = { x, y, yaw, v, a, c, d, s, t };
    return check_ext(h, 1, n_paths, len, flags, arr, overrides, max_stop_distance, n_static, static_xy, mode, S, Pn, T,
                     dyn, status_out);
}

}  // extern "C"
