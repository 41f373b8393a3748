"""Diagnostic: wave timeline of one k_evaluate launch on the bench workload (needs a -DFOT_TIMELINE build of libfot,
see scripts/timeline.sh).  Prints slot utilisation, the concurrency curve and what list scheduling of the measured
wave durations would give at workgroup and at wave granularity."""
import ctypes as C
import heapq
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from integrated_path_planning_amd.batch import request_from_instance                                   # noqa: E402
from integrated_path_planning_amd import _abi, synthetic as syn            # noqa: E402
from integrated_path_planning_amd.batch import PackedBatch                 # noqa: E402
from integrated_path_planning_amd.planner import BatchPlanner              # noqa: E402

N_INST, WPB, SLOTS = int(os.environ.get("FOT_TIMELINE_INST", "256")), int(os.environ.get("FOT_TIMELINE_WPB", "4")), int(os.environ.get("FOT_TIMELINE_SLOTS", "4096"))                       # 256 CUs x 4 SIMDs x 3 waves (VGPR-limited)


def list_schedule(durations, slots):
    ends = [0.0] * slots
    heapq.heapify(ends)
    last = 0.0
    for d in durations:
        e = heapq.heappop(ends) + d
        last = max(last, e)
        heapq.heappush(ends, e)
    return last


def main():
    dev = torch.device("cuda:0")
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(N_INST)]
    pb = PackedBatch(reqs, obstacle_dtype=np.float32)
    bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **syn.CONFIG3_PLANNER)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    out = torch.zeros(N_INST * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    bs = pb.with_device_obstacles(None, dyn.data_ptr())
    st = torch.cuda.current_stream(dev)
    for _ in range(int(os.environ.get("FOT_TIMELINE_LAUNCHES", "4000"))):    # > 1 s of back-to-back launches: the clock settles
        bp.plan_packed_device(bs, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    L = _abi.lib()
    if not hasattr(L, "fot_timeline_read"):
        raise SystemExit("libfot was built without -DFOT_TIMELINE (scripts/timeline.sh)")
    raw = np.zeros(4 * 16384, dtype=np.uint64)
    L.fot_timeline_read.argtypes = [C.c_void_p, C.c_int]
    assert L.fot_timeline_read(raw.ctypes.data, raw.size) == 0
    t = raw.reshape(-1, 4).astype(np.int64)
    if hasattr(L, "fot_timeline_read_rows"):
        rw = np.zeros(16384, dtype=np.uint64)
        L.fot_timeline_read_rows.argtypes = [C.c_void_p, C.c_int]
        if L.fot_timeline_read_rows(rw.ctypes.data, rw.size) == 0:
            rw = rw.astype(np.int64)[: t.shape[0]]
            okr = t[:, 2] > t[:, 1]
            print(f"shader cycles lane 0 spent waiting for its LDS rows, per tile: median {np.median(rw[okr]):.0f} "
                  f"p95 {np.percentile(rw[okr], 95):.0f} (tile loop: median "
                  f"{np.median((t[okr, 2] - t[okr, 1]) * 10 * 2.36):.0f} cycles)")
    if hasattr(L, "fot_timeline_read_clock"):                              # in-kernel shader clock of the wave loops
        ck = np.zeros(2 * 16384, dtype=np.uint64)
        L.fot_timeline_read_clock.argtypes = [C.c_void_p, C.c_int]
        if L.fot_timeline_read_clock(ck.ctypes.data, ck.size) == 0:
            ck = ck.reshape(-1, 2).astype(np.int64)
            okc = (t[:, 2] > t[:, 1]) & (ck[:, 1] > ck[:, 0])
            ghz = (ck[okc, 1] - ck[okc, 0]) / ((t[okc, 2] - t[okc, 1]) * 10.0)          # cycles per ns (100 MHz ticks)
            print(f"in-kernel shader clock over the wave loops: median {np.median(ghz):.3f} GHz "
                  f"(p5 {np.percentile(ghz, 5):.3f}, p95 {np.percentile(ghz, 95):.3f})")
    if hasattr(L, "fot_timeline_read_phase"):                              # where a tile's time goes before its loop
        ph = np.zeros(4 * 16384, dtype=np.uint64)
        L.fot_timeline_read_phase.argtypes = [C.c_void_p, C.c_int]
        if L.fot_timeline_read_phase(ph.ctypes.data, ph.size) == 0:
            ph = ph.reshape(-1, 4).astype(np.int64)
            okp = (t[:, 2] > t[:, 1]) & (ph[:, 0] > 0)
            us = lambda a: "median %.2f p95 %.2f" % (np.median(a) / 100.0, np.percentile(a, 95) / 100.0)
            print("per tile [us]: entry -> spline staged " + us((ph[:, 1] - ph[:, 0])[okp]) + "; -> tile start " + us((t[:, 0] - ph[:, 1])[okp])
                  + "; -> summaries " + us((ph[:, 2] - t[:, 0])[okp]) + "; -> rows built " + us((t[:, 1] - ph[:, 2])[okp])
                  + "; loop " + us((ph[:, 3] - t[:, 1])[okp]) + "; epilogue " + us((t[:, 2] - ph[:, 3])[okp]))
    if N_INST < 16:                                                         # a few egos: the phases above are the answer
        okw = t[:, 2] > 0
        print(f"waves {int(okw.sum())}; first entry -> last end {(t[okw, 2].max() - t[okw, 0].min()) / 100.0:.1f} us")
        return
    n_waves = int(np.nonzero(t[:, 2])[0].max()) + 1
    n_waves = (n_waves + WPB - 1) // WPB * WPB
    t = t[:n_waves]
    ok = t[:, 2] > 0
    t0 = t[ok, 0].min()
    tb, tw, te = [(t[:, i] - t0) / 100.0 for i in range(3)]                 # us
    dur = np.where(ok, te - tw, 0.0)
    prologue = float(np.median((tw - tb)[ok]))
    span = float(te[ok].max())
    print(f"waves {int(ok.sum())} (+{int((~ok).sum())} all-padding), kernel span {span:.1f} us, prologue {prologue:.1f} us")
    print(f"wave duration: median {np.median(dur[ok]):.1f} p95 {np.percentile(dur[ok], 95):.1f} max {dur[ok].max():.1f} us")
    work = float((te - tb)[ok].sum()) / SLOTS
    print(f"sum of wave residency / {SLOTS} slots = {work:.1f} us -> slot utilisation {work / span:.2f}")
    d4 = dur.reshape(-1, WPB)
    full = ok.reshape(-1, WPB).all(1)
    print(f"inside a workgroup: slowest wave {d4[full].max(1).mean():.1f} us, mean wave {d4[full].mean(1).mean():.1f} us")
    ev = np.concatenate([np.stack([tb[ok], np.ones(ok.sum())], 1), np.stack([te[ok], -np.ones(ok.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    marks = np.arange(0.0, span, 20.0)
    print("resident waves at t [us]: " + "  ".join(f"{int(m)}:{int(conc[min(np.searchsorted(ev[:, 0], m), len(conc) - 1)])}" for m in marks))
    wg = d4.max(1) + prologue
    item = (dur + prologue)[ok]
    print(f"list scheduling, workgroups in dispatch order: {list_schedule(wg, SLOTS // WPB):.1f} us; longest first: "
          f"{list_schedule(np.sort(wg)[::-1], SLOTS // WPB):.1f} us")
    print(f"list scheduling, single waves in dispatch order: {list_schedule(item, SLOTS):.1f} us; longest first: "
          f"{list_schedule(np.sort(item)[::-1], SLOTS):.1f} us")
    tag = (t[:, 3] >> 32)[ok]
    t[:, 3] &= 0xffffffff
    xcc, qx, base, rnd = tag & 15, (tag >> 4) & 15, (tag >> 8) & 15, (tag >> 12) & 15
    print("tiles by XCD id:", np.bincount(xcc, minlength=8).tolist(), " label offset seen:", np.unique(base).tolist())
    print("tiles by steal round (0 = own queue):", np.bincount(rnd, minlength=8).tolist())
    for r in range(int(rnd.max()) + 1):
        m = rnd == r
        print(f"  round {r}: {int(m.sum())} tiles, median duration {np.median(dur[ok][m]):.1f} us")
    ch = t[:, 3][ok].astype(float)
    A = np.stack([np.ones_like(ch), ch], 1)
    coef = np.linalg.lstsq(A, dur[ok], rcond=None)[0]
    r2 = 1 - ((dur[ok] - A @ coef) ** 2).sum() / ((dur[ok] - dur[ok].mean()) ** 2).sum()
    print(f"duration ~ {coef[0]:.1f} + {coef[1]:.3f} x chunks in the wave's strip ranges (R^2 {r2:.2f})")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "timeline.npy"), t)


if __name__ == "__main__":
    main()
