"""Sharding + all-gather of selected-path records with world_size 2 on CPU (gloo).

The planner itself needs a GPU; what is covered here is the N>1 logic the 8-GPU
run depends on: contiguous shards, padding of uneven shards, global ordering
after the all-gather.  Each rank fabricates recognisable records for its shard.
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.distributed import (all_gather_records, max_shard, pack_records_host, records_from_bytes,
                                                      shard_bounds, unpack_records, wire_record_bytes)


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 255, 256, 4096, 4097):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == max_shard(n, w)
    assert shard_bounds(4096, 8)[3] == (1536, 2048)          # BASELINE config 5: 512 per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_total, world)[rank]
        recs = (_abi.Result * max(hi - lo, 1))()
        for i in range(hi - lo):
            g = lo + i
            recs[i].status = g % 3
            recs[i].best_index = g
            recs[i].cost = 0.5 * g
            recs[i].n_keep = 2
            recs[i].x[0] = float(g)
            recs[i].x[1] = float(rank)
        local = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8)[: (hi - lo) * _abi.RESULT_BYTES].clone()
        full = all_gather_records(local, n_total, world, rank)
        out = records_from_bytes(full.numpy(), n_total)
        bounds = shard_bounds(n_total, world)
        ok = True
        for g in range(n_total):
            owner = [r for r, (a, b) in enumerate(bounds) if a <= g < b][0]
            r = out[g]
            ok &= (r.best_index == g and r.status == g % 3 and r.cost == 0.5 * g and r.x[0] == float(g)
                   and r.x[1] == float(owner))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_all_gather_world2(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))


def _pipe_worker(rank, world, port, ret, depth=2):
    from integrated_path_planning_amd.distributed import PipelinedAllGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nb = 3 * _abi.RESULT_BYTES
        pg = PipelinedAllGather(nb, world, torch.device("cpu"), depth=depth)
        ok = True
        seen = {}
        for step in range(7):                                   # more steps than buffers: every pair is reused
            j, send, recv = pg.slot()
            if j in seen:                                       # the pair is free again: its previous gather is complete
                s0 = seen[j]
                for r in range(world):
                    ok &= bool((recv[r * nb:(r + 1) * nb] == (s0 * 16 + r) % 251).all())
            send.fill_((step * 16 + rank) % 251)
            pg.launch(j)
            seen[j] = step
        last = pg.drain()
        for r in range(world):
            ok &= bool((last[r * nb:(r + 1) * nb] == (6 * 16 + r) % 251).all())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_pipelined_all_gather_world2(depth):
    """Multi-buffered asynchronous gather (what bench.py runs at N > 1, one buffer pair per plan call in flight: three
    by default): buffers rotate, nothing is overwritten early."""
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, ret, depth)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))


def _fabricate(lo, hi, rank, n_total=51):
    recs = (_abi.Result * max(hi - lo, 1))()
    for i in range(hi - lo):
        g = lo + i
        r = recs[i]
        r.status, r.best_index, r.n_cand, r.n_keep = g % 2, g, 2240, 2 + g % (n_total - 1)
        r.cost, r.stats_valid, r.new_last_kappa, r.new_prev_s = 0.25 * g + 1e-9, 1, 1e-3 * g, 10.0 + g
        for k in range(8):
            r.stats[k] = g + k
        for k in range(6):
            r.frenet0[k], r.ref0[k] = g + 0.125 * k, g - 0.125 * k           # (offsets of s, x, y stay representable)
        for f_i, f in enumerate(_abi.PATH_FIELDS):
            arr = getattr(r, f)
            for k in range(r.n_keep):
                arr[k] = (f_i + 1) * 1.5 + k * 0.125 + g + rank * 0.0          # exactly representable in float32
    return recs


def test_wire_record_round_trip():
    """fot_pack_records_host / fot_unpack_records (pure format conversion, no GPU): header fields bit for bit, path
    samples through float32; 3 328 bytes per record at 51 samples instead of the 7 856 of fot_result."""
    assert wire_record_bytes(51) == 3328 and wire_record_bytes(64) == 4096
    assert C.sizeof(_abi.WireHeader) == 176
    recs = _fabricate(0, 5, 0)
    wire = pack_records_host(recs, 5, 51)
    assert wire.nbytes == 5 * 3328
    back = unpack_records(wire, 5, 51)
    for i in range(5):
        a, b = recs[i], back[i]
        for f in ("status", "best_index", "n_cand", "n_keep", "cost", "stats_valid", "new_last_kappa", "new_prev_s"):
            assert getattr(a, f) == getattr(b, f), f
        assert list(a.stats) == list(b.stats) and list(a.frenet0) == list(b.frenet0) and list(a.ref0) == list(b.ref0)
        for f in _abi.PATH_FIELDS:
            assert list(getattr(a, f)[: a.n_keep]) == list(getattr(b, f)[: a.n_keep]), f
            assert all(v == 0.0 for v in getattr(b, f)[a.n_keep:])
    # float32 rounding of a value that is not representable: within 2^-24 of its OFFSET from the record's reference
    # point -- so a path far from the origin keeps its resolution (x, y, s travel as offsets)
    recs[0].x[1] = 97.123456789
    back = unpack_records(pack_records_host(recs, 5, 51), 5, 51)
    assert abs(back[0].x[1] - 97.123456789) <= (97.123456789 - recs[0].ref0[1]) * 2.0 ** -24
    far = _fabricate(0, 1, 0)
    far[0].ref0[1], far[0].ref0[2], far[0].frenet0[0] = 1.0e4, -2.0e4, 5.0e3
    for k in range(far[0].n_keep):
        far[0].x[k], far[0].y[k], far[0].s[k] = 1.0e4 + 0.7 * k + 1e-7, -2.0e4 - 0.3 * k + 1e-7, 5.0e3 + 0.9 * k + 1e-7
    back = unpack_records(pack_records_host(far, 1, 51), 1, 51)
    for f in ("x", "y", "s"):
        err = max(abs(getattr(back[0], f)[k] - getattr(far[0], f)[k]) for k in range(far[0].n_keep))
        assert err <= 1e-6, (f, err)                  # (plain float32 of 2e4 would be off by 1e-3)


def _wire_worker(rank, world, port, n_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_total, world)[rank]
        wb = wire_record_bytes(51)
        local = torch.from_numpy(pack_records_host(_fabricate(lo, hi, rank), hi - lo, 51)[: (hi - lo) * wb].copy())
        full = all_gather_records(local, n_total, world, rank, record_bytes=wb)
        out = unpack_records(full.numpy(), n_total, 51)
        want = _fabricate(0, n_total, 0)
        ok = all(out[g].best_index == g and out[g].cost == want[g].cost and out[g].n_keep == want[g].n_keep
                 and list(out[g].y[: want[g].n_keep]) == list(want[g].y[: want[g].n_keep]) for g in range(n_total))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5])
def test_all_gather_of_wire_records_world2(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_wire_worker, args=(r, world, port, n_total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))
