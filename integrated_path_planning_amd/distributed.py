"""Instance sharding over GPUs: one process per GPU, contiguous shards, one all-gather.

Each ego/scenario instance is planned independently, so the
only exchange is an all-gather of the fixed-size ``fot_result`` records of the
selected paths.  With backend "nccl" that is RCCL over xGMI; the same code runs
on "gloo" for the CPU tests of the sharding/gather logic.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import ctypes as C

import numpy as np

from . import _abi


def wire_record_bytes(n_total_samples: int) -> int:
    """Bytes of one record in the compact wire form (include/fot.h): header + float32 path samples, 256-aligned."""
    rb = _abi.lib().fot_wire_record_bytes(int(n_total_samples))
    if rb < 0:
        raise ValueError(f"n_total_samples {n_total_samples} out of range")
    return rb


def pack_records_host(records, n: int, n_total_samples: int) -> np.ndarray:
    """ctypes array of fot_result (host) -> uint8 wire bytes (pure format conversion, no GPU)."""
    out = np.zeros(n * wire_record_bytes(n_total_samples), dtype=np.uint8)
    _abi.check(None, _abi.lib().fot_pack_records_host(int(n_total_samples), n, C.cast(records, C.c_void_p),
                                                      out.ctypes.data))
    return out


def unpack_records(wire: np.ndarray, n: int, n_total_samples: int):
    """uint8 wire bytes -> ctypes array of ``fot_result`` (the host view; path samples widened from float32)."""
    wire = np.ascontiguousarray(wire[: n * wire_record_bytes(n_total_samples)])
    out = (_abi.Result * max(n, 1))()
    _abi.check(None, _abi.lib().fot_unpack_records(int(n_total_samples), n, wire.ctypes.data, out))
    return out


def shard_bounds(n_total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced shards: the first ``n_total % world`` ranks get one extra instance."""
    if world < 1 or n_total < 0:
        raise ValueError("world >= 1 and n_total >= 0 required")
    base, extra = divmod(n_total, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def max_shard(n_total: int, world: int) -> int:
    return -(-n_total // world) if n_total else 0


def all_gather_records(local, n_total: int, world: int, rank: int, group=None, record_bytes: int = 0):
    """All-gather result records.

    ``local``: uint8 torch tensor with this rank's records (``(hi-lo) * record_bytes`` bytes, device or CPU);
    ``record_bytes`` = 0: full ``fot_result`` records, else the compact wire form (``wire_record_bytes``).
    Returns a uint8 tensor of ``n_total * record_bytes`` bytes in global instance order on the same device.
    Shards are padded to the largest shard so that one equal-count all-gather suffices.
    """
    import torch
    import torch.distributed as dist

    rb = record_bytes or _abi.RESULT_BYTES
    bounds = shard_bounds(n_total, world)
    lo, hi = bounds[rank]
    if local.numel() != (hi - lo) * rb:
        raise ValueError(f"rank {rank}: expected {(hi - lo) * rb} bytes, got {local.numel()}")
    if world == 1:
        return local
    m = max_shard(n_total, world)
    if hi - lo == m:
        send = local
    else:
        send = torch.zeros(m * rb, dtype=torch.uint8, device=local.device)
        send[: (hi - lo) * rb] = local
    gathered = torch.empty(world * m * rb, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(gathered, send, group=group)
    if n_total == world * m:
        return gathered
    parts = [gathered[r * m * rb: r * m * rb + (b[1] - b[0]) * rb] for r, b in enumerate(bounds)]
    return torch.cat(parts)


class PipelinedAllGather:
    """Double-buffered asynchronous all-gather for back-to-back plan steps.

    The gather of step i runs on the collective's own stream (RCCL over xGMI) while step i+1 already plans into the
    other buffer pair; a buffer pair is handed out again only after the gather that used it has completed.  Every rank
    contributes the same number of bytes (equal shards, as in the weak-scaling benchmark)."""

    def __init__(self, n_bytes_local: int, world: int, device, depth: int = 2, group=None):
        import torch
        self.world, self.depth, self.group = world, depth, group
        self.send = [torch.zeros(n_bytes_local, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.recv = [torch.zeros(world * n_bytes_local, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.step = 0

    def slot(self):
        """(index, send buffer, receive buffer) for the next step."""
        j = self.step % self.depth
        if self.work[j] is not None:
            self.work[j].wait()                  # device-side wait on the current stream for "nccl", host wait for "gloo"
            self.work[j] = None
        return j, self.send[j], self.recv[j]

    def launch(self, j: int) -> None:
        """Start gathering send[j] (written on the current stream) into recv[j]."""
        import torch.distributed as dist
        self.work[j] = dist.all_gather_into_tensor(self.recv[j], self.send[j], group=self.group, async_op=True)
        self.step += 1

    def drain(self):
        """Wait for every gather in flight; returns the receive buffer of the most recent step (None before any)."""
        for j in range(self.depth):
            if self.work[j] is not None:
                self.work[j].wait()
                self.work[j] = None
        return self.recv[(self.step - 1) % self.depth] if self.step else None


def records_from_bytes(buf: np.ndarray, n: int):
    """uint8 array -> ctypes array of ``fot_result``."""
    return (_abi.Result * n).from_buffer_copy(np.ascontiguousarray(buf[: n * _abi.RESULT_BYTES]).tobytes())


class ShardedPlanner:
    """Plans this rank's shard on its GPU and all-gathers the selected paths.

    ``torch`` provides device memory, the stream and the collective; the planning itself is libfot.
    """

    def __init__(self, waypoints, device_index: int, world: int, rank: int, group=None, **planner_kwargs):
        import torch
        from .planner import BatchPlanner

        self.torch = torch
        self.world, self.rank, self.group = world, rank, group
        self.device = torch.device("cuda", device_index)
        self.planner = BatchPlanner(waypoints=waypoints, device=device_index, **planner_kwargs)
        self._keep = None
        self._stream = None

    def plan(self, requests: Sequence, obstacle_dtype=np.float32):
        """``requests``: the GLOBAL list of PlanRequest (every rank passes the same list).
        Returns (records of all instances in global order, uint8 tensor they live in)."""
        from .batch import PackedBatch

        torch = self.torch
        n_total = len(requests)
        lo, hi = shard_bounds(n_total, self.world)[self.rank]
        pb = PackedBatch(requests[lo:hi], obstacle_dtype)
        dyn = torch.from_numpy(pb.dyn_xy).to(self.device) if pb.dyn_xy.size else None
        st = torch.from_numpy(pb.static_xy).to(self.device) if pb.static_xy.size else None
        out = torch.zeros(max(hi - lo, 1) * _abi.RESULT_BYTES, dtype=torch.uint8, device=self.device)
        # an explicit stream: the NULL (default) stream would mean "the handle's own stream" to libfot, which the
        # collective below does not wait for
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        self._stream.wait_stream(torch.cuda.current_stream(self.device))     # the uploads above
        with torch.cuda.stream(self._stream):
            if hi > lo:
                self.planner.plan_packed_device(
                    pb.with_device_obstacles(st.data_ptr() if st is not None else None,
                                             dyn.data_ptr() if dyn is not None else None),
                    out.data_ptr(), self._stream.cuda_stream)
            self._keep = (pb, dyn, st)                   # inputs must outlive the enqueued work
            # the records travel in the compact wire form (3 328 instead of 7 856 bytes at 51 samples)
            nt = self.planner.n_total_samples
            wb = wire_record_bytes(nt)
            wire = torch.zeros(max(hi - lo, 1) * wb, dtype=torch.uint8, device=self.device)
            if hi > lo:
                self.planner.pack_records_device(hi - lo, out.data_ptr(), wire.data_ptr(), self._stream.cuda_stream)
            full = all_gather_records(wire[: (hi - lo) * wb], n_total, self.world, self.rank, self.group, record_bytes=wb)
            self._stream.synchronize()
        host = full.cpu().numpy()
        return unpack_records(host, n_total, nt), full
