"""Row-strip sharding of one frame over the GPUs of a node, with a gather to the root.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  The frame's rows are cut into strips of `rows_per_strip` rows dealt
round-robin to the ranks (contiguous bands would leave the ranks holding the set's interior with
several times the work of the others: the C4 view is 74.7 % interior).
Pixels are independent, so there is NO collective on the compute path; the only exchange is the
final gather of disjoint byte ranges:

    rank r renders its strips, packed, into shard_buf          (HIP kernel, compute stream)
    dist.gather(shard_buf -> root)                             (RCCL send/recv, comm stream)
    root: de-interleave the gathered strips into the frame     (one strided copy, comm stream)

For a frame sequence (.franim sweep) the buffers are double-buffered so that gather(frame n)
overlaps render(frame n+1).  The reference has no multi-GPU path; this is new design
(BASELINE.json north_star: "disjoint row bands with a final RCCL gather over xGMI").
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from .renderer import Shard


def pick_rows_per_strip(height: int, world: int, target: int = 32) -> int:
    """Largest strip height <= target that divides the frame evenly over the ranks
    (height % (world * R) == 0), so every rank owns the same number of full strips."""
    for r in range(min(target, max(1, height // world)), 0, -1):
        if height % (world * r) == 0:
            return r
    return 1


class StripGather:
    """Gathers per-rank packed strips into the root's row-major frame.

    render_fn(shard, out_tensor, frame_index) must fill out_tensor (rows_local x W x C) for
    `shard`; on the GPU path it must only ENQUEUE work on the current torch stream.
    """

    def __init__(self, width: int, height: int, channels: int = 4, dtype=torch.float32,
                 device: Optional[torch.device] = None, rows_per_strip: int = 0,
                 group=None, root: int = 0, buffers: int = 2):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.root = root
        self.W, self.H, self.C = width, height, channels
        self.device = device if device is not None else torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"
        self.R = rows_per_strip or pick_rows_per_strip(height, self.world)
        self.shard = Shard(self.rank, self.world, self.R)
        self.rows_local = self.shard.rows(height)
        self.even = (height % (self.world * self.R) == 0)
        self.nbuf = buffers
        mk = lambda *shape: torch.empty(shape, dtype=dtype, device=self.device)  # noqa: E731
        self.shard_buf = [mk(self.rows_local, width, channels) for _ in range(buffers)]
        self.frames: List[torch.Tensor] = []
        self.recv: List[List[torch.Tensor]] = []
        if self.rank == root:
            self.frames = [mk(height, width, channels) for _ in range(buffers)]
            for _ in range(buffers):
                # one receive tensor per rank, sized for THAT rank's rows (ragged when H is not a
                # multiple of world*R); the root's own entry aliases its shard buffer's shape
                self.recv.append([mk(Shard(r, self.world, self.R).rows(height), width, channels)
                                  for r in range(self.world)])
        self._rows_of = [Shard(r, self.world, self.R).global_rows(height) for r in range(self.world)] \
            if self.rank == root else []
        if self.on_gpu:
            self.compute = torch.cuda.Stream(device=self.device)
            self.comm = torch.cuda.Stream(device=self.device)
            self.rendered = [torch.cuda.Event() for _ in range(buffers)]
            self.gathered = [torch.cuda.Event() for _ in range(buffers)]
            self._used = [False] * buffers

    # -- root-side reassembly ---------------------------------------------------------------
    def _assemble(self, b: int) -> None:
        frame, parts = self.frames[b], self.recv[b]
        if self.even:
            # frame rows = [strip s of rank 0, strip s of rank 1, ...] for s = 0, 1, ...:
            # view the frame as (S, world, R, W, C) and copy rank p's (S, R, W, C) block into [:, p]
            S = self.H // (self.world * self.R)
            fv = frame.view(S, self.world, self.R, self.W, self.C)
            for p, part in enumerate(parts):
                fv[:, p].copy_(part.view(S, self.R, self.W, self.C))
        else:
            for p, part in enumerate(parts):
                if part.shape[0]:
                    frame.index_copy_(0, torch.from_numpy(self._rows_of[p]).to(frame.device), part)

    def _gather(self, b: int) -> None:
        if self.world == 1:
            self.recv[b][0].copy_(self.shard_buf[b])
        else:
            # a gather IS grouped send/recv in RCCL (there is no ncclGather primitive); sizes may
            # differ per rank (ragged last strip), so each rank sends exactly its own bytes and the
            # root posts one receive per peer, all batched into one group (7 concurrent xGMI
            # point-to-point transfers into the root at world 8)
            if self.rank == self.root:
                self.recv[b][self.root].copy_(self.shard_buf[b])
                ops = [dist.P2POp(dist.irecv, self.recv[b][r], r, self.group)
                       for r in range(self.world) if r != self.root and self.recv[b][r].numel()]
            else:
                ops = [dist.P2POp(dist.isend, self.shard_buf[b], self.root, self.group)] \
                    if self.shard_buf[b].numel() else []
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()          # NCCL: makes the current stream wait; gloo: blocks the host
        if self.rank == self.root:
            self._assemble(b)

    # -- one frame, synchronous (CPU/gloo tests, single frames) ---------------------------------
    def render_frame(self, render_fn: Callable, frame_index: int = 0) -> Optional[torch.Tensor]:
        b = 0
        if self.on_gpu:
            with torch.cuda.stream(self.compute):
                render_fn(self.shard, self.shard_buf[b], frame_index)
                self.rendered[b].record(self.compute)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.rendered[b])
                self._gather(b)
            self.comm.synchronize()
        else:
            render_fn(self.shard, self.shard_buf[b], frame_index)
            self._gather(b)
        return self.frames[b] if self.rank == self.root else None

    # -- a frame sequence, pipelined: gather(n) overlaps render(n+1) -----------------------------
    def submit(self, render_fn: Callable, frame_index: int) -> int:
        """Enqueue render + gather of one frame; returns the buffer slot holding it.
        The slot's frame (root) is valid after wait(slot) / drain()."""
        b = frame_index % self.nbuf
        if not self.on_gpu:
            render_fn(self.shard, self.shard_buf[b], frame_index)
            self._gather(b)
            return b
        if self._used[b]:
            self.compute.wait_event(self.gathered[b])      # slot b's previous gather must be done
        with torch.cuda.stream(self.compute):
            render_fn(self.shard, self.shard_buf[b], frame_index)
            self.rendered[b].record(self.compute)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.rendered[b])
            self._gather(b)
            self.gathered[b].record(self.comm)
        self._used[b] = True
        return b

    def drain(self) -> None:
        if self.on_gpu:
            self.compute.synchronize()
            self.comm.synchronize()
