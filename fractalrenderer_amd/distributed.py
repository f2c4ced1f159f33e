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

StripGather (above) brings ONE frame to one root.  For a frame sequence (.franim sweep, bench.py)
FrameExchange rotates the root from frame to frame and ships the smooth-count plane instead of the
colour, which takes the exchange off the critical path (see its docstring for the link budget), and
export_animation builds the reference's offline animation render on top of it.  Its default layout for
sequences is the north-star's disjoint row BANDS, the band of a rank rotating over the frames of a group
(balanced across the group, received in place); interleaved strips remain as the other layout.  The reference has no
multi-GPU path; this is new design (BASELINE.json north_star: "disjoint row bands with a final RCCL
gather over xGMI").
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
        # gloo cannot move device tensors point-to-point: rehearsals of the GPU path on one card bounce the shares
        # through pinned host memory (RCCL moves device memory directly)
        self.stage = self.on_gpu and dist.is_initialized() and dist.get_backend(group) == "gloo"
        if self.stage:
            pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)  # noqa: E731
            self._hsend = pin(self.shard_buf[0])
            self._hrecv = [pin(t) for t in self.recv[0]] if self.rank == root else []
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
            src = self.shard_buf[b]
            dst = self.recv[b] if self.rank == self.root else []
            if self.stage:
                self._hsend.copy_(src)
                torch.cuda.current_stream().synchronize()
                src, dst = self._hsend, self._hrecv
            if self.rank == self.root:
                self.recv[b][self.root].copy_(self.shard_buf[b])
                ops = [dist.P2POp(dist.irecv, dst[r], r, self.group)
                       for r in range(self.world) if r != self.root and dst[r].numel()]
            else:
                ops = [dist.P2POp(dist.isend, src, self.root, self.group)] if src.numel() else []
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()          # NCCL: makes the current stream wait; gloo: blocks the host
            if self.stage and self.rank == self.root:
                for r in range(self.world):
                    if r != self.root:
                        self.recv[b][r].copy_(dst[r], non_blocking=True)
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
            for lane in getattr(self, "lanes", [self.compute]):
                lane.synchronize()
            self.comm.synchronize()


class FrameExchange:
    """Row-strip sharded frames of a SEQUENCE, gathered with rotating roots (an all-to-all).

    Why not StripGather for a sequence: a finished frame is 16 B/pixel of colour, one MI355X renders
    the C2 frame at ~18.5 Gpixel/s = ~300 GB/s of output, and an xGMI link moves ~77 GB/s per
    direction.  Gathering every frame to ONE root funnels (N-1)/N of all output through the root's
    N-1 links: at N = 2 the single link is 4x too slow, at N = 8 the root's 7 links still carry only
    ~1/4 of what 8 GPUs produce.  Two changes remove that wall:

      * rotating roots: frames are processed in groups of N; frame g*N + j is gathered to rank j.
        Every rank renders its strips of all N frames of the group and ONE all-to-all (RCCL grouped
        send/recv over the full xGMI mesh) delivers strip set j to rank j, so all N*(N-1) directed
        links carry traffic instead of the 7 into one root, and each rank ends the group owning one
        complete frame -- which is also where the per-frame export / PNG work then runs, in parallel;
      * compact payload: when the colour is a function of the smooth count alone
        (fr_colorize_supported) the ranks render and ship only the nu plane (8 B/pixel fp64, 4 B fp32)
        and the destination recolours the assembled frame (fr_colorize_async), bit-identically to a
        direct render.  Otherwise the 16-byte colour plane is shipped.

    render_fn(shard, out, frame_index, plane, lane) fills `out` (rows_local x W [x 4]) with plane "nu"
    or "rgba" of `shard` of frame `frame_index`; colorize_fn(nu_frame, rgba_frame, frame_index)
    recolours.  On the GPU path both must only ENQUEUE on the current torch stream.  Double-buffered:
    the exchange of group g overlaps the rendering of group g+1.

    render_lanes: a rank's N renders per group are each 1/N of a frame -- small launches whose ramp-up
    and tail are not hidden (measured on C2, 1/8 shards: 0.25 ms each against 0.11 ms of arithmetic).
    With render_lanes = L > 1 they are spread over L streams, `lane` telling render_fn which of its L
    render contexts to use (one fr_ctx is not re-entrant, distinct contexts run concurrently), so one
    render's tail overlaps the next one's body (measured: 1.99 -> 1.30 ms per group of 8 with L = 4).
    """

    def __init__(self, width: int, height: int, *, payload: str = "nu", nu_dtype=torch.float64,
                 device: Optional[torch.device] = None, rows_per_strip: int = 0, group=None, slots: int = 2,
                 stage_through_host: Optional[bool] = None, render_lanes: int = 1, layout: str = "strips"):
        if payload not in ("nu", "rgba"):
            raise ValueError("payload must be 'nu' or 'rgba'")
        if layout not in ("strips", "bands"):
            raise ValueError("layout must be 'strips' or 'bands'")
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.W, self.H = width, height
        self.payload = payload
        self.device = device if device is not None else torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"
        # layout "bands": ONE contiguous band of H/N rows per rank and frame, and the band a rank renders ROTATES over the
        # frames of a group -- rank r renders band (r + j) mod N of frame j -- so that over a group every rank renders
        # every band once: the load balances across the (similar) frames of the group instead of inside each frame,
        # and a received band is a contiguous row range of the destination's frame: it is received IN PLACE, no
        # de-interleave pass (0.06 ms per group of 8 C2 frames, 134 MB read + written).  Needs H % N == 0; else strips.
        self.bands = layout == "bands" and height % self.world == 0 and self.world > 1
        self.R = (height // self.world) if self.bands else (rows_per_strip or pick_rows_per_strip(height, self.world))
        self.shard = Shard(self.rank, self.world, self.R)
        self.rows_local = self.shard.rows(height)
        self.rows_of = [Shard(r, self.world, self.R).rows(height) for r in range(self.world)]
        self.even = (height % (self.world * self.R) == 0)
        self.nslots = slots
        # gloo cannot move device tensors point-to-point: bounce through pinned host memory (rehearsals
        # of the GPU path on one card; RCCL moves device memory directly)
        if stage_through_host is None:
            stage_through_host = self.on_gpu and dist.is_initialized() and dist.get_backend(group) == "gloo"
        self.stage = stage_through_host
        tail = (4,) if payload == "rgba" else ()
        pdt = torch.float32 if payload == "rgba" else nu_dtype
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=self.device)  # noqa: E731
        N = self.world
        self.send = [mk((N, self.rows_local, width) + tail, pdt) for _ in range(slots)]
        if self.bands:
            self.recv = [None] * slots            # received in place: rows of frame_nu / frame_rgba, set below
        elif self.even:
            # equal shares: one receive tensor per slot, so that the de-interleave is ONE strided copy
            self.recv_all = [mk((N, self.rows_local, width) + tail, pdt) for _ in range(slots)]
            self.recv = [[ra[r] for r in range(N)] for ra in self.recv_all]
        else:
            self.recv = [[mk((self.rows_of[r], width) + tail, pdt) for r in range(N)] for _ in range(slots)]
        self.frame_nu = [mk((height, width), nu_dtype) for _ in range(slots)] if payload == "nu" else [None] * slots
        self.frame_rgba = [mk((height, width, 4), torch.float32) for _ in range(slots)]
        self.frame_index = [-1] * slots          # which frame this rank's slot holds (-1: none)
        self._index = [torch.from_numpy(Shard(r, N, self.R).global_rows(height)).to(self.device)
                       for r in range(N)] if not self.even else []
        if self.bands:
            frames = self.frame_nu if payload == "nu" else self.frame_rgba
            self.recv = [[f[k * self.R:(k + 1) * self.R] for k in range(N)] for f in frames]    # slot b, band k
        if self.stage:
            pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)  # noqa: E731
            self._hsend = pin(self.send[0])
            self._hrecv = [pin(t) for t in self.recv[0]]
        self.nlanes = max(1, int(render_lanes))
        if self.on_gpu:
            self.lanes = [torch.cuda.Stream(device=self.device) for _ in range(self.nlanes)]
            self.compute = self.lanes[0]
            self.comm = torch.cuda.Stream(device=self.device)
            self.rendered = [[torch.cuda.Event() for _ in range(self.nlanes)] for _ in range(slots)]
            self.delivered = [torch.cuda.Event() for _ in range(slots)]
            self._used = [False] * slots
        self._groups = 0

    # -- the exchange of one group: strip set j -> rank j, j < count ------------------------------
    def _exchange(self, b: int, count: int) -> None:
        N, me = self.world, self.rank
        send, recv = self.send[b], self.recv[b]
        # where the share of rank r lands at destination `me`: its own slot (strips) / the band r rendered of frame `me`
        at = (lambda r: (r + me) % N) if self.bands else (lambda r: r)
        if me < count:
            recv[at(me)].copy_(send[me])                                 # own share stays on the card
        if N == 1:
            return
        if self.stage:
            self._hsend.copy_(send)
            torch.cuda.current_stream().synchronize()
            src, dst = self._hsend, [self._hrecv[at(r)] for r in range(N)]
        else:
            src, dst = send, [recv[at(r)] for r in range(N)]
        ops = []
        for j in range(count):
            if j != me and self.rows_local:
                ops.append(dist.P2POp(dist.isend, src[j], j, self.group))
        if me < count:
            for r in range(N):
                if r != me and self.rows_of[r]:
                    ops.append(dist.P2POp(dist.irecv, dst[r], r, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()              # RCCL: the current stream waits; gloo: the host blocks
        if self.stage and me < count:
            for r in range(N):
                if r != me:
                    recv[at(r)].copy_(dst[r], non_blocking=True)

    def _assemble(self, b: int, colorize_fn, frame_index: int) -> None:
        N = self.world
        frame = self.frame_nu[b] if self.payload == "nu" else self.frame_rgba[b]
        if self.bands:
            pass                                                         # every band was received in place
        elif self.even:
            # frame rows = [strip s of rank 0, strip s of rank 1, ...]: (S, N, R, W) <- (N, S, R, W) transposed
            S = self.H // (N * self.R)
            tail = tuple(frame.shape[2:])
            frame.view((S, N, self.R, self.W) + tail).copy_(
                self.recv_all[b].view((N, S, self.R, self.W) + tail).transpose(0, 1))
        else:
            for p, part in enumerate(self.recv[b]):
                if part.shape[0]:
                    frame.index_copy_(0, self._index[p], part)
        if self.payload == "nu":
            colorize_fn(self.frame_nu[b], self.frame_rgba[b], frame_index)

    def submit_group(self, render_fn: Callable, first_frame: int, count: int,
                     colorize_fn: Optional[Callable] = None) -> int:
        """Render and exchange frames first_frame .. first_frame + count - 1 (count <= world); frame
        first_frame + j lands on rank j.  Returns the slot; this rank's frame (if it got one) is valid
        after wait(slot) / drain() in frame_rgba[slot] (and frame_nu[slot] for the nu payload)."""
        if not 1 <= count <= self.world:
            raise ValueError("a group holds 1..world frames")
        if self.payload == "nu" and colorize_fn is None:
            raise ValueError("the nu payload needs a colorize_fn")
        b = self._groups % self.nslots
        self._groups += 1
        mine = first_frame + self.rank if self.rank < count else -1

        def render_all(lane=0, nlanes=1):
            for j in range(lane, count, nlanes):
                if self.rows_local:
                    shard = Shard((self.rank + j) % self.world, self.world, self.R) if self.bands else self.shard
                    render_fn(shard, self.send[b][j], first_frame + j, self.payload, lane)

        def deliver():
            self._exchange(b, count)
            if mine >= 0:
                self._assemble(b, colorize_fn, mine)

        if not self.on_gpu:
            render_all()
            deliver()
        else:
            for k, lane in enumerate(self.lanes):
                if self._used[b]:
                    lane.wait_event(self.delivered[b])         # the slot's send buffers are free again
                with torch.cuda.stream(lane):
                    render_all(k, self.nlanes)
                    self.rendered[b][k].record(lane)
            with torch.cuda.stream(self.comm):
                for k in range(self.nlanes):
                    self.comm.wait_event(self.rendered[b][k])
                deliver()
                self.delivered[b].record(self.comm)
            self._used[b] = True
        self.frame_index[b] = mine
        return b

    def prime(self) -> None:
        """Open every pairwise connection (RCCL sets up a peer connection on first use) before timing."""
        if self.world == 1:
            return
        dev = self.device if not self.stage else torch.device("cpu")
        tx = torch.zeros(self.world, 16, device=dev)
        rx = torch.zeros(self.world, 16, device=dev)
        ops = []
        for r in range(self.world):
            if r != self.rank:
                ops.append(dist.P2POp(dist.isend, tx[r], r, self.group))
                ops.append(dist.P2POp(dist.irecv, rx[r], r, self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.on_gpu:
            torch.cuda.synchronize(self.device)

    def wait(self, slot: int) -> None:
        if self.on_gpu and self._used[slot]:
            self.delivered[slot].synchronize()

    def drain(self) -> None:
        if self.on_gpu:
            for lane in getattr(self, "lanes", [self.compute]):
                lane.synchronize()
            self.comm.synchronize()


def export_animation(anim, renderers, output_folder: str, *, fractal_type=None, precision=None,
                     width: Optional[int] = None, height: Optional[int] = None, frames=None,
                     device: Optional[torch.device] = None, rows_per_strip: int = 0, group=None) -> List[str]:
    """AnimationRenderer::start_render (src/animation_renderer.cpp:26-152) over the GPUs of a node: the
    BASELINE.json C5 pipeline (.franim sweep, row-band sharded frames, RCCL exchange).

    Every frame is cut into row strips over all ranks; frames are processed in groups of `world`, frame
    j of a group is gathered to rank j (FrameExchange), which then runs what the reference's
    RenderFrameCallback does after its dispatch (src/vk_engine.cpp:1266-1381): the 8-bit export (second
    tonemap, u8, flip -- fr_export_rgb8 on the GPU) and the PNG (`frame_%06d.png`), so the readback and
    the deflate are spread over the ranks too.  The files are byte-identical to the ones
    Renderer.render_frame writes on one GPU.  `renderers`: this rank's render contexts, one per render
    lane.  Returns the paths THIS rank wrote.  Call on every rank with the same arguments."""
    import numpy as np  # noqa: F401  (torch -> numpy for the PNG writer)
    from .renderer import write_png, frame_path, Renderer
    from .state import FractalType, Precision

    fractal_type = FractalType.Mandelbrot if fractal_type is None else fractal_type
    precision = Precision.F32 if precision is None else precision       # what the reference's shaders compute in
    info = anim.info
    if info.keyframe_count < 2:
        raise ValueError("need at least 2 keyframes")                    # src/animation_renderer.cpp:35-42
    W, H = width or info.export_width, height or info.export_height
    total = anim.frame_count()                                           # :48
    todo = [f for f in (frames if frames is not None else range(total)) if 0 <= f < total]
    states = {f: anim.interpolate(anim.frame_time(f)) for f in todo}     # :80-83
    r0 = renderers[0]
    deep = fractal_type == FractalType.Deep_Zoom
    payload = "nu" if all(Renderer.colorize_supported(s, fractal_type, precision) for s in states.values()) else "rgba"
    nu_dtype = torch.float64 if precision == Precision.F64 else torch.float32
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    # consecutive animation frames cost about the same band by band: rotating bands balance over a group and are
    # received in place (rows_per_strip given: the caller asks for interleaved strips)
    fx = FrameExchange(W, H, payload=payload, nu_dtype=nu_dtype, device=dev, rows_per_strip=rows_per_strip,
                       group=group, render_lanes=len(renderers), layout="strips" if rows_per_strip else "bands")

    def render_fn(shard, out, idx, plane, lane=0):
        # the storage image of the reference holds the post-chained colour (shaders/mandelbrot.comp:233-237)
        renderers[lane].render(states[todo[idx]], W, H, fractal_type=fractal_type, precision=precision,
                               post_chain=(plane == "rgba" and not deep), shard=shard, sync=False,
                               stream=torch.cuda.current_stream().cuda_stream, **{plane: out})

    def colorize_fn(nu_frame, rgba_frame, idx):
        r0.colorize(states[todo[idx]], nu_frame, rgba_frame, fractal_type=fractal_type, precision=precision,
                    post_chain=True, stream=torch.cuda.current_stream().cuda_stream)

    written: List[str] = []
    # the PNG writer deflates on worker threads: share the host's cores between the ranks of the node
    import os as _os
    _os.environ.setdefault("FR_PNG_THREADS", str(max(1, (_os.cpu_count() or 1) // max(1, fx.world))))

    def finish(slot):
        fx.wait(slot)
        idx = fx.frame_index[slot]
        if idx < 0:
            return
        rgb8 = r0.export_rgb8(fx.frame_rgba[slot], W, H, through_half=True)      # src/vk_engine.cpp:1313-1371
        path = frame_path(output_folder, todo[idx])                              # src/animation_renderer.cpp:86-88
        write_png(path, rgb8.cpu().numpy())                                      # src/vk_engine.cpp:1374-1381
        written.append(path)

    fx.prime()
    pending = []
    i = 0
    while i < len(todo):
        count = min(fx.world, len(todo) - i)
        pending.append(fx.submit_group(render_fn, i, count, colorize_fn if payload == "nu" else None))
        if len(pending) == fx.nslots:            # group g's export + PNG overlap the rendering of group g+1
            finish(pending.pop(0))
        i += count
    for slot in pending:
        finish(slot)
    fx.drain()
    return written
