"""Frame-sharded data parallelism for the video rasterizer.

All frames of a clip share one Gaussian parameter set, so the natural multi-GPU axis is the
frame: every rank holds a replica, renders the frames ``f = rank (mod world)`` of the step's frame
batch forward+backward, and the Gaussian gradients (accumulated locally over the rank's frames in
ONE flat fp32 bucket) are summed with a single ``all_reduce`` per optimiser step -- RCCL over xGMI on
MI355X (``backend="nccl"``), gloo in the CPU tests.  The reference initialises a process group and a
DistributedSampler but never synchronises gradients (src/train.py:31,212); this is the missing
piece, not a translation of anything.

The module is backend-agnostic plumbing (no kernels): it only depends on torch / torch.distributed.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def frames_of_rank(frames: Sequence[int], rank: int, world: int) -> List[int]:
    """Round-robin frame shard: rank r takes frames[r], frames[r+world], ..."""
    return list(frames[rank::world])


class FlatGradBucket:
    """Parameters as views of one flat tensor, gradients as views of another.

    ``bucket.params[name]`` are leaf tensors (``requires_grad=True``) whose ``.grad`` aliases a slice of
    ``bucket.flat_grad``; autograd (or a kernel with a gradient sink) accumulates in place, so after the local frames'
    backward passes ``all_reduce()`` moves exactly one contiguous buffer.

    ``buffers=2`` double-buffers the gradients: ``swap()`` at the start of a step points every ``.grad`` (and
    ``grad(name)``) at the other buffer, ``all_reduce(async_op=True)`` at its end leaves the collective running on
    RCCL's stream while the next step's frames fill the other buffer -- the reduction is off the critical path instead
    of between two steps (``swap()`` waits for the collective that last used the buffer it switches to)."""

    def __init__(self, tensors: Dict[str, torch.Tensor], buffers: int = 1, pad_to: int = 1):
        if buffers < 1:
            raise ValueError("buffers must be >= 1")
        names = list(tensors)
        dev = tensors[names[0]].device
        total = sum(t.numel() for t in tensors.values())
        # ``pad_to``: the flat buffers' length rounded up to a multiple (ZeRO-1 deals the buffer in `world` equal blocks of whole
        # float4: pad_to = 4 * world).  The tail belongs to no parameter: zero parameters, zero gradients, Adam leaves it at 0.
        total = -(-total // int(pad_to)) * int(pad_to)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grads = [torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(buffers)]
        self.pending = [None] * buffers            # outstanding collective per buffer
        self.active = 0
        self.params: Dict[str, torch.Tensor] = {}
        self.slices: Dict[str, Tuple[int, int]] = {}
        o = 0
        for n in names:
            t = tensors[n]
            k = t.numel()
            v = self.flat_param[o:o + k].view(t.shape)
            with torch.no_grad():
                v.copy_(t)
            v.requires_grad_(True)
            self.params[n] = v
            self.slices[n] = (o, o + k)
            o += k
        self._point_grads()

    @property
    def flat_grad(self) -> torch.Tensor:
        return self.flat_grads[self.active]

    def _point_grads(self) -> None:
        for n, v in self.params.items():
            v.grad = self.grad(n)

    def swap(self) -> None:
        """make the next buffer the active one (no-op with a single buffer), after its last collective has finished"""
        if len(self.flat_grads) == 1:
            self.wait(0)
            return
        self.active = (self.active + 1) % len(self.flat_grads)
        self.wait(self.active)
        self._point_grads()

    def zero_grad(self) -> None:
        """the active gradient buffer := 0.  On the GPU one float4 fill launch of the library (splat_fill_f32) on the current
        stream; the CPU buffers of the gloo tests use the framework's."""
        g = self.flat_grad
        if g.is_cuda:
            import ctypes

            from . import _lib as L
            L.check(L.lib().splat_fill_f32(L.ptr(g), ctypes.c_size_t(g.numel()), L.cf(0.0), L.stream()))
        else:
            g.zero_()

    def activate(self, buffer: int) -> None:
        """make ``buffer`` the active gradient buffer (after its last collective has finished)"""
        self.wait(buffer)
        self.active = buffer
        self._point_grads()

    def all_reduce(self, average: bool = False, async_op: bool = False) -> None:
        """One collective per step over the active buffer; a no-op for a single process.  ``async_op``: return at once,
        ``wait()`` / ``swap()`` synchronise later."""
        if dist.is_available() and dist.is_initialized():
            flat = self.flat_grad
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)  # world size 1: identity, still exercises RCCL
            if async_op:
                self.pending[self.active] = (work, average)
            elif average:
                flat.div_(dist.get_world_size())

    def fold(self, src: int, dst: int) -> None:
        """flat_grads[dst] += flat_grads[src] (both reductions finished): the two half-batch sums of overlapped_halves_step"""
        self.wait(src)
        self.wait(dst)
        self.flat_grads[dst].add_(self.flat_grads[src])

    def wait(self, buffer: int = None) -> None:
        """finish the outstanding collective of ``buffer`` (default: all buffers)"""
        for b in (range(len(self.flat_grads)) if buffer is None else (buffer,)):
            if self.pending[b] is not None:
                work, average = self.pending[b]
                work.wait()
                if average:
                    self.flat_grads[b].div_(dist.get_world_size())
                self.pending[b] = None

    def grad(self, name: str, buffer: int = None) -> torch.Tensor:
        a, b = self.slices[name]
        flat = self.flat_grads[self.active if buffer is None else buffer]
        return flat[a:b].view(self.params[name].shape)


def reduce_visibility(visibility: torch.Tensor, radii: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Densification statistics the renderer ORs / maxes over its frames
    (reference: dptr_ortho_enhanced.py:430-431) reduced over the ranks as well."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        v = visibility.to(torch.int32)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        r = radii.clone()
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        return v.bool(), r
    return visibility, radii


def reduce_densify_batch(viewspace_grad: torch.Tensor, visibility: torch.Tensor, radii: torch.Tensor) -> None:
    """The batch statistics densification consumes, reduced over the ranks IN PLACE so that every rank takes the same
    clone / split / prune decisions: the reference's single process sums the gradient taps of all frames of a batch, ORs
    their visibility and maxes their radii (dptr_ortho_enhanced.py:425-431, frag_model.py:326-343); with the batch's
    frames spread over the ranks that is SUM / MAX / MAX across the process group.  Call it once per step, after the
    local frames' backward passes and before ``DensifyState.update()``.  No-op without a process group."""
    if not _group_on():
        return
    dist.all_reduce(viewspace_grad, op=dist.ReduceOp.SUM)
    v = visibility.to(torch.int32)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    visibility.copy_(v.to(visibility.dtype))
    dist.all_reduce(radii, op=dist.ReduceOp.MAX)


def overlapped_halves_step(bucket: FlatGradBucket, first_half, second_half, optimizer=None, average: bool = False) -> None:
    """One gradient step with HALF of the all-reduce hidden and the SAME result as ``sharded_step`` (to fp32 summation
    order): the rank's frames are split into two half-batches with a gradient buffer each (``FlatGradBucket(buffers=2)``);
    the first half's buffer is all-reduced asynchronously (RCCL's own stream) UNDER the second half's forward + backward,
    the second half's buffer synchronously behind it, the two sums are added and the optimiser steps once.  Nothing is
    stale: the parameters change only after both halves' gradients have been reduced.  ``first_half`` / ``second_half``
    render their frames forward + backward into the bucket's ACTIVE buffer (``bucket.grad(name)`` / ``.grad`` of the
    parameters at call time).  What stays exposed is the second half's collective (half of the bytes when the link is
    bandwidth bound) -- every gradient of a frame-sharded step completes in its last kernels, so a full-size bucket
    cannot be hidden without staleness."""
    if len(bucket.flat_grads) < 2:
        raise ValueError("overlapped_halves_step needs FlatGradBucket(buffers=2)")
    bucket.activate(0)
    bucket.zero_grad()
    first_half()
    bucket.all_reduce(async_op=True)
    bucket.activate(1)
    bucket.zero_grad()
    second_half()
    bucket.all_reduce(async_op=False)
    bucket.fold(0, 1)                     # waits for the first half's collective; the sum sits in the active buffer
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if optimizer is not None:
        optimizer.step(grad_scale=(1.0 / world) if average else 1.0)
    elif average and world > 1:
        bucket.flat_grad.mul_(1.0 / world)    # a caller with its own optimiser gets the MEAN it asked for, not the sum


def sharded_step(bucket: FlatGradBucket, frames: Iterable[int], render_and_backward, optimizer=None,
                 average: bool = False) -> None:
    """One SYNCHRONOUS data-parallel gradient step: zero -> local frames forward+backward (gradients accumulate in the
    bucket) -> one all-reduce -> the optimiser step every rank applies identically -> (caller) next step's forward.
    ``optimizer`` is anything with ``step(grad_scale=...)`` over the bucket (``optim.FlatAdam`` on the GPU)."""
    bucket.zero_grad()
    for f in frames:
        render_and_backward(f)
    bucket.all_reduce()
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if optimizer is not None:
        optimizer.step(grad_scale=(1.0 / world) if average else 1.0)
    elif average and world > 1:
        bucket.flat_grad.mul_(1.0 / world)    # (no optimiser: the averaged gradient stays in the bucket)


# ---------------------------------------------------------------------------------------------------------------------
# Owner-sharded step: the spline table of the reference's model is 87 % of its parameters (4 * I * 3 floats per Gaussian against
# 72 for everything else at 200 frames, src/dynamic_gaussian_with_base_point_cloud.py:73-75) and is addressed BY TIME: segment s
# serves the frames of its five-frame interval.  With the clip's frames dealt to the ranks in contiguous time blocks, rank r
# OWNS the segments of its block: their gradient is reduced TO the owner only (reduce-scatter), the owner alone keeps their Adam
# moments and steps them, and the updated blocks are all-gathered, so that every replica still evaluates position(ids2) of any
# pair frame.  Same bytes on the wire as the ring all-reduce (2 (W-1)/W B per rank), but the optimiser streams 1/W of the
# table's 28 bytes per parameter, its state takes 1/W of the memory, and the two halves are separate collectives: the
# reduce-scatter can start as soon as the backward's last kernel is done while the replicated part is still in flight.
class OwnerShards:
    """The flat buffer split into a REPLICATED part and an OWNED parameter (``name``: leading dimension = the units that are
    dealt out, e.g. the spline table stored segment-major [I, N, 4, 3]) cut into ``world`` contiguous blocks of whole units:
    rank r owns units [units * r // world, units * (r + 1) // world).  The owned parameter must come FIRST in the bucket (its
    blocks then start 16-byte aligned for the streaming Adam kernel)."""

    def __init__(self, bucket: FlatGradBucket, name: str, world: int, rank: int):
        a, b = bucket.slices[name]
        if a != 0:
            raise ValueError(f"the owned parameter {name!r} must be the first tensor of the bucket")
        units = int(bucket.params[name].shape[0])
        per = (b - a) // units
        if units < world:
            raise ValueError(f"{units} units cannot be dealt to {world} ranks")
        self.name, self.world, self.rank = name, int(world), int(rank)
        self.unit_bounds = [units * r // world for r in range(world + 1)]
        self.bounds = [a + u * per for u in self.unit_bounds]
        self.a, self.b, self.total = a, b, bucket.flat_param.numel()
        self.equal = len({self.bounds[r + 1] - self.bounds[r] for r in range(world)}) == 1

    @property
    def own(self) -> Tuple[int, int]:
        return self.bounds[self.rank], self.bounds[self.rank + 1]

    def frames_of_rank(self, unit_of_frame: Sequence[int]) -> List[int]:
        """the clip's frames in contiguous TIME BLOCKS: rank r renders the frames whose unit (``unit_of_frame[f]``: the spline
        segment of frame f, ``FrameClock.scalars(f)[0]``) it owns -- their render-path gradient of the table is then already
        complete on its owner"""
        u0, u1 = self.unit_bounds[self.rank], self.unit_bounds[self.rank + 1]
        return [f for f, u in enumerate(unit_of_frame) if u0 <= u < u1]


# A process group of ONE rank normally skips the owner collectives (nothing to exchange).  ``MIN_WORLD = 1`` makes them run
# anyway: the RCCL first-contact test and ``bench.py --force-process-group`` execute every collective signature of the 8-GPU
# run -- reduce_scatter_tensor, all_gather_into_tensor, the asynchronous all-reduce beside them -- on the one GPU a box has.
MIN_WORLD = 2


def _group_on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() >= MIN_WORLD


def owner_reduce(bucket: FlatGradBucket, shards, reduce_owned: bool = True) -> None:
    """gradients after the local backward: all-reduce of the replicated part || reduce-scatter of the owned part (afterwards a
    rank's gradient of the owned parameter is complete in ITS block only; the other blocks hold partial sums nobody reads).
    ``shards``: ``OwnerShards`` (one owned parameter + a replicated rest) or ``Zero1Shards`` (the whole buffer is owned).
    ``reduce_owned=False``: the owned part's gradient is already complete on its owner (position exchange): only the replicated
    part is reduced."""
    if not _group_on():
        return
    if not reduce_owned:
        if shards.total > shards.b:
            dist.all_reduce(bucket.flat_grad[shards.b:shards.total], op=dist.ReduceOp.SUM)
        return
    world = dist.get_world_size()
    g = bucket.flat_grad
    lo, hi = shards.own
    # the replicated part's all-reduce is issued first and waited for last: on RCCL both collectives are enqueued, in this
    # order on every rank, on the process group's one stream -- the reduce-scatter runs behind it, not beside it
    work = None
    if shards.total > shards.b:
        work = dist.all_reduce(g[shards.b:shards.total], op=dist.ReduceOp.SUM, async_op=True)
    if dist.get_backend() == "nccl" and shards.equal:
        own = torch.empty(hi - lo, dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(own, g[shards.a:shards.b], op=dist.ReduceOp.SUM)
        g[lo:hi].copy_(own)
    else:   # uneven blocks / backends without reduce-scatter (gloo): one reduce per owner -- the same bytes
        for r in range(world):
            dist.reduce(g[shards.bounds[r]:shards.bounds[r + 1]], dst=r, op=dist.ReduceOp.SUM)
    if work is not None:
        work.wait()


def owner_gather(bucket: FlatGradBucket, shards, flat: torch.Tensor = None) -> None:
    """the owners' updated blocks to every rank (``flat``: another buffer laid out like the bucket, e.g. assembled moments)"""
    if not _group_on():
        return
    p = bucket.flat_param if flat is None else flat
    lo, hi = shards.own
    with torch.no_grad():
        if dist.get_backend() == "nccl" and shards.equal:
            dist.all_gather_into_tensor(p[shards.a:shards.b], p[lo:hi].clone())
        else:
            for r in range(dist.get_world_size()):
                dist.broadcast(p[shards.bounds[r]:shards.bounds[r + 1]], src=r)


def owner_sharded_step(bucket: FlatGradBucket, shards, frames: Iterable[int], render_and_backward, optimizer,
                       average: bool = False) -> None:
    """One SYNCHRONOUS step with the owned parameter's gradient reduced to its owners only: zero -> local frames forward +
    backward -> all-reduce of the replicated part || reduce-scatter of the owned part -> ``optimizer.step`` (replicated part on
    every rank, own block on its owner: ``optim.OwnerShardedAdam``) -> all-gather of the updated blocks.  Parameters after the
    step = ``sharded_step``'s (to fp32 summation order), replicas bit-identical."""
    bucket.zero_grad()
    for f in frames:
        render_and_backward(f)
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    world = dist.get_world_size() if on else 1
    owner_reduce(bucket, shards)
    optimizer.step(grad_scale=(1.0 / world) if average else 1.0)
    owner_gather(bucket, shards)


class Zero1Shards:
    """ZeRO-1 over the WHOLE flat buffer: `world` equal blocks of whole float4 (the bucket is built with
    ``FlatGradBucket(..., pad_to=4 * world)``), rank r owns block r of the parameters' Adam moments and steps only that block:
    reduce-scatter of the gradient -> Adam on 1/world of the buffer -> all-gather of the updated parameters.  On the wire the
    bytes of the ring all-reduce, as two collectives RCCL can run over all seven xGMI links of a rank (every peer receives its
    own block); the optimiser streams and stores 1/world.  Same interface as ``OwnerShards`` with nothing replicated:
    ``owner_reduce`` / ``owner_gather`` / ``owner_sharded_step`` / ``optim.OwnerShardedAdam`` take either."""

    def __init__(self, bucket: FlatGradBucket, world: int, rank: int):
        total = bucket.flat_param.numel()
        if total % (4 * int(world)):
            raise ValueError(f"the flat buffer ({total} floats) is not {world} blocks of whole float4: build the bucket with "
                             f"FlatGradBucket(..., pad_to={4 * int(world)})")
        per = total // int(world)
        self.name, self.world, self.rank = None, int(world), int(rank)
        self.a, self.b, self.total = 0, total, total
        self.bounds = [r * per for r in range(int(world) + 1)]
        self.equal = True

    @property
    def own(self) -> Tuple[int, int]:
        return self.bounds[self.rank], self.bounds[self.rank + 1]


def zero1_step(bucket: FlatGradBucket, shards: Zero1Shards, frames: Iterable[int], render_and_backward, optimizer,
               average: bool = False) -> None:
    """``owner_sharded_step`` with the whole buffer owned (ZeRO-1): parameters after the step = ``sharded_step``'s to fp32
    summation order, replicas bit-identical (every rank receives the owners' bits)."""
    owner_sharded_step(bucket, shards, frames, render_and_backward, optimizer, average=average)


# ---------------------------------------------------------------------------------------------------------------------
# Position exchange: with the spline table owner-sharded by time blocks, what a rank needs of ANOTHER block is never
# the table -- 14.4 MB per segment at 300k Gaussians -- but position(t) of the pair frames it drew there (3.6 MB per frame), and
# what it owes the owner afterwards is the gradient of that position.  The owner evaluates, sends, later receives dL/dposition and
# runs the positions' backward into ITS block; the table, its gradient and its Adam moments never leave their owner: per rank and
# step 2 F frames of [N, 3] instead of 2 (W - 1) / W of the table's gradient + parameters (331 instead of 1159 MB at 8 GPUs and 200
# frames).
class PositionExchangePlan:
    """Who evaluates which requested frame.  ``times2_all[r][k]`` = pair frame k of rank r (every rank holds the whole table of
    requests: one small all-gather); ``owner_of(t)`` = rank that owns the spline segment of time t.  ``serve`` = the (requester,
    k, time) this rank evaluates, ordered by (requester, k) -- the order of the messages between any two ranks on both sides."""

    def __init__(self, times2_all, owner_of, rank: int):
        self.rank = int(rank)
        self.owners = [[int(owner_of(t)) for t in row] for row in times2_all]
        self.serve = [(r, k, float(t)) for r, row in enumerate(times2_all) for k, t in enumerate(row) if self.owners[r][k] == self.rank]
        self.mine = self.owners[self.rank]          # owner of each of this rank's requested frames


def gather_times(times, world: int, rank: int):
    """every rank's list of frame times (host floats) on every rank: [world][len(times)].  A host synchronisation per step -- the
    frame tables are built on the host anyway."""
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not on:
        return [[float(t) for t in times]]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(t) for t in times], dtype=torch.float64, device=dev)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [[float(x) for x in o.cpu().tolist()] for o in out]


def exchange_frames(sends, recvs) -> None:
    """point-to-point exchange of whole tensors: ``sends`` = [(peer, tensor)], ``recvs`` = [(peer, tensor)] (dense [N, 3] frames;
    between any two ranks both sides list their messages in the same order).  RCCL: one grouped batch of isend / irecv on the
    device tensors; gloo (the CPU rehearsal of the tests: it moves no CUDA tensors point to point): staged through host copies."""
    if not sends and not recvs:
        return
    if dist.get_backend() == "nccl":
        ops = [dist.P2POp(dist.isend, t, peer) for peer, t in sends] + [dist.P2POp(dist.irecv, t, peer) for peer, t in recvs]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return
    host_s = [(peer, t.detach().cpu().contiguous()) for peer, t in sends]
    host_r = [(peer, torch.empty(t.shape, dtype=t.dtype)) for peer, t in recvs]
    works = [dist.isend(h, peer) for peer, h in host_s] + [dist.irecv(h, peer) for peer, h in host_r]
    for w in works:
        w.wait()
    for (_, t), (_, h) in zip(recvs, host_r):
        t.copy_(h)
