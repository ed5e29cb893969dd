"""Head-parallel GAT levels: one process per GPU, heads sharded over the ranks.

The reference has no parallelism at all (single process, heads looped in Python,
models.py:32,34).  Heads are independent given the input x and the graph, so the
natural MI355X mapping is head-per-GPU:

  hidden level (models.py:32, torch.cat):   rank r computes its heads' columns, then an
      RCCL ALL-GATHER over xGMI concatenates them; its backward is the matching
      REDUCE-SCATTER (each rank back-propagated only its own next-level heads, so the
      incoming gradients are partial sums).
  last level (models.py:34, mean of stack): every rank averages its local heads, scaled by
      h_loc/H; an ALL-REDUCE(sum) finishes the mean; backward is the identity.

Parameter gradients stay local (model parallelism: no gradient all-reduce).  x and
the CSR graph are replicated.  `level_fn` lets the CPU tests (gloo, world_size 2)
swap the HIP level for the oracle to check the sharding algebra without a GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def partition_heads(H: int, world: int) -> List[Tuple[int, int]]:
    """Balanced contiguous blocks; the first H % world ranks get one head more."""
    base, rem = divmod(H, world)
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((s, s + n))
        s += n
    return out


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# Test hook: with a process group of ONE rank the collectives below are still issued (a 1-GPU box can host one RCCL rank:
# tests/test_gpu_dist.py drives all_gather_into_tensor(async_op=True), reduce_scatter_tensor and all_reduce through the
# "nccl" backend this way).  Off: a world of one takes the short cuts.
FORCE_COLLECTIVES = False


def _single():
    """True when the collectives may be skipped (no process group, or one rank and no FORCE_COLLECTIVES)."""
    _, world = _world()
    return world == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def gather_columns_layout(buf: torch.Tensor) -> torch.Tensor:
    """[world, N, w] (what an all-gather delivers: rank-major column blocks) -> [N, world*w] (torch.cat(dim=1),
    models.py:32).  One strided copy."""
    world, N, w = buf.shape
    return buf.permute(1, 0, 2).reshape(N, world * w)


def all_gather_columns_raw(local: torch.Tensor, widths: Sequence[int]) -> torch.Tensor:
    """[N, widths[rank]] on every rank -> [N, sum(widths)] on every rank (no autograd)."""
    rank, world = _world()
    if _single():
        return local
    N, wmax = local.shape[0], max(widths)
    if all(w == wmax for w in widths):
        buf = torch.empty(world * N, wmax, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, local.contiguous())
        return gather_columns_layout(buf.view(world, N, wmax))
    pad = torch.zeros(N, wmax, dtype=local.dtype, device=local.device)
    pad[:, :widths[rank]] = local
    buf = torch.empty(world * N, wmax, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad)
    buf = buf.view(world, N, wmax)
    return torch.cat([buf[r, :, :widths[r]] for r in range(world)], dim=1)


class AllGatherColumns(torch.autograd.Function):
    """forward: all-gather of column blocks; backward: reduce-scatter(sum) of the gradient."""

    @staticmethod
    def forward(ctx, local, widths):
        ctx.widths = tuple(widths)
        return all_gather_columns_raw(local, widths)

    @staticmethod
    def backward(ctx, G):
        rank, world = _world()
        widths = ctx.widths
        if _single():
            return G, None
        N, wmax = G.shape[0], max(widths)
        offs = [0]
        for w in widths:
            offs.append(offs[-1] + w)
        if all(w == wmax for w in widths):
            # [N, world w] -> [world, N, w], the layout the reduce-scatter sends: ONE strided copy (rounds 1-3: a zero fill and
            # a Python loop of `world` slice copies).  Writing the input gradient in this layout to begin with needs a
            # column-blocked C in the dX GEMM of the next level: DESIGN.md section 5, not built.
            stacked = G.reshape(N, world, wmax).permute(1, 0, 2).contiguous()
        else:
            stacked = torch.zeros(world, N, wmax, dtype=G.dtype, device=G.device)
            for r in range(world):
                stacked[r, :, :widths[r]] = G[:, offs[r]:offs[r + 1]]
        out = torch.empty(N, wmax, dtype=G.dtype, device=G.device)
        if dist.get_backend() == "nccl":
            dist.reduce_scatter_tensor(out, stacked.view(world * N, wmax), op=dist.ReduceOp.SUM)
        else:  # gloo has no reduce_scatter: all-reduce then slice (CPU tests, 1-card rehearsals)
            dist.all_reduce(stacked, op=dist.ReduceOp.SUM)
            out = stacked[rank]
        return out[:, :widths[rank]].contiguous(), None


import os as _os

# row chunks of a pipelined hidden level (1 = no pipeline).  PYGAT_DIST_CHUNKS is read ONCE, at import: every rank must issue
# the same number and shapes of collectives, so the count may not change under a running job (a per-call read on each rank
# could disagree and hang); set the environment identically on all ranks, or assign dist.PIPELINE_CHUNKS on all of them.
PIPELINE_CHUNKS = int(_os.environ.get("PYGAT_DIST_CHUNKS", 4))
PIPELINE_MIN_ROWS = 1 << 15  # below this a level is launch-bound and one blocking all-gather is cheaper


_layout = {}


def _layout_stream(dev) -> "torch.cuda.Stream":
    """One side stream per device for the layout copies of gathered chunks."""
    key = torch.device(dev).index
    if key not in _layout:
        _layout[key] = torch.cuda.Stream(device=dev)
    return _layout[key]


class _GatheredColumns(torch.autograd.Function):
    """Ties the gathered activation `full` (assembled chunk by chunk from the ranks' column blocks while the
    level was still running) to the local block it was gathered from: forward returns `full`, backward is the
    reduce-scatter(sum) of AllGatherColumns."""

    @staticmethod
    def forward(ctx, local, full, widths):
        ctx.widths = tuple(widths)
        return full

    @staticmethod
    def backward(ctx, G):
        return AllGatherColumns.backward(ctx, G)[0], None, None


def _pipelined_concat_level(x, graph, Ws, As, sk, alpha, widths, nchunks):
    """Hidden level with its heads sharded over the ranks, row-chunk pipelined: K2 runs chunk by
    chunk; as soon as chunk c's launches are enqueued its rows are all-gathered on RCCL's stream (which first waits
    for them), so the exchange of chunk c overlaps the computation of chunk c+1, and so on.  The gathered chunks
    arrive as [world, rows, w] blocks and are copied into their column slices of the [N, sum(widths)] activation
    as they land."""
    from .ops import gat_level
    rank, world = _world()
    N, w = x.shape[0], widths[rank]
    full = torch.empty(N, sum(widths), dtype=torch.float32, device=x.device)
    works = []
    main = torch.cuda.current_stream(x.device)
    side = _layout_stream(x.device)
    side.wait_stream(main)                                  # `full` was allocated on the compute stream

    def land(item):
        """chunk `item` into its column slices of `full`, on the SIDE stream: it waits for the chunk's all-gather there and
        the strided copy runs beside the computation of the later chunks (round 4; rounds 2-3 made all these copies on the
        compute stream after the level, ~0.17 ms per level at config 5)."""
        work, buf, r0, r1 = item
        with torch.cuda.stream(side):
            work.wait()
            full[r0:r1].view(r1 - r0, world, w).copy_(buf.permute(1, 0, 2))
        buf.record_stream(side)

    def on_chunk(c, r0, r1, out):
        buf = torch.empty(world, r1 - r0, w, dtype=out.dtype, device=out.device)
        works.append((dist.all_gather_into_tensor(buf.view(world * (r1 - r0), w), out[r0:r1], async_op=True), buf, r0, r1))
        if c >= 1:
            land(works[c - 1])                              # the previous chunk's exchange has had a chunk of compute to finish

    local = gat_level(x, graph, Ws, As, sk, alpha, True, pipeline=(nchunks, on_chunk))
    land(works[-1])
    main.wait_stream(side)                                  # the next level reads `full` on the compute stream
    return _GatheredColumns.apply(local, full, widths)


class AllReduceSum(torch.autograd.Function):
    """forward: all-reduce(sum); backward: identity (the result is replicated)."""

    @staticmethod
    def forward(ctx, t):
        if _single():
            return t
        t = t.contiguous().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    @staticmethod
    def backward(ctx, G):
        return G


def gat_level_head_parallel(x, graph, Ws, As, Wskips, alpha: float, concat: bool, dropout: float = 0.0,
                            level_fn: Optional[Callable] = None) -> torch.Tensor:
    """One level with its heads sharded over the ranks; returns the full (replicated) output."""
    rank, world = _world()
    H = len(Ws)
    Fo = Ws[0].shape[1]
    parts = partition_heads(H, world)
    s, e = parts[rank]
    hip_level = level_fn is None
    if level_fn is None:
        if dropout > 0.0:
            from .dropout import gat_level_dropout
            level_fn = lambda x_, g_, W_, a_, sk_, al_, cc_: gat_level_dropout(  # noqa: E731
                x_, g_, W_, a_, sk_, al_, cc_, dropout, head_mean=not cc_)
        else:
            from .ops import gat_level
            level_fn = gat_level
    sk = None if Wskips is None else list(Wskips[s:e])
    widths = [(b - a) * Fo for a, b in parts]
    nchunks = PIPELINE_CHUNKS
    if (concat and hip_level and dropout == 0.0 and not _single() and nchunks > 1 and x.shape[0] >= PIPELINE_MIN_ROWS
            and len(set(widths)) == 1 and e > s):
        return _pipelined_concat_level(x, graph, list(Ws[s:e]), list(As[s:e]), sk, alpha, widths, nchunks)
    if concat:
        if e > s:
            local = level_fn(x, graph, list(Ws[s:e]), list(As[s:e]), sk, alpha, True)
        else:
            # no local head: stay connected to x so this rank still takes part in the backward
            # reduce-scatter of the previous level (with a zero contribution)
            local = x[:, :0] * 0.0
            if not local.requires_grad:     # x is a graph input: become a leaf so backward still runs here
                local = local.detach().requires_grad_(True)
        return AllGatherColumns.apply(local, widths)
    if e > s:
        local = level_fn(x, graph, list(Ws[s:e]), list(As[s:e]), sk, alpha, False) * ((e - s) / H)
    else:
        local = x[:, :1].expand(x.shape[0], Fo) * 0.0   # zero contribution, autograd-connected to x
    return AllReduceSum.apply(local)
