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


from .config import config as _config

# row chunks of a pipelined hidden level (1 = no pipeline).  PYGAT_DIST_CHUNKS is read ONCE, at import: every rank must issue
# the same number and shapes of collectives, so the count may not change under a running job (a per-call read on each rank
# could disagree and hang); set the environment identically on all ranks, or assign dist.PIPELINE_CHUNKS on all of them.
PIPELINE_CHUNKS = _config.dist_chunks
PIPELINE_MIN_ROWS = 1 << 15  # below this a level is launch-bound and one exchange after the whole level is cheaper


# ---------------------------------------------------------------------------------------------------------------------
# The copy-free exchange (round 5).  The activation of a sharded hidden level lives COLUMN-BLOCKED: `full [world, N, w]`,
# block r = rank r's head columns of EVERY node (w = heads per rank x F').  Rank r's K2 writes its rows straight into
# full[r] (the level's output buffer); per row chunk [r0, r1) the exchange sends the contiguous view full[r, r0:r1] to
# every peer and receives the peers' views full[r', r0:r1] in place -- no staging buffer, no permute, no self copy (rounds
# 2-4 gathered into [world, rows, w] chunks and copied every chunk into [N, world w]: at config 5 a 512 MB permute
# beside HBM-bound kernels, 0.7 ms of a 3.1 ms step at RCCL world 1).  The next level reads the blocks in place: the
# projection / weight-gradient GEMMs take a column-blocked operand (include/pygat_amd.h, pygat_col_blocks: with whole-N
# blocks the block base is uniform per k-step), and the input gradient is written in the same blocks, which is the layout
# reduce_scatter_tensor sends (models.py:32 torch.cat and its autograd, without either copy).
# ---------------------------------------------------------------------------------------------------------------------
def blocked_width_ok(w: int) -> bool:
    """Block widths the GEMMs read in place: a power of two >= 16 floats (one 64-byte sector and up)."""
    return w >= 16 and (w & (w - 1)) == 0


def exchange_blocks(full: torch.Tensor, r0: int, r1: int):
    """Rows [r0, r1) of this rank's block full[rank] are enqueued on the current stream: deliver them into every peer's
    full[rank, r0:r1] and receive the peers' rows into full[peer, r0:r1].  Returns the pending works (wait() on each
    before `full` is read).  RCCL: ONE grouped send/recv launch (ncclGroupStart ... End: every pair's transfer runs
    concurrently over its own xGMI link -- the direct all-gather of SURVEY.md 8(e)); gloo (CPU tests, 1-card rehearsals):
    one broadcast per owner."""
    rank, world = _world()
    if dist.get_backend() == "nccl":
        if world == 1:      # FORCE_COLLECTIVES on one rank: the in-place all-gather of the (contiguous) view -- drives RCCL
            v = full[0, r0:r1]
            return [dist.all_gather_into_tensor(v, v, async_op=True)]
        if r0 == 0 and r1 == full.shape[1]:
            # the whole level at once (a level too small to pipeline, bench.py --chunks 1): [world, N, w] IS the all-gather's
            # output layout and full[rank] its in-place input -- RCCL's own all-gather, still without a copy
            N, w = full.shape[1], full.shape[2]
            return [dist.all_gather_into_tensor(full.view(world * N, w), full[rank], async_op=True)]
        ops = []
        for d in range(1, world):       # rank -> rank + d, rank - d -> rank: every step of the loop is a perfect matching
            ops.append(dist.P2POp(dist.isend, full[rank, r0:r1], (rank + d) % world))
            ops.append(dist.P2POp(dist.irecv, full[(rank - d) % world, r0:r1], (rank - d) % world))
        return list(dist.batch_isend_irecv(ops))
    return [dist.broadcast(full[o, r0:r1], src=o, async_op=True) for o in range(world)]


def reduce_scatter_blocks(G: torch.Tensor) -> torch.Tensor:
    """[world, N, w] partial gradients on every rank -> this rank's block [N, w], summed over the ranks: the layout is
    already the one reduce_scatter_tensor sends."""
    rank, world = _world()
    G = G.contiguous()
    _, N, w = G.shape
    if dist.get_backend() == "nccl":
        out = torch.empty(N, w, dtype=G.dtype, device=G.device)
        dist.reduce_scatter_tensor(out, G.view(world * N, w), op=dist.ReduceOp.SUM)
        return out
    G = G.clone()           # gloo has no reduce_scatter: all-reduce, then the own block (CPU tests, 1-card rehearsals)
    dist.all_reduce(G, op=dist.ReduceOp.SUM)
    return G[rank].contiguous()


def unblock(xb: torch.Tensor) -> torch.Tensor:
    """[world, N, w] -> [N, world w] (torch.cat(dim=1), models.py:32): ONE strided copy, differentiable.  Only for
    consumers that cannot read blocks (a level function injected by a test, dropout levels)."""
    world, N, w = xb.shape
    return xb.permute(1, 0, 2).reshape(N, world * w)


class _GatheredBlocks(torch.autograd.Function):
    """Ties the exchanged activation `full [world, N, w]` to the local block it was assembled around: forward returns
    `full`, backward is the reduce-scatter(sum) of the gradient blocks."""

    @staticmethod
    def forward(ctx, local, full):
        return full

    @staticmethod
    def backward(ctx, G):
        return reduce_scatter_blocks(G), None


def _blocked_concat_level(x, graph, Ws, As, sk, alpha, w, nchunks):
    """Hidden level with its heads sharded over the ranks, copy-free (see above), row-chunk pipelined: K2
    runs chunk by chunk into full[rank]; as soon as chunk c's launches are enqueued its rows go out to the peers (RCCL's
    stream waits for them first), so the exchange of chunk c overlaps the computation of chunk c + 1."""
    from .ops import gat_level
    rank, world = _world()
    N = x.shape[0] if x.dim() == 2 else x.shape[1]
    full = torch.empty(world, N, w, dtype=torch.float32, device=x.device)
    works = []

    def on_chunk(c, r0, r1, out):
        works.extend(exchange_blocks(full, r0, r1))

    local = gat_level(x, graph, Ws, As, sk, alpha, True, pipeline=(nchunks, on_chunk, full[rank]))
    for wk in works:
        wk.wait()                                           # the next level reads `full` on the compute stream
    return _GatheredBlocks.apply(local, full)


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
    blocked_in = x.dim() == 3
    N = x.shape[1] if blocked_in else x.shape[0]
    reads_blocks = hip_level and dropout == 0.0 and e > s        # the HIP level reads a column-blocked x in place
    if blocked_in and not reads_blocks:
        x = unblock(x)
    if (concat and hip_level and dropout == 0.0 and not _single() and x.is_cuda and len(set(widths)) == 1 and e > s
            and blocked_width_ok(widths[0])):
        nchunks = PIPELINE_CHUNKS if N >= PIPELINE_MIN_ROWS else 1
        return _blocked_concat_level(x, graph, list(Ws[s:e]), list(As[s:e]), sk, alpha, widths[0], max(1, nchunks))
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
