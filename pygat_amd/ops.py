"""torch.autograd.Function over the C ABI: one GAT level (all local heads) per call.

This is the op boundary that replaces `SpecialSpmmFunction` (reference
layers.py:70-90) and the ATen op sequence of `GraphAttentionLayer.forward` /
`SpGraphAttentionLayer.forward` (layers.py:32-53, 125-173) for ALL heads of a
level at once (the reference loops over heads in Python, models.py:32,34).

    out = gat_level(x, graph, Ws, As, Wskips, alpha, concat)

concat=True  -> [N, H*F'], each head ELU'd         (hidden level, models.py:32)
concat=False -> [N, F'], mean over heads, no ELU   (last level, models.py:34)

PyTorch owns memory and the stream; every FLOP of the path runs in the HIP
library.  There is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import lib, check, padded_width
from .config import config as _config
from .graph import CSRGraph, slot_edges_for


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event spans around the kernel launches (bench.py uses it for the roofline
    figure).  Events are recorded on the stream the kernels are launched on."""

    def __init__(self):
        self.spans = []   # (name, start_event, end_event)

    def times_ms(self):
        out = {}
        for name, a, b in self.spans:
            out.setdefault(name, []).append(a.elapsed_time(b))
        return out


TIMER: Optional[KernelTimer] = None

# Backward flavour.  False: the column pass K4 writes dz per transposed edge and a light row-sum pass (4H-byte
# records fetched through perm_f) takes ds_i = sum_j dz_ij -- no second gather of a Wh row per edge.
# True: the row pass K3b recomputes dz from a gathered Wh_j first, then K4.  None (default) = by row width:
# a record fetch costs one 64-byte sector whatever H is, so below 64 floats per row (1-2 heads of 16: the
# shard of an 8- or 4-GPU head-parallel run) the second gather is the cheaper one (config-5 graph, one
# head: 1.15 ms against 1.46 ms per step; four heads: 2.45 against 2.34; eight: 4.30 against 3.84).
# PYGAT_TWO_GATHER_BACKWARD=1/0 forces either.  Both are free of atomics and bitwise reproducible; they
# differ only in the summation order of ds.
TWO_GATHER_BACKWARD: Optional[bool] = _config.two_gather_backward

# Default since round 2 for rows of up to 256 floats (above: "rowsum"): "rowlocal".  The training forward (K2) also accumulates the share of every row sum that
# went through the alpha branch of the LeakyReLU (aneg, qneg); since sum_j de_ij = 0 the row sums of dz follow
# row-locally in K3a, ds_i = -(1 - alpha)(Gp_i . aneg_i - D_i qneg_i), so the backward needs neither the second
# gather (K3b) nor per-edge dz records and their row-sum pass (K3c): K3a -> K4 -> da -> dW.  Costs one more [N, R]
# table (written by K2, read by K3a).  BACKWARD_FLAVOUR / PYGAT_BACKWARD = rowlocal | rowsum | two-gather forces a
# flavour; a non-None TWO_GATHER_BACKWARD (the older switch) selects between the two older ones.
BACKWARD_FLAVOUR: Optional[str] = _config.backward


def two_gather_backward(row_floats: int) -> bool:
    return TWO_GATHER_BACKWARD if TWO_GATHER_BACKWARD is not None else row_floats < 64


def backward_flavour(row_floats: int) -> str:
    if BACKWARD_FLAVOUR:
        if BACKWARD_FLAVOUR not in ("rowlocal", "rowsum", "two-gather"):
            raise ValueError(f"PYGAT_BACKWARD={BACKWARD_FLAVOUR!r}: expected rowlocal, rowsum or two-gather")
        return BACKWARD_FLAVOUR
    if TWO_GATHER_BACKWARD is not None:
        return "two-gather" if TWO_GATHER_BACKWARD else "rowsum"
    # measured, config-5 graph, 8 heads x F': step with rowlocal / rowsum at 128 floats per row 3.69 / 3.81 ms, at 256
    # 6.90 / 7.23, at 512 14.76 / 14.43, at 1024 29.4 / 28.3: from two 16-byte chunks per lane on the extra accumulators
    # of the training forward cost more than the row-sum pass they replace
    return "rowlocal" if row_floats <= 256 else "rowsum"


# Heads per window of the backward passes = layout of a GR row (include/pygat_amd.h, pygat_head_group): None = the library's
# default for the level's (N, H, F'); a number of FLOATS = windows of at most that many floats on rows wider than it,
# whatever the table size (tests walk tiny graphs in windows of 256 this way; until ABI 12 an environment variable of the
# C library did that).  The choice is fixed per level in its forward and handed to every backward entry point.
BWD_WINDOW_FLOATS: Optional[int] = None

# da inside the column pass (pygat_gat_backward_col with da_part + pygat_a_grad_fold) instead of a pass of its own over Wh,
# ds and dt: on tables of DA_MIN_BYTES and more (the a-gradient stream is HBM time there: 0.12 ms at config 5); a small
# graph's epoch is launch-bound and gains nothing from it (two launches either way).  PYGAT_DA_IN_K4=0 switches it off.
DA_IN_K4 = _config.da_in_k4
DA_MIN_BYTES = 32 << 20


# INTERNAL node order (round 5; CSRGraph.degree_ordered).  A level whose input carries no gradient (a first
# level: its features are the same tensor every epoch) and whose gathered table is RENUMBER_MIN_BYTES and more runs on the
# graph renumbered by descending degree: x is permuted ONCE per feature tensor (features.permuted_rows, cached like the padded
# and the sparse copies), every node table between the kernels is in internal order, K2 writes `out` and K3a reads G / the saved
# output at the caller's rows through the map inside the kernels -- no permutation pass in the step.  Results: the caller-order
# results up to the summation order inside a softmax row.  RENUMBER = False / PYGAT_RENUMBER=0 switches it off.
# Measured on the headline graph (same run, bench.py `alt_node_order`): 8 heads x 16 (512 MB table) 3.16 -> 3.06 ms (K2 0.98 -> 0.95,
# K4 1.24 -> 1.17, K3a 0.37 -> 0.40: its G / y rows are now gathers); 4 heads (256 MB) 1.90 -> 1.84; 2 heads (128 MB) 1.114 -> 1.111;
# 1 head (64 MB) 0.860 -> 0.887 -- narrow rows lose: K3a's gathers cost what K2 / K4 gain.  Hence the threshold.
# With the self-loop-only tail streamed by itself (TAIL below) every width gains on that graph: one head 0.853 -> 0.816 ms, two
# 1.107 -> 1.036, four 1.949 -> 1.751, eight 3.10 -> 2.77 (8 x 8: 1.96 -> 1.77, 8 x 64: 12.76 -> 12.34).  So: tables of
# RENUMBER_MIN_BYTES and more always, tables from RENUMBER_MIN_BYTES_TAIL when the graph has such a tail.
RENUMBER = _config.renumber
RENUMBER_MIN_BYTES = 160 << 20
RENUMBER_MIN_BYTES_TAIL = 48 << 20
# In internal order the self-loop-only nodes (alpha_ii = 1: forward = ELU(Wh_i), backward dWh_i = Gp_i) are a contiguous tail of
# the rows and a suffix of the slots: the fused kernels run on the slot prefix, two plain streams take the tail
# (csrc/k12_tail.hip).  Symmetric patterns, concat levels, the row-local backward in one head window, tails of TAIL_MIN_SHARE of
# the nodes and more.  TAIL = False / PYGAT_TAIL=0 switches it off.
TAIL = _config.tail
TAIL_MIN_SHARE = 0.05


def head_group(N: int, H: int, Fo: int) -> int:
    """Heads per backward window for a level of H heads (see BWD_WINDOW_FLOATS)."""
    if BWD_WINDOW_FLOATS is None:
        return lib.pygat_head_group(N, H, Fo)
    Fp = padded_width(Fo)
    return H if H * Fp <= BWD_WINDOW_FLOATS else max(1, min(H, BWD_WINDOW_FLOATS // Fp))


class _span:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TIMER is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if TIMER is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            TIMER.spans.append((self.name, self.a, b))
        return False


def _ptr(t):
    return None if t is None else t.data_ptr()


# Side stream for the part of the backward that does not feed the weight-gradient GEMM: the `a` gradient (an
# HBM-bound stream over Wh, ds, dt) runs beside the MFMA-bound dW GEMM.  One per device; fork/join with events, so a
# HIP-graph capture of the level records the two branches as parallel graph nodes.
_side = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    key = torch.device(dev).index
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=dev)
    return _side[key]


_K1_SLAB = _config.k1_split_min_k   # measured: Cora epoch 0.614 / 0.596 / 0.595 ms at 256 / 128 / 64
OVERLAP_BACKWARD = _config.overlap_backward   # measured: the fork/join costs more than it hides


def _segments(cols_ptr_ld) -> _lib.OutSegments:
    seg = _lib.OutSegments()
    seg.nseg = len(cols_ptr_ld)
    c = 0
    for k, (ncol, t, ld) in enumerate(cols_ptr_ld):
        seg.col_start[k] = c
        seg.ptr[k] = t.data_ptr()
        seg.ld[k] = ld
        c += ncol
    seg.col_start[len(cols_ptr_ld)] = c
    return seg


def _x3g_takes(transA: bool, transB: bool, M: int, N: int, K: int) -> bool:
    """Shapes the general split-bf16 kernel takes (csrc/k1_gemm_x3.hip try_gemm_x3g; rows are assumed 16-byte aligned
    and, for a k-strided operand, padded to a multiple of 4 columns)."""
    kca, kcb = not transA, transB
    return M >= 64 and N > 64 and K >= 32 and ((K % 4) == 0 or not (kca or kcb))


def _split_k(M: int, N: int, K: int, streamed_k: bool = False, mode: Optional[str] = None, transB: bool = False) -> int:
    """K slabs of a GEMM.  General kernel: up to ~4 work-groups per CU, slabs of >= 256, at most 256 of them,
    and the partial sums (written + re-read) below a quarter of the operand bytes -- the dropout projection
    [N, H*Fin] x [H*Fin, R] streams a 100+ MB operand through few output tiles and wants 5-20 slabs, not the 2
    that "one work-group per CU" gives.  Streamed-K weight gradient (transA, K >= 4096, tools/gemm_tn_sweep.py):
    exactly one work-group per CU is the optimum at every width -- slabs x column tiles = 256 (128 columns:
    256 slabs 0.37 ms; 512 columns: 64 slabs 1.41 ms against 2.68 ms with 256; 1024 columns: 32 slabs)."""
    nt = -(-N // 32)
    if streamed_k and K >= 4096 and M * N <= 512 * 512 and (mode or get_gemm_mode()) == "split-bf16" and M > 64 and N > 64 \
            and M % 4 == 0 and N % 4 == 0:
        # the split-bf16 kernel (128 x 128 tiles through 60 KB of LDS): two work-groups per CU
        tiles = -(-M // 128) * -(-N // 128)
        return max(1, min(512 // tiles, K // 256)) if tiles < 512 else 1
    if streamed_k and K >= 4096 and M * N <= 512 * 512:
        tiles = -(-M // 128) * -(-N // (32 * min(nt, 5 if nt == 5 else 4)))
        return max(1, min(256 // tiles, K // 256)) if tiles < 256 else 1
    if (mode or get_gemm_mode()) == "split-bf16" and _x3g_takes(streamed_k, transB, M, N, K) and not (not streamed_k and K <= 256 and M >= 8192):
        # general split kernel, 128 x 128 tiles, two work-groups per CU.  Its partial sums are expensive (written and
        # re-read by the reduce launch): measured on PPI level 2 (tools/gemm_bench.py --splits), projection 3144 x 2056 x
        # 1024 (425 tiles) 145 / 221 us at 1 / 2 slabs, input gradient 3144 x 1024 x 1024 (200 tiles) 82 / 115 us, weight
        # gradient 1024 x 1024 x 3144 (64 tiles) 217 / 140 / 105 / 89 / 82 us at 1 / 2 / 3 / 4 / 8: slabs only when the
        # tiles alone leave most of the chip idle, then up to 512 work-groups
        tiles = -(-M // 128) * -(-N // 128)
        return 1 if tiles >= 96 else max(1, min(K // 256, 512 // tiles))
    bn = 32 * (nt if nt <= 5 else 4)
    tiles = -(-M // 128) * -(-N // bn)
    by_traffic = int(0.125 * K * (M + N) / (M * N))
    # weight-gradient shapes of the general kernel (K = nodes, few output tiles) want ~2 work-groups per CU: the PPI
    # level-2 dW (1024 x 1024 x 3144) takes 160 us with 4 slabs, 127 us with 8 (tools/diag/gemm_split_sweep.py);
    # projections and input gradients (many tiles) lose with any split: their partial sums are as big as the output
    want = (512 if streamed_k else 256) // tiles
    return max(1, min(256, 1024 // tiles, K // 256, max(by_traffic, want))) if tiles < 1024 else 1


GEMM_MODES = {"split-bf16": 0, "fp32-mfma": 1}     # PYGAT_GEMM_SPLIT_BF16 / PYGAT_GEMM_FP32_MFMA
_mode_tls = threading.local()                        # the calling thread's choice; None = the library's default


def _mode_code(mode: Optional[str] = None) -> int:
    """The `gemm_mode` argument of a C call: an explicit name, else this thread's set_gemm_mode choice, else
    PYGAT_GEMM_DEFAULT (-1: split-bf16 unless PYGAT_GEMM_F32=1 was in the environment when the library was loaded)."""
    mode = mode or getattr(_mode_tls, "mode", None)
    return -1 if mode is None else GEMM_MODES[mode]


def set_gemm_mode(mode: Optional[str]) -> None:
    """'split-bf16': the GEMMs cut every fp32 operand exactly into three bf16 pieces and sum all nine piece products
    in fp32 on the bf16 MFMA pipe; 'fp32-mfma': fp32 MFMA throughout; None: back to the library default
    (include/pygat_amd.h).  The C library holds no mode: this is the CALLING THREAD's choice, handed to every GEMM
    entry point as an argument; a level's backward uses the mode its forward ran with."""
    if mode is not None and mode not in GEMM_MODES:
        raise ValueError(f"gemm mode {mode!r}: expected one of {sorted(GEMM_MODES)}")
    _mode_tls.mode = mode


def get_gemm_mode() -> str:
    code = _mode_code()
    if code < 0:
        code = lib.pygat_default_gemm_mode()
    return next(k for k, v in GEMM_MODES.items() if v == code)


class gemm_mode:
    """with pygat_amd.gemm_mode("fp32-mfma"): ...  -- the product mode of the calls made inside, this thread only."""

    def __init__(self, mode: Optional[str]):
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_mode_tls, "mode", None)
        set_gemm_mode(self.mode)

    def __exit__(self, *exc):
        _mode_tls.mode = self.prev
        return False


def gemm(transA: bool, transB: bool, M: int, N: int, K: int, A: torch.Tensor, lda: int, B: torch.Tensor,
         ldb: int, segments, accumulate: bool = False, split_k: Optional[int] = None, mode: Optional[str] = None,
         a_blocks=None, c_blocks=None) -> None:
    """C = op(A) op(B), fp32 in / fp32 accumulate; `segments` = [(ncols, tensor, ld), ...]; mode: see set_gemm_mode.
    a_blocks / c_blocks: _lib.ColBlocks of a column-blocked A (stored matrix) / C (one segment), include/pygat_amd.h."""
    if c_blocks is not None:
        split_k = 1
    if split_k is None:
        split_k = _split_k(M, N, K, streamed_k=transA and not transB, mode=mode, transB=transB)
    ws = None
    if split_k > 1:
        ws = torch.empty(lib.pygat_gemm_workspace_bytes(M, N, split_k) // 4, dtype=torch.float32, device=A.device)
    seg = _segments(segments)
    if a_blocks is None and c_blocks is None:
        check(lib.pygat_gemm_f32(int(transA), int(transB), M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb,
                                 C.byref(seg), int(accumulate), split_k, _ptr(ws), _mode_code(mode), _stream()), "gemm_f32")
    else:
        check(lib.pygat_gemm_f32_blocked(int(transA), int(transB), M, N, K, A.data_ptr(), lda,
                                         None if a_blocks is None else C.byref(a_blocks), B.data_ptr(), ldb, C.byref(seg),
                                         None if c_blocks is None else C.byref(c_blocks), int(accumulate), split_k, _ptr(ws),
                                         _mode_code(mode), _stream()), "gemm_f32_blocked")


class _Level:
    """Shapes and packed operands of one call."""

    def __init__(self, x, H, Fo, skip):
        # x [N, Fin], or column-blocked [blocks, N, w] (Fin = blocks * w: the head exchange's layout, pygat_amd/dist.py)
        self.blocks = None if x.dim() == 2 else _lib.ColBlocks(int(x.shape[2]), int(x.shape[1]) * int(x.shape[2]))
        self.N, self.Fin = (x.shape[0], x.shape[1]) if x.dim() == 2 else (x.shape[1], x.shape[0] * x.shape[2])
        self.ldx = self.Fin if x.dim() == 2 else int(x.shape[2])
        self.H, self.Fo, self.skip = H, Fo, skip
        self.Fp = padded_width(Fo)
        self.R = H * self.Fp
        self.hg = head_group(self.N, H, Fo)   # heads per backward pass = layout of a GR row
        self.ldw = -(-(self.R * (2 if skip else 1) + 2 * H) // 4) * 4
        self.ts = 0        # slot length of the nnz-split kernels (rows cut by a slot border cost a partial record)

    def xb(self):
        """pygat_col_blocks* of the level's input (None: an ordinary matrix)."""
        return None if self.blocks is None else C.byref(self.blocks)

    def gp_windows(self):
        """(first column in the R-wide tables, width, first column inside a GR row) of each head window's Gp."""
        for h0 in range(0, self.H, self.hg):
            hc = min(self.hg, self.H - h0)
            yield h0 * self.Fp, hc * self.Fp, h0 * (self.Fp + 4)

    def gp_col(self, h):
        """Column of head h's Gp slice inside a GR row (include/pygat_amd.h, K3a)."""
        h0 = (h // self.hg) * self.hg
        return h0 * (self.Fp + 4) + (h - h0) * self.Fp


def _in_features(x) -> int:
    """Input width of a level: x [N, Fin], or column-blocked [blocks, N, w] with Fin = blocks * w."""
    return x.shape[1] if x.dim() == 2 else x.shape[0] * x.shape[2]


def blocked_input_ok(x3: torch.Tensor) -> bool:
    """Can the level read this column-blocked activation [blocks, N, w] in place (include/pygat_amd.h, pygat_col_blocks)?
    w a power of two >= 16; float32 on the GPU."""
    w = int(x3.shape[2])
    return x3.dim() == 3 and x3.is_cuda and x3.dtype == torch.float32 and w >= 16 and (w & (w - 1)) == 0


class GATLevelFn(torch.autograd.Function):
    """forward(x, W[H,Fin,F'], a[H,2F'], Wskip[H,Fin,F']|None, graph, alpha, concat[, bwd_heads]) -> out.

    x may be COLUMN-BLOCKED: a 3-D tensor [blocks, N, w] = the activation of a head-parallel hidden level as the exchange
    left it (block b = rank b's head columns of every node; pygat_amd/dist.py), read in place by the projection and the
    weight-gradient GEMMs; its gradient comes back in the same layout -- the one the reduce-scatter sends.

    bwd_heads = (first, count): the forward covers all H heads, the backward only that range -- dW / da of the
    other heads come back as zeros.  For a rank of a head-parallel run that computes every head's forward
    itself (cheaper than receiving the outputs over xGMI) and owns the gradients of its own heads only.
    Needs Wskip = None and no gradient into x.

    pipeline = (nchunks, on_chunk): the edge-softmax + aggregation pass (K2) runs chunk of rows by chunk of rows
    and `on_chunk(c, row_first, row_end, out)` is called after chunk c's launches are enqueued -- rows
    [row_first, row_end) of `out` are final once they have run -- so that a caller can send them off (RCCL
    all-gather on its own stream, pygat_amd/dist.py) while the next chunk is computed.  concat levels only."""

    @staticmethod
    def forward(ctx, x, W, a, Wskip, graph: CSRGraph, alpha: float, concat: bool, bwd_heads=None, pipeline=None):
        if not x.is_cuda:
            raise RuntimeError("pygat_amd: inputs must be on the GPU; the hot path has no CPU fallback")
        # the path computes in float32 (like the reference's sparse layer, layers.py:150); other float dtypes are cast
        # on the way in and their gradients cast back on the way out
        ctx.in_dtypes = (x.dtype, W.dtype, a.dtype, None if Wskip is None else Wskip.dtype)
        x = x.contiguous().float()
        W = W.contiguous().float(); a = a.contiguous().float()
        H, Fin, Fo = W.shape
        if _in_features(x) != Fin or a.shape != (H, 2 * Fo):
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, W {tuple(W.shape)}, a {tuple(a.shape)}")
        skip = Wskip is not None
        if skip:
            Wskip = Wskip.contiguous().float()

        def pack(Wcat, ldw, a_pad, st):
            check(lib.pygat_pack_params(H, Fin, Fo, W.data_ptr(), a.data_ptr(), _ptr(Wskip), Wcat.data_ptr(), ldw,
                                        a_pad.data_ptr(), st), "pack_params")
        return _level_forward(ctx, tuple(ctx.needs_input_grad[:4]), x, H, Fo, skip, pack, graph, alpha, concat, bwd_heads, pipeline)

    @staticmethod
    def backward(ctx, G):
        dx, dW, da, dWs = _level_backward(ctx, G)
        cast = lambda g_, k: g_ if g_ is None or g_.dtype == ctx.in_dtypes[k] else g_.to(ctx.in_dtypes[k])  # noqa: E731
        return (cast(dx, 0), cast(dW, 1), cast(da, 2), cast(dWs, 3), None, None, None, None, None)


PAD_K = _config.pad_k     # development knob: 0 = run odd input widths as they are
MAX_HEAD_TABLE = 16     # PYGAT_MAX_HEADS_TABLE: heads whose parameter pointers travel as kernel arguments


class GATLevelHeadsFn(torch.autograd.Function):
    """The same level with its parameters given as they live in the model -- one W [Fin,F'], a (2F' values) and
    skip_projection [Fin,F'] tensor PER HEAD (layers.py:21-28,111-119; models.py:15-27) -- instead of stacked: the packing
    kernel reads them through a pointer table (pygat_pack_params_heads), so no torch.stack (a cat launch per parameter
    kind, level and forward: a tenth of a small graph's epoch) precedes the level.
    forward(x, graph, alpha, concat, pipeline, H, skip, *Ws, *As[, *Wskips]) -> out."""

    @staticmethod
    def forward(ctx, x, graph: CSRGraph, alpha: float, concat: bool, pipeline, H: int, skip: bool, xs, *params):
        if not x.is_cuda:
            raise RuntimeError("pygat_amd: inputs must be on the GPU; the hot path has no CPU fallback")
        Ws, As = params[:H], params[H:2 * H]
        Ss = params[2 * H:3 * H] if skip else ()
        Fin, Fo = Ws[0].shape
        ctx.in_dtype_x = x.dtype
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.param_dtypes = [p.dtype for p in params]
        x = x.contiguous().float()
        Ws = [w.contiguous().float() for w in Ws]; As = [v.contiguous().float() for v in As]; Ss = [w.contiguous().float() for w in Ss]
        if _in_features(x) != Fin or any(w.shape != (Fin, Fo) for w in Ws) or any(v.numel() != 2 * Fo for v in As):
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, W {tuple(Ws[0].shape)}, a {tuple(As[0].shape)}")
        PT = C.c_void_p * H
        wp, ap = PT(*[w.data_ptr() for w in Ws]), PT(*[v.data_ptr() for v in As])
        sp = PT(*[w.data_ptr() for w in Ss]) if skip else None

        def pack(Wcat, ldw, a_pad, st):
            check(lib.pygat_pack_params_heads(H, Fin, Fo, wp, ap, sp, Wcat.data_ptr(), ldw, a_pad.data_ptr(), st), "pack_params_heads")
        n = ctx.needs_input_grad
        need = (n[0], any(n[8:8 + H]), any(n[8 + H:8 + 2 * H]), skip and any(n[8 + 2 * H:8 + 3 * H]))
        ctx.H, ctx.skip = H, skip
        return _level_forward(ctx, need, x, H, Fo, skip, pack, graph, alpha, concat, None, pipeline, xs=xs)

    @staticmethod
    def backward(ctx, G):
        dx, dW, da, dWs = _level_backward(ctx, G)
        H = ctx.H
        if dx is not None and dx.dtype != ctx.in_dtype_x:
            dx = dx.to(ctx.in_dtype_x)
        outs = []
        for k in range(H):
            outs.append(None if dW is None else dW[k])
        for k in range(H):
            outs.append(None if da is None else da[k].reshape(ctx.param_shapes[H + k]))
        if ctx.skip:
            for k in range(H):
                outs.append(None if dWs is None else dWs[k])
        outs = [g_ if g_ is None or g_.dtype == dt else g_.to(dt) for g_, dt in zip(outs, ctx.param_dtypes)]
        return (dx, None, None, None, None, None, None, None) + tuple(outs)


def _level_forward(ctx, need, x, H, Fo, skip, pack, graph: CSRGraph, alpha: float, concat: bool, bwd_heads, pipeline, xs=None):
    """Body of the level's forward, shared by GATLevelFn (stacked parameters) and GATLevelHeadsFn (one tensor per head).
    need = (x, W, a, Wskip) gradient flags; pack(Wcat, ldw, a_pad, stream) launches the parameter packing; xs: the
    SparseFeatures of x (features.py) -- projection and weight gradient on the non-zeros only -- or None."""
    if x.dim() == 3 and not blocked_input_ok(x):
        raise ValueError(f"column-blocked input {tuple(x.shape)}: blocks must be a power of two >= 16 floats wide (float32, GPU)")
    L = _Level(x, H, Fo, skip)
    Fin = L.Fin
    if L.N != graph.n:
        raise ValueError(f"x has {L.N} rows but the graph has {graph.n} nodes")
    if L.blocks is not None:
        xs = None
    user_row = None
    tail = None      # (first tail row, pygat_graph* of the slot prefix) when the self-loop-only tail goes through its own streams
    single_out = (not concat) and H == 1
    if (RENUMBER and not need[0] and xs is None and pipeline is None and x.dim() == 2 and (concat or single_out)
            and graph.user_row is None and not graph.degree_sorted and L.N * L.R * 4 >= min(RENUMBER_MIN_BYTES, RENUMBER_MIN_BYTES_TAIL)
            and not torch.cuda.is_current_stream_capturing()):     # (a captured graph would bake this epoch's permuted copy of x in)
        from .features import permuted_rows
        g_int, to_user, _ = graph.degree_ordered()
        worth = L.N * L.R * 4 >= RENUMBER_MIN_BYTES
        if not worth and TAIL and concat and graph.symmetric:     # a narrower table: only with a self-loop-only tail to stream
            t = g_int.fwd.self_loop_tail(slot_edges_for(L.R, g_int.slot_edges))
            worth = t is not None and L.N - t[0] >= TAIL_MIN_SHARE * L.N
        xp = permuted_rows(x, to_user) if worth else None
        if xp is not None:
            x, graph, user_row = xp, g_int, to_user
    L.ts = slot_edges_for(L.R, graph.slot_edges)
    L.mode = get_gemm_mode()     # this thread's product mode, fixed for the level: its backward (another thread) uses it too
    dev, f32 = x.device, torch.float32
    need_grad = any(need)
    ctx.need = need
    with torch.cuda.device(dev):
        st = _stream()
        Wcat = torch.empty(Fin, L.ldw, dtype=f32, device=dev)
        a_pad = torch.empty(H, 2, L.Fp, dtype=f32, device=dev)
        pack(Wcat, L.ldw, a_pad, st)
        # K1: [Wh | Sk | s | t] = x @ Wcat
        Wh = torch.empty(L.N, L.R, dtype=f32, device=dev)
        Sk = torch.empty(L.N, L.R, dtype=f32, device=dev) if skip else None
        s = torch.empty(L.N, H, dtype=f32, device=dev)
        ncols = L.R * (2 if skip else 1) + H
        tiles = -(-L.N // 128) * -(-ncols // 128)
        # few row tiles (Cora: 22): K slabs of >= 128 until ~1.5 work-groups per CU (PYGAT_K1_SPLIT_MIN_K: slab floor)
        split_k = max(1, min(-(-384 // tiles), Fin // _K1_SLAB)) if tiles < 256 else 1
        ws = torch.empty(lib.pygat_gemm_workspace_bytes(L.N, ncols, split_k) // 4, dtype=f32, device=dev) \
            if split_k > 1 else None
        if xs is not None and need[0]:
            xs = None                                   # a gradient into x: the dense path forms it
        with _span("k1_project"):
            if xs is not None:
                check(lib.pygat_project_sparse(L.N, Fin, H, Fo, xs.rowptr.data_ptr(), xs.col.data_ptr(), xs.val.data_ptr(),
                                               Wcat.data_ptr(), L.ldw, 0.0, None, 0, None, Wh.data_ptr(), _ptr(Sk), s.data_ptr(),
                                               st), "project_sparse")
            else:
                check(lib.pygat_project_blocked(L.N, Fin, H, Fo, x.data_ptr(), L.ldx, L.xb(), Wcat.data_ptr(), L.ldw, a_pad.data_ptr(),
                                                Wh.data_ptr(), _ptr(Sk), s.data_ptr(), split_k, _ptr(ws), GEMM_MODES[L.mode], st),
                      "project")
        # K2
        flags = (_lib.F_ELU if concat else 0) | (_lib.F_SKIP if skip else 0)
        # mean over ONE head = that head (Cora / Citeseer output level, train.py:55,66): K2 writes `out` itself (its epilogue
        # adds the skip rows, no ELU), no head-mean launch; hattn is then only kept for the backward
        single = (not concat) and H == 1
        hattn = torch.empty(L.N, L.R, dtype=f32, device=dev) if (not concat and (need_grad or not single)) else None
        m = torch.empty(L.N, H, dtype=f32, device=dev) if need_grad else None
        Z = torch.empty(L.N, H, dtype=f32, device=dev) if need_grad else None
        flavour = backward_flavour(L.R) if need_grad else None
        aneg = torch.empty(L.N, L.R, dtype=f32, device=dev) if flavour == "rowlocal" else None
        qneg = torch.empty(L.N, H, dtype=f32, device=dev) if flavour == "rowlocal" else None
        out = pipeline[2] if (pipeline is not None and len(pipeline) > 2 and concat) else None    # the caller's buffer (dist.py:
        if out is not None:                                                                           # the rank's block of the exchange)
            if tuple(out.shape) != (L.N, H * Fo) or out.dtype != f32 or not out.is_contiguous() or out.device != dev:
                raise ValueError(f"pipeline output buffer {tuple(out.shape)}: expected a contiguous float32 [{L.N}, {H * Fo}] on {dev}")
            # the buffer is the rank's block of a larger tensor whose OTHER blocks are received in place afterwards: alias its
            # memory with a tensor of its own (no view relation, own version counter), or autograd takes the peers' writes for
            # in-place changes of this output
            out = torch.empty(0, dtype=f32, device=dev).set_(out.untyped_storage(), out.storage_offset(), out.shape, out.stride())
        else:
            out = torch.empty(L.N, H * Fo if concat else Fo, dtype=f32, device=dev)
        part = torch.empty(lib.pygat_partials_bytes(graph.nnz, L.ts, H, L.Fp) // 4, dtype=f32,
                           device=dev)
        # (a degree-ordered pattern: reached through the renumbering above -- user_row maps `out` / G to the caller's rows -- or
        # handed in by a model that keeps ALL its node arrays in internal order, graph.InternalOrderView: no map)
        if (TAIL and graph.degree_sorted and concat and graph.symmetric
                and (not need_grad or (flavour == "rowlocal" and L.hg >= H and bwd_heads is None))):
            t = graph.fwd.self_loop_tail(L.ts)
            if t is not None and L.N - t[0] >= TAIL_MIN_SHARE * L.N:
                tail = (t[0], t[2], t[1])
        chunks = [(graph.fwd.ref(L.ts) if tail is None else tail[1], 0, L.N if tail is None else tail[0])]
        if pipeline is not None and concat and pipeline[0] > 1:
            chunks = (graph.fwd.row_chunks(int(pipeline[0]), L.ts) if tail is None
                      else graph.fwd.row_chunks(int(pipeline[0]), L.ts, nslots=tail[2], row_end=tail[0]))
        # a pipelined level: chunk c's fix-up launch (a few thousand cut rows, latency-bound: 25-30 us at config 5) and the
        # caller's hand-off of the chunk run on a SIDE stream beside chunk c + 1's main launch (the partial records are per
        # slot: chunks share none) -- in line they cost the step ~70 us per chunk border
        phases = len(chunks) > 1 and TIMER is None and bool(lib.pygat_gat_forward_phases_ok(L.N, H, Fo))
        if phases:
            main_s, side_s = torch.cuda.current_stream(dev), _side_stream(dev)

        def k2(gref, fl, stream):
            check(lib.pygat_gat_forward(gref, H, Fo, float(alpha), fl, Wh.data_ptr(), s.data_ptr(),
                                        a_pad.data_ptr(), _ptr(Sk), None, out.data_ptr() if (concat or single) else None,
                                        _ptr(hattn), _ptr(m), _ptr(Z), _ptr(aneg), _ptr(qneg), part.data_ptr(), stream),
                  "gat_forward")
        for c, (gref, r0, r1) in enumerate(chunks):
            if phases:
                k2(gref, flags | _lib.F_MAIN_ONLY, st)
                side_s.wait_stream(main_s)
                with torch.cuda.stream(side_s):
                    k2(gref, flags | _lib.F_FIXUP_ONLY, side_s.cuda_stream)
                    pipeline[1](c, r0, r1, out)        # (a collective issued here waits for the side stream = for this chunk only)
                continue
            with _span("k2_forward"):
                k2(gref, flags, st)
                if tail is not None and c == len(chunks) - 1:     # the self-loop-only rows: out = ELU(Wh (+ skip)), one stream
                    check(lib.pygat_gat_forward_tail(tail[0], L.N - tail[0], H, Fo, flags, Wh.data_ptr(), 0, _ptr(Sk), out.data_ptr(),
                                                     _ptr(user_row), _ptr(m), _ptr(Z), _ptr(qneg), st), "gat_forward_tail")
            if pipeline is not None and concat:
                pipeline[1](c, r0, r1 if not (tail is not None and c == len(chunks) - 1) else L.N, out)
        if phases:
            main_s.wait_stream(side_s)
            if tail is not None:       # (pipelined chunks: the tail is one more hand-off, after the last chunk's)
                check(lib.pygat_gat_forward_tail(tail[0], L.N - tail[0], H, Fo, flags, Wh.data_ptr(), 0, _ptr(Sk), out.data_ptr(),
                                                 _ptr(user_row), _ptr(m), _ptr(Z), _ptr(qneg), st), "gat_forward_tail")
                pipeline[1](len(chunks), tail[0], L.N, out)
        if not concat and not single:
            check(lib.pygat_head_mean(L.N, H, Fo, hattn.data_ptr(), _ptr(Sk), out.data_ptr(), st), "head_mean")
    if need_grad:
        # concat: the backward recovers hattn from `out` (no second [N,R] table is written)
        ctx.save_for_backward(x, Wcat, a_pad, Wh, s, Sk, out if concat else hattn, m, Z, aneg, qneg)
        ctx.graph, ctx.L, ctx.alpha, ctx.concat, ctx.flags = graph, L, float(alpha), concat, flags
        ctx.user_row = user_row
        ctx.tail = tail
        ctx.xs = xs
        ctx.flavour = flavour
        ctx.bwd_heads = None
        if bwd_heads is not None:
            hb, hr = int(bwd_heads[0]), int(bwd_heads[1])
            if not (0 <= hb and 0 < hr and hb + hr <= H):
                raise ValueError(f"bwd_heads {bwd_heads} outside the {H} heads of the level")
            if skip or need[0]:
                raise ValueError("pygat_amd: bwd_heads supports neither a skip projection nor a gradient into x")
            ctx.bwd_heads = (hb, hr)
    return out

def _level_backward(ctx, G):
    """Body of the level's backward: -> (dx | None, dW [H,Fin,F'] | None, da [H,2F'] | None, dWskip | None)."""
    x, Wcat, a_pad, Wh, s, Sk, y, m, Z, aneg, qneg = ctx.saved_tensors
    graph, L, H, Fo = ctx.graph, ctx.L, ctx.L.H, ctx.L.Fo
    dev, f32 = x.device, torch.float32
    G = G.contiguous().float()
    ranged = ctx.bwd_heads is not None
    hb, hr = ctx.bwd_heads if ranged else (0, 0)       # (0, 0) = all heads in the C ABI
    Hb = hr if ranged else H                            # heads this backward covers
    with torch.cuda.device(dev):
        st = _stream()
        RW = Hb * (L.Fp + 4)                            # GR is compact for the covered heads
        GR = torch.empty(L.N, RW, dtype=f32, device=dev)      # per head window: [Gp | (s, m, 1/Z, D) per head]
        ds = torch.empty(L.N, H, dtype=f32, device=dev)
        dt = torch.empty(L.N, H, dtype=f32, device=dev)
        dWh = torch.empty(L.N, L.R, dtype=f32, device=dev)
        part = torch.empty(lib.pygat_partials_bytes(graph.nnz, L.ts, H, L.Fp) // 4, dtype=f32,
                           device=dev)
        rowlocal = ctx.flavour == "rowlocal"
        # heads per window: the level's choice for all its heads, or the library default for the range a ranged backward covers
        hgw = lib.pygat_head_group(L.N, Hb, Fo) if (ranged and BWD_WINDOW_FLOATS is None) else (min(L.hg, Hb) if not ranged else head_group(L.N, Hb, Fo))
        # (symmetric pattern with a self-loop-only tail in internal order: the column pass and its fold run on the slot prefix)
        tail = getattr(ctx, "tail", None)
        gT = graph.bwd.ref(L.ts) if tail is None else tail[1]
        # da along with the column pass (no separate stream over Wh, ds, dt), when the pass can and the table is large
        da_part = None
        if DA_IN_K4 and not ranged and ctx.flavour != "rowsum" and ctx.need[2] and L.N * L.R * 4 >= DA_MIN_BYTES:
            nb = lib.pygat_gat_backward_col_da_bytes(gT, H, Fo, hgw)
            if nb:
                da_part = torch.empty(nb // 4, dtype=f32, device=dev)
        # no skip projection: nothing but the column pass reads the tail's Gp -- its whole backward is one stream from G / out
        # (pygat_gat_backward_tail) and K3a runs on the rows before it
        fused_tail = tail is not None and not L.skip
        with _span("k3a_prepare"):
            check(lib.pygat_gat_backward_prepare(tail[0] if fused_tail else L.N, H, Fo, ctx.flags, 0 if ctx.concat else 1, G.data_ptr(),
                                                 y.data_ptr(), _ptr(Sk), s.data_ptr(), m.data_ptr(), Z.data_ptr(),
                                                 GR.data_ptr(), _ptr(aneg), _ptr(qneg), ctx.alpha,
                                                 ds.data_ptr() if rowlocal else None, hb, hr, hgw, _ptr(getattr(ctx, "user_row", None)), st),
                      "gat_backward_prepare")
        two_gather = ctx.flavour == "two-gather"
        if rowlocal:        # ds is known: the column pass finishes dWh on its own
            with _span("k4_backward_col"):
                check(lib.pygat_gat_backward_col(gT, None, H, Fo, ctx.alpha, Wh.data_ptr(),
                                                 a_pad.data_ptr(), GR.data_ptr(), None, ds.data_ptr(),
                                                 dWh.data_ptr(), dt.data_ptr(), None, part.data_ptr(), _ptr(da_part), hb, hr, hgw, st),
                      "gat_backward_col")
                if fused_tail:           # the self-loop-only rows: dWh_j = G_u ELU'(out_u), ds_j = dt_j = 0 (csrc/k12_tail.hip)
                    check(lib.pygat_gat_backward_tail(tail[0], L.N - tail[0], H, Fo, ctx.flags, G.data_ptr(), y.data_ptr(),
                                                      _ptr(getattr(ctx, "user_row", None)), dWh.data_ptr(), 0, 0, ds.data_ptr(), dt.data_ptr(), st),
                          "gat_backward_tail")
                elif tail is not None:   # ... with a skip projection (its weight gradient reads every row's Gp): dWh_j = Gp_j, dt_j = 0
                    check(lib.pygat_gat_backward_col_tail(tail[0], L.N - tail[0], H, Fo, GR.data_ptr(), dWh.data_ptr(), dt.data_ptr(), st),
                          "gat_backward_col_tail")
        elif two_gather:
            with _span("k3b_row"):
                check(lib.pygat_gat_backward_row(graph.fwd.ref(L.ts), H, Fo, ctx.alpha, Wh.data_ptr(),
                                                 a_pad.data_ptr(), GR.data_ptr(), None, ds.data_ptr(),
                                                 part.data_ptr(), hb, hr, hgw, st), "gat_backward_row")
            with _span("k4_backward_col"):
                check(lib.pygat_gat_backward_col(graph.bwd.ref(L.ts), None, H, Fo, ctx.alpha, Wh.data_ptr(),
                                                 a_pad.data_ptr(), GR.data_ptr(), None, ds.data_ptr(),
                                                 dWh.data_ptr(), dt.data_ptr(), None, part.data_ptr(), _ptr(da_part), hb, hr, hgw, st),
                      "gat_backward_col")
        else:
            dz_t = torch.empty(graph.nnz, H, dtype=f32, device=dev)
            with _span("k4_backward_col"):
                check(lib.pygat_gat_backward_col(graph.bwd.ref(L.ts), None, H, Fo, ctx.alpha, Wh.data_ptr(),
                                                 a_pad.data_ptr(), GR.data_ptr(), None, None,
                                                 dWh.data_ptr(), dt.data_ptr(), dz_t.data_ptr(), part.data_ptr(), None, hb, hr, hgw, st),
                      "gat_backward_col")
            with _span("k3c_rowsum"):
                check(lib.pygat_gat_backward_rowsum(graph.fwd.ref(L.ts), graph.perm_f.data_ptr(), H, Fo,
                                                    dz_t.data_ptr(), ds.data_ptr(), part.data_ptr(), hb, hr, st),
                      "gat_backward_rowsum")
        # da; after the row-sum flavour the same stream also finishes dWh_i += ds_i a_src
        da = (torch.zeros if ranged else torch.empty)(H, 2 * Fo, dtype=f32, device=dev)
        ws = torch.empty(lib.pygat_agrad_workspace_bytes(H, Fo) // 4, dtype=f32, device=dev)
        # ... unless nothing but the weight-gradient GEMM consumes dWh: there ds rides along as extra columns
        # (pygat_wgrad) and dWh is never rewritten
        rowsum = ctx.flavour == "rowsum"
        xs = getattr(ctx, "xs", None)
        sparse_w = xs is not None and not ranged and (not L.skip or L.hg >= H)    # weight gradients on the non-zeros of x
        fold_ds = (rowsum and ctx.need[1] and not ctx.need[0] and not sparse_w
                   and (Hb * L.Fp) % 32 == 0 and L.N >= 4096)   # the streamed-K GEMM takes [dWh | ds] in one pass
        finish = rowsum and not fold_ds
        # when a_grad does not rewrite dWh, nothing downstream depends on it: run it on the side stream,
        # beside the weight-gradient GEMM (TIMER spans stay on the main stream: no fork while timing kernels)
        fork = OVERLAP_BACKWARD and not finish and TIMER is None and ctx.need[1]
        if fork:
            main, side = torch.cuda.current_stream(), _side_stream(dev)
            side.wait_stream(main)      # the tensors it touches stay referenced until the join below
        if da_part is not None:      # the column pass left one record per work-group: fold them (and the cut rows) in a fixed order
            with _span("k5_afold"):
                check(lib.pygat_a_grad_fold(gT, H, Fo, Wh.data_ptr(), ds.data_ptr(), dt.data_ptr(),
                                            da_part.data_ptr(), da.data_ptr(), ws.data_ptr(), hgw,
                                            side.cuda_stream if fork else st), "a_grad_fold")
        else:
            with _span("k5_agrad"):
                check(lib.pygat_a_grad(L.N, H, Fo, Wh.data_ptr(), ds.data_ptr(), dt.data_ptr(), da.data_ptr(),
                                       ws.data_ptr(), a_pad.data_ptr() if finish else None,
                                       dWh.data_ptr() if finish else None, None, hb, hr,
                                       side.cuda_stream if fork else st), "a_grad")
        # dW = x^T dWh (split-K over the nodes), dWskip = x^T Gp
        dW = dWs = dx = None
        if sparse_w and (ctx.need[1] or (L.skip and ctx.need[3])):
            dW = torch.empty(H, L.Fin, Fo, dtype=f32, device=dev)
            want_s = L.skip and ctx.need[3]
            dWs = torch.empty(H, L.Fin, Fo, dtype=f32, device=dev) if want_s else None
            wss = torch.empty(lib.pygat_wgrad_sparse_workspace_bytes(xs.nseg, H, Fo, int(want_s)) // 4 + 4, dtype=f32, device=dev)
            with _span("k5_wgrad"):
                check(lib.pygat_wgrad_sparse(L.N, L.Fin, H, Fo, xs.nseg, xs.colseg.data_ptr(), xs.seg_col.data_ptr(),
                                             xs.seg_begin.data_ptr(), xs.seg_end.data_ptr(), xs.trow.data_ptr(), xs.tval.data_ptr(),
                                             0.0, None, 0, None, dWh.data_ptr(), GR.data_ptr() if want_s else None, RW,
                                             wss.data_ptr(), dW.data_ptr(), _ptr(dWs), st), "wgrad_sparse")
        elif ctx.need[1]:
            split_k = _split_k(L.Fin, Hb * L.Fp + (Hb if fold_ds else 0), L.N, streamed_k=True, mode=L.mode)
            wsw = torch.empty(lib.pygat_wgrad_workspace_bytes(L.Fin, H, Fo, split_k) // 4, dtype=f32, device=dev)
            dW = (torch.zeros if ranged else torch.empty)(H, L.Fin, Fo, dtype=f32, device=dev)
            with _span("k5_wgrad"):
                check(lib.pygat_wgrad_blocked(L.N, L.Fin, H, Fo, x.data_ptr(), L.ldx, L.xb(), dWh.data_ptr(),
                                              ds.data_ptr() if fold_ds else None, a_pad.data_ptr(), dW.data_ptr(), split_k,
                                              wsw.data_ptr(), hb, hr, GEMM_MODES[L.mode], st), "wgrad")
        if L.skip and ctx.need[3] and not sparse_w:
            dSc = torch.empty(L.Fin, L.R, dtype=f32, device=dev)
            for c0, w, g0 in L.gp_windows():
                gemm(True, False, L.Fin, w, L.N, x, L.ldx, GR[:, g0:], RW, [(w, dSc[:, c0:], L.R)], mode=L.mode, a_blocks=L.blocks)
            dWs = torch.empty(H, L.Fin, Fo, dtype=f32, device=dev)
            check(lib.pygat_unpack_wgrad(H, L.Fin, Fo, dSc.data_ptr(), L.R, 0, dWs.data_ptr(), st), "unpack")
        # dx = dWh Wcat[:, :R]^T (+ Gp Wcat[:, R:2R]^T)
        if ctx.need[0]:
            # (a column-blocked x: its gradient is written in the same blocks -- what the reduce-scatter of the exchange sends)
            dx = torch.empty(x.shape, dtype=f32, device=dev)
            with _span("k5_xgrad"):
                gemm(False, True, L.N, L.Fin, L.R, dWh, L.R, Wcat, L.ldw, [(L.Fin, dx, L.ldx)], mode=L.mode, c_blocks=L.blocks)
                if L.skip:
                    for c0, w, g0 in L.gp_windows():
                        gemm(False, True, L.N, L.Fin, w, GR[:, g0:], RW, Wcat[:, L.R + c0:], L.ldw,
                             [(L.Fin, dx, L.ldx)], accumulate=True, split_k=1, mode=L.mode, c_blocks=L.blocks)
        if fork:
            main.wait_stream(side)
    return dx, dW, (da if ctx.need[2] else None), dWs


class StackHeads(torch.autograd.Function):
    """(W [H,Fin,F'], a [H,2F'], Wskip [H,Fin,F'] | None) from the per-head parameter tensors in ONE launch (torch.stack: a
    cat launch per parameter kind); backward: views of the stacked gradients.  forward(H, skip, rows, *Ws, *As[, *Wskips]);
    rows > Fin: W and Wskip come out as [H, rows, F'] with zero rows appended (gat_level on padded input columns)."""

    @staticmethod
    def forward(ctx, H: int, skip: bool, rows: int, *params):
        Ws, As = params[:H], params[H:2 * H]
        Ss = params[2 * H:3 * H] if skip else ()
        dev, f32 = Ws[0].device, torch.float32
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.H, ctx.skip = H, skip
        ok = Ws[0].is_cuda and H <= MAX_HEAD_TABLE and all(p.dtype == f32 and p.is_contiguous() for p in params)
        pad = rows > 0 and rows != Ws[0].shape[0]
        if not ok:
            grow = (lambda t: torch.nn.functional.pad(t, (0, 0, 0, rows - t.shape[0]))) if pad else (lambda t: t)   # noqa: E731
            return (torch.stack([grow(w) for w in Ws], 0), torch.stack([q.reshape(-1) for q in As], 0),
                    torch.stack([grow(w) for w in Ss], 0) if skip else None)
        wshape = (rows, *Ws[0].shape[1:]) if pad else tuple(Ws[0].shape)
        W = torch.empty(H, *wshape, dtype=f32, device=dev)
        a = torch.empty(H, As[0].numel(), dtype=f32, device=dev)
        S = None
        if skip:                                                                        # (GATv2: W is [2 Fin, F'], the skip [Fin, F'])
            S = torch.empty(H, *((rows, *Ss[0].shape[1:]) if pad else Ss[0].shape), dtype=f32, device=dev)
        PT = C.c_void_p * H
        with torch.cuda.device(dev):
            check(lib.pygat_stack_heads_padded(H, Ws[0].numel(), W[0].numel(), As[0].numel(), Ss[0].numel() if skip else 0,
                                               S[0].numel() if skip else 0, PT(*[w.data_ptr() for w in Ws]),
                                               PT(*[v.data_ptr() for v in As]), PT(*[w.data_ptr() for w in Ss]) if skip else None,
                                               W.data_ptr(), a.data_ptr(), _ptr(S), _stream()), "stack_heads")
        return W, a, S

    @staticmethod
    def backward(ctx, dW, da, dS):
        H = ctx.H
        rows_w = ctx.shapes[0][0]
        outs = [None if dW is None else dW[k][:rows_w] for k in range(H)]               # (a view: the first Fin rows are contiguous)
        outs += [None if da is None else da[k].reshape(ctx.shapes[H + k]) for k in range(H)]
        if ctx.skip:
            rows_s = ctx.shapes[2 * H][0]
            outs += [None if dS is None else dS[k][:rows_s] for k in range(H)]
        return (None, None, None) + tuple(outs)


def stack_heads(Ws, As, Wskips):
    """-> (W, a, Wskip) stacked; one launch."""
    return StackHeads.apply(len(Ws), Wskips is not None, 0, *Ws, *As, *(Wskips if Wskips is not None else ()))


def gat_level(x: torch.Tensor, graph: CSRGraph, Ws: Sequence[torch.Tensor], As: Sequence[torch.Tensor],
              Wskips: Optional[Sequence[torch.Tensor]], alpha: float, concat: bool, pipeline=None, xs=None) -> torch.Tensor:
    """All heads of one level. Ws: H tensors [Fin,F']; As: H tensors with 2F' elements
    ([2F',1] as in GraphAttentionLayer, layers.py:23, or [1,2F'] as in SpGraphAttentionLayer,
    layers.py:114); Wskips: H tensors [Fin,F'] or None.  pipeline: see GATLevelFn.  xs: features.SparseFeatures of x
    (a first level on sparse input features) or None."""
    H = len(Ws)
    Fin = x.shape[1]
    if (xs is None and PAD_K and Fin % 16 and 16 < Fin <= 1024 and x.is_cuda and not x.requires_grad and x.dim() == 2
            and H <= MAX_HEAD_TABLE and Ws[0].dim() == 2 and Ws[0].is_cuda):
        # input columns zero-padded to a multiple of 16 (cached with x, like the sparse pattern) and zero rows appended to the
        # weights while they are stacked: the split-bf16 kernels take K in steps of 16; on PPI's 50 input features the fp32
        # fallback ran the projection at 45 us and the two weight gradients at 60-67 us, against 12 / 10 us on 64
        from .features import padded_columns
        xp = padded_columns(x, 16)
        W, a, Wskip = StackHeads.apply(H, Wskips is not None, xp.shape[1], *Ws, *As, *(Wskips if Wskips is not None else ()))
        return GATLevelFn.apply(xp, W, a, Wskip, graph, alpha, concat, None, pipeline)
    if H <= MAX_HEAD_TABLE:        # parameters read in place through a pointer table: no torch.stack launches
        return GATLevelHeadsFn.apply(x, graph, alpha, concat, pipeline, H, Wskips is not None, xs, *Ws, *As,
                                     *(Wskips if Wskips is not None else ()))
    W = torch.stack(list(Ws), 0)
    a = torch.stack([p.reshape(-1) for p in As], 0)
    Wskip = torch.stack(list(Wskips), 0) if Wskips is not None else None
    return GATLevelFn.apply(x, W, a, Wskip, graph, alpha, concat, None, pipeline)
