"""GATv2 layers of the reference (layers.py:179-316), on the same HIP machinery.

`SpGraphAttentionLayerV2` (layers.py:234-316): e_ij = a . LeakyReLU(W_l h_i + W_r h_j), softmax over
the row, aggregation of the LEFT projection at the neighbour (`special_spmm(..., Whi)`, layers.py:296).
One projection GEMM yields [Whi | Whj] rows; K2 (V2 variant) gathers one such row per edge; the
backward recomputes everything per edge in a row pass and a column pass (csrc/k6_gatv2_backward.hip).

`GraphAttentionLayerV2` (layers.py:179-232): as written, the logit is ONE value per node broadcast
along its row (layers.py:214-217), so attention is uniform and the layer is the neighbour mean of
`h W[Fin:]`; `a` and `W[:Fin]` receive exactly-zero gradients.  Reproduced as is.
"""
from __future__ import annotations

import ctypes as C  # noqa: F401
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import lib, check
from .graph import CSRGraph, as_graph
from .ops import _Level, _ptr, _span, _stream, gemm, gat_level, gemm_mode, get_gemm_mode, stack_heads


class GATv2LevelFn(torch.autograd.Function):
    """forward(x, W[H,2Fin,F'], a[H,F'], Wskip[H,Fin,F']|None, graph, alpha, concat, masks|None).

    masks (train-mode dropout, layers.py:266,271-272,293): dict of pre-scaled keep masks
    {"x": [H,N,Fin], "whi": [H,N,F'], "whj": [H,N,F'], "att": [E,H]}; each head draws its own input mask
    (models.py:32), so the projection then runs per head on the masked input."""

    @staticmethod
    def forward(ctx, x, W, a, Wskip, graph: CSRGraph, alpha: float, concat: bool, masks=None):
        if not x.is_cuda:
            raise RuntimeError("pygat_amd: inputs must be on the GPU; the hot path has no CPU fallback")
        x = x.contiguous().float(); W = W.contiguous().float(); a = a.contiguous().float()
        H, Fin2, Fo = W.shape
        Fin = Fin2 // 2
        if x.shape[1] != Fin or a.shape != (H, Fo):
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, W {tuple(W.shape)}, a {tuple(a.shape)}")
        skip = Wskip is not None
        L = _Level(x, H, Fo, skip)
        # Internal node order + self-loop-only tail, as for the v1 level (ops._level_forward): no masks, no
        # gradient into x, concat.  A node whose only edge is its self loop has alpha_ii = 1: h'_i = ELU(Whi_i (+ skip_i))
        # (layers.py:296 with one edge), dWhi_i = Gp_i, dWhj_i = 0, no share in da.
        user_row = tail = None
        if (ops.RENUMBER and masks is None and not ctx.needs_input_grad[0] and concat and graph.user_row is None
                and not graph.degree_sorted and L.N * 2 * L.R * 4 >= min(ops.RENUMBER_MIN_BYTES, ops.RENUMBER_MIN_BYTES_TAIL)
                and not torch.cuda.is_current_stream_capturing()):
            from .features import permuted_rows
            g_int, to_user, _ = graph.degree_ordered()
            worth = L.N * 2 * L.R * 4 >= ops.RENUMBER_MIN_BYTES
            if not worth and ops.TAIL and graph.symmetric:
                t = g_int.fwd.self_loop_tail(graph.slot_edges)
                worth = t is not None and L.N - t[0] >= ops.TAIL_MIN_SHARE * L.N
            xp = permuted_rows(x, to_user) if worth else None
            if xp is not None:
                x, graph, user_row = xp, g_int, to_user
        L.ts = graph.slot_edges
        if ops.TAIL and masks is None and graph.degree_sorted and concat and graph.symmetric:
            t = graph.fwd.self_loop_tail(L.ts)
            if t is not None and L.N - t[0] >= ops.TAIL_MIN_SHARE * L.N:
                tail = (t[0], t[2])
        if 2 * L.R > 2048:
            raise ValueError("pygat_amd: GATv2 row too wide; shard the heads")
        dev, f32 = x.device, torch.float32
        R, Fp = L.R, L.Fp
        # operand of the projection: columns [Wi heads | Wj heads | skip heads], heads padded to Fp
        ncols = 2 * R + (R if skip else 0)
        Wcat = torch.zeros(Fin, ncols, dtype=f32, device=dev)
        Wv = Wcat.view(Fin, ncols // Fp, Fp)
        Wv[:, 0:H, :Fo] = W[:, :Fin, :].permute(1, 0, 2)
        Wv[:, H:2 * H, :Fo] = W[:, Fin:, :].permute(1, 0, 2)
        if skip:
            Wv[:, 2 * H:3 * H, :Fo] = Wskip.contiguous().float().permute(1, 0, 2)
        a2 = torch.zeros(H, Fp, dtype=f32, device=dev)
        a2[:, :Fo] = a
        need_grad = any(ctx.needs_input_grad[:4])
        ctx.gemm_mode = get_gemm_mode()     # the backward (an autograd thread) forms its GEMM products the same way
        with torch.cuda.device(dev):
            st = _stream()
            WW = torch.empty(L.N, 2 * R, dtype=f32, device=dev)
            Sk = torch.empty(L.N, R, dtype=f32, device=dev) if skip else None
            mask_x = mww = matt = None
            if masks is None:
                segs = [(2 * R, WW, 2 * R)] + ([(R, Sk, R)] if skip else [])
                with _span("v2_project"):
                    gemm(False, False, L.N, ncols, Fin, x, Fin, Wcat, ncols, segs)
            else:
                mask_x = masks["x"].to(f32).contiguous()
                matt = masks["att"].to(f32).contiguous()
                mww = torch.zeros(L.N, 2 * H, Fp, dtype=f32, device=dev)      # [Whi | Whj] mask, padded layout
                mww[:, :H, :Fo] = masks["whi"].to(f32).permute(1, 0, 2)
                mww[:, H:, :Fo] = masks["whj"].to(f32).permute(1, 0, 2)
                for h in range(H):
                    xh = x * mask_x[h]
                    c0 = h * Fp
                    gemm(False, False, L.N, Fp, Fin, xh, Fin, Wcat[:, c0:], ncols, [(Fp, WW[:, c0:], 2 * R)])
                    gemm(False, False, L.N, Fp, Fin, xh, Fin, Wcat[:, R + c0:], ncols, [(Fp, WW[:, R + c0:], 2 * R)])
                    if skip:
                        gemm(False, False, L.N, Fp, Fin, xh, Fin, Wcat[:, 2 * R + c0:], ncols, [(Fp, Sk[:, c0:], R)])
                WW.mul_(mww.view(L.N, 2 * R))
                need_grad = True
                m = Z = None
            flags = (_lib.F_ELU if concat else 0) | (_lib.F_SKIP if skip else 0)
            hattn = torch.empty(L.N, R, dtype=f32, device=dev) if not concat else None
            m = torch.empty(L.N, H, dtype=f32, device=dev) if need_grad else None
            Z = torch.empty(L.N, H, dtype=f32, device=dev) if need_grad else None
            out = torch.empty(L.N, H * Fo if concat else Fo, dtype=f32, device=dev)
            part = torch.empty(lib.pygat_partials_bytes(graph.nnz, L.ts, H, Fp) // 4, dtype=f32, device=dev)
            with _span("v2_forward"):
                check(lib.pygat_gatv2_forward(graph.fwd.ref(L.ts) if tail is None else tail[1], H, Fo, float(alpha), flags,
                                              WW.data_ptr(), a2.data_ptr(),
                                              _ptr(Sk), _ptr(matt), out.data_ptr() if concat else None, _ptr(hattn), _ptr(m),
                                              _ptr(Z), part.data_ptr(), st), "gatv2_forward")
                if tail is not None:
                    check(lib.pygat_gat_forward_tail(tail[0], L.N - tail[0], H, Fo, flags, WW.data_ptr(), 2 * R, _ptr(Sk), out.data_ptr(),
                                                     _ptr(user_row), _ptr(m), _ptr(Z), None, st), "gat_forward_tail")
            if not concat:
                check(lib.pygat_head_mean(L.N, H, Fo, hattn.data_ptr(), _ptr(Sk), out.data_ptr(), st), "head_mean")
        if need_grad:
            ctx.save_for_backward(x, Wcat, a2, WW, Sk, out if concat else hattn, m, Z, mask_x, mww, matt)
            ctx.graph, ctx.L, ctx.alpha, ctx.concat, ctx.flags, ctx.Fin = graph, L, float(alpha), concat, flags, Fin
            ctx.user_row, ctx.tail = user_row, tail
        return out

    @staticmethod
    def backward(ctx, G):
        x, Wcat, a2, WW, Sk, y, m, Z, mask_x, mww, matt = ctx.saved_tensors
        graph, L, H, Fo, Fin = ctx.graph, ctx.L, ctx.L.H, ctx.L.Fo, ctx.Fin
        R, Fp = L.R, L.Fp
        dev, f32 = x.device, torch.float32
        G = G.contiguous().float()
        with torch.cuda.device(dev), gemm_mode(ctx.gemm_mode):
            st = _stream()
            LG = 2 * R + 4 * H
            GRW = torch.empty(L.N, LG, dtype=f32, device=dev)
            Gp = GRW[:, :R]
            user_row, tail = getattr(ctx, "user_row", None), getattr(ctx, "tail", None)
            fused_tail = tail is not None and not L.skip       # (a skip projection's weight gradient reads every row's Gp from GRW)
            with _span("v2_prepare"):
                check(lib.pygat_gatv2_backward_prepare(tail[0] if fused_tail else L.N, H, Fo, ctx.flags, 0 if ctx.concat else 1,
                                                       G.data_ptr(), y.data_ptr(),
                                                       _ptr(Sk), m.data_ptr(), Z.data_ptr(), WW.data_ptr(), GRW.data_ptr(), _ptr(user_row), st),
                      "gatv2_backward_prepare")
            dWW = torch.empty(L.N, 2 * R, dtype=f32, device=dev)
            da_p = torch.empty(H, Fo, dtype=f32, device=dev)
            ws = torch.empty(lib.pygat_gatv2_workspace_bytes(graph.nnz, L.ts, H, Fo) // 4 + 4, dtype=f32, device=dev)
            with _span("v2_backward_row_col"):
                check(lib.pygat_gatv2_backward(graph.fwd.ref(L.ts) if tail is None else tail[1],
                                               graph.bwd.ref(L.ts) if tail is None else tail[1],
                                               graph.perm_t.data_ptr() if matt is not None else None, graph.perm_f.data_ptr(),
                                               H, Fo, ctx.alpha, WW.data_ptr(), a2.data_ptr(), GRW.data_ptr(), _ptr(matt),
                                               dWW.data_ptr(), da_p.data_ptr(), ws.data_ptr(), st), "gatv2_backward")
                if fused_tail:         # the self-loop-only rows: dWW_i = [G_u ELU'(out_u) | 0] straight from the caller's rows
                    check(lib.pygat_gat_backward_tail(tail[0], L.N - tail[0], H, Fo, ctx.flags, G.data_ptr(), y.data_ptr(), _ptr(user_row),
                                                      dWW.data_ptr(), 2 * R, R, None, None, st), "gat_backward_tail")
                elif tail is not None:  # ... with a skip projection: Gp_i from GRW (prepared for every row)
                    dWW[tail[0]:, :R].copy_(GRW[tail[0]:, :R])
                    dWW[tail[0]:, R:].zero_()
            ncols = Wcat.shape[1]
            dW = dWs = dx = None
            if mask_x is not None:   # dropout: back through the Whi/Whj masks, then per head through its input mask
                dWW.mul_(mww.view(L.N, 2 * R))
                dWc = torch.zeros(Fin, 2 * R, dtype=f32, device=dev)
                dSc = torch.zeros(Fin, R, dtype=f32, device=dev) if L.skip else None
                need_dx = ctx.needs_input_grad[0]
                dx = torch.zeros(L.N, Fin, dtype=f32, device=dev) if need_dx else None
                dxh = torch.empty(L.N, Fin, dtype=f32, device=dev) if need_dx else None
                for h in range(H):
                    xh = x * mask_x[h]
                    c0 = h * Fp
                    for blk in (0, R):
                        gemm(True, False, Fin, Fp, L.N, xh, Fin, dWW[:, blk + c0:], 2 * R, [(Fp, dWc[:, blk + c0:], 2 * R)])
                    if L.skip:
                        gemm(True, False, Fin, Fp, L.N, xh, Fin, Gp[:, c0:], LG, [(Fp, dSc[:, c0:], R)])
                    if need_dx:
                        gemm(False, True, L.N, Fin, Fp, dWW[:, c0:], 2 * R, Wcat[:, c0:], ncols, [(Fin, dxh, Fin)], split_k=1)
                        gemm(False, True, L.N, Fin, Fp, dWW[:, R + c0:], 2 * R, Wcat[:, R + c0:], ncols, [(Fin, dxh, Fin)],
                             accumulate=True, split_k=1)
                        if L.skip:
                            gemm(False, True, L.N, Fin, Fp, Gp[:, c0:], LG, Wcat[:, 2 * R + c0:], ncols, [(Fin, dxh, Fin)],
                                 accumulate=True, split_k=1)
                        dx.addcmul_(dxh, mask_x[h])
                dv = dWc.view(Fin, 2 * H, Fp)
                dW = torch.cat([dv[:, 0:H, :Fo].permute(1, 0, 2), dv[:, H:2 * H, :Fo].permute(1, 0, 2)], dim=1).contiguous()
                if L.skip:
                    dWs = dSc.view(Fin, H, Fp)[:, :, :Fo].permute(1, 0, 2).contiguous()
                return dx, dW, da_p, dWs, None, None, None, None
            # dWcat[:, :2R] = x^T dWW ; skip columns = x^T Gp
            dWc = torch.empty(Fin, 2 * R, dtype=f32, device=dev)
            with _span("v2_wgrad"):
                gemm(True, False, Fin, 2 * R, L.N, x, Fin, dWW, 2 * R, [(2 * R, dWc, 2 * R)])
            dv = dWc.view(Fin, 2 * H, Fp)
            dW = torch.cat([dv[:, 0:H, :Fo].permute(1, 0, 2), dv[:, H:2 * H, :Fo].permute(1, 0, 2)], dim=1).contiguous()
            if L.skip:
                dSc = torch.empty(Fin, R, dtype=f32, device=dev)
                gemm(True, False, Fin, R, L.N, x, Fin, Gp, LG, [(R, dSc, R)])
                dWs = dSc.view(Fin, H, Fp)[:, :, :Fo].permute(1, 0, 2).contiguous()
            if ctx.needs_input_grad[0]:
                dx = torch.empty(L.N, Fin, dtype=f32, device=dev)
                gemm(False, True, L.N, Fin, 2 * R, dWW, 2 * R, Wcat, ncols, [(Fin, dx, Fin)], split_k=1)
                if L.skip:
                    gemm(False, True, L.N, Fin, R, Gp, LG, Wcat[:, 2 * R:], ncols, [(Fin, dx, Fin)], accumulate=True,
                         split_k=1)
        return dx, dW, da_p, dWs, None, None, None, None


def draw_masks_v2(p: float, H: int, N: int, Fin: int, Fo: int, E: int, device, generator=None):
    keep = 1.0 - p

    def mk(*shape):
        return (torch.rand(*shape, device=device, generator=generator) < keep).to(torch.float32) / keep
    return {"x": mk(H, N, Fin), "whi": mk(H, N, Fo), "whj": mk(H, N, Fo), "att": mk(E, H)}


def gatv2_level(x, graph: CSRGraph, Ws: Sequence[torch.Tensor], As: Sequence[torch.Tensor],
                Wskips: Optional[Sequence[torch.Tensor]], alpha: float, concat: bool, dropout: float = 0.0,
                masks: Optional[dict] = None) -> torch.Tensor:
    """All heads of one SpGraphAttentionLayerV2 level.  Ws: H tensors [2Fin,F']; As: H tensors of F' elements.
    dropout > 0 (training): per-head masks are drawn here unless given (`masks`, tests)."""
    W, a, Wskip = stack_heads(list(Ws), list(As), None if Wskips is None else list(Wskips))   # one launch, not a cat per kind
    if masks is None and dropout > 0.0:
        H, Fin2, Fo = W.shape
        masks = draw_masks_v2(dropout, H, x.shape[0], Fin2 // 2, Fo, graph.nnz, x.device)
    return GATv2LevelFn.apply(x, W, a, Wskip, graph, alpha, concat, masks)


class _V2Base(nn.Module):
    def __init__(self, in_features, out_features, dropout, alpha, concat=True, skip_connection=False):
        super().__init__()
        self.dropout, self.in_features, self.out_features = dropout, in_features, out_features
        self.alpha, self.concat, self.skip_connection = alpha, concat, skip_connection

    def __repr__(self):  # layers.py:231-232,315-316
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class SpGraphAttentionLayerV2(_V2Base):
    """Same constructor / parameters / initialisers as reference layers.py:239-256."""
    pattern_mode = "nonzero"

    def __init__(self, in_features, out_features, dropout, alpha, concat=True, skip_connection=False):
        super().__init__(in_features, out_features, dropout, alpha, concat, skip_connection)
        self.W = nn.Parameter(torch.empty(size=(2 * in_features, out_features)))
        nn.init.xavier_normal_(self.W.data, gain=1.414)
        self.a = nn.Parameter(torch.zeros(size=(1, out_features)))
        nn.init.xavier_normal_(self.a.data, gain=1.414)
        if self.skip_connection:
            self.skip_projection = nn.Parameter(torch.empty(size=(in_features, out_features)))
            nn.init.xavier_uniform_(self.skip_projection.data, gain=1.414)

    def forward(self, input, adj):
        return gatv2_level(input, as_graph(adj, self.pattern_mode), [self.W], [self.a],
                           [self.skip_projection] if self.skip_connection else None, self.alpha, self.concat,
                           self.dropout if self.training else 0.0)


class GraphAttentionLayerV2(_V2Base):
    """Same constructor / parameters / initialisers as reference layers.py:183-202; same function as
    the reference computes (neighbour mean of h W[Fin:], see module docstring), train-mode dropout included."""
    pattern_mode = "positive"

    def __init__(self, in_features, out_features, dropout, alpha, concat=True, skip_connection=False):
        super().__init__(in_features, out_features, dropout, alpha, concat, skip_connection)
        self.W = nn.Parameter(torch.empty(size=(2 * in_features, out_features)))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        self.a = nn.Parameter(torch.empty(size=(out_features, 1)))
        nn.init.xavier_uniform_(self.a.data, gain=1.414)
        if self.skip_connection:
            self.skip_projection = nn.Parameter(torch.empty(size=(in_features, out_features)))
            nn.init.xavier_uniform_(self.skip_projection.data, gain=1.414)

    def forward(self, h, adj, masks=None):
        """`masks` (tests only): explicit pre-scaled keep masks {"x" [1,N,Fin], "wh" [1,N,F'] (the Wh2 mask,
        layers.py:212), "att" [E,1]} instead of in-kernel draws."""
        Fo = self.out_features
        zero_a = torch.zeros(2 * Fo, 1, dtype=self.W.dtype, device=self.W.device)   # uniform attention
        graph = as_graph(adj, self.pattern_mode)
        skips = [self.skip_projection] if self.skip_connection else None
        if self.training and self.dropout > 0.0:
            # layers.py:206-221: dropout on h, on Wh1 and Wh2, on the attention; the skip term uses the dropped h
            # (layers.py:225).  Wh1 only feeds the logits, which are constant along a row, so its mask never
            # reaches the output: the layer is gat_level_dropout on W[Fin:] with a = 0.
            from .dropout import gat_level_dropout
            out = gat_level_dropout(h, graph, [self.W[self.in_features:]], [zero_a], skips, self.alpha, self.concat,
                                    self.dropout, masks=masks)
        else:
            out = gat_level(h, graph, [self.W[self.in_features:]], [zero_a], skips, self.alpha, self.concat)
        # the reference's autograd gives a and W[:Fin] exactly-zero gradients (not None): keep them in the graph
        return out + 0.0 * (self.a.sum() + self.W[:self.in_features].sum())
