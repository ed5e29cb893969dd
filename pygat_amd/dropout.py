"""Train-mode dropout (p > 0) for one GAT level, on top of the same HIP kernels.

The reference applies three dropouts inside every head (layers.py:34,37,43 dense /
132,136,153 sparse), each head drawing ITS OWN masks because `models.GAT` calls the
heads one after another (models.py:32,34):

    x_h  = dropout(x)                 -> Wh_h = x_h W_h      (and the skip term x_h Wskip_h)
    Wh_h = dropout(Wh_h)              -> s, t from the dropped Wh
    alpha = softmax(...)              -> alpha~ = dropout(alpha) weights the aggregation
                                         (sparse layer: numerators dropped AFTER the row sum)

A per-head input mask means the fused all-heads projection is no longer one GEMM:
this path runs one MFMA GEMM per head on the masked input (the citation graphs that
train with dropout are small), re-derives s,t from the dropped Wh (pygat_attn_scores)
and hands the attention mask to K2/K3b.  Masks are drawn with torch's Philox RNG and
applied with elementwise multiplies -- the only torch arithmetic in the package; the
reference's own RNG stream cannot be reproduced bit-for-bit by any other implementation,
so parity is tested with EXPLICIT masks against the oracle (tests/test_gpu_dropout.py).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import lib, check, padded_width
from .graph import CSRGraph, slot_edges_for
from .ops import _Level, _ptr, _span, _stream, gemm


def draw_masks(p: float, H: int, N: int, Fin: int, Fo: int, E: int, device, generator=None):
    """Pre-scaled keep masks (0 or 1/(1-p)): x [H,N,Fin], wh [H,N,Fo], att [E,H]."""
    keep = 1.0 - p

    def mk(*shape):
        return (torch.rand(*shape, device=device, generator=generator) < keep).to(torch.float32) / keep
    return {"x": mk(H, N, Fin), "wh": mk(H, N, Fo), "att": mk(E, H)}


class GATLevelDropoutFn(torch.autograd.Function):
    """forward(x, W[H,Fin,F'], a[H,2F'], Wskip|None, graph, alpha, concat, mask_x, mask_wh, mask_att)."""

    @staticmethod
    def forward(ctx, x, W, a, Wskip, graph: CSRGraph, alpha, concat, mask_x, mask_wh, mask_att):
        if not x.is_cuda:
            raise RuntimeError("pygat_amd: inputs must be on the GPU; the hot path has no CPU fallback")
        x = x.contiguous().float(); W = W.contiguous().float(); a = a.contiguous().float()
        H, Fin, Fo = W.shape
        skip = Wskip is not None
        if skip:
            Wskip = Wskip.contiguous().float()
        L = _Level(x, H, Fo, skip)
        L.ts = slot_edges_for(L.R, graph.slot_edges)
        dev, f32 = x.device, torch.float32
        mask_x = mask_x.to(f32).contiguous(); mask_att = mask_att.to(f32).contiguous()
        # Wh mask in the padded head-interleaved layout [N, H, Fp]
        mwh = torch.zeros(L.N, H, L.Fp, dtype=f32, device=dev)
        mwh[:, :, :Fo] = mask_wh.to(f32).permute(1, 0, 2)
        with torch.cuda.device(dev):
            st = _stream()
            Wcat = torch.empty(Fin, L.ldw, dtype=f32, device=dev)
            a_pad = torch.empty(H, 2, L.Fp, dtype=f32, device=dev)
            check(lib.pygat_pack_params(H, Fin, Fo, W.data_ptr(), a.data_ptr(), _ptr(Wskip), Wcat.data_ptr(), L.ldw,
                                        a_pad.data_ptr(), st), "pack_params")
            Wh = torch.empty(L.N, L.R, dtype=f32, device=dev)
            Sk = torch.empty(L.N, L.R, dtype=f32, device=dev) if skip else None
            for h in range(H):                     # per-head masked input (layers.py:34,132)
                xh = x * mask_x[h]
                c0 = h * L.Fp
                gemm(False, False, L.N, L.Fp, Fin, xh, Fin, Wcat[:, c0:], L.ldw, [(L.Fp, Wh[:, c0:], L.R)])
                if skip:                           # h_prime += mm(h, skip) uses the dropped h (layers.py:48,166)
                    gemm(False, False, L.N, L.Fp, Fin, xh, Fin, Wcat[:, L.R + c0:], L.ldw, [(L.Fp, Sk[:, c0:], L.R)])
            Wh.mul_(mwh.view(L.N, L.R))            # layers.py:37,136
            s = torch.empty(L.N, H, dtype=f32, device=dev); t = torch.empty(L.N, H, dtype=f32, device=dev)
            check(lib.pygat_attn_scores(L.N, H, Fo, Wh.data_ptr(), a_pad.data_ptr(), s.data_ptr(), t.data_ptr(), st),
                  "attn_scores")
            flags = (_lib.F_ELU if concat else 0) | (_lib.F_SKIP if skip else 0)
            hattn = torch.empty(L.N, L.R, dtype=f32, device=dev) if not concat else None
            m = torch.empty(L.N, H, dtype=f32, device=dev); Z = torch.empty(L.N, H, dtype=f32, device=dev)
            out = torch.empty(L.N, H * Fo if concat else Fo, dtype=f32, device=dev)
            part = torch.empty(lib.pygat_partials_bytes(graph.nnz, L.ts, H, L.Fp) // 4, dtype=f32, device=dev)
            with _span("k2_forward"):
                check(lib.pygat_gat_forward(graph.fwd.ref(L.ts), H, Fo, float(alpha), flags, Wh.data_ptr(), s.data_ptr(),
                                            a_pad.data_ptr(), _ptr(Sk), mask_att.data_ptr(),
                                            out.data_ptr() if concat else None, _ptr(hattn), m.data_ptr(), Z.data_ptr(),
                                            part.data_ptr(), st), "gat_forward")
            if not concat:
                check(lib.pygat_head_mean(L.N, H, Fo, hattn.data_ptr(), _ptr(Sk), out.data_ptr(), st), "head_mean")
        ctx.save_for_backward(x, Wcat, a_pad, Wh, s, t, Sk, out if concat else hattn, m, Z, mask_x, mwh, mask_att)
        ctx.graph, ctx.L, ctx.alpha, ctx.concat, ctx.flags = graph, L, float(alpha), concat, flags
        return out

    @staticmethod
    def backward(ctx, G):
        x, Wcat, a_pad, Wh, s, t, Sk, y, m, Z, mask_x, mwh, mask_att = ctx.saved_tensors
        graph, L, H, Fo = ctx.graph, ctx.L, ctx.L.H, ctx.L.Fo
        dev, f32 = x.device, torch.float32
        G = G.contiguous().float()
        with torch.cuda.device(dev):
            st = _stream()
            RW = L.R + 4 * H
            GR = torch.empty(L.N, RW, dtype=f32, device=dev)
            ds = torch.empty(L.N, H, dtype=f32, device=dev); dt = torch.empty(L.N, H, dtype=f32, device=dev)
            dWh = torch.empty(L.N, L.R, dtype=f32, device=dev)
            part = torch.empty(lib.pygat_partials_bytes(graph.nnz, L.ts, H, L.Fp) // 4, dtype=f32, device=dev)
            check(lib.pygat_gat_backward_prepare(L.N, H, Fo, ctx.flags, 0 if ctx.concat else 1, G.data_ptr(), y.data_ptr(),
                                                 _ptr(Sk), s.data_ptr(), m.data_ptr(), Z.data_ptr(), GR.data_ptr(), st),
                  "gat_backward_prepare")
            check(lib.pygat_gat_backward_row(graph.fwd.ref(L.ts), H, Fo, ctx.alpha, Wh.data_ptr(), a_pad.data_ptr(),
                                             GR.data_ptr(), mask_att.data_ptr(), ds.data_ptr(), part.data_ptr(), st),
                  "gat_backward_row")
            check(lib.pygat_gat_backward_col(graph.bwd.ref(L.ts), graph.perm_t.data_ptr(), H, Fo, ctx.alpha,
                                             Wh.data_ptr(), a_pad.data_ptr(), GR.data_ptr(), mask_att.data_ptr(),
                                             ds.data_ptr(), dWh.data_ptr(), dt.data_ptr(), part.data_ptr(), st),
                  "gat_backward_col")
            da = torch.empty(H, 2 * Fo, dtype=f32, device=dev)
            ws = torch.empty(lib.pygat_agrad_workspace_bytes(H, Fo) // 4, dtype=f32, device=dev)
            check(lib.pygat_a_grad(L.N, H, Fo, Wh.data_ptr(), ds.data_ptr(), dt.data_ptr(), da.data_ptr(), ws.data_ptr(), st),
                  "a_grad")
            dWh.mul_(mwh.view(L.N, L.R))           # back through the Wh dropout
            need_dx = ctx.needs_input_grad[0]
            dWc = torch.zeros(L.Fin, L.R, dtype=f32, device=dev)
            dSc = torch.zeros(L.Fin, L.R, dtype=f32, device=dev) if L.skip else None
            dx = torch.zeros(L.N, L.Fin, dtype=f32, device=dev) if need_dx else None
            dxh = torch.empty(L.N, L.Fin, dtype=f32, device=dev) if need_dx else None
            for h in range(H):
                xh = x * mask_x[h]
                c0 = h * L.Fp
                gemm(True, False, L.Fin, L.Fp, L.N, xh, L.Fin, dWh[:, c0:], L.R, [(L.Fp, dWc[:, c0:], L.R)])
                if L.skip:
                    gemm(True, False, L.Fin, L.Fp, L.N, xh, L.Fin, GR[:, L.gp_col(h):], RW, [(L.Fp, dSc[:, c0:], L.R)])
                if need_dx:
                    gemm(False, True, L.N, L.Fin, L.Fp, dWh[:, c0:], L.R, Wcat[:, c0:], L.ldw, [(L.Fin, dxh, L.Fin)],
                         split_k=1)
                    if L.skip:
                        gemm(False, True, L.N, L.Fin, L.Fp, GR[:, L.gp_col(h):], RW, Wcat[:, L.R + c0:], L.ldw,
                             [(L.Fin, dxh, L.Fin)], accumulate=True, split_k=1)
                    dx.addcmul_(dxh, mask_x[h])    # back through the per-head input dropout
            dW = torch.empty(H, L.Fin, Fo, dtype=f32, device=dev)
            check(lib.pygat_unpack_wgrad(H, L.Fin, Fo, dWc.data_ptr(), L.R, 0, dW.data_ptr(), st), "unpack")
            dWs = None
            if L.skip:
                dWs = torch.empty(H, L.Fin, Fo, dtype=f32, device=dev)
                check(lib.pygat_unpack_wgrad(H, L.Fin, Fo, dSc.data_ptr(), L.R, 0, dWs.data_ptr(), st), "unpack")
        return dx, dW, da, dWs, None, None, None, None, None, None


def gat_level_dropout(x, graph: CSRGraph, Ws: Sequence[torch.Tensor], As: Sequence[torch.Tensor],
                      Wskips: Optional[Sequence[torch.Tensor]], alpha: float, concat: bool, p: float,
                      head_mean: bool = False, masks: Optional[dict] = None, generator=None) -> torch.Tensor:
    """One level in training mode with dropout p.  `masks` (tests) = {"x","wh","att"} pre-scaled."""
    del head_mean  # implied by `concat` (models.py:23): concat=False <=> last level <=> head mean
    W = torch.stack(list(Ws), 0)
    a = torch.stack([q.reshape(-1) for q in As], 0)
    Wskip = torch.stack(list(Wskips), 0) if Wskips is not None else None
    H, Fin, Fo = W.shape
    if masks is None:
        masks = draw_masks(p, H, x.shape[0], Fin, Fo, graph.nnz, x.device, generator)
    return GATLevelDropoutFn.apply(x, W, a, Wskip, graph, alpha, concat, masks["x"], masks["wh"], masks["att"])
