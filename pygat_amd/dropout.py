"""Train-mode dropout (p > 0) for one GAT level, on top of the same HIP kernels.

The reference applies three dropouts inside every head (layers.py:34,37,43 dense /
132,136,153 sparse), each head drawing ITS OWN masks because `models.GAT` calls the
heads one after another (models.py:32,34):

    x_h  = dropout(x)                 -> Wh_h = x_h W_h      (and the skip term x_h Wskip_h)
    Wh_h = dropout(Wh_h)              -> s, t from the dropped Wh
    alpha = softmax(...)              -> alpha~ = dropout(alpha) weights the aggregation
                                         (sparse layer: numerators dropped AFTER the row sum)

A per-head input mask means the fused all-heads projection is no longer one GEMM on x.
Round 2 (default where supported: H <= 8, row of one head <= 256 floats): the decisions of all
heads for x[i,k] are ONE byte (bit h = head h keeps it, csrc/k7_dropout.hip dropout_bits_kernel),
and the projection and its weight gradient run for all heads in one launch with x read once --
the A tile sits in LDS with its mask bytes and the heads are an inner loop over the MFMA fragments
(csrc/k1_gemm.hip gemm_headmask_kernel).  Memory N*Fin bytes instead of N*H*Fin*4 (Citeseer:
12 MB instead of 394 MB), no products with zero blocks.  Round 1's form remains as the fallback
(PYGAT_DROPOUT_WIDE=1 forces it): the masked input written once as one wide operand A' [N, H*Fin]
(A'[i, h*Fin+k] = x[i,k] m_h[i,k]) multiplied with the block-diagonal stack of the head weights,
ONE MFMA GEMM with K = H*Fin for all heads.  s,t are re-derived from the dropped Wh
(pygat_attn_scores) and the attention mask goes to K2/K3b/K4.  In training the masks are
drawn in-kernel (Philox-4x32-10) from one int64 seed taken from torch's generator; the
reference's own RNG stream cannot be reproduced bit-for-bit by any other implementation, so
parity is tested with EXPLICIT masks against the oracle (tests/test_gpu_dropout.py), through
the same kernels.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib, ops
from .config import config as _config
from ._lib import lib, check, padded_width
from .graph import CSRGraph, slot_edges_for
from .ops import _Level, _ptr, _span, _stream, gemm, gemm_mode, get_gemm_mode, stack_heads


STREAM_X, STREAM_WH, STREAM_ATT = 1, 2, 3     # Philox stream ids of the three masks drawn from one seed
FORCE_WIDE = _config.dropout_wide   # round 1's wide-operand projection everywhere


def _headmask_splits(tiles_m: int, H: int, Fp: int, skip: bool, K: int) -> int:
    """K slabs of a head-masked GEMM: enough work-groups (tiles x head groups x slabs) for the 256 CUs, slabs of >= 64."""
    nth = max(1, (Fp * (2 if skip else 1) + 31) // 32)
    groups = -(-H // max(1, 8 // nth))
    return max(1, min(-(-384 // (tiles_m * groups)), K // 64))


def _narrow_slabs(n: int) -> int:
    """Row slabs of the narrow weight gradient (one wave each): 4-32 rows, about a thousand waves on a large graph."""
    return max(1, -(-n // max(4, min(32, n // 1024))))


def _pack_bits(mask_x: torch.Tensor) -> torch.Tensor:
    """Explicit per-head input masks [H,N,Fin] (0 or 1/(1-p)) -> one byte per input element, bit h = head h keeps."""
    H = mask_x.shape[0]
    w = (1 << torch.arange(H, device=mask_x.device, dtype=torch.int32)).view(H, 1, 1)
    return ((mask_x != 0).to(torch.int32) * w).sum(0).to(torch.uint8).contiguous()


def draw_masks(p: float, H: int, N: int, Fin: int, Fo: int, E: int, device, generator=None):
    """Explicit pre-scaled keep masks (0 or 1/(1-p)) from torch's RNG: x [H,N,Fin], wh [H,N,Fo], att [E,H].
    Only tests need them materialised; training draws the masks inside the kernels from a seed."""
    keep = 1.0 - p

    def mk(*shape):
        return (torch.rand(*shape, device=device, generator=generator) < keep).to(torch.float32) / keep
    return {"x": mk(H, N, Fin), "wh": mk(H, N, Fo), "att": mk(E, H)}


class GATLevelDropoutFn(torch.autograd.Function):
    """forward(x, W[H,Fin,F'], a[H,2F'], Wskip|None, graph, alpha, concat, p, mask_x, mask_wh, mask_att, seed).

    Either the three masks are given (pre-scaled, shapes of `draw_masks`) or `seed` (int64 [1] on the GPU)
    is, and the masks are drawn in-kernel (csrc/k7_dropout.hip)."""

    @staticmethod
    def forward(ctx, x, W, a, Wskip, graph: CSRGraph, alpha, concat, p, mask_x, mask_wh, mask_att, seed, xs=None):
        if not x.is_cuda:
            raise RuntimeError("pygat_amd: inputs must be on the GPU; the hot path has no CPU fallback")
        ctx.in_dtypes = (x.dtype, W.dtype, a.dtype, None if Wskip is None else Wskip.dtype)
        x = x.contiguous().float(); W = W.contiguous().float(); a = a.contiguous().float()
        H, Fin, Fo = W.shape
        skip = Wskip is not None
        if skip:
            Wskip = Wskip.contiguous().float()
        explicit = mask_x is not None
        if explicit == (seed is not None) or (explicit and (mask_wh is None or mask_att is None)):
            raise ValueError("pygat_amd: pass either the three masks or a seed tensor")
        L = _Level(x, H, Fo, skip)
        L.ts = slot_edges_for(L.R, graph.slot_edges)
        dev, f32 = x.device, torch.float32
        HF, R, E = H * Fin, L.R, graph.nnz
        ncb = R * (2 if skip else 1)
        p = float(p)
        ctx.gemm_mode = get_gemm_mode()     # the backward (an autograd thread) forms its GEMM products the same way
        with torch.cuda.device(dev):
            st = _stream()
            if explicit:
                mask_x = mask_x.to(f32).contiguous(); matt = mask_att.to(f32).contiguous()
                mwh = torch.zeros(L.N, H, L.Fp, dtype=f32, device=dev)      # padded head-interleaved layout
                mwh[:, :, :Fo] = mask_wh.to(f32).permute(1, 0, 2)
                mwh = mwh.view(L.N, R)
            else:
                seed = seed.contiguous()
                mwh = torch.empty(L.N, R, dtype=f32, device=dev)
                matt = torch.empty(E, H, dtype=f32, device=dev)
                check(lib.pygat_dropout_mask2(p, seed.data_ptr(), L.N * R, STREAM_WH, mwh.data_ptr(), E * H, STREAM_ATT,
                                              matt.data_ptr(), st), "dropout_mask2")     # both masks, one launch
            Wcat = torch.empty(Fin, L.ldw, dtype=f32, device=dev)
            a_pad = torch.empty(H, 2, L.Fp, dtype=f32, device=dev)
            check(lib.pygat_pack_params(H, Fin, Fo, W.data_ptr(), a.data_ptr(), _ptr(Wskip), Wcat.data_ptr(), L.ldw,
                                        a_pad.data_ptr(), st), "pack_params")
            Wh = torch.empty(L.N, R, dtype=f32, device=dev)
            Sk = torch.empty(L.N, R, dtype=f32, device=dev) if skip else None   # mm(h, skip) uses the dropped h (layers.py:48,166)
            use_bits = (not FORCE_WIDE) and bool(lib.pygat_headmask_supported(H, Fo, int(skip))) and L.hg == H
            Ae = Bp = bits = None
            # sparse input features (features.py): the projection on the non-zeros of x, same per-head decisions (drawn per
            # non-zero from the seed, or the explicit masks' bit bytes); no gradient into x is formed on this path
            sparse = xs is not None and use_bits and not ctx.needs_input_grad[0] and (not explicit or H <= 8)
            ctx.xs = xs if sparse else None
            if sparse:
                bits = _pack_bits(mask_x) if explicit else None
                with _span("k1_project"):
                    check(lib.pygat_project_sparse(L.N, Fin, H, Fo, xs.rowptr.data_ptr(), xs.col.data_ptr(), xs.val.data_ptr(),
                                                   Wcat.data_ptr(), L.ldw, p, None if explicit else seed.data_ptr(), STREAM_X,
                                                   _ptr(bits), Wh.data_ptr(), _ptr(Sk), None, st), "project_sparse")
                if bits is None:
                    bits = torch.empty(0, dtype=torch.uint8, device=dev)     # (placeholder among the saved tensors)
            elif use_bits:
                # per-head input masks as one byte per element; all heads in one launch, x read once (layers.py:34,132)
                if explicit:
                    bits = _pack_bits(mask_x)
                else:
                    bits = torch.empty(L.N, Fin, dtype=torch.uint8, device=dev)
                    check(lib.pygat_dropout_bits(L.N, Fin, H, p, seed.data_ptr(), STREAM_X, bits.data_ptr(), st), "dropout_bits")
                split_k = _headmask_splits(-(-L.N // 128), H, L.Fp, skip, Fin)
                wsp = torch.empty(lib.pygat_project_dropout_workspace_bytes(L.N, H, Fo, int(skip), split_k) // 4 + 1, dtype=f32,
                                  device=dev)
                with _span("k1_project"):
                    check(lib.pygat_project_dropout(L.N, Fin, H, Fo, x.data_ptr(), Fin, bits.data_ptr(), p, Wcat.data_ptr(),
                                                    L.ldw, Wh.data_ptr(), _ptr(Sk), split_k, wsp.data_ptr(), st), "project_dropout")
            else:
                # per-head masked input as ONE operand A' [N, H*Fin] against the block-diagonal weights B'
                Bp = torch.empty(HF, ncb, dtype=f32, device=dev)
                check(lib.pygat_pack_blockdiag(H, Fin, Fo, W.data_ptr(), _ptr(Wskip), Bp.data_ptr(), ncb, st), "pack_blockdiag")
                Ae = torch.empty(L.N, HF, dtype=f32, device=dev)
                check(lib.pygat_dropout_expand(L.N, Fin, H, x.data_ptr(), Fin, _ptr(mask_x) if explicit else None, p,
                                               None if explicit else seed.data_ptr(), STREAM_X, Ae.data_ptr(), HF, st),
                      "dropout_expand")
                with _span("k1_project"):
                    gemm(False, False, L.N, ncb, HF, Ae, HF, Bp, ncb, [(R, Wh, R)] + ([(R, Sk, R)] if skip else []))
            # the Wh dropout (layers.py:37,136) is applied in place by the score kernel: Wh *= mwh, then s, t from the masked rows
            s = torch.empty(L.N, H, dtype=f32, device=dev); t = torch.empty(L.N, H, dtype=f32, device=dev)
            check(lib.pygat_attn_scores(L.N, H, Fo, Wh.data_ptr(), mwh.data_ptr(), a_pad.data_ptr(), s.data_ptr(), t.data_ptr(), st),
                  "attn_scores")
            flags = (_lib.F_ELU if concat else 0) | (_lib.F_SKIP if skip else 0)
            hattn = torch.empty(L.N, R, dtype=f32, device=dev) if not concat else None
            m = torch.empty(L.N, H, dtype=f32, device=dev); Z = torch.empty(L.N, H, dtype=f32, device=dev)
            flavour = ops.backward_flavour(L.R)
            aneg = torch.empty(L.N, R, dtype=f32, device=dev) if flavour == "rowlocal" else None
            qneg = torch.empty(L.N, H, dtype=f32, device=dev) if flavour == "rowlocal" else None
            out = torch.empty(L.N, H * Fo if concat else Fo, dtype=f32, device=dev)
            part = torch.empty(lib.pygat_partials_bytes(E, L.ts, H, L.Fp) // 4, dtype=f32, device=dev)
            with _span("k2_forward"):
                check(lib.pygat_gat_forward(graph.fwd.ref(L.ts), H, Fo, float(alpha), flags, Wh.data_ptr(), s.data_ptr(),
                                            a_pad.data_ptr(), _ptr(Sk), matt.data_ptr(),
                                            out.data_ptr() if (concat or H == 1) else None, _ptr(hattn), m.data_ptr(), Z.data_ptr(),
                                            _ptr(aneg), _ptr(qneg), part.data_ptr(), st), "gat_forward")
            if not concat and H > 1:     # (the mean over one head is that head: K2 wrote `out` itself, see ops._level_forward)
                check(lib.pygat_head_mean(L.N, H, Fo, hattn.data_ptr(), _ptr(Sk), out.data_ptr(), st), "head_mean")
        ctx.save_for_backward(x if use_bits else Ae, bits if use_bits else Bp, a_pad, Wh, s, Sk, out if concat else hattn, m, Z,
                              mask_x if explicit else seed, mwh, matt, aneg, qneg, Wcat if use_bits else None)
        ctx.graph, ctx.L, ctx.alpha, ctx.concat, ctx.flags, ctx.p, ctx.explicit = \
            graph, L, float(alpha), concat, flags, p, explicit
        ctx.use_bits = use_bits
        ctx.flavour = flavour
        return out

    @staticmethod
    def backward(ctx, G):
        Ae, Bp, a_pad, Wh, s, Sk, y, m, Z, mx_or_seed, mwh, matt, aneg, qneg, Wcat = ctx.saved_tensors
        graph, L, H, Fo, p = ctx.graph, ctx.L, ctx.L.H, ctx.L.Fo, ctx.p
        dev, f32 = Ae.device, torch.float32
        G = G.contiguous().float()
        HF, R, Fin = H * L.Fin, L.R, L.Fin
        ncb = Bp.shape[1] if Bp.dim() == 2 else 0     # (wide path only: Bp is the block-diagonal weight stack there)
        with torch.cuda.device(dev), gemm_mode(ctx.gemm_mode):
            st = _stream()
            RW = R + 4 * H
            GR = torch.empty(L.N, RW, dtype=f32, device=dev)
            ds = torch.empty(L.N, H, dtype=f32, device=dev); dt = torch.empty(L.N, H, dtype=f32, device=dev)
            dWh = torch.empty(L.N, R, dtype=f32, device=dev)
            part = torch.empty(lib.pygat_partials_bytes(graph.nnz, L.ts, H, L.Fp) // 4, dtype=f32, device=dev)
            rowlocal = ctx.flavour == "rowlocal"
            check(lib.pygat_gat_backward_prepare(L.N, H, Fo, ctx.flags, 0 if ctx.concat else 1, G.data_ptr(), y.data_ptr(),
                                                 _ptr(Sk), s.data_ptr(), m.data_ptr(), Z.data_ptr(), GR.data_ptr(),
                                                 _ptr(aneg), _ptr(qneg), ctx.alpha, ds.data_ptr() if rowlocal else None, 0, 0, L.hg, None, st),
                  "gat_backward_prepare")
            two_gather = ctx.flavour != "rowsum"       # below: does a_grad still have to finish dWh += ds a_src ?
            if rowlocal:       # ds known from the forward's alpha-branch shares (ops.BACKWARD_FLAVOUR)
                check(lib.pygat_gat_backward_col(graph.bwd.ref(L.ts), graph.perm_t.data_ptr(), H, Fo, ctx.alpha,
                                                 Wh.data_ptr(), a_pad.data_ptr(), GR.data_ptr(), matt.data_ptr(),
                                                 ds.data_ptr(), dWh.data_ptr(), dt.data_ptr(), None, part.data_ptr(), None, 0, 0, L.hg, st),
                      "gat_backward_col")
            elif two_gather:
                check(lib.pygat_gat_backward_row(graph.fwd.ref(L.ts), H, Fo, ctx.alpha, Wh.data_ptr(), a_pad.data_ptr(),
                                                 GR.data_ptr(), matt.data_ptr(), ds.data_ptr(), part.data_ptr(), 0, 0, L.hg, st),
                      "gat_backward_row")
                check(lib.pygat_gat_backward_col(graph.bwd.ref(L.ts), graph.perm_t.data_ptr(), H, Fo, ctx.alpha,
                                                 Wh.data_ptr(), a_pad.data_ptr(), GR.data_ptr(), matt.data_ptr(),
                                                 ds.data_ptr(), dWh.data_ptr(), dt.data_ptr(), None, part.data_ptr(), None, 0, 0, L.hg, st),
                      "gat_backward_col")
            else:      # K4 writes dz per transposed edge, the row sums come from those records (ops.GATLevelFn)
                dz_t = torch.empty(graph.nnz, H, dtype=f32, device=dev)
                check(lib.pygat_gat_backward_col(graph.bwd.ref(L.ts), graph.perm_t.data_ptr(), H, Fo, ctx.alpha,
                                                 Wh.data_ptr(), a_pad.data_ptr(), GR.data_ptr(), matt.data_ptr(),
                                                 None, dWh.data_ptr(), dt.data_ptr(), dz_t.data_ptr(), part.data_ptr(), None, 0, 0, L.hg, st),
                      "gat_backward_col")
                check(lib.pygat_gat_backward_rowsum(graph.fwd.ref(L.ts), graph.perm_f.data_ptr(), H, Fo, dz_t.data_ptr(),
                                                    ds.data_ptr(), part.data_ptr(), 0, 0, st), "gat_backward_rowsum")
            da = torch.empty(H, 2 * Fo, dtype=f32, device=dev)
            ws = torch.empty(lib.pygat_agrad_workspace_bytes(H, Fo) // 4, dtype=f32, device=dev)
            # the same pass takes dWh back through the Wh dropout (x mwh), after finishing it where the flavour asks for that
            check(lib.pygat_a_grad(L.N, H, Fo, Wh.data_ptr(), ds.data_ptr(), dt.data_ptr(), da.data_ptr(), ws.data_ptr(),
                                   None if two_gather else a_pad.data_ptr(), dWh.data_ptr(), mwh.data_ptr(), 0, 0, st),
                  "a_grad")
            xs = getattr(ctx, "xs", None)
            if xs is not None:   # weight gradients on the non-zeros of x under the same decisions (forward: project_sparse)
                dW = torch.empty(H, Fin, Fo, dtype=f32, device=dev)
                dWs = torch.empty(H, Fin, Fo, dtype=f32, device=dev) if L.skip else None
                wss = torch.empty(lib.pygat_wgrad_sparse_workspace_bytes(xs.nseg, H, Fo, int(L.skip)) // 4 + 4, dtype=f32, device=dev)
                with _span("k5_wgrad"):
                    check(lib.pygat_wgrad_sparse(L.N, Fin, H, Fo, xs.nseg, xs.colseg.data_ptr(), xs.seg_col.data_ptr(),
                                                 xs.seg_begin.data_ptr(), xs.seg_end.data_ptr(), xs.trow.data_ptr(),
                                                 xs.tval.data_ptr(), p, None if ctx.explicit else mx_or_seed.data_ptr(), STREAM_X,
                                                 Bp.data_ptr() if ctx.explicit else None, dWh.data_ptr(),
                                                 GR.data_ptr() if L.skip else None, RW, wss.data_ptr(), dW.data_ptr(), _ptr(dWs), st),
                          "wgrad_sparse")
                cast = lambda g_, k: g_ if g_ is None or g_.dtype == ctx.in_dtypes[k] else g_.to(ctx.in_dtypes[k])  # noqa: E731
                return (None, cast(dW, 1), cast(da, 2), cast(dWs, 3), None, None, None, None, None, None, None, None, None)
            if ctx.use_bits:     # Ae is x and Bp the mask bytes on this path
                dx, dW, dWs = _backward_bits(ctx, Ae, Bp, Wcat, dWh, GR, RW, st)
                cast = lambda g_, k: g_ if g_ is None or g_.dtype == ctx.in_dtypes[k] else g_.to(ctx.in_dtypes[k])  # noqa: E731
                return cast(dx, 0), cast(dW, 1), cast(da, 2), cast(dWs, 3), None, None, None, None, None, None, None, None, None
            # dW_h = (x o m_h)^T dWh_h: the diagonal blocks of A'^T dWh
            dBp = torch.empty(HF, R, dtype=f32, device=dev)
            with _span("k5_wgrad"):
                gemm(True, False, HF, R, L.N, Ae, HF, dWh, R, [(R, dBp, R)])
            dW = torch.empty(H, Fin, Fo, dtype=f32, device=dev)
            check(lib.pygat_unpack_blockdiag(H, Fin, Fo, dBp.data_ptr(), R, 0, dW.data_ptr(), st), "unpack_blockdiag")
            dWs = None
            if L.skip:
                for c0, w, g0 in L.gp_windows():
                    gemm(True, False, HF, w, L.N, Ae, HF, GR[:, g0:], RW, [(w, dBp[:, c0:], R)])
                dWs = torch.empty(H, Fin, Fo, dtype=f32, device=dev)
                check(lib.pygat_unpack_blockdiag(H, Fin, Fo, dBp.data_ptr(), R, 0, dWs.data_ptr(), st), "unpack_blockdiag")
            dx = None
            if ctx.needs_input_grad[0]:
                # dxe[i, h*Fin + k] = dWh_h[i,:] . W_h[k,:] (+ Gp_h . Wskip_h); folded over the heads under the input masks
                dxe = torch.empty(L.N, HF, dtype=f32, device=dev)
                gemm(False, True, L.N, HF, R, dWh, R, Bp, ncb, [(HF, dxe, HF)], split_k=1)
                if L.skip:
                    for c0, w, g0 in L.gp_windows():
                        gemm(False, True, L.N, HF, w, GR[:, g0:], RW, Bp[:, R + c0:], ncb, [(HF, dxe, HF)],
                             accumulate=True, split_k=1)
                dx = torch.empty(L.N, Fin, dtype=f32, device=dev)
                check(lib.pygat_dropout_head_sum(L.N, Fin, H, dxe.data_ptr(), HF,
                                                 mx_or_seed.data_ptr() if ctx.explicit else None, p,
                                                 None if ctx.explicit else mx_or_seed.data_ptr(), STREAM_X,
                                                 dx.data_ptr(), Fin, 0, st), "dropout_head_sum")
        cast = lambda g_, k: g_ if g_ is None or g_.dtype == ctx.in_dtypes[k] else g_.to(ctx.in_dtypes[k])  # noqa: E731
        return cast(dx, 0), cast(dW, 1), cast(da, 2), cast(dWs, 3), None, None, None, None, None, None, None, None, None


def _backward_bits(ctx, x, bits, Wcat, dWh, GR, RW, st):
    """dW / dWskip / dX of the projection under per-head input dropout, bits form (forward: pygat_project_dropout)."""
    L, H, Fo, p = ctx.L, ctx.L.H, ctx.L.Fo, ctx.p
    dev, f32 = x.device, torch.float32
    Fin, R = L.Fin, L.R
    ntot = R * (2 if L.skip else 1)
    # dW_h = (x o m_h)^T dWh_h, dWskip_h = (x o m_h)^T Gp_h: one launch for all heads, x read once; K slabs over the nodes
    narrow = bool(lib.pygat_dropout_narrow(Fin, H, Fo, int(L.skip)))       # k10_narrow.hip: slabs = one wave's rows each
    split_k = _narrow_slabs(L.N) if narrow else _headmask_splits(-(-Fin // 128), H, L.Fp, L.skip, L.N)
    ws = torch.empty(lib.pygat_wgrad_dropout_workspace_bytes(Fin, H, Fo, int(L.skip), split_k) // 4, dtype=f32, device=dev)
    dWc = torch.empty(Fin, ntot, dtype=f32, device=dev)
    with _span("k5_wgrad"):
        check(lib.pygat_wgrad_dropout(L.N, Fin, H, Fo, x.data_ptr(), Fin, bits.data_ptr(), p, dWh.data_ptr(),
                                      GR.data_ptr() if L.skip else None, RW, dWc.data_ptr(), split_k, ws.data_ptr(), st),
              "wgrad_dropout")
    dW = torch.empty(H, Fin, Fo, dtype=f32, device=dev)
    check(lib.pygat_unpack_wgrad(H, Fin, Fo, dWc.data_ptr(), ntot, 0, dW.data_ptr(), st), "unpack_wgrad")
    dWs = None
    if L.skip:
        dWs = torch.empty(H, Fin, Fo, dtype=f32, device=dev)
        check(lib.pygat_unpack_wgrad(H, Fin, Fo, dWc.data_ptr(), ntot, R, dWs.data_ptr(), st), "unpack_wgrad")
    dx = None
    if ctx.needs_input_grad[0] and narrow:
        dx = torch.empty(L.N, Fin, dtype=f32, device=dev)
        check(lib.pygat_dx_dropout(L.N, Fin, H, Fo, dWh.data_ptr(), GR.data_ptr() if L.skip else None, RW, bits.data_ptr(), p,
                                   Wcat.data_ptr(), L.ldw, dx.data_ptr(), Fin, 0, st), "dx_dropout")
    elif ctx.needs_input_grad[0]:
        # dxe[i, h*Fin + k] = dWh_h[i,:] . W_h[k,:] (+ Gp_h . Wskip_h), head by head, then folded under the mask bits
        HF = H * Fin
        dxe = torch.empty(L.N, HF, dtype=f32, device=dev)
        for h in range(H):
            c0 = h * L.Fp
            gemm(False, True, L.N, Fin, L.Fp, dWh[:, c0:], R, Wcat[:, c0:], L.ldw, [(Fin, dxe[:, h * Fin:], HF)], split_k=1)
            if L.skip:
                gemm(False, True, L.N, Fin, L.Fp, GR[:, c0:], RW, Wcat[:, R + c0:], L.ldw, [(Fin, dxe[:, h * Fin:], HF)],
                     accumulate=True, split_k=1)
        dx = torch.empty(L.N, Fin, dtype=f32, device=dev)
        check(lib.pygat_dropout_head_sum_bits(L.N, Fin, H, dxe.data_ptr(), HF, bits.data_ptr(), p, dx.data_ptr(), Fin, 0, st),
              "dropout_head_sum_bits")
    return dx, dW, dWs


def gat_level_dropout(x, graph: CSRGraph, Ws: Sequence[torch.Tensor], As: Sequence[torch.Tensor],
                      Wskips: Optional[Sequence[torch.Tensor]], alpha: float, concat: bool, p: float,
                      head_mean: bool = False, masks: Optional[dict] = None, generator=None, xs=None) -> torch.Tensor:
    """One level in training mode with dropout p.  `masks` (tests) = {"x","wh","att"} pre-scaled; without them the
    masks are drawn in-kernel from one int64 seed taken from torch's (graph-safe) generator."""
    del head_mean  # implied by `concat` (models.py:23): concat=False <=> last level <=> head mean
    W, a, Wskip = stack_heads(list(Ws), list(As), None if Wskips is None else list(Wskips))   # one launch, not a cat per kind
    if masks is not None:
        return GATLevelDropoutFn.apply(x, W, a, Wskip, graph, alpha, concat, p, masks["x"], masks["wh"], masks["att"], None, xs)
    seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=x.device, generator=generator)
    return GATLevelDropoutFn.apply(x, W, a, Wskip, graph, alpha, concat, p, None, None, None, seed, xs)
