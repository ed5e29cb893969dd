"""Train-mode dropout with EXPLICIT masks: HIP path vs the oracle (fp64 autograd).

The reference draws per-head masks from torch's global RNG (layers.py:34,37,43 / 132,136,153);
no other implementation can reproduce that stream, so parity is defined on given masks.
"""
import numpy as np
import os

import pytest
import torch

from oracle import gat_oracle as O
from parity import check_autograd, close_grad
from test_gpu_parity import params, pg  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,Fin,Fo,skip,concat", [(3, 10, 8, False, True), (2, 7, 5, True, False), (8, 20, 8, True, True),
                                                  (3, 9, 128, True, True), (5, 6, 100, True, False), (6, 5, 128, True, True)])   # last two: head windows
@pytest.mark.parametrize("two_gather", [None, False, True, "wide"])
def test_dropout_explicit_masks(pg, monkeypatch, two_gather, H, Fin, Fo, skip, concat):  # noqa: F811
    from pygat_amd import dropout as D
    from pygat_amd.dropout import gat_level_dropout
    if two_gather == "wide":       # round 1's wide-operand projection instead of the mask-byte one (default backward)
        monkeypatch.setattr(D, "FORCE_WIDE", True)
        two_gather = None
    monkeypatch.setattr(pg.ops, "TWO_GATHER_BACKWARD", two_gather)   # all backward flavours carry the attention mask (None: the default, row-local one)
    monkeypatch.setattr(pg.ops, "BWD_WINDOW_FLOATS", 256)   # rows > 256 floats: backward in head windows
    N, p = 70, 0.6
    rowptr, col = O.random_symmetric_csr(N, 5, 3, hub=(2, 50))
    E = len(col)
    W, a, Sk = params(H, Fin, Fo, skip, 4)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    keep = lambda *s: (torch.rand(*s, generator=gen) >= p).double() / (1 - p)  # noqa: E731
    mx, mwh, matt = keep(H, N, Fin), keep(H, N, Fo), keep(E, H)
    # oracle (sparse formulation, masks per head; att mask is [H,E] there)
    leaves = [x, W, a] + ([Sk] if skip else [])

    def oracle(*lv):
        mk = dict(x=mx.to(lv[0].dtype), wh=mwh.to(lv[0].dtype), att=matt.t().contiguous().to(lv[0].dtype))
        return O.level_forward(lv[0], (rowptr, col), lv[1], lv[2], 0.2, concat, lv[3] if skip else None, "sparse", mk)
    dev = "cuda:0"
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev), slot_edges=16)
    xd = x.float().to(dev).requires_grad_(True)
    Ws = [W[h].float().to(dev).requires_grad_(True) for h in range(H)]
    As = [a[h].float().to(dev).reshape(1, -1).requires_grad_(True) for h in range(H)]
    Ss = [Sk[h].float().to(dev).requires_grad_(True) for h in range(H)] if skip else None
    masks = dict(x=mx.float().to(dev), wh=mwh.float().to(dev), att=matt.float().to(dev))
    out = gat_level_dropout(xd, g, Ws, As, Ss, 0.2, concat, p, masks=masks)
    out.backward(G.float().to(dev))
    got = [xd.grad, torch.stack([w.grad for w in Ws]), torch.stack([w.grad.reshape(-1) for w in As])]
    if skip:
        got.append(torch.stack([w.grad for w in Ss]))
    check_autograd(out, got, oracle, leaves, G, ["dX", "dW", "da", "dW_skip"], what=f"dropout[{H},{Fin},{Fo},{skip},{concat}]")


def test_dropout_statistics_and_model_train_mode(pg, topologies):  # noqa: F811
    """models.GAT in train mode with dropout 0.6 on the Cora topology: runs, is random across calls,
    keeps the expected scale, and p=0 in train mode equals eval (layers.py F.dropout semantics)."""
    rowptr, col = topologies["cora"]
    N = len(rowptr) - 1
    torch.manual_seed(0)
    model = pg.GAT([64, 8, 7], [8, 1], 2, 0.6, 0.2, pg.SpGraphAttentionLayer).cuda()
    x = torch.randn(N, 64, device="cuda")
    g = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    model.train()
    y1 = model(x, g); y2 = model(x, g)
    assert y1.shape == (N, 7) and torch.isfinite(y1).all() and not torch.equal(y1, y2)
    y1.sum().backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in model.parameters())
    model.eval()
    with torch.no_grad():
        ye = model(x, g)
    model0 = pg.GAT([64, 8, 7], [8, 1], 2, 0.0, 0.2, pg.SpGraphAttentionLayer).cuda()
    model0.load_state_dict(model.state_dict())
    model0.train()
    assert torch.allclose(model0(x, g), ye, atol=1e-6)


def test_in_kernel_masks_statistics_and_consistency(pg):  # noqa: F811
    """The Philox masks drawn inside the K7 kernels: keep rate and scale (F.dropout semantics), independence
    between heads / streams / seeds, determinism for one seed, and the SAME mask in the expand (forward) and
    head-sum (backward) kernels."""
    from pygat_amd._lib import lib, check
    dev = torch.device("cuda", 0)
    p, keep = 0.6, 0.4
    seed = torch.tensor([123456789], dtype=torch.int64, device=dev)
    seed2 = torch.tensor([123456790], dtype=torch.int64, device=dev)
    n = 1 << 20
    m1 = torch.empty(n + 3, device=dev); m2 = torch.empty(n + 3, device=dev); m3 = torch.empty(n + 3, device=dev)
    check(lib.pygat_dropout_mask(n + 3, p, seed.data_ptr(), 2, m1.data_ptr(), None))
    check(lib.pygat_dropout_mask(n + 3, p, seed.data_ptr(), 3, m2.data_ptr(), None))
    check(lib.pygat_dropout_mask(n + 3, p, seed2.data_ptr(), 2, m3.data_ptr(), None))
    m1b = torch.empty_like(m1)
    check(lib.pygat_dropout_mask(n + 3, p, seed.data_ptr(), 2, m1b.data_ptr(), None))
    assert torch.equal(m1, m1b)                                              # one seed, one stream -> one mask
    vals = torch.unique(m1)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / keep) < 1e-6
    sig = (keep * (1 - keep) / n) ** 0.5
    for m in (m1, m2, m3):
        assert abs(float((m > 0).float().mean()) - keep) < 5 * sig            # keep rate
        assert abs(float(m.mean()) - 1.0) < 5 * sig / keep                   # E[mask] = 1
    for a_, b_ in ((m1, m2), (m1, m3)):                                      # streams / seeds are independent
        both = float(((a_ > 0) & (b_ > 0)).float().mean())
        assert abs(both - keep * keep) < 5 * (keep * keep * (1 - keep * keep) / n) ** 0.5
    # expand: A'[i, h*Fin + k] = x[i,k] * m_h[i,k]; with x = 1 it IS the mask, [N, H, Fin]
    N, Fin, H = 501, 37, 8          # H*Fin not a multiple of 1024, Fin odd
    x = torch.ones(N, Fin, device=dev)
    A = torch.empty(N, H * Fin, device=dev)
    check(lib.pygat_dropout_expand(N, Fin, H, x.data_ptr(), Fin, None, p, seed.data_ptr(), 1, A.data_ptr(), H * Fin, None))
    M = A.view(N, H, Fin)
    assert abs(float((M > 0).float().mean()) - keep) < 5 * (keep * (1 - keep) / M.numel()) ** 0.5
    agree = float(((M[:, 0] > 0) == (M[:, 1] > 0)).float().mean())            # two heads: independent masks
    assert abs(agree - (keep * keep + (1 - keep) ** 2)) < 0.02
    xr = torch.randn(N, Fin, device=dev)
    A2 = torch.empty_like(A)
    check(lib.pygat_dropout_expand(N, Fin, H, xr.data_ptr(), Fin, None, p, seed.data_ptr(), 1, A2.data_ptr(), H * Fin, None))
    assert torch.equal(A2.view(N, H, Fin), xr[:, None, :] * M)
    # head_sum regenerates the same masks: dx[i,k] = sum_h m_h[i,k] dxe[i,h,k]
    dxe = torch.randn(N, H * Fin, device=dev)
    dx = torch.empty(N, Fin, device=dev)
    check(lib.pygat_dropout_head_sum(N, Fin, H, dxe.data_ptr(), H * Fin, None, p, seed.data_ptr(), 1, dx.data_ptr(), Fin, 0, None))
    ref = (M.double() * dxe.view(N, H, Fin).double()).sum(1)
    assert float((dx.double() - ref).abs().max()) < 1e-5
    # explicit mask [H,N,Fin] through the same kernels
    mex = (torch.rand(H, N, Fin, device=dev) < keep).float() / keep
    check(lib.pygat_dropout_expand(N, Fin, H, xr.data_ptr(), Fin, mex.data_ptr(), p, None, 0, A2.data_ptr(), H * Fin, None))
    assert torch.equal(A2.view(N, H, Fin), xr[:, None, :] * mex.permute(1, 0, 2))
    # block-diagonal packing round trip
    Fo = 5
    W = torch.randn(H, Fin, Fo, device=dev); Wsk = torch.randn(H, Fin, Fo, device=dev)
    Fp = pg.padded_width(Fo); R = H * Fp
    Bp = torch.empty(H * Fin, 2 * R, device=dev)
    check(lib.pygat_pack_blockdiag(H, Fin, Fo, W.data_ptr(), Wsk.data_ptr(), Bp.data_ptr(), 2 * R, None))
    B4 = Bp.view(H, Fin, 2, H, Fp)
    for h in range(H):
        assert torch.equal(B4[h, :, 0, h, :Fo], W[h]) and torch.equal(B4[h, :, 1, h, :Fo], Wsk[h])
    assert float(Bp.abs().sum()) == pytest.approx(float(W.abs().sum() + Wsk.abs().sum()), rel=1e-5)   # zero elsewhere
    back = torch.empty(H, Fin, Fo, device=dev)
    check(lib.pygat_unpack_blockdiag(H, Fin, Fo, Bp.data_ptr(), 2 * R, R, back.data_ptr(), None))
    assert torch.equal(back, Wsk)


def test_dropout_p_one_and_float64_inputs(pg):  # noqa: F811
    """F.dropout accepts p = 1 (everything dropped): the layer then outputs ELU(0) = 0 with zero gradients, as the
    reference's formulas give (layers.py:132-170 with h = 0).  float64 inputs are computed in float32 (the path is
    fp32, like the reference's sparse layer, which is fp32-only: layers.py:150) -- with gradients routed back to the
    float64 leaves."""
    N, Fin, Fo = 60, 10, 8
    rowptr, col = O.random_symmetric_csr(N, 5, 1)
    g = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    layer = pg.SpGraphAttentionLayer(Fin, Fo, 1.0, 0.2).cuda().train()
    x = torch.randn(N, Fin, device="cuda", requires_grad=True)
    y = layer(x, g)
    assert float(y.abs().max()) == 0.0
    y.sum().backward()
    assert float(layer.W.grad.abs().max()) == 0.0 and float(x.grad.abs().max()) == 0.0
    # float64 in -> float32 compute -> float32 out, float64 gradient on the float64 leaf
    layer = pg.SpGraphAttentionLayer(Fin, Fo, 0.0, 0.2).cuda()
    x64 = torch.randn(N, Fin, device="cuda", dtype=torch.float64, requires_grad=True)
    y = layer(x64, g)
    assert y.dtype == torch.float32
    y.sum().backward()
    assert x64.grad is not None and x64.grad.dtype == torch.float64 and torch.isfinite(x64.grad).all()
    yr = layer(x64.detach().float(), g)
    assert torch.equal(y.detach(), yr.detach())


@pytest.mark.parametrize("N,Fin,H,Fo,skip", [(3000, 1433, 8, 8, False), (700, 50, 4, 16, True), (1300, 77, 3, 64, True),
                                            (900, 33, 1, 128, True), (5000, 64, 8, 3, False), (2708, 64, 1, 7, False),
                                            (800, 100, 2, 20, True), (3, 5, 2, 1, True), (67, 128, 8, 8, False)])
def test_headmask_projection_and_weight_gradient(pg, N, Fin, H, Fo, skip):  # noqa: F811
    """The mask-byte projection (pygat_project_dropout) and its weight gradient (pygat_wgrad_dropout, K slabs over the
    nodes) through the C ABI against fp64 torch on the same decisions; and the statistics of pygat_dropout_bits
    (keep rate, independent heads, one seed -> one mask, bit h of the byte = head h)."""
    from pygat_amd._lib import lib, check
    dev = torch.device("cuda", 0)
    p, keep = 0.6, 0.4
    Fp = pg.padded_width(Fo); R = H * Fp
    assert lib.pygat_headmask_supported(H, Fo, int(skip)) == 1
    seed = torch.tensor([987654321], dtype=torch.int64, device=dev)
    bits = torch.empty(N, Fin, dtype=torch.uint8, device=dev)
    check(lib.pygat_dropout_bits(N, Fin, H, p, seed.data_ptr(), 1, bits.data_ptr(), None))
    bits2 = torch.empty_like(bits)
    check(lib.pygat_dropout_bits(N, Fin, H, p, seed.data_ptr(), 1, bits2.data_ptr(), None))
    assert torch.equal(bits, bits2)
    M = torch.stack([((bits >> h) & 1).bool() for h in range(H)])                 # [H,N,Fin]
    assert int(bits.max()) < (1 << H)
    n_el = N * Fin
    for h in range(H):
        assert abs(float(M[h].float().mean()) - keep) < 5 * (keep * (1 - keep) / n_el) ** 0.5
    if H > 1 and n_el >= 20000:          # (a 2 % band needs a sample)
        agree = float((M[0] == M[1]).float().mean())
        assert abs(agree - (keep * keep + (1 - keep) ** 2)) < 0.02
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(N, Fin, generator=gen)
    W = torch.randn(H, Fin, Fo, generator=gen) * 0.2
    a = torch.randn(H, 2 * Fo, generator=gen) * 0.2
    Ws = torch.randn(H, Fin, Fo, generator=gen) * 0.2 if skip else None
    xd, Wd, ad = x.to(dev), W.to(dev), a.to(dev)
    Wsd = Ws.to(dev) if skip else None
    ldw = -(-(R * (2 if skip else 1) + 2 * H) // 4) * 4
    Wcat = torch.empty(Fin, ldw, device=dev); a_pad = torch.empty(H, 2, Fp, device=dev)
    check(lib.pygat_pack_params(H, Fin, Fo, Wd.data_ptr(), ad.data_ptr(), Wsd.data_ptr() if skip else None, Wcat.data_ptr(), ldw,
                                a_pad.data_ptr(), None))
    Mc = M.cpu().double() / keep
    for fsplit in (1, 5):
      Wh = torch.zeros(N, R, device=dev); Sk = torch.zeros(N, R, device=dev) if skip else None
      wsp = torch.empty(lib.pygat_project_dropout_workspace_bytes(N, H, Fo, int(skip), fsplit) // 4 + 1, device=dev)
      check(lib.pygat_project_dropout(N, Fin, H, Fo, xd.data_ptr(), Fin, bits.data_ptr(), p, Wcat.data_ptr(), ldw, Wh.data_ptr(),
                                      Sk.data_ptr() if skip else None, fsplit, wsp.data_ptr(), None))
      for h in range(H):
        xm = x.double() * Mc[h]
        ref = xm @ W[h].double()
        got = Wh.view(N, H, Fp)[:, h, :Fo].double().cpu()
        close_grad(got, ref, (xm.float() @ W[h]).double(), f"Wh fsplit {fsplit} head {h}")
        if Fp > Fo:
            assert float(Wh.view(N, H, Fp)[:, h, Fo:].abs().max()) == 0.0
        if skip:
            refs = xm @ Ws[h].double()
            close_grad(Sk.view(N, H, Fp)[:, h, :Fo], refs, (xm.float() @ Ws[h]).double(), f"Sk fsplit {fsplit} head {h}")
    # weight gradient: dWc [Fin, R (+R)] = (x o m_h)^T [dWh_h | Gp_h]
    dWh = torch.randn(N, R, generator=gen).to(dev)
    RW = R + 4 * H
    GR = torch.randn(N, RW, generator=gen).to(dev) if skip else None
    ntot = R * (2 if skip else 1)
    # K slabs as the level passes them (pygat_amd.dropout._headmask_splits: slabs of >= 64 nodes until the chip is full) and
    # one odd count; a single 1300-node chain (split_k = 1) is not something the level ever asks for -- one MFMA
    # accumulator over 1300 rows sits at 4.1 x the fp32 CPU product's error, the slabs' partial sums at 1-2 x
    # (the narrow kernels of k10_narrow.hip -- cases with Fin <= 128 and <= 128 output columns -- take slabs of a few rows, one
    # wave each: pygat_amd.dropout._narrow_slabs, and a second count with ragged last slab)
    from pygat_amd.dropout import _headmask_splits, _narrow_slabs
    narrow = bool(lib.pygat_dropout_narrow(Fin, H, Fo, int(skip)))
    assert narrow == (Fin <= 128 and R * (2 if skip else 1) <= 128 and Fp <= 64) or os.environ.get("PYGAT_NARROW") == "0"
    for split_k in sorted({_narrow_slabs(N), N // 37 + 1} if narrow else {7, _headmask_splits(-(-Fin // 128), H, Fp, skip, N)}):
        ws = torch.empty(max(1, lib.pygat_wgrad_dropout_workspace_bytes(Fin, H, Fo, int(skip), split_k) // 4), device=dev)
        dWc = torch.empty(Fin, ntot, device=dev)
        check(lib.pygat_wgrad_dropout(N, Fin, H, Fo, xd.data_ptr(), Fin, bits.data_ptr(), p, dWh.data_ptr(),
                                      GR.data_ptr() if skip else None, RW, dWc.data_ptr(), split_k, ws.data_ptr(), None))
        for h in range(H):
            xm = x.double() * Mc[h]
            d_h = dWh.view(N, H, Fp)[:, h].cpu()
            ref = xm.t() @ d_h.double()
            got = dWc[:, h * Fp:(h + 1) * Fp].double().cpu()
            close_grad(got, ref, (xm.float().t() @ d_h).double(), f"dWc split_k {split_k} head {h}")
            if skip:
                g_h = GR[:, h * Fp:(h + 1) * Fp].cpu()
                refs = xm.t() @ g_h.double()
                gots = dWc[:, R + h * Fp:R + (h + 1) * Fp].double().cpu()
                close_grad(gots, refs, (xm.float().t() @ g_h).double(), f"dWc skip split_k {split_k} head {h}")
    # head sum under the same bytes
    dxe = torch.randn(N, H * Fin, generator=gen).to(dev)
    dx = torch.empty(N, Fin, device=dev)
    check(lib.pygat_dropout_head_sum_bits(N, Fin, H, dxe.data_ptr(), H * Fin, bits.data_ptr(), p, dx.data_ptr(), Fin, 0, None))
    ref = (Mc.permute(1, 0, 2) * dxe.view(N, H, Fin).double().cpu()).sum(1)
    close_grad(dx, ref, (Mc.permute(1, 0, 2).float() * dxe.view(N, H, Fin).cpu()).sum(1).double(), "head sum under the mask bytes")
    if narrow:
        # gradient into x in one launch (pygat_dx_dropout) = sum_h m_h o (dWh_h W_h^T + Gp_h Wskip_h^T)
        dxn = torch.full((N, Fin), float("nan"), device=dev)
        check(lib.pygat_dx_dropout(N, Fin, H, Fo, dWh.data_ptr(), GR.data_ptr() if skip else None, RW, bits.data_ptr(), p,
                                   Wcat.data_ptr(), ldw, dxn.data_ptr(), Fin, 0, None))
        ref64 = torch.zeros(N, Fin, dtype=torch.float64); ref32 = torch.zeros(N, Fin)
        for h in range(H):
            d_h = dWh.view(N, H, Fp)[:, h, :Fo].cpu()
            t64 = d_h.double() @ W[h].double().t(); t32 = d_h @ W[h].t()
            if skip:
                g_h = GR[:, h * Fp:h * Fp + Fo].cpu()
                t64 = t64 + g_h.double() @ Ws[h].double().t(); t32 = t32 + g_h @ Ws[h].t()
            ref64 += Mc[h] * t64; ref32 += Mc[h].float() * t32
        close_grad(dxn, ref64, ref32.double(), "dx under the mask bytes (narrow)")
        base = torch.randn(N, Fin, generator=gen).to(dev)
        acc = base.clone()
        check(lib.pygat_dx_dropout(N, Fin, H, Fo, dWh.data_ptr(), GR.data_ptr() if skip else None, RW, bits.data_ptr(), p,
                                   Wcat.data_ptr(), ldw, acc.data_ptr(), Fin, 1, None))
        assert torch.allclose(acc, base + dxn, rtol=0, atol=1e-6 * float(dxn.abs().max()) + 1e-6)
