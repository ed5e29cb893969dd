"""Size-independent properties of the HIP level on the GPU: equivariance under node renumbering, independence of
the order in which slots are walked, and the register / scratch footprint the headline instantiations were tuned at."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _permuted_csr(rowptr, col, perm):
    """CSR of P A P^T: new node p = old node perm[p]."""
    N = len(rowptr) - 1
    inv = np.empty(N, dtype=np.int64); inv[perm] = np.arange(N)
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    r2, c2 = inv[rows], inv[col]
    order = np.lexsort((c2, r2))
    r2, c2 = r2[order], c2[order]
    rp2 = np.zeros(N + 1, dtype=np.int64); np.cumsum(np.bincount(r2, minlength=N), out=rp2[1:])
    return rp2.astype(np.int32), c2.astype(np.int32)


@pytest.mark.parametrize("cfg", [dict(N=6000, H=8, Fo=16, Fin=128, skip=False, concat=True, deg=9, hub=(11, 3000)),   # the headline shape
                                 dict(N=2500, H=4, Fo=64, Fin=48, skip=True, concat=True, deg=12, hub=(5, 800)),
                                 dict(N=3000, H=3, Fo=7, Fin=33, skip=True, concat=False, deg=5, hub=(2, 1500))])
def test_level_is_equivariant_under_node_renumbering(cfg):
    """level(P x, P A P^T) = P level(x, A); dX permutes with the nodes; dW, da, dWskip are sums over nodes / edges and agree up to
    summation order (the slots, the cut rows and the fix-up lists of the two runs are entirely different)."""
    import pygat_amd as pg
    from oracle import gat_oracle as O
    dev = torch.device("cuda", 0)
    N, H, Fo, Fin = cfg["N"], cfg["H"], cfg["Fo"], cfg["Fin"]
    rowptr, col = O.random_symmetric_csr(N, cfg["deg"], 17, hub=cfg["hub"])
    perm = np.random.default_rng(3).permutation(N)
    rp2, c2 = _permuted_csr(np.asarray(rowptr), np.asarray(col), perm)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(N, Fin, generator=g)
    W = torch.randn(H, Fin, Fo, generator=g) * (1.414 * (2.0 / (Fin + Fo)) ** 0.5)
    a = torch.randn(H, 2 * Fo, generator=g) * 0.4
    S = torch.randn(H, Fin, Fo, generator=g) * 0.2 if cfg["skip"] else None
    G = torch.randn(N, H * Fo if cfg["concat"] else Fo, generator=g)
    pt = torch.from_numpy(perm)

    def run(rp, c, xx, GG):
        graph = pg.CSRGraph(torch.as_tensor(rp, device=dev), torch.as_tensor(c, device=dev))
        leaves = [t.to(dev).clone().requires_grad_(True) if t is not None else None for t in (xx, W, a, S)]
        out = pg.GATLevelFn.apply(leaves[0], leaves[1], leaves[2], leaves[3], graph, 0.2, cfg["concat"])
        out.backward(GG.to(dev))
        return [out.detach().cpu()] + [None if t is None else t.grad.cpu() for t in leaves]
    o1, dx1, dW1, da1, dS1 = run(rowptr, col, x, G)
    o2, dx2, dW2, da2, dS2 = run(rp2, c2, x[pt], G[pt])
    sc = lambda t: max(1.0, float(t.abs().max()))        # noqa: E731
    assert float((o2 - o1[pt]).abs().max()) <= 5e-6 * sc(o1)
    assert float((dx2 - dx1[pt]).abs().max()) <= 2e-5 * sc(dx1)
    for p, q in ((dW1, dW2), (da1, da2), (dS1, dS2)):
        if p is not None:
            assert float((p - q).abs().max()) <= 5e-5 * sc(p)


def test_results_do_not_depend_on_the_slot_order():
    """pygat_graph.slot_order (ABI 14) only changes which rows' gathers are in flight together: out, dX and dW come out BIT FOR
    BIT the same under a random order of the slots (rows, partial records and fix-ups go by the slot id), da up to the order in
    which the column pass folds its per-work-group records."""
    import pygat_amd as pg
    from pygat_amd import graph as G_
    from pygat_amd.rmat import rmat_csr
    dev = torch.device("cuda", 0)
    rowptr, col = rmat_csr(17, 700_000, seed=5, device=dev)      # 131 072 nodes: tables of 64 MB, the da-in-K4 path included
    N = rowptr.numel() - 1
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, 128, generator=g, device=dev)
    W = torch.randn(8, 128, 16, generator=g, device=dev) * 0.17
    a = torch.randn(8, 32, generator=g, device=dev) * 0.3
    Gr = torch.randn(N, 128, generator=g, device=dev)

    def run():
        graph = pg.CSRGraph(rowptr, col)
        leaves = [t.clone().requires_grad_(True) for t in (x, W, a)]
        out = pg.GATLevelFn.apply(leaves[0], leaves[1], leaves[2], None, graph, 0.2, True)
        out.backward(Gr)
        return [out.detach()] + [t.grad for t in leaves]
    ref = run()
    try:
        G_.SLOT_ORDER_FN = lambda pat, meta: torch.randperm(meta.shape[0], generator=torch.Generator().manual_seed(2)).to(torch.int32).to(meta.device)
        got = run()
    finally:
        G_.SLOT_ORDER_FN = None
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
    assert float((got[3] - ref[3]).abs().max()) <= 1e-5 * max(1.0, float(ref[3].abs().max()))


def test_headline_kernels_keep_their_register_footprint():
    """the headline instantiations of K2 / K4 were tuned at four waves per SIMD without scratch (K4 with the
    a-gradient sums: 122 of 128 VGPRs; variants at 130 VGPRs or with scratch measured 10-75 % slower).  `amdgpu_waves_per_eu` makes
    hipcc SPILL rather than fail, so a compiler or flag change would show up only as a timing drift: ask the loaded code object
    (pygat_kernel_footprint -> hipFuncGetAttributes)."""
    from pygat_amd._lib import lib
    for name, max_regs in (("k2_headline", 128), ("k4_headline_da", 128), ("tn_x3w", 256), ("x3gw", 256)):
        regs, scratch = C.c_int(-1), C.c_int(-1)
        rc = lib.pygat_kernel_footprint(name.encode(), C.byref(regs), C.byref(scratch))
        assert rc == 0, (name, lib.pygat_last_error())
        assert scratch.value == 0 and 0 < regs.value <= max_regs, (name, regs.value, scratch.value)


@pytest.mark.parametrize("cfg", [dict(N=5000, H=8, Fo=16, Fin=128, skip=False, concat=True, sym=True),      # the headline shape
                                 dict(N=3000, H=4, Fo=64, Fin=40, skip=True, concat=True, sym=True),        # skip rows ride in internal order
                                 dict(N=2000, H=1, Fo=7, Fin=33, skip=True, concat=False, sym=True),        # one head, mean: K2 writes `out` itself
                                 dict(N=2500, H=3, Fo=8, Fin=20, skip=False, concat=True, sym=False),       # asymmetric pattern: own transpose
                                 dict(N=6000, H=8, Fo=16, Fin=64, skip=False, concat=True, sym=True, iso=0.5),   # half the nodes self-loop-only: the tail streams
                                 dict(N=4000, H=2, Fo=5, Fin=24, skip=True, concat=True, sym=True, iso=0.3)])    # ... with skip rows and padded heads
def test_internal_degree_order_is_invisible(cfg, monkeypatch):
    """ops.RENUMBER (round 5): a first level may run on the graph renumbered by descending degree, x permuted once, `out` written
    and G / the saved output read at the caller's rows inside K2 / K3a.  Forced here on small graphs: equal to the caller-order run
    (summation order inside a softmax row differs) and to the oracle under the one parity rule."""
    import pygat_amd as pg
    from oracle import gat_oracle as O
    import parity
    dev = torch.device("cuda", 0)
    N, H, Fo, Fin = cfg["N"], cfg["H"], cfg["Fo"], cfg["Fin"]
    iso = cfg.get("iso", 0.0)
    if iso:      # a connected part + nodes with nothing but their self loop (55 % of the R-MAT workload), ids shuffled
        n0 = int(N * (1 - iso))
        rp0, c0 = O.random_symmetric_csr(n0, 6, 29, hub=(4, min(n0 - 1, 1200)))
        rp0, c0 = np.asarray(rp0, dtype=np.int64), np.asarray(c0, dtype=np.int64)
        relabel = np.random.default_rng(30).permutation(N)            # node v of the construction -> caller's id
        rows = np.concatenate([np.repeat(np.arange(n0), np.diff(rp0)), np.arange(n0, N)])
        cols = np.concatenate([c0, np.arange(n0, N)])
        r2, c2 = relabel[rows], relabel[cols]
        o = np.lexsort((c2, r2))
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=N))]).astype(np.int32)
        col = c2[o].astype(np.int32)
    else:
        rowptr, col = O.random_symmetric_csr(N, 6, 29, hub=(4, min(N - 1, 1200)))
    rowptr, col = np.asarray(rowptr), np.asarray(col)
    if not cfg["sym"]:       # drop every third off-diagonal edge: an asymmetric pattern (self loops kept)
        rows = np.repeat(np.arange(N), np.diff(rowptr))
        keep = (rows == col) | ((np.arange(len(col)) % 3) != 0)
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=N))]).astype(np.int32)
        col = col[keep].astype(np.int32)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, Fin, generator=g, dtype=torch.float64)
    W = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) * (1.414 * (2.0 / (Fin + Fo)) ** 0.5)
    a = torch.randn(H, 2 * Fo, generator=g, dtype=torch.float64) * 0.4
    S = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) * 0.2 if cfg["skip"] else None
    G = torch.randn(N, H * Fo if cfg["concat"] else Fo, generator=g, dtype=torch.float64)

    def run(renumber):
        monkeypatch.setattr(pg.ops, "RENUMBER", renumber)
        monkeypatch.setattr(pg.ops, "RENUMBER_MIN_BYTES", 0)
        graph = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
        xd = x.float().to(dev)
        ps = [None if t is None else t.float().to(dev).requires_grad_(True) for t in (W, a, S)]
        out = pg.GATLevelFn.apply(xd, ps[0], ps[1], ps[2], graph, 0.2, cfg["concat"])
        out.backward(G.float().to(dev))
        used = graph._ordered is not None
        if used and iso:     # ... and the self-loop-only tail went through its own streams
            tails = [v for k_, v in graph._ordered[0].fwd._alt.items() if isinstance(k_, tuple) and k_[0] == "tail"]
            assert tails and tails[0] is not None and N - tails[0][0] >= 0.9 * iso * N
        return out.detach().cpu(), [None if p is None else p.grad.cpu() for p in ps], used
    o0, g0, used0 = run(False)
    o1, g1, used1 = run(True)
    assert used1 and not used0                      # the renumbered run really took the internal order
    sc = lambda t: max(1.0, float(t.abs().max()))   # noqa: E731
    assert float((o1 - o0).abs().max()) <= 5e-6 * sc(o0)
    for p, q in zip(g0, g1):
        if p is not None:
            assert float((p - q).abs().max()) <= 5e-5 * sc(p)
    grads = {"dX": None, "dW": g1[0].numpy(), "da": g1[1].numpy()}
    if S is not None:
        grads["dW_skip"] = g1[2].numpy()
    parity.check_level(o1.numpy(), grads, x.numpy(), rowptr, col, W.numpy(), a.numpy(), 0.2, cfg["concat"], G.numpy(),
                       None if S is None else S.numpy(), what=f"internal-order{cfg}", verbose=False)


def test_model_runs_in_internal_order(monkeypatch):
    """pygat_amd.GAT on a large graph: x permuted once, EVERY level (hidden ones too: their input is the previous level's output,
    already internal) on the degree-ordered pattern, the self-loop-only tail through its streams at every concat level, the final
    logits put back.  Forced on a small graph: logits and every parameter gradient equal the caller-order run's."""
    import pygat_amd as pg
    from oracle import gat_oracle as O
    dev = torch.device("cuda", 0)
    N, n0 = 5000, 3000                                      # 40 % of the nodes have nothing but their self loop
    rp0, c0 = O.random_symmetric_csr(n0, 6, 41, hub=(9, 1500))
    rp0, c0 = np.asarray(rp0, dtype=np.int64), np.asarray(c0, dtype=np.int64)
    relabel = np.random.default_rng(42).permutation(N)
    rows = np.concatenate([np.repeat(np.arange(n0), np.diff(rp0)), np.arange(n0, N)])
    cols = np.concatenate([c0, np.arange(n0, N)])
    r2, c2 = relabel[rows], relabel[cols]
    o = np.lexsort((c2, r2))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=N))]).astype(np.int32)
    col = c2[o].astype(np.int32)
    graph = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
    torch.manual_seed(5)
    kw = dict(nfeat=[24, 16, 16, 6], nheads=[4, 2, 3], nlayers=3, dropout=0.0, alpha=0.2, layer_type=pg.SpGraphAttentionLayer,
              skip_connection=True)
    model = pg.GAT(**kw).to(dev)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(N, 24, generator=g).to(dev)
    G = torch.randn(N, 6, generator=g).to(dev)
    monkeypatch.setattr(pg.ops, "RENUMBER_MIN_BYTES", 0)
    monkeypatch.setattr(pg.ops, "RENUMBER_MIN_BYTES_TAIL", 0)
    outs = {}
    for renumber in (False, True):
        monkeypatch.setattr(pg.ops, "RENUMBER", renumber)
        model.zero_grad(set_to_none=True)
        used = []
        orig = pg.CSRGraph.internal_view
        monkeypatch.setattr(pg.CSRGraph, "internal_view", lambda self: (used.append(1), orig(self))[1])
        y = model(x, graph)
        y.backward(G)
        monkeypatch.setattr(pg.CSRGraph, "internal_view", orig)
        assert bool(used) == renumber
        outs[renumber] = (y.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    y0, g0 = outs[False]
    y1, g1 = outs[True]
    sc = lambda t: max(1.0, float(t.abs().max()))        # noqa: E731
    assert float((y1 - y0).abs().max()) <= 5e-6 * sc(y0)
    for k in g0:
        assert float((g1[k] - g0[k]).abs().max()) <= 5e-5 * sc(g0[k]), k


@pytest.mark.parametrize("skip", [False, True])
def test_gatv2_level_internal_order_and_tail(skip, monkeypatch):
    """The SpGraphAttentionLayerV2 level (layers.py:234-316) in the internal degree order with the self-loop-only tail streamed
    (h'_i = ELU(Whi_i (+ skip_i)), dWW_i = [Gp_i | 0]): equal to the caller-order run, forced on a small graph."""
    import pygat_amd as pg
    from oracle import gat_oracle as O
    dev = torch.device("cuda", 0)
    N, n0, H, Fo, Fin = 4000, 2400, 4, 16, 32
    rp0, c0 = O.random_symmetric_csr(n0, 6, 51, hub=(7, 1000))
    rp0, c0 = np.asarray(rp0, dtype=np.int64), np.asarray(c0, dtype=np.int64)
    relabel = np.random.default_rng(52).permutation(N)
    rows = np.concatenate([np.repeat(np.arange(n0), np.diff(rp0)), np.arange(n0, N)])
    cols = np.concatenate([c0, np.arange(n0, N)])
    r2, c2 = relabel[rows], relabel[cols]
    o = np.lexsort((c2, r2))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=N))]).astype(np.int32)
    col = c2[o].astype(np.int32)
    g = torch.Generator().manual_seed(53)
    x = torch.randn(N, Fin, generator=g).to(dev)
    W = (torch.randn(H, 2 * Fin, Fo, generator=g) * 0.2).to(dev)
    a = (torch.randn(H, Fo, generator=g) * 0.4).to(dev)
    S = (torch.randn(H, Fin, Fo, generator=g) * 0.2).to(dev) if skip else None
    G = torch.randn(N, H * Fo, generator=g).to(dev)
    monkeypatch.setattr(pg.ops, "RENUMBER_MIN_BYTES", 0)
    monkeypatch.setattr(pg.ops, "RENUMBER_MIN_BYTES_TAIL", 0)
    res = {}
    for renumber in (False, True):
        monkeypatch.setattr(pg.ops, "RENUMBER", renumber)
        graph = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
        ps = [None if t is None else t.clone().requires_grad_(True) for t in (W, a, S)]
        out = pg.GATv2LevelFn.apply(x, ps[0], ps[1], ps[2], graph, 0.2, True, None)
        out.backward(G)
        assert (graph._ordered is not None) == renumber
        res[renumber] = (out.detach(), [None if p is None else p.grad for p in ps])
    sc = lambda t: max(1.0, float(t.abs().max()))        # noqa: E731
    assert float((res[True][0] - res[False][0]).abs().max()) <= 5e-6 * sc(res[False][0])
    for p, q in zip(res[False][1], res[True][1]):
        if p is not None:
            assert float((p - q).abs().max()) <= 5e-5 * sc(p)
