"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Oracle = oracle/gat_oracle.py (PARITY UNPINNED, see its header: the reference
cannot be imported here and ships no fixtures; the oracle restates
layers.py:32-64 / 125-173 / models.py:29-35).  Ground truth is the oracle in
fp64; the tolerance is the ONE rule of tests/parity.py: forward values
within 1e-5 absolute where |ref| < 1, everything else within max(1e-5, 4 x the error of the
same oracle run in fp32), gradients after accounting for LeakyReLU branch flips of the edges
within a rounding band of the kink (bounded and verified there).
"""
import numpy as np
import pytest
import torch

from oracle import gat_oracle as O
from parity import check_level, close_fwd, close_grad

pytestmark = pytest.mark.gpu



@pytest.fixture(scope="module")
def pg():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: run with gpurun")
    import pygat_amd
    return pygat_amd


def check(outs, x, rowptr, col, W, a, Sk, concat, G, what):
    """(out, dX, dW, da, dW_skip) of run_level against the oracle, tests/parity.py's rule."""
    out, dx, dW, da, dS = outs
    grads = {"dX": dx, "dW": dW, "da": da}
    if Sk is not None:
        grads["dW_skip"] = dS
    return check_level(out, grads, x.numpy(), rowptr, col, W.numpy(), a.numpy(), 0.2, concat, G.numpy(),
                       None if Sk is None else Sk.numpy(), what=what)


def close_product(got, A64, B64, what):
    """A GEMM result against the fp64 product; ref32 = the same product formed in fp32 on the CPU."""
    ref64 = A64 @ B64
    ref32 = (A64.float() @ B64.float()).double()
    return close_grad(got, ref64.numpy(), ref32.numpy(), what)


def params(H, Fin, Fo, skip, seed):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) * (1.414 * (2.0 / (Fin + Fo)) ** 0.5)
    a = torch.randn(H, 2 * Fo, generator=g, dtype=torch.float64) * (1.414 * (2.0 / (1 + 2 * Fo)) ** 0.5)
    Sk = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) * (1.414 * (6.0 / (Fin + Fo)) ** 0.5) if skip else None
    return W, a, Sk


def run_level(pg, x64, rowptr, col, W, a, Sk, concat, G64, slot=64, need_dx=True):
    dev = "cuda:0"
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev), slot_edges=slot)
    x = x64.float().to(dev).requires_grad_(need_dx)
    Wd = W.float().to(dev).requires_grad_(True)
    ad = a.float().to(dev).requires_grad_(True)
    Sd = Sk.float().to(dev).requires_grad_(True) if Sk is not None else None
    out = pg.GATLevelFn.apply(x, Wd, ad, Sd, g, 0.2, concat)
    out.backward(G64.float().to(dev))
    torch.cuda.synchronize()
    return out, x.grad, Wd.grad, ad.grad, (Sd.grad if Sd is not None else None)


# ------------------------------------------------------------------- K0 graph
def test_dense_to_csr_and_perm(pg):
    N = 300
    rowptr, col = O.random_symmetric_csr(N, 6, 3, hub=(5, 200))
    adj = O.dense_from_csr(rowptr, col, N)
    adj[adj > 0] = torch.rand(int((adj > 0).sum())) + 0.1       # values are irrelevant, only the pattern
    g = pg.CSRGraph.from_dense(adj.cuda(), "nonzero")
    assert g.symmetric and g.n == N and g.nnz == len(col)
    assert np.array_equal(g.fwd.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(g.fwd.col.cpu().numpy(), col)
    # mirror permutation: edge k=(i,j) -> position of (j,i)
    src = np.repeat(np.arange(N), np.diff(rowptr))
    perm = g.perm_t.cpu().numpy()
    assert g.perm_f is g.perm_t                                   # involution for symmetric patterns
    assert np.array_equal(src[perm], col) and np.array_equal(col[perm], src)
    # "adj > 0" (dense layer) ignores negative entries, "adj != 0" (sparse layer) keeps them
    adj2 = adj.clone(); adj2[0, 1] = adj2[1, 0] = -1.0
    gp = pg.CSRGraph.from_dense(adj2.cuda(), "positive")
    gn = pg.CSRGraph.from_dense(adj2.cuda(), "nonzero")
    assert gn.nnz - gp.nnz in (0, 2) and gn.nnz >= g.nnz


def test_scan_large(pg):
    from pygat_amd._lib import lib, check
    n = 1_000_003
    v = torch.randint(0, 7, (n,), dtype=torch.int32, device="cuda")
    out = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.pygat_scan_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    check(lib.pygat_exclusive_scan_i32(v.data_ptr(), n, out.data_ptr(), ws.data_ptr(), None))
    ref = np.concatenate([[0], np.cumsum(v.cpu().numpy().astype(np.int64))])
    assert np.array_equal(out.cpu().numpy().astype(np.int64), ref)   # bit exact (integer work)


def test_empty_row_rejected(pg):
    rowptr = torch.tensor([0, 1, 1, 2], dtype=torch.int32, device="cuda")
    col = torch.tensor([0, 2], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        pg.CSRGraph(rowptr, col)


@pytest.mark.parametrize("H,Fo", [(2, 8), (8, 16)])   # narrow rows: K3b scatters; wide rows: K4 gathers
def test_asymmetric_pattern_transpose(pg, H, Fo):
    N, Fin = 64, 8
    rng = np.random.default_rng(0)
    dense = (rng.random((N, N)) < 0.08) | np.eye(N, dtype=bool)
    rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    col = np.nonzero(dense)[1].astype(np.int32)
    W, a, _ = params(H, Fin, Fo, False, 1)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo, dtype=torch.float64, generator=gen)
    check(run_level(pg, x, rowptr, col, W, a, None, True, G), x, rowptr, col, W, a, None, True, G, f"asymmetric {H}x{Fo}")


# ------------------------------------------------------------------- K1 GEMM
@pytest.mark.parametrize("tA,tB,M,N,K", [
    (False, False, 300, 80, 1433), (False, False, 1000, 144, 128), (False, False, 257, 272, 50),
    (True, False, 128, 128, 5000), (True, False, 1433, 64, 2708), (False, True, 999, 50, 1024),
    (False, True, 2708, 1433, 64), (False, False, 5, 3, 7),
    # fast paths: small-K register-streamed A (M >= 8192, K % 32 == 0), weight-gradient stream (huge K)
    (False, False, 10007, 144, 128), (False, True, 9000, 128, 128), (False, False, 8200, 40, 96),
    (False, False, 8193, 300, 256), (False, True, 8192, 50, 64), (True, False, 128, 128, 20011),
    (True, False, 200, 80, 5000), (True, False, 50, 1024, 4500), (True, False, 256, 132, 9001),
])
def test_gemm(pg, tA, tB, M, N, K):
    gen = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((K, M) if tA else (M, K), generator=gen)
    B = torch.randn((N, K) if tB else (K, N), generator=gen)
    A64 = A.double().t() if tA else A.double()
    B64 = B.double().t() if tB else B.double()
    ref = A64 @ B64
    Ad, Bd = A.cuda(), B.cuda()
    C1 = torch.full((M, N), float("nan"), device="cuda")
    pg.gemm(tA, tB, M, N, K, Ad, Ad.shape[1], Bd, Bd.shape[1], [(N, C1, N)])
    # two column segments + accumulate
    n1 = max(1, N // 3)
    S1 = torch.ones(M, n1, device="cuda"); S2 = torch.ones(M, N - n1 + 2, device="cuda") if N > n1 else None
    segs = [(n1, S1, n1)] + ([(N - n1, S2, N - n1 + 2)] if N > n1 else [])
    pg.gemm(tA, tB, M, N, K, Ad, Ad.shape[1], Bd, Bd.shape[1], segs, accumulate=True, split_k=1)
    torch.cuda.synchronize()
    e, own = close_product(C1, A64, B64, "C")
    close_product(S1 - 1, A64, B64[:, :n1], "seg0")
    if S2 is not None:
        close_product(S2[:, :N - n1] - 1, A64, B64[:, n1:], "seg1")
        assert bool((S2[:, N - n1:] == 1).all())
    print(f"gemm {tA},{tB} {M}x{N}x{K}: err {e:.2e} (fp32 CPU product {own:.2e})")


# ------------------------------------------------------------------- fused level
SHAPES = [  # (H, Fin, Fo, skip, concat)
    (8, 16, 8, False, True),     # Cora level 1 shape (R=64)
    (1, 64, 7, False, False),    # Cora level 2: 1 head, F'=7 (padded), mean
    (8, 64, 3, False, False),    # Pubmed level 2: 8 heads x 3, mean
    (8, 32, 16, False, True),    # RMAT headline shape (R=128)
    (4, 50, 256, True, True),    # PPI level 1: skip, R=1024 -> 4 head windows of 256 floats
    (6, 40, 121, True, False),   # PPI level 3: 6 heads x 121, mean, skip -> 3 windows of 2 heads
    (3, 10, 128, True, True),    # uneven windows: 2 heads + 1 head
    (5, 12, 100, True, False),   # mean over 5 heads in windows 2 + 2 + 1, padded F'
    (12, 8, 128, False, True),   # R = 1536: wider than any single pass ever took
    (1, 12, 16, True, True),     # one head of 16 (8-GPU shard of the headline shape)
    (2, 9, 4, False, True),
    (3, 10, 8, True, True),      # NCH = 6: idle lanes in the group
    (8, 24, 64, False, True),    # R = 512 -> 2 windows of 4 heads
]


@pytest.fixture(params=["rowlocal", "rowsum", "two-gather", "rowlocal+da"])
def backward_mode(request, pg, monkeypatch):
    """The three backward flavours of pygat_amd.ops: row sums of dz from the forward's alpha-branch shares (default) /
    K4 + row sums of its per-edge dz records / K3b (second gather) + K4.  "+da": the attention-vector gradient taken along by
    the column pass (pygat_gat_backward_col with da_part + pygat_a_grad_fold: the default on tables of 32 MB and more)
    instead of by pygat_a_grad -- wherever the pass can (one-chunk rows, one head window), else the level falls back itself."""
    if request.param.endswith("+da"):
        # the column pass takes the sums along for 8 heads x 16 in one window only (pygat_gat_backward_col_da_bytes answers 0
        # otherwise and the level keeps pygat_a_grad): every other shape would repeat the plain "rowlocal" run under a da label
        cs = getattr(request.node, "callspec", None)
        shape = (cs.params.get("H"), cs.params.get("Fo")) if cs is not None else (None, None)
        if shape != (8, 16):
            pytest.skip("da rides in the column pass for 8 heads x 16 only; this shape is the plain rowlocal run")
    monkeypatch.setattr(pg.ops, "BACKWARD_FLAVOUR", request.param.split("+")[0])
    monkeypatch.setattr(pg.ops, "DA_MIN_BYTES", 0 if request.param.endswith("+da") else 1 << 60)
    return request.param


@pytest.mark.parametrize("H,Fin,Fo,skip,concat", SHAPES)
@pytest.mark.parametrize("chunk", [64, 8])
def test_level_fwd_bwd_small(pg, backward_mode, H, Fin, Fo, skip, concat, chunk):
    N = 96
    rowptr, col = O.random_symmetric_csr(N, 5, 11 + H, hub=(3, 70))
    W, a, Sk = params(H, Fin, Fo, skip, 12 + Fo)
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    check(run_level(pg, x, rowptr, col, W, a, Sk, concat, G, slot=chunk), x, rowptr, col, W, a, Sk, concat, G,
          f"small[{H},{Fin},{Fo},{skip},{concat},slot {chunk},{backward_mode}]")


WIDE = [s for s in SHAPES if s[0] * max(4, 1 << (s[2] - 1).bit_length()) > 512]


@pytest.mark.parametrize("H,Fin,Fo,skip,concat", WIDE)
def test_level_backward_head_windows(pg, monkeypatch, backward_mode, H, Fin, Fo, skip, concat):
    """Rows wider than 512 floats on a LARGE graph run the backward in head windows of <= 256 floats (GR laid
    out window by window).  ops.BWD_WINDOW_FLOATS = 256 asks for that on a small graph (whose own default is windows
    of <= 512 floats: every other wide-row test of this file runs those): the heads per window are an ARGUMENT of the
    backward entry points since ABI 13, no environment variable of the library."""
    monkeypatch.setattr(pg.ops, "BWD_WINDOW_FLOATS", 256)
    N = 80
    hg = pg.ops.head_group(N, H, Fo)
    assert 1 <= hg < H and hg * pg.padded_width(Fo) <= 256
    rowptr, col = O.random_symmetric_csr(N, 6, 21 + H, hub=(5, 60))
    W, a, Sk = params(H, Fin, Fo, skip, 22 + Fo)
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    check(run_level(pg, x, rowptr, col, W, a, Sk, concat, G, slot=16), x, rowptr, col, W, a, Sk, concat, G,
          f"windows[{H},{Fin},{Fo},{skip},{concat},{backward_mode}]")
    monkeypatch.setattr(pg.ops, "BWD_WINDOW_FLOATS", None)
    # (a cache-resident table wider than 512 floats: windows of <= 512 floats -- attn_common.h head_group_bwd)
    assert pg.ops.head_group(N, H, Fo) == pg._lib.lib.pygat_head_group(N, H, Fo) == min(H, max(1, 512 // pg.padded_width(Fo)))


def test_da_of_the_column_pass_is_the_a_grad_pass(pg, monkeypatch):
    """da from the column pass's per-work-group records + the cut-row list (pygat_a_grad_fold) against da from the pass
    of its own (pygat_a_grad): same sums in another order, and bit-identical between two runs (no atomics across lanes:
    the LDS adds of a lane go to words only that lane touches)."""
    N, Fin, Fo, H = 3000, 32, 16, 8
    rowptr, col = O.random_symmetric_csr(N, 9, 31, hub=(7, 2500))      # a hub row: cut by every slot border
    W, a, _ = params(H, Fin, Fo, False, 32)
    gen = torch.Generator().manual_seed(33)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo, dtype=torch.float64, generator=gen)
    res = {}
    for tag, minb in (("fold", 0), ("fold2", 0), ("pass", 1 << 60)):
        monkeypatch.setattr(pg.ops, "DA_MIN_BYTES", minb)
        res[tag] = run_level(pg, x, rowptr, col, W, a, None, True, G, slot=16)
    assert torch.equal(res["fold"][3], res["fold2"][3])
    assert torch.equal(res["fold"][2], res["pass"][2])                 # dW does not depend on where da is taken
    d = (res["fold"][3] - res["pass"][3]).abs().max().item()
    assert d <= 2e-5 * res["pass"][3].abs().max().item(), d
    check(res["fold"], x, rowptr, col, W, a, None, True, G, "da in the column pass")


def test_eval_matches_both_oracle_formulations(pg):
    """dense (layers.py:32-64) and sparse (layers.py:125-173) oracles, eval mode."""
    N, Fin, Fo, H = 120, 20, 8, 4
    rowptr, col = O.random_symmetric_csr(N, 6, 5)
    W, a, _ = params(H, Fin, Fo, False, 6)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
    yd = O.level_forward(x, O.dense_from_csr(rowptr, col, N, torch.float64), W, a, 0.2, True, None, "dense")
    ys = O.level_forward(x, (rowptr, col), W, a, 0.2, True, None, "sparse")
    g = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    with torch.no_grad():
        y = pg.gat_level(x.float().cuda(), g, list(W.float().cuda()), list(a.float().cuda()), None, 0.2, True)
    y32 = O.level_forward(x.float(), (rowptr, col), W.float(), a.float(), 0.2, True, None, "sparse").double()
    close_fwd(y, yd.numpy(), "vs dense oracle", y32.numpy()); close_fwd(y, ys.numpy(), "vs sparse oracle", y32.numpy())


@pytest.mark.parametrize("name,Fin,H,Fo", [("cora", 1433, 8, 8), ("citeseer", 3703, 8, 8), ("pubmed", 500, 8, 8)])
def test_real_topology_level1(pg, topologies, name, Fin, H, Fo):
    """Real Cora / Citeseer / Pubmed topology (tests/golden/*_csr.npz), synthetic row-normalised
    sparse features (the reference's feature blobs are missing), level-1 shape of train.py:47-87."""
    rowptr, col = topologies[name]
    N = len(rowptr) - 1
    gen = torch.Generator().manual_seed(72)
    x = (torch.rand(N, Fin, generator=gen) < 0.013).double()
    x = x / x.sum(1, keepdim=True).clamp(min=1)
    W, a, _ = params(H, Fin, Fo, False, 72)
    G = torch.randn(N, H * Fo, dtype=torch.float64, generator=gen)
    rep = check(run_level(pg, x, rowptr, col, W, a, None, True, G, need_dx=(name != "citeseer")), x, rowptr, col, W, a, None,
                True, G, name)
    assert rep["out_err"] <= 1e-5          # row-normalised features: the north star's absolute bar


def test_hub_and_degree_one_rows(pg):
    """Rows cut across many slots (hub of 5000 edges, 64-edge slots) and rows that only have their self loop."""
    N, Fin, Fo, H = 6000, 16, 16, 8
    rng = np.random.default_rng(1)
    nb = rng.choice(np.arange(1, N), size=5000, replace=False)
    r = np.concatenate([np.zeros(5000, dtype=np.int64), nb, np.arange(N)])
    c = np.concatenate([nb, np.zeros(5000, dtype=np.int64), np.arange(N)])
    key = np.unique(r * N + c)
    rr, cc = key // N, (key % N).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rr, minlength=N))]).astype(np.int32)
    assert (np.diff(rowptr) == 1).sum() > 900 and np.diff(rowptr).max() == 5001
    W, a, _ = params(H, Fin, Fo, False, 3)
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo, dtype=torch.float64, generator=gen)
    check(run_level(pg, x, rowptr, cc, W, a, None, True, G), x, rowptr, cc, W, a, None, True, G, "hub + degree-1 rows")


def test_softmax_shift_invariance_and_extreme_logits(pg):
    """Large |e_ij| must not overflow: the row max is subtracted (layers.py:145-146)."""
    N, Fin, Fo, H = 64, 8, 8, 2
    rowptr, col = O.random_symmetric_csr(N, 6, 9)
    W, a, _ = params(H, Fin, Fo, False, 10)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=torch.Generator().manual_seed(11)) * 40.0
    G = torch.ones(N, H * Fo, dtype=torch.float64)
    check(run_level(pg, x, rowptr, col, W, a, None, True, G), x, rowptr, col, W, a, None, True, G, "extreme logits")


# ------------------------------------------------------------------- drop-in classes
def test_dropin_model_state_dict_and_logits(pg, topologies):
    """models.GAT drop-in: same state_dict keys/shapes (models.py:27, layers.py:21-28,111-119) and
    eval logits equal to the oracle's model_forward on the Cora topology."""
    rowptr, col = topologies["cora"]
    N, nfeat, nheads = len(rowptr) - 1, [1433, 8, 7], [8, 1]
    for cls, ashape in ((pg.SpGraphAttentionLayer, (1, 16)), (pg.GraphAttentionLayer, (16, 1))):
        torch.manual_seed(72)
        model = pg.GAT(nfeat, nheads, 2, 0.6, 0.2, cls).cuda().eval()
        sd = model.state_dict()
        assert set(sd) == {f"attention_layer_1_head_{j}.{p}" for j in range(1, 9) for p in "Wa"} | \
            {"attention_layer_2_head_1.W", "attention_layer_2_head_1.a"}
        assert tuple(sd["attention_layer_1_head_1.W"].shape) == (1433, 8)
        assert tuple(sd["attention_layer_1_head_1.a"].shape) == ashape
        assert tuple(sd["attention_layer_2_head_1.W"].shape) == (64, 7)
        gen = torch.Generator().manual_seed(1)
        x = (torch.rand(N, 1433, generator=gen) < 0.013).float()
        x = x / x.sum(1, keepdim=True).clamp(min=1)
        adj = O.dense_from_csr(rowptr, col, N)
        with torch.no_grad():
            y = model(x.cuda(), adj.cuda())
            y2 = model(x.cuda(), adj.cuda())          # cached graph
        levels = []
        for li, nh in enumerate(nheads):
            Ws = torch.stack([sd[f"attention_layer_{li+1}_head_{j+1}.W"].double().cpu() for j in range(nh)])
            As = torch.stack([sd[f"attention_layer_{li+1}_head_{j+1}.a"].double().cpu().reshape(-1) for j in range(nh)])
            levels.append(dict(W=Ws, a=As))
        ref = O.model_forward(x.double(), (rowptr, col), levels, 0.2)
        assert y.shape == (N, 7)
        close_fwd(y, ref.numpy(), f"{cls.__name__} logits")       # |logits| < 1: absolute 1e-5
        assert torch.equal(y, y2)                      # deterministic: no atomics anywhere


def test_single_layer_dropin_matches_oracle_head(pg):
    N, Fin, Fo = 80, 12, 8
    rowptr, col = O.random_symmetric_csr(N, 5, 2)
    adj = O.dense_from_csr(rowptr, col, N)
    x = torch.randn(N, Fin, generator=torch.Generator().manual_seed(3))
    for cls, form in ((pg.GraphAttentionLayer, "dense"), (pg.SpGraphAttentionLayer, "sparse")):
        for concat in (True, False):
            torch.manual_seed(5)
            layer = cls(Fin, Fo, 0.0, 0.2, concat=concat, skip_connection=True).cuda()
            y = layer(x.cuda(), adj.cuda())
            W, a, sk = (layer.W.detach().double().cpu(), layer.a.detach().double().cpu(),
                        layer.skip_projection.detach().double().cpu())
            if form == "dense":
                ref = O.dense_head_forward(x.double(), adj.double(), W, a, 0.2, concat, sk)
            else:
                ref = O.sparse_head_forward(x.double(), rowptr, col, W, a, 0.2, concat, sk)
            if form == "dense":
                ref32 = O.dense_head_forward(x, adj, W.float(), a.float(), 0.2, concat, sk.float())
            else:
                ref32 = O.sparse_head_forward(x, rowptr, col, W.float(), a.float(), 0.2, concat, sk.float())
            close_fwd(y, ref.numpy(), f"{cls.__name__} concat={concat}", ref32.double().numpy())
            y.sum().backward()
            assert layer.W.grad is not None and layer.a.grad.shape == layer.a.shape


def test_more_heads_than_one_call_holds(pg):
    """8 heads x 256 = 2048 floats per node row: one call, walked as 8 head windows inside the library."""
    N, Fin, Fo, H = 40, 6, 256, 8
    rowptr, col = O.random_symmetric_csr(N, 4, 17)
    W, a, _ = params(H, Fin, Fo, False, 18)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=torch.Generator().manual_seed(19))
    g = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    for concat in (True, False):
        ref = O.level_forward(x, (rowptr, col), W, a, 0.2, concat)
        with torch.no_grad():
            y = pg.gat_level(x.float().cuda(), g, list(W.float().cuda()), list(a.float().cuda()), None, 0.2, concat)
        ref32 = O.level_forward(x.float(), (rowptr, col), W.float(), a.float(), 0.2, concat).double()
        close_fwd(y, ref.numpy(), f"concat={concat}", ref32.numpy())


@pytest.mark.parametrize("H,Fo,skip", [(8, 16, False), (2, 64, False), (1, 128, False), (8, 16, True), (16, 8, False), (3, 32, False)])
def test_project(pg, H, Fo, skip):
    """pygat_project (tall-skinny fast path: n >= 8192, Fin % 32 == 0): Wh, Sk and s_i = Wh_i . a_src against fp64."""
    from pygat_amd._lib import lib, check
    n, Fin = 8200, 64
    Fp = pg.padded_width(Fo); R = H * Fp
    gen = torch.Generator().manual_seed(H * 100 + Fo)
    X = torch.randn(n, Fin, generator=gen); W = torch.randn(H, Fin, Fo, generator=gen) * 0.3
    a = torch.randn(H, 2 * Fo, generator=gen) * 0.5
    Ws = torch.randn(H, Fin, Fo, generator=gen) * 0.3 if skip else None
    dev = "cuda"
    ldw = -(-(R * (2 if skip else 1) + 2 * H) // 4) * 4
    Wcat = torch.empty(Fin, ldw, device=dev); a_pad = torch.empty(H, 2, Fp, device=dev)
    Wd, ad = W.to(dev).contiguous(), a.to(dev).contiguous()
    Wsd = Ws.to(dev).contiguous() if skip else None
    check(lib.pygat_pack_params(H, Fin, Fo, Wd.data_ptr(), ad.data_ptr(), Wsd.data_ptr() if skip else None, Wcat.data_ptr(),
                                ldw, a_pad.data_ptr(), None))
    Xd = X.to(dev); Wh = torch.full((n, R), float("nan"), device=dev); s = torch.full((n, H), float("nan"), device=dev)
    Sk = torch.full((n, R), float("nan"), device=dev) if skip else None
    check(lib.pygat_project(n, Fin, H, Fo, Xd.data_ptr(), Fin, Wcat.data_ptr(), ldw, a_pad.data_ptr(), Wh.data_ptr(),
                            Sk.data_ptr() if skip else None, s.data_ptr(), 1, None, -1, None))
    torch.cuda.synchronize()
    ref_wh = torch.einsum("nk,hkf->nhf", X.double(), W.double())
    wh32 = torch.einsum("nk,hkf->nhf", X, W)
    close_grad(Wh.view(n, H, Fp)[:, :, :Fo], ref_wh.numpy(), wh32.double().numpy(), "Wh")
    close_grad(s, torch.einsum("nhf,hf->nh", ref_wh, a[:, :Fo].double()).numpy(),
               torch.einsum("nhf,hf->nh", wh32, a[:, :Fo]).double().numpy(), "s")
    if skip:
        close_grad(Sk.view(n, H, Fp)[:, :, :Fo], torch.einsum("nk,hkf->nhf", X.double(), Ws.double()).numpy(),
                   torch.einsum("nk,hkf->nhf", X, Ws).double().numpy(), "Sk")
    if Fp > Fo:
        assert bool((Wh.view(n, H, Fp)[:, :, Fo:] == 0).all())


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_level(pg, monkeypatch, seed):
    """Random shapes / patterns / slot lengths (fixed seeds): forward + all gradients vs the fp64 oracle."""
    monkeypatch.setattr(pg.ops, "BACKWARD_FLAVOUR", ("two-gather", "rowlocal", "rowsum", "rowlocal")[seed % 4])   # half on the default
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(1, 400))
    H = int(rng.choice([1, 2, 3, 4, 6, 8]))
    Fo = int(rng.choice([1, 3, 4, 5, 8, 16, 17, 32, 64, 100]))
    Fin = int(rng.integers(1, 70))
    skip, concat = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    slot = int(rng.choice([4, 8, 16, 64]))
    if rng.integers(0, 2):
        rowptr, col = O.random_symmetric_csr(N, float(rng.uniform(0.5, 12)), seed, hub=(0, int(rng.integers(1, N + 1))))
    else:  # asymmetric, still with self loops (no empty row)
        dense = (rng.random((N, N)) < rng.uniform(0.01, 0.3)) | np.eye(N, dtype=bool)
        rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
        col = np.nonzero(dense)[1].astype(np.int32)
    W, a, Sk = params(H, Fin, Fo, skip, seed)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    tag = f"fuzz {seed}: N={N} H={H} Fo={Fo} Fin={Fin} skip={skip} concat={concat} slot={slot} E={len(col)}"
    check(run_level(pg, x, rowptr, col, W, a, Sk, concat, G, slot=slot), x, rowptr, col, W, a, Sk, concat, G, tag)


@pytest.mark.parametrize("n,Fin,H,Fo,with_ds", [
    (5000, 50, 8, 16, True),     # streamed-K path, [dWh | ds] as one 5-tile GEMM (the headline shape)
    (5000, 50, 8, 16, False),
    (9000, 20, 8, 32, True),     # R = 256: three column tiles, the ds columns in the last one
    (5000, 33, 3, 5, True),      # R = 24 is not a multiple of 32: general path, one GEMM per operand
    (300, 7, 2, 4, True),        # small K: general path
    (4500, 129, 4, 100, True),   # padded heads (Fp 128), M not a multiple of 128
])
def test_wgrad_with_ds_columns(pg, n, Fin, H, Fo, with_ds):
    """pygat_wgrad: dW_h = X^T (dWh'_h + ds_h (x) a_src_h), the ds term riding along as extra GEMM columns."""
    from pygat_amd._lib import lib, check
    from pygat_amd.ops import _split_k
    dev = torch.device("cuda", 0)
    Fp = pg.padded_width(Fo); R = H * Fp
    gen = torch.Generator().manual_seed(n + Fo)
    X = torch.randn(n, Fin, generator=gen, dtype=torch.float64)
    dWh = torch.zeros(n, H, Fp, dtype=torch.float64); dWh[:, :, :Fo] = torch.randn(n, H, Fo, generator=gen, dtype=torch.float64)
    ds = torch.randn(n, H, generator=gen, dtype=torch.float64)
    a_pad = torch.zeros(H, 2, Fp, dtype=torch.float64); a_pad[:, :, :Fo] = torch.randn(H, 2, Fo, generator=gen, dtype=torch.float64)
    full = dWh + (ds[:, :, None] * a_pad[None, :, 0, :] if with_ds else 0.0)
    ref = torch.einsum("nk,nhf->hkf", X, full[:, :, :Fo])
    ref32 = torch.einsum("nk,nhf->hkf", X.float(), (dWh.float() + (ds.float()[:, :, None] * a_pad.float()[None, :, 0, :] if with_ds
                                                                     else 0.0))[:, :, :Fo])
    split_k = _split_k(Fin, R + (H if with_ds else 0), n, streamed_k=True)
    ws = torch.empty(lib.pygat_wgrad_workspace_bytes(Fin, H, Fo, split_k) // 4, device=dev)
    dW = torch.empty(H, Fin, Fo, device=dev)
    Xd, dd, sd, ad = (t.float().to(dev).contiguous() for t in (X, dWh.view(n, R), ds, a_pad))
    check(lib.pygat_wgrad(n, Fin, H, Fo, Xd.data_ptr(), Fin, dd.data_ptr(), sd.data_ptr() if with_ds else None,
                          ad.data_ptr(), dW.data_ptr(), split_k, ws.data_ptr(), 0, 0, -1, None))
    torch.cuda.synchronize()
    e, own = close_grad(dW, ref.numpy(), ref32.double().numpy(), "dW")
    print(f"wgrad n={n} Fin={Fin} {H}x{Fo} ds={with_ds}: err {e:.2e} (fp32 CPU {own:.2e})")


@pytest.mark.parametrize("H,Fo,rng", [(8, 16, (2, 4)), (8, 16, (7, 1)), (5, 7, (0, 2)), (6, 64, (3, 3))])
def test_backward_head_range(pg, backward_mode, H, Fo, rng):
    """GATLevelFn(..., bwd_heads): forward for all heads, gradients only for a head range -- equal to the same
    rows of the full backward, zeros elsewhere (the rank of a head-parallel run that recomputes the others'
    forward instead of receiving it)."""
    N, Fin = 150, 12
    rowptr, col = O.random_symmetric_csr(N, 6, 31 + H, hub=(4, 100))
    W, a, _ = params(H, Fin, Fo, False, 32 + Fo)
    gen = torch.Generator().manual_seed(33)
    x = torch.randn(N, Fin, generator=gen)
    G = torch.randn(N, H * Fo, generator=gen)
    dev = "cuda:0"
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev), slot_edges=16)

    def run(bh):
        Wd = W.float().to(dev).requires_grad_(True); ad = a.float().to(dev).requires_grad_(True)
        out = pg.GATLevelFn.apply(x.to(dev), Wd, ad, None, g, 0.2, True, bh)
        out.backward(G.to(dev))
        torch.cuda.synchronize()
        return out.detach(), Wd.grad, ad.grad
    out_f, dW_f, da_f = run(None)
    out_r, dW_r, da_r = run(rng)
    hb, hr = rng
    assert torch.equal(out_f, out_r)
    sel = torch.zeros(H, dtype=torch.bool); sel[hb:hb + hr] = True
    scale = lambda t: max(1.0, float(t.abs().max()))  # noqa: E731
    assert float((dW_r[sel] - dW_f[sel]).abs().max()) <= 2e-6 * scale(dW_f)     # summation order may differ
    assert float((da_r[sel] - da_f[sel]).abs().max()) <= 2e-6 * scale(da_f)
    assert float(dW_r[~sel].abs().max()) == 0.0 and float(da_r[~sel].abs().max()) == 0.0
    with pytest.raises(ValueError):
        pg.GATLevelFn.apply(x.to(dev), W.float().to(dev).requires_grad_(True), a.float().to(dev), None, g, 0.2, True, (H - 1, 2))


def test_edge_list_intake_symmetrize_self_loops(pg, topologies):
    """CSRGraph.from_edge_index(symmetrize, self_loops) rebuilds the A + A^T + I pattern of utils.py:49-52 from a
    one-directional edge list: the Cora fixture (built that way on the CPU from data/cora/cora.cites) comes back
    bit for bit from its strict upper triangle."""
    rowptr, col = topologies["cora"]
    N = len(rowptr) - 1
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    keep = rows < col                                  # one direction, no diagonal
    dev = "cuda:0"
    r = torch.as_tensor(rows[keep], device=dev); c = torch.as_tensor(col[keep].astype(np.int64), device=dev)
    r = torch.cat([r, r[:7]]); c = torch.cat([c, c[:7]])   # duplicates are dropped
    g = pg.CSRGraph.from_edge_index(r, c, N, symmetrize=True, self_loops=True)
    assert np.array_equal(g.fwd.rowptr.cpu().numpy(), rowptr) and np.array_equal(g.fwd.col.cpu().numpy(), col)
    plain = pg.CSRGraph.from_edge_index(torch.as_tensor(rows, device=dev), torch.as_tensor(col.astype(np.int64), device=dev), N)
    assert np.array_equal(plain.fwd.col.cpu().numpy(), col)


@pytest.mark.parametrize("rowptr,col", [
    ([0, 2, 3, 5], [0, 3, 1, 0, 2]),        # column index out of range
    ([0, 2, 3, 5], [0, -1, 1, 0, 2]),       # negative column index
    ([0, 3, 2, 5], [0, 1, 1, 0, 2]),        # rowptr not monotone
    ([0, 2, 3, 4], [0, 1, 1, 0, 2]),        # rowptr[-1] != nnz
    ([1, 2, 3, 5], [0, 1, 1, 0, 2]),        # rowptr[0] != 0
    ([0, 2, 3, 5], [1, 0, 1, 0, 2]),        # unsorted row
    ([0, 2, 3, 5], [0, 0, 1, 0, 2]),        # duplicate column in a row
])
def test_malformed_csr_rejected(pg, rowptr, col):
    """A bad pattern must be a ValueError at graph build, never an out-of-bounds gather in a kernel."""
    with pytest.raises(ValueError):
        pg.CSRGraph(torch.tensor(rowptr, dtype=torch.int32, device="cuda"), torch.tensor(col, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        pg.as_graph((torch.tensor(rowptr, dtype=torch.int32, device="cuda"), torch.tensor(col, dtype=torch.int32, device="cuda")))


def test_pubmed_full_model_logits_and_output_level(pg, topologies):
    """BASELINE.json config 3 at full size: the Pubmed topology (19 717 nodes, 108 365 edges) through train.py:74-87's
    model -- 500 -> 8 heads x 8 (concat) -> 8 heads x 3 averaged (models.py:34; Cora / Citeseer have ONE output head).
    Logits against the fp64 oracle under the 8(c) rule, and the output level's gradients (8 x 3 mean, F' = 3 padded
    to 4) under the flip-aware rule of tests/parity.py."""
    from parity import close_grad, close_level_grads
    rowptr, col = topologies["pubmed"]
    N, Fin = len(rowptr) - 1, 500
    gen = torch.Generator().manual_seed(72)
    x = (torch.rand(N, Fin, generator=gen) < 0.1).double()
    x = x / x.sum(1, keepdim=True).clamp(min=1)                       # utils.normalize_features: rows sum to 1
    torch.manual_seed(72)
    model = pg.GAT([Fin, 8, 3], [8, 8], 2, 0.6, 0.2, pg.SpGraphAttentionLayer).cuda().eval()
    g = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    with torch.no_grad():
        y = model(x.float().cuda(), g)
    assert y.shape == (N, 3)
    levels = []
    for heads in model.gat_layers:
        levels.append({"W": torch.stack([h.W.detach().double().cpu() for h in heads]),
                       "a": torch.stack([h.a.detach().double().cpu().reshape(-1) for h in heads])})
    y64 = O.model_forward(x, (rowptr, col), levels, 0.2)
    y32 = O.model_forward(x.float(), (rowptr, col), [{k: v.float() for k, v in lv.items()} for lv in levels], 0.2)
    e, own = close_grad(y, y64, y32, "pubmed logits", factor=4.0)
    assert e <= 1e-5, e                                              # row-normalised features: the north star's absolute bar
    # output level alone, fed with the oracle's hidden activations
    hid = O.level_forward(x, (rowptr, col), levels[0]["W"], levels[0]["a"], 0.2, True).numpy()
    W2, a2 = levels[1]["W"].numpy(), levels[1]["a"].numpy()
    G = np.random.default_rng(3).standard_normal((N, 3))
    hd = torch.as_tensor(hid, dtype=torch.float32).cuda().requires_grad_(True)
    Wd = torch.as_tensor(W2, dtype=torch.float32).cuda().requires_grad_(True)
    ad = torch.as_tensor(a2, dtype=torch.float32).cuda().requires_grad_(True)
    out = pg.GATLevelFn.apply(hd, Wd, ad, None, g, 0.2, False)
    out.backward(torch.as_tensor(G, dtype=torch.float32).cuda())
    rep = close_level_grads({"dX": hd.grad, "dW": Wd.grad, "da": ad.grad}, hid, rowptr, col, W2, a2, 0.2, False, G,
                            what="pubmed level 2")
    print(f"pubmed: logits err {e:.2e} (fp32 oracle {own:.2e}); level-2 grads after {len(rep['hip_flips'])} flips: "
          + ", ".join(f"{n} {rep['hip'][n]:.2e} (fp32 oracle {rep['fp32'][n]:.2e})" for n in ("dX", "dW", "da")))


# ------------------------------------------------------------------- cut rows of every length, all row widths
def _ladder_graph(N=360, seed=5):
    """Symmetric pattern + self loops whose degrees run from 3 to ~180: with 4-edge slots the cut-row list holds chains of 2 ...
    40+ pieces -- packed entries (several rows per wave of the list-driven fix-ups), every remainder of the last wave, and
    `wide` entries (more than 32 pieces: a whole work-group each)."""
    rng = np.random.default_rng(seed)
    r, c = [], []
    for i in range(N):
        d = 170 if i % 45 == 0 else (40 if i % 45 == 7 else 1 + (i * 7) % 6)   # forward neighbours, symmetrised below
        nb = (i + 1 + rng.choice(N - 1, size=d, replace=False)) % N
        r.append(np.full(d, i)); c.append(nb)
    r = np.concatenate(r); c = np.concatenate(c)
    rr = np.concatenate([r, c, np.arange(N)]); cc = np.concatenate([c, r, np.arange(N)])
    key = np.unique(rr.astype(np.int64) * N + cc)
    rr = (key // N).astype(np.int32); cc = (key % N).astype(np.int32)
    rowptr = np.zeros(N + 1, dtype=np.int64); np.add.at(rowptr, rr + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cc


@pytest.mark.parametrize("H,Fo", [(1, 16), (2, 16), (4, 16), (8, 16), (8, 64)])   # 4 / 8 / 16 / 32 / 64 lanes per row
def test_cut_rows_of_every_length(pg, H, Fo):
    rowptr, col = _ladder_graph()
    deg = np.diff(rowptr)
    assert deg.max() > 32 * 4 + 8 and (deg > 8).sum() > 100 and (deg <= 8).sum() > 20   # wide chains, a long list of packed ones, uncut rows
    N, Fin = len(rowptr) - 1, 32
    W, a, _ = params(H, Fin, Fo, False, 11)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo, dtype=torch.float64, generator=gen)
    for slot in (4, 8):
        check(run_level(pg, x, rowptr, col, W, a, None, True, G, slot=slot), x, rowptr, col, W, a, None, True, G,
              f"cut rows, {H}x{Fo}, {slot}-edge slots")
