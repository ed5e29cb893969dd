"""SURVEY.md 7.3 property tests on the CPU oracle, driven by `hypothesis` (shape / degree / pattern generator): the two
formulations of the reference -- dense N x N (layers.py:32-64) and edge list (layers.py:125-173) -- are the same function on
every symmetric pattern with a full diagonal, and a level is equivariant under node renumbering.  (The oracle is this repo's
restatement, parity unpinned: oracle/gat_oracle.py header.)"""
import numpy as np
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import gat_oracle as O


@st.composite
def level_case(draw):
    N = draw(st.integers(2, 40))
    Fin = draw(st.integers(1, 12))
    Fo = draw(st.integers(1, 9))
    H = draw(st.integers(1, 3))
    density = draw(st.sampled_from([0.0, 0.05, 0.3, 1.0]))       # 0: self loops only ... 1: complete graph
    hub = draw(st.booleans())
    seed = draw(st.integers(0, 2 ** 16))
    rng = np.random.default_rng(seed)
    adj = rng.random((N, N)) < density
    if hub:
        adj[draw(st.integers(0, N - 1)), :] = True
    adj = adj | adj.T | np.eye(N, dtype=bool)                      # utils.py:49-52: symmetrised, self loops
    rowptr = np.concatenate([[0], np.cumsum(adj.sum(1))]).astype(np.int32)
    col = np.nonzero(adj)[1].astype(np.int32)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Fin, generator=g, dtype=torch.float64) * draw(st.sampled_from([0.1, 1.0, 8.0]))
    W = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64)
    a = torch.randn(H, 2 * Fo, generator=g, dtype=torch.float64)
    Sk = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) if draw(st.booleans()) else None
    return dict(N=N, adj=adj, rowptr=rowptr, col=col, x=x, W=W, a=a, Sk=Sk, concat=draw(st.booleans()), seed=seed, H=H, Fo=Fo)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(level_case())
def test_dense_equals_sparse_on_generated_patterns(c):
    adj = torch.from_numpy(c["adj"].astype(np.float64))
    yd = O.level_forward(c["x"], adj, c["W"], c["a"], 0.2, c["concat"], c["Sk"], "dense")
    ys = O.level_forward(c["x"], (c["rowptr"], c["col"]), c["W"], c["a"], 0.2, c["concat"], c["Sk"], "sparse")
    scale = max(1.0, float(yd.abs().max()))
    assert float((yd - ys).abs().max()) <= 1e-11 * scale


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(level_case())
def test_level_is_equivariant_under_node_renumbering(c):
    """P a permutation of the nodes: level(P x, P A P^T) = P level(x, A), and the hand-derived gradients follow (dX permutes,
    dW / da are sums over nodes / edges: equal up to summation order)."""
    N = c["N"]
    perm = torch.from_numpy(np.random.default_rng(c["seed"] + 1).permutation(N))      # new position p holds old node perm[p]
    adjp = c["adj"][perm.numpy()][:, perm.numpy()]
    rowptr_p = np.concatenate([[0], np.cumsum(adjp.sum(1))]).astype(np.int32)
    col_p = np.nonzero(adjp)[1].astype(np.int32)
    G = torch.randn(N, c["H"] * c["Fo"] if c["concat"] else c["Fo"], dtype=torch.float64, generator=torch.Generator().manual_seed(c["seed"] + 2))
    x, W, a, Sk = (None if t is None else t.numpy() for t in (c["x"], c["W"], c["a"], c["Sk"]))
    p, Gn = perm.numpy(), G.numpy()
    r = O.csr_layer_fwd_bwd(x, c["rowptr"], c["col"], W, a, 0.2, c["concat"], Gn, Sk)
    rp = O.csr_layer_fwd_bwd(x[p], rowptr_p, col_p, W, a, 0.2, c["concat"], Gn[p], Sk)
    tol = lambda t: 1e-10 * max(1.0, float(np.abs(t).max()))       # noqa: E731
    assert np.abs(rp["out"] - r["out"][p]).max() <= tol(r["out"])
    assert np.abs(rp["dX"] - r["dX"][p]).max() <= tol(r["dX"])
    assert np.abs(rp["dW"] - r["dW"]).max() <= tol(r["dW"]) and np.abs(rp["da"] - r["da"]).max() <= tol(r["da"])
    if Sk is not None:
        assert np.abs(rp["dW_skip"] - r["dW_skip"]).max() <= tol(r["dW_skip"])
