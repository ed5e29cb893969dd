"""BASELINE.json config 4 on the GPU: the PPI-shaped batch of tests/ppi_case.py through the 3-level
4/4/6-head skip model (reference train_ppi.py:43-55), forward + every gradient against the oracle, the
block-diagonal batching (load_data_ppi.py:71-88) bit for bit, and the same model head-sharded over 2 and 4
ranks (6 heads of the last level -> 2/2/1/1)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import ppi_case as P
from oracle import gat_oracle as O
from parity import check_level, close_grad

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pg():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: run with gpurun")
    import pygat_amd
    return pygat_amd


def _dev_graphs(pg, parts):
    return [pg.CSRGraph(torch.as_tensor(rp, device=DEV), torch.as_tensor(c, device=DEV)) for rp, c in parts]


def test_block_diag_equals_dense_block_diag(pg):
    """CSRGraph.block_diag == the pattern of torch.block_diag of the dense adjacencies (load_data_ppi.py:77-84),
    bit for bit (integer work), and == the K0 intake of that dense matrix."""
    parts = P.graphs()
    gs = _dev_graphs(pg, parts)
    batch = pg.CSRGraph.block_diag(gs)
    dense = torch.block_diag(*[O.dense_from_csr(rp, c, len(rp) - 1) for rp, c in parts])      # [1612, 1612]
    nz = dense.nonzero()                                                                      # row-major = CSR order
    want_rowptr = np.concatenate([[0], np.cumsum(np.bincount(nz[:, 0].numpy(), minlength=dense.shape[0]))]).astype(np.int32)
    want_col = nz[:, 1].numpy().astype(np.int32)
    assert batch.n == sum(P.SIZES) and batch.nnz == len(want_col)
    assert np.array_equal(batch.fwd.rowptr.cpu().numpy(), want_rowptr)
    assert np.array_equal(batch.fwd.col.cpu().numpy(), want_col)
    assert batch.symmetric
    rp_h, col_h = P.batch_csr(parts)
    assert np.array_equal(rp_h, want_rowptr) and np.array_equal(col_h, want_col)
    via_dense = pg.CSRGraph.from_dense(dense.to(DEV), "nonzero")
    assert torch.equal(via_dense.fwd.rowptr, batch.fwd.rowptr) and torch.equal(via_dense.fwd.col, batch.fwd.col)
    assert torch.equal(via_dense.perm_t, batch.perm_t)
    # no edge crosses a graph border
    src = torch.repeat_interleave(torch.arange(batch.n, device=DEV), (batch.fwd.rowptr[1:] - batch.fwd.rowptr[:-1]).long())
    assert bool(((src < P.SIZES[0]) == (batch.fwd.col < P.SIZES[0])).all())


def test_batched_level_equals_per_graph_levels(pg):
    """A level over the batch == the same level over each graph on its own rows (the blocks do not interact)."""
    parts = P.graphs()
    gs = _dev_graphs(pg, parts)
    batch = pg.CSRGraph.block_diag(gs)
    x = torch.as_tensor(P.features(), device=DEV)
    torch.manual_seed(3)
    layer = [pg.SpGraphAttentionLayer(50, 256, 0.0, 0.2, True, True).to(DEV) for _ in range(4)]
    args = ([l.W for l in layer], [l.a for l in layer], [l.skip_projection for l in layer], 0.2, True)
    whole = pg.gat_level(x, batch, *args).detach()
    o = 0
    for g in gs:
        alone = pg.gat_level(x[o:o + g.n].contiguous(), g, *args).detach()
        # same arithmetic, another slot partition (summation order of rows cut by a slot border): rounding only
        assert float((whole[o:o + g.n] - alone).abs().max()) <= 2e-6 * max(1.0, float(alone.abs().max()))
        o += g.n


def test_ppi_levels_forward_backward_vs_oracle(pg):
    """Each of the three levels at its real shape (50 -> 256 x 4 +skip, 1024 -> 256 x 4 +skip, 1024 -> 121 x 6 mean
    +skip), fed with the fp64 oracle's activations of the level below: forward and dX, dW, da, dW_skip against the
    fp64 oracle under the SURVEY 8(c) rule, LeakyReLU kinks accounted for (tests/parity.py)."""
    parts = P.graphs()
    rowptr, col = P.batch_csr(parts)
    batch = pg.CSRGraph.block_diag(_dev_graphs(pg, parts))
    torch.manual_seed(11)
    model = pg.GAT(P.NFEAT, P.NHEADS, 3, 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=True)
    levels = P.oracle_levels(model, torch.float64)
    rng = np.random.default_rng(7)
    x = P.features().astype(np.float64)
    for li, lv in enumerate(levels):
        concat = li < 2
        H, Fin, Fo = lv["W"].shape
        W, a, Sk = (lv[k].detach().numpy() for k in ("W", "a", "skip"))
        G = rng.standard_normal((batch.n, H * Fo if concat else Fo))
        xd = torch.as_tensor(x, dtype=torch.float32, device=DEV).requires_grad_(True)
        Wd, ad, Sd = (torch.as_tensor(v, dtype=torch.float32, device=DEV).requires_grad_(True) for v in (W, a, Sk))
        out = pg.GATLevelFn.apply(xd, Wd, ad, Sd, batch, 0.2, concat)
        out.backward(torch.as_tensor(G, dtype=torch.float32, device=DEV))
        torch.cuda.synchronize()
        rep = check_level(out, {"dX": xd.grad, "dW": Wd.grad, "da": ad.grad, "dW_skip": Sd.grad}, x, rowptr, col, W, a,
                          0.2, concat, G, Sk, what=f"ppi level {li + 1}")
        x = rep["ref64"]["out"]


def test_ppi_model_forward_backward_vs_oracle(pg):
    """The whole model (models.py:29-35 chaining: concat, concat, mean; skip projections; state_dict naming): logits
    under the 8(c) rule, and the END-TO-END gradients under the same flip-aware rule as a single level
    (parity.close_model_grads): the near-kink edges of every level are screened in the fp64 oracle
    (oracle.model_logits_z, band parity.KINK_TAU), the effect of a branch flip at one of them on every parameter of
    the levels below comes from the oracle run with that branch overridden, and what is left after the (bounded,
    in-band) flips must be within max(1e-5, 4 x fp32-oracle error).  No magnitude-scaled escape."""
    from parity import KINK_TAU, close_model_grads
    parts = P.graphs()
    rowptr, col = P.batch_csr(parts)
    batch = pg.CSRGraph.block_diag(_dev_graphs(pg, parts))
    x_h = P.features()
    G_h = np.random.default_rng(7).standard_normal((batch.n, P.NFEAT[-1])).astype(np.float32)
    torch.manual_seed(11)
    model = pg.GAT(P.NFEAT, P.NHEADS, 3, 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=True).to(DEV)
    assert sorted(k for k, _ in model.named_parameters()) == sorted(
        f"attention_layer_{l}_head_{h}.{n}" for l, H in enumerate(P.NHEADS, 1) for h in range(1, H + 1)
        for n in ("W", "a", "skip_projection"))                                  # models.py:27 naming
    x = torch.as_tensor(x_h, device=DEV).requires_grad_(True)
    y = model(x, batch)
    y.backward(torch.as_tensor(G_h, device=DEV))
    torch.cuda.synchronize()
    y64, dx64, g64 = P.oracle_run(model, x_h, rowptr, col, G_h, torch.float64)
    y32, dx32, g32 = P.oracle_run(model, x_h, rowptr, col, G_h, torch.float32)
    assert y.shape == (batch.n, 121)
    # three stacked levels on features up to 23: the fp32 oracle itself sits 8e-6 from fp64 here
    e, own = close_grad(y, y64, y32, "logits", factor=2.0)
    assert e <= 2e-5, e
    print(f"ppi logits: err {e:.2e} (fp32 oracle {own:.2e}), max |y| {float(y64.abs().max()):.3g}")

    levels64 = P.oracle_levels(model, torch.float64)
    kinks = []
    for lv, (z, zs) in enumerate(O.model_logits_z(torch.as_tensor(x_h, dtype=torch.float64), (rowptr, col),
                                                  [{k: (None if v is None else v.detach()) for k, v in l.items()} for l in levels64], 0.2)):
        rel = (z.abs() / zs.clamp(min=1e-300)).numpy()
        for h, ed in zip(*np.nonzero(rel <= KINK_TAU)):
            kinks.append((lv, int(h), int(ed), float(rel[h, ed])))

    def oracle_grads(dtype, flips):
        _, dx, g = P.oracle_run(model, x_h, rowptr, col, G_h, dtype, flips=flips)
        return dict(g, dX=dx)
    oracle_grads.flip_shapes = [(H, len(col)) for H in P.NHEADS]
    got = {name: p.grad for name, p in model.named_parameters()}
    got["dX"] = x.grad
    rep = close_model_grads(got, oracle_grads, kinks, what="ppi model")
    worst = max(rep["hip"], key=lambda n: rep["hip"][n] / max(1e-5, 4 * rep["fp32"][n]))
    print(f"{rep['flips']}; worst tensor {worst}: err {rep['hip'][worst]:.2e} (raw {rep['hip_raw'][worst]:.2e}; fp32 oracle "
          f"{rep['fp32'][worst]:.2e}, raw {rep['fp32_raw'][worst]:.2e})")


def _spawn(world, mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker_gpu.py"), str(r), str(world), str(port), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]


@pytest.mark.parametrize("world", [2, 4])
def test_ppi_model_head_sharded_on_one_card(world):
    """6 heads of the last level over 4 ranks -> 2/2/1/1; all ranks share cuda:0 and exchange over gloo (a 1-GPU
    box cannot host several RCCL ranks).  At most 4 worker processes + pytest touch the card (limit 6)."""
    _spawn(world, "ppi")
