"""Column-blocked operands (include/pygat_amd.h, pygat_col_blocks; pygat_amd/dist.py "copy-free exchange"): the activation of a
head-parallel hidden level stays as the exchange delivered it -- [world, N, w], block r = rank r's head columns (models.py:32
torch.cat without the concatenating copy) -- and the next level's GEMMs read / write it in place.  Here on one GPU: blocked ==
ordinary for the GEMMs in both product modes and for a whole level (outputs, dX in blocks, dW, da)."""
import os
import subprocess
import sys
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocked(x2, nb):
    """[N, F] -> [nb, N, F / nb] (block b = columns b w .. of every row), contiguous."""
    N, F = x2.shape
    return x2.view(N, nb, F // nb).permute(1, 0, 2).contiguous()


def _ref_bound(A64, B64, A32, B32):
    """|fp32 CPU product - fp64 product|: the price of fp32 on this input (tests/parity.py prices GEMMs at 4 x that)."""
    return float((A32 @ B32).double().sub(A64 @ B64).abs().max())


@pytest.mark.parametrize("mode", ["split-bf16", "fp32-mfma"])
@pytest.mark.parametrize("shape", [(3144, 1024, 2056, 4), (700, 128, 40, 8), (5000, 256, 130, 2), (130, 64, 200, 4)])
def test_blocked_a_plain_and_transposed(mode, shape):
    """C = A B with A [M x K] blocked along K (the projection), and C = A^T B with the stored A [K x M] blocked along M (the
    weight gradient): against the fp64 product, priced like every GEMM test, and against the same call on the ordinary layout."""
    import pygat_amd as pg
    from pygat_amd import ops, _lib
    M, K, N, nb = shape
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g); B = torch.randn(K, N, generator=g); D = torch.randn(M, N, generator=g)
    Ad, Bd, Dd = A.to(dev), B.to(dev), D.to(dev)
    Ab = _blocked(Ad, nb)
    blocks = _lib.ColBlocks(K // nb, M * (K // nb))
    with pg.gemm_mode(mode):
        C1 = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev)
        ops.gemm(False, False, M, N, K, Ab, K // nb, Bd, N, [(N, C1, N)], a_blocks=blocks)
        ops.gemm(False, False, M, N, K, Ad, K, Bd, N, [(N, C2, N)])
        ref = A.double() @ B.double()
        bound = max(1e-5, 4 * _ref_bound(A.double(), B.double(), A, B))
        assert float((C1.cpu().double() - ref).abs().max()) <= bound
        assert float((C1 - C2).abs().max()) <= bound
        # transposed: W = A^T D  (K x N), stored A [M rows x K cols] blocked along its columns
        W1 = torch.empty(K, N, device=dev); W2 = torch.empty(K, N, device=dev)
        ops.gemm(True, False, K, N, M, Ab, K // nb, Dd, N, [(N, W1, N)], a_blocks=blocks)
        ops.gemm(True, False, K, N, M, Ad, K, Dd, N, [(N, W2, N)])
        ref = A.double().t() @ D.double()
        bound = max(1e-5, 4 * _ref_bound(A.double().t(), D.double(), A.t().contiguous(), D))
        assert float((W1.cpu().double() - ref).abs().max()) <= bound
        assert float((W1 - W2).abs().max()) <= bound


@pytest.mark.parametrize("mode", ["split-bf16", "fp32-mfma"])
def test_blocked_c_is_the_reduce_scatter_layout(mode):
    """dX = dWh W^T written as [world, N, w] blocks (+ the accumulate pass of the skip term)."""
    import pygat_amd as pg
    from pygat_amd import ops, _lib
    dev = torch.device("cuda", 0)
    M, K, N, nb = 2100, 96, 256, 4
    g = torch.Generator().manual_seed(6)
    A = torch.randn(M, K, generator=g).to(dev); W = torch.randn(N, K, generator=g).to(dev)
    blocks = _lib.ColBlocks(N // nb, M * (N // nb))
    with pg.gemm_mode(mode):
        Cb = torch.empty(nb, M, N // nb, device=dev); C = torch.empty(M, N, device=dev)
        ops.gemm(False, True, M, N, K, A, K, W, K, [(N, Cb, N // nb)], c_blocks=blocks)
        ops.gemm(False, True, M, N, K, A, K, W, K, [(N, Cb, N // nb)], accumulate=True, split_k=1, c_blocks=blocks)
        ops.gemm(False, True, M, N, K, A, K, W, K, [(N, C, N)])
    got = Cb.permute(1, 0, 2).reshape(M, N)
    ref = 2 * (A.cpu().double() @ W.cpu().double().t())
    bound = max(1e-5, 8 * float(((A.cpu() @ W.cpu().t()).double() * 2 - ref).abs().max()))
    assert float((got.cpu().double() - ref).abs().max()) <= bound
    assert float((got - 2 * C).abs().max()) <= bound


def test_bad_blocks_are_rejected():
    from pygat_amd import ops, _lib
    dev = torch.device("cuda", 0)
    A = torch.zeros(4, 64, 24, device=dev); B = torch.zeros(96, 80, device=dev); C = torch.zeros(64, 80, device=dev)
    with pytest.raises(ValueError):          # 24 floats per block: not a power of two
        ops.gemm(False, False, 64, 80, 96, A, 24, B, 80, [(80, C, 80)], a_blocks=_lib.ColBlocks(24, 64 * 24))
    A = torch.zeros(8, 64, 8, device=dev); B = torch.zeros(64, 80, device=dev)
    with pytest.raises(ValueError):          # 8 floats per block: below a 64-byte sector
        ops.gemm(False, False, 64, 80, 64, A, 8, B, 80, [(80, C, 80)], a_blocks=_lib.ColBlocks(8, 64 * 8))


@pytest.mark.parametrize("cfg", [dict(N=3000, nb=4, w=256, H=2, Fo=64, skip=True, concat=True),     # PPI level 2 / 3 on 4 ranks
                                 dict(N=5000, nb=8, w=16, H=1, Fo=16, skip=False, concat=True),      # one 16-float head per rank of 8
                                 dict(N=2000, nb=2, w=32, H=3, Fo=7, skip=True, concat=False)])      # an output level reading blocks
def test_level_reads_blocks_in_place(cfg):
    """The whole level on a column-blocked input == the level on the concatenated input: outputs, dW, da, dWskip, and dX --
    returned in the blocks' layout."""
    import pygat_amd as pg
    from oracle import gat_oracle as O
    dev = torch.device("cuda", 0)
    N, nb, w, H, Fo = cfg["N"], cfg["nb"], cfg["w"], cfg["H"], cfg["Fo"]
    Fin = nb * w
    rowptr, col = O.random_symmetric_csr(N, 6, 3, hub=(7, min(N - 1, 900)))
    graph = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, Fin, generator=g).to(dev)
    W = (torch.randn(H, Fin, Fo, generator=g) * 0.1).to(dev)
    a = (torch.randn(H, 2 * Fo, generator=g) * 0.3).to(dev)
    S = (torch.randn(H, Fin, Fo, generator=g) * 0.1).to(dev) if cfg["skip"] else None
    G = torch.randn(N, H * Fo if cfg["concat"] else Fo, generator=g).to(dev)

    def run(xin):
        xin = xin.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) if p is not None else None for p in (W, a, S)]
        out = pg.GATLevelFn.apply(xin, ps[0], ps[1], ps[2], graph, 0.2, cfg["concat"])
        out.backward(G)
        return out.detach(), xin.grad, [None if p is None else p.grad for p in ps]
    o1, dx1, g1 = run(x)
    o2, dx2, g2 = run(_blocked(x, nb))
    assert dx2.shape == (nb, N, w)
    scale = lambda t: max(1.0, float(t.abs().max()))     # noqa: E731
    assert float((o1 - o2).abs().max()) <= 2e-5 * scale(o1)
    assert float((dx1 - dx2.permute(1, 0, 2).reshape(N, Fin)).abs().max()) <= 5e-5 * scale(dx1)
    for p, q in zip(g1, g2):
        if p is not None:
            assert float((p - q).abs().max()) <= 5e-5 * scale(p)


def _run_bench(extra, timeout=420):
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "16", "--draws", "300000", "--steps", "3",
                        "--warmup", "1", "--verify"] + extra, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]           # stdout carries exactly ONE line
    assert "max |sharded - unsharded|" in r.stderr
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` as the driver types it, NO launcher: the parent never touches the GPU and starts N child
    ranks (here over gloo, sharing the one card: RCCL refuses two ranks on a device); rank 0's single JSON line comes back
    with n_gpus = N, and --verify compares the exchanged column-blocked activation with the unsharded level."""
    line = _run_bench(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["config"]["heads_per_gpu"] == 4 and line["value"] > 0
    assert "cpu_baseline" not in line


def test_bench_self_launch_one_head_per_rank():
    line = _run_bench(["--gpus", "4", "--heads", "4"])
    assert line["n_gpus"] == 4 and line["config"]["heads_per_gpu"] == 1


def test_bench_self_launch_propagates_a_failing_rank():
    env = dict(os.environ, BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--heads", "3", "--scale", "12", "--draws", "20000",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()       # 3 heads over 2 ranks: unequal shards, every rank refuses
