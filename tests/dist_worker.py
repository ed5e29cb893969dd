"""Worker of tests/test_dist_cpu.py: one rank of the world_size-2 gloo run (CPU, oracle as level_fn)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gat_oracle as O  # noqa: E402


def _oracle_level(x, graph, Ws, As, Sk, alpha, concat):
    W = torch.stack(list(Ws)); a = torch.stack([p.reshape(-1) for p in As])
    S = torch.stack(list(Sk)) if Sk is not None else None
    return O.level_forward(x, graph, W, a, alpha, concat, S, "sparse")


def model_step_and_sync(rank, world):
    """pygat_amd.GAT(head_parallel=True): one optimiser step moves only the owned heads; after
    sync_head_parameters() every rank's state_dict equals the unsharded model's after the same step
    (the reference checkpoints model.state_dict() every epoch, train.py:201,233)."""
    import pygat_amd as pg
    N = 30
    graph = O.random_symmetric_csr(N, 4, 2)
    torch.manual_seed(5)
    kw = dict(nfeat=[5, 4, 3], nheads=[3, 3], nlayers=2, dropout=0.0, alpha=0.2, layer_type=pg.SpGraphAttentionLayer,
              skip_connection=True, level_fn=_oracle_level)
    sharded = pg.GAT(head_parallel=True, **kw).double()
    plain = pg.GAT(**kw).double()
    plain.load_state_dict(sharded.state_dict())
    x = torch.randn(N, 5, dtype=torch.float64); G = torch.randn(N, 3, dtype=torch.float64)
    before = {k: v.clone() for k, v in sharded.state_dict().items()}
    for m in (sharded, plain):
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
        m(x, graph).backward(G)
        opt.step()
    sd, ref = sharded.state_dict(), plain.state_dict()
    stale = [k for k in sd if torch.equal(sd[k], before[k])]
    assert stale, "with 3 heads on 2 ranks every rank has heads it does not own"
    assert any(not torch.allclose(sd[k], ref[k], atol=1e-12) for k in stale)      # un-synced replicas differ
    sharded.sync_head_parameters()
    sd = sharded.state_dict()
    assert list(sd) == list(ref)
    for k in sd:
        assert torch.allclose(sd[k], ref[k], atol=1e-12), k
    # and every rank holds the same bytes
    for k in sd:
        t = sd[k].clone()
        dist.broadcast(t, src=0)
        assert torch.equal(t, sd[k]), k


def main():
    rank, world, port, H2 = (int(v) for v in sys.argv[1:5])
    H1_arg = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pygat_amd.dist import gat_level_head_parallel, partition_heads
    torch.manual_seed(0)                      # identical replicas of every parameter
    N, Fin, F1, H1, C = 40, 6, 4, H1_arg, 5
    rowptr, col = O.random_symmetric_csr(N, 4, 1)
    graph = (rowptr, col)
    x = torch.randn(N, Fin, dtype=torch.float64)
    mk = lambda *s: torch.randn(*s, dtype=torch.float64, requires_grad=True)  # noqa: E731
    W1 = [mk(Fin, F1) for _ in range(H1)]; a1 = [mk(1, 2 * F1) for _ in range(H1)]
    W2 = [mk(H1 * F1, C) for _ in range(H2)]; a2 = [mk(2 * C, 1) for _ in range(H2)]
    S2 = [mk(H1 * F1, C) for _ in range(H2)]
    G = torch.randn(N, C, dtype=torch.float64)
    # sharded: level 1 concat (all-gather / reduce-scatter), level 2 mean (all-reduce)
    h = gat_level_head_parallel(x, graph, W1, a1, None, 0.2, True, level_fn=_oracle_level)
    y = gat_level_head_parallel(h, graph, W2, a2, S2, 0.2, False, level_fn=_oracle_level)
    y.backward(G)
    # unsharded reference on the same parameters
    cl = lambda L: [w.detach().clone().requires_grad_(True) for w in L]  # noqa: E731
    W1r, a1r, W2r, a2r, S2r = cl(W1), cl(a1), cl(W2), cl(a2), cl(S2)
    yr = _oracle_level(_oracle_level(x, graph, W1r, a1r, None, 0.2, True), graph, W2r, a2r, S2r, 0.2, False)
    yr.backward(G)
    assert torch.allclose(y, yr, atol=1e-12), "sharded output differs"
    for H, mine, ref in ((H1, W1, W1r), (H1, a1, a1r), (H2, W2, W2r), (H2, a2, a2r), (H2, S2, S2r)):
        s, e = partition_heads(H, world)[rank]
        for k in range(H):
            if s <= k < e:      # gradients of the local heads are complete and equal the unsharded ones
                assert torch.allclose(mine[k].grad, ref[k].grad, atol=1e-11), f"grad of local head {k}"
            else:               # non-local heads are never touched on this rank
                assert mine[k].grad is None
    model_step_and_sync(rank, world)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
