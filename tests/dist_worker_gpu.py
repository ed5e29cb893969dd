"""Worker of tests/test_gpu_dist.py: one rank of a world_size-2 run whose ranks SHARE cuda:0 (gloo moves the
tensors; RCCL refuses two ranks on one device).  The HIP level function runs for real; the sharded model
must reproduce the unsharded one."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gat_oracle as O  # noqa: E402


def ppi(rank, world, pg, dev):
    """BASELINE.json config 4: the PPI-shaped batch (tests/ppi_case.py) through the 4/4/6-head skip model with the
    heads sharded over `world` ranks (last level: 6 heads -> 2/2/1/1 on 4 ranks), against the unsharded model on
    the same card; then one optimiser step + sync_head_parameters() and the state_dicts must agree."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ppi_case as P
    from pygat_amd.dist import partition_heads
    parts = P.graphs()
    batch = pg.CSRGraph.block_diag([pg.CSRGraph(torch.as_tensor(rp, device=dev), torch.as_tensor(c, device=dev))
                                    for rp, c in parts])
    if world == 4:
        assert partition_heads(6, 4) == [(0, 2), (2, 4), (4, 5), (5, 6)]
    torch.manual_seed(21)
    kw = dict(nfeat=P.NFEAT, nheads=P.NHEADS, nlayers=3, dropout=0.0, alpha=0.2, layer_type=pg.SpGraphAttentionLayer,
              skip_connection=True)
    sharded = pg.GAT(head_parallel=True, **kw).to(dev)
    plain = pg.GAT(**kw).to(dev)
    plain.load_state_dict(sharded.state_dict())
    x = torch.as_tensor(P.features(), device=dev)
    G = torch.randn(batch.n, P.NFEAT[-1], generator=torch.Generator().manual_seed(4)).to(dev)
    # plain SGD: the step is linear in the gradient, so rounding differences between the sharded and the unsharded
    # gradients stay rounding differences in the parameters (Adam's first step is lr * g / (|g| + 1e-8): sign-like)
    opts = [torch.optim.SGD(m.parameters(), lr=1e-3) for m in (sharded, plain)]
    y = sharded(x, batch); y.backward(G)
    yr = plain(x, batch); yr.backward(G)
    assert float((y - yr).abs().max()) < 2e-5, float((y - yr).abs().max())
    ps, pr = dict(sharded.named_parameters()), dict(plain.named_parameters())
    for lvl, H in enumerate(P.NHEADS, start=1):
        s, e = partition_heads(H, world)[rank]
        for h in range(H):
            for nm in ("W", "a", "skip_projection"):
                key = f"attention_layer_{lvl}_head_{h + 1}.{nm}"
                if s <= h < e:
                    scale = max(1.0, float(pr[key].grad.abs().max()))
                    assert float((ps[key].grad - pr[key].grad).abs().max()) < 5e-5 * scale, key
                else:
                    assert ps[key].grad is None, key
    for o in opts:
        o.step()
    sharded.sync_head_parameters()
    sd, ref = sharded.state_dict(), plain.state_dict()
    assert list(sd) == list(ref)
    for k in sd:
        assert float((sd[k] - ref[k]).abs().max()) < 1e-5, (k, float((sd[k] - ref[k]).abs().max()))   # steps up to ~1 in size
        t = sd[k].clone()
        dist.broadcast(t, src=0)
        assert torch.equal(t, sd[k]), k            # identical bytes on every rank after the sync


def pipeline(rank, world, pg, dev, internal=False):
    """The copy-free, row-chunk pipelined hidden level of pygat_amd/dist.py (K2 chunk by chunk into the rank's block of the
    column-blocked activation, each chunk exchanged while the next is computed; the next level reads the blocks in place
    and writes its input gradient in the same blocks) against the unsharded model: outputs, gradients of the local heads,
    reduce-scatter of the gradient blocks into the previous level."""
    import pygat_amd.dist as D
    from pygat_amd.dist import partition_heads
    D.PIPELINE_MIN_ROWS, D.PIPELINE_CHUNKS = 0, 3
    N = 9000
    rowptr, col = O.random_symmetric_csr(N, 7, 5, hub=(4000, 3000))
    if internal:
        # the model-level internal node order (pygat_amd.GAT on a large graph), forced: a third of the nodes lose their edges
        # (self loop only: the tail streams), every level and every exchange in the degree order all ranks share
        import numpy as np
        rowptr, col = np.asarray(rowptr, dtype=np.int64), np.asarray(col, dtype=np.int64)
        rows = np.repeat(np.arange(N), np.diff(rowptr))
        iso = (np.arange(N) % 3) == 1
        keep = (rows == col) | (~iso[rows] & ~iso[col])
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=N))]).astype(np.int32)
        col = col[keep].astype(np.int32)
        pg.ops.RENUMBER_MIN_BYTES = pg.ops.RENUMBER_MIN_BYTES_TAIL = 0
    else:
        pg.ops.RENUMBER = False
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev), slot_edges=32)
    nfeat, nheads = [12, 16, 16, 5], [4, 2, 3]         # levels 1 and 2 shard evenly over 2 ranks (blocks of 32 / 16 floats) -> copy-free + pipelined
    torch.manual_seed(0)
    sharded = pg.GAT(nfeat, nheads, 3, 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=True, head_parallel=True).to(dev)
    plain = pg.GAT(nfeat, nheads, 3, 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=True).to(dev)
    plain.load_state_dict(sharded.state_dict())
    calls = []
    orig = D._blocked_concat_level
    D._blocked_concat_level = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    copies = []
    orig_unblock = D.unblock
    D.unblock = lambda *a, **k: (copies.append(1), orig_unblock(*a, **k))[1]
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, nfeat[0], generator=gen).to(dev)
    G = torch.randn(N, nfeat[-1], generator=gen).to(dev)
    y = sharded(x, g); y.backward(G)
    yr = plain(x, g); yr.backward(G)
    assert len(calls) == 2, calls                        # both hidden levels took the copy-free pipelined path
    assert not copies, copies                            # ... and nobody asked for the concatenated layout
    assert float((y - yr).abs().max()) < 2e-5, float((y - yr).abs().max())
    ps, pr = dict(sharded.named_parameters()), dict(plain.named_parameters())
    for lvl, H in enumerate(nheads, start=1):
        s, e = partition_heads(H, world)[rank]
        for h in range(s, e):
            for nm in ("W", "a", "skip_projection"):
                key = f"attention_layer_{lvl}_head_{h + 1}.{nm}"
                scale = max(1.0, float(pr[key].grad.abs().max()))
                assert float((ps[key].grad - pr[key].grad).abs().max()) < 5e-5 * scale, key


def rccl_world1(pg, dev):
    """ONE rank on the "nccl" backend (= RCCL; a 1-GPU box hosts exactly one) with dist.FORCE_COLLECTIVES: the pipelined
    hidden level issues its per-chunk exchange on RCCL's stream (with one rank: the in-place all_gather_into_tensor of the
    chunk's view of the column-blocked activation; with peers: one grouped send/recv), its backward reduce_scatter_tensor
    of the gradient blocks, the output level all_reduce.  With one rank every collective is the identity, so the model must
    equal the unsharded one."""
    import pygat_amd.dist as D
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    D.FORCE_COLLECTIVES = True
    N = 1 << 16                                   # >= dist.PIPELINE_MIN_ROWS: the row-chunk pipeline runs
    assert N >= D.PIPELINE_MIN_ROWS and D.PIPELINE_CHUNKS > 1
    from pygat_amd.rmat import rmat_csr
    rowptr, col = rmat_csr(16, 300_000, seed=4, device=dev)
    g = pg.CSRGraph(rowptr, col)
    torch.manual_seed(2)
    kw = dict(nfeat=[32, 16, 16, 5], nheads=[4, 4, 3], nlayers=3, dropout=0.0, alpha=0.2, layer_type=pg.SpGraphAttentionLayer)
    sharded = pg.GAT(head_parallel=True, **kw).to(dev)
    plain = pg.GAT(**kw).to(dev)
    plain.load_state_dict(sharded.state_dict())
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(N, 32, generator=gen).to(dev); G = torch.randn(N, 5, generator=gen).to(dev)
    calls = {"ag": 0, "rs": 0, "ar": 0}
    real = (dist.all_gather_into_tensor, dist.reduce_scatter_tensor, dist.all_reduce)

    def counted(name, fn):
        def f(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return f
    dist.all_gather_into_tensor = counted("ag", real[0]); dist.reduce_scatter_tensor = counted("rs", real[1])
    dist.all_reduce = counted("ar", real[2])
    try:
        y = sharded(x, g); y.backward(G)
    finally:
        dist.all_gather_into_tensor, dist.reduce_scatter_tensor, dist.all_reduce = real
    torch.cuda.synchronize()
    yr = plain(x, g); yr.backward(G)
    torch.cuda.synchronize()
    assert calls["ag"] == 2 * D.PIPELINE_CHUNKS and calls["rs"] == 2 and calls["ar"] >= 1, calls    # two hidden levels, one output level
    assert float((y - yr).abs().max()) <= 1e-6 * max(1.0, float(yr.abs().max())), float((y - yr).abs().max())
    worst = 0.0
    for (k, p), (_, q) in zip(sharded.named_parameters(), plain.named_parameters()):
        # (the sharded model's levels 2 / 3 read the column-blocked activation through the general GEMM kernels, the plain one
        # through the streamed fast paths: other fp32 summation orders -- and with 2.5 M logits per level a handful sit within
        # rounding of the LeakyReLU kink and take the other branch, each moving a gradient by ~1e-4 of its maximum)
        assert p.grad is not None, k
        rel = float((p.grad - q.grad).abs().max()) / max(1.0, float(q.grad.abs().max()))
        worst = max(worst, rel)
        assert rel <= 5e-4, (k, rel)
    D.FORCE_COLLECTIVES = False
    print("rccl world-1: collectives", calls, f"worst relative gradient difference {worst:.2e}")


def main():
    rank, world, port = (int(v) for v in sys.argv[1:4])
    mode = sys.argv[4] if len(sys.argv) > 4 else "small"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "rccl1":
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import pygat_amd as pg
        rccl_world1(pg, dev)
        torch.cuda.synchronize()
        dist.destroy_process_group()
        print("rank 0 ok")
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pygat_amd as pg
    from pygat_amd.dist import partition_heads
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if mode in ("ppi", "pipeline", "pipeline_internal"):
        if mode == "ppi":
            ppi(rank, world, pg, dev)
        else:
            pipeline(rank, world, pg, dev, internal=(mode == "pipeline_internal"))
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        print(f"rank {rank} ok")
        return
    N = 300
    rowptr, col = O.random_symmetric_csr(N, 6, 1, hub=(2, 150))
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
    nfeat, nheads = [10, 16, 8, 7], [3, 4, 5]          # 3 and 5 heads over 2 ranks: uneven shards
    torch.manual_seed(0)                                # identical replicas on both ranks
    sharded = pg.GAT(nfeat, nheads, 3, 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=True, head_parallel=True).to(dev)
    plain = pg.GAT(nfeat, nheads, 3, 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=True).to(dev)
    plain.load_state_dict(sharded.state_dict())
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, nfeat[0], generator=gen).to(dev)
    G = torch.randn(N, nfeat[-1], generator=gen).to(dev)
    y = sharded(x, g); y.backward(G)
    yr = plain(x, g); yr.backward(G)
    assert float((y - yr).abs().max()) < 2e-5, float((y - yr).abs().max())
    ps, pr = dict(sharded.named_parameters()), dict(plain.named_parameters())
    for lvl, H in enumerate(nheads, start=1):
        s, e = partition_heads(H, world)[rank]
        for h in range(H):
            for nm in ("W", "a", "skip_projection"):
                key = f"attention_layer_{lvl}_head_{h + 1}.{nm}"
                if s <= h < e:
                    scale = max(1.0, float(pr[key].grad.abs().max()))
                    assert float((ps[key].grad - pr[key].grad).abs().max()) < 5e-5 * scale, key
                else:
                    assert ps[key].grad is None, key
    # train mode with dropout: runs, finite, ranks agree on the replicated output
    drop = pg.GAT(nfeat, nheads, 3, 0.5, 0.2, pg.SpGraphAttentionLayer, head_parallel=True).to(dev).train()
    yd = drop(x, g)
    yd.sum().backward()
    assert torch.isfinite(yd).all()
    t = yd.detach().clone()
    dist.broadcast(t, src=0)
    assert torch.equal(t, yd.detach())                  # all-reduce result identical on every rank
    # SpGraphAttentionLayerV2 levels shard the same way (layers.py:234-316 heads are independent too)
    torch.manual_seed(3)
    v2s = pg.GAT([10, 8, 5], [3, 2], 2, 0.0, 0.2, pg.SpGraphAttentionLayerV2, skip_connection=True, head_parallel=True).to(dev)
    v2p = pg.GAT([10, 8, 5], [3, 2], 2, 0.0, 0.2, pg.SpGraphAttentionLayerV2, skip_connection=True).to(dev)
    v2p.load_state_dict(v2s.state_dict())
    G2 = torch.randn(N, 5, generator=gen).to(dev)
    y2 = v2s(x, g); y2.backward(G2)
    y2r = v2p(x, g); y2r.backward(G2)
    assert float((y2 - y2r).abs().max()) < 2e-5
    s0, e0 = partition_heads(3, world)[rank]
    for h in range(s0, e0):
        key = f"attention_layer_1_head_{h + 1}.W"
        gs_, gr_ = dict(v2s.named_parameters())[key].grad, dict(v2p.named_parameters())[key].grad
        assert float((gs_ - gr_).abs().max()) < 5e-5 * max(1.0, float(gr_.abs().max())), key
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
