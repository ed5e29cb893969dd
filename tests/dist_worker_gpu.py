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


def main():
    rank, world, port = (int(v) for v in sys.argv[1:4])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pygat_amd as pg
    from pygat_amd.dist import partition_heads
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
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
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
