"""Head-parallel sharding algebra on CPU: world_size 2 over gloo, the HIP level swapped for the
oracle (pygat_amd.dist takes `level_fn`), checked against the unsharded oracle result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(world, H2, H1):
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")       # world processes on the container's 8 cores
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(H2), str(H1)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in out, out[-2000:]


@pytest.mark.parametrize("H2", [2, 1])        # H2 = 1: rank 1 owns no head of the last level
def test_head_parallel_world2_gloo(H2):
    _run_world(2, H2, 3)


def test_head_parallel_world8_gloo():
    """The shape of the driver's 8-GPU run (bench.py --gpus 8: 8 heads, ONE per rank) plus an uneven last level (6 heads on
    8 ranks: two ranks own none and still take part in every collective): all-gather / reduce-scatter / all-reduce algebra
    and sync_head_parameters over 8 gloo ranks against the unsharded oracle (the N = 8 path had only ever
    been rehearsed at 2 and 4 ranks)."""
    _run_world(8, 6, 8)


def test_partition_heads():
    from pygat_amd.dist import partition_heads
    assert partition_heads(8, 8) == [(i, i + 1) for i in range(8)]
    assert partition_heads(6, 4) == [(0, 2), (2, 4), (4, 5), (5, 6)]     # PPI last level on 4 ranks: 2/2/1/1
    assert partition_heads(1, 2) == [(0, 1), (1, 1)]
    for H in range(1, 12):
        for w in range(1, 9):
            p = partition_heads(H, w)
            assert p[0][0] == 0 and p[-1][1] == H and all(a[1] == b[0] for a, b in zip(p, p[1:]))


def test_rmat_generator_is_deterministic_and_well_formed():
    from pygat_amd.rmat import rmat_csr
    rp, c = rmat_csr(scale=10, n_draws=5000, seed=1)
    rp2, c2 = rmat_csr(scale=10, n_draws=5000, seed=1)
    assert torch.equal(rp, rp2) and torch.equal(c, c2)
    n = 1 << 10
    rp, c = rp.numpy().astype(np.int64), c.numpy().astype(np.int64)
    src = np.repeat(np.arange(n), np.diff(rp))
    assert (np.diff(rp) >= 1).all()                                   # self loops: no empty row
    keys = set(zip(src.tolist(), c.tolist()))
    assert len(keys) == len(c) and all((j, i) in keys for i, j in list(keys)[:2000])   # dedup + symmetric
    assert all(np.all(np.diff(c[rp[i]:rp[i + 1]]) > 0) for i in range(0, n, 37))       # sorted rows


def test_bench_legs_draw_the_same_graph():
    """bench.py's GPU leg (pygat_amd.rmat.rmat_csr_numpy, uploaded) and its cpu_baseline leg (oracle/cpu_bench.py, a process
    without torch that carries its own copy of the generator) must time the identical graph."""
    import numpy as np
    from pygat_amd.rmat import rmat_csr_numpy
    from oracle.cpu_bench import rmat_csr_numpy as cpu_leg
    for scale, draws in ((10, 5000), (13, 40000)):
        a, b = rmat_csr_numpy(scale, draws), cpu_leg(scale, draws)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        rp, c = a
        assert rp[0] == 0 and rp[-1] == len(c) and (np.diff(rp) >= 1).all()      # self loops: no empty row
