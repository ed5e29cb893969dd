"""Head-parallel model on the GPU kernels, world_size 2, both ranks on cuda:0 over gloo (a 1-GPU box cannot
host two RCCL ranks): sharded == unsharded for outputs and local-head gradients, uneven shards included."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_head_parallel_model_world2_on_one_card():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker_gpu.py"), str(r), "2", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]


def test_pipelined_hidden_levels_world2_on_one_card():
    """Row-chunk pipeline of the head-parallel hidden levels (K2 per chunk + all-gather of the finished chunk)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker_gpu.py"), str(r), "2", str(port), "pipeline"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]


def test_pipelined_hidden_levels_in_internal_order_world2_on_one_card():
    """The same model in the graph's internal node order (pygat_amd.GAT on large graphs; forced here): x permuted once, every level
    and every exchanged row chunk in the degree order both ranks share, the self-loop-only tail as one more hand-off of the
    pipeline, the final logits put back -- against the unsharded model."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker_gpu.py"), str(r), "2", str(port), "pipeline_internal"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]


def test_rccl_backend_world1_drives_the_collectives():
    """SURVEY.md 7.3 "world_size 1 RCCL smoke on the single GPU": a fresh child initialises the "nccl" backend (RCCL) with
    one rank and runs a 3-level head-parallel model whose collectives are forced on (dist.FORCE_COLLECTIVES): the row-chunk
    pipeline's all_gather_into_tensor(async_op=True), the reduce_scatter_tensor of its backward and the output level's
    all_reduce all execute through RCCL, and the result equals the unsharded model."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker_gpu.py"), "0", "1", str(port), "rccl1"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    try:
        o, _ = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    o = o.decode()
    assert p.returncode == 0 and "rank 0 ok" in o and "rccl world-1: collectives" in o, o[-3000:]
