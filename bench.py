#!/usr/bin/env python3
"""Headline benchmark: one GAT level forward+backward on a synthetic R-MAT graph.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json config 5 / SURVEY.md 8(d)): R-MAT scale 20 (N = 1,048,576 nodes),
5,000,000 edge draws with (a,b,c,d) = (0.57,0.19,0.19,0.05), seed 1, symmetrised, de-duplicated,
self loops added (E ~ 10.76 M); X ~ N(0,1) [N,128]; 8 heads x F' = 16 (H*F' = 128); concat + ELU;
eval-mode semantics (dropout 0); fp32.  A step = forward + backward of the level (dW, da; the
input of a first level carries no gradient in the reference, train.py:132,158; --dx adds dX).

The step launches the level's kernels one by one on the current stream (pygat_amd.GATLevelFn, the op the
drop-in model calls).  --hip-graph replays the level from two captured HIP graphs instead
(pygat_amd.GraphedLevel): that wins on the reference's small graphs, where launches dominate (epoch_ms
below is measured that way), and measured 5 % SLOWER here, where each kernel runs for ~1 ms.

N GPUs (`python bench.py --gpus N` starts the N ranks itself; under torch.distributed.run it is one of them): heads
sharded head-per-GPU (pygat_amd/dist.py): each rank projects and attends its H/N heads, the head outputs are
exchanged over xGMI (RCCL, copy-free: straight into the column-blocked activation the next level reads,
models.py:32) while the rank's later row chunks and its backward run; parameter gradients stay local.  Total work
is fixed -> "scaling": "strong".
(--forward-exchange replicate is a labelled experiment: no collective, every rank recomputes all
heads' forward.)

One JSON line on rank 0: value = E / step time (max over ranks, K steps between barriers), plus
  roofline      the kernel with the largest share of the step (HIP events around every launch in an
                instrumented pass of the same steps right after the timed region, so that the timed
                steps carry no event records; SURVEY.md 8(d) algorithmic bytes / flops);
  kernels       the same figures for every kernel of the step;
  cpu_baseline  oracle/gat_oracle.c (CPU port of the same level) on this host's cores, N=1 only;
  epoch_ms      Cora / Pubmed / PPI-shaped epochs (train.py:151-179, train_ppi.py) replayed from one HIP graph, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--draws", type=int, default=5_000_000)
    ap.add_argument("--fin", type=int, default=128)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--fout", type=int, default=16)
    ap.add_argument("--dx", action="store_true", help="also back-propagate into the input features")
    ap.add_argument("--hip-graph", action="store_true", help="replay the level from captured HIP graphs (pygat_amd.GraphedLevel)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-dense-cpu", action="store_true", help="skip the dense-formulation CPU figure inside the cpu_baseline leg (~15 s)")
    ap.add_argument("--no-epoch", action="store_true", help="skip the Cora / Pubmed epoch_ms leg")
    ap.add_argument("--no-v2", action="store_true", help="skip the GATv2 level leg")
    ap.add_argument("--no-alt", action="store_true", help="skip the two `alt` legs (fp32-MFMA GEMMs; caller's node order): a profiler run "
                                                          "of the headline mode alone, whose per-kernel averages then mix no other mode in")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed steps of the cpu_baseline leg (>= 1; its value is their median)")
    ap.add_argument("--verify", action="store_true", help="check the gathered sharded output against the unsharded level")
    ap.add_argument("--forward-exchange", choices=["allgather", "replicate"], default="allgather",
                    help="N>1, how every rank gets the other ranks' head outputs: exchanged over xGMI through RCCL (default, the "
                         "design of pygat_amd/dist.py: see --chunks), or -- experiment -- no collective: run the forward of ALL heads "
                         "and back-propagate only the own ones (GATLevelFn bwd_heads)")
    ap.add_argument("--chunks", type=int, default=4,
                    help="N>1: row chunks of the pipelined level -- chunk c's head outputs travel (one grouped RCCL send/recv per chunk, "
                         "straight into the column-blocked activation) while chunk c+1 is computed; 1 = RCCL's in-place all-gather "
                         "of the whole level after its forward (also copy-free), hidden behind the backward only")
    ap.add_argument("--as-rank-of", type=int, default=0,
                    help="single process: run the work of rank 0 of a world of this size (per-rank time model, no collectives)")
    args = ap.parse_args()
    if args.cpu_steps < 1:
        ap.error("--cpu-steps must be >= 1")
    return args


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_child(args):
    """oracle/gat_oracle.c (kind "port") on the host's physical cores, in a process of its own (oracle/cpu_bench.py): started
    from main() BEFORE this process touches the GPU or wakes torch's OpenMP pool; returns the child's JSON record."""
    import subprocess
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        # under rocprofv3 the profiler's library has initialised the GPU already: no child process from here
        return {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "port", "cpu": cpu_model(),
                "sample": "skipped: running under rocprofv3 (no child process after the GPU is initialised)"}
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--scale", str(args.scale), "--draws", str(args.draws),
           "--fin", str(args.fin), "--heads", str(args.heads), "--fout", str(args.fout), "--steps", str(args.cpu_steps)]
    if args.dx:
        cmd.append("--dx")
    env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_") and k != "PYGAT_CPU_BENCH_CHILD"}
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0:
        raise RuntimeError(f"oracle/cpu_bench.py rc={r.returncode}: {r.stderr[-400:]}")
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    if not args.no_dense_cpu:
        # the reference's own DENSE O(N^2) formulation (layers.py:32-64 restated: oracle.gat_oracle.dense_head_forward,
        # stock torch CPU autograd) on this box's host cores -- Pubmed topology, one head 500 -> 8, fwd+bwd; the reference itself
        # measured 3.17 s/step for this case on the build container's 8 cores.  Same pre-GPU discipline: a child.
        try:
            d = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_dense_bench.py"), "--reps", "3"], capture_output=True,
                               text=True, env=env, timeout=600)
            if d.returncode != 0:
                raise RuntimeError(d.stderr[-300:])
            rec["dense_reference_formulation"] = json.loads(d.stdout.strip().splitlines()[-1])
        except Exception as ex:
            rec["dense_reference_formulation"] = {"seconds_per_step": None, "error": repr(ex)}
    return rec


# ---------------------------------------------------------------------------------------------------------
# second half of BASELINE.json's metric: epoch time on the reference's small configurations
# ---------------------------------------------------------------------------------------------------------
EPOCH_CFG = {  # train.py:47-87, train_ppi.py:43-57
    # density: share of non-zero input features of the real dataset (bag of words; Cora 49 216 of 2708 x 1433, Pubmed
    # 988 031 of 19 717 x 500 TF-IDF values): the synthetic features are drawn at that density
    "cora": dict(nheads=[8, 1], nfeats=[1433, 8, 7], dropout=0.6, lr=5e-3, wd=5e-4, ntrain=140, density=0.0127),
    "pubmed": dict(nheads=[8, 8], nfeats=[500, 8, 3], dropout=0.6, lr=1e-2, wd=1e-3, ntrain=60, density=0.1002),
    "ppi": dict(nheads=[4, 4, 6], nfeats=[50, 256, 256, 121], dropout=0.0, lr=5e-3, wd=0.0),
}


def ppi_batch(pg, dev):
    """A PPI-shaped batch (BASELINE.json config 4): two graphs with the node counts of the first two training graphs
    (tests/golden/ppi_graph_sizes.npz; the reference's edge lists are missing blobs), seeded random symmetric edges of
    mean degree 28 plus self loops each, batched block-diagonally (load_data_ppi.py:71-88)."""
    sizes = np.load(os.path.join(ROOT, "tests", "golden", "ppi_graph_sizes.npz"), allow_pickle=False)["train"][:2]
    gen = torch.Generator().manual_seed(100)
    graphs = []
    for n in (int(v) for v in sizes):
        m = n * 14
        r = torch.randint(0, n, (m,), generator=gen)
        c = torch.randint(0, n, (m,), generator=gen)
        graphs.append(pg.CSRGraph.from_edge_index(r.to(dev), c.to(dev), n, symmetrize=True, self_loops=True))
    return pg.CSRGraph.block_diag(graphs)


def epoch_ms(pg, dev, name, epochs=200):
    """One epoch of train.py:151-179 (train step with dropout + eval forward) on the REAL topology
    (tests/golden/<name>_csr.npz) with seeded synthetic features / labels (the reference's feature blobs are
    missing), replayed from ONE HIP graph (pygat_amd.FusedEpoch)."""
    import torch.nn.functional as F
    c = EPOCH_CFG[name]
    g = torch.Generator().manual_seed(72)
    if name == "ppi":
        graph = ppi_batch(pg, dev)
        N, E = graph.n, graph.nnz
        x = torch.randn(N, c["nfeats"][0], generator=g).to(dev)
        y = (torch.rand(N, c["nfeats"][-1], generator=g) < 0.3).float().to(dev)
        loss_fn = pg.BCEWithLogits(y)            # train_ppi.py:114,157 BCEWithLogitsLoss(reduction='mean'): one launch each way
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", f"{name}_csr.npz"), allow_pickle=False)
        rowptr, col = z["rowptr"], z["col"]
        N, E = len(rowptr) - 1, int(len(col))
        x = (torch.rand(N, c["nfeats"][0], generator=g) < c["density"]).float()
        x = (x / x.sum(1, keepdim=True).clamp(min=1)).to(dev)          # utils.normalize_features
        y = torch.randint(0, c["nfeats"][-1], (N,), generator=g).to(dev)
        it = torch.arange(c["ntrain"], device=dev)
        graph = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
        # train.py:151-152,159: nll_loss(log_softmax(elu(out))[idx_train], labels[idx_train]) -- one launch forward, one backward
        loss_fn = pg.EluLogSoftmaxNLL(it, y, N)
    torch.manual_seed(72)
    model = pg.GAT(c["nfeats"], c["nheads"], len(c["nheads"]), c["dropout"], 0.2, pg.SpGraphAttentionLayer,
                   skip_connection=(name == "ppi")).to(dev)
    # train.py:64-66's optim.Adam update as one launch over all parameters (pygat_amd.Adam, csrc/k11_adam.hip; torch's foreach
    # form spends ~36 small launches per step, its fused capturable form two: 15 us of a 0.25 ms Cora epoch)
    opt = pg.Adam(model.parameters(), lr=c["lr"], weight_decay=c["wd"])
    ep = pg.FusedEpoch(model, opt, x, graph, loss_fn)
    for _ in range(10):
        ep.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        ep.run()
    torch.cuda.synchronize()
    shape = f"{c['nfeats'][0]} -> " + " -> ".join(f"{h} x {f}" for h, f in zip(c["nheads"], c["nfeats"][1:]))
    return {"ms": (time.perf_counter() - t0) / epochs * 1e3, "nodes": N, "edges": E,
            "config": f"{shape}{', skip connections' if name == 'ppi' else ''}, dropout {c['dropout']}, Adam (pygat_adam_step), train step + "
                      f"eval forward, one HIP-graph replay per epoch",
            "data": ("two synthetic graphs with real PPI node counts, block-diagonal batch, synthetic features/labels"
                     if name == "ppi" else f"real topology, synthetic row-normalised features at the dataset's density "
                                           f"({c['density']:.2%} non-zero), synthetic labels")}


def gatv2_level_record(pg, ops, graph, X, H, Fo, steps=10):
    """one SpGraphAttentionLayerV2 level (layers.py:234-316) forward + backward on the same graph and
    input, same head count and width: step time and HIP-event spans of its launches against their algorithmic bytes.
    The V2 score a . LeakyReLU(W_l h_i + W_r h_j) needs the gathered node's whole F'-vector and the layer aggregates
    Whi_j, so every edge gathers a 2R-float row [Whi_j | Whj_j] (forward, backward row pass) or [Gp_i | m, 1/Z, D | Whi_i]
    (backward column pass): 2.7 x the bytes of the v1 level per edge."""
    N, E, Fin = graph.n, graph.nnz, X.shape[1]
    dev = X.device
    g = torch.Generator(device=dev).manual_seed(5)
    W = (torch.randn(H, 2 * Fin, Fo, generator=g, device=dev) * (1.414 * (2.0 / (2 * Fin + Fo)) ** 0.5)).requires_grad_(True)
    a = (torch.randn(H, Fo, generator=g, device=dev) * (1.414 * (2.0 / (1 + Fo)) ** 0.5)).requires_grad_(True)
    G = torch.randn(N, H * Fo, generator=g, device=dev)

    def step():
        W.grad = a.grad = None
        pg.GATv2LevelFn.apply(X, W, a, None, graph, 0.2, True, None).backward(G)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    timer = ops.KernelTimer()
    ops.TIMER = timer
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ops.TIMER = None
    R = H * pg.padded_width(Fo)
    model = {
        "v2_project": 4.0 * N * (Fin + 2 * R),
        "v2_forward": E * (8 + 8 * R) + N * (4 + 8 * R + 8 * H),
        "v2_prepare": N * (12 * R + 8 * H + 4 * (2 * R + 4 * H)),
        # column pass: per edge (i, j) pair, the gathered [Gp_i | m,1/Z,D | Whi_i] row, the de record it writes; per node WW_j
        # row-local and dWW_j written.  Row pass: per edge pair, perm, the Whj HALF of WW_j, the de record; per node Whi_i
        # row-local and dWhi_i read-modify-written
        "v2_backward_row_col": E * (8 + 4 * (2 * R + 4 * H) + 4 * H) + N * (16 * R) + E * (12 + 4 * R + 4 * H) + N * (12 * R),
        "v2_wgrad": 4.0 * N * (Fin + 2 * R),
    }
    kernels = []
    for name, v in timer.times_ms().items():
        t = float(np.mean(v))
        b = model.get(name)
        kernels.append({"kernel": name, "avg_ms": t, "algorithmic_bytes": None if b is None else int(b),
                        "frac_of_8TBps": None if b is None else b / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    return {"layer": "SpGraphAttentionLayerV2 (layers.py:234-316)", "ms_per_step": ms, "edges_per_s": E / (ms * 1e-3),
            "heads": H, "f_out": Fo, "kernels": kernels,
            "note": "v2_backward_row_col = column pass (gathers [Gp_i | m,1/Z,D | Whi_i], leaves de per transposed edge) + row pass "
                    "(gathers the Whj half of [Whi_j | Whj_j], de through perm) + their fix-ups, one C call; per-kernel times: "
                    "profiles/ rocprof summary of tools/v2_bench.py"}


def self_launch(args):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes of this one (which never
    touches the GPU: no HIP call before or after), one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would; rank 0's stdout (the one JSON line) is relayed, every rank's stderr is this process's
    stderr; any rank failing ends the others and the exit code is non-zero.  Under torch.distributed.run (WORLD_SIZE set)
    this function is never reached."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks0 = []
    reader = threading.Thread(target=lambda: chunks0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    try:
        pending = list(procs)
        while pending:          # a rank that died takes the job with it: the others would wait in a collective for ever
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    reader.join(timeout=10)
    lines = [ln for ln in b"".join(chunks0).decode().splitlines() if ln.startswith("{")]
    if rc == 0 and not lines:
        print("bench: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    sys.exit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # stdout carries exactly ONE line, the JSON record: native libraries (RCCL prints a version banner on its
    # first collective) get stderr instead
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_record = None
    if world == 1 and not args.no_cpu:
        try:        # the cpu_baseline leg: library rebuilt for this host (-march=native) and the whole leg run in a child
                    # process NOW, before this process touches the GPU (round 3 ran it in-process, after the GPU legs, on
                    # every hardware thread beside torch's own OpenMP pool: 2.37 s/step one round, 4.63 the next)
            from oracle import c_oracle
            c_oracle.build_for_host()
            cpu_record = cpu_baseline_child(args)
        except Exception as ex:  # the GPU number stays valid without the CPU leg
            print(f"bench: cpu_baseline leg failed ({ex!r})", file=sys.stderr)
            cpu_record = {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "port", "cpu": cpu_model(),
                          "sample": f"failed: {ex!r}"}
    if world != args.gpus:
        if rank == 0:
            print(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    # BENCH_BACKEND=gloo is a rehearsal aid only: several ranks may then share one card (local_rank modulo the
    # device count) to exercise the N>1 code path on a 1-GPU box; the driver's runs use RCCL ("nccl").
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import pygat_amd as pg
    from pygat_amd import ops
    from pygat_amd.dist import partition_heads
    from pygat_amd.rmat import rmat_csr_numpy

    H, Fo, Fin = args.heads, args.fout, args.fin
    # ONE graph for both legs: drawn on the host from the numpy stream the cpu_baseline child replays (oracle/cpu_bench.py), uploaded
    rp_h, col_h = rmat_csr_numpy(args.scale, args.draws, seed=1)
    rowptr, col = torch.from_numpy(rp_h).to(dev), torch.from_numpy(col_h).to(dev)
    graph = pg.CSRGraph(rowptr, col)
    N, E = graph.n, graph.nnz
    g2 = torch.Generator(device=dev).manual_seed(2)
    X = torch.randn(N, Fin, generator=g2, device=dev)
    g3 = torch.Generator(device=dev).manual_seed(3)
    W = torch.randn(H, Fin, Fo, generator=g3, device=dev) * (1.414 * (2.0 / (Fin + Fo)) ** 0.5)
    a = torch.randn(H, 2 * Fo, generator=g3, device=dev) * (1.414 * (2.0 / (1 + 2 * Fo)) ** 0.5)
    g4 = torch.Generator(device=dev).manual_seed(4)
    G = torch.randn(N, H * Fo, generator=g4, device=dev)

    model_world = args.as_rank_of if (world == 1 and args.as_rank_of > 1) else world
    parts = partition_heads(H, model_world)
    hs, he = parts[rank]
    h_loc = he - hs
    replicate = model_world > 1 and args.forward_exchange == "replicate"
    use_pg = (world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1") and not replicate
    G_loc = G[:, hs * Fo:he * Fo].contiguous()
    k2_heads = H if replicate else h_loc          # heads the forward kernels process on this rank

    # ---- the step -----------------------------------------------------------------------------------
    if replicate:       # experiment: forward of ALL heads in one call, backward of the own ones (GATLevelFn bwd_heads)
        W_all = W.clone().requires_grad_(True)
        a_all = a.clone().requires_grad_(True)

        def level_fwd():
            W_all.grad = a_all.grad = None
            return pg.GATLevelFn.apply(X, W_all, a_all, None, graph, 0.2, True, (hs, h_loc))

        def level_bwd(out):
            out.backward(G)                   # only the columns of the own heads are read
    elif use_pg or not args.hip_graph:
        W_loc = W[hs:he].contiguous().requires_grad_(True)
        a_loc = a[hs:he].contiguous().requires_grad_(True)
        Xb = X.clone().requires_grad_(True) if args.dx else X

        def level_fwd():
            W_loc.grad = a_loc.grad = None
            if args.dx:
                Xb.grad = None
            return pg.GATLevelFn.apply(Xb, W_loc, a_loc, None, graph, 0.2, True)

        def level_bwd(out):
            out.backward(G_loc)
    else:               # the level captured once, replayed every step
        lvl = pg.GraphedLevel(graph, X, W[hs:he], a[hs:he], None, 0.2, True, need_dx=args.dx)

        def level_fwd():
            return lvl.forward()

        def level_bwd(out):
            lvl.backward(G_loc)

    # N > 1: the level runs row chunk by row chunk (GATLevelFn pipeline): K2 writes the rank's head columns straight into ITS block
    # of the column-blocked activation full_out [world, N, w] (models.py:32 torch.cat, as the next level's GEMMs read it:
    # pygat_amd/dist.py "copy-free exchange"), and as soon as chunk c's launches are enqueued its rows go out to the peers and
    # the peers' rows come in -- one grouped RCCL send/recv launch per chunk, straight between the contiguous views
    # full_out[r, r0:r1]: no staging buffer, no landing copy (rounds 2-4: all-gather into [world, rows, w] chunks + a strided
    # copy of every chunk).  The exchange of chunk c overlaps chunk c + 1; the rest hides behind this level's backward, which
    # does not depend on it, and is joined at the end of the step.
    nchunks = max(1, args.chunks) if use_pg else 0
    works = []
    full_out = None
    to_int = None
    if use_pg:
        from pygat_amd import dist as pgdist
        if ops.RENUMBER and not args.dx:
            # a head-parallel MODEL runs in its graph's internal node order (pygat_amd.GAT: x permuted once, every level and every
            # exchange in that order -- all ranks hold the same one --, only the final [N, C] logits put back): the level of this
            # step is fed the way such a model feeds it, and the exchanged activation stays in internal order, as the next level
            # would read it.  (--verify compares it, un-permuted, with the unsharded level in the caller's order.)
            view = graph.internal_view()
            to_int = view.to_internal.long()
            X_run = X.index_select(0, view.to_user.long()).contiguous()
            G_run = G_loc.index_select(0, view.to_user.long()).contiguous()
            graph_run = view
        else:
            X_run, G_run, graph_run = Xb, G_loc, graph
        w_loc = h_loc * Fo
        if any(b - a != h_loc for a, b in parts) or not pgdist.blocked_width_ok(w_loc):
            raise SystemExit(f"bench: {H} heads x {Fo} over {world} ranks: the exchange needs equal shards of a power-of-two width >= 16 "
                             f"floats (got {[(b - a) * Fo for a, b in parts]})")
        full_out = torch.empty(world, N, w_loc, device=dev)

    def on_chunk(c, r0, r1, out):
        works.extend(pgdist.exchange_blocks(full_out, r0, r1))

    def step():
        if use_pg:
            works.clear()
            W_loc.grad = a_loc.grad = None
            if args.dx:
                Xb.grad = None
            out = pg.GATLevelFn.apply(X_run, W_loc, a_loc, None, graph_run, 0.2, True, None, (nchunks, on_chunk, full_out[rank]))
            out.backward(G_run)
            for wk in works:            # the step ends when every peer's rows have arrived
                wk.wait()
            return full_out
        out = level_fwd()
        level_bwd(out)
        return out

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = dt / args.steps * 1e3
    step_ms = sorted(a_.elapsed_time(b_) for a_, b_ in ev)
    ms_median = step_ms[len(step_ms) // 2]

    # ---- the same K steps with the two streamed GEMMs on the fp32 MFMA pipe (the gemm_mode argument of the C ABI): reported beside
    # the headline so that the effect of the split-bf16 products is on record in every run.  Single GPU, stream launches.
    alt = None
    if world == 1 and pg.get_gemm_mode() == "split-bf16" and not (args.hip_graph and not replicate) and not args.no_alt:
        try:
            pg.set_gemm_mode("fp32-mfma")
            for _ in range(max(2, args.warmup)):
                step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            alt = {"gemm_products": "fp32-mfma", "ms_per_step": (time.perf_counter() - t1) / args.steps * 1e3}
        finally:
            pg.set_gemm_mode(None)
        for _ in range(2):   # (back in the headline mode for the instrumented pass)
            step()
        barrier()

    # ---- the same K steps in the CALLER's node order (ops.RENUMBER off): since round 5 a first level of this size runs its node
    # tables in an internal degree order (x permuted once per feature tensor, outputs / gradients at the caller's rows through a
    # map inside the kernels: pygat_amd/ops.py); the line reports what that layout choice is worth.
    alt_order = None
    if world == 1 and not (args.hip_graph and not replicate) and ops.RENUMBER and not args.no_alt:
        try:
            ops.RENUMBER = False
            for _ in range(max(2, args.warmup)):
                step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            alt_order = {"node_order": "caller's", "ms_per_step": (time.perf_counter() - t1) / args.steps * 1e3}
        finally:
            ops.RENUMBER = True
        for _ in range(2):
            step()
        barrier()

    # ---- instrumented pass: the same steps launched eagerly with HIP events around every kernel ------
    timer = ops.KernelTimer()
    if replicate:
        inst_fwd, inst_bwd = level_fwd, level_bwd
    else:
        W_i = W[hs:he].contiguous().requires_grad_(True)
        a_i = a[hs:he].contiguous().requires_grad_(True)
        X_i = X.clone().requires_grad_(True) if args.dx else X

        def inst_fwd():
            W_i.grad = a_i.grad = None
            if args.dx:
                X_i.grad = None
            return pg.GATLevelFn.apply(X_i, W_i, a_i, None, graph, 0.2, True)

        def inst_bwd(out):
            out.backward(G_loc)
    for _ in range(2):
        inst_bwd(inst_fwd())
    torch.cuda.synchronize()
    ops.TIMER = timer
    for _ in range(min(args.steps, 20)):
        inst_bwd(inst_fwd())
    torch.cuda.synchronize()
    ops.TIMER = None

    if args.verify:
        full = step()
        if rank == 0:
            # the gathered concat output of the sharded run against the unsharded level on this rank
            ref = pg.GATLevelFn.apply(X, W, a, None, graph, 0.2, True)
            got = full if (use_pg or replicate or model_world == 1) else None
            if got is not None and got.dim() == 3:          # column-blocked [world, N, w] -> [N, H F'] for the comparison only
                got = got.permute(1, 0, 2).reshape(N, -1)
                if to_int is not None:                       # ... and out of the internal node order
                    got = got.index_select(0, to_int)
            if got is not None:
                err = float((got - ref).abs().max())
                print(f"bench --verify: max |sharded - unsharded| = {err:.3e} over {tuple(ref.shape)}", file=sys.stderr)
                assert err < 1e-5, err

    if rank == 0:
        kt = {k: float(np.mean(v)) for k, v in timer.times_ms().items()}
        Fp = pg.padded_width(Fo)
        Rf, Hf = k2_heads * Fp, k2_heads            # forward kernels
        Rb, Hb = h_loc * Fp, h_loc                  # backward kernels
        # rows K3a prepares: all of them, or -- internal degree order with the self-loop-only tail streamed by itself and no skip
        # projection (pygat_amd/ops.py TAIL; csrc/k12_tail.hip) -- the rows before the tail: the tail's Gp never goes through GR
        n_k3a, tail_rows = N, 0
        tbytes = N * h_loc * Fp * 4
        renumbered = ops.RENUMBER and world == 1 and not args.dx and tbytes >= ops.RENUMBER_MIN_BYTES
        if ops.RENUMBER and world == 1 and not args.dx and ops.TAIL and graph.symmetric and tbytes >= ops.RENUMBER_MIN_BYTES_TAIL:
            from pygat_amd.graph import slot_edges_for
            t = graph.degree_ordered()[0].fwd.self_loop_tail(slot_edges_for(h_loc * Fp, graph.slot_edges))
            if t is not None and N - t[0] >= ops.TAIL_MIN_SHARE * N:
                n_k3a, tail_rows, renumbered = t[0], N - t[0], True
        # SURVEY.md 8(d) byte / flop models
        model = {
            "k1_project": ("mfma", 2.0 * N * Fin * (Rf + 2 * Hf)),
            "k2_forward": ("hbm", E * (4 + 4 * Hf + 4 * Rf) + N * (4 + 4 * Hf + 4 * Rf + 8 * Hf)),
            # (SURVEY 8(d) predates the row-local backward: K3a also reads the forward's alpha-branch share `aneg` [N, R] for the
            # rows with logits on both sides of the LeakyReLU kink -- 44 % of the rows at config 5, 0.24 GB -- which is the whole of
            # its PMC traffic above this model: 2.15 GB measured = 1.85 + 0.24 + qneg; nothing is read twice)
            "k3a_prepare": ("hbm", n_k3a * (12 * Rb + 28 * Hb)),
            "k3b_row": ("hbm", E * (4 + 4 * Rb + 8 * Hb) + N * (4 + 8 * Rb + 16 * Hb) - N * (12 * Rb + 28 * Hb)),
            "k4_backward_col": ("hbm", E * (8 + 4 * Rb + 8 * Hb) + N * (4 + 8 * Rb + 8 * Hb)),
            "k3c_rowsum": ("hbm", E * (12 + 4 * Hb) + N * 4 * Hb),
            "k5_agrad": ("hbm", N * (4 * Rb + 8 * Hb)),
            # da taken along by the column pass (round 4): what is left is the fold of its per-work-group records [2 R] (one per
            # 8 slots of ~64 edges) and of the rows its fix-up finished -- priced on the records' bytes
            "k5_afold": ("hbm", -(-(-(-E // 64)) // 8) * 8 * Rb),
            "k5_wgrad": ("mfma", 2.0 * N * Fin * (Rb + Hb)),
            "k5_xgrad": ("mfma", 2.0 * N * Fin * Rb),
        }
        traffic, traffic_source = {}, None
        try:   # HBM traffic from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
               # this same command; FETCH_SIZE doubled as the gfx950 guide prescribes): main + fix-up launches
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pmc.get("workload_edges") == E and pmc.get("heads_per_gpu") == h_loc:
                traffic = pmc["traffic_bytes"]
                # NOT measured in this run: counters need their own rocprofv3 --pmc passes; the line says where they are from
                traffic_source = ("replayed from the committed profiles/pmc_latest.json = " + str(pmc.get("source", "?"))
                                  + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, "
                                    "FETCH_SIZE x 2 per the gfx950 guide); not measured in this run")
        except Exception:
            traffic = {}
        # split-bf16 mode: the streamed GEMMs run 9 bf16 MFMAs per fp32 32x32x16 block (288 cycles instead of the fp32
        # pipe's 512) and are priced against the HBM roof on their operand + result bytes; their fp32-equivalent flops
        # stay in the record
        split = pg.get_gemm_mode() == "split-bf16"
        gemm_bytes = {"k1_project": 4.0 * N * (Fin + Rf + Hf), "k5_wgrad": 4.0 * N * (Fin + Rb), "k5_xgrad": 4.0 * N * (Fin + Rb)}
        kernels = []
        for name, t in kt.items():
            bound, work = model.get(name, ("hbm", None))
            if work is None:
                continue
            if bound == "mfma" and split:
                ach = gemm_bytes[name] / (t * 1e-3) / 1e9
                kernels.append({"kernel": name, "bound": "hbm", "avg_ms": t, "achieved": ach, "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes": int(gemm_bytes[name]),
                                "flops": work, "fp32_equivalent_tflops": work / (t * 1e-3) / 1e12,
                                "products": "exact three-way bf16 split of both fp32 operands, 9 v_mfma_f32_32x32x16_bf16 per "
                                            "32x32x16 block, fp32 accumulators",
                                "traffic": traffic.get(name)})
                continue
            if bound == "hbm":
                ach = work / (t * 1e-3) / 1e9
                kernels.append({"kernel": name, "bound": "hbm", "avg_ms": t, "achieved": ach, "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes": int(work),
                                "traffic": traffic.get(name)})
            else:
                ach = work / (t * 1e-3) / 1e12
                kernels.append({"kernel": name, "bound": "mfma", "avg_ms": t, "achieved": ach, "peak": MFMA_F32_PEAK_TF,
                                "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "flops": work,
                                "traffic": traffic.get(name)})
        for r in kernels:
            if r["kernel"] == "k4_backward_col" and "k5_afold" in kt:
                # since round 4 the column pass also takes the attention-vector gradient along (pygat_gat_backward_col with
                # da_part): `frac` above prices the FUSED kernel on K4's own SURVEY 8(d) bytes (every byte the da sums need
                # -- Wh_j, ds_j, dt_j -- is already in them); beside it, the pair (K4 + fold) on the bytes of the two passes
                # it replaces (K4 + the a-gradient stream), comparable with earlier rounds' K4 + k5_agrad
                pair_bytes = model["k4_backward_col"][1] + model["k5_agrad"][1]
                pair_ms = r["avg_ms"] + kt["k5_afold"]
                r["fused"] = {"with": "k5_agrad (da = sum_j [ds_j | dt_j] (x) Wh_j), folded by k5_afold",
                              "pair_algorithmic_bytes": int(pair_bytes), "pair_ms": pair_ms,
                              "pair_frac": pair_bytes / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        dominant = max(kernels, key=lambda r: r["avg_ms"])
        roof = dict(dominant)
        roof["timing"] = "HIP events around each launch, instrumented eager pass of the same steps after the timed region"
        roof["traffic_source"] = traffic_source
        line = {
            "metric": "GAT-layer fwd+bwd edges/sec", "value": E / (ms * 1e-3), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "ms_per_step_median": ms_median,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"RMAT scale {args.scale} ({N} nodes, {E} edges incl. self loops, max degree "
                                   f"{int((rowptr[1:] - rowptr[:-1]).max())}), Fin {Fin}, {H} heads x {Fo}, concat+ELU, "
                                   f"dropout 0, fwd+bwd (dW, da{', dX' if args.dx else ''})",
                       "nodes": N, "edges": E, "fin": Fin, "heads": H, "f_out": Fo,
                       "parallelism": (f"head-parallel x{model_world}, "
                                       + ("forward replicated (no collective; experiment)" if replicate
                                          else "copy-free exchange of the head outputs (grouped RCCL send/recv per row chunk into the "
                                               "column-blocked activation) overlapped with the forward's later chunks and the backward")
                                       + (" (rank-0 work only, modelled)" if model_world != world else ""))
                       if model_world > 1 else "single GPU",
                       "heads_per_gpu": h_loc,
                       "gemm_products": pg.get_gemm_mode(),
                       "node_order": ("internal degree order (x permuted once per feature tensor and cached; out / G / saved output "
                                      "addressed at the caller's rows inside K2 / K3a; no permutation pass in the step)"
                                      + (f"; the {tail_rows} self-loop-only nodes (alpha_ii = 1 exactly) as a contiguous tail through two "
                                         f"plain streams inside the k2_forward / k4_backward_col spans, K3a on the {n_k3a} rows before them"
                                         if tail_rows else "")
                                      if renumbered else "caller's"),
                       "launch": "HIP-graph replay (pygat_amd.GraphedLevel)" if (args.hip_graph and not replicate)
                       else "stream launches (pygat_amd.GATLevelFn)"},
            "roofline": roof,
            "kernels": kernels,
            "kernels_ms_sum": float(sum(kt.values())),
            "traffic_source": traffic_source,
            "alt": alt,
            "alt_node_order": alt_order,
        }
        if cpu_record is not None:
            line["cpu_baseline"] = cpu_record
        if world == 1 and model_world == 1 and not args.no_v2 and not args.dx:
            try:
                line["gatv2"] = gatv2_level_record(pg, ops, graph, X, H, Fo)
            except Exception as ex:
                line["gatv2"] = {"ms_per_step": None, "error": repr(ex)}
        if world == 1 and model_world == 1 and not args.no_epoch:
            line["epoch_ms"] = {}
            for name in ("cora", "pubmed", "ppi"):
                try:
                    line["epoch_ms"][name] = epoch_ms(pg, dev, name, epochs=100 if name == "ppi" else 200)
                except Exception as ex:
                    line["epoch_ms"][name] = {"ms": None, "error": repr(ex)}
        print(json.dumps(line), file=real_stdout, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
