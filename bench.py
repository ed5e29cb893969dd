#!/usr/bin/env python3
"""Headline benchmark: one GAT level forward+backward on a synthetic R-MAT graph.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json config 5 / SURVEY.md 8(d)): R-MAT scale 20 (N = 1,048,576 nodes),
5,000,000 edge draws with (a,b,c,d) = (0.57,0.19,0.19,0.05), seed 1, symmetrised, de-duplicated,
self loops added (E ~ 10.76 M); X ~ N(0,1) [N,128]; 8 heads x F' = 16 (H*F' = 128); concat + ELU;
eval-mode semantics (dropout 0); fp32.  A step = forward + backward of the level (dW, da; the
input of a first level carries no gradient in the reference, train.py:132,158).

N GPUs: heads sharded head-per-GPU (pygat_amd/dist.py): each rank projects and attends its
H/N heads, an RCCL all-gather concatenates the head outputs, backward is local.  Total work is
fixed -> "scaling": "strong".

One JSON line on rank 0: value = E / step time (max over ranks), plus `roofline` for the
dominant kernel (K2 fused edge-softmax+aggregate, HIP events around every launch in the timed
region, algorithmic bytes of SURVEY.md 8(d)) and `cpu_baseline` (oracle/gat_oracle.c, the CPU
port of the same level, timed on this host's cores; N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


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
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--verify", action="store_true", help="check the gathered sharded output against the unsharded level")
    ap.add_argument("--forward-exchange", choices=["auto", "allgather", "replicate"], default="auto",
                    help="N>1, how every rank gets the other ranks' head outputs: RCCL all-gather over xGMI, or no "
                         "collective at all -- run the forward of ALL heads (K1+K2) and back-propagate only the own ones "
                         "(GATLevelFn bwd_heads).  auto = replicate at 2 and 4 GPUs (receiving 256 / 3x128 MB over 1 / 3 "
                         "xGMI links takes longer than 0.4-0.9 ms of extra forward), all-gather at 8 (7 links in parallel)")
    ap.add_argument("--as-rank-of", type=int, default=0,
                    help="single process: run the work of rank 0 of a world of this size (per-rank time model, no collectives)")
    return ap.parse_args()


def cpu_baseline(args, rowptr, col, X, W, a, G):
    """oracle/gat_oracle.c (kind "port") on the host cores: the same level, same inputs."""
    subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import c_oracle
    lib = c_oracle.load()
    tp = c_oracle.transpose_pattern(rowptr, col)
    times = []
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        c_oracle.level(X, rowptr, col, W, a, 0.2, True, G, want_dx=args.dx, lib=lib, tp=tp)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": len(col) / best, "unit": "edges/s", "cores": int(lib.gat_oracle_threads()), "kind": "port",
            "sample": f"full workload (N={len(rowptr)-1}, E={len(col)}), best of {args.cpu_steps} fwd+bwd steps, "
                      f"{best:.2f} s/step, OpenMP C port oracle/gat_oracle.c"}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON record: native libraries (RCCL prints a version banner on its
    # first collective) get stderr instead
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    from pygat_amd.dist import partition_heads, all_gather_columns_raw
    from pygat_amd.rmat import rmat_csr

    H, Fo, Fin = args.heads, args.fout, args.fin
    rowptr, col = rmat_csr(args.scale, args.draws, seed=1, device=dev)
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
    widths = [(e - s) * Fo for s, e in parts]
    W_loc = W[hs:he].contiguous().requires_grad_(True)
    a_loc = a[hs:he].contiguous().requires_grad_(True)
    G_loc = G[:, hs * Fo:he * Fo].contiguous()
    Xb = X.requires_grad_(True) if args.dx else X

    replicate = model_world > 1 and (args.forward_exchange == "replicate" or
                                     (args.forward_exchange == "auto" and model_world <= 4))
    use_pg = (world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1") and not replicate
    gbuf = torch.empty(world * N, h_loc * Fo, device=dev) if use_pg else None
    if replicate:     # forward of ALL heads in one call, backward of the own ones (GATLevelFn bwd_heads)
        W_all = W.clone().requires_grad_(True)
        a_all = a.clone().requires_grad_(True)

    def step():
        W_loc.grad = a_loc.grad = None
        if args.dx:
            Xb.grad = None
        if replicate:
            W_all.grad = a_all.grad = None
            full = pg.GATLevelFn.apply(X.detach(), W_all, a_all, None, graph, 0.2, True, (hs, h_loc))
            full.backward(G)                              # only the columns of the own heads are read
            return full
        out = pg.GATLevelFn.apply(Xb, W_loc, a_loc, None, graph, 0.2, True)
        if use_pg:
            # RCCL all-gather of the head outputs (models.py:32 torch.cat) on RCCL's own stream; this
            # level's backward does not depend on it, so it overlaps K3/K4/K5 and is joined at the end.
            work = dist.all_gather_into_tensor(gbuf, out.detach(), async_op=True)
            out.backward(G_loc)
            work.wait()
            full = gbuf.view(world, N, h_loc * Fo).permute(1, 0, 2).reshape(N, world * h_loc * Fo)
        else:
            out.backward(G_loc)
            full = out
        return full

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timer = ops.KernelTimer()
    ops.TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = dt / args.steps * 1e3
    if args.verify and rank == 0:
        # the gathered concat output of the sharded run against the unsharded level on this rank
        full = step()
        ref = pg.GATLevelFn.apply(X.detach(), W, a, None, graph, 0.2, True)
        err = float((full - ref).abs().max())
        print(f"bench --verify: max |sharded - unsharded| = {err:.3e} over {tuple(ref.shape)}", file=sys.stderr)
        assert err < 1e-5, err
    if args.verify and rank != 0 and use_pg:
        step()

    if rank == 0:
        kt = {k: float(np.mean(v)) for k, v in timer.times_ms().items()}
        Fp = pg.padded_width(Fo)
        R = h_loc * Fp
        # per edge col index + t_j per local head + one Wh row; per node rowptr + s_i + out row + (m,Z)
        b_fwd = E * (4 + 4 * h_loc + 4 * R) + N * (4 + 4 * h_loc + 4 * R + 8 * h_loc)
        k2 = kt.get("k2_forward", float("nan"))
        achieved = b_fwd / (k2 * 1e-3) / 1e9
        # HBM traffic of K2 from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # runs of this same command; FETCH_SIZE doubled as the gfx950 guide prescribes): main + fix-up launch.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_k2_latest.json")))
            if pmc.get("workload_edges") == E and pmc.get("heads_per_gpu") == h_loc:
                traffic = pmc["k2_hbm_traffic_bytes"]
        except Exception:
            traffic = None
        line = {
            "metric": "GAT-layer fwd+bwd edges/sec", "value": E / (ms * 1e-3), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"RMAT scale {args.scale} ({N} nodes, {E} edges incl. self loops, max degree "
                                   f"{int((rowptr[1:] - rowptr[:-1]).max())}), Fin {Fin}, {H} heads x {Fo}, concat+ELU, "
                                   f"dropout 0, fwd+bwd (dW, da{', dX' if args.dx else ''})",
                       "nodes": N, "edges": E, "fin": Fin, "heads": H, "f_out": Fo,
                       "parallelism": (f"head-parallel x{model_world}" + (", forward replicated" if replicate else "")
                                       + (" (rank-0 work only, modelled)" if model_world != world else ""))
                       if model_world > 1 else "single GPU",
                       "heads_per_gpu": h_loc},
            "roofline": {"kernel": "k2_forward (gat_fwd_kernel + gat_fwd_fixup_kernel)", "bound": "hbm", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "algorithmic_bytes": b_fwd, "avg_ms": k2},
            "kernels_ms": kt,
        }
        if world == 1 and not args.no_cpu:
            try:
                line["cpu_baseline"] = cpu_baseline(args, rowptr.cpu().numpy(), col.cpu().numpy(), X.detach().cpu().numpy(),
                                                    W.cpu().numpy(), a.cpu().numpy(), G.cpu().numpy())
            except Exception as ex:  # the GPU number stays valid without the CPU leg
                line["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {ex!r}"}
        print(json.dumps(line), file=real_stdout, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
