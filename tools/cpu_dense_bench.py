#!/usr/bin/env python3
"""Time the reference's DENSE O(N^2) formulation on the host CPU.

    python tools/cpu_dense_bench.py [--dataset pubmed|cora|citeseer] [--fout 8] [--reps 5]

One GraphAttentionLayer head (layers.py:32-64: [N,N] logits, where(adj>0), softmax, att @ Wh), dropout
0, forward + backward through stock torch CPU autograd, on the real topology with synthetic features
-- the oracle's restatement `oracle.gat_oracle.dense_head_forward`, NOT the reference module (which
cannot be imported, see oracle header).  BASELINE.md section 2 measured the reference itself at 3.17 s
per step for the Pubmed case (500 -> 8) on the build container's 8 cores; this script must land near
that there, and on the GPU box it gives the "same box's host cores" figure of the north star.
Prints one JSON line (median of --reps, core count, CPU model).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import gat_oracle as O  # noqa: E402

FIN = {"cora": 1433, "citeseer": 3703, "pubmed": 500}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="pubmed")
    ap.add_argument("--fout", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0, help="torch CPU threads (0: one per physical core the cgroup quota grants)")
    a = ap.parse_args()
    from oracle.cpu_bench import physical_cores
    cores, hw, quota = physical_cores()
    threads = a.threads or (cores if quota is None else max(1, min(cores, int(quota + 1e-9))))
    torch.set_num_threads(threads)
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{a.dataset}_csr.npz"))
    rowptr, col = z["rowptr"], z["col"]
    N, E, Fin = len(rowptr) - 1, len(col), FIN[a.dataset]
    g = torch.Generator().manual_seed(72)
    x = (torch.rand(N, Fin, generator=g) < 0.013).float()
    x = x / x.sum(1, keepdim=True).clamp(min=1)
    adj = O.dense_from_csr(rowptr, col, N)
    W = torch.empty(Fin, a.fout); torch.nn.init.xavier_uniform_(W, gain=1.414, generator=g); W.requires_grad_()
    av = torch.empty(2 * a.fout, 1); torch.nn.init.xavier_uniform_(av, gain=1.414, generator=g); av.requires_grad_()
    G = torch.randn(N, a.fout, generator=g)
    ts = []
    for _ in range(a.reps + 1):
        t0 = time.perf_counter()
        out = O.dense_head_forward(x, adj, W, av, 0.2, True)
        out.backward(G)
        ts.append(time.perf_counter() - t0)
        W.grad = av.grad = None
    med = statistics.median(ts[1:])
    print(json.dumps({"case": f"{a.dataset} one dense head {Fin}->{a.fout}, dropout 0, fwd+bwd", "N": N, "E": E,
                      "seconds_per_step": med, "edges_per_s": E / med, "threads": torch.get_num_threads(), "reps": a.reps,
                      "hw_threads_visible": hw, "cgroup_cpu_quota": quota,
                      "cpu": cpu_model(), "kind": "oracle restatement of layers.py:32-64 (dense N x N)"}))


if __name__ == "__main__":
    main()
