#!/usr/bin/env python3
"""Kernel-level micro-benchmarks for K2/K3/K4 on the headline workload (development tool).

    python tools/kbench.py [--graph rmat|regular|cora|citeseer|pubmed] [--chunk 256] [--iters 20] [--heads 8 --fout 16]

Times each kernel through the C ABI with HIP events (median of --iters) and prints algorithmic
GB/s (SURVEY.md 8(d) byte model).  Also usable under `rocprofv3 --pmc ... -- python3 tools/kbench.py`.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="rmat")
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ts", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--fout", type=int, default=16)
    ap.add_argument("--fin", type=int, default=128)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import pygat_amd as pg
    from pygat_amd import _lib
    from pygat_amd._lib import lib, check
    from pygat_amd.rmat import rmat_csr
    dev = torch.device("cuda", 0)
    H, Fo = args.heads, args.fout
    if args.graph == "rmat":
        rowptr, col = rmat_csr(args.scale, 5_000_000 * (1 << args.scale) // (1 << 20), seed=1, device=dev)
    elif args.graph in ("cora", "citeseer", "pubmed"):   # real topology (BASELINE.json configs 2-3)
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", f"{args.graph}_csr.npz"))
        rowptr = torch.from_numpy(z["rowptr"]).to(dev); col = torch.from_numpy(z["col"]).to(dev)
    else:  # every row: self loop + 9 random neighbours (not symmetric; forward-only experiments)
        n = 1 << args.scale
        g = torch.Generator(device=dev).manual_seed(1)
        nb = torch.randint(0, n, (n, 9), generator=g, device=dev)
        cols = torch.cat([torch.arange(n, device=dev)[:, None], nb], 1).sort(1).values
        rowptr = (torch.arange(n + 1, device=dev) * 10).to(torch.int32)
        col = cols.reshape(-1).to(torch.int32)
    graph = pg.CSRGraph(rowptr, col, slot_edges=args.ts)
    N, E = graph.n, graph.nnz
    Fp = pg.padded_width(Fo); R = H * Fp
    g2 = torch.Generator(device=dev).manual_seed(2)
    Wh = torch.randn(N, R, generator=g2, device=dev)
    s = torch.randn(N, H, generator=g2, device=dev); t = torch.randn(N, H, generator=g2, device=dev)
    G = torch.randn(N, H * Fo, generator=g2, device=dev)
    a_pad = torch.randn(H, 2, Fp, generator=g2, device=dev)
    out = torch.empty(N, H * Fo, device=dev); hattn = torch.empty(N, R, device=dev)
    m = torch.empty(N, H, device=dev); Z = torch.empty(N, H, device=dev)
    ds = torch.empty(N, H, device=dev); dt = torch.empty(N, H, device=dev); dWh = torch.empty(N, R, device=dev)
    part = torch.empty(max(1, lib.pygat_partials_bytes(E, args.ts, H, Fp) // 4), device=dev)
    P = lambda x: None if x is None else x.data_ptr()

    aneg = torch.empty(N, R, device=dev); qneg = torch.empty(N, H, device=dev)

    def k2(train=True, aux=False):
        check(lib.pygat_gat_forward(graph.fwd.ref(), H, Fo, 0.2, _lib.F_ELU, P(Wh), P(s), P(a_pad), None, None, P(out),
                                    None, P(m) if train else None, P(Z) if train else None, P(aneg) if aux else None,
                                    P(qneg) if aux else None, P(part), None))

    GR = torch.empty(N, R + 4 * H, device=dev)

    def k3a():
        check(lib.pygat_gat_backward_prepare(N, H, Fo, _lib.F_ELU, 0, P(G), P(out), None, P(s), P(m), P(Z), P(GR), None, None,
                                             0.2, None, 0, 0, 0, None, None))

    def k3a_ds():
        check(lib.pygat_gat_backward_prepare(N, H, Fo, _lib.F_ELU, 0, P(G), P(out), None, P(s), P(m), P(Z), P(GR), P(aneg),
                                             P(qneg), 0.2, P(ds), 0, 0, 0, None, None))

    def k3b():
        check(lib.pygat_gat_backward_row(graph.fwd.ref(), H, Fo, 0.2, P(Wh), P(a_pad), P(GR), None, P(ds), P(part), 0, 0, 0, None))

    def k4():
        check(lib.pygat_gat_backward_col(graph.bwd.ref(), None, H, Fo, 0.2, P(Wh), P(a_pad), P(GR), None, P(ds),
                                         P(dWh), P(dt), None, P(part), None, 0, 0, 0, None))

    dz_t = torch.empty(E, H, device=dev)

    def k4dz():   # default backward: K4 writes its dz per transposed edge ...
        check(lib.pygat_gat_backward_col(graph.bwd.ref(), None, H, Fo, 0.2, P(Wh), P(a_pad), P(GR), None, None,
                                         P(dWh), P(dt), P(dz_t), P(part), None, 0, 0, 0, None))

    def k3c():    # ... and the row sums are taken from those records
        check(lib.pygat_gat_backward_rowsum(graph.fwd.ref(), graph.perm_f.data_ptr(), H, Fo, P(dz_t), P(ds), P(part), 0, 0, None))

    b_fwd = E * (4 + 4 * H + 4 * R) + N * (4 + 4 * H + 4 * R + 8 * H)
    b_k3a = N * (12 * R + 12 * H + 16 * H)
    b_k3 = E * (4 + 4 * R + 8 * H) + N * (4 + 8 * R + 16 * H)
    b_k4 = E * (8 + 4 * R + 8 * H) + N * (4 + 8 * R + 8 * H)
    runs = [("k2_train", lambda: k2(True), b_fwd), ("k2_aux", lambda: k2(True, True), b_fwd + N * (4 * R + 4 * H)),
            ("k2_eval", lambda: k2(False), b_fwd - N * 8 * H),
            ("k3a_prep", k3a, b_k3a), ("k3a_ds", k3a_ds, b_k3a + N * (4 * R + 8 * H)), ("k3b_row", k3b, b_k3 - b_k3a), ("k4_col", k4, b_k4),
            ("k4_col_dz", k4dz, b_k4 + E * 4 * H), ("k3c_rowsum", k3c, E * (12 + 4 * H) + N * 4 * H)]
    if args.only:
        runs = [r for r in runs if r[0] in args.only.split(",")]
    deg = (rowptr[1:] - rowptr[:-1])
    res = {"graph": args.graph, "N": N, "E": E, "ts": args.ts, "H": H, "Fo": Fo, "max_deg": int(deg.max())}
    for name, fn, nbytes in runs:
        for _ in range(3):
            fn()
        ts = []
        for _ in range(args.iters):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        med = float(np.median(ts))
        res[name] = {"ms": round(med, 4), "alg_GBps": round(nbytes / med / 1e6, 1), "frac_8TBps": round(nbytes / med / 1e6 / 8000, 4)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
