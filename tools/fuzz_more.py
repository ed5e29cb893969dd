#!/usr/bin/env python3
"""Extended fuzz campaign (development tool): 400 random shapes / patterns / slot lengths / backward flavours /
head-window settings, forward + all gradients of one level against the fp64 oracle.  tests/ keeps 24 of these."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pygat_amd as pg
from oracle import gat_oracle as O
from test_gpu_parity import params, close, run_level
bad = cond = 0
for seed in range(24, 424):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(1, 400)); H = int(rng.choice([1, 2, 3, 4, 6, 8, 12])); Fo = int(rng.choice([1, 3, 4, 5, 8, 16, 17, 32, 64, 100]))
    Fin = int(rng.integers(1, 70)); skip, concat = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    slot = int(rng.choice([4, 8, 16, 64]))
    if rng.integers(0, 2):
        rowptr, col = O.random_symmetric_csr(N, float(rng.uniform(0.5, 12)), seed, hub=(0, int(rng.integers(1, N + 1))))
    else:
        dense = (rng.random((N, N)) < rng.uniform(0.01, 0.3)) | np.eye(N, dtype=bool)
        rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32); col = np.nonzero(dense)[1].astype(np.int32)
    pg.ops.TWO_GATHER_BACKWARD = [None, True, False][seed % 3]
    pg.ops.BWD_WINDOW_FLOATS = 256 if seed % 2 else None     # odd seeds: backward in head windows of <= 256 floats
    W, a, Sk = params(H, Fin, Fo, skip, seed)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen); G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    ref = O.csr_layer_fwd_bwd(x.numpy(), rowptr, col, W.numpy(), a.numpy(), 0.2, concat, G.numpy(), None if Sk is None else Sk.numpy())
    try:
        out, dx, dW, da, dS = run_level(pg, x, rowptr, col, W, a, Sk, concat, G, slot=slot)
        close(out, ref["out"], "out"); close(dx, ref["dX"], "dX"); close(dW, ref["dW"], "dW"); close(da, ref["da"], "da")
        if skip: close(dS, ref["dW_skip"], "dW_skip")
    except AssertionError as e:
        # SURVEY.md 8(c) tolerance for gradients: <= max(1e-5, 4 x the error of an fp32 run of the reference
        # algorithm against the same fp64 ground truth).  An edge whose logit s_i + t_j rounds to the other side of
        # zero in fp32 flips LeakyReLU' between 1 and alpha: any fp32 implementation then differs from fp64 there.
        f32 = lambda t: None if t is None else t.numpy().astype(np.float32)  # noqa: E731
        r32 = O.csr_layer_fwd_bwd(f32(x), rowptr, col, f32(W), f32(a), 0.2, concat, f32(G), f32(Sk))
        worst = 0.0
        for got, key in ((dx, "dX"), (dW, "dW"), (da, "da")):
            mine = float(np.abs(got.detach().double().cpu().numpy() - ref[key]).max())
            theirs = float(np.abs(r32[key].astype(np.float64) - ref[key]).max())
            worst = max(worst, mine / max(theirs, 1e-5 * max(1.0, float(np.abs(ref[key]).max()))))
        if worst <= 4.0:
            cond += 1; print("seed", seed, "outside 1e-5 but within 4x of the fp32 oracle's own error (x%.2f): conditioning" % worst, flush=True)
        else:
            bad += 1; print("FAIL seed", seed, N, H, Fo, Fin, skip, concat, slot, str(e)[:120], flush=True)
    if seed % 50 == 0: print("seed", seed, "ok so far, bad =", bad, flush=True)
print("done, bad =", bad, "conditioning-limited =", cond)
