"""Synthetic R-MAT graphs for the headline benchmark (BASELINE.json config 5).

Not part of the reference (which cannot run beyond N ~ 20k: dense N x N adjacency,
utils.py:55).  Edge draws follow the R-MAT recursion with quadrant probabilities
(a,b,c,d); the result is symmetrised, de-duplicated and given self loops, i.e. the
pattern shape utils.py:49-52 produces.  Generated with torch ops on whatever device
is asked for (graph construction is outside the timed region).
"""
from __future__ import annotations

import torch


def rmat_csr(scale: int = 20, n_draws: int = 5_000_000, abcd=(0.57, 0.19, 0.19, 0.05), seed: int = 1,
             device="cpu"):
    """-> (rowptr int32 [N+1], col int32 [E]) of the symmetric pattern with self loops, N = 2**scale."""
    n = 1 << scale
    gen = torch.Generator(device=device).manual_seed(seed)
    a, b, c, _ = abcd
    r = torch.zeros(n_draws, dtype=torch.int64, device=device)
    cidx = torch.zeros(n_draws, dtype=torch.int64, device=device)
    for _bit in range(scale):
        u = torch.rand(n_draws, generator=gen, device=device)
        # quadrants: [0,a) -> (0,0); [a,a+b) -> (0,1); [a+b,a+b+c) -> (1,0); rest -> (1,1)
        rb = (u >= a + b).to(torch.int64)
        cb = ((u >= a) & (u < a + b) | (u >= a + b + c)).to(torch.int64)
        r = (r << 1) | rb
        cidx = (cidx << 1) | cb
    ar = torch.arange(n, device=device, dtype=torch.int64)
    rr = torch.cat([r, cidx, ar])
    cc = torch.cat([cidx, r, ar])
    key = torch.unique(rr * n + cc)
    rr, cc = key // n, key % n
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rr, minlength=n), 0)
    return rowptr.to(torch.int32), cc.to(torch.int32)


def rmat_csr_numpy(scale: int = 20, n_draws: int = 5_000_000, abcd=(0.57, 0.19, 0.19, 0.05), seed: int = 1):
    """The same recipe from a NUMPY stream on the host -> (rowptr int32 [N+1], col int32 [E]) numpy arrays.  bench.py draws the
    headline graph this way and uploads it, so that its GPU leg and its cpu_baseline leg (a process without torch:
    oracle/cpu_bench.py carries a copy of this function, pinned to it by tests/test_dist_cpu.py) time the IDENTICAL graph
    (rounds 1-4: one graph per leg, E different in the fourth digit)."""
    import numpy as np
    n = 1 << scale
    rng = np.random.default_rng(seed)
    a, b, c, _ = abcd
    r = np.zeros(n_draws, np.int64)
    ci = np.zeros(n_draws, np.int64)
    for _bit in range(scale):
        u = rng.random(n_draws, dtype=np.float32)
        rb = (u >= a + b).astype(np.int64)
        cb = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.int64)
        r = (r << 1) | rb
        ci = (ci << 1) | cb
    ar = np.arange(n, dtype=np.int64)
    key = np.unique(np.concatenate([r, ci, ar]) * n + np.concatenate([ci, r, ar]))
    rr, cc = key // n, key % n
    rowptr = np.zeros(n + 1, np.int64)
    rowptr[1:] = np.cumsum(np.bincount(rr, minlength=n))
    return rowptr.astype(np.int32), cc.astype(np.int32)
