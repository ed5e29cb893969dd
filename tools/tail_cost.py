#!/usr/bin/env python3
"""What the self-loop-only rows cost the fused forward (K2) once the graph is in internal degree order (they are a contiguous tail
of the row range then): the training forward over ALL slots against the same launch over the slots before the tail
(pygat_graph.slot_first / slot_count).  The tail's floor as a pure stream: N1 x (read Wh 4R + s 4H, write out 4R + m, Z, qneg 12H).
    python3 tools/tail_cost.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygat_amd as pg  # noqa: E402
from pygat_amd import _lib  # noqa: E402
from pygat_amd._lib import lib, check  # noqa: E402
from pygat_amd.rmat import rmat_csr_numpy  # noqa: E402

dev = torch.device("cuda", 0)
rp, col = rmat_csr_numpy(20, 5_000_000, seed=1)
g0 = pg.CSRGraph(torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev))
g, to_user, _ = g0.degree_ordered()
N, E, H, Fo, R = g.n, g.nnz, 8, 16, 128
deg = (g.fwd.rowptr[1:] - g.fwd.rowptr[:-1]).long()
n1 = int((deg > 1).sum())                       # internal rows [n1, N) have only their self loop
st = g.fwd._make(64, True)
_, sb, cut, meta, _ = g.fwd._alt[(64, True)]
first_row = meta[:, 2].long()
k_tail = int(torch.searchsorted(first_row, torch.tensor(n1, device=dev)))
while k_tail < meta.shape[0] and int(first_row[k_tail]) < n1:
    k_tail += 1
gen = torch.Generator(device=dev).manual_seed(2)
Wh = torch.randn(N, R, generator=gen, device=dev); s = torch.randn(N, H, generator=gen, device=dev)
a_pad = torch.randn(H, 2, 16, generator=gen, device=dev)
out = torch.empty(N, R, device=dev); m = torch.empty(N, H, device=dev); Z = torch.empty(N, H, device=dev)
aneg = torch.empty(N, R, device=dev); qneg = torch.empty(N, H, device=dev)
part = torch.empty(lib.pygat_partials_bytes(E, 64, H, 16) // 4, device=dev)
P = lambda t: t.data_ptr()  # noqa: E731


def k2(struct):
    check(lib.pygat_gat_forward(C.byref(struct), H, Fo, 0.2, _lib.F_ELU, P(Wh), P(s), P(a_pad), None, None, P(out), None, P(m), P(Z),
                                P(aneg), P(qneg), P(part), None))


head = _lib.Graph()
C.memmove(C.byref(head), C.byref(st), C.sizeof(_lib.Graph))
head.slot_first, head.slot_count = 0, k_tail
for name, struct in (("all slots", st), ("slots before the self-loop tail", head), ("all slots", st), ("slots before the self-loop tail", head)):
    for _ in range(3):
        k2(struct)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); k2(struct); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"{name:34s} {np.median(ts):.4f} ms")
n_tail = N - n1
print(f"tail: {n_tail} rows = {meta.shape[0] - k_tail} of {meta.shape[0]} slots; as a pure stream {n_tail * (8 * R + 16 * H) / 1e9:.3f} GB = "
      f"{n_tail * (8 * R + 16 * H) / 5.5e12 * 1e3:.3f} ms at 5.5 TB/s")
