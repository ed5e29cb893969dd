#!/usr/bin/env python3
"""How many (row, head) pairs of the bench workload have edges on BOTH LeakyReLU branches?  Only those need the
negative-branch aggregate `aneg` of the row-local backward (ds_i = 0 otherwise).  Development diagnostic."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from pygat_amd.rmat import rmat_csr  # noqa: E402

dev = torch.device("cuda", 0)
scale, draws, Fin, H, Fo = 20, 6_000_000, 128, 8, 16
if len(sys.argv) > 1:
    draws = int(sys.argv[1])
rowptr, col = rmat_csr(scale, draws, seed=1, device=dev)
N, E = rowptr.numel() - 1, col.numel()
X = torch.randn(N, Fin, generator=torch.Generator(device=dev).manual_seed(2), device=dev)
g3 = torch.Generator(device=dev).manual_seed(3)
W = torch.randn(H, Fin, Fo, generator=g3, device=dev) * (1.414 * (2.0 / (Fin + Fo)) ** 0.5)
a = torch.randn(H, 2 * Fo, generator=g3, device=dev) * (1.414 * (2.0 / (1 + 2 * Fo)) ** 0.5)
Wh = torch.einsum("nk,hkf->nhf", X, W)
s = torch.einsum("nhf,hf->nh", Wh, a[:, :Fo]); t = torch.einsum("nhf,hf->nh", Wh, a[:, Fo:])
deg = (rowptr[1:] - rowptr[:-1]).long()
row = torch.repeat_interleave(torch.arange(N, device=dev), deg)
z = s[row] + t[col.long()]                      # [E, H]
pos = torch.zeros(N, H, device=dev).index_add_(0, row, (z > 0).float())
neg = deg[:, None].float() - pos
mixed = (pos > 0) & (neg > 0)
print(f"N {N} E {E}: mixed (row, head) pairs {float(mixed.float().mean()):.3f}; rows with ANY mixed head "
      f"{float(mixed.any(1).float().mean()):.3f}; rows with degree 1: {float((deg == 1).float().mean()):.3f}, <= 2: "
      f"{float((deg <= 2).float().mean()):.3f}")
