"""Sparse input features: the non-zero pattern of the first level's input, extracted once.

The reference hands its models a dense [N, Fin] feature tensor (utils.py:38-41,60: `np.array(features.todense())`) that is a
row-normalised bag of words on its citation datasets -- Cora 1.27 % non-zero, Citeseer 0.85 %, Pubmed 10 % -- and multiplies
it, zeros included, with every head's W in every epoch (layers.py:35,134).  The input never changes between epochs, so its
pattern is extracted ONCE (cached on the tensor like the adjacency, graph.as_graph) as CSR for the projection and as its
transpose for the weight gradient; csrc/k9_sparse.hip then does both products on the non-zeros only, under the same
per-head dropout decisions as the dense kernels.  Dense inputs (density above MAX_DENSITY, or feature columns longer than MAX_COLUMN_NNZ on average), inputs that require a gradient
and hidden levels keep the dense GEMMs.  PYGAT_SPARSE_X=0 switches the whole path off.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch

MAX_DENSITY = 0.05          # above this the dense MFMA GEMM wins
MAX_COLUMN_NNZ = 256        # mean non-zeros per feature column: the weight gradient runs one wave per column (Cora 34,
                            # Citeseer 28; Pubmed's 500 TF-IDF columns hold 1976 each and stay on the dense kernels: measured
                            # 1.22 ms per epoch sparse against 0.70 dense)
MAX_COLUMNS = 512           # output columns 2 R + H the sparse kernels take (k9_sparse.hip SP_CPL)
ENABLED = os.environ.get("PYGAT_SPARSE_X", "1") != "0"


class SparseFeatures:
    """CSR (rowptr, col, val) and transposed CSR (colptr, row, tval) of a dense [N, Fin] float tensor, on its device."""

    def __init__(self, x: torch.Tensor):
        if not x.is_cuda or x.dim() != 2:
            raise ValueError("SparseFeatures: a 2-D GPU tensor is expected (there is no CPU path)")
        n, fin = x.shape
        nz = x.nonzero()                                   # row-major order = CSR order
        rows, cols = nz[:, 0], nz[:, 1]
        self.n, self.fin, self.nnz = n, fin, int(rows.numel())
        self.val = x[rows, cols].float().contiguous()
        self.col = cols.to(torch.int32).contiguous()
        self.rowptr = torch.zeros(n + 1, dtype=torch.int32, device=x.device)
        self.rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0).to(torch.int32)
        order = torch.argsort(cols, stable=True)           # by column, rows ascending inside a column
        self.trow = rows[order].to(torch.int32).contiguous()
        self.tval = self.val[order].contiguous()
        self.colptr = torch.zeros(fin + 1, dtype=torch.int32, device=x.device)
        self.colptr[1:] = torch.cumsum(torch.bincount(cols, minlength=fin), 0).to(torch.int32)
        self.density = self.nnz / max(1, n * fin)


_cache: "dict" = {}


def as_sparse_features(x: torch.Tensor, out_columns: int) -> Optional[SparseFeatures]:
    """The cached SparseFeatures of `x` if the sparse kernels should take its projection, else None."""
    if not ENABLED or not isinstance(x, torch.Tensor) or not x.is_cuda or x.requires_grad or x.dim() != 2 \
            or x.dtype != torch.float32 or out_columns > MAX_COLUMNS:
        return None
    key = (x.data_ptr(), tuple(x.shape), x._version, str(x.device))
    hit = _cache.get(key)
    if hit is not None and hit[0]() is x:
        return hit[1]
    density = float(torch.count_nonzero(x)) / max(1, x.numel())      # one device sync per feature tensor, then cached
    xs = SparseFeatures(x) if (0.0 < density <= MAX_DENSITY and density * x.shape[0] <= MAX_COLUMN_NNZ) else None
    if len(_cache) > 16:
        _cache.clear()
    _cache[key] = (weakref.ref(x), xs)
    return xs
