"""Sparse input features: the non-zero pattern of the first level's input, extracted once.

The reference hands its models a dense [N, Fin] feature tensor (utils.py:38-41,60: `np.array(features.todense())`) that is a
row-normalised bag of words on its citation datasets -- Cora 1.27 % non-zero, Citeseer 0.85 %, Pubmed 10 % -- and multiplies
it, zeros included, with every head's W in every epoch (layers.py:35,134).  The input never changes between epochs, so its
pattern is extracted ONCE (cached on the tensor like the adjacency, graph.as_graph) as CSR for the projection and as its
transpose for the weight gradient; csrc/k9_sparse.hip then does both products on the non-zeros only, under the same
per-head dropout decisions as the dense kernels.  Dense inputs (density above MAX_DENSITY), inputs that require a gradient
and hidden levels keep the dense GEMMs.  PYGAT_SPARSE_X=0 switches the whole path off.  Inputs narrower than
MIN_PROBE_COLUMNS (128) are not probed implicitly (they are dense feature vectors, e.g. PPI's 50 columns): call
prepare_features(x) to put a narrow sparse input on the path.
"""
from __future__ import annotations

import collections
import weakref
from typing import Optional

import torch

from .config import config as _config

MAX_DENSITY = _config.sparse_max_density   # above this the dense MFMA GEMMs win (Pubmed: 10 %)
SEGMENT = 128               # entries per weight-gradient segment (k9_sparse.hip SP_SEG)
MAX_COLUMNS = 512           # output columns 2 R + H the sparse kernels take (k9_sparse.hip SP_CPL)
ENABLED = _config.sparse_x


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
        cnt = torch.bincount(cols, minlength=fin)
        colptr = torch.zeros(fin + 1, dtype=torch.int64, device=x.device)
        colptr[1:] = torch.cumsum(cnt, 0)
        # segments of at most SEGMENT entries, none crossing a column, at least one per column (k9_sparse.hip: a wave each)
        nseg_k = torch.clamp((cnt + SEGMENT - 1) // SEGMENT, min=1)
        colseg = torch.zeros(fin + 1, dtype=torch.int64, device=x.device)
        colseg[1:] = torch.cumsum(nseg_k, 0)
        self.nseg = int(colseg[-1])
        seg_col = torch.repeat_interleave(torch.arange(fin, device=x.device), nseg_k)
        within = torch.arange(self.nseg, device=x.device) - colseg[seg_col]
        seg_begin = colptr[seg_col] + SEGMENT * within
        seg_end = torch.minimum(seg_begin + SEGMENT, colptr[seg_col + 1])
        self.colseg = colseg.to(torch.int32).contiguous()
        self.seg_col = seg_col.to(torch.int32).contiguous()
        self.seg_begin = seg_begin.to(torch.int32).contiguous()
        self.seg_end = seg_end.to(torch.int32).contiguous()
        self.density = self.nnz / max(1, n * fin)


CACHE_ENTRIES = 64          # per cache; least recently used entries leave first (a dead tensor's entry when its key is reused)
MIN_PROBE_COLUMNS = 128     # narrower inputs are never probed: they are dense features (PPI: 50), not bags of words


class _LRU(collections.OrderedDict):
    """key -> (weakref of the tensor, value); a hit moves the entry to the young end, an insert beyond CACHE_ENTRIES drops the
    OLDEST one only (rounds 2-3 cleared the whole cache at 17 entries: a loop over 17 feature tensors missed every time)."""

    def lookup(self, key, x):
        hit = self.get(key)
        if hit is not None and hit[0]() is x:
            self.move_to_end(key)
            return True, hit[1]
        return False, None

    def insert(self, key, x, value):
        self[key] = (weakref.ref(x), value)
        self.move_to_end(key)
        while len(self) > CACHE_ENTRIES:
            self.popitem(last=False)


_cache = _LRU()


def eligible(x, min_columns: int = 0) -> bool:
    """The ONE eligibility rule of the sparse-feature path (both entry points below): a 2-D float32 GPU tensor that carries no
    gradient, at least `min_columns` wide, with the path switched on."""
    return (ENABLED and isinstance(x, torch.Tensor) and x.is_cuda and not x.requires_grad and x.dim() == 2
            and x.dtype == torch.float32 and x.shape[1] >= min_columns)


def prepare_features(x: torch.Tensor, min_columns: int = 0) -> Optional[SparseFeatures]:
    """The EXPLICIT form of what as_sparse_features does on a model's first forward: extract (and cache) the sparse pattern
    of an input feature tensor, outside any timed region or stream capture.  Returns the SparseFeatures, or None when the
    tensor is too dense for the sparse kernels (density above MAX_DENSITY).  Costs one host sync (the density probe) and a
    few sort / scan launches.  Raises ValueError for a tensor the sparse path can never take (see `eligible`: not a 2-D
    float32 GPU tensor, requires a gradient, or the path is switched off) -- rounds 3-4 built and cached a pattern for
    anything handed in.  `min_columns`: the implicit probe skips inputs narrower than MIN_PROBE_COLUMNS (128: dense feature
    vectors such as PPI's 50 columns, train_ppi.py:118, are never synchronised on); an explicit call probes any width
    unless told otherwise."""
    if not eligible(x, min_columns):
        raise ValueError("prepare_features: expected a 2-D float32 GPU tensor without requires_grad"
                         + (f", at least {min_columns} columns wide" if min_columns else "")
                         + ("" if ENABLED else " (the sparse-feature path is switched off: PYGAT_SPARSE_X=0)"))
    key = (x.data_ptr(), tuple(x.shape), x._version, str(x.device))
    ok, val = _cache.lookup(key, x)
    if ok:
        return val
    density = float(torch.count_nonzero(x)) / max(1, x.numel())      # the one device sync per feature tensor
    xs = SparseFeatures(x) if 0.0 < density <= MAX_DENSITY else None
    _cache.insert(key, x, xs)
    return xs


def as_sparse_features(x: torch.Tensor, out_columns: int) -> Optional[SparseFeatures]:
    """The cached SparseFeatures of `x` if the sparse kernels should take its projection, else None.
    A tensor seen for the first time is probed (prepare_features: one host sync) -- unless it is narrower than
    MIN_PROBE_COLUMNS (too narrow to be a bag of words: a per-batch loop over fresh dense feature tensors, train_ppi.py:118, is
    never synchronised; a genuinely sparse narrow input gets the path through an explicit prepare_features(x), whose cache
    entry is honoured here), or a stream capture is in progress (a sync would abort it: the level then runs dense; call
    prepare_features(x) before capturing)."""
    if not eligible(x) or out_columns > MAX_COLUMNS:
        return None
    key = (x.data_ptr(), tuple(x.shape), x._version, str(x.device))
    ok, val = _cache.lookup(key, x)
    if ok:
        return val
    if x.shape[1] < MIN_PROBE_COLUMNS or torch.cuda.is_current_stream_capturing():
        return None
    return prepare_features(x)


_perm_cache = _LRU()


def permuted_rows(x: torch.Tensor, to_user: torch.Tensor) -> Optional[torch.Tensor]:
    """x[to_user] -- the input features in a graph's INTERNAL node order (CSRGraph.degree_ordered) -- built once per
    (feature tensor, graph) and cached like the padded copy below: a first level's input does not change between epochs
    (train.py:93-99 loads `features` once).  None inside a stream capture when the copy is not cached yet (the level then
    runs in the caller's order)."""
    key = (x.data_ptr(), tuple(x.shape), x._version, str(x.device), str(x.dtype), to_user.data_ptr())
    ok, val = _perm_cache.lookup(key, x)
    if ok:
        return val[0]
    if torch.cuda.is_current_stream_capturing():
        return None
    xp = x.index_select(0, to_user.long()).float().contiguous()
    _perm_cache.insert(key, x, (xp, to_user))     # (the entry keeps the map alive: its address, part of the key, cannot be reused under it)
    weakref.finalize(x, _perm_cache.pop, key, None)     # a copy this size (512 MB at config 5) leaves with its tensor, not with the LRU
    return xp


_pad_cache = _LRU()


def padded_columns(x: torch.Tensor, multiple: int) -> torch.Tensor:
    """x [N, Fin] with zero columns appended up to a multiple of `multiple`, float32 -- built once per feature tensor (same
    cache discipline as the sparse pattern: keyed by storage, shape and version, dropped with the tensor).  For a first level
    whose Fin is not a multiple of 16 (PPI: 50): ops.gat_level runs it on the padded copy with zero rows appended to W."""
    fin = x.shape[1]
    cols = -(-fin // multiple) * multiple
    if cols == fin and x.dtype == torch.float32 and x.is_contiguous():
        return x
    key = (x.data_ptr(), tuple(x.shape), x._version, str(x.device), str(x.dtype), multiple)
    ok, val = _pad_cache.lookup(key, x)
    if ok:
        return val
    xp = torch.zeros(x.shape[0], cols, dtype=torch.float32, device=x.device)
    xp[:, :fin] = x
    _pad_cache.insert(key, x, xp)
    return xp
