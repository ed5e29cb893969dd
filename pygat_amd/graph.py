"""Graph intake: adjacency -> cached CSR pattern on the GPU.

The reference hands every layer a dense [N,N] float adjacency (utils.py:55,
load_data_ppi.py:153) and re-derives the edge list from it in every forward of
every head (`adj.nonzero().t()`, layers.py:129), or masks with `adj > 0`
(layers.py:41).  Only the PATTERN is ever used.  Here the pattern is extracted
once (K0 kernels) and cached; all kernels then work on CSR:

  row i of the CSR  = softmax row i  = { j : adj[i, j] != 0 }   (edge[0]=i, edge[1]=j)

For the atomics-free backward the transposed pattern and, per transposed edge,
the index of its forward edge are needed.  All reference graphs are symmetric
(utils.py:49, load_data_ppi.py:157), where the transpose has the same CSR and
only a mirror permutation is built (HIP binary search); asymmetric patterns are
accepted and transposed with a device sort.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import lib, check
from .config import config as _config

# Experiment hook (tools/slot_order_probe.py): f(pattern, slot_meta [n_slots, 4] int32) -> int32 permutation
# [n_slots] handed to the kernels as pygat_graph.slot_order -- the order in which the main launches of K2 / K4 walk the slots.
# None (the shipped default): grid order = slot order.
SLOT_ORDER_FN = None
FORCED_SLOT_EDGES = _config.slot_edges     # PYGAT_SLOT_EDGES
USE_SLOT_META = _config.slot_meta          # PYGAT_NO_SLOT_META=1 switches the slot records off

DEFAULT_SLOT_EDGES = 64   # edges per work slot of the nnz-split kernels on large graphs (multiple of 4)


def auto_slot_edges(nnz: int) -> int:
    """Slot length for a graph of `nnz` edges.  A lane group walks its slot serially (dependent gather
    rounds of 4 edges), so a small graph wants short slots -- enough of them (>= 8192) to occupy the
    256 CUs -- while a large one wants 64-edge slots (fewer cut rows, less fix-up work).  Measured K2 on
    MI355X, 8 heads x 8: Cora (13 264 edges) 48 us at 64 -> 12.7 us at 4; Pubmed (108 365) 56 -> 20 us at 8."""
    if FORCED_SLOT_EDGES:     # development knob (config.slot_edges)
        return int(FORCED_SLOT_EDGES)
    ts = DEFAULT_SLOT_EDGES
    while ts > 4 and nnz // ts < 8192:
        ts //= 2
    return ts


def slot_edges_for(row_floats: int, base: int = DEFAULT_SLOT_EDGES) -> int:
    """Slot length for a head-interleaved row of `row_floats` floats.  A wave carries 64/LPR slots
    (LPR = lanes per row): with 64-byte rows (one 16-float head per GPU) that is 16 slots, and 32-edge
    slots measured ~5 % faster than 64-edge ones there (more waves to balance; round 4, with the fix-up kernels merging
    four cut rows per wave: 0.904 against 0.910 ms for the 8-GPU shard of config 5); wider rows keep `base`."""
    return min(base, 32) if row_floats <= 16 else base


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class _Pattern:
    """rowptr/col plus the (row, col) edge pairs, as the pygat_graph struct."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, slot_edges: int, user_row: Optional[torch.Tensor] = None):
        self.rowptr, self.col, self.slot_edges = rowptr, col, slot_edges
        self.user_row = user_row     # internal renumbering of a caller's graph: caller's row of node i (pygat_graph.user_row)
        self.n = rowptr.numel() - 1
        self.nnz = col.numel()
        self.edge_rc = torch.empty(self.nnz, 2, dtype=torch.int32, device=rowptr.device)
        with torch.cuda.device(rowptr.device):
            check(lib.pygat_edge_pairs(self.n, rowptr.data_ptr(), col.data_ptr(), self.edge_rc.data_ptr(), _stream()),
                  "edge_pairs")
        self._alt = {}          # (slot_edges, snapped) -> (struct, slot_begin tensor)
        self.struct = self._make(slot_edges, True)

    def _make(self, slot_edges: int, snapped: bool):
        key = (slot_edges, snapped)
        if key not in self._alt:
            sb = None
            if snapped:   # row-snapped slot borders (K2 / K4: fewer cut rows, fewer partial records)
                nslots = -(-self.nnz // slot_edges)
                sb = torch.empty(nslots + 1, dtype=torch.int32, device=self.rowptr.device)
                with torch.cuda.device(self.rowptr.device):
                    check(lib.pygat_slot_bounds(self.n, self.nnz, self.rowptr.data_ptr(), self.edge_rc.data_ptr(),
                                                slot_edges, sb.data_ptr(), _stream()), "slot_bounds")
            cut, n_cut, n_wide = None, 0, 0
            if snapped:   # rows cut by a slot border: (owner slot, row, pieces), longest chains first
                sbl = sb.long()
                e0, e1 = sbl[:-1], sbl[1:]
                r_last = self.edge_rc[e1 - 1, 0].long()
                rp = self.rowptr.long()
                row_end = rp[r_last + 1]
                k = torch.nonzero((row_end > e1) & (rp[r_last] >= e0)).flatten()
                if k.numel():
                    k_e = torch.searchsorted(sbl, row_end[k] - 1, right=True) - 1
                    pieces = k_e - k + 1
                    order = torch.argsort(pieces, descending=True, stable=True)
                    cut = torch.stack([k, r_last[k], pieces], 1)[order].to(torch.int32).contiguous()
                    n_cut, n_wide = int(k.numel()), int((pieces > 32).sum().item())
                else:     # no row is cut: an EMPTY list (non-NULL, n_cut = 0) -- the fix-up launches are skipped, not replaced by
                    cut = torch.zeros(1, 3, dtype=torch.int32, device=self.rowptr.device)     # a screening pass over every slot
            # one 16-byte record per slot (first edge, end edge, first row, cut flags): the kernels' start-up in one load
            meta = torch.empty(-(-self.nnz // slot_edges), 4, dtype=torch.int32, device=self.rowptr.device)
            with torch.cuda.device(self.rowptr.device):
                check(lib.pygat_slot_meta(self.n, self.nnz, self.rowptr.data_ptr(), self.edge_rc.data_ptr(), slot_edges, _ptr(sb),
                                          meta.data_ptr(), _stream()), "slot_meta")
            if not USE_SLOT_META:      # development knob (A/B of the slot records; config.slot_meta)
                meta = None
            order = SLOT_ORDER_FN(self, meta) if (SLOT_ORDER_FN is not None and meta is not None) else None
            st = _lib.Graph(self.n, self.nnz, _ptr(self.rowptr), _ptr(self.edge_rc), slot_edges, _ptr(sb), _ptr(cut),
                            n_cut, n_wide, 0, 0, _ptr(meta), _ptr(order), _ptr(self.user_row))
            self._alt[key] = (st, sb, cut, meta, order)
        return self._alt[key][0]

    def row_chunks(self, nchunks: int, slot_edges: Optional[int] = None, nslots: Optional[int] = None, row_end: Optional[int] = None):
        """Cut the pattern into up to `nchunks` ranges of WHOLE rows with about equal numbers of slots, for pipelining a
        level by row chunks (pygat_amd/dist.py).  -> [(pygat_graph* for pygat_gat_forward, row_first, row_end)].
        A chunk border is a slot border at which a row starts, so no row -- and no chain of partial records -- crosses it.
        nslots / row_end: cut only the slot PREFIX [0, nslots), which ends in front of row `row_end` (the slots before a
        self-loop-only tail, self_loop_tail)."""
        ts = slot_edges or self.slot_edges
        key = ("chunks", ts, nchunks, nslots)
        if key not in self._alt:
            base = self._make(ts, True)
            _, sb, cut, meta, _order = self._alt[(ts, True)]
            all_slots = sb.numel() - 1
            nslots = all_slots if nslots is None else int(nslots)
            sbl, rp = sb.long(), self.rowptr.long()
            first_row = self.edge_rc[sbl[:-1], 0].long()                 # row of the first edge of every slot
            starts_row = rp[first_row] == sbl[:-1]
            ok = torch.nonzero(starts_row[:nslots]).flatten()
            borders = [0]
            for c in range(1, nchunks):
                tgt = c * nslots // nchunks
                j = int(torch.searchsorted(ok, torch.tensor(tgt, device=ok.device)))
                if j < ok.numel() and int(ok[j]) > borders[-1]:
                    borders.append(int(ok[j]))
            borders.append(nslots)
            out, keep = [], []
            for b0, b1 in zip(borders[:-1], borders[1:]):
                r0 = int(first_row[b0])
                r1 = int(first_row[b1]) if b1 < nslots else (self.n if (row_end is None or nslots == all_slots) else int(row_end))
                sub, n_cut, n_wide = None, 0, 0
                if cut is not None:
                    sel = (cut[:, 0] >= b0) & (cut[:, 0] < b1)
                    sub = cut[sel].contiguous()                           # order (pieces descending) is kept
                    n_cut, n_wide = int(sub.shape[0]), int((sub[:, 2] > 32).sum().item()) if sub.numel() else 0
                    if n_cut == 0:
                        sub = cut[:1].contiguous()                       # a non-NULL list with n_cut = 0: nothing to fix up
                st = _lib.Graph(self.n, self.nnz, _ptr(self.rowptr), _ptr(self.edge_rc), ts, _ptr(sb), _ptr(sub), n_cut,
                                n_wide, b0, b1 - b0, _ptr(meta), None, _ptr(self.user_row))
                out.append((st, r0, r1))
                keep.append(sub)
            self._alt[key] = (out, keep)
            del base
        return [(C.byref(st), r0, r1) for st, r0, r1 in self._alt[key][0]]

    def self_loop_tail(self, slot_edges: Optional[int] = None):
        """For a pattern whose rows are in descending-degree order (CSRGraph.degree_ordered): (row_first, first_slot, prefix) --
        the rows [row_first, n) have only their self loop and occupy exactly the slots [first_slot, n_slots); `prefix` is the
        pygat_graph* of the slots before them (slot_first 0, slot_count first_slot) for pygat_gat_forward / pygat_gat_backward_col /
        pygat_a_grad_fold, the tail goes to pygat_gat_forward_tail / pygat_gat_backward_col_tail (csrc/k12_tail.hip).
        None when there is no such tail.  Computed once per slot length (two host reads)."""
        ts = slot_edges or self.slot_edges
        key = ("tail", ts)
        if key not in self._alt:
            st = self._make(ts, True)
            _, sb, cut, meta, order = self._alt[(ts, True)]
            out = None
            if meta is not None and order is None:
                rp = self.rowptr.long()
                deg = rp[1:] - rp[:-1]
                n1 = int((deg > 1).sum().item())                      # descending degree: rows [n1, n) are the self-loop-only ones
                if n1 < self.n and bool((deg[n1:] == 1).all().item()) and bool((self.col[rp[n1]:].long() == torch.arange(n1, self.n, device=self.col.device)).all().item()):
                    first_row = meta[:, 2].long()
                    k = int(torch.searchsorted(first_row, torch.tensor(n1, device=first_row.device)).item())
                    if k < meta.shape[0]:
                        row_first = int(first_row[k].item())
                        pre = _lib.Graph()
                        C.memmove(C.byref(pre), C.byref(st), C.sizeof(_lib.Graph))
                        pre.slot_first, pre.slot_count = 0, k
                        out = (row_first, k, pre)
            self._alt[key] = out
        t = self._alt[key]
        return None if t is None else (t[0], t[1], C.byref(t[2]))

    def ref(self, slot_edges: Optional[int] = None, snapped: bool = True):
        """pygat_graph* for a call.  `slot_edges` overrides the slot length for this call only (the edge
        arrays do not depend on it); `snapped=False` gives uniform slots (K3b has no row reduction)."""
        return C.byref(self._make(slot_edges or self.slot_edges, snapped))


class _UnmappedPattern:
    """A degree-ordered _Pattern handed to kernels whose node arrays are THEMSELVES in internal order (a model that permutes its
    input once and un-permutes its final output, CSRGraph.internal_view): the same device arrays, pygat_graph structs without
    the user_row map."""

    def __init__(self, base: "_Pattern"):
        self._base = base
        self._copies = {}
        self.user_row = None

    def __getattr__(self, name):               # rowptr, col, edge_rc, n, nnz, slot_edges, ...
        return getattr(self._base, name)

    def _strip(self, key, struct_ref):
        if key not in self._copies:
            st = _lib.Graph()
            C.memmove(C.byref(st), struct_ref, C.sizeof(_lib.Graph))
            st.user_row = None
            self._copies[key] = st
        return C.byref(self._copies[key])

    def ref(self, slot_edges: Optional[int] = None, snapped: bool = True):
        return self._strip(("ref", slot_edges or self._base.slot_edges, snapped), self._base.ref(slot_edges, snapped))

    def row_chunks(self, nchunks: int, slot_edges: Optional[int] = None, nslots: Optional[int] = None, row_end: Optional[int] = None):
        out = self._base.row_chunks(nchunks, slot_edges, nslots, row_end)
        return [(self._strip(("chunk", slot_edges or self._base.slot_edges, nchunks, nslots, c), ref), r0, r1)
                for c, (ref, r0, r1) in enumerate(out)]

    def self_loop_tail(self, slot_edges: Optional[int] = None):
        t = self._base.self_loop_tail(slot_edges)
        return None if t is None else (t[0], t[1], self._strip(("tail", slot_edges or self._base.slot_edges), t[2]))


class InternalOrderView:
    """CSRGraph.internal_view(): the degree-ordered pattern for callers whose node arrays are in INTERNAL order themselves --
    pygat_amd.GAT permutes x once (cached per feature tensor), runs every level of the model on this view (hidden levels too:
    their input is the previous level's output, already internal) and un-permutes the final [N, C] output; a head-parallel
    model exchanges internal-order rows, every rank holding the same order.  Duck-types CSRGraph for ops / dist."""
    degree_sorted = True
    user_row = None
    _ordered = None

    def __init__(self, g_int: "CSRGraph", to_user: torch.Tensor, to_internal: torch.Tensor):
        self.base = g_int
        self.n, self.nnz, self.device, self.slot_edges, self.symmetric = g_int.n, g_int.nnz, g_int.device, g_int.slot_edges, g_int.symmetric
        self.fwd = _UnmappedPattern(g_int.fwd)
        self.bwd = self.fwd if g_int.bwd is g_int.fwd else _UnmappedPattern(g_int.bwd)
        self.perm_t, self.perm_f = g_int.perm_t, g_int.perm_f
        self.to_user, self.to_internal = to_user, to_internal


class CSRGraph:
    degree_sorted = False      # True for the patterns CSRGraph.degree_ordered builds

    """Device-resident CSR pattern (+ transpose info) consumed by the HIP kernels."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, slot_edges: Optional[int] = None,
                 validate: bool = True, user_row: Optional[torch.Tensor] = None):
        if not (rowptr.is_cuda and col.is_cuda):
            raise ValueError("CSRGraph: rowptr/col must live on the GPU (there is no CPU path)")
        rowptr = rowptr.to(torch.int32).contiguous()
        col = col.to(torch.int32).contiguous()
        if rowptr.dim() != 1 or col.dim() != 1 or rowptr.numel() < 2:
            raise ValueError("CSRGraph: rowptr [N+1] and col [E] expected")
        self.device = rowptr.device
        self.n = rowptr.numel() - 1
        self.nnz = col.numel()
        if self.nnz == 0:
            raise ValueError("CSRGraph: empty pattern")
        if validate:
            self._validate(rowptr, col)
        if slot_edges is None:
            slot_edges = auto_slot_edges(int(col.numel()))
        if slot_edges < 4 or slot_edges % 4:
            raise ValueError("slot_edges must be a multiple of 4, >= 4")
        self.slot_edges = slot_edges
        self.user_row = user_row
        self._ordered = None
        self.fwd = _Pattern(rowptr, col, slot_edges, user_row)
        # mirror permutation (symmetric pattern, sorted rows) via the HIP binary search
        perm = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
        flags = torch.zeros(2, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.pygat_csr_symmetric_perm(self.n, rowptr.data_ptr(), col.data_ptr(), perm.data_ptr(),
                                               flags.data_ptr(), _stream()), "csr_symmetric_perm")
        asym, empty = (int(v) for v in flags.tolist())
        if validate and empty:
            # the reference NaN-asserts (sparse layer, layers.py:157,162) or silently attends to
            # ALL nodes (dense layer, softmax of a constant -9e15 row) on an empty row
            raise ValueError("CSRGraph: some node has no neighbour (add self loops as utils.py:52 does)")
        self.symmetric = not asym
        if self.symmetric:
            # the mirror permutation is an involution: it maps forward positions to transposed ones and back
            self.bwd, self.perm_t, self.perm_f = self.fwd, perm, perm
        else:
            self._build_transpose()

    @staticmethod
    def _validate(rowptr: torch.Tensor, col: torch.Tensor):
        """The kernels index Wh[col[k]] and rowptr[col[k]] unchecked: a malformed pattern would be an out-of-bounds
        device access.  One pass of device-side checks at graph build (outside any timed region)."""
        n, nnz = rowptr.numel() - 1, col.numel()
        rp = rowptr.long()
        deg = rp[1:] - rp[:-1]
        bad_rp = (rp[0] != 0) | (rp[-1] != nnz) | (deg < 0).any()
        bad_col = (col < 0).any() | (col >= n).any()
        if bool(bad_rp | bad_col):
            raise ValueError("CSRGraph: malformed CSR (need rowptr[0] == 0, rowptr[-1] == nnz, rowptr non-decreasing, "
                             "0 <= col < n)")
        if nnz > 1:
            # columns strictly increasing inside every row (sorted, no duplicates): the mirror permutation of a
            # symmetric pattern is found by binary search and must be a bijection
            first = torch.zeros(nnz, dtype=torch.bool, device=col.device)
            starts = rp[:-1][deg > 0]
            first[starts] = True
            if bool(((col[1:] <= col[:-1]) & ~first[1:]).any()):
                raise ValueError("CSRGraph: columns must be strictly increasing within each row (sort and "
                                 "de-duplicate the edge list; from_edge_index does)")

    def _build_transpose(self):
        rowptr, col = self.fwd.rowptr.long(), self.fwd.col.long()
        deg = rowptr[1:] - rowptr[:-1]
        src = torch.repeat_interleave(torch.arange(self.n, device=self.device), deg)
        key = col * self.n + src                       # sort by (j, i): stable order inside a column
        order = torch.argsort(key, stable=True)
        cnt = torch.bincount(col, minlength=self.n)
        rp_t = torch.zeros(self.n + 1, dtype=torch.int64, device=self.device)
        rp_t[1:] = torch.cumsum(cnt, 0)
        self.bwd = _Pattern(rp_t.to(torch.int32), src[order].to(torch.int32).contiguous(), self.slot_edges, self.user_row)
        self.perm_t = order.to(torch.int32).contiguous()          # transposed position -> forward edge
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=self.device)
        self.perm_f = inv.to(torch.int32).contiguous()            # forward edge -> transposed position

    def degree_ordered(self) -> "Tuple[CSRGraph, torch.Tensor, torch.Tensor]":
        """-> (g, to_user, to_internal): the same pattern with its nodes renumbered by DESCENDING DEGREE (stable: the caller's
        order inside a degree), built once and cached; internal node p is the caller's node to_user[p].  A level may run all
        its node tables in this INTERNAL order (ops.RENUMBER): the most-gathered rows of the gathered
        tables become neighbours and the self-loop-only nodes (55 % of the R-MAT workload) a contiguous tail -- measured on the
        headline graph: K2 0.99 -> 0.91 ms, K4 1.23 -> 1.17 at almost unchanged HBM traffic.  Results are the caller-order
        results up to the summation order inside a softmax row (its neighbours are visited in the internal order); `out` is
        written, G and the saved output are read, at the caller's rows through g.user_row inside the kernels."""
        if self._ordered is None:
            rp = self.fwd.rowptr.long()
            deg = rp[1:] - rp[:-1]
            to_user = torch.argsort(deg, descending=True, stable=True)
            to_int = torch.empty_like(to_user)
            to_int[to_user] = torch.arange(self.n, device=self.device)
            rows = torch.repeat_interleave(torch.arange(self.n, device=self.device), deg)
            key = torch.sort(to_int[rows] * self.n + to_int[self.fwd.col.long()]).values       # unique keys: (row, col) pairs
            r2, c2 = key // self.n, key % self.n
            rp2 = torch.zeros(self.n + 1, dtype=torch.int64, device=self.device)
            rp2[1:] = torch.cumsum(torch.bincount(r2, minlength=self.n), 0)
            urow = to_user.to(torch.int32).contiguous()
            g = CSRGraph(rp2.to(torch.int32), c2.to(torch.int32), self.slot_edges, validate=False, user_row=urow)
            g.degree_sorted = True
            self._ordered = (g, urow, to_int.to(torch.int32).contiguous())
        return self._ordered

    def internal_view(self) -> InternalOrderView:
        """The degree-ordered pattern for node arrays that are themselves in internal order (see InternalOrderView)."""
        g, to_user, to_int = self.degree_ordered()
        if getattr(g, "_view", None) is None:
            g._view = InternalOrderView(g, to_user, to_int)
        return g._view

    # ------------------------------------------------------------------ builders
    @staticmethod
    def from_dense(adj: torch.Tensor, mode: str = "nonzero", slot_edges: Optional[int] = None) -> "CSRGraph":
        """mode "nonzero": pattern adj != 0 (SpGraphAttentionLayer, layers.py:129);
        mode "positive": pattern adj > 0 (GraphAttentionLayer, layers.py:41)."""
        if adj.dim() != 2 or adj.shape[0] != adj.shape[1]:
            raise ValueError("adjacency must be square [N,N]")
        if not adj.is_cuda:
            raise ValueError("adjacency must live on the GPU (there is no CPU path)")
        a = adj if (adj.dtype == torch.float32 and adj.stride(1) == 1) else adj.float().contiguous()
        n, ld, m = a.shape[0], a.stride(0), (1 if mode == "positive" else 0)
        dev = a.device
        with torch.cuda.device(dev):
            counts = torch.empty(n, dtype=torch.int32, device=dev)
            check(lib.pygat_dense_row_counts(a.data_ptr(), n, ld, m, counts.data_ptr(), _stream()), "row_counts")
            rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
            ws = torch.empty(lib.pygat_scan_workspace_bytes(n), dtype=torch.uint8, device=dev)
            check(lib.pygat_exclusive_scan_i32(counts.data_ptr(), n, rowptr.data_ptr(), ws.data_ptr(), _stream()),
                  "scan")
            nnz = int(rowptr[-1].item())
            if nnz == 0:
                raise ValueError("adjacency has no edges")
            col = torch.empty(nnz, dtype=torch.int32, device=dev)
            check(lib.pygat_dense_fill_cols(a.data_ptr(), n, ld, m, rowptr.data_ptr(), col.data_ptr(), _stream()),
                  "fill_cols")
        return CSRGraph(rowptr, col, slot_edges)

    @staticmethod
    def block_diag(graphs: "list[CSRGraph]") -> "CSRGraph":
        """Batch of graphs as one block-diagonal pattern (replaces torch.block_diag of the dense
        adjacencies in the reference's PPI collate, load_data_ppi.py:71-88): CSR concatenation with
        row and column offsets, no N x N tensor."""
        rps, cols, noff, eoff = [], [], 0, 0
        for g in graphs:
            rp = g.fwd.rowptr.long()
            rps.append(rp[(1 if rps else 0):] + eoff)
            cols.append(g.fwd.col.long() + noff)
            noff += g.n
            eoff += g.nnz
        return CSRGraph(torch.cat(rps).to(torch.int32), torch.cat(cols).to(torch.int32))

    @staticmethod
    def from_edge_index(row: torch.Tensor, col: torch.Tensor, n: int, slot_edges: Optional[int] = None,
                        symmetrize: bool = False, self_loops: bool = False) -> "CSRGraph":
        """COO (row=i, col=j), duplicates removed, rows sorted.  symmetrize + self_loops build the pattern of
        A + A^T + I straight from a citation edge list, i.e. what the reference's loader produces through
        `adj + adj.T.multiply(adj.T > adj) - ...` and `normalize_adj(adj + sp.eye(n))` before densifying
        (utils.py:49-55); the values of that normalisation are never used by any layer (layers.py:41,129)."""
        row, col = row.long(), col.long()
        if symmetrize:
            row, col = torch.cat([row, col]), torch.cat([col, row])
        if self_loops:
            d = torch.arange(n, device=row.device)
            row, col = torch.cat([row, d]), torch.cat([col, d])
        key = torch.unique(row * n + col)
        r, c = key // n, key % n
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=row.device)
        rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
        return CSRGraph(rowptr.to(torch.int32), c.to(torch.int32), slot_edges)


# ---------------------------------------------------------------------------
# cache: the reference passes the same dense adj tensor to every head of every
# level in every epoch; convert once.
# ---------------------------------------------------------------------------
_cache: "dict[Tuple, Tuple[weakref.ref, CSRGraph]]" = {}


def as_graph(adj, mode: str = "nonzero") -> CSRGraph:
    """Accepts a CSRGraph, a (rowptr, col) pair, a torch sparse CSR tensor, or a dense [N,N] tensor."""
    if isinstance(adj, CSRGraph):
        return adj
    if isinstance(adj, (tuple, list)) and len(adj) == 2:
        return CSRGraph(adj[0], adj[1])
    if isinstance(adj, torch.Tensor) and adj.layout == torch.sparse_csr:
        return CSRGraph(adj.crow_indices(), adj.col_indices())
    if not isinstance(adj, torch.Tensor):
        raise TypeError(f"unsupported adjacency type {type(adj)}")
    key = (adj.data_ptr(), tuple(adj.shape), adj._version, str(adj.device), mode)
    hit = _cache.get(key)
    if hit is not None and hit[0]() is adj:
        return hit[1]
    g = CSRGraph.from_dense(adj, mode)
    if len(_cache) > 16:
        _cache.clear()
    _cache[key] = (weakref.ref(adj), g)
    return g
