"""CPU oracle for the GAT attention layer hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product (pygat_amd/) never does; it fails loudly when the HIP
library is missing.

PARITY UNPINNED.  The reference (ArielleRosinski/pyGAT) ships no tests, golden
vectors or fixtures for this path, and its module `layers.py` cannot be
imported in the build container: `layers.py:5` imports `torch_scatter`
(rusty1s/pytorch_scatter, not pinned in the reference's requirements.txt),
which is not installed and stays absent.  This file is therefore a restatement,
in our own words, of the algorithm as read from the reference source:

  GATv2 variants      layers.py:179-232 (dense V2, incl. its row-broadcast of the
                      logits = neighbour mean), layers.py:234-316 (sparse V2)
  dense formulation   layers.py:32-64   (GraphAttentionLayer.forward and
                                         _prepare_attentional_mechanism_input)
  sparse formulation  layers.py:125-173 (SpGraphAttentionLayer.forward) with
                      layers.py:72-90   (SpecialSpmmFunction fwd/bwd) and the
                      published semantics of torch_scatter.scatter_max
                      (out[i] = max over {src[k] : index[k] == i})
  multi-head model    models.py:29-35   (concat for hidden levels, mean of the
                                         stacked heads for the last level)

What pins it instead (tests/test_oracle.py): the two formulations agree with
each other; the hand-derived CSR gradients (`csr_layer_fwd_bwd`) agree with
stock torch autograd run through the dense formulation in fp64; gradcheck.

All functions take/return CPU torch tensors; dtype follows the inputs
(fp32 like the reference, or fp64 for ground truth).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# single head, dense formulation -- layers.py:32-64
# --------------------------------------------------------------------------
def dense_head_forward(h, adj, W, a, alpha, concat, W_skip=None,
                       mask_x=None, mask_wh=None, mask_att=None):
    """One head of GraphAttentionLayer.

    h [N,Fin]; adj [N,N] (only `adj > 0` is used, layers.py:41); W [Fin,F];
    a [2F,1] (layers.py:23); W_skip [Fin,F] or None (layers.py:26-28).
    mask_* are pre-scaled dropout masks (0 or 1/(1-p)) standing in for the
    three F.dropout calls at layers.py:34,37,43; None == eval mode.
    """
    Fo = W.shape[1]
    if mask_x is not None:                       # layers.py:34
        h = h * mask_x
    Wh = h @ W                                   # layers.py:35
    if mask_wh is not None:                      # layers.py:37
        Wh = Wh * mask_wh
    a = a.reshape(-1)
    s = Wh @ a[:Fo]                              # layers.py:60
    t = Wh @ a[Fo:]                              # layers.py:61
    e = F.leaky_relu(s[:, None] + t[None, :], alpha)      # layers.py:63-64
    att = torch.where(adj > 0, e, torch.full_like(e, -9e15))  # layers.py:40-41
    att = torch.softmax(att, dim=1)              # layers.py:42
    if mask_att is not None:                     # layers.py:43 (dense [N,N] mask)
        att = att * mask_att
    hp = att @ Wh                                # layers.py:44
    if W_skip is not None:                       # layers.py:47-48 (h is post-dropout)
        hp = hp + h @ W_skip
    return F.elu(hp) if concat else hp           # layers.py:50-53


# --------------------------------------------------------------------------
# single head, sparse (edge-list) formulation -- layers.py:125-173
# --------------------------------------------------------------------------
def _leaky(z, alpha, flip=None):
    """LeakyReLU (layers.py:30,144).  `flip` [E] bool marks edges whose branch is taken on the OTHER side of the
    kink: for a logit within fp32 rounding distance of 0 the side -- and with it the derivative, 1 or alpha -- is
    decided by rounding, so both sides are legitimate fp32 results (tests/parity.py, flip-aware comparison).
    The forward value moves by at most (1 - alpha)|z|, i.e. nothing at that distance."""
    if flip is None:
        return F.leaky_relu(z, alpha)
    pos = (z > 0) ^ torch.as_tensor(flip, dtype=torch.bool)
    return torch.where(pos, z, alpha * z)


def sparse_head_forward(h, rowptr, col, W, a, alpha, concat, W_skip=None,
                        mask_x=None, mask_wh=None, mask_edge=None, slope_flip=None):
    """One head of SpGraphAttentionLayer over a CSR pattern.

    The reference derives `edge = adj.nonzero().t()` (layers.py:129), which is
    row-major, i.e. exactly CSR order; edge[0] = row (i), edge[1] = col (j).
    a is [1,2F] here (layers.py:114).  mask_edge [E] stands in for the dropout
    on the numerators AFTER the row sum is taken (layers.py:150-153).
    """
    N = h.shape[0]
    rowptr = torch.as_tensor(rowptr, dtype=torch.int64)
    col = torch.as_tensor(col, dtype=torch.int64)
    deg = rowptr[1:] - rowptr[:-1]
    src = torch.repeat_interleave(torch.arange(N), deg)   # edge[0]
    Fo = W.shape[1]
    if mask_x is not None:                       # layers.py:132
        h = h * mask_x
    Wh = h @ W                                   # layers.py:134
    if mask_wh is not None:                      # layers.py:136
        Wh = Wh * mask_wh
    a = a.reshape(-1)
    # layers.py:141-144: a . [Wh_i ; Wh_j]  ==  a[:F].Wh_i + a[F:].Wh_j
    edge_e = _leaky((Wh @ a[:Fo])[src] + (Wh @ a[Fo:])[col], alpha, slope_flip)
    # layers.py:145: torch_scatter.scatter_max(edge_e, edge[0]) -> per-row max
    m = torch.full((N,), -float("inf"), dtype=h.dtype).scatter_reduce(
        0, src, edge_e.detach(), "amax", include_self=True)
    p = torch.exp(edge_e - m[src])               # layers.py:146
    Z = torch.zeros(N, dtype=h.dtype).index_add(0, src, p)   # layers.py:150
    if mask_edge is not None:                    # layers.py:153
        p = p * mask_edge
    hp = torch.zeros(N, Fo, dtype=h.dtype).index_add(0, src, p[:, None] * Wh[col])  # layers.py:156
    hp = hp / Z[:, None]                         # layers.py:160
    if W_skip is not None:                       # layers.py:165-166
        hp = hp + h @ W_skip
    return F.elu(hp) if concat else hp           # layers.py:168-173


# --------------------------------------------------------------------------
# GATv2 variants as the reference WRITES them -- layers.py:179-232 (dense), 234-316 (sparse)
# --------------------------------------------------------------------------
def dense_head_forward_v2(h, adj, W, a, alpha, concat, W_skip=None,
                          mask_x=None, mask_wh1=None, mask_wh2=None, mask_att=None):
    """GraphAttentionLayerV2.forward.  W [2Fin,F] (layers.py:192), a [F,1] (layers.py:194).
    NOTE (faithful to the reference): e is [N,1] (layers.py:214) and broadcasts
    along each ROW in torch.where (layers.py:217), so every neighbour of i gets the same logit and
    the softmax is uniform: the layer is a neighbour-MEAN of Wh2."""
    Fin = W.shape[0] // 2
    if mask_x is not None:                       # layers.py:206
        h = h * mask_x
    Wh1 = h @ W[:Fin]                            # layers.py:208
    Wh2 = h @ W[Fin:]                            # layers.py:209
    if mask_wh1 is not None:                     # layers.py:211-212
        Wh1 = Wh1 * mask_wh1
    if mask_wh2 is not None:
        Wh2 = Wh2 * mask_wh2
    e = F.leaky_relu(Wh1 + Wh2, alpha) @ a.reshape(-1, 1)          # layers.py:213-215 -> [N,1]
    att = torch.where(adj > 0, e, torch.full_like(adj, -9e15))      # layers.py:217-218 (row broadcast)
    att = torch.softmax(att, dim=1)              # layers.py:219
    if mask_att is not None:                     # layers.py:220
        att = att * mask_att
    hp = att @ Wh2                               # layers.py:221
    if W_skip is not None:                       # layers.py:224-225
        hp = hp + h @ W_skip
    return F.elu(hp) if concat else hp           # layers.py:227-230


def sparse_head_forward_v2(h, rowptr, col, W, a, alpha, concat, W_skip=None,
                           mask_x=None, mask_whi=None, mask_whj=None, mask_edge=None):
    """SpGraphAttentionLayerV2.forward.  W [2Fin,F] (layers.py:246), a [1,F] (layers.py:249).
    Scores e_ij = a . LeakyReLU(Whi_i + Whj_j) (layers.py:280-283); NOTE the aggregation gathers
    Whi at the neighbour, `special_spmm(edge, edge_e, ., Whi)` (layers.py:296), not Whj."""
    N = h.shape[0]
    Fin = W.shape[0] // 2
    rowptr = torch.as_tensor(rowptr, dtype=torch.int64)
    col = torch.as_tensor(col, dtype=torch.int64)
    src = torch.repeat_interleave(torch.arange(N), rowptr[1:] - rowptr[:-1])
    if mask_x is not None:                       # layers.py:266
        h = h * mask_x
    Whi = h @ W[:Fin]                            # layers.py:268
    Whj = h @ W[Fin:]                            # layers.py:269
    if mask_whi is not None:                     # layers.py:271-272
        Whi = Whi * mask_whi
    if mask_whj is not None:
        Whj = Whj * mask_whj
    edge_h = Whi[src] + Whj[col]                 # layers.py:280
    edge_e = F.leaky_relu(edge_h, alpha) @ a.reshape(-1)           # layers.py:283
    m = torch.full((N,), -float("inf"), dtype=h.dtype).scatter_reduce(
        0, src, edge_e.detach(), "amax", include_self=True)        # layers.py:285 scatter_max
    p = torch.exp(edge_e - m[src])               # layers.py:286
    Z = torch.zeros(N, dtype=h.dtype).index_add(0, src, p)         # layers.py:290
    if mask_edge is not None:                    # layers.py:293
        p = p * mask_edge
    hp = torch.zeros(N, Whi.shape[1], dtype=h.dtype).index_add(0, src, p[:, None] * Whi[col])  # layers.py:296
    hp = hp / Z[:, None]                         # layers.py:300
    if W_skip is not None:                       # layers.py:305-306
        hp = hp + h @ W_skip
    return F.elu(hp) if concat else hp           # layers.py:308-313


def level_forward_v2(x, graph, Ws, As, alpha, concat, W_skips=None, formulation="sparse"):
    """All heads of a V2 level (models.py:29-35).  Ws [H,2Fin,F], As [H,F]."""
    outs = []
    for hd in range(Ws.shape[0]):
        sk = None if W_skips is None else W_skips[hd]
        if formulation == "dense":
            outs.append(dense_head_forward_v2(x, graph, Ws[hd], As[hd], alpha, concat, sk))
        else:
            outs.append(sparse_head_forward_v2(x, graph[0], graph[1], Ws[hd], As[hd], alpha, concat, sk))
    if concat:
        return torch.cat(outs, dim=1)
    return torch.mean(torch.stack(outs, dim=1), dim=1)


# --------------------------------------------------------------------------
# one GAT level = all heads of one layer -- models.py:29-35
# --------------------------------------------------------------------------
def level_forward(x, graph, Ws, As, alpha, concat, W_skips=None,
                  formulation="sparse", masks=None, flips=None):
    """All heads of one level.  Ws [H,Fin,F], As [H,2F], W_skips [H,Fin,F]|None.

    concat=True  -> hidden level: cat(heads, dim=1), each head ELU'd (models.py:32)
    concat=False -> last level: mean(stack(heads, 1), 1), no ELU     (models.py:34)
    graph: (rowptr, col) for "sparse", dense adj for "dense".
    masks: optional dict of per-head pre-scaled masks {"x":[H,N,Fin],
           "wh":[H,N,F], "att":[H,E] (sparse) or [H,N,N] (dense)}.
    flips: optional [H,E] bool, LeakyReLU branch overrides (sparse formulation only, see `_leaky`).
    """
    outs = []
    for hd in range(Ws.shape[0]):
        mk = {} if masks is None else {k: v[hd] for k, v in masks.items() if v is not None}
        sk = None if W_skips is None else W_skips[hd]
        if formulation == "dense":
            o = dense_head_forward(x, graph, Ws[hd], As[hd].reshape(-1, 1), alpha, concat, sk,
                                   mk.get("x"), mk.get("wh"), mk.get("att"))
        else:
            o = sparse_head_forward(x, graph[0], graph[1], Ws[hd], As[hd].reshape(1, -1), alpha,
                                    concat, sk, mk.get("x"), mk.get("wh"), mk.get("att"),
                                    None if flips is None else flips[hd])
        outs.append(o)
    if concat:
        return torch.cat(outs, dim=1)
    return torch.mean(torch.stack(outs, dim=1), dim=1)


def model_forward(x, graph, levels, alpha, formulation="sparse", flips=None):
    """models.GAT.forward (eval mode): `levels` is a list of dicts
    {"W":[H,Fin,F], "a":[H,2F], "skip":[H,Fin,F]|None}; every level but the
    last concatenates (models.py:23,30-34)."""
    for li, lv in enumerate(levels):
        x = level_forward(x, graph, lv["W"], lv["a"], alpha, li < len(levels) - 1,
                          lv.get("skip"), formulation, flips=None if flips is None else flips[li])
    return x


def model_logits_z(x, graph, levels, alpha):
    """Per level, the pre-activation logits z_ij = s_i + t_j of every head and edge, [H,E], and their rounding
    scale |s_i| + |t_j| (eval mode, no grad): what the flip-aware comparison screens for kinks."""
    rowptr = torch.as_tensor(graph[0], dtype=torch.int64); col = torch.as_tensor(graph[1], dtype=torch.int64)
    src = torch.repeat_interleave(torch.arange(x.shape[0]), rowptr[1:] - rowptr[:-1])
    out = []
    with torch.no_grad():
        for li, lv in enumerate(levels):
            Fo = lv["W"].shape[2]
            Wh = torch.einsum("nk,hkf->hnf", x, lv["W"])
            s = torch.einsum("hnf,hf->hn", Wh, lv["a"][:, :Fo]); t = torch.einsum("hnf,hf->hn", Wh, lv["a"][:, Fo:])
            out.append((s[:, src] + t[:, col], s[:, src].abs() + t[:, col].abs()))
            x = level_forward(x, graph, lv["W"], lv["a"], alpha, li < len(levels) - 1, lv.get("skip"), "sparse")
    return out


# --------------------------------------------------------------------------
# hand-derived CSR forward+backward (the math the HIP kernels implement)
# --------------------------------------------------------------------------
def csr_layer_fwd_bwd(X, rowptr, col, Ws, As, alpha, concat, G, W_skips=None, flips=None):
    """Multi-head level, eval mode / dropout 0, explicit gradients (numpy).

    Follows the same forward as sparse_head_forward; the backward is the chain
    rule through layers.py:134-170 and SpecialSpmmFunction.backward
    (layers.py:81-90) written per edge, never forming N x N:
        dp_ij = G'_i . Wh_j ; D_i = sum_j alpha_ij dp_ij
        de_ij = alpha_ij (dp_ij - D_i) ; dz_ij = de_ij * (z_ij > 0 ? 1 : alpha)
        ds_i = sum_j dz_ij ; dt_j = sum_i dz_ij
        dWh_j = sum_i alpha_ij G'_i + ds_j a_src + dt_j a_dst
        da_src = sum_i ds_i Wh_i ; da_dst = sum_j dt_j Wh_j
        dW = X^T dWh ; dX = dWh W^T (+ G' W_skip^T) ; dW_skip = X^T G'
    G is dL/d(out) with out [N,H*F] (concat) or [N,F] (mean).
    flips: optional [H,E] bool, LeakyReLU branch overrides (see `_leaky`).
    Returns dict(out, dX, dW [H,Fin,F], da [H,2F], dW_skip) plus what the flip-aware comparison of
    tests/parity.py needs: z [H,E], zscale [H,E] = |s_i| + |t_j|, de [H,E] (dL/de_ij before the LeakyReLU slope),
    Wh [H,N,F].
    """
    X = np.asarray(X); dt_ = X.dtype
    rowptr = np.asarray(rowptr, dtype=np.int64); col = np.asarray(col, dtype=np.int64)
    Ws = np.asarray(Ws, dtype=dt_); As = np.asarray(As, dtype=dt_); G = np.asarray(G, dtype=dt_)
    N = X.shape[0]; H, Fin, Fo = Ws.shape
    deg = np.diff(rowptr); src = np.repeat(np.arange(N), deg)
    outs = []; dX = np.zeros_like(X); dW = np.zeros_like(Ws); dA = np.zeros_like(As)
    dSk = None if W_skips is None else np.zeros_like(np.asarray(W_skips, dtype=dt_))
    zs, zscales, des, Whs = [], [], [], []
    for h in range(H):
        W = Ws[h]; a_s = As[h, :Fo]; a_d = As[h, Fo:]
        Wh = X @ W
        s = Wh @ a_s; t = Wh @ a_d
        z = s[src] + t[col]
        zs.append(z); zscales.append(np.abs(s[src]) + np.abs(t[col]))
        pos = (z > 0) if flips is None else ((z > 0) ^ np.asarray(flips[h], dtype=bool))
        e = np.where(pos, z, alpha * z)
        m = np.full(N, -np.inf, dtype=dt_); np.maximum.at(m, src, e)
        p = np.exp(e - m[src])
        Z = np.zeros(N, dtype=dt_); np.add.at(Z, src, p)
        al = p / Z[src]
        hp = np.zeros((N, Fo), dtype=dt_); np.add.at(hp, src, al[:, None] * Wh[col])
        pre = hp if W_skips is None else hp + X @ np.asarray(W_skips[h], dtype=dt_)
        out = np.where(pre > 0, pre, np.expm1(np.minimum(pre, 0))) if concat else pre
        outs.append(out)
        # ---- backward
        Gh = G[:, h * Fo:(h + 1) * Fo] if concat else G / H
        Gp = Gh * np.where(pre > 0, 1.0, np.exp(np.minimum(pre, 0))).astype(dt_) if concat else Gh
        dp = np.einsum("ef,ef->e", Gp[src], Wh[col])
        D = np.zeros(N, dtype=dt_); np.add.at(D, src, al * dp)
        de = al * (dp - D[src])
        des.append(de); Whs.append(Wh)
        dz = de * np.where(pos, 1.0, alpha).astype(dt_)
        ds = np.zeros(N, dtype=dt_); np.add.at(ds, src, dz)
        dtt = np.zeros(N, dtype=dt_); np.add.at(dtt, col, dz)
        dWh = np.zeros((N, Fo), dtype=dt_); np.add.at(dWh, col, al[:, None] * Gp[src])
        dWh += ds[:, None] * a_s[None, :] + dtt[:, None] * a_d[None, :]
        dA[h, :Fo] = ds @ Wh; dA[h, Fo:] = dtt @ Wh
        dW[h] = X.T @ dWh
        dX += dWh @ W.T
        if W_skips is not None:
            Sk = np.asarray(W_skips[h], dtype=dt_)
            dSk[h] = X.T @ Gp
            dX += Gp @ Sk.T
    out = np.concatenate(outs, 1) if concat else np.mean(np.stack(outs, 1), 1)
    return dict(out=out, dX=dX, dW=dW, da=dA, dW_skip=dSk, z=np.stack(zs), zscale=np.stack(zscales),
                de=np.stack(des), Wh=np.stack(Whs))


# --------------------------------------------------------------------------
# the citation scripts' training loss -- train.py:151-152,159
# --------------------------------------------------------------------------
def train_loss(out, idx, labels):
    """output = F.log_softmax(F.elu(model(features, adj)), dim=1)   (train.py:151-152)
    loss = F.nll_loss(output[idx], labels[idx])                       (train.py:159; idx_val / idx_test at 169, 186)"""
    idx = torch.as_tensor(idx, dtype=torch.int64)
    return F.nll_loss(F.log_softmax(F.elu(out), dim=1)[idx], torch.as_tensor(labels, dtype=torch.int64)[idx])


# --------------------------------------------------------------------------
# graph helpers used by tests and by bench.py's CPU leg
# --------------------------------------------------------------------------
def dense_from_csr(rowptr, col, N, dtype=torch.float32):
    adj = torch.zeros(N, N, dtype=dtype)
    deg = np.diff(np.asarray(rowptr))
    src = np.repeat(np.arange(N), deg)
    adj[torch.as_tensor(src), torch.as_tensor(np.asarray(col, dtype=np.int64))] = 1.0
    return adj


def random_symmetric_csr(N, avg_deg, seed, hub=None):
    """Random symmetric pattern + self loops (the shape utils.py:49-52 produces).
    hub: optional (node, degree) to force one heavy row."""
    rng = np.random.default_rng(seed)
    n_e = max(1, int(N * avg_deg / 2))
    r = rng.integers(0, N, n_e); c = rng.integers(0, N, n_e)
    if hub is not None:
        hn, hd = hub
        nb = rng.choice(N, size=min(hd, N), replace=False)
        r = np.concatenate([r, np.full(nb.shape, hn)]); c = np.concatenate([c, nb])
    rr = np.concatenate([r, c, np.arange(N)]); cc = np.concatenate([c, r, np.arange(N)])
    key = np.unique(rr.astype(np.int64) * N + cc)
    rr = (key // N).astype(np.int32); cc = (key % N).astype(np.int32)
    rowptr = np.zeros(N + 1, dtype=np.int32); np.add.at(rowptr, rr + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    return rowptr, cc
