"""Tolerance rule of the parity tests, in one place.

Ground truth is the oracle in fp64.  The HIP path computes in fp32, as the reference does, so
  forward   : |got - ref64| <= 1e-5 absolute (the north star's bar) wherever max |ref64| < 1 (the reference's
              row-normalised datasets); larger outputs (N(0,1) features: |out| up to 11, one ulp = 1e-6) take the
              gradients' rule,
  gradients : |got - ref64| <= max(1e-5, 4 x |ref32 - ref64|), ref32 = the SAME oracle run in fp32 --
              i.e. no worse than four times what the reference's own precision costs on that input
              (gradients are sums over up to 10^6 nodes and have no fixed magnitude).
ONE rule for every -m gpu test and for __graft_entry__.smoke(): `check_level` (a whole level), `close_fwd`,
`close_grad` (single tensors, GEMMs included: ref32 = the same product formed in fp32 on the CPU).
The oracle itself is PARITY UNPINNED (oracle/gat_oracle.py header).
"""
import numpy as np
import torch

ATOL = 1e-5


def _np64(t):
    if isinstance(t, torch.Tensor):
        return t.detach().double().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def err(got, ref):
    return float(np.abs(_np64(got) - _np64(ref)).max())


def close_fwd(got, ref64, what, ref32=None, atol=ATOL, factor=4.0):
    """Forward values.  max |ref64| < 1 (or no fp32 oracle run given): absolute `atol`.  Otherwise the gradients'
    rule, max(atol, factor x the fp32 oracle's own error)."""
    got, ref64 = _np64(got), _np64(ref64)
    assert got.shape == ref64.shape, f"{what}: shape {got.shape} vs {ref64.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    e = float(np.abs(got - ref64).max()) if got.size else 0.0
    big = float(np.abs(ref64).max()) if got.size else 0.0
    if ref32 is None or big < 1.0:
        assert e <= atol, f"{what}: max abs err {e:.3e} > {atol:.1e} (max |ref| {big:.3g})"
        return e
    return close_grad(got, ref64, ref32, what, factor, atol)[0]


def close_grad(got, ref64, ref32, what, factor=4.0, floor=ATOL):
    got, ref64, ref32 = _np64(got), _np64(ref64), _np64(ref32)
    assert got.shape == ref64.shape, f"{what}: shape {got.shape} vs {ref64.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    own = float(np.abs(ref32 - ref64).max()) if got.size else 0.0
    tol = max(floor, factor * own)
    e = float(np.abs(got - ref64).max()) if got.size else 0.0
    assert e <= tol, (f"{what}: max abs err {e:.3e} > {tol:.3e} = max({floor:.0e}, {factor:g} x fp32-oracle err "
                      f"{own:.3e}); max |ref| {np.abs(ref64).max():.3g}")
    return e, own


# ---------------------------------------------------------------------------------------------------
# Flip-aware comparison of one level's gradients.
#
# LeakyReLU has a kink at z_ij = s_i + t_j = 0 (layers.py:30,144).  For a logit within fp32 rounding distance
# of 0 the branch -- and with it dz_ij = de_ij * (1 or alpha) -- is decided by rounding: the reference's own fp32
# run lands on either side, and no fp32 implementation can agree with fp64 there (a single such edge moves
# gradients by 1e-4 of their maximum, 100x the rounding bar).  So the comparison first explains the residual
# by branch flips of the near-kink edges, then applies the SURVEY 8(c) rule to what is left:
#   * candidates: edges with |z| <= KINK_TAU * (|s_i| + |t_j|) in the fp64 oracle (at most KINK_MAX, nearest first).
#     KINK_TAU is a ROUNDING band: s_i and t_j are fp32 dot products over Fin (+ F') terms, each off by a few
#     1e-7 of its magnitude, so only a logit within ~1e-6..1e-5 of (|s| + |t|) can land on the other side of 0.
#     8e-6 = 67 ulp (the full-size tests use 4e-6); a kernel that took the wrong branch anywhere outside that band
#     fails, because no candidate exists to explain it;
#   * a flip of edge e = (i, j) in head h adds D = de_e * (slope' - slope) to dz_e, hence (everything downstream
#     is linear in dz):  dX_i += D a_src W_h^T, dX_j += D a_dst W_h^T, dW_h += D (X_i (x) a_src + X_j (x) a_dst),
#     da_src,h += D Wh_i, da_dst,h += D Wh_j;
#   * the 0/1 flip vector is a least-squares fit of the residual, rounded; the SAME vector must explain dX, dW
#     and da together.
#   * the leash: every accepted flip must be one of those in-band candidates (asserted on the fp64 z), and the HIP
#     path may not flip more than FLIP_FACTOR x the fp32 oracle's own flips + FLIP_SLACK -- both run fp32 arithmetic
#     on the same inputs, so their flip counts are draws from the same distribution.
# The fp32 oracle gets the same treatment, so `own` is its pure rounding error.
# ---------------------------------------------------------------------------------------------------
KINK_TAU = 8e-6
KINK_MAX = 512
FLIP_FACTOR = 2
FLIP_SLACK = 3


def _flip_leash(report, rel, tau, what):
    """rel[k] = |z64| / (|s| + |t|) of candidate k.  Asserts the leash described above; returns the log line."""
    for side in ("hip", "fp32"):
        for k in report[side + "_flip_idx"]:
            assert rel[k] <= tau, f"{what}: {side} flip at candidate {k} with |z|/(|s|+|t|) = {rel[k]:.2e} outside the band {tau:.1e}"
    nh, nf = len(report["hip_flip_idx"]), len(report["fp32_flip_idx"])
    assert nh <= FLIP_FACTOR * nf + FLIP_SLACK, (
        f"{what}: the HIP path takes the other LeakyReLU branch at {nh} of {report['candidates']} near-kink edges, the fp32 "
        f"oracle at {nf}: more than {FLIP_FACTOR} x + {FLIP_SLACK}")
    far = max([rel[k] for k in report["hip_flip_idx"]], default=0.0)
    return (f"{what}: {report['candidates']} edges within {tau:.0e} (|s|+|t|) of the kink; branch flips hip {nh} "
            f"(farthest at {far:.1e}), fp32 oracle {nf}")


def _candidate_deltas(hh, ee, zz, dee, X, W, a, rowptr, col, alpha, with_dx, Wh=None):
    """Gradient change caused by taking the other LeakyReLU branch at each candidate (head, edge)."""
    H, Fin, Fo = W.shape
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    cols = []
    for h, e, z, de in zip(hh, ee, zz, dee):
        i, j = int(np.searchsorted(rowptr, e, side="right") - 1), int(col[e])
        D = de * ((alpha - 1.0) if z > 0 else (1.0 - alpha))
        a_s, a_d = a[h, :Fo], a[h, Fo:]
        whi = Wh[h, i] if Wh is not None else X[i] @ W[h]
        whj = Wh[h, j] if Wh is not None else X[j] @ W[h]
        dW = np.zeros((H, Fin, Fo)); dW[h] = D * (np.outer(X[i], a_s) + np.outer(X[j], a_d))
        da = np.zeros((H, 2 * Fo)); da[h, :Fo] = D * whi; da[h, Fo:] = D * whj
        d = {"dW": dW, "da": da}
        if with_dx:
            dX = np.zeros((len(rowptr) - 1, Fin))
            dX[i] += D * (W[h] @ a_s); dX[j] += D * (W[h] @ a_d)
            d["dX"] = dX
        cols.append(d)
    return cols, list(zip((int(v) for v in hh), (int(v) for v in ee)))


def _kink_deltas(ref, X, W, a, rowptr, col, alpha, with_dx):
    z, zs = ref["z"], ref["zscale"]
    rel = np.abs(z) / np.maximum(zs, 1e-300)
    hh, ee = np.nonzero(rel <= KINK_TAU)
    order = np.argsort(rel[hh, ee])[:KINK_MAX]
    hh, ee = hh[order], ee[order]
    cols, cand = _candidate_deltas(hh, ee, z[hh, ee], ref["de"][hh, ee], X, W, a, rowptr, col, alpha, with_dx, ref["Wh"])
    return cols, cand, rel[hh, ee]


def _explain(resid, cols, names):
    """0/1 flip vector that best explains the residual, all tensors jointly.  Every tensor is scaled to unit maximum
    of its RESIDUAL (a tensor no candidate touches -- an upper level's parameters in a multi-level model -- then simply
    does not take part, instead of having its rounding noise blown up).  Start: rounded least squares and the empty
    set; then single toggles are accepted while they lower |resid - A sigma|^2 by at least 0.1 %: the result never
    explains less than "no flips" does."""
    if not cols:
        return np.zeros(0)
    sc = {n: 1.0 / max(float(np.abs(resid[n]).max()), 1e-300) for n in names}
    A = np.stack([np.concatenate([(c[n] * sc[n]).ravel() for n in names]) for c in cols], 1)
    b = np.concatenate([(resid[n] * sc[n]).ravel() for n in names])
    Gm, cb, bb = A.T @ A, A.T @ b, float(b @ b)
    cost = lambda s_: bb - 2.0 * float(s_ @ cb) + float(s_ @ Gm @ s_)  # noqa: E731
    lsq, *_ = np.linalg.lstsq(A, b, rcond=None)
    best, best_cost = None, None
    for start in (np.zeros(len(cols)), (lsq > 0.5).astype(np.float64)):
        s_ = start.copy()
        cur = cost(s_)
        improved = True
        while improved:
            improved = False
            gs = Gm @ s_
            for k in range(len(cols)):
                d = 1.0 - 2.0 * s_[k]                                   # +1: switch flip k on, -1: off
                delta = d * (2.0 * gs[k] - 2.0 * cb[k]) + Gm[k, k]       # change of the cost (d^2 = 1)
                if delta < -1e-3 * max(cur, 1e-300):
                    s_[k] += d
                    cur += delta
                    gs = Gm @ s_
                    improved = True
        if best is None or cur < best_cost:
            best, best_cost = s_, cur
    return best


def close_level_grads(got, X, rowptr, col, W, a, alpha, concat, G, Wskip=None, what="level", factor=4.0, floor=ATOL):
    """got = dict(dX (or None), dW [H,Fin,F], da [H,2F][, dW_skip]) from the HIP path; X, W, a, G, Wskip are the
    fp64 arrays whose fp32 roundings were fed to it.  Returns a report dict."""
    from oracle import gat_oracle as O
    X = np.asarray(X, np.float64); W = np.asarray(W, np.float64); a = np.asarray(a, np.float64); G = np.asarray(G, np.float64)
    Sk = None if Wskip is None else np.asarray(Wskip, np.float64)
    f32 = lambda v: None if v is None else v.astype(np.float32)  # noqa: E731
    ref64 = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, alpha, concat, G, Sk)
    ref32 = O.csr_layer_fwd_bwd(f32(X), rowptr, col, f32(W), f32(a), alpha, concat, f32(G), f32(Sk))
    with_dx = got.get("dX") is not None
    names = (["dX"] if with_dx else []) + ["dW", "da"]
    cols, cand, rel = _kink_deltas(ref64, X, W, a, rowptr, col, alpha, with_dx)
    report = {"candidates": len(cand)}
    for side, vals in (("hip", {n: _np64(got[n]).reshape(ref64[n].shape) for n in names}),
                       ("fp32", {n: np.asarray(ref32[n], np.float64) for n in names})):
        resid = {n: vals[n] - ref64[n] for n in names}
        sig = _explain(resid, cols, names)
        for k, c in enumerate(cols):
            if sig[k]:
                for n in names:
                    resid[n] = resid[n] - c[n]
        report[side] = {n: float(np.abs(resid[n]).max()) for n in names}
        report[side + "_flips"] = [cand[k] for k in range(len(cand)) if sig[k]]
        report[side + "_flip_idx"] = [k for k in range(len(cand)) if sig[k]]
    report["flips"] = _flip_leash(report, rel, KINK_TAU, what)
    for n in names:
        assert np.isfinite(_np64(got[n])).all(), f"{what} {n}: non-finite values"
        tol = max(floor, factor * report["fp32"][n])
        assert report["hip"][n] <= tol, (
            f"{what} {n}: max abs err {report['hip'][n]:.3e} > {tol:.3e} = max({floor:.0e}, {factor:g} x fp32-oracle err "
            f"{report['fp32'][n]:.3e}) after {len(report['hip_flips'])} LeakyReLU branch flips "
            f"({len(cand)} near-kink edges); max |ref| {np.abs(ref64[n]).max():.3g}")
    if Wskip is not None and got.get("dW_skip") is not None:        # no kink on this path
        close_grad(got["dW_skip"], ref64["dW_skip"], ref32["dW_skip"], f"{what} dW_skip", factor, floor)
    report["ref64"] = ref64
    report["ref32"] = ref32
    return report


def close_fullsize_grads(got, r64, r32, X, W, a, rowptr, col, alpha, what="level", factor=4.0, floor=ATOL, names=("dW", "da"),
                         tau=KINK_TAU):
    """The same flip-aware rule where the python oracle cannot run (10^7 edges): r64 / r32 come from the two builds of
    oracle/gat_oracle.c, r64["kinks"] lists the near-kink edges of the fp64 run (c_oracle.level(kink_tau=...)).  Only
    the parameter gradients take part in the fit when dX is absent (a first level); with got["dX"] the rows of dX the candidates
    touch join it."""
    X = np.asarray(X, np.float64); W = np.asarray(W, np.float64); a = np.asarray(a, np.float64)
    k = r64["kinks"]
    cols, cand = _candidate_deltas(k["h"], k["e"], k["z"], k["de"], X, W, a, rowptr, col, alpha, False)
    # |z| / (|s_i| + |t_j|) of every candidate, recomputed here in fp64 (the C oracle selected them by that ratio)
    rp64 = np.asarray(rowptr, dtype=np.int64); Fo = W.shape[2]
    rel = np.zeros(len(cand))
    for q, (h, e) in enumerate(cand):
        i, j = int(np.searchsorted(rp64, e, side="right") - 1), int(col[e])
        sc = abs(float(X[i] @ W[h] @ a[h, :Fo])) + abs(float(X[j] @ W[h] @ a[h, Fo:]))
        rel[q] = abs(float(k["z"][q])) / max(sc, 1e-300)
    report = {"candidates": len(cand)}
    # dX takes part in the FIT (round 5), through the rows the candidates touch: a flip moves two rows of dX by D W_h a_src /
    # D W_h a_dst -- ~1e-3 for a D that is invisible in dW (max 3845, fp32 noise 3e-3) -- so a flip vector chosen by the parameter
    # gradients alone is arbitrary exactly where dX is sensitive (config-5 graph drawn from the numpy stream: dX raw 1.3e-4, after
    # the parameter gradients' 33 flips 5.3e-4).  The untouched rows of dX do not depend on the flips and are priced afterwards.
    with_dx = got.get("dX") is not None and r64.get("dX") is not None
    fit_names, touched = tuple(names), None
    if with_dx and len(cand):
        Fo_ = W.shape[2]
        ij = [(int(np.searchsorted(rp64, e, side="right") - 1), int(col[e])) for _, e in cand]
        touched = np.unique(np.asarray(ij).ravel())
        pos = {int(r): q for q, r in enumerate(touched)}
        for q, c in enumerate(cols):
            h, _e = cand[q]
            i, j = ij[q]
            D = float(k["de"][q]) * ((alpha - 1.0) if float(k["z"][q]) > 0 else (1.0 - alpha))
            dxt = np.zeros((len(touched), X.shape[1]))
            dxt[pos[i]] += D * (W[h] @ a[h, :Fo_]); dxt[pos[j]] += D * (W[h] @ a[h, Fo_:])
            c["dXT"] = dxt
        fit_names = tuple(names) + ("dXT",)
    for side, vals in (("hip", {n: _np64(got[n]).reshape(r64[n].shape) for n in names}),
                       ("fp32", {n: np.asarray(r32[n], np.float64) for n in names})):
        resid = {n: vals[n] - r64[n] for n in names}
        report[side + "_raw"] = {n: float(np.abs(resid[n]).max()) for n in names}
        if touched is not None:
            v = _np64(got["dX"]) if side == "hip" else np.asarray(r32["dX"], np.float64)
            resid["dXT"] = v.reshape(r64["dX"].shape)[touched] - r64["dX"][touched]
        sig = _explain(resid, cols, fit_names)
        resid.pop("dXT", None)
        for q, c in enumerate(cols):
            if sig[q]:
                for n in names:
                    resid[n] = resid[n] - c[n]
        report[side] = {n: float(np.abs(resid[n]).max()) for n in names}
        report[side + "_flips"] = [cand[q] for q in range(len(cand)) if sig[q]]
        report[side + "_flip_idx"] = [q for q in range(len(cand)) if sig[q]]
    report["flips"] = _flip_leash(report, rel, tau, what)
    if with_dx:
        # dX [N, Fin] at full size: a candidate's effect on it is two rows (dX_i += D W_h a_src, dX_j += D W_h a_dst), kept
        # sparse; ONE flip vector (fitted above on dW, da and the touched rows of dX together) has to explain all three
        Fo_ = W.shape[2]
        for side, v in (("hip", _np64(got["dX"])), ("fp32", np.asarray(r32["dX"], np.float64))):
            res = v.reshape(r64["dX"].shape) - r64["dX"]
            report[side + "_raw"]["dX"] = float(np.abs(res).max())
            for q in report[side + "_flip_idx"]:
                h, e = cand[q]
                i, j = int(np.searchsorted(rp64, e, side="right") - 1), int(col[e])
                D = float(k["de"][q]) * ((alpha - 1.0) if float(k["z"][q]) > 0 else (1.0 - alpha))
                res[i] -= D * (W[h] @ a[h, :Fo_]); res[j] -= D * (W[h] @ a[h, Fo_:])
            report[side]["dX"] = float(np.abs(res).max())
            del res
        names = tuple(names) + ("dX",)
    for n in names:
        assert np.isfinite(_np64(got[n])).all(), f"{what} {n}: non-finite values"
        tol = max(floor, factor * report["fp32"][n])
        assert report["hip"][n] <= tol, (
            f"{what} {n}: max abs err {report['hip'][n]:.3e} (raw {report['hip_raw'][n]:.3e}) > {tol:.3e} = max({floor:.0e}, "
            f"{factor:g} x fp32-oracle err {report['fp32'][n]:.3e}) after {len(report['hip_flips'])} LeakyReLU branch flips "
            f"({len(cand)} near-kink edges); max |ref| {np.abs(r64[n]).max():.3g}")
    return report


def check_level(out, grads, X, rowptr, col, W, a, alpha, concat, G, Wskip=None, what="level", verbose=True):
    """A whole level against the oracle under the one rule: `out` by close_fwd, grads = dict(dX|None, dW, da[, dW_skip])
    by the flip-aware close_level_grads.  X, W, a, G, Wskip: the fp64 arrays whose fp32 roundings the HIP path got."""
    rep = close_level_grads(grads, X, rowptr, col, W, a, alpha, concat, G, Wskip, what)
    rep["out_err"] = close_fwd(out, rep["ref64"]["out"], f"{what} out", rep["ref32"]["out"])
    if verbose:
        names = [n for n in ("dX", "dW", "da") if n in rep["hip"]]
        print(f"{rep['flips']}; out {rep['out_err']:.2e}; "
              + ", ".join(f"{n} {rep['hip'][n]:.2e} (fp32 oracle {rep['fp32'][n]:.2e})" for n in names))
    return rep


def close_model_grads(got, oracle_run, kinks, what="model", factor=4.0, floor=ATOL):
    """End-to-end gradients of a multi-level model under the same flip-aware rule.

    got: {name: tensor}; oracle_run(dtype, flips) -> {name: array} runs the oracle end to end (autograd through
    oracle.model_forward) with per-level LeakyReLU branch overrides `flips` (list of None | [H,E] bool);
    kinks: [(level, head, edge, |z|/(|s|+|t|))] = the in-band near-kink edges of the fp64 run (oracle.model_logits_z).
    The effect of one flip on every gradient is the difference of two fp64 oracle runs (it is carried densely
    through the levels below, so it cannot be written per edge as in close_level_grads); the 0/1 fit picks the
    flips; the residual is then taken against the fp64 oracle run WITH exactly those branches -- an exact reference
    for that branch pattern, no linearisation left -- and priced by the 8(c) rule.  Same leash as above."""
    names = sorted(got)
    base64 = {n: _np64(v) for n, v in oracle_run(torch.float64, None).items()}
    vals32 = {n: _np64(v) for n, v in oracle_run(torch.float32, None).items()}
    kinks = sorted(kinks, key=lambda k: k[3])[:16]
    rel = np.array([k[3] for k in kinks])
    nlev = 1 + max([k[0] for k in kinks], default=0)

    def flipset(sel, shapes):
        fl = [None] * len(shapes)
        for q in sel:
            lv, h, e, _ = kinks[q]
            if fl[lv] is None:
                fl[lv] = np.zeros(shapes[lv], dtype=bool)
            fl[lv][h, e] = True
        return fl

    shapes = oracle_run.flip_shapes
    cols = []
    for q in range(len(kinks)):
        r = oracle_run(torch.float64, flipset([q], shapes))
        cols.append({n: _np64(r[n]) - base64[n] for n in names})
    report = {"candidates": len(kinks)}
    for side, vals in (("hip", {n: _np64(got[n]).reshape(base64[n].shape) for n in names}), ("fp32", vals32)):
        resid = {n: vals[n] - base64[n] for n in names}
        sig = _explain(resid, cols, names)
        sel = [q for q in range(len(kinks)) if sig[q]]
        truth = base64 if not sel else {n: _np64(v) for n, v in oracle_run(torch.float64, flipset(sel, shapes)).items()}
        report[side] = {n: float(np.abs(vals[n] - truth[n]).max()) for n in names}
        report[side + "_raw"] = {n: float(np.abs(resid[n]).max()) for n in names}
        report[side + "_flip_idx"] = sel
        report[side + "_flips"] = [kinks[q][:3] for q in sel]
    report["flips"] = _flip_leash(report, rel, KINK_TAU, what)
    for n in names:
        assert np.isfinite(_np64(got[n])).all(), f"{what} {n}: non-finite values"
        tol = max(floor, factor * report["fp32"][n])
        assert report["hip"][n] <= tol, (
            f"{what} {n}: max abs err {report['hip'][n]:.3e} (raw {report['hip_raw'][n]:.3e}) > {tol:.3e} = max({floor:.0e}, "
            f"{factor:g} x fp32-oracle err {report['fp32'][n]:.3e}); {report['flips']}; max |ref| {np.abs(base64[n]).max():.3g}")
    return report


def check_autograd(got_out, got_grads, fn, leaves64, G64, names, what="level"):
    """For the cases whose oracle is evaluated through torch autograd (explicit dropout masks, GATv2): `fn(*leaves)`
    is run in fp64 (ground truth) and in fp32 (the reference's own precision) and the HIP results are priced by the
    same rule: out by close_fwd, every gradient by close_grad.  got_grads / names follow `leaves64`; a leaf the
    oracle does not use (autograd returns None) must come back as exact zeros."""
    def run(dtype):
        lv = [t.detach().to(dtype).clone().requires_grad_(True) for t in leaves64]
        y = fn(*lv)
        gr = torch.autograd.grad(y, lv, G64.to(dtype), allow_unused=True)
        return y.detach(), gr
    y64, g64 = run(torch.float64)
    y32, g32 = run(torch.float32)
    rep = {"out": close_fwd(got_out, y64, f"{what} out", y32)}
    for name, got, r64, r32 in zip(names, got_grads, g64, g32):
        if got is None:
            continue
        if r64 is None:
            assert float(got.abs().max()) == 0.0, f"{what} {name}: the oracle's gradient is identically zero"
            continue
        rep[name] = close_grad(got, r64.reshape(got.shape), r32.reshape(got.shape), f"{what} {name}")[0]
    return rep, (y64, g64)
