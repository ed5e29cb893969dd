"""The parity rule of tests/parity.py, checked on the CPU against the oracle itself: it accepts the fp32 run of the
oracle, accepts a LeakyReLU branch flip at an edge whose logit sits inside the rounding band of the kink (and reports
it), and REJECTS the same flip anywhere else -- i.e. the flip-aware comparison cannot explain away a kernel that takes
the wrong branch at an ordinary edge."""
import numpy as np
import pytest

import parity
from oracle import gat_oracle as O


def _case(seed=0, N=60, H=3, Fin=12, Fo=8):
    rng = np.random.default_rng(seed)
    rowptr, col = O.random_symmetric_csr(N, 5, seed, hub=(1, 30))
    X = rng.standard_normal((N, Fin)); W = rng.standard_normal((H, Fin, Fo)) * 0.4
    a = rng.standard_normal((H, 2 * Fo)) * 0.4
    G = rng.standard_normal((N, H * Fo))
    return X, rowptr, col, W, a, G


def _as_got(r):
    return {"dX": r["dX"], "dW": r["dW"], "da": r["da"]}


def _put_edge_on_the_kink(X, rowptr, col, W, a, h, e, eps):
    """Shift a_src of head h along Wh_i so that the logit of edge e becomes eps * (|s| + |t|)."""
    Fo = W.shape[2]
    i = int(np.searchsorted(rowptr, e, side="right") - 1); j = int(col[e])
    whi, whj = X[i] @ W[h], X[j] @ W[h]
    for _ in range(30):
        s, t = whi @ a[h, :Fo], whj @ a[h, Fo:]
        want = eps * (abs(s) + abs(t))
        a[h, :Fo] += (want - (s + t)) * whi / (whi @ whi)
    return a


def test_fp32_oracle_passes_and_reports_no_flip_without_kinks():
    X, rowptr, col, W, a, G = _case()
    r32 = O.csr_layer_fwd_bwd(X.astype(np.float32), rowptr, col, W.astype(np.float32), a.astype(np.float32), 0.2, True,
                              G.astype(np.float32))
    rep = parity.check_level(r32["out"], _as_got(r32), X, rowptr, col, W, a, 0.2, True, G, what="fp32 oracle", verbose=False)
    assert rep["candidates"] == 0 and rep["hip_flips"] == []


def test_flip_inside_the_band_is_explained_and_counted():
    X, rowptr, col, W, a, G = _case(1)
    h, e = 1, 37
    a = _put_edge_on_the_kink(X, rowptr, col, W, a, h, e, 1e-7)
    flips = np.zeros((W.shape[0], len(col)), dtype=bool); flips[h, e] = True
    other = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, True, G, flips=flips)       # fp64, other branch at that edge
    rep = parity.check_level(other["out"], _as_got(other), X, rowptr, col, W, a, 0.2, True, G, what="in-band flip", verbose=False)
    assert rep["hip_flips"] == [(h, e)] and rep["candidates"] >= 1
    assert max(rep["hip"].values()) < 1e-6           # nothing but the flip (and (1 - alpha)|z| ~ 1e-7 in the forward) separates the two fp64 runs


def test_flip_outside_the_band_is_rejected():
    X, rowptr, col, W, a, G = _case(2)
    ref = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, True, G)
    rel = np.abs(ref["z"]) / ref["zscale"]
    # an ordinary edge with a sizeable gradient through it: far from the kink
    score = np.where(rel > 100 * parity.KINK_TAU, np.abs(ref["de"]), 0.0)
    h, e = np.unravel_index(np.argmax(score), score.shape)
    flips = np.zeros_like(rel, dtype=bool); flips[h, e] = True
    wrong = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, True, G, flips=flips)
    assert max(np.abs(wrong[n] - ref[n]).max() for n in ("dX", "dW", "da")) > 1e-4     # the flip is visible ...
    with pytest.raises(AssertionError):                                               # ... and nothing may explain it away
        parity.check_level(wrong["out"], _as_got(wrong), X, rowptr, col, W, a, 0.2, True, G, what="wrong branch", verbose=False)


def test_too_many_flips_are_rejected(monkeypatch):
    """The leash on the count: more than FLIP_FACTOR x the fp32 oracle's flips + FLIP_SLACK in-band flips fail."""
    X, rowptr, col, W, a, G = _case(3, H=1)
    monkeypatch.setattr(parity, "FLIP_SLACK", 1)
    monkeypatch.setattr(parity, "FLIP_FACTOR", 0)      # (the fp32 oracle run inside the rule may flip at these edges too)
    edges = [5, 50, 90]
    flips = np.zeros((1, len(col)), dtype=bool)
    # three edges of different rows / columns put on the kink one after another (each fix leaves the earlier ones
    # within the band: the shifts are along different Wh rows and tiny)
    for _ in range(200):
        for e in edges:
            a = _put_edge_on_the_kink(X, rowptr, col, W, a, 0, e, 1e-8)
    ref = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, True, G)
    rel = np.abs(ref["z"]) / ref["zscale"]
    inband = [e for e in edges if rel[0, e] <= parity.KINK_TAU]
    if len(inband) < 2:
        pytest.skip("could not put two edges on the kink at once for this seed")
    flips[0, inband] = True
    other = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, True, G, flips=flips)
    with pytest.raises(AssertionError, match="other LeakyReLU branch"):
        parity.check_level(other["out"], _as_got(other), X, rowptr, col, W, a, 0.2, True, G, what="flip count", verbose=False)
