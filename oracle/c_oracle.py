"""ctypes wrapper of oracle/gat_oracle.c (TEST INFRASTRUCTURE / cpu_baseline only; parity unpinned,
see gat_oracle.py).  Built by `make -C oracle` (also from __graft_entry__.build())."""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libgat_oracle.so")


def build_for_host():
    """gat_oracle.c is compiled -march=native, so the library built in the build container is rebuilt once on another
    host (the GPU box).  Callers (tests/conftest.py at collection time, bench.py at start-up) run this BEFORE their
    process initialises the GPU: no child process is ever started from one that holds the device."""
    import subprocess
    stamp = os.path.join(_DIR, "_build", ".host")
    try:
        host = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except (OSError, StopIteration):
        host = "unknown"
    try:
        same = open(stamp).read() == host and os.path.exists(_SO)
    except OSError:
        same = False
    subprocess.run(["make", "-C", _DIR] + ([] if same else ["-B"]), check=True, capture_output=True)
    if not same:
        open(stamp, "w").write(host)


def load():
    lib = C.CDLL(_SO)
    p = C.c_void_p
    lib.gat_oracle_level.argtypes = [C.c_int64, C.c_int64, p, p, p, p, p, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_int, p, p, p, p, p, p, p, p]
    lib.gat_oracle_level.restype = C.c_int
    lib.gat_oracle_level_f64.argtypes = [C.c_int64, C.c_int64, p, p, p, p, p, C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_int, p, p, p, p, p, p, p, p]
    lib.gat_oracle_level_f64.restype = C.c_int
    for fn, real in ((lib.gat_oracle_level_v2, C.c_float), (lib.gat_oracle_level_v2_f64, C.c_double)):
        fn.argtypes = [C.c_int64, C.c_int64, p, p, p, p, p, C.c_int, C.c_int, C.c_int, real, C.c_int, p, p, p, p, p, p, p, p]
        fn.restype = C.c_int
    lib.gat_oracle_threads.restype = C.c_int
    for fn in (lib.gat_oracle_capture_kinks, lib.gat_oracle_capture_kinks_f64):
        fn.argtypes = [C.c_double, C.c_int64, p, p, p, p, p]
        fn.restype = None
    return lib


def transpose_pattern(rowptr, col):
    """(rowptr_t, col_t, perm_t) with perm_t[k] = forward edge index of transposed edge k."""
    rowptr = np.asarray(rowptr, dtype=np.int64); col = np.asarray(col, dtype=np.int64)
    n = len(rowptr) - 1
    src = np.repeat(np.arange(n), np.diff(rowptr))
    order = np.argsort(col * n + src, kind="stable")
    rp_t = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=n))])
    return rp_t.astype(np.int32), src[order].astype(np.int32), order.astype(np.int32)


def level(X, rowptr, col, W, a, alpha, concat, G, want_dx=True, lib=None, tp=None, dtype=np.float32, kink_tau=0.0,
          kink_cap=4096):
    """dtype=np.float32: the fp32 port (gat_oracle_level); np.float64: the same source built with REAL = double
    (gat_oracle_level_f64) -- ground truth for the full-size tests.  kink_tau > 0: also returns "kinks" = dict(h, e,
    z, de) of the edges whose logit lies within kink_tau (|s_i| + |t_j|) of the LeakyReLU kink."""
    lib = lib or load()
    X = np.ascontiguousarray(X, dtype=dtype); W = np.ascontiguousarray(W, dtype=dtype)
    a = np.ascontiguousarray(a, dtype=dtype); G = np.ascontiguousarray(G, dtype=dtype)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    rp_t, col_t, perm_t = tp if tp is not None else transpose_pattern(rowptr, col)
    N, Fin = X.shape; H, _, F = W.shape
    out = np.empty_like(G); dW = np.empty_like(W); da = np.empty_like(a)
    dX = np.empty_like(X) if want_dx else None
    ptr = lambda v: None if v is None else v.ctypes.data
    f64 = np.dtype(dtype) == np.float64
    fn = lib.gat_oracle_level_f64 if f64 else lib.gat_oracle_level
    cap_fn = lib.gat_oracle_capture_kinks_f64 if f64 else lib.gat_oracle_capture_kinks
    if kink_tau > 0:
        kh = np.zeros(kink_cap, np.int32); ke = np.zeros(kink_cap, np.int64)
        kz = np.zeros(kink_cap, np.float64); kde = np.zeros(kink_cap, np.float64); kc = np.zeros(1, np.int64)
        cap_fn(float(kink_tau), kink_cap, ptr(kh), ptr(ke), ptr(kz), ptr(kde), ptr(kc))
    rc = fn(N, len(col), ptr(rowptr), ptr(col), ptr(rp_t), ptr(col_t), ptr(perm_t), Fin, H, F,
                              alpha, int(concat), ptr(X), ptr(W), ptr(a), ptr(G), ptr(out), ptr(dW), ptr(da), ptr(dX))
    if kink_tau > 0:
        cap_fn(0.0, 0, None, None, None, None, None)
    if rc != 0:
        raise MemoryError("gat_oracle_level: allocation failed")
    res = dict(out=out, dW=dW, da=da, dX=dX)
    if kink_tau > 0:
        n = int(kc[0])
        if n > kink_cap:
            raise RuntimeError(f"c_oracle: {n} near-kink edges exceed kink_cap={kink_cap}; lower kink_tau")
        order = np.lexsort((ke[:n], kh[:n]))          # thread timing decides the capture order: make it canonical
        res["kinks"] = dict(h=kh[:n][order], e=ke[:n][order], z=kz[:n][order], de=kde[:n][order])
    return res


def level_v2(X, rowptr, col, W, a, alpha, concat, G, want_dx=True, lib=None, tp=None, dtype=np.float32):
    """One SpGraphAttentionLayerV2 level (layers.py:258-313; gat_oracle.c gat_oracle_level_v2): W [H, 2 Fin, F], a [H, F];
    -> dict(out, dW [H, 2 Fin, F], da [H, F], dX).  dtype as in level()."""
    lib = lib or load()
    X = np.ascontiguousarray(X, dtype=dtype); W = np.ascontiguousarray(W, dtype=dtype)
    a = np.ascontiguousarray(a, dtype=dtype).reshape(W.shape[0], -1); G = np.ascontiguousarray(G, dtype=dtype)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    rp_t, col_t, perm_t = tp if tp is not None else transpose_pattern(rowptr, col)
    N, Fin = X.shape; H, twoFin, F = W.shape
    assert twoFin == 2 * Fin and a.shape == (H, F)
    out = np.empty_like(G); dW = np.empty_like(W); da = np.empty_like(a)
    dX = np.empty_like(X) if want_dx else None
    ptr = lambda v: None if v is None else v.ctypes.data
    fn = lib.gat_oracle_level_v2_f64 if np.dtype(dtype) == np.float64 else lib.gat_oracle_level_v2
    rc = fn(N, len(col), ptr(rowptr), ptr(col), ptr(rp_t), ptr(col_t), ptr(perm_t), Fin, H, F, alpha, int(concat),
            ptr(X), ptr(W), ptr(a), ptr(G), ptr(out), ptr(dW), ptr(da), ptr(dX))
    if rc != 0:
        raise MemoryError("gat_oracle_level_v2: allocation failed")
    return dict(out=out, dW=dW, da=da, dX=dX)
