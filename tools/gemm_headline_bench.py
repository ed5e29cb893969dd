#!/usr/bin/env python3
"""The two streamed GEMMs of the headline step (config 5: projection [Wh | s] = X Wcat, weight gradient dW = X^T dWh) timed
through the C ABI, HIP events, median of --iters.  With PYGAT_AMD_LIB=<variant .so> (tools/build_variant.sh) it times a
diagnostic build.   python3 tools/gemm_headline_bench.py [--heads 8] [--fout 16] [--fin 128] [--iters 30]"""
import argparse
import os
import time
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pygat_amd as pg  # noqa: E402
from pygat_amd._lib import lib, check  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--fin", type=int, default=128)
ap.add_argument("--heads", type=int, default=8)
ap.add_argument("--fout", type=int, default=16)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--mode", default="split-bf16")
ap.add_argument("--stamps", action="store_true", help="read the stamps of a PYGAT_DIAG_K1 & 16 build (pygat_diag_k1_stamps)")
ap.add_argument("--tag", default=os.path.basename(os.environ.get("PYGAT_AMD_LIB", "default")))
ap.add_argument("--split-k", type=int, default=0, help="slabs of the weight gradient (0: the package's choice)")
ap.add_argument("--gap-ms", type=float, default=0.0, help="idle time between timed launches (lets the clocks recover)")
a_ = ap.parse_args()
n, Fin, H, Fo = a_.n, a_.fin, a_.heads, a_.fout
torch.manual_seed(0)
x = torch.randn(n, Fin, device="cuda")
W = torch.randn(H, Fin, Fo, device="cuda") * 0.2
a = torch.randn(H, 2 * Fo, device="cuda")
Fp = pg.padded_width(Fo); R = H * Fp
ldw = -(-(R + 2 * H) // 4) * 4
Wcat = torch.empty(Fin, ldw, device="cuda"); a_pad = torch.empty(H, 2, Fp, device="cuda")
check(lib.pygat_pack_params(H, Fin, Fo, W.data_ptr(), a.data_ptr(), None, Wcat.data_ptr(), ldw, a_pad.data_ptr(), None), "pack")
Wh = torch.empty(n, R, device="cuda"); s = torch.zeros(n, H, device="cuda")
dWh = torch.randn(n, R, device="cuda")
mode = pg.ops.GEMM_MODES[a_.mode]
split_k = a_.split_k or pg.ops._split_k(Fin, R, n, streamed_k=True, mode=a_.mode)
wsw = torch.empty(lib.pygat_wgrad_workspace_bytes(Fin, H, Fo, split_k) // 4, device="cuda")
dW = torch.empty(H, Fin, Fo, device="cuda")


def timed(fn):
    ts = []
    for _ in range(a_.iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        if a_.gap_ms:
            time.sleep(a_.gap_ms * 1e-3)
    return float(np.median(ts[5:]))


def project():
    check(lib.pygat_project(n, Fin, H, Fo, x.data_ptr(), Fin, Wcat.data_ptr(), ldw, a_pad.data_ptr(), Wh.data_ptr(), None,
                            s.data_ptr(), 1, None, mode, None), "project")


def wgrad():
    check(lib.pygat_wgrad(n, Fin, H, Fo, x.data_ptr(), Fin, dWh.data_ptr(), None, a_pad.data_ptr(), dW.data_ptr(), split_k,
                          wsw.data_ptr(), 0, 0, mode, None), "wgrad")


tp, tw = timed(project), timed(wgrad)
if a_.stamps:
    import ctypes
    lib.pygat_diag_k1_stamps.restype = ctypes.c_int
    lib.pygat_diag_k1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros(2048 * 64, dtype=np.uint64)
    assert lib.pygat_diag_k1_stamps(buf.ctypes.data, buf.size) == buf.size
    raw = buf.reshape(-1, 8).astype(np.int64)
    raw = raw[raw[:, 3] > 0]
    tot, mf, ep, nt = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]
    t0 = raw[:, 4].min()
    entry, first, end = (raw[:, 4] - t0) * 0.01, (raw[:, 5] - t0) * 0.01, (raw[:, 6] - t0) * 0.01    # us since the first wave's entry
    pct = lambda v: " / ".join(f"{np.percentile(v, q):.1f}" for q in (0, 10, 50, 90, 100))  # noqa: E731
    print(f"stamps: {len(raw)} waves, tiles per wave {nt.mean():.1f}; per tile (shader clocks): total {np.median(tot / nt):.0f}  MFMA phase "
          f"{np.median(mf / nt):.0f}  epilogue {np.median(ep / nt):.0f}  (288 MFMAs alone = 9216; two waves share a SIMD's pipe)")
    print(f"        kernel {tp*1e3:.1f} us by HIP events; shader clock {np.median(tot / np.maximum(raw[:, 6] - raw[:, 5], 1)) * 0.1:.2f} GHz (s_memtime against the "
          f"100 MHz s_memrealtime)")
    print(f"        us since the first wave's entry, min / p10 / median / p90 / max over the waves: entry {pct(entry)}; first tile {pct(first)}; "
          f"end {pct(end)}; span {pct(end - first)}")
    wg_end = end.reshape(-1, 8).max(axis=1) if len(end) % 8 == 0 else end
    for x in range(8):
        print(f"        work-groups = {x} mod 8: end {pct(wg_end[x::8])}")
gbp, gbw = 4.0 * n * (Fin + R + H) / 1e9, 4.0 * n * (Fin + R) / 1e9
# a checksum so that a diagnostic build that computes something else shows it
print(f"{a_.tag:28s} project {tp*1e3:7.1f} us ({gbp/tp:5.2f} TB/s)  wgrad {tw*1e3:7.1f} us ({gbw/tw:5.2f} TB/s, split_k {split_k})  "
      f"|Wh| {float(Wh.abs().sum()):.6e} |dW| {float(dW.abs().sum()):.6e}", flush=True)
