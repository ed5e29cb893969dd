#!/usr/bin/env python3
"""pygat_project ([Wh | s] of one level) in both GEMM modes over a few (Fin, H, F') -- which kernel should take which
shape.  Development diagnostic."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pygat_amd as pg  # noqa: E402
from pygat_amd._lib import lib, check  # noqa: E402

n = 1 << 20
for Fin, H, Fo in [(128, 8, 16), (128, 8, 8), (128, 4, 16), (128, 2, 16), (128, 1, 16), (64, 8, 8), (64, 8, 16), (128, 8, 32), (128, 8, 64)]:
    x = torch.randn(n, Fin, device="cuda")
    W = torch.randn(H, Fin, Fo, device="cuda") * 0.2
    a = torch.randn(H, 2 * Fo, device="cuda")
    Fp = pg.padded_width(Fo); R = H * Fp
    ldw = -(-(R + 2 * H) // 4) * 4
    Wcat = torch.empty(Fin, ldw, device="cuda"); a_pad = torch.empty(H, 2, Fp, device="cuda")
    check(lib.pygat_pack_params(H, Fin, Fo, W.data_ptr(), a.data_ptr(), None, Wcat.data_ptr(), ldw, a_pad.data_ptr(), None), "pack")
    Wh = torch.empty(n, R, device="cuda"); s = torch.empty(n, H, device="cuda")
    res = {}
    for mode in ("fp32-mfma", "split-bf16"):
        ts = []
        for _ in range(12):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.pygat_project(n, Fin, H, Fo, x.data_ptr(), Fin, Wcat.data_ptr(), ldw, a_pad.data_ptr(), Wh.data_ptr(), None,
                                    s.data_ptr(), 1, None, pg.ops.GEMM_MODES[mode], None), "project")
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res[mode] = float(np.median(ts[3:]))
    gb = 4.0 * n * (Fin + R + H) / 1e9
    print(f"Fin {Fin:4d} H {H} F' {Fo:3d} (R {R:4d}): fp32-mfma {res['fp32-mfma']*1e3:7.1f} us  split-bf16 {res['split-bf16']*1e3:7.1f} us   "
          f"({gb:.2f} GB: {gb/res['split-bf16']*1e3/1e3:.2f} TB/s)", flush=True)
