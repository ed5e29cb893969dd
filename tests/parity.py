"""Tolerance rule of the parity tests, in one place.

Ground truth is the oracle in fp64.  The HIP path computes in fp32, as the reference does, so
  forward   : |got - ref64| <= 1e-5 absolute (the north star's bar),
  gradients : |got - ref64| <= max(1e-5, 4 x |ref32 - ref64|), ref32 = the SAME oracle run in fp32 --
              i.e. no worse than four times what the reference's own precision costs on that input
              (gradients are sums over up to 10^6 nodes and have no fixed magnitude).
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


def close_fwd(got, ref64, what, atol=ATOL):
    got, ref64 = _np64(got), _np64(ref64)
    assert got.shape == ref64.shape, f"{what}: shape {got.shape} vs {ref64.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    e = float(np.abs(got - ref64).max())
    assert e <= atol, f"{what}: max abs err {e:.3e} > {atol:.1e} (max |ref| {np.abs(ref64).max():.3g})"
    return e


def close_grad(got, ref64, ref32, what, factor=4.0, floor=ATOL):
    got, ref64, ref32 = _np64(got), _np64(ref64), _np64(ref32)
    assert got.shape == ref64.shape, f"{what}: shape {got.shape} vs {ref64.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    own = float(np.abs(ref32 - ref64).max())
    tol = max(floor, factor * own)
    e = float(np.abs(got - ref64).max())
    assert e <= tol, (f"{what}: max abs err {e:.3e} > {tol:.3e} = max({floor:.0e}, {factor:g} x fp32-oracle err "
                      f"{own:.3e}); max |ref| {np.abs(ref64).max():.3g}")
    return e, own
