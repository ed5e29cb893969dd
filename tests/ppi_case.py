"""BASELINE.json config 4 as a test case: a PPI-shaped batch (reference train_ppi.py:43-55, load_data_ppi.py:71-88).

Two graphs with the REAL node counts 591 and 1021 (tests/golden/ppi_graph_sizes.npz, from train_graph_id.npy);
the reference's edge lists are missing blobs, so each graph gets seeded random symmetric edges of mean degree 28
plus self loops; node features are real rows of valid_feats.npy
(tests/golden/ppi_feats_sample.npz).  Model = train_ppi.py's: 50 -> 256 x 4 -> 256 x 4 -> 121 x 6 (mean), skip
connections on, alpha 0.2, dropout 0 (train_ppi.py:49-55)."""
import os

import numpy as np
import torch

from oracle import gat_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NFEAT, NHEADS = [50, 256, 256, 121], [4, 4, 6]
SIZES = (591, 1021)


def graphs():
    sizes = np.load(os.path.join(GOLDEN, "ppi_graph_sizes.npz"), allow_pickle=False)["train"]
    assert all(s in sizes for s in SIZES)
    return [O.random_symmetric_csr(n, 28, 100 + k) for k, n in enumerate(SIZES)]


def features():
    return np.load(os.path.join(GOLDEN, "ppi_feats_sample.npz"), allow_pickle=False)["feats"]   # [1612, 50] float32


def batch_csr(parts):
    """CSR of the block-diagonal batch, built on the host from the per-graph CSRs (the expected answer)."""
    rps, cols, noff, eoff = [np.zeros(1, np.int64)], [], 0, 0
    for rp, c in parts:
        rps.append(np.asarray(rp[1:], np.int64) + eoff)
        cols.append(np.asarray(c, np.int64) + noff)
        noff += len(rp) - 1
        eoff += len(c)
    return np.concatenate(rps).astype(np.int32), np.concatenate(cols).astype(np.int32)


def oracle_levels(model, dtype):
    """The model's parameters as the oracle's level dicts (leaf tensors of `dtype` that require grad)."""
    levels = []
    for heads in model.gat_layers:
        mk = lambda name: torch.stack([getattr(h, name).detach().cpu().reshape(getattr(h, name).shape if name != "a" else (-1,))  # noqa: E731
                                       for h in heads]).to(dtype).requires_grad_(True)
        levels.append({"W": mk("W"), "a": mk("a"), "skip": mk("skip_projection") if model.skip_connection else None})
    return levels


def oracle_run(model, x, rowptr, col, G, dtype, flips=None):
    """Logits and every gradient from the oracle (autograd through oracle.model_forward) in `dtype`.
    flips: per level None or [H,E] bool, LeakyReLU branch overrides (oracle._leaky; tests/parity.py)."""
    levels = oracle_levels(model, dtype)
    xx = torch.as_tensor(x).to(dtype).requires_grad_(True)
    y = O.model_forward(xx, (rowptr, col), levels, model.alpha, "sparse", flips=flips)
    y.backward(torch.as_tensor(G).to(dtype))
    grads = {}
    for li, lv in enumerate(levels, start=1):
        for hd in range(lv["W"].shape[0]):
            grads[f"attention_layer_{li}_head_{hd + 1}.W"] = lv["W"].grad[hd]
            grads[f"attention_layer_{li}_head_{hd + 1}.a"] = lv["a"].grad[hd]
            if lv["skip"] is not None:
                grads[f"attention_layer_{li}_head_{hd + 1}.skip_projection"] = lv["skip"].grad[hd]
    return y.detach(), xx.grad, grads
