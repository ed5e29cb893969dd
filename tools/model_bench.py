#!/usr/bin/env python3
"""A multi-level pygat_amd.GAT on the headline graph, forward + backward (hidden levels carry a gradient into their input): what the
model-level internal node order (pygat_amd.GAT on large graphs: x permuted once, every level on the degree-ordered pattern, tail
streams at every concat level, logits put back) is worth against the caller's order.
    python3 tools/model_bench.py [--levels 128 16 16 8 --heads 8 8 4]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygat_amd as pg  # noqa: E402
from pygat_amd.rmat import rmat_csr_numpy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--levels", type=int, nargs="*", default=[128, 16, 16, 8])
ap.add_argument("--heads", type=int, nargs="*", default=[8, 8, 4])
ap.add_argument("--skip", action="store_true")
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
rp, col = rmat_csr_numpy(20, 5_000_000, seed=1)
graph = pg.CSRGraph(torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev))
N = graph.n
torch.manual_seed(0)
model = pg.GAT(a.levels, a.heads, len(a.heads), 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=a.skip).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, a.levels[0], generator=g, device=dev)
G = torch.randn(N, a.levels[-1], generator=g, device=dev)
res = {}
for renumber in (True, False, True, False):
    pg.ops.RENUMBER = renumber

    def step():
        model.zero_grad(set_to_none=True)
        y = model(x, graph)
        y.backward(G)
        return y
    for _ in range(3):
        y = step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record(); torch.cuda.synchronize()
    res.setdefault(renumber, []).append(round(e0.elapsed_time(e1) / a.steps, 3))
    res.setdefault(("y", renumber), y.detach())
err = float((res[("y", True)] - res[("y", False)]).abs().max())
print(json.dumps({"model": f"GAT {a.levels} x heads {a.heads}{' + skip' if a.skip else ''}", "nodes": N, "edges": graph.nnz,
                  "ms_per_step_internal_order": res[True], "ms_per_step_caller_order": res[False], "max_abs_diff_logits": err}))
