"""Drop-in layer classes for the reference's `layer_type` plug-in point.

`models.GAT(..., layer_type=Cls)` builds `Cls(in_features=, out_features=, dropout=, alpha=,
concat=, skip_connection=)` by keyword (reference models.py:18-25) and calls `att(x, adj)`
(models.py:32,34).  The classes here accept the same keywords, expose the same parameter names and
shapes and draw them from the same initialisers in the same order (so a seeded construction yields the
reference's weights and `state_dict`s interchange), and run the fused HIP path -- one head per call
here; `pygat_amd.models.GAT` batches all heads of a level into one call.

    GraphAttentionLayer    reference layers.py:8-67    a: [2F',1]  xavier_uniform   pattern adj > 0
    SpGraphAttentionLayer  reference layers.py:98-176  a: [1,2F']  xavier_normal    pattern adj != 0
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .graph import as_graph
from .ops import gat_level

_GAIN = 1.414   # layers.py:22,24,28


class _FusedGATLayer(nn.Module):
    pattern_mode = "nonzero"
    _a_shape = staticmethod(lambda f: (1, 2 * f))
    _init = staticmethod(nn.init.xavier_normal_)

    def __init__(self, in_features, out_features, dropout, alpha, concat=True, skip_connection=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.dropout, self.alpha = dropout, alpha
        self.concat, self.skip_connection = concat, skip_connection
        # creation order W, a, skip_projection = the reference's RNG consumption order
        shapes = [("W", (in_features, out_features), self._init), ("a", self._a_shape(out_features), self._init)]
        if skip_connection:   # the skip projection is xavier_uniform in BOTH reference layers (layers.py:28,119)
            shapes.append(("skip_projection", (in_features, out_features), nn.init.xavier_uniform_))
        for name, shape, init in shapes:
            p = nn.Parameter(torch.empty(shape))
            init(p.data, gain=_GAIN)
            setattr(self, name, p)

    def forward(self, h, adj):
        graph = as_graph(adj, self.pattern_mode)
        skips = [self.skip_projection] if self.skip_connection else None
        if self.training and self.dropout > 0.0:
            from .dropout import gat_level_dropout
            return gat_level_dropout(h, graph, [self.W], [self.a], skips, self.alpha, self.concat, self.dropout)
        return gat_level(h, graph, [self.W], [self.a], skips, self.alpha, self.concat)

    def __repr__(self):
        return f"{type(self).__name__} ({self.in_features} -> {self.out_features})"


class GraphAttentionLayer(_FusedGATLayer):
    """Dense-adjacency flavour: mask `adj > 0` (layers.py:41), a stored as a column (layers.py:23)."""
    pattern_mode = "positive"
    _a_shape = staticmethod(lambda f: (2 * f, 1))
    _init = staticmethod(nn.init.xavier_uniform_)


class SpGraphAttentionLayer(_FusedGATLayer):
    """Edge-list flavour: pattern `adj.nonzero()` (layers.py:129), a stored as a row (layers.py:114)."""
    pattern_mode = "nonzero"
    _a_shape = staticmethod(lambda f: (1, 2 * f))
    _init = staticmethod(nn.init.xavier_normal_)
