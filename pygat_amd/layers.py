"""Drop-in layer classes for the reference's `layer_type` plug-in point.

`models.GAT(..., layer_type=Cls)` builds `Cls(in_features=, out_features=,
dropout=, alpha=, concat=, skip_connection=)` by keyword (reference
models.py:18-25) and calls `att(x, adj)` (models.py:32,34).  These classes keep
the constructor, the parameter names / shapes / initialisers (so seeds and
`state_dict`s line up) and `forward(h, adj) -> [N, out_features]`, and run the
fused HIP path (one head per call here; `pygat_amd.models.GAT` batches all heads
of a level into one call).

    GraphAttentionLayer    layers.py:8-67    a is [2F',1], xavier_uniform, mask adj > 0
    SpGraphAttentionLayer  layers.py:98-176  a is [1,2F'], xavier_normal, pattern adj != 0
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .graph import as_graph
from .ops import gat_level


class _FusedGATLayer(nn.Module):
    pattern_mode = "nonzero"

    def __init__(self, in_features, out_features, dropout, alpha, concat=True, skip_connection=False):
        super().__init__()
        self.dropout = dropout
        self.in_features = in_features
        self.out_features = out_features
        self.alpha = alpha
        self.concat = concat
        self.skip_connection = skip_connection

    def forward(self, h, adj):
        if self.training and self.dropout > 0.0:
            from .dropout import gat_level_dropout
            return gat_level_dropout(h, as_graph(adj, self.pattern_mode), [self.W], [self.a],
                                     [self.skip_projection] if self.skip_connection else None,
                                     self.alpha, self.concat, self.dropout, head_mean=False)
        out = gat_level(h, as_graph(adj, self.pattern_mode), [self.W], [self.a],
                        [self.skip_projection] if self.skip_connection else None, self.alpha, self.concat)
        return out

    def __repr__(self):  # layers.py:66-67,175-176
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class GraphAttentionLayer(_FusedGATLayer):
    """Same interface and initialisation as reference layers.py:12-30."""
    pattern_mode = "positive"   # torch.where(adj > 0, ...), layers.py:41

    def __init__(self, in_features, out_features, dropout, alpha, concat=True, skip_connection=False):
        super().__init__(in_features, out_features, dropout, alpha, concat, skip_connection)
        self.W = nn.Parameter(torch.empty(size=(in_features, out_features)))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        self.a = nn.Parameter(torch.empty(size=(2 * out_features, 1)))
        nn.init.xavier_uniform_(self.a.data, gain=1.414)
        if self.skip_connection:
            self.skip_projection = nn.Parameter(torch.empty(size=(in_features, out_features)))
            nn.init.xavier_uniform_(self.skip_projection.data, gain=1.414)


class SpGraphAttentionLayer(_FusedGATLayer):
    """Same interface and initialisation as reference layers.py:103-123."""
    pattern_mode = "nonzero"    # adj.nonzero(), layers.py:129

    def __init__(self, in_features, out_features, dropout, alpha, concat=True, skip_connection=False):
        super().__init__(in_features, out_features, dropout, alpha, concat, skip_connection)
        self.W = nn.Parameter(torch.zeros(size=(in_features, out_features)))
        nn.init.xavier_normal_(self.W.data, gain=1.414)
        self.a = nn.Parameter(torch.zeros(size=(1, 2 * out_features)))
        nn.init.xavier_normal_(self.a.data, gain=1.414)
        if self.skip_connection:
            self.skip_projection = nn.Parameter(torch.empty(size=(in_features, out_features)))
            nn.init.xavier_uniform_(self.skip_projection.data, gain=1.414)
