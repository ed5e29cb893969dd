"""Drop-in for the reference's `models.GAT` (models.py:7-35), fused per level.

Same constructor, same registered sub-modules and therefore the same
`state_dict` keys `attention_layer_{L}_head_{H}.{W,a,skip_projection}`
(models.py:27).  forward(x, adj) runs ONE fused call per level for all heads
(the reference runs the heads one after another in a Python list
comprehension, models.py:32,34):

    hidden level : cat(heads, dim=1)            -> written in place by K2
    last level   : mean(stack(heads, 1), 1)     -> head-mean kernel

With `torch.distributed` initialised and `head_parallel=True` the heads of each
level are sharded over the ranks (pygat_amd.dist).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .graph import as_graph
from .layers import GraphAttentionLayer, SpGraphAttentionLayer
from .ops import gat_level


class GAT(nn.Module):
    def __init__(self, nfeat, nheads, nlayers, dropout, alpha, layer_type=GraphAttentionLayer,
                 skip_connection=False, head_parallel=False):
        super().__init__()
        self.dropout = dropout
        self.alpha = alpha
        self.skip_connection = skip_connection
        self.head_parallel = head_parallel
        nheads = [1] + list(nheads)
        self.gat_layers = []
        for i in range(nlayers):
            self.gat_layers.append([])
            for j in range(nheads[i + 1]):
                layer = layer_type(
                    in_features=nfeat[i] * nheads[i],
                    out_features=nfeat[i + 1],
                    dropout=dropout,
                    alpha=alpha,
                    concat=True if i < nlayers - 1 else False,
                    skip_connection=skip_connection,
                )
                self.gat_layers[i].append(layer)
                self.add_module('attention_layer_{}_head_{}'.format(i + 1, j + 1), layer)
        self.pattern_mode = getattr(layer_type, "pattern_mode", "nonzero")
        from .gatv2 import SpGraphAttentionLayerV2
        self._kind = ("v1" if issubclass(layer_type, (GraphAttentionLayer, SpGraphAttentionLayer))
                      else "v2sp" if issubclass(layer_type, SpGraphAttentionLayerV2) else "other")

    def forward(self, x, adj):
        graph = as_graph(adj, self.pattern_mode)
        nl = len(self.gat_layers)
        for i, heads in enumerate(self.gat_layers):
            concat = i < nl - 1
            if self._kind == "other":   # e.g. GraphAttentionLayerV2: one head per call, as the reference does
                ys = [att(x, graph) for att in heads]
                x = torch.cat(ys, dim=1) if concat else torch.mean(torch.stack(ys, dim=1), dim=1)
                continue
            if self._kind == "v2sp":
                from .gatv2 import gatv2_level
                x = gatv2_level(x, graph, [h.W for h in heads], [h.a for h in heads],
                                [h.skip_projection for h in heads] if self.skip_connection else None, self.alpha, concat,
                                self.dropout if self.training else 0.0)
                continue
            Ws = [h.W for h in heads]
            As = [h.a for h in heads]
            Sk = [h.skip_projection for h in heads] if self.skip_connection else None
            if self.head_parallel:
                from .dist import gat_level_head_parallel
                x = gat_level_head_parallel(x, graph, Ws, As, Sk, self.alpha, concat,
                                            self.dropout if self.training else 0.0)
            elif self.training and self.dropout > 0.0:
                from .dropout import gat_level_dropout
                x = gat_level_dropout(x, graph, Ws, As, Sk, self.alpha, concat, self.dropout, head_mean=not concat)
            else:
                x = gat_level(x, graph, Ws, As, Sk, self.alpha, concat)
        return x
