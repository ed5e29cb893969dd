"""Drop-in for the reference's `models.GAT` (models.py:7-35), fused per level.

Same constructor keywords; the per-head sub-modules are registered under the reference's names, so
`state_dict` keys are `attention_layer_{L}_head_{H}.{W,a,skip_projection}` (models.py:27).  forward(x,
adj) issues ONE fused call per level for all its heads (the reference walks the heads in a Python
list comprehension, models.py:32,34):

    hidden level : cat(heads, dim=1)            -> written in place by K2
    last level   : mean(stack(heads, 1), 1)     -> head-mean kernel

With `torch.distributed` initialised and `head_parallel=True` the heads of each level are sharded over
the ranks (pygat_amd.dist).  Every rank keeps ALL parameters (the module is a full replica, so `state_dict`
has the reference's keys everywhere), but only the heads a rank owns receive gradients and are stepped by its
optimiser: call `sync_head_parameters()` before `state_dict()` / checkpointing (train.py:201,233) or before
switching `head_parallel` off.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .features import as_sparse_features
from .graph import as_graph
from .layers import GraphAttentionLayer, SpGraphAttentionLayer
from .ops import gat_level


class GAT(nn.Module):
    def __init__(self, nfeat, nheads, nlayers, dropout, alpha, layer_type=GraphAttentionLayer,
                 skip_connection=False, head_parallel=False, level_fn=None):
        super().__init__()
        self.dropout, self.alpha = dropout, alpha
        self.skip_connection, self.head_parallel = skip_connection, head_parallel
        # test hook (CPU gloo tests of the sharding algebra swap the HIP level for the oracle); None = HIP path
        self.level_fn = level_fn
        widths = [1] + list(nheads)          # heads feeding level i (the input counts as one head)
        self.gat_layers = []                 # plain lists like the reference: registration is by name below
        for lvl in range(nlayers):
            last = lvl == nlayers - 1
            row = [layer_type(in_features=nfeat[lvl] * widths[lvl], out_features=nfeat[lvl + 1], dropout=dropout,
                              alpha=alpha, concat=not last, skip_connection=skip_connection)
                   for _ in range(widths[lvl + 1])]
            for hd, layer in enumerate(row, start=1):
                self.add_module(f"attention_layer_{lvl + 1}_head_{hd}", layer)
            self.gat_layers.append(row)
        self.pattern_mode = getattr(layer_type, "pattern_mode", "nonzero")
        from .gatv2 import SpGraphAttentionLayerV2
        self._kind = ("v1" if issubclass(layer_type, (GraphAttentionLayer, SpGraphAttentionLayer))
                      else "v2sp" if issubclass(layer_type, SpGraphAttentionLayerV2) else "other")

    def forward(self, x, adj):
        graph = adj if self.level_fn is not None else as_graph(adj, self.pattern_mode)
        p_drop = self.dropout if self.training else 0.0
        # Large graphs (ops.RENUMBER): the whole model runs in the graph's INTERNAL node order (descending degree: CSRGraph.
        # internal_view) -- x is permuted once (cached per feature tensor when it carries no gradient), every level reads the
        # previous level's output as it lies (hidden levels too), the self-loop-only nodes go through their tail streams, a
        # head-parallel model exchanges internal-order rows, and only the final [N, C] output is put back into the caller's order.
        to_internal = None
        if self._internal_order_pays(x, graph, p_drop):
            from .features import permuted_rows
            view = graph.internal_view()
            xp = permuted_rows(x, view.to_user) if not x.requires_grad else x.index_select(0, view.to_user.long())
            if xp is not None:
                x, graph, to_internal = xp, view, view.to_internal
        for lvl, heads in enumerate(self.gat_layers):
            concat = lvl < len(self.gat_layers) - 1
            if self._kind == "other":   # e.g. GraphAttentionLayerV2: one head per call, as the reference does
                ys = [att(x, graph) for att in heads]
                x = torch.cat(ys, dim=1) if concat else torch.mean(torch.stack(ys, dim=1), dim=1)
                continue
            Ws, As = [h.W for h in heads], [h.a for h in heads]
            Sk = [h.skip_projection for h in heads] if self.skip_connection else None
            fn = self.level_fn
            if self._kind == "v2sp" and fn is None:
                from .gatv2 import gatv2_level
                fn = lambda x_, g_, W_, a_, sk_, al_, cc_: gatv2_level(x_, g_, W_, a_, sk_, al_, cc_, p_drop)  # noqa: E731
            if self.head_parallel:
                from .dist import gat_level_head_parallel
                x = gat_level_head_parallel(x, graph, Ws, As, Sk, self.alpha, concat, p_drop, level_fn=fn)
            elif fn is not None:
                x = fn(x, graph, Ws, As, Sk, self.alpha, concat)
            else:
                # a first level on sparse input features (bag-of-words X: features.py) multiplies the non-zeros only
                xs = None
                if lvl == 0:
                    Fp = 1 << max(2, (heads[0].W.shape[1] - 1).bit_length())
                    xs = as_sparse_features(x, len(heads) * Fp * (2 if self.skip_connection else 1) + len(heads))
                if p_drop > 0.0:
                    from .dropout import gat_level_dropout
                    x = gat_level_dropout(x, graph, Ws, As, Sk, self.alpha, concat, p_drop, xs=xs)
                else:
                    x = gat_level(x, graph, Ws, As, Sk, self.alpha, concat, xs=xs)
        if to_internal is not None:
            x = x.index_select(0, to_internal.long())        # the final [N, C] output back in the caller's order (differentiable)
        return x

    def _internal_order_pays(self, x, graph, p_drop) -> bool:
        """Model-level internal node order: the v1 layers without dropout (the dropout path and GATv2 keep the caller's order), a
        2-D GPU input, not under stream capture, more than one level (a single level renumbers itself inside ops._level_forward,
        with the row maps in its kernels), and tables of ops.RENUMBER_MIN_BYTES(_TAIL) and more at the widest hidden level."""
        from . import ops
        if not (ops.RENUMBER and self.level_fn is None and self._kind == "v1" and p_drop == 0.0 and len(self.gat_layers) > 1
                and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 2 and hasattr(graph, "degree_ordered")
                and not torch.cuda.is_current_stream_capturing()):
            return False
        widest = max(len(h) * ops.padded_width(h[0].W.shape[1]) for h in self.gat_layers[:-1])
        nbytes = graph.n * widest * 4
        if nbytes >= ops.RENUMBER_MIN_BYTES:
            return True
        if nbytes < ops.RENUMBER_MIN_BYTES_TAIL or not (ops.TAIL and graph.symmetric):
            return False
        from .graph import slot_edges_for
        t = graph.degree_ordered()[0].fwd.self_loop_tail(slot_edges_for(widest, graph.slot_edges))
        return t is not None and graph.n - t[0] >= ops.TAIL_MIN_SHARE * graph.n

    @torch.no_grad()
    def sync_head_parameters(self):
        """Head-parallel training steps only the heads a rank owns (`dist.partition_heads`); the other heads'
        parameters on that rank go stale.  This broadcasts every head's W / a / skip_projection from its owner
        so that all replicas -- and hence `state_dict()` on any rank, which the reference saves every epoch and
        reloads at the end (train.py:201,233) -- hold the trained values.  One flat buffer per (level, owner)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        from .dist import partition_heads
        world = dist.get_world_size()
        for heads in self.gat_layers:
            for owner, (s, e) in enumerate(partition_heads(len(heads), world)):
                ps = [p for h in heads[s:e] for p in h.parameters()]
                if not ps:
                    continue
                flat = torch.cat([p.detach().reshape(-1) for p in ps])
                dist.broadcast(flat, src=owner)
                o = 0
                for p in ps:
                    p.copy_(flat[o:o + p.numel()].view_as(p))
                    o += p.numel()
