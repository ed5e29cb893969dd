"""pygat_amd -- MI355X-native GAT attention layer (hand-written HIP for gfx950).

Drop-in for the hot path of ArielleRosinski/pyGAT: `GraphAttentionLayer`,
`SpGraphAttentionLayer` (reference layers.py) and `GAT` (reference models.py).
Importing this package loads libpygat_amd.so and fails loudly when it is absent:
there is no CPU / eager fallback.
"""
from ._lib import lib, LIB_PATH, padded_width          # noqa: F401  (raises if the .so is missing)
from .graph import CSRGraph, as_graph                   # noqa: F401
from .ops import gat_level, GATLevelFn, gemm, set_gemm_mode, get_gemm_mode, gemm_mode            # noqa: F401
from .layers import GraphAttentionLayer, SpGraphAttentionLayer  # noqa: F401
from .models import GAT                                 # noqa: F401
from .graphed import GraphedLevel, FusedEpoch           # noqa: F401
from .losses import EluLogSoftmaxNLL, BCEWithLogits                    # noqa: F401
from .optim import Adam                                 # noqa: F401
from .gatv2 import GraphAttentionLayerV2, SpGraphAttentionLayerV2, gatv2_level, GATv2LevelFn  # noqa: F401

__all__ = ["Adam", "BCEWithLogits", "CSRGraph", "as_graph", "gat_level", "GATLevelFn", "gemm", "set_gemm_mode", "get_gemm_mode", "gemm_mode", "GraphAttentionLayer",
           "SpGraphAttentionLayer", "GAT", "padded_width", "LIB_PATH", "GraphAttentionLayerV2",
           "SpGraphAttentionLayerV2", "gatv2_level", "GATv2LevelFn", "GraphedLevel", "FusedEpoch", "EluLogSoftmaxNLL"]
