"""Every development switch of the Python layer in ONE object, read from the environment ONCE, when the package is imported.

The switches exist for A/B measurements (none changes results beyond summation order); the shipped
defaults are the measured winners.  Modules copy the values they need into module-level names at import (ops.BACKWARD_FLAVOUR,
dist.PIPELINE_CHUNKS, features.ENABLED, ...: tests and tools assign those), so nothing in the package reads `os.environ`
while it runs, and two ranks of a job cannot come to different answers about, say, the number of row chunks of a pipelined
level half way through it.  The C library reads two process defaults of its own when it is loaded (PYGAT_GEMM_F32,
PYGAT_NARROW: include/pygat_amd.h).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, fields
from typing import Optional


def _flag(name: str, default: bool) -> bool:
    v = os.environ.get(name)
    return default if v is None else v not in ("0", "", "false", "False")


def _opt_flag(name: str) -> Optional[bool]:
    v = os.environ.get(name)
    return None if v is None else v == "1"


def _opt_int(name: str) -> Optional[int]:
    v = os.environ.get(name)
    return int(v) if v else None


@dataclass(frozen=True)
class Config:
    lib_path: Optional[str]               # PYGAT_AMD_LIB            another build of libpygat_amd.so (tools/build_variant.sh)
    dist_chunks: int                      # PYGAT_DIST_CHUNKS        row chunks of a pipelined head-parallel level (dist.PIPELINE_CHUNKS)
    dropout_wide: bool                    # PYGAT_DROPOUT_WIDE=1     round 1's wide-operand dropout projection everywhere
    sparse_max_density: float             # PYGAT_SPARSE_MAX_DENSITY input features denser than this take the dense GEMMs
    sparse_x: bool                        # PYGAT_SPARSE_X=0         never take the sparse-feature path
    slot_edges: Optional[int]             # PYGAT_SLOT_EDGES         force the slot length of the nnz-split kernels
    slot_meta: bool                       # PYGAT_NO_SLOT_META=1     kernels walk slot_begin -> edge_rc -> rowptr instead of slot records
    two_gather_backward: Optional[bool]   # PYGAT_TWO_GATHER_BACKWARD=1/0  (the older switch between the two older backward flavours)
    backward: Optional[str]               # PYGAT_BACKWARD           rowlocal | rowsum | two-gather
    da_in_k4: bool                        # PYGAT_DA_IN_K4=0         the a-gradient as a pass of its own
    k1_split_min_k: int                   # PYGAT_K1_SPLIT_MIN_K     slab floor of a small graph's split-K projection
    overlap_backward: bool                # PYGAT_OVERLAP_BACKWARD=1 a-gradient on a side stream beside the weight gradient
    pad_k: bool                           # PYGAT_PAD_K=0            odd input widths run as they are
    renumber: bool                        # PYGAT_RENUMBER=0         large first levels in the caller's node order (no internal degree order)
    tail: bool                            # PYGAT_TAIL=0             self-loop-only nodes through the fused kernels like every other row

    @staticmethod
    def from_env() -> "Config":
        return Config(
            lib_path=os.environ.get("PYGAT_AMD_LIB") or None,
            dist_chunks=int(os.environ.get("PYGAT_DIST_CHUNKS", 4)),
            dropout_wide=_flag("PYGAT_DROPOUT_WIDE", False),
            sparse_max_density=float(os.environ.get("PYGAT_SPARSE_MAX_DENSITY", 0.15)),
            sparse_x=_flag("PYGAT_SPARSE_X", True),
            slot_edges=_opt_int("PYGAT_SLOT_EDGES"),
            slot_meta=not _flag("PYGAT_NO_SLOT_META", False),
            two_gather_backward=_opt_flag("PYGAT_TWO_GATHER_BACKWARD"),
            backward=os.environ.get("PYGAT_BACKWARD") or None,
            da_in_k4=_flag("PYGAT_DA_IN_K4", True),
            k1_split_min_k=int(os.environ.get("PYGAT_K1_SPLIT_MIN_K", 128)),
            overlap_backward=_flag("PYGAT_OVERLAP_BACKWARD", False),
            pad_k=_flag("PYGAT_PAD_K", True),
            renumber=_flag("PYGAT_RENUMBER", True),
            tail=_flag("PYGAT_TAIL", True),
        )

    def describe(self) -> dict:
        return {f.name: getattr(self, f.name) for f in fields(self)}


config = Config.from_env()
