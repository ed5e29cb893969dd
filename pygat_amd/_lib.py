"""ctypes binding of libpygat_amd.so (the C ABI in include/pygat_amd.h).

There is NO fallback: if the HIP library is missing or does not export the
expected symbols, importing this module raises.  PyTorch is used by the callers
only to own device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: it brings its own HIP runtime (torch/lib/libamdhip64.so); libpygat_amd.so must bind to that one.  Loaded
# before torch, the library pulls in /opt/rocm/lib/libamdhip64.so.7 instead, the process then holds two HIP runtimes and
# the library's launches fail with "no ROCm-capable device is detected" while torch sees the GPU (met on the GPU box when
# __graft_entry__.build() imported the package before anything had imported torch).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpygat_amd.so")
# development aid (tools/build_variant.sh, same-lease A/B runs): another build of the same library.  Still a HIP library or
# nothing -- the symbol check below applies to it as well.
from .config import config as _config  # noqa: E402
if _config.lib_path:
    LIB_PATH = os.path.abspath(_config.lib_path)

ABI_VERSION = 14
F_ELU = 1
F_SKIP = 2
F_MAIN_ONLY = 4
F_FIXUP_ONLY = 8

# every entry point declared in include/pygat_amd.h
SYMBOLS = [
    "pygat_abi_version", "pygat_last_error", "pygat_padded_width", "pygat_device_count",
    "pygat_device_name", "pygat_kernel_footprint", "pygat_default_gemm_mode", "pygat_dense_row_counts", "pygat_scan_workspace_bytes",
    "pygat_exclusive_scan_i32", "pygat_dense_fill_cols", "pygat_csr_symmetric_perm",
    "pygat_gemm_workspace_bytes", "pygat_gemm_f32", "pygat_gemm_f32_blocked", "pygat_project_blocked", "pygat_wgrad_blocked", "pygat_pack_params", "pygat_pack_params_heads", "pygat_stack_heads", "pygat_stack_heads_padded", "pygat_project", "pygat_attn_scores",
    "pygat_unpack_wgrad",
    "pygat_edge_pairs", "pygat_slot_bounds", "pygat_slot_meta", "pygat_partials_bytes", "pygat_head_group", "pygat_gat_forward", "pygat_gat_forward_phases_ok", "pygat_gat_forward_tail", "pygat_gat_backward_col_tail", "pygat_gat_backward_tail", "pygat_head_mean",
    "pygat_gat_backward_prepare", "pygat_gat_backward_row", "pygat_gat_backward_col", "pygat_gat_backward_rowsum",
    "pygat_gat_backward_col_da_bytes", "pygat_a_grad_fold",
    "pygat_agrad_workspace_bytes", "pygat_a_grad", "pygat_wgrad_workspace_bytes", "pygat_wgrad",
    "pygat_gatv2_forward", "pygat_gatv2_backward_prepare", "pygat_gatv2_workspace_bytes", "pygat_gatv2_backward",
    "pygat_dropout_mask", "pygat_dropout_mask2", "pygat_dropout_expand", "pygat_dropout_head_sum", "pygat_pack_blockdiag",
    "pygat_unpack_blockdiag",
    "pygat_headmask_supported", "pygat_dropout_bits", "pygat_project_dropout_workspace_bytes", "pygat_project_dropout",
    "pygat_wgrad_dropout_workspace_bytes",
    "pygat_wgrad_dropout", "pygat_dropout_head_sum_bits",
    "pygat_nll_workspace_bytes", "pygat_elu_logsoftmax_nll", "pygat_elu_logsoftmax_nll_backward",
    "pygat_project_sparse", "pygat_wgrad_sparse", "pygat_wgrad_sparse_workspace_bytes", "pygat_dropout_narrow", "pygat_dx_dropout", "pygat_adam_step", "pygat_bce_workspace_bytes", "pygat_bce_with_logits", "pygat_bce_with_logits_backward",
]


MAX_ADAM_TENSORS = 48    # PYGAT_ADAM_MAX_TENSORS
MAX_SEGMENTS = 4    # PYGAT_MAX_SEGMENTS


class OutSegments(C.Structure):
    _fields_ = [("nseg", C.c_int), ("col_start", C.c_int32 * (MAX_SEGMENTS + 1)), ("ptr", C.c_void_p * MAX_SEGMENTS),
                ("ld", C.c_int64 * MAX_SEGMENTS)]


class ColBlocks(C.Structure):
    """pygat_col_blocks: a [rows x cols] matrix stored as cols / w blocks of [rows x w], `stride` floats apart."""
    _fields_ = [("w", C.c_int), ("stride", C.c_int64)]


class Graph(C.Structure):
    _fields_ = [("n", C.c_int), ("nnz", C.c_int64), ("rowptr", C.c_void_p), ("edge_rc", C.c_void_p),
                ("slot_edges", C.c_int), ("slot_begin", C.c_void_p), ("cut_rows", C.c_void_p),
                ("n_cut", C.c_int), ("n_cut_wide", C.c_int), ("slot_first", C.c_int64), ("slot_count", C.c_int64),
                ("slot_meta", C.c_void_p), ("slot_order", C.c_void_p), ("user_row", C.c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"pygat_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C pygat_amd/csrc`. There is no CPU/PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"pygat_amd: {LIB_PATH} lacks symbols {missing}")
    p, i, i64, f, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
    lib.pygat_abi_version.restype = i
    lib.pygat_last_error.restype = C.c_char_p
    lib.pygat_padded_width.argtypes = [i]
    lib.pygat_device_name.argtypes = [C.c_char_p, i]
    lib.pygat_kernel_footprint.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pygat_default_gemm_mode.argtypes = []
    lib.pygat_dense_row_counts.argtypes = [p, i, i64, i, p, p]
    lib.pygat_scan_workspace_bytes.argtypes = [i64]
    lib.pygat_scan_workspace_bytes.restype = sz
    lib.pygat_exclusive_scan_i32.argtypes = [p, i64, p, p, p]
    lib.pygat_dense_fill_cols.argtypes = [p, i, i64, i, p, p, p]
    lib.pygat_csr_symmetric_perm.argtypes = [i, p, p, p, p, p]
    lib.pygat_gemm_workspace_bytes.argtypes = [i, i, i]
    lib.pygat_gemm_workspace_bytes.restype = sz
    lib.pygat_gemm_f32.argtypes = [i, i, i, i, i64, p, i64, p, i64, C.POINTER(OutSegments), i, i, p, i, p]
    CB = C.POINTER(ColBlocks)
    lib.pygat_gemm_f32_blocked.argtypes = [i, i, i, i, i64, p, i64, CB, p, i64, C.POINTER(OutSegments), CB, i, i, p, i, p]
    lib.pygat_project_blocked.argtypes = [i, i, i, i, p, i64, CB, p, i64, p, p, p, p, i, p, i, p]
    lib.pygat_wgrad_blocked.argtypes = [i, i, i, i, p, i64, CB, p, p, p, p, i, p, i, i, i, p]
    lib.pygat_pack_params.argtypes = [i, i, i, p, p, p, p, i64, p, p]
    lib.pygat_pack_params_heads.argtypes = [i, i, i, p, p, p, p, i64, p, p]
    lib.pygat_stack_heads.argtypes = [i, i64, i, i64, p, p, p, p, p, p, p]
    lib.pygat_stack_heads_padded.argtypes = [i, i64, i64, i, i64, i64, p, p, p, p, p, p, p]
    lib.pygat_unpack_wgrad.argtypes = [i, i, i, p, i64, i, p, p]
    lib.pygat_attn_scores.argtypes = [i, i, i, p, p, p, p, p, p]
    lib.pygat_project.argtypes = [i, i, i, i, p, i64, p, i64, p, p, p, p, i, p, i, p]
    lib.pygat_edge_pairs.argtypes = [i, p, p, p, p]
    lib.pygat_slot_bounds.argtypes = [i, i64, p, p, i, p, p]
    lib.pygat_slot_meta.argtypes = [i, i64, p, p, i, p, p, p]
    lib.pygat_partials_bytes.argtypes = [i64, i, i, i]
    lib.pygat_partials_bytes.restype = sz
    lib.pygat_head_group.argtypes = [i, i, i]
    lib.pygat_head_group.restype = i
    lib.pygat_gat_forward.argtypes = [C.POINTER(Graph), i, i, f, i, p, p, p, p, p, p, p, p, p, p, p, p, p]
    lib.pygat_gat_forward_phases_ok.argtypes = [i, i, i]
    lib.pygat_gat_forward_tail.argtypes = [i, i, i, i, i, p, i64, p, p, p, p, p, p, p]
    lib.pygat_gat_backward_col_tail.argtypes = [i, i, i, i, p, p, p, p]
    lib.pygat_gat_backward_tail.argtypes = [i, i, i, i, i, p, p, p, p, i64, i, p, p, p]
    lib.pygat_head_mean.argtypes = [i, i, i, p, p, p, p]
    lib.pygat_gat_backward_prepare.argtypes = [i, i, i, i, i, p, p, p, p, p, p, p, p, p, f, p, i, i, i, p, p]
    lib.pygat_gat_backward_row.argtypes = [C.POINTER(Graph), i, i, f, p, p, p, p, p, p, i, i, i, p]
    lib.pygat_gat_backward_col.argtypes = [C.POINTER(Graph), p, i, i, f, p, p, p, p, p, p, p, p, p, p, i, i, i, p]
    lib.pygat_gat_backward_col_da_bytes.argtypes = [C.POINTER(Graph), i, i, i]
    lib.pygat_gat_backward_col_da_bytes.restype = sz
    lib.pygat_a_grad_fold.argtypes = [C.POINTER(Graph), i, i, p, p, p, p, p, p, i, p]
    lib.pygat_gat_backward_rowsum.argtypes = [C.POINTER(Graph), p, i, i, p, p, p, i, i, p]
    lib.pygat_agrad_workspace_bytes.argtypes = [i, i]
    lib.pygat_agrad_workspace_bytes.restype = sz
    lib.pygat_a_grad.argtypes = [i, i, i, p, p, p, p, p, p, p, p, i, i, p]
    lib.pygat_gatv2_forward.argtypes = [C.POINTER(Graph), i, i, f, i, p, p, p, p, p, p, p, p, p, p]
    lib.pygat_gatv2_backward_prepare.argtypes = [i, i, i, i, i, p, p, p, p, p, p, p, p, p]
    lib.pygat_gatv2_workspace_bytes.argtypes = [i64, i, i, i]
    lib.pygat_gatv2_workspace_bytes.restype = sz
    lib.pygat_gatv2_backward.argtypes = [C.POINTER(Graph), C.POINTER(Graph), p, p, i, i, f, p, p, p, p, p, p, p, p]
    u32 = C.c_uint32
    lib.pygat_wgrad_workspace_bytes.argtypes = [i, i, i, i]
    lib.pygat_wgrad_workspace_bytes.restype = sz
    lib.pygat_wgrad.argtypes = [i, i, i, i, p, i64, p, p, p, p, i, p, i, i, i, p]
    lib.pygat_dropout_mask.argtypes = [i64, f, p, u32, p, p]
    lib.pygat_dropout_mask2.argtypes = [f, p, i64, u32, p, i64, u32, p, p]
    lib.pygat_dropout_expand.argtypes = [i, i, i, p, i64, p, f, p, u32, p, i64, p]
    lib.pygat_dropout_head_sum.argtypes = [i, i, i, p, i64, p, f, p, u32, p, i64, i, p]
    lib.pygat_pack_blockdiag.argtypes = [i, i, i, p, p, p, i64, p]
    lib.pygat_unpack_blockdiag.argtypes = [i, i, i, p, i64, i, p, p]
    lib.pygat_headmask_supported.argtypes = [i, i, i]
    lib.pygat_dropout_bits.argtypes = [i, i, i, f, p, i, p, p]
    lib.pygat_project_dropout_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.pygat_project_dropout_workspace_bytes.restype = sz
    lib.pygat_project_dropout.argtypes = [i, i, i, i, p, i64, p, f, p, i64, p, p, i, p, p]
    lib.pygat_wgrad_dropout_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.pygat_wgrad_dropout_workspace_bytes.restype = sz
    lib.pygat_wgrad_dropout.argtypes = [i, i, i, i, p, i64, p, f, p, p, i64, p, i, p, p]
    lib.pygat_dropout_head_sum_bits.argtypes = [i, i, i, p, i64, p, f, p, i64, i, p]
    lib.pygat_project_sparse.argtypes = [i, i, i, i, p, p, p, p, i64, f, p, i, p, p, p, p, p]
    lib.pygat_wgrad_sparse.argtypes = [i, i, i, i, i, p, p, p, p, p, p, f, p, i, p, p, p, i64, p, p, p, p]
    lib.pygat_wgrad_sparse_workspace_bytes.argtypes = [i, i, i, i]
    lib.pygat_dropout_narrow.argtypes = [i, i, i, i]
    lib.pygat_bce_workspace_bytes.argtypes = [i64]
    lib.pygat_bce_workspace_bytes.restype = sz
    lib.pygat_bce_with_logits.argtypes = [i64, p, p, p, p, p]
    lib.pygat_bce_with_logits_backward.argtypes = [i64, p, p, p, p, p]
    lib.pygat_adam_step.argtypes = [i, p, p, p, p, p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, p, p]
    lib.pygat_dx_dropout.argtypes = [i, i, i, i, p, p, i64, p, f, p, i64, p, i64, i, p]
    lib.pygat_wgrad_sparse_workspace_bytes.restype = sz
    lib.pygat_nll_workspace_bytes.argtypes = [i]
    lib.pygat_nll_workspace_bytes.restype = sz
    lib.pygat_elu_logsoftmax_nll.argtypes = [i, i, p, i64, p, p, p, p, p]
    lib.pygat_elu_logsoftmax_nll_backward.argtypes = [i, i, p, i64, p, p, p, p, i64, p]
    for s in SYMBOLS:
        fn = getattr(lib, s)
        if fn.restype is C.c_int or s in ("pygat_abi_version", "pygat_padded_width", "pygat_device_count", "pygat_default_gemm_mode"):
            fn.restype = i
    if lib.pygat_abi_version() != ABI_VERSION:
        raise ImportError(f"pygat_amd: ABI version {lib.pygat_abi_version()} != {ABI_VERSION}")
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    """Turn a negative C return code into the reference-style Python error."""
    if rc == 0:
        return
    msg = lib.pygat_last_error().decode(errors="replace")
    if rc == -1:
        raise ValueError(f"pygat_amd {what}: {msg}")
    raise RuntimeError(f"pygat_amd {what}: {msg} (code {rc})")


def padded_width(f_out: int) -> int:
    fp = lib.pygat_padded_width(int(f_out))
    if fp == 0:
        raise ValueError(f"pygat_amd: head width {f_out} unsupported (1..256)")
    return fp
