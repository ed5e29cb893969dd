"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/pygat_amd.h declares; argument validation works without a GPU (no compute is launched)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "pygat_amd", "libpygat_amd.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "pygat_amd", "csrc"), "-j4"], check=True, capture_output=True)
    from pygat_amd import _lib
    return _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "pygat_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pygat_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert len(names) >= 20
    raw = C.CDLL(lib.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert sorted(lib.SYMBOLS) == names       # the python binding covers the whole header


def test_head_windows_host_logic(lib, monkeypatch):
    """pygat_head_group is pure host code (csrc/attn_common.h head_group_bwd): whole rows up to 256 floats; a table beyond the
    caches (>= 256 MB) in windows of 256 floats; a cache-resident one whole up to 512 floats, in windows of 512 beyond."""
    hg = lib.lib.pygat_head_group
    assert hg(1000, 8, 16) == 8 and hg(1 << 20, 8, 16) == 8 and hg(1 << 20, 8, 32) == 8      # R <= 256
    assert hg(1000, 8, 64) == 8 and hg(1 << 20, 8, 64) == 4      # R = 512: whole on a small table, 2 x 256 on a large one
    assert hg(3000, 4, 256) == 2                       # PPI-sized: table 12 MB, rows of 1024 floats -> two windows of 512
    assert hg(1 << 20, 8, 128) == 2 and hg(1 << 20, 4, 256) == 1 and hg(1 << 20, 6, 121) == 2
    assert hg(1000, 12, 128) == 4                      # R = 1536 on a small table: windows of 512 floats
    assert hg(0, 8, 16) == 0 and hg(10, 0, 16) == 0 and hg(10, 8, 300) == 0
    # a pure function of its arguments: no environment variable moves it (ABI 13; PYGAT_BWD_WINDOW_BYTES used to)
    monkeypatch.setenv("PYGAT_BWD_WINDOW_BYTES", "0")
    assert hg(10, 6, 121) == 4 and hg(10, 8, 64) == 8


def test_abi_version_and_padding(lib):
    assert lib.lib.pygat_abi_version() == lib.ABI_VERSION
    assert [lib.lib.pygat_padded_width(f) for f in (1, 3, 4, 7, 8, 16, 121, 256, 257, 0)] == \
        [4, 4, 4, 8, 8, 16, 128, 256, 0, 0]


def test_argument_validation_returns_einval(lib):
    L = lib.lib
    # null pointers / bad sizes are rejected before anything is launched
    assert L.pygat_dense_row_counts(None, 4, 4, 0, None, None) == -1
    assert b"bad arguments" in L.pygat_last_error()
    assert L.pygat_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 0, 1, None, -1, None) == -1
    g = lib.Graph(0, 0, None, None, 64, None, None, 0, 0)
    assert L.pygat_gat_forward(C.byref(g), 8, 16, 0.2, 1, None, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert b"graph" in L.pygat_last_error()
    g2 = lib.Graph(4, 4, 1, 1, 6, None, None, 0, 0)             # slot_edges not a multiple of 4
    assert L.pygat_gat_forward(C.byref(g2), 8, 16, 0.2, 1, None, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert b"slot_edges" in L.pygat_last_error()
    with pytest.raises(ValueError):
        lib.check(-1, "x")
    with pytest.raises(RuntimeError):
        lib.check(-2, "x")
    assert lib.lib.pygat_partials_bytes(1000, 64, 8, 16) == 2 * 16 * (2 * 128 + 3 * 8) * 4   # widest record: training forward
    assert lib.lib.pygat_gemm_workspace_bytes(128, 128, 4) == 4 * 128 * 128 * 4


QUERIES = {"pygat_abi_version", "pygat_last_error", "pygat_padded_width", "pygat_device_count", "pygat_device_name",
           "pygat_scan_workspace_bytes", "pygat_gemm_workspace_bytes", "pygat_partials_bytes", "pygat_head_group",
           "pygat_agrad_workspace_bytes", "pygat_gatv2_workspace_bytes", "pygat_wgrad_workspace_bytes",
           "pygat_headmask_supported", "pygat_project_dropout_workspace_bytes", "pygat_wgrad_dropout_workspace_bytes",
           "pygat_default_gemm_mode", "pygat_nll_workspace_bytes", "pygat_wgrad_sparse_workspace_bytes", "pygat_dropout_narrow", "pygat_bce_workspace_bytes",
           "pygat_gat_backward_col_da_bytes", "pygat_gat_forward_phases_ok"}


def test_gemm_mode_is_a_call_argument(lib, monkeypatch):
    """The product mode of a GEMM is an ARGUMENT of the entry points that run one (ABI 11): the library exports no
    setter -- no process-global mutable state besides the thread-local error string -- only the immutable default
    (split-bf16 unless PYGAT_GEMM_F32=1 was in the environment at load).  Unknown modes are refused per call.  The
    Python mirror keeps the choice per THREAD and hands it down with every call."""
    import threading
    L = lib.lib
    assert not hasattr(L, "pygat_set_gemm_mode") and not hasattr(L, "pygat_get_gemm_mode")
    assert L.pygat_default_gemm_mode() in (0, 1)
    seg = lib.OutSegments()
    assert L.pygat_gemm_f32(0, 0, 4, 4, 4, 1, 4, 1, 4, C.byref(seg), 0, 1, None, 7, None) == -1
    assert b"unknown product mode" in L.pygat_last_error()
    assert L.pygat_project(4, 4, 1, 4, 1, 4, 1, 8, None, 1, None, 1, 1, None, -2, None) == -1
    assert b"unknown product mode" in L.pygat_last_error()
    assert L.pygat_wgrad(4, 4, 1, 4, 1, 4, 1, None, None, 1, 1, 1, 0, 0, 2, None) == -1
    assert b"unknown product mode" in L.pygat_last_error()
    from pygat_amd import ops
    ops.set_gemm_mode(None)
    assert ops._mode_code() == -1 and ops.get_gemm_mode() == ("split-bf16", "fp32-mfma")[L.pygat_default_gemm_mode()]
    seen = {}
    with ops.gemm_mode("fp32-mfma"):
        assert ops.get_gemm_mode() == "fp32-mfma" and ops._mode_code() == 1 and ops._mode_code("split-bf16") == 0
        t = threading.Thread(target=lambda: seen.setdefault("other", ops._mode_code()))   # another thread: its own choice
        t.start(); t.join()
    assert seen["other"] == -1 and ops._mode_code() == -1
    with pytest.raises(ValueError):
        ops.set_gemm_mode("fp16")


def test_every_launcher_rejects_null_arguments(lib):
    """Each compute entry point, called with null pointers and zero sizes, returns a negative code and
    leaves a message -- it neither launches nor crashes (the checks run before any HIP call)."""
    L = lib.lib
    launchers = [n for n in lib.SYMBOLS if n not in QUERIES]
    assert len(launchers) >= 18
    for name in launchers:
        fn = getattr(L, name)
        args = []
        for t in fn.argtypes:
            if t in (C.c_int, C.c_int64, C.c_size_t, C.c_uint32):
                args.append(0)
            elif t in (C.c_float, C.c_double):
                args.append(0.0)
            else:                      # void*, pygat_graph*, pygat_out_segments*
                args.append(None)
        rc = fn(*args)
        assert rc < 0, (name, rc)
        assert len(L.pygat_last_error()) > 0, name
    # size queries answer 0 for nonsense instead of failing
    assert L.pygat_partials_bytes(0, 64, 8, 16) == 0 and L.pygat_agrad_workspace_bytes(0, 16) == 0


def test_no_cpu_fallback(lib):
    """The product path fails loudly off-GPU instead of silently computing elsewhere."""
    import torch
    import pygat_amd as pg
    x = torch.randn(4, 3)
    with pytest.raises((ValueError, RuntimeError)):
        pg.CSRGraph(torch.tensor([0, 1, 2, 3, 4], dtype=torch.int32), torch.tensor([0, 1, 2, 3], dtype=torch.int32))
    layer = pg.SpGraphAttentionLayer(3, 2, 0.0, 0.2)
    with pytest.raises((ValueError, RuntimeError)):
        layer(x, torch.eye(4))


def test_dropin_constructor_and_state_dict_cpu(lib):
    """Parameter names, shapes and initialisers of the drop-ins (reference layers.py:21-28,111-119,
    models.py:27) -- checked on CPU, no kernel involved."""
    import torch
    import pygat_amd as pg
    torch.manual_seed(0)
    d = pg.GraphAttentionLayer(in_features=5, out_features=3, dropout=0.6, alpha=0.2, concat=True, skip_connection=True)
    s = pg.SpGraphAttentionLayer(in_features=5, out_features=3, dropout=0.6, alpha=0.2, concat=False)
    assert tuple(d.W.shape) == (5, 3) and tuple(d.a.shape) == (6, 1) and tuple(d.skip_projection.shape) == (5, 3)
    assert tuple(s.W.shape) == (5, 3) and tuple(s.a.shape) == (1, 6) and not hasattr(s, "skip_projection")
    assert repr(d) == "GraphAttentionLayer (5 -> 3)" and repr(s) == "SpGraphAttentionLayer (5 -> 3)"
    # xavier_uniform with gain 1.414 bounds |W| by gain*sqrt(6/(fan_in+fan_out))
    assert float(d.W.abs().max()) <= 1.414 * (6 / 8) ** 0.5 + 1e-6
    m = pg.GAT(nfeat=[50, 256, 256, 121], nheads=[4, 4, 6], nlayers=3, dropout=0.0, alpha=0.2,
               layer_type=pg.GraphAttentionLayer, skip_connection=True)
    keys = set(m.state_dict())
    assert "attention_layer_3_head_6.skip_projection" in keys and len(keys) == (4 + 4 + 6) * 3
    assert tuple(m.state_dict()["attention_layer_2_head_1.W"].shape) == (1024, 256)
    assert tuple(m.state_dict()["attention_layer_3_head_1.W"].shape) == (1024, 121)


def test_host_policies(lib, monkeypatch):
    """Pure host decisions of the python layer (no kernel involved): slot length by graph size, backward flavour by
    row width, K slabs of the GEMM wrapper."""
    from pygat_amd import ops
    from pygat_amd.graph import auto_slot_edges, slot_edges_for
    assert auto_slot_edges(13_264) == 4 and auto_slot_edges(108_365) == 8 and auto_slot_edges(10_758_702) == 64
    assert slot_edges_for(16, 64) == 32 and slot_edges_for(128, 64) == 64 and slot_edges_for(16, 8) == 8
    monkeypatch.setattr(ops, "TWO_GATHER_BACKWARD", None)
    assert ops.two_gather_backward(16) and ops.two_gather_backward(32) and not ops.two_gather_backward(64)
    monkeypatch.setattr(ops, "TWO_GATHER_BACKWARD", True)
    assert ops.two_gather_backward(1024)
    monkeypatch.setattr(ops, "TWO_GATHER_BACKWARD", False)
    assert not ops.two_gather_backward(16)
    # flavour of the backward: row-local by default, the older switch selects between the two older flavours
    monkeypatch.setattr(ops, "BACKWARD_FLAVOUR", None)
    assert ops.backward_flavour(16) == "rowsum"
    monkeypatch.setattr(ops, "TWO_GATHER_BACKWARD", True)
    assert ops.backward_flavour(128) == "two-gather"
    monkeypatch.setattr(ops, "TWO_GATHER_BACKWARD", None)
    assert ops.backward_flavour(16) == ops.backward_flavour(256) == "rowlocal" and ops.backward_flavour(512) == "rowsum"
    monkeypatch.setattr(ops, "BACKWARD_FLAVOUR", "rowsum")
    assert ops.backward_flavour(128) == "rowsum"
    # streamed-K weight gradient: slabs x column tiles = 256 (one work-group per CU)
    mode = ops.get_gemm_mode()
    try:
        ops.set_gemm_mode("fp32-mfma")
        assert ops._split_k(128, 128, 1 << 20, streamed_k=True) == 256
        assert ops._split_k(128, 136, 1 << 20, streamed_k=True) == 256      # [dWh | ds]: still one 5-tile column block
        assert ops._split_k(128, 520, 1 << 20, streamed_k=True) == 51
        ops.set_gemm_mode("split-bf16")                                      # 128 x 128 tiles through LDS: two work-groups per CU
        assert ops._split_k(128, 128, 1 << 20, streamed_k=True) == 512
        assert ops._split_k(256, 256, 1 << 20, streamed_k=True) == 128
        assert ops._split_k(128, 16, 1 << 20, streamed_k=True) == 256       # narrow outputs keep the fp32 kernel
    finally:
        ops.set_gemm_mode(mode)
    # general kernel: the dropout projection shapes get more than "one work-group per CU"
    assert ops._split_k(2708, 64, 11464) == 22 and ops._split_k(11464, 64, 2708, streamed_k=True) == 5
    assert ops._split_k(1 << 20, 128, 128) == 1


def test_python_switches_are_read_once_in_one_place():
    """the development knobs of the Python layer live in pygat_amd.config (read from the environment once, at
    import); no other product module reads os.environ while it runs."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pygat_amd")
    offenders = []
    for path in sorted(glob.glob(os.path.join(root, "*.py"))):
        if os.path.basename(path) == "config.py":
            continue
        for n, line in enumerate(open(path), 1):
            code = line.split("#", 1)[0]
            if re.search(r"os\.environ|getenv", code):
                offenders.append(f"{os.path.basename(path)}:{n}")
    assert not offenders, offenders
    from pygat_amd.config import Config, config
    assert config.describe().keys() == Config.from_env().describe().keys() and config.dist_chunks >= 1
