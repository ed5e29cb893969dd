"""The streamed GEMMs in their two arithmetic modes (include/pygat_amd.h: the `gemm_mode` argument of the GEMM entry points).

'split-bf16' cuts every fp32 operand exactly into three bf16 pieces and sums all nine piece products into fp32
accumulators on the bf16 MFMA pipe (pygat_amd/csrc/k1_gemm_x3.hip); 'fp32-mfma' is v_mfma_f32_32x32x2_f32 throughout.
These tests hold the split mode to the fp32 mode's own accuracy against float64 (the parity bar of the level tests is
priced on that accuracy), and to EXACT results wherever fp32 arithmetic is exact -- the latter fails if
any of the 24 operand bits or any of the nine products went missing.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def pg():
    import pygat_amd as pg
    start = pg.get_gemm_mode()
    yield pg
    pg.set_gemm_mode(start)


def _run(pg, mode, tA, tB, M, N, K, A, B, **kw):
    C = torch.full((M, N), float("nan"), device="cuda")
    pg.gemm(tA, tB, M, N, K, A, A.shape[1], B, B.shape[1], [(N, C, N)], mode=mode, **kw)
    torch.cuda.synchronize()
    return C


# (transA, transB, M, N, K): projection shapes (K small, M = nodes) incl. ragged M / N, the input-gradient form
# (transB), weight-gradient shapes (K = nodes) incl. a K tail and ragged M / N
SHAPES = [
    (False, False, 20000, 128, 128), (False, False, 8192 + 77, 136, 64), (False, False, 9001, 72, 32),
    (False, False, 33000, 160, 96), (False, False, 12345, 300, 128), (False, True, 20000, 128, 128),
    (False, True, 10007, 100, 64), (False, True, 70001, 100, 64), (False, False, 70001, 72, 128), (False, False, 300000, 128, 64),
    (True, False, 128, 128, 100000), (True, False, 100, 200, 65537), (True, False, 72, 130, 50001),
    (True, False, 256, 256, 20011),
    # the general split kernel (gemm_x3g_kernel: any layout, K > 256 or few rows): the PPI levels' projections, input and
    # weight gradients, ragged M / N, K tails of a step (K % 16 != 0) and of the accumulator flush period
    (False, False, 3144, 2056, 1024), (False, True, 3144, 1024, 1024), (True, False, 1024, 1024, 3144),
    (False, False, 1612, 1548, 1024), (False, True, 1000, 1024, 768), (True, False, 1024, 260, 3144),
    (False, False, 131, 132, 36), (False, True, 65, 67, 100), (True, False, 68, 72, 333), (False, False, 4097, 520, 260),
    (False, True, 20000, 128, 512),
]


@pytest.mark.parametrize("tA,tB,M,N,K", SHAPES)
def test_split_mode_is_as_accurate_as_fp32_mfma(pg, tA, tB, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + 7 * N + K)
    A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
    B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
    # a spread of magnitudes inside one sum, and exact zeros
    A *= torch.exp2(torch.randint(-6, 7, A.shape, device="cuda", generator=g).float())
    A[::7] = 0
    ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double()
    scale = float(ref.abs().max())
    err = {}
    for mode in ("fp32-mfma", "split-bf16"):
        C = _run(pg, mode, tA, tB, M, N, K, A, B)
        assert torch.isfinite(C).all()
        err[mode] = float((C.double() - ref).abs().max()) / scale
    print(f"\n  {'T' if tA else 'N'}{'T' if tB else 'N'} {M}x{N}x{K}: fp32-mfma {err['fp32-mfma']:.2e}  split-bf16 {err['split-bf16']:.2e}")
    assert err["fp32-mfma"] < 1e-5
    # same accuracy class: the two modes differ in the ORDER of fp32 additions only (a factor 2 covers that)
    assert err["split-bf16"] <= 2.0 * err["fp32-mfma"] + 1e-7


@pytest.mark.parametrize("tA,big", [(False, False), (True, False), (False, True)])
def test_split_mode_is_exact_where_fp32_is(pg, tA, big):
    """Operands with all 24 significant bits set at random against a signed power-of-two selection matrix: every
    product and every sum is exact in fp32, so the result must equal the float64 product BIT FOR BIT in both modes.
    A lost low piece (or a lost piece product) shows up as a wrong last bit."""
    # big: 1024 row tiles over 256 persistent work-groups -- the straight-line tile loop with its prefetch ring and more
    # than 63 memory operations in flight per wave
    M, N, K = (128, 128, 65536) if tA else ((262144 + 77, 128, 128) if big else (16384, 128, 128))
    g = torch.Generator(device="cuda").manual_seed(5)
    rows, cols = ((K, M) if tA else (M, K))
    mant = torch.randint(1 << 23, 1 << 24, (rows, cols), device="cuda", generator=g).float()   # 24-bit integers, exact
    expo = torch.randint(-20, 21, (rows, cols), device="cuda", generator=g).float()
    sign = torch.randint(0, 2, (rows, cols), device="cuda", generator=g).float() * 2 - 1
    A = sign * mant * torch.exp2(expo)
    # B: one nonzero per output column (so a sum has a single term), a signed power of two
    B = torch.zeros(K, N, device="cuda")
    pick = torch.randint(0, K, (N,), device="cuda", generator=g)
    B[pick, torch.arange(N, device="cuda")] = torch.exp2(torch.randint(-8, 9, (N,), device="cuda", generator=g).float()) * \
        (torch.randint(0, 2, (N,), device="cuda", generator=g).float() * 2 - 1)
    ref = ((A.t() if tA else A).double() @ B.double()).float()
    for mode in ("fp32-mfma", "split-bf16"):
        C = _run(pg, mode, tA, False, M, N, K, A, B)
        assert torch.equal(C, ref), mode
    # and the mirror image: full-mantissa B against a selecting A
    Bf = (torch.randint(1 << 23, 1 << 24, (K, N), device="cuda", generator=g).float() *
          torch.exp2(torch.randint(-20, 21, (K, N), device="cuda", generator=g).float()))
    As = torch.zeros(rows, cols, device="cuda")
    if tA:
        As[torch.randint(0, K, (M,), device="cuda", generator=g), torch.arange(M, device="cuda")] = -2.0
    else:
        As[torch.arange(M, device="cuda"), torch.randint(0, K, (M,), device="cuda", generator=g)] = -2.0
    ref = ((As.t() if tA else As).double() @ Bf.double()).float()
    for mode in ("fp32-mfma", "split-bf16"):
        C = _run(pg, mode, tA, False, M, N, K, As, Bf)
        assert torch.equal(C, ref), mode


@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False)])
def test_general_split_kernel_is_exact_where_fp32_is(pg, tA, tB):
    """The same bit-exactness argument for the general split kernel (K = 1024, a PPI-sized operand), every operand layout:
    full-mantissa values on one side against a signed power-of-two selection matrix on the other, both ways round, plus
    split-K slabs and accumulation onto an existing C (exact as long as C holds a multiple of the result's ulp: 0 here)."""
    M, N, K = 1612, 1028, 1024
    g = torch.Generator(device="cuda").manual_seed(17)

    def full(r, c):
        return ((torch.randint(0, 2, (r, c), device="cuda", generator=g).float() * 2 - 1) *
                torch.randint(1 << 23, 1 << 24, (r, c), device="cuda", generator=g).float() *
                torch.exp2(torch.randint(-20, 21, (r, c), device="cuda", generator=g).float()))

    def select(rows, cols, along_rows):
        """one nonzero (a signed power of two) per row (along_rows) or per column of a [rows, cols] matrix"""
        S = torch.zeros(rows, cols, device="cuda")
        n = rows if along_rows else cols
        val = torch.exp2(torch.randint(-8, 9, (n,), device="cuda", generator=g).float()) * \
            (torch.randint(0, 2, (n,), device="cuda", generator=g).float() * 2 - 1)
        if along_rows:
            S[torch.arange(rows, device="cuda"), torch.randint(0, cols, (rows,), device="cuda", generator=g)] = val
        else:
            S[torch.randint(0, rows, (cols,), device="cuda", generator=g), torch.arange(cols, device="cuda")] = val
        return S
    opA = lambda T: T.t().contiguous() if tA else T      # noqa: E731  stored form of a logical [M, K] / [K, N] operand
    opB = lambda T: T.t().contiguous() if tB else T      # noqa: E731
    cases = [(full(M, K), select(K, N, False)),           # every output = one full-mantissa A value x a power of two
             (select(M, K, True), full(K, N))]
    for Al, Bl in cases:
        ref = (Al.double() @ Bl.double()).float()
        A, B = opA(Al), opB(Bl)
        for mode in ("fp32-mfma", "split-bf16"):
            for sk in (1, 3):
                C = _run(pg, mode, tA, tB, M, N, K, A, B, split_k=sk)
                assert torch.equal(C, ref), (mode, sk)
        C = torch.zeros(M, N, device="cuda")
        pg.gemm(tA, tB, M, N, K, A, A.shape[1], B, B.shape[1], [(N // 2, C, N), (N - N // 2, C[:, N // 2:], N)], accumulate=True,
                split_k=1, mode="split-bf16")
        assert torch.equal(C, ref)


def test_general_split_kernel_padded_rows(pg):
    """A k-strided operand whose extent is not a multiple of 4 but whose rows are padded to one (the projection's Wcat: 2 R + H
    = 1542 columns of PPI level 3 in rows of 1544): the kernel reads up to 3 columns into the padding, whatever it holds --
    NaN here -- without it reaching C."""
    g = torch.Generator(device="cuda").manual_seed(23)
    M, N, K, ldb = 3144, 1542, 1024, 1544
    A = torch.randn(M, K, device="cuda", generator=g)
    Bp = torch.full((K, ldb), float("nan"), device="cuda")
    Bp[:, :N] = torch.randn(K, N, device="cuda", generator=g)
    ref = A.double() @ Bp[:, :N].double()
    err = {}
    for mode in ("fp32-mfma", "split-bf16"):
        C = torch.full((M, N + 3), 7.0, device="cuda")
        pg.gemm(False, False, M, N, K, A, K, Bp, ldb, [(N, C, N + 3)], mode=mode)
        assert torch.isfinite(C).all() and bool((C[:, N:] == 7.0).all()), mode
        err[mode] = float((C[:, :N].double() - ref).abs().max() / ref.abs().max())
    assert err["fp32-mfma"] < 1e-5 and err["split-bf16"] <= 2.0 * err["fp32-mfma"] + 1e-7, err
    # the transposed-A form (weight gradient of a 121-wide head table): M = 726 + 2 rows of padding
    Mt, ldat = 726 + 1, 728
    At = torch.full((K, ldat), float("nan"), device="cuda"); At[:, :Mt] = torch.randn(K, Mt, device="cuda", generator=g)
    Bt = torch.randn(K, 256, device="cuda", generator=g)
    reft = At[:, :Mt].double().t() @ Bt.double()
    Ct = torch.empty(Mt, 256, device="cuda")
    pg.gemm(True, False, Mt, 256, K, At, ldat, Bt, 256, [(256, Ct, 256)], mode="split-bf16", split_k=1)
    assert float((Ct.double() - reft).abs().max() / reft.abs().max()) < 1e-6


def test_small_integers_sum_exactly(pg):
    """Integer operands whose dot products stay below 2^24: any summation order is exact, both modes must give the
    integer result."""
    g = torch.Generator(device="cuda").manual_seed(9)
    for tA, M, N, K in ((False, 10000, 144, 96), (True, 96, 144, 30000)):
        A = torch.randint(-15, 16, (K, M) if tA else (M, K), device="cuda", generator=g).float()
        B = torch.randint(-15, 16, (K, N), device="cuda", generator=g).float()
        ref = ((A.t() if tA else A).double() @ B.double()).float()
        for mode in ("fp32-mfma", "split-bf16"):
            assert torch.equal(_run(pg, mode, tA, False, M, N, K, A, B), ref), (mode, tA)


@pytest.mark.parametrize("n,Fin,H,Fo,with_a", [(30011, 128, 8, 16, True), (30011, 128, 8, 16, False), (70003, 64, 8, 8, True),
                                                (70003, 64, 4, 16, True), (20000, 128, 6, 7, True), (20000, 96, 8, 16, True),
                                                (40001, 128, 8, 7, True), (40001, 64, 16, 13, True)])
def test_projection_with_s_columns_both_modes(pg, n, Fin, H, Fo, with_a):
    """pygat_project (Wh, s in one launch) gives the same level in both modes: s on the VALU of the streaming lanes
    (no a_pad), or -- split mode, heads of 8 / 16 columns, Fin 64 / 128 -- from the Wh accumulators in the epilogue."""
    from pygat_amd import ops
    from pygat_amd._lib import lib, check
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, Fin, device="cuda", generator=g)
    W = torch.randn(H, Fin, Fo, device="cuda", generator=g) * 0.2
    a = torch.randn(H, 2 * Fo, device="cuda", generator=g)
    Fp = pg.padded_width(Fo)
    R = H * Fp
    ldw = -(-(R + 2 * H) // 4) * 4
    Wcat = torch.empty(Fin, ldw, device="cuda"); a_pad = torch.empty(H, 2, Fp, device="cuda")
    check(lib.pygat_pack_params(H, Fin, Fo, W.data_ptr(), a.data_ptr(), None, Wcat.data_ptr(), ldw, a_pad.data_ptr(), None),
          "pack")
    Wh64 = torch.einsum("nk,hkf->nhf", x.double(), W.double())
    s64 = torch.einsum("nhf,hf->nh", Wh64, a[:, :Fo].double())
    res = {}
    for mode in ("fp32-mfma", "split-bf16"):
        Wh = torch.full((n, R), float("nan"), device="cuda"); s = torch.full((n, H), float("nan"), device="cuda")
        check(lib.pygat_project(n, Fin, H, Fo, x.data_ptr(), Fin, Wcat.data_ptr(), ldw, a_pad.data_ptr() if with_a else None,
                                Wh.data_ptr(), None, s.data_ptr(), 1, None, pg.ops.GEMM_MODES[mode], None), "project")
        torch.cuda.synchronize()
        eW = float((Wh.view(n, H, Fp)[:, :, :Fo].double() - Wh64).abs().max() / Wh64.abs().max())
        eS = float((s.double() - s64).abs().max() / s64.abs().max())
        res[mode] = (eW, eS)
    print("\n  projection (Wh, s) rel err:", res)
    assert res["fp32-mfma"][0] < 1e-5 and res["split-bf16"][0] <= 2 * res["fp32-mfma"][0] + 1e-7
    assert res["split-bf16"][1] <= 2 * res["fp32-mfma"][1] + 1e-7


def test_projection_with_skip_both_modes(pg):
    """[Wh | Sk | s] with a skip projection: in split mode Wh and s come from the accumulator kernel, Sk from a second
    launch over the skip columns of Wcat."""
    from pygat_amd._lib import lib, check
    n, Fin, H, Fo = 25013, 128, 8, 16
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(n, Fin, device="cuda", generator=g)
    W = torch.randn(H, Fin, Fo, device="cuda", generator=g) * 0.2
    Ws = torch.randn(H, Fin, Fo, device="cuda", generator=g) * 0.2
    a = torch.randn(H, 2 * Fo, device="cuda", generator=g)
    Fp = pg.padded_width(Fo)
    R = H * Fp
    ldw = -(-(2 * R + 2 * H) // 4) * 4
    Wcat = torch.empty(Fin, ldw, device="cuda"); a_pad = torch.empty(H, 2, Fp, device="cuda")
    check(lib.pygat_pack_params(H, Fin, Fo, W.data_ptr(), a.data_ptr(), Ws.data_ptr(), Wcat.data_ptr(), ldw, a_pad.data_ptr(), None),
          "pack")
    Wh64 = torch.einsum("nk,hkf->nhf", x.double(), W.double())
    Sk64 = torch.einsum("nk,hkf->nhf", x.double(), Ws.double())
    s64 = torch.einsum("nhf,hf->nh", Wh64, a[:, :Fo].double())
    for mode in ("fp32-mfma", "split-bf16"):
        Wh = torch.full((n, R), float("nan"), device="cuda"); Sk = torch.full((n, R), float("nan"), device="cuda")
        s = torch.full((n, H), float("nan"), device="cuda")
        check(lib.pygat_project(n, Fin, H, Fo, x.data_ptr(), Fin, Wcat.data_ptr(), ldw, a_pad.data_ptr(), Wh.data_ptr(), Sk.data_ptr(),
                                s.data_ptr(), 1, None, pg.ops.GEMM_MODES[mode], None), "project")
        torch.cuda.synchronize()
        for got, ref, what in ((Wh.view(n, H, Fp)[:, :, :Fo], Wh64, "Wh"), (Sk.view(n, H, Fp)[:, :, :Fo], Sk64, "Sk"), (s, s64, "s")):
            err = float((got.double() - ref).abs().max() / ref.abs().max())
            assert err < 2e-6, (mode, what, err)
