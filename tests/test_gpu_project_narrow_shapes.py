"""pygat_project on the shapes of the citation models' second level (64 inputs, 1 x 7 / 8 x 3 outputs) and other narrow
levels: [Wh | Sk | s] = X Wcat against the fp64 product of the same packed operand.  (A vector-ALU kernel for these shapes
was tried and dropped -- DESIGN.md section 8 -- the MFMA kernels keep them; the dropout kernels of csrc/k10_narrow.hip are
covered by tests/test_gpu_dropout.py::test_headmask_projection_and_weight_gradient.)"""
import pytest
import torch

from parity import close_grad

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Fin,H,Fo,skip", [(2708, 64, 1, 7, False), (3000, 128, 8, 8, True), (501, 20, 3, 5, True),
                                            (1000, 64, 8, 3, False), (40000, 64, 8, 8, False),
                                            # the streamed kernel with s on the VALU, one / two column tiles, up to four and more s columns,
                                            # ragged row counts (the shards of a head-parallel level)
                                            (20000, 128, 1, 16, False), (20011, 64, 2, 7, False), (9001, 128, 3, 5, False),
                                            (30000, 128, 1, 7, True), (12345, 128, 6, 5, False), (8200, 64, 5, 3, True)])
def test_plain_projection(N, Fin, H, Fo, skip):
    import pygat_amd as pg
    from pygat_amd._lib import lib, check
    dev = torch.device("cuda", 0)
    Fp = pg.padded_width(Fo); R = H * Fp
    nw = R * (2 if skip else 1)
    gen = torch.Generator().manual_seed(N + Fin)
    x = torch.randn(N, Fin, generator=gen)
    W = torch.randn(H, Fin, Fo, generator=gen) * 0.3
    a = torch.randn(H, 2 * Fo, generator=gen) * 0.3
    Ws = torch.randn(H, Fin, Fo, generator=gen) * 0.3 if skip else None
    xd, Wd, ad = x.to(dev), W.to(dev), a.to(dev)
    Wsd = Ws.to(dev) if skip else None
    ldw = -(-(nw + 2 * H) // 4) * 4
    Wcat = torch.zeros(Fin, ldw, device=dev); a_pad = torch.empty(H, 2, Fp, device=dev)
    check(lib.pygat_pack_params(H, Fin, Fo, Wd.data_ptr(), ad.data_ptr(), Wsd.data_ptr() if skip else None, Wcat.data_ptr(), ldw,
                                a_pad.data_ptr(), None))
    Wh = torch.full((N, R), float("nan"), device=dev)
    Sk = torch.full((N, R), float("nan"), device=dev) if skip else None
    s = torch.full((N, H), float("nan"), device=dev)
    check(lib.pygat_project(N, Fin, H, Fo, xd.data_ptr(), Fin, Wcat.data_ptr(), ldw, a_pad.data_ptr(), Wh.data_ptr(),
                            Sk.data_ptr() if skip else None, s.data_ptr(), 1, None, -1, None))
    Wc = Wcat.cpu()
    ref64 = x.double() @ Wc.double(); ref32 = (x @ Wc).double()
    close_grad(Wh, ref64[:, :R], ref32[:, :R], "Wh")
    if skip:
        close_grad(Sk, ref64[:, R:nw], ref32[:, R:nw], "Sk")
    # s = Wh . a_src (layers.py:60): either as the packed columns W a_src of the operand, or from the Wh rows in the epilogue
    s64 = torch.einsum("nhf,hf->nh", (x.double() @ W.double().permute(1, 0, 2).reshape(Fin, H * Fo)).view(N, H, Fo), a[:, :Fo].double())
    s32 = torch.einsum("nhf,hf->nh", (x @ W.permute(1, 0, 2).reshape(Fin, H * Fo)).view(N, H, Fo), a[:, :Fo]).double()
    close_grad(s, s64, s32, "s")
    for h in range(H):
        assert float(Wh.view(N, H, Fp)[:, h, Fo:].abs().max() if Fp > Fo else 0.0) == 0.0
