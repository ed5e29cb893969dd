"""The training loss of the reference's citation-network script as two HIP launches.

train.py:151-152,159 computes `F.nll_loss(F.log_softmax(F.elu(model(x, adj)), dim=1)[idx_train], labels[idx_train])`;
ATen spends 17 launches on it per training step (an index_put with a radix sort among them) in an epoch whose attention
kernels take microseconds.  `EluLogSoftmaxNLL(idx, labels, n)(out)` is the same value and the same gradient in one launch
forward, one backward (csrc/k8_loss.hip).  No CPU path: the tensors must live on the GPU.
"""
from __future__ import annotations

import torch

from ._lib import lib, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _NLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, crit):
        if not out.is_cuda:
            raise RuntimeError("pygat_amd: EluLogSoftmaxNLL needs GPU tensors; there is no CPU path")
        out = out.contiguous().float()
        n, C = out.shape
        if n != crit.n:
            raise ValueError(f"EluLogSoftmaxNLL: built for {crit.n} rows, got {n}")
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        with torch.cuda.device(out.device):
            check(lib.pygat_elu_logsoftmax_nll(n, C, out.data_ptr(), C, crit.label.data_ptr(), crit.weight.data_ptr(),
                                               crit.ws.data_ptr(), loss.data_ptr(), _stream()), "elu_logsoftmax_nll")
        ctx.save_for_backward(out)
        ctx.crit = crit
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        crit = ctx.crit
        n, C = out.shape
        g = g.reshape(1).float().contiguous()
        dout = torch.empty_like(out)
        with torch.cuda.device(out.device):
            check(lib.pygat_elu_logsoftmax_nll_backward(n, C, out.data_ptr(), C, crit.label.data_ptr(), crit.weight.data_ptr(),
                                                        g.data_ptr(), dout.data_ptr(), C, _stream()),
                  "elu_logsoftmax_nll_backward")
        return dout, None


class EluLogSoftmaxNLL:
    """loss(out) == F.nll_loss(F.log_softmax(F.elu(out), dim=1)[idx], labels[idx])  (train.py:151-152,159).

    idx: the rows the loss is taken over (idx_train / idx_val / idx_test; repeats count as often as they occur),
    labels: [n] class per row, n: rows of the model output."""

    def __init__(self, idx: torch.Tensor, labels: torch.Tensor, n: int):
        if not (idx.is_cuda and labels.is_cuda):
            raise RuntimeError("pygat_amd: EluLogSoftmaxNLL needs GPU tensors; there is no CPU path")
        self.n = int(n)
        idx = idx.long()
        # one-time host-side checks (F.nll_loss raises on these; the kernels would read out of bounds or drop rows silently)
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.n):
            raise IndexError(f"EluLogSoftmaxNLL: idx outside [0, {self.n})")
        if labels.numel() < self.n:
            raise ValueError(f"EluLogSoftmaxNLL: {labels.numel()} labels for {self.n} rows")
        self.weight = (torch.bincount(idx, minlength=self.n).float() / max(1, idx.numel())).contiguous()
        self.label = labels.to(torch.int32).contiguous()
        self._label_range = ((int(labels[idx].min()), int(labels[idx].max())) if idx.numel() else (0, 0))   # of the weighted rows
        self._checked_C = None
        self.ws = torch.zeros(lib.pygat_nll_workspace_bytes(self.n) // 4, dtype=torch.float32, device=idx.device)

    def __call__(self, out: torch.Tensor) -> torch.Tensor:
        C_ = out.shape[1]
        if self._checked_C != C_:        # the class count is only known from the output: checked once per width, no sync
            if out.shape[0] != self.n:
                raise ValueError(f"EluLogSoftmaxNLL: output has {out.shape[0]} rows, the criterion was built for {self.n}")
            if self._label_range[0] < 0 or self._label_range[1] >= C_:
                raise IndexError(f"EluLogSoftmaxNLL: labels of the weighted rows span {self._label_range}, outside [0, {C_})")
            self._checked_C = C_
        return _NLLFn.apply(out, self)


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, crit):
        if not logits.is_cuda:
            raise RuntimeError("pygat_amd: BCEWithLogits needs GPU tensors; there is no CPU path")
        x = logits.contiguous().float()
        if x.shape != crit.target.shape:
            raise ValueError(f"BCEWithLogits: logits {tuple(x.shape)} vs targets {tuple(crit.target.shape)}")
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.pygat_bce_with_logits(x.numel(), x.data_ptr(), crit.target.data_ptr(), crit.ws.data_ptr(), loss.data_ptr(),
                                            _stream()), "bce_with_logits")
        ctx.save_for_backward(x)
        ctx.crit, ctx.in_dtype = crit, logits.dtype
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.reshape(1).float().contiguous()
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.pygat_bce_with_logits_backward(x.numel(), x.data_ptr(), ctx.crit.target.data_ptr(), g.data_ptr(),
                                                     dx.data_ptr(), _stream()), "bce_with_logits_backward")
        return (dx if ctx.in_dtype == torch.float32 else dx.to(ctx.in_dtype)), None


class BCEWithLogits:
    """loss(logits) == nn.BCEWithLogitsLoss(reduction='mean')(logits, targets)  (train_ppi.py:114,157), one launch forward and
    one backward (csrc/k8_loss.hip) instead of ATen's dozen.  targets: float tensor of the logits' shape, on the GPU."""

    def __init__(self, targets: torch.Tensor):
        if not targets.is_cuda:
            raise RuntimeError("pygat_amd: BCEWithLogits needs GPU tensors; there is no CPU path")
        self.target = targets.contiguous().float()
        self.ws = torch.zeros(lib.pygat_bce_workspace_bytes(self.target.numel()) // 4 + 1, dtype=torch.float32, device=targets.device)

    def __call__(self, logits: torch.Tensor) -> torch.Tensor:
        return _BCEFn.apply(logits, self)
