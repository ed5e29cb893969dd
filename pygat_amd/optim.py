"""torch.optim.Adam's update as ONE HIP launch over all parameters (csrc/k11_adam.hip, pygat_adam_step).

The reference trains with `optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)` and one
`optimizer.step()` per epoch (train.py:64-66,122; train_ppi.py:58-60).  `pygat_amd.Adam` takes the same arguments (betas,
eps, weight_decay as L2 added to the gradient -- not AdamW) and keeps torch's state names (`exp_avg`, `exp_avg_sq`), but its
step counter is a device integer the kernel advances itself: the step is capturable into a HIP graph as it is
(pygat_amd.FusedEpoch), and a replay costs one launch instead of torch's two (its fused, capturable Adam: counters 3 us +
update 12 us on the citation models).  fp32 parameters on one device; `maximize` / `amsgrad` are not offered."""
import ctypes as C
from typing import Iterable

import torch

from ._lib import lib, check, MAX_ADAM_TENSORS


class Adam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if not (lr >= 0 and 0 <= betas[0] < 1 and 0 <= betas[1] < 1 and eps >= 0 and weight_decay >= 0):
            raise ValueError("pygat_amd.Adam: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._chunks = {}       # group index -> list of (params, state tensor) chunks of <= MAX_ADAM_TENSORS

    def _chunks_of(self, gi, group):
        ps = [p for p in group["params"] if p.requires_grad]
        key = tuple(id(p) for p in ps)
        hit = self._chunks.get(gi)
        if hit is None or hit[0] != key:
            for p in ps:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise TypeError("pygat_amd.Adam: parameters must be contiguous float32 tensors on the GPU")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            chunks = [(ps[i:i + MAX_ADAM_TENSORS], torch.zeros(6, dtype=torch.int32, device=ps[0].device))     # PYGAT_ADAM_STATE_BYTES
                      for i in range(0, len(ps), MAX_ADAM_TENSORS)] if ps else []
            self._chunks[gi] = hit = (key, chunks)
        return hit[1]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            for ps, state in self._chunks_of(gi, group):
                live = [p for p in ps if p.grad is not None]
                if len(live) != len(ps):
                    # torch skips parameters without a gradient; a chunk's step counter is shared, so all or none
                    if not live:
                        continue
                    raise RuntimeError("pygat_amd.Adam: some parameters of a group have no gradient")
                n = len(ps)
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
                for g_, p in zip(grads, ps):
                    if g_.dtype != torch.float32 or g_.shape != p.shape:
                        raise TypeError("pygat_amd.Adam: gradients must be float32 of the parameter's shape")
                arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])     # noqa: E731
                numel = (C.c_int64 * n)(*[p.numel() for p in ps])
                st = torch.cuda.current_stream(ps[0].device).cuda_stream
                check(lib.pygat_adam_step(n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]),
                                          arr([self.state[p]["exp_avg_sq"] for p in ps]), numel, float(group["lr"]), float(b1),
                                          float(b2), float(group["eps"]), float(group["weight_decay"]), state.data_ptr(), st),
                      "adam_step")
        return loss

    def steps_taken(self) -> int:
        """The device step counter of the first chunk (a synchronising read)."""
        for _, chunks in self._chunks.values():
            for _, state in chunks:
                return int(state[0].item())
        return 0
