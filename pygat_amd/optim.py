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
        # group index -> (parameter list, [(params, state tensor) chunks of <= MAX_ADAM_TENSORS]).  The 24-byte device state of a
        # chunk (step counter + running beta^t products) travels with state_dict() / load_state_dict() (below) and with
        # copy.deepcopy (the chunk is found again by the IDENTITY of its parameters, which a deep copy preserves pairwise).
        self._chunks = {}
        self._loaded_steps = None      # chunk states of a load_state_dict, applied when the chunks are next built

    def _chunks_of(self, gi, group):
        ps = [p for p in group["params"] if p.requires_grad]
        hit = self._chunks.get(gi)
        if hit is None or len(hit[0]) != len(ps) or any(a is not b for a, b in zip(hit[0], ps)):
            for p in ps:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise TypeError("pygat_amd.Adam: parameters must be contiguous float32 tensors on the GPU")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            chunks = [(ps[i:i + MAX_ADAM_TENSORS], torch.zeros(6, dtype=torch.int32, device=ps[0].device))     # PYGAT_ADAM_STATE_BYTES
                      for i in range(0, len(ps), MAX_ADAM_TENSORS)] if ps else []
            loaded = (self._loaded_steps or {}).pop(gi, None)
            if loaded is not None:
                if len(loaded) != len(chunks):
                    raise ValueError("pygat_amd.Adam: the loaded step state does not match this group's parameters")
                for (_, st), words in zip(chunks, loaded):
                    st.copy_(torch.tensor(words, dtype=torch.int32))
            self._chunks[gi] = hit = (ps, chunks)
        return hit[1]

    def _step_words(self):
        return {gi: [st.cpu().tolist() for _, st in hit[1]] for gi, hit in self._chunks.items()}

    def __getstate__(self):          # pickling / copy.deepcopy: torch keeps defaults, state and param_groups only
        st = super().__getstate__()
        st["_pygat_steps"] = self._step_words()
        return st

    def __setstate__(self, state):
        steps = state.pop("_pygat_steps", None)
        super().__setstate__(state)
        self._chunks = {}
        self._loaded_steps = steps or None

    def state_dict(self):
        """torch's state_dict plus `pygat_adam_steps`: per group, per chunk, the six 32-bit words of the device step state
        (step counter, running beta1^t and beta2^t as doubles).  Without them a resumed run would restart the bias corrections
        on warm moments and take mis-scaled steps."""
        sd = super().state_dict()
        sd["pygat_adam_steps"] = {gi: [st.cpu().tolist() for _, st in self._chunks_of(gi, g)]
                                  for gi, g in enumerate(self.param_groups)}
        return sd

    def load_state_dict(self, state_dict):
        steps = state_dict.get("pygat_adam_steps")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "pygat_adam_steps"})
        self._chunks = {}                                  # rebuilt (moments from self.state) on the next step
        self._loaded_steps = {int(k): v for k, v in steps.items()} if steps is not None else None
        if steps is None and any("exp_avg" in st for st in self.state.values()):
            raise ValueError("pygat_amd.Adam.load_state_dict: moments without `pygat_adam_steps` (a torch.optim.Adam "
                             "checkpoint?): the step counter is needed for the bias corrections")

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
        for gi, g in enumerate(self.param_groups):
            self._chunks_of(gi, g)
        for _, chunks in self._chunks.values():
            for _, state in chunks:
                return int(state[0].item())
        return 0
