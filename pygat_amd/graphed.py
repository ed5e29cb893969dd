"""HIP-graph replay of the hot path: one level (`GraphedLevel`) or a whole training epoch (`FusedEpoch`).

The reference launches every ATen op of every head of every level from Python, each epoch again
(models.py:32,34; train.py:151-179).  On the reference's own graphs (Cora: 13 264 edges) the fused kernels
take microseconds and the time goes to launches; on the large synthetic graph 7 % of a step is launch gaps and
allocator calls.  A HIP graph captures the launch sequence once -- kernels, their arguments, the memory they
use, and the fork/join of the side stream (ops._side_stream) as parallel branches -- and replays it with one
call.  Tensors handed in are copied into the captured ("static") buffers only if they are not those buffers
themselves; dropout masks stay fresh on every replay because the kernels draw them from a seed in DEVICE
memory that the captured graph advances (pygat_amd.dropout, csrc/k7_dropout.hip).

    lvl = GraphedLevel(graph, x, W, a)            # captures forward and backward
    y = lvl(x, W, a); y.backward(G)               # two replays

    ep = FusedEpoch(model, optimizer, x, graph, loss_fn)     # train.py:151-179 as ONE graph
    loss_train, loss_val = ep.run()               # train step + eval forward, one replay
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from .graph import CSRGraph
from .ops import GATLevelFn


def _side_warmup(fn, iters: int):
    """Run `fn` a few times on a side stream before capture (allocator warm-up, lazy initialisations)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(iters):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()


class _Replay(torch.autograd.Function):
    """forward = replay of the captured forward graph, backward = replay of the captured backward graph."""

    @staticmethod
    def forward(ctx, lvl, *inputs):
        for dst, src in zip(lvl.inputs, inputs):
            if src is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        lvl.g_fwd.replay()
        ctx.lvl = lvl
        return lvl.out.detach()

    @staticmethod
    def backward(ctx, G):
        lvl = ctx.lvl
        if G.data_ptr() != lvl.G.data_ptr():
            lvl.G.copy_(G)
        lvl.g_bwd.replay()
        # copies: the static buffers are overwritten by the next replay, and autograd may adopt a returned tensor as
        # param.grad (then accumulate into it in place)
        if ctx.needs_input_grad[1] and lvl.grads[0] is None:
            raise RuntimeError("GraphedLevel: x requires grad but the level was captured with need_dx=False")
        return (None,) + tuple(None if g is None else g.detach().clone() for g in lvl.grads)


class GraphedLevel:
    """One GAT level, forward + backward, as two HIP graphs (the op boundary of ops.GATLevelFn, replayed).

    x [N,Fin], W [H,Fin,F'], a [H,2F'], Wskip [H,Fin,F'] | None are the SHAPES (and initial values) to capture
    with; `need_dx` selects whether the backward graph also produces the gradient into x (every level but the
    first needs it, layers.py:85-89).  The returned gradients are views of static buffers: consume them (or
    copy) before the next replay."""

    def __init__(self, graph: CSRGraph, x: torch.Tensor, W: torch.Tensor, a: torch.Tensor,
                 Wskip: Optional[torch.Tensor] = None, alpha: float = 0.2, concat: bool = True, need_dx: bool = False,
                 warmup: int = 3):
        if not x.is_cuda:
            raise RuntimeError("pygat_amd: GraphedLevel needs GPU tensors; there is no CPU path")
        self.graph, self.alpha, self.concat = graph, float(alpha), bool(concat)
        mk = lambda t, g: None if t is None else t.detach().clone().float().contiguous().requires_grad_(g)  # noqa: E731
        self.x, self.W, self.a, self.Wskip = mk(x, need_dx), mk(W, True), mk(a, True), mk(Wskip, True)
        self.inputs = [self.x, self.W, self.a] + ([self.Wskip] if Wskip is not None else [])
        diff = [t for t in self.inputs if t.requires_grad]

        def fwd():
            return GATLevelFn.apply(self.x, self.W, self.a, self.Wskip, graph, self.alpha, self.concat)

        def both():
            o = fwd()
            torch.autograd.grad(o, diff, torch.ones_like(o))

        with torch.cuda.device(x.device):
            _side_warmup(both, warmup)
            self.g_fwd, self.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fwd):
                self.out = fwd()
            self.G = torch.zeros_like(self.out)
            with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool()):
                got = torch.autograd.grad(self.out, diff, self.G)
        it = iter(got)
        self.grads = [next(it) if t.requires_grad else None for t in self.inputs]

    # -- without autograd: the two replays as plain calls (bench.py, pipelines that own their buffers)
    @torch.no_grad()
    def forward(self, x=None, W=None, a=None, Wskip=None) -> torch.Tensor:
        """Replay the forward; returns the static output buffer.  None = keep the static input as it is."""
        for dst, src in zip(self.inputs, [x, W, a] + ([Wskip] if self.Wskip is not None else [])):
            if src is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.g_fwd.replay()
        return self.out

    @torch.no_grad()
    def backward(self, G: torch.Tensor):
        """Replay the backward of the last forward for upstream gradient G; returns the static gradients
        [dx | None, dW, da(, dWskip)]."""
        if G.data_ptr() != self.G.data_ptr():
            self.G.copy_(G)
        self.g_bwd.replay()
        return self.grads

    def __call__(self, x=None, W=None, a=None, Wskip=None) -> torch.Tensor:
        args = [x, W, a] + ([Wskip] if self.Wskip is not None else [])
        # tensors that require grad must be passed for autograd to route gradients to them; None = the static buffer
        args = [s if t is None else t for t, s in zip(args, self.inputs)]
        return _Replay.apply(self, *args)


class FusedEpoch:
    """The reference's epoch (train.py:151-179: training step = forward, loss on the training subset, backward,
    optimiser step; then an eval-mode forward + loss) captured into ONE HIP graph.

    model: pygat_amd.GAT (or any module over the HIP levels); optimizer: pygat_amd.Adam (its step counter is a device
    integer the kernel advances itself) or a torch optimiser built with `capturable=True` (Adam / AdamW); `loss_fn(out) ->
    scalar` closes over labels and index sets; `eval_fn(out) -> tensor` (default: loss_fn) is evaluated on the eval-mode
    output, `evaluate=False` drops that half (train.py's --fastmode).  Dropout masks differ on every replay.
    The `warmup` epochs run before the capture are real epochs: they train the model.

    What the capture FREEZES (run() raises instead of silently replaying stale values): the input features `x` -- a first
    level on sparse features bakes x's non-zero pattern and values into the graph, so x must not be modified in place
    afterwards (x._version is checked) -- and the optimiser's hyper-parameters, which reach the kernels as launch arguments
    (a changed lr / weight_decay / betas / eps, e.g. by an LR scheduler, needs a new FusedEpoch)."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, x: torch.Tensor, graph,
                 loss_fn: Callable[[torch.Tensor], torch.Tensor], eval_fn: Optional[Callable] = None,
                 evaluate: bool = True, warmup: int = 3, capture: bool = True):
        for grp in optimizer.param_groups:
            if "capturable" in grp and not grp["capturable"]:
                raise ValueError("FusedEpoch: build the optimiser with capturable=True (its step counter must live "
                                 "on the device to be advanced by a replayed graph)")
        self.model, self.opt, self.x, self.graph = model, optimizer, x, graph
        self._hyper_now()                                 # (refuses tensor-valued hyper-parameters BEFORE anything is captured)
        self.loss_fn, self.eval_fn, self.evaluate = loss_fn, (eval_fn or loss_fn), evaluate
        self.epochs = 0
        self.g = None
        self._one = torch.ones((), dtype=torch.float32, device=x.device)   # d loss / d loss, made once (backward() would fill one per step)
        if capture:                                       # capture=False: the same epoch body, launched eagerly
            with torch.cuda.device(x.device):
                _side_warmup(self._eager_epoch, warmup)   # NOTE: these warm-up epochs DO train the model
                self.g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g):
                    self.static = self._eager_epoch()
        self.epochs = 0
        self._x_version = x._version
        self._hyper = self._hyper_now()

    def _hyper_now(self):
        """Everything of the optimiser a captured launch may have taken as an argument: EVERY non-'params' entry of every
        parameter group (lr, weight_decay, betas, eps, amsgrad, maximize, ...) plus the identity of the parameters in it.
        Tensor-valued entries (capturable torch.optim.Adam allows a tensor lr) are refused at capture time: comparing them
        per run() would cost a device sync, and their value is read by the kernels from device memory anyway, which a
        replay cannot see change."""
        snap = []
        for grp in self.opt.param_groups:
            items = []
            for k in sorted(grp):
                if k == "params":
                    continue
                v = grp[k]
                if isinstance(v, torch.Tensor):
                    raise ValueError(f"FusedEpoch: optimiser hyper-parameter {k!r} is a tensor; use a Python number (the value is "
                                     "a launch argument of the captured kernels)")
                items.append((k, tuple(v) if isinstance(v, (list, tuple)) else v))
            snap.append((tuple(items), tuple(id(p) for p in grp["params"])))
        return snap

    def _eager_epoch(self):
        self.model.train()
        self.opt.zero_grad(set_to_none=True)     # no fill kernels; the first gradient of a step is adopted, not added
        loss = self.loss_fn(self.model(self.x, self.graph))
        loss.backward(self._one if loss.dtype == torch.float32 and loss.dim() == 0 else None)
        self.opt.step()
        val = loss
        if self.evaluate:
            self.model.eval()
            with torch.no_grad():
                val = self.eval_fn(self.model(self.x, self.graph))
            self.model.train()
        self.epochs += 1
        return loss, val

    def run(self):
        """One epoch = one graph replay.  Returns (train loss, eval value) as views of static device tensors:
        read them (`.item()`, `.clone()`) before the next call if they are to be kept."""
        if self.g is None:
            return self._eager_epoch()
        if self.x._version != self._x_version:
            raise RuntimeError("FusedEpoch: x was modified in place after the capture; the captured graph holds the old "
                               "features (and, for a sparse first level, their non-zero pattern).  Build a new FusedEpoch.")
        if self._hyper_now() != self._hyper:
            raise RuntimeError("FusedEpoch: the optimiser's hyper-parameters changed after the capture; they are launch "
                               "arguments of the captured kernels and would not take effect.  Build a new FusedEpoch.")
        self.g.replay()
        self.epochs += 1
        return self.static
