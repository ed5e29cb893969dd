"""pygat_amd.Adam (csrc/k11_adam.hip: torch.optim.Adam's update in one launch, device step counter) against torch.optim.Adam
on the same parameters and gradients (train.py:64-66: lr 0.005, weight_decay 5e-4), eagerly and replayed from a HIP graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(8, 1433, 8), (8, 16), (64, 7), (14,), (5000,), (1,), (4097,)]


def _params(seed):
    gen = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=gen).cuda().requires_grad_(True) for s in SHAPES]


def _grads(step):
    gen = torch.Generator().manual_seed(100 + step)
    return [torch.randn(*s, generator=gen).cuda() * (0.1 + step) for s in SHAPES]


@pytest.mark.parametrize("wd", [0.0, 5e-4])
def test_adam_matches_torch(wd):
    import pygat_amd as pg
    ours, theirs = _params(1), _params(1)
    a = pg.Adam(ours, lr=5e-3, weight_decay=wd)
    b = torch.optim.Adam(theirs, lr=5e-3, weight_decay=wd)
    for step in range(6):
        for p, q, g in zip(ours, theirs, _grads(step)):
            p.grad = g.clone(); q.grad = g.clone()
        a.step(); b.step()
        for k, (p, q) in enumerate(zip(ours, theirs)):
            # a step moves a parameter by <= lr; the two differ in rounding of the bias corrections and of m / denom
            assert torch.allclose(p, q, rtol=0, atol=2e-6 * 5e-3 * (step + 1) + 1e-7 * float(q.abs().max())), \
                f"step {step} tensor {k}: {float((p - q).abs().max()):.3e}"
            assert torch.allclose(a.state[p]["exp_avg"], b.state[q]["exp_avg"], rtol=1e-6, atol=1e-7)
            assert torch.allclose(a.state[p]["exp_avg_sq"], b.state[q]["exp_avg_sq"], rtol=1e-6, atol=1e-9)
    assert a.steps_taken() == 6


def test_adam_many_tensors_and_graph_replay():
    """More tensors than one launch's table (two chunks, each with its own counter), then the step captured once and
    replayed: the device counter advances, so every replay is the next step."""
    import pygat_amd as pg
    gen = torch.Generator().manual_seed(5)
    ours = [torch.randn(33, generator=gen).cuda().requires_grad_(True) for _ in range(60)]
    theirs = [p.detach().clone().requires_grad_(True) for p in ours]
    a = pg.Adam(ours, lr=1e-2, weight_decay=1e-3)
    b = torch.optim.Adam(theirs, lr=1e-2, weight_decay=1e-3)
    static = [torch.zeros(33, device="cuda") for _ in ours]
    for p, g in zip(ours, static):
        p.grad = g                                  # the gradients live in static buffers, as under FusedEpoch

    def feed():
        for sg, q in zip(static, theirs):
            g = torch.randn(33, generator=gen).cuda()
            sg.copy_(g); q.grad = g.clone()

    def same(what):
        for k, (p, q) in enumerate(zip(ours, theirs)):
            assert torch.allclose(p, q, rtol=0, atol=1e-6), f"{what} tensor {k}: {float((p - q).abs().max()):.3e}"

    for step in range(2):                           # eager, both chunks
        feed(); a.step(); b.step(); same(f"eager step {step}")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                   # (capture executes nothing)
        a.step()
    for step in range(3):
        feed(); graph.replay(); b.step(); same(f"replayed step {step}")
    assert a.steps_taken() == 5


def test_adam_unaligned_and_tiny_tensors():
    """Parameters that are views at odd offsets (no 16-byte alignment: the scalar path) and of 1-3 elements."""
    import pygat_amd as pg
    gen = torch.Generator().manual_seed(9)
    base_a = torch.randn(10000, generator=gen).cuda()
    base_b = base_a.clone()
    cuts = [(1, 4098), (4099, 4102), (4103, 4104), (4105, 9999)]
    ours = [base_a[i:j].detach().requires_grad_(True) for i, j in cuts]
    theirs = [base_b[i:j].detach().requires_grad_(True) for i, j in cuts]
    assert any(p.data_ptr() % 16 for p in ours)
    a = pg.Adam(ours, lr=3e-3, weight_decay=1e-2)
    b = torch.optim.Adam(theirs, lr=3e-3, weight_decay=1e-2)
    for step in range(3):
        for p, q in zip(ours, theirs):
            g = torch.randn(p.shape, generator=gen).cuda()
            p.grad = g.clone(); q.grad = g.clone()
        a.step(); b.step()
    assert torch.allclose(base_a, base_b, rtol=0, atol=1e-6)          # the views write through; untouched elements equal


def test_adam_state_round_trip_keeps_the_step_counter():
    """save after k steps, load into a FRESH optimiser, take step k + 1: equal to torch.optim.Adam's step k + 1 (the device
    step counter and the running beta^t products travel with state_dict(); ADVICE round 3: they used to be dropped, so a
    resumed run restarted its bias corrections on warm moments).  copy.deepcopy keeps them too."""
    import copy
    import pygat_amd as pg
    ours, theirs = _params(3), _params(3)
    a = pg.Adam(ours, lr=5e-3, weight_decay=5e-4)
    b = torch.optim.Adam(theirs, lr=5e-3, weight_decay=5e-4)
    for step in range(4):
        for p, q, g in zip(ours, theirs, _grads(step)):
            p.grad = g.clone(); q.grad = g.clone()
        a.step(); b.step()
    sd = a.state_dict()
    assert "pygat_adam_steps" in sd and sd["pygat_adam_steps"][0][0][0] == 4
    fresh_params = [p.detach().clone().requires_grad_(True) for p in ours]
    fresh = pg.Adam(fresh_params, lr=5e-3, weight_decay=5e-4)
    fresh.load_state_dict(copy.deepcopy(sd))
    dup = copy.deepcopy(a)                               # the optimiser object itself (its parameters are copied along)
    dup_params = dup.param_groups[0]["params"]
    for p, q, r, g in zip(fresh_params, theirs, dup_params, _grads(4)):
        p.grad = g.clone(); q.grad = g.clone(); r.grad = g.clone()
    fresh.step(); b.step(); dup.step()
    assert fresh.steps_taken() == 5 and dup.steps_taken() == 5
    for k, (p, q, r) in enumerate(zip(fresh_params, theirs, dup_params)):
        tol = 2e-6 * 5e-3 * 5 + 1e-7 * float(q.abs().max())
        assert torch.allclose(p, q, rtol=0, atol=tol), f"resumed tensor {k}: {float((p - q).abs().max()):.3e}"
        assert torch.allclose(r, q, rtol=0, atol=tol), f"deep-copied tensor {k}: {float((r - q).abs().max()):.3e}"
    # a checkpoint with moments but no step words (torch.optim.Adam's) is refused, not silently mis-stepped
    with pytest.raises(ValueError):
        pg.Adam([p.detach().clone().requires_grad_(True) for p in ours], lr=5e-3).load_state_dict(
            {k: v for k, v in sd.items() if k != "pygat_adam_steps"})
