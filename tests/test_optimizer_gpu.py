"""The step tail against torch.optim.Adam's own bookkeeping (scripts/adapt_codeps.py:289-291, misc/train_utils.py:194-201):
parameters no active loss reaches have grad None in the reference -- Adam leaves their values, moments and step counts alone --
and every parameter carries its own step count for the bias correction."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _steps(g, pi):
    return g.steps if isinstance(g.steps, int) else g.steps[pi]


def _runner():
    import bench
    from codeps_amd.step import StepRunner
    dev = torch.device(DEV)
    net, cfg = bench.build(dev, 18, 64, 128, "adapt")
    return net, StepRunner(net, cfg, "adapt", 1), dev


def test_full_step_reaches_every_trainable_parameter():
    """The autograd-graph walk sees the conv / BN weights whose gradients the kernels write straight into the flat buffer."""
    from codeps_amd import synthetic
    net, runner, dev = _runner()
    data, _ = synthetic.adapt_batch(64, 128, dev, 5, 1, 1, 1, 2)
    runner.forward_backward(data)
    assert runner._touched is None
    for g in runner.groups:                                 # ... and none of those gradients is identically zero
        for p, a, b in zip(g.params, g.offsets[:-1], g.offsets[1:]):
            assert float(g.grad[a:b].abs().max()) > 0, tuple(p.shape)


def test_unreached_parameters_are_left_alone_like_grad_none():
    from codeps_amd import synthetic
    net, runner, dev = _runner()
    full, _ = synthetic.adapt_batch(64, 128, dev, 5, 1, 1, 1, 2)
    depth_only, _ = synthetic.adapt_batch(64, 128, dev, 6, 1, 1, 0, 0)        # no labelled item: the semantic losses are absent
    # torch.optim.Adam over clones of the parameters, fed with the gradients of the product's own backward pass
    ref_params = [[torch.nn.Parameter(g.flat[a:b].clone()) for a, b in zip(g.offsets[:-1], g.offsets[1:])] for g in runner.groups]
    opt = torch.optim.Adam([q for ps in ref_params for q in ps], lr=runner.lr)
    ema_ref = [None if g.ema is None else g.ema.clone() for g in runner.groups]
    start = [g.flat.clone() for g in runner.groups]
    seen_partial = False
    for step, data in enumerate((depth_only, depth_only, full, depth_only)):
        runner.forward_backward(data)
        touched = runner._touched
        opt.zero_grad(set_to_none=True)
        for gi, g in enumerate(runner.groups):
            for pi, (q, a, b) in enumerate(zip(ref_params[gi], g.offsets[:-1], g.offsets[1:])):
                if touched is None or touched[gi][pi]:
                    q.grad = g.grad[a:b].clone()
        opt.step()
        for gi, g in enumerate(runner.groups):              # online_adap.py:279-289 over every parameter of an EMA'd module
            if ema_ref[gi] is not None:
                ema_ref[gi].mul_(runner.ema_alpha).add_(torch.cat([q.detach() for q in ref_params[gi]]), alpha=1 - runner.ema_alpha)
        runner.optimizer_step()
        if touched is not None:
            seen_partial = True
            sem = {id(p) for p in net.semantic_head.parameters()}
            for gi, g in enumerate(runner.groups):
                for pi, p in enumerate(g.params):
                    assert touched[gi][pi] == (id(p) not in sem), "exactly the semantic head is out of reach of the depth losses"
        for gi, g in enumerate(runner.groups):
            want = torch.cat([q.detach() for q in ref_params[gi]])
            assert torch.allclose(g.flat, want, rtol=0, atol=2e-7), (step, gi, float((g.flat - want).abs().max()))
            if ema_ref[gi] is not None:
                assert torch.allclose(g.ema, ema_ref[gi], rtol=0, atol=2e-7), (step, gi)
        if step == 1:                                       # two depth-only steps: the semantic head has not moved, moments are zero
            for gi, g in enumerate(runner.groups):
                for pi, (p, a, b) in enumerate(zip(g.params, g.offsets[:-1], g.offsets[1:])):
                    if not touched[gi][pi]:
                        assert torch.equal(g.flat[a:b], start[gi][a:b]) and float(g.m[a:b].abs().max()) == 0 and float(g.v[a:b].abs().max()) == 0
                        assert _steps(g, pi) == 0
                    else:
                        assert _steps(g, pi) == 2
    assert seen_partial
    counts = {_steps(g, pi) for g in runner.groups for pi in range(len(g.params))}
    assert counts == {1, 4}                                 # per-parameter step counts: 1 for the semantic head, 4 for the rest
