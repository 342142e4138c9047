"""Host logic of the deferred (grouped) weight gradients of equiformer_amd/ops.py on CPU, with the launch replaced by a
recorder: one queue and one engine callback per graph task (nested passes keep their own), the flush writes into the tensor that
IS the parameter's .grad when the pass ends, a pass that dies inside backward takes its entries with it, torch.autograd.grad
passes are recognised and not deferred, an early flush (gradient hooks) launches what is queued so far."""
import gc

import pytest
import torch

from equiformer_amd import ops


@pytest.fixture()
def recorder(monkeypatch):
    calls = []
    monkeypatch.setattr(ops, "_lin_wgrad_descs", lambda x, dy, spec, tw, tb: [(tw.data_ptr(), tw.numel(), spec)])
    monkeypatch.setattr(ops, "_gemm_group", lambda descs, st: calls.append(list(descs)))
    monkeypatch.setattr(ops, "_stream", lambda: None)
    prev = ops.set_deferred_weight_gradients(True)
    ops.deferred_weight_gradient_stats(reset=True)
    yield calls
    ops.set_deferred_weight_gradients(prev)


def _queued():
    gc.collect()
    return sum(len(q.entries) for q in ops._task_queues.values())


class _Lin(torch.autograd.Function):
    """y = x w (CPU stand-in for _IrrepsLinear): its backward hands the weight gradient to the deferral queue when allowed"""

    @staticmethod
    def forward(ctx, x, w, tag, fail):
        ctx.save_for_backward(x, w)
        ctx.tag, ctx.fail, ctx.w = tag, fail, w
        return x @ w

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if ops._can_defer(ctx.w):
            dw = torch.zeros(w.numel())
            ops._defer_lin_wgrad(ctx.w, None, x, dy, ctx.tag, False, dw, None)
            dw = dw.view_as(w)
        else:
            dw = x.t() @ dy
        if ctx.fail:
            raise RuntimeError("backward dies here")
        return dy @ w.t(), dw, None, None


def test_alias_keeps_the_memory_not_the_tensor():
    t = torch.zeros(12)[4:10]
    a = ops._alias(t)
    u = ops._from_alias(a)
    assert u.data_ptr() == t.data_ptr() and u.numel() == 6
    u.fill_(3.0)
    assert float(t.sum()) == 18.0


def test_one_flush_per_pass_into_the_tensor_that_is_grad(recorder):
    w1, w2 = torch.randn(4, 3, requires_grad=True), torch.randn(3, 2, requires_grad=True)
    x = torch.randn(5, 4)
    _Lin.apply(_Lin.apply(x, w1, "first", False), w2, "second", False).sum().backward()
    assert len(recorder) == 1 and _queued() == 0 and len(ops._task_queues) == 0
    group = recorder[0]
    assert [d[2] for d in group] == ["second", "first"]  # backward order, ONE grouped launch
    # AccumulateGrad adopted the zero tensors: the launch is aimed at the memory of .grad itself
    assert group[0][0] == w2.grad.data_ptr() and group[1][0] == w1.grad.data_ptr()
    assert group[0][1] == w2.numel() and group[1][1] == w1.numel()
    assert ops.deferred_weight_gradient_stats() == {"queued": 2, "flushes": 1}
    # a second pass: .grad exists now -> these gradients are computed at once (no deferral), the queue stays empty
    with torch.no_grad():
        assert not ops._can_defer(w1)


def test_a_pass_that_dies_takes_its_entries_with_it(recorder):
    w1, w2 = torch.randn(4, 3, requires_grad=True), torch.randn(3, 2, requires_grad=True)
    x = torch.randn(5, 4)
    with pytest.raises(RuntimeError):
        _Lin.apply(_Lin.apply(x, w1, "dead-first", True), w2, "dead-second", False).sum().backward()
    # the queue was owned by the dead graph task's callback: nothing is left behind, nothing was launched for it later
    assert _queued() == 0
    del recorder[:]
    w1.grad = w2.grad = None
    _Lin.apply(_Lin.apply(x, w1, "first", False), w2, "second", False).sum().backward()
    assert len(recorder) == 1 and [d[2] for d in recorder[0]] == ["second", "first"]
    assert _queued() == 0


class _Reentrant(torch.autograd.Function):
    """identity whose backward runs ANOTHER backward pass (what a re-entrant checkpoint does) before returning"""

    @staticmethod
    def forward(ctx, x, inner):
        ctx.inner = inner
        return x.clone()

    @staticmethod
    def backward(ctx, dy):
        with torch.enable_grad():
            ctx.inner()
        return dy, None


def test_nested_backward_keeps_the_outer_queue(recorder):
    """a nested graph task has its own id; it must not drop the entries the outer pass has queued so far."""
    w_out2, w_out1 = torch.randn(3, 2, requires_grad=True), torch.randn(4, 3, requires_grad=True)
    w_in = torch.randn(4, 4, requires_grad=True)
    x = torch.randn(5, 4)

    def inner():
        _Lin.apply(x, w_in, "inner", False).sum().backward()

    h = _Lin.apply(x, w_out1, "outer-first", False)
    h = _Reentrant.apply(h, inner)
    _Lin.apply(h, w_out2, "outer-second", False).sum().backward()
    tags = sorted(tuple(d[2] for d in g) for g in recorder)
    assert tags == [("inner",), ("outer-second", "outer-first")]  # each pass flushed its own entries, none lost
    for w in (w_out1, w_out2, w_in):
        assert w.grad is not None
    by_tag = {d[2]: d[0] for g in recorder for d in g}
    assert by_tag["inner"] == w_in.grad.data_ptr() and by_tag["outer-first"] == w_out1.grad.data_ptr()
    assert _queued() == 0


def test_autograd_grad_passes_are_not_deferred(recorder):
    """torch.autograd.grad captures gradients (and sums several contributions of one parameter out of place): the engine
    itself says that the parameter's AccumulateGrad node will not run, and the gradient is computed at once."""
    w = torch.randn(4, 4, requires_grad=True)
    x = torch.randn(5, 4)
    y = _Lin.apply(_Lin.apply(x, w, "a", False), w, "b", False).sum()  # two contributions to one parameter
    (g,) = torch.autograd.grad(y, [w])
    assert not recorder and _queued() == 0
    ref = torch.autograd.grad(((x @ w) @ w).sum(), [w])[0]
    assert torch.allclose(g, ref, atol=1e-5)
    # .backward() restricted to other inputs: the parameter receives nothing, nothing is queued
    x2 = x.clone().requires_grad_(True)
    _Lin.apply(x2, w, "c", False).sum().backward(inputs=[x2])
    assert not recorder and w.grad is None and _queued() == 0


def test_early_flush_from_a_gradient_hook(recorder):
    """FlatGradAllReduce's tail hook: what is queued when a post-accumulate-grad hook fires is launched there, the rest of
    the pass at its end."""
    w1, w2 = torch.randn(4, 3, requires_grad=True), torch.randn(3, 2, requires_grad=True)
    w1._eqf_flushes = w2._eqf_flushes = True  # what FlatGradAllReduce sets: "my hook flushes before it reads"
    x = torch.randn(5, 4)
    seen = []

    def hook(p):
        ops.flush_deferred_weight_gradients()
        seen.append([d[2] for g in recorder for d in g])

    h = w2.register_post_accumulate_grad_hook(hook)
    _Lin.apply(_Lin.apply(x, w1, "first", False), w2, "second", False).sum().backward()
    h.remove()
    assert seen == [["second"]]
    assert [[d[2] for d in g] for g in recorder] == [["second"], ["first"]]
    assert recorder[0][0][0] == w2.grad.data_ptr() and recorder[1][0][0] == w1.grad.data_ptr()
    ops.flush_deferred_weight_gradients()  # outside a pass: no-op
    assert len(recorder) == 2


def test_switch_conditions(recorder):
    w = torch.randn(3, 3, requires_grad=True)
    with torch.no_grad():  # outside a graph task nothing will run the AccumulateGrad node: never deferred
        assert not ops._can_defer(w)
    res = {}

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w_):
            ctx.w = w_
            return x @ w_

        @staticmethod
        def backward(ctx, dy):
            res["leaf"] = ops._can_defer(ctx.w, None)
            res["nonleaf"] = ops._can_defer(ctx.w * 2)
            res["nograd"] = ops._can_defer(torch.randn(3))
            with torch.enable_grad():  # grad mode on = a create_graph backward
                res["create_graph"] = ops._can_defer(ctx.w)
            prev = ops.set_deferred_weight_gradients(False)
            res["off"] = ops._can_defer(ctx.w)
            ops.set_deferred_weight_gradients(prev)
            return None, torch.zeros_like(ctx.w)

    Probe.apply(torch.randn(2, 3), w).sum().backward()
    assert res == {"leaf": True, "nonleaf": False, "nograd": False, "create_graph": False, "off": False}


def test_a_foreign_gradient_hook_switches_deferral_off_for_its_parameter(recorder):
    """anything that reads a gradient DURING backward (tensor hooks, post-accumulate hooks, DDP's bucket hooks)
    would see the zero tensor of a deferred gradient.  A parameter with a hook that does not declare the flush is not deferred:
    the hook sees the real gradient; the other parameter still is."""
    w1, w2 = torch.randn(4, 3, requires_grad=True), torch.randn(3, 2, requires_grad=True)
    x = torch.randn(5, 4)
    seen = {}
    h1 = w2.register_post_accumulate_grad_hook(lambda p: seen.__setitem__("post", p.grad.clone()))
    h2 = w1.register_hook(lambda g: seen.__setitem__("tensor", g.clone()))
    y = _Lin.apply(_Lin.apply(x, w1, "first", False), w2, "second", False)
    y.sum().backward()
    h1.remove(), h2.remove()
    assert not recorder and _queued() == 0  # both parameters are hooked: nothing was queued
    ref1, ref2 = torch.autograd.grad(((x @ w1) @ w2).sum(), [w1, w2])
    assert torch.allclose(seen["post"], ref2, atol=1e-6) and torch.allclose(seen["tensor"], ref1, atol=1e-6)
    assert torch.allclose(w1.grad, ref1, atol=1e-6) and torch.allclose(w2.grad, ref2, atol=1e-6)


def test_early_flush_keeps_entries_whose_parameter_is_not_accumulated_yet(recorder, monkeypatch):
    """a weight used three times; a (flush-declaring) hook on a layer in between fires while the shared
    weight's .grad is still None and its first deferred zero tensor has already been summed out of place with the second
    contribution.  The early flush must leave that entry queued (a launch into the dead zero tensor would lose it); at the end of
    the pass every queued entry is aimed at the tensor that IS .grad."""
    launched = []
    monkeypatch.setattr(ops, "_gemm_group", lambda descs, st: launched.append(list(descs)))
    w = torch.randn(4, 4, requires_grad=True)
    v = torch.randn(4, 4, requires_grad=True)
    w._eqf_flushes = v._eqf_flushes = True
    x = torch.randn(5, 4)
    h = v.register_post_accumulate_grad_hook(lambda p: ops.flush_deferred_weight_gradients())
    # v sits between the uses of w: its AccumulateGrad (and the hook) runs while w still waits for its remaining contributions
    y = _Lin.apply(_Lin.apply(_Lin.apply(_Lin.apply(x, w, "w-1", False), w, "w-2", False), v, "v", False), w, "w-3", False)
    y.sum().backward()
    h.remove()
    tags = [[d[2] for d in g] for g in launched]
    # the hook's early flush launched v only (w.grad was still None); everything of w went out when the pass ended
    assert tags[0] == ["v"], tags
    assert sorted(t for g in tags[1:] for t in g) == ["w-1", "w-2", "w-3"], tags
    assert all(d[0] == w.grad.data_ptr() for g in launched[1:] for d in g)  # ... into the tensor that IS .grad, not a dead zero tensor


def test_an_initialised_process_group_without_the_reducer_switches_deferral_off(recorder):
    """Stock DistributedDataParallel hangs its bucket hooks on the AccumulateGrad nodes (not visible from Python): with a process
    group up and parameters that no FlatGradAllReduce has marked, nothing is deferred and DDP's gradients are exact."""
    import socket
    import torch.distributed as dist
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w1 = torch.nn.Parameter(torch.randn(4, 3))
                self.w2 = torch.nn.Parameter(torch.randn(3, 2))

            def forward(self, x):
                return _Lin.apply(_Lin.apply(x, self.w1, "first", False), self.w2, "second", False)
        net = Net()
        ddp = torch.nn.parallel.DistributedDataParallel(net)
        x = torch.randn(5, 4)
        ddp(x).sum().backward()
        assert not recorder and _queued() == 0
        ref1, ref2 = torch.autograd.grad(((x @ net.w1) @ net.w2).sum(), [net.w1, net.w2])
        assert torch.allclose(net.w1.grad, ref1, atol=1e-6) and torch.allclose(net.w2.grad, ref2, atol=1e-6)
        # ... and FlatGradAllReduce's parameters keep it (the reducer's own hook flushes first)
        from equiformer_amd.parallel import FlatGradAllReduce
        net2 = Net()
        FlatGradAllReduce(net2)
        with torch.no_grad():
            assert not ops._hooked(net2.w1) and ops._hooked(torch.nn.Parameter(torch.randn(2)))
    finally:
        dist.destroy_process_group()
