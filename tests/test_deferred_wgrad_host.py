"""Host logic of the deferred (grouped) weight gradients of equiformer_amd/ops.py on CPU, with the launch replaced by a
recorder: one engine callback per backward pass, the flush writes into the tensor that IS the parameter's .grad when the pass
ends, entries of a pass that died inside backward are dropped by the next pass, and the switch conditions (`_can_defer`)."""
import pytest
import torch

from equiformer_amd import ops


@pytest.fixture()
def recorder(monkeypatch):
    calls = []
    monkeypatch.setattr(ops, "_lin_wgrad_descs", lambda x, dy, spec, tw, tb: [(tw.data_ptr(), tw.numel(), spec)])
    monkeypatch.setattr(ops, "_gemm_group", lambda descs, st: calls.append(list(descs)))
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "_seen_create_graph", [False])
    prev = ops.set_deferred_weight_gradients(True)
    del ops._deferred[:]
    ops._deferred_task[0] = -1
    yield calls
    ops.set_deferred_weight_gradients(prev)
    del ops._deferred[:]
    ops._deferred_task[0] = -1


class _Lin(torch.autograd.Function):
    """y = x w (CPU stand-in for _IrrepsLinear): its backward hands the weight gradient to the deferral queue"""

    @staticmethod
    def forward(ctx, x, w, tag, fail):
        ctx.save_for_backward(x, w)
        ctx.tag, ctx.fail, ctx.w = tag, fail, w
        return x @ w

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        assert ops._can_defer(ctx.w)
        dw = torch.zeros(w.numel())
        ops._defer_lin_wgrad(ctx.w, None, x, dy, ctx.tag, False, dw, None)
        if ctx.fail:
            raise RuntimeError("backward dies here")
        return dy @ w.t(), dw.view_as(w), None, None


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
    assert len(recorder) == 1 and not ops._deferred and ops._deferred_task[0] == -1
    group = recorder[0]
    assert [d[2] for d in group] == ["second", "first"]  # backward order, ONE grouped launch
    # AccumulateGrad adopted the zero tensors: the launch is aimed at the memory of .grad itself
    assert group[0][0] == w2.grad.data_ptr() and group[1][0] == w1.grad.data_ptr()
    assert group[0][1] == w2.numel() and group[1][1] == w1.numel()
    # a second pass: .grad exists now -> these gradients are computed at once (no deferral), the queue stays empty
    assert not ops._can_defer(w1)


def test_entries_of_a_pass_that_died_are_dropped(recorder):
    w1, w2 = torch.randn(4, 3, requires_grad=True), torch.randn(3, 2, requires_grad=True)
    x = torch.randn(5, 4)
    with pytest.raises(RuntimeError):
        _Lin.apply(_Lin.apply(x, w1, "dead-first", True), w2, "dead-second", False).sum().backward()
    del recorder[:]  # (whether the engine ran the dead pass's callback does not matter)
    w1.grad = w2.grad = None
    _Lin.apply(_Lin.apply(x, w1, "first", False), w2, "second", False).sum().backward()
    assert len(recorder) == 1 and [d[2] for d in recorder[0]] == ["second", "first"]
    assert not ops._deferred


def test_switch_conditions(recorder):
    w = torch.randn(3, 3, requires_grad=True)
    with torch.no_grad():  # backward of a first-order pass runs with grad mode off
        assert ops._can_defer(w, None)
        assert not ops._can_defer(w * 2)  # not a leaf
        assert not ops._can_defer(torch.randn(3))  # does not require grad
        ops.note_create_graph()  # a create_graph pass was seen: off for good in this process
        assert not ops._can_defer(w)
    ops._seen_create_graph[0] = False
    assert not ops._can_defer(w)  # grad mode on = a create_graph backward
    with torch.no_grad():
        ops.set_deferred_weight_gradients(False)
        assert not ops._can_defer(w)
