"""Autograd operators of the Equiformer hot path; every forward and backward is one or more launches of
libequiformer_hip.so through its C ABI (include/equiformer_hip.h).  No operator here has a CPU / eager fallback:
tensors must live on the GPU, otherwise `HipOnlyError` is raised.

The operators are the closed primitive set of SURVEY.md Appendix B; `equiformer_amd.nets` composes them into the
reference's module tree.
"""
import contextlib
import ctypes
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib
from .lib import EqfGemmDesc, EqfRows, call


class HipOnlyError(RuntimeError):
    pass


_dummy = {}


class _Flag:
    on = False


# PROCESS-global on purpose, not thread-local: autograd executes the backward of GPU operators on its per-device worker
# thread, not on the thread that called autograd.grad(), so a threading.local set by the caller would be invisible to the
# very backward it is meant for.  The assumption is the reference's: one Python thread drives one model per process
# (one process per GPU); two threads running force passes and training backwards concurrently are not supported.
_input_grads_only = _Flag()


class input_grads_only:
    """Context of a `torch.autograd.grad(energy, pos, create_graph=True)` force pass
    [ref: nets/graph_attention_transformer_md17.py:318-325]: only gradients wrt operator INPUTS are wanted, so the
    differentiable (create_graph) backward of every operator skips its parameter gradients -- `needs_input_grad` cannot
    tell, it is static.  Outside this context the create_graph backward also returns the parameter gradients
    (first-order kernels; differentiating THROUGH those raises)."""

    def __enter__(self):
        self.prev = _input_grads_only.on
        _input_grads_only.on = True

    def __exit__(self, *exc):
        _input_grads_only.on = self.prev
        return False


def _want_param_grads():
    return not _input_grads_only.on


# ------------------------------------------------------------------------------------------------- matrix-step arithmetic
# How the matrix steps of the fused SeparableFCTP kernels multiply (csrc/sfcx.hip; C ABI eqf_sfcx_*):
#   "fp32"    exact-fp32 MFMA (csrc/sfc.hip): bit-equal to an fmaf chain; 1/16 of the bf16 matrix rate, shares the VALU lanes
#   "split"   fp32 operands split into bf16 planes (activations 2, weights 3), 5 products on the bf16 matrix cores,
#             fp32 accumulation: fp32-class results (energies 2e-6, gradients 1e-5 vs fp64; tools/split_model_error.py)
#   "bf16"    plain bf16 operands, fp32 accumulation = torch.autocast(bfloat16), which the reference's drivers switch on by
#             default (main_qm9.py:117-119,197-201); BASELINE config #2.  Tolerance vs fp32: see tests/test_gpu_sfcx.py
#   "split6"  3 + 3 planes, 6 products (3e-7): cross-checks
_MATRIX_MODES = {"fp32": None, "split": 0, "bf16": 1, "split6": 2}
_matrix_mode = ["split"]


def set_matrix_mode(name):
    """Process-wide arithmetic of the matrix steps; returns the previous mode."""
    if name not in _MATRIX_MODES:
        raise ValueError("matrix mode must be one of %s, got %r" % (sorted(_MATRIX_MODES), name))
    prev = _matrix_mode[0]
    _matrix_mode[0] = name
    return prev


def get_matrix_mode():
    return _matrix_mode[0]


class matrix_mode:
    """with ops.matrix_mode("bf16"): ...  -- the autocast-equivalent scope.  Every operator records the mode of its FORWARD
    and its gradients (first and second order) run in that recorded mode whatever the global is when backward executes, as
    autocast's backward does; the packed weight planes saved by the forward are those of the recorded mode."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = set_matrix_mode(self.name)

    def __exit__(self, *exc):
        set_matrix_mode(self.prev)
        return False


def _guard_opt(t, dep, what):
    """First-order parameter gradient handed out by a create_graph backward: usable as a value, raises when something
    tries to differentiate through it (second derivatives wrt parameters-of-parameters are not implemented)."""
    if t is None:
        return None
    return _Guard.apply(t, dep, what) if dep.requires_grad else t



def _nonnull(t):
    """Device address of a tensor; an EMPTY tensor (a batch without any edge) has data_ptr() == 0, which the C ABI would
    reject as a missing argument before it looks at the (zero) row count: hand it a valid one-element buffer instead."""
    a = t.data_ptr()
    if a == 0 and t.is_cuda:
        d = _dummy.get(t.device)
        if d is None:
            d = _dummy[t.device] = torch.zeros(64, dtype=torch.float32, device=t.device)
        return d.data_ptr()
    return a


def _p(t, off=0):
    if t is None:
        return None
    return ctypes.c_void_p(_nonnull(t) + 4 * off)


def _stream():
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipOnlyError("the Equiformer hot path runs on MI355X only (got a %s tensor); there is no CPU fallback"
                               % t.device)
        if t.dtype not in (torch.float32, torch.int32):
            raise HipOnlyError("kernels take fp32 data / int32 indices, got %s" % t.dtype)
        if not t.is_contiguous():
            raise HipOnlyError("non-contiguous tensor handed to a HIP kernel")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def rows(d, ld, inner):
    return EqfRows(int(d), int(ld), int(inner))


try:  # the raw handle of the current stream without building a torch.cuda.Stream object (9 us -> 0.3 us per launch; the
    # host, not the GPU, bounds the small-graph steps: tools/host_profile.py)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    _raw_stream = None



_capturing = getattr(torch._C, "_cuda_isCurrentStreamCapturing", lambda: False)


class _ZeroArena:
    """Zero-initialised fp32 accumulators (the targets of atomically accumulated weight / bias gradients) are carved
    out of slabs that are filled ONCE, instead of one fill launch per tensor (209 fill launches per QM9 step before).
    A slab is ordinary caching-allocator memory; it stays alive through the views handed out and no byte of it is
    handed out twice, so the semantics are exactly those of torch.zeros."""
    SLAB = 4 << 20      # floats (16 MB)
    MAX_REQ = 1 << 20   # larger requests get their own fill

    def __init__(self):
        self.slabs = {}
        self.capture_slabs = None  # a dict while a capture that declared itself is running (capture_scope)

    def take(self, numel, device):
        cap = _capturing()
        if numel == 0 or numel > self.MAX_REQ or device.type != "cuda" or (cap and self.capture_slabs is None):
            # (under a HIP-graph capture that did not open a capture_scope every accumulator gets its own fill NODE: a replay
            # must zero it again, and a slab that was filled before the capture would be accumulated into once per replay)
            return torch.zeros(numel, device=device, dtype=torch.float32)
        # inside a capture_scope the slabs are allocated DURING the capture (their fill is a node of the graph, replayed with
        # it) and dropped when the scope ends: ~4 fill nodes per QM9 step instead of 116 (5 % of the replayed step, round 6)
        slabs = self.capture_slabs if cap else self.slabs
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)
        slab, used = slabs.get(key, (None, 0))
        need = (numel + 63) & ~63
        if slab is None or used + need > self.SLAB:
            slab, used = torch.zeros(self.SLAB, device=device, dtype=torch.float32), 0
        slabs[key] = (slab, used + need)
        return slab[used:used + numel]

    @contextlib.contextmanager
    def capture_scope(self):
        """Around a stream capture (equiformer_amd/capture.py): accumulators of the captured launches share slabs of their own."""
        prev = self.capture_slabs
        self.capture_slabs = {}
        try:
            yield
        finally:
            self.capture_slabs = prev


_arena = _ZeroArena()


def _zeros(shape, device=None, dtype=torch.float32):
    if dtype != torch.float32:
        return torch.zeros(shape, device=device, dtype=dtype)
    if isinstance(shape, int):
        shape = (shape,)
    n = 1
    for d in shape:
        n *= int(d)
    return _arena.take(n, torch.device(device)).view(tuple(shape))


def _zeros_like(t):
    return _zeros(tuple(t.shape), t.device, t.dtype)


def _zeros2(n1, n2, device):
    """Two zero-initialised fp32 accumulators."""
    return _zeros(n1, device), _zeros(n2, device)


# ------------------------------------------------------------------------------------------------- layer norm
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layout, eps):
        x = _c(x)
        _chk(x, weight, bias)
        n = x.shape[0]
        y = torch.empty_like(x)
        rstd = torch.empty((n, len(layout.segs)), device=x.device, dtype=torch.float32)
        mean0 = torch.empty((n,), device=x.device, dtype=torch.float32)
        call("eqf_layernorm_fwd", _p(x), _p(weight), _p(bias), _p(y), _p(rstd), _p(mean0), n, layout.c_ref,
             float(eps), _stream())
        ctx.save_for_backward(x, weight, rstd, mean0, bias)
        ctx.layout = layout
        ctx.nb = bias.numel()
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, rstd, mean0, bias = ctx.saved_tensors
        if torch.is_grad_enabled():  # create_graph: the backward is itself a differentiable HIP operator
            dx, dw, db = _LayerNormBwd.apply(x, weight, dy, rstd, mean0, ctx.layout, ctx.eps, ctx.nb)
            return dx, _guard_opt(dw, dy, "layer-norm weight gradient"), _guard_opt(db, dy, "layer-norm bias gradient"), \
                None, None
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        dw, db = _zeros2(weight.numel(), ctx.nb, x.device)
        call("eqf_layernorm_bwd", _p(x), _p(weight), _p(dy), _p(rstd), _p(mean0), _p(dx), _p(dw), _p(db), x.shape[0],
             ctx.layout.c_ref, _stream())
        return dx, dw, db, None, None


class _LayerNormBwd(Function):
    """dx (and, outside a force pass, the first-order dw / db) of the equivariant layer norm as a differentiable op;
    its backward is eqf_layernorm_bwd2."""

    @staticmethod
    def forward(ctx, x, weight, dy, rstd, mean0, layout, eps, nb):
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        want = _want_param_grads()
        dw, db = _zeros2(weight.numel(), nb, x.device) if want else (None, None)
        call("eqf_layernorm_bwd", _p(x), _p(weight), _p(dy), _p(rstd), _p(mean0), _p(dx), _p(dw), _p(db), x.shape[0],
             layout.c_ref, _stream())
        ctx.save_for_backward(x, weight, dy)
        ctx.layout, ctx.eps = layout, eps
        if not want:
            return dx, None, None
        ctx.mark_non_differentiable(dw, db)
        return dx, dw, db

    @staticmethod
    @once_differentiable
    def backward(ctx, c, _cw=None, _cb=None):
        x, weight, dy = ctx.saved_tensors
        c = _c(c)
        _chk(c)
        g_x, g_dy = torch.empty_like(x), torch.empty_like(x)
        g_w = _zeros_like(weight)
        call("eqf_layernorm_bwd2", _p(x), _p(weight), _p(dy), _p(c), _p(g_x), _p(g_w), _p(g_dy), x.shape[0],
             ctx.layout.c_ref, float(ctx.eps), _stream())
        return g_x, g_w, g_dy, None, None, None, None, None


def layer_norm(x, weight, bias, layout, eps=1e-5):
    return _LayerNorm.apply(x, weight, bias, layout, eps)


class _AddLayerNorm(Function):
    """(s, y) = (a + b, LayerNorm(a + b)): the residual add in front of a norm and the norm in one launch; the backward
    folds the gradient arriving at s from the residual branch into the norm's input gradient (no element-wise add
    launches in either direction) and accumulates the affine-parameter gradients in the same kernel."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, layout, eps):
        a, b = _c(a), _c(b)
        _chk(a, b, weight, bias)
        n = a.shape[0]
        s = torch.empty_like(a)
        y = torch.empty_like(a)
        rstd = torch.empty((n, len(layout.segs)), device=a.device, dtype=torch.float32)
        mean0 = torch.empty((n,), device=a.device, dtype=torch.float32)
        call("eqf_add_layernorm_fwd", _p(a), _p(b), _p(s), _p(weight), _p(bias), _p(y), _p(rstd), _p(mean0), n,
             layout.c_ref, float(eps), _stream())
        ctx.save_for_backward(s, weight, rstd, mean0)
        ctx.layout, ctx.nb, ctx.eps = layout, bias.numel(), eps
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        s, weight, rstd, mean0 = ctx.saved_tensors
        if dy is None:  # the normalised branch is unused: identity on the sum
            return ds, ds, None, None, None, None
        if torch.is_grad_enabled():  # create_graph: differentiable norm backward + a differentiable add
            dx, dw, db = _LayerNormBwd.apply(s, weight, dy, rstd, mean0, ctx.layout, ctx.eps, ctx.nb)
            d = dx if ds is None else dx + ds
            return d, d, _guard_opt(dw, dy, "layer-norm weight gradient"), _guard_opt(db, dy, "layer-norm bias gradient"), \
                None, None
        dy = _c(dy)
        ds = _c(ds) if ds is not None else None
        _chk(dy, ds)
        d = torch.empty_like(s)
        want = _want_param_grads()
        dw, db = _zeros2(weight.numel(), ctx.nb, s.device) if want else (None, None)
        call("eqf_add_layernorm_bwd", _p(s), _p(weight), _p(dy), _p(ds), _p(rstd), _p(mean0), _p(d), _p(dw), _p(db), s.shape[0],
             ctx.layout.c_ref, _stream())
        return d, d, dw, db, None, None


def add_layer_norm(a, b, weight, bias, layout, eps=1e-5):
    """(a + b, LayerNorm(a + b))."""
    return _AddLayerNorm.apply(a, b, weight, bias, layout, eps)


# ------------------------------------------------------------------------------------------------- per-degree linear
class LinearSpec:
    """Pairs (degree-wise GEMMs) of a LinearRS / FCTP-with-scalar-attr between two row layouts.

    weight offsets follow e3nn's flat `tp.weight`: instructions ordered by input segment, each [mul_in, mul_out].
    `in_irreps` may be unsimplified (several consecutive segments of the same degree, e.g. the DTP output): the
    consecutive [mul_i, N] blocks of one degree form one row-major [K, N] matrix."""

    def __init__(self, in_layout, out_layout):
        self.in_layout, self.out_layout = in_layout, out_layout
        self.pairs = []  # (l, in_off, K, out_off, N, w_off)
        w_off = 0
        for (K, l), par, in_off in zip(in_layout.segs, in_layout.par, in_layout.offsets):
            j = out_layout.seg_index(l, par)  # a scalar second operand couples equal degree AND parity only
            if j is None:
                continue
            N = out_layout.segs[j][0]
            self.pairs.append((l, in_off, K, out_layout.offsets[j], N, w_off))
            w_off += K * N
        self.weight_numel = w_off
        self.out_covered = len(self.pairs) == len(out_layout.segs)
        self.in_covered = len(self.pairs) == len(in_layout.segs)
        self.bias_dim = out_layout.mul_of(0)  # bias on 0e only
        # the pair that carries the bias is the one writing the 0e segment: a 0o segment (E(3) irreps) also has l == 0 but
        # no bias (round 4: the parity-blind `l == 0` test added the 0e bias to the 0o outputs and its column sums to the
        # bias gradient -- invisible with zero-initialised biases, 11 % on the OC20 auxiliary head with filled ones)
        j0 = out_layout.seg_index(0, 1)
        self.bias_out_off = out_layout.offsets[j0] if j0 is not None else None

    def has_bias(self, l, out_off):
        return l == 0 and out_off == self.bias_out_off


# per-degree / dense linears on the bf16 matrix cores (csrc/gemmx.hip) in every matrix mode but "fp32"; the switch exists for
# A/B measurements (tools/) and cross-checks
_gemmx = [True]


def _gemm_group(descs, st):
    """One launch per kind for up to 8 GEMMs: eqf_gemmx_group (split-precision / bf16 planes on the bf16 matrix cores) in the
    matrix modes split / bf16 / split6, eqf_gemm_group (exact-fp32 MFMA) in the mode fp32."""
    m = _MATRIX_MODES[_matrix_mode[0]] if _gemmx[0] else None
    step = 8 if m is None else 24
    for i in range(0, len(descs), step):
        chunk = descs[i:i + step]
        arr = (EqfGemmDesc * len(chunk))(*chunk)
        if m is None:
            call("eqf_gemm_group", arr, len(chunk), st)
        else:
            call("eqf_gemmx_group", arr, len(chunk), m, st)


def _desc(kind, A, ra, B, ldb, C, rc, bias, M, N, K):
    return EqfGemmDesc(_p(A[0], A[1]), _p(B[0], B[1]), _p(C[0], C[1]), _p(bias), ra, rc, int(ldb), int(M), int(N), int(K),
                       0, kind)


def _lin_fwd_descs(x, weight, bias, spec):
    """(output tensor, descriptors of the per-degree GEMMs out_l = x_l W_l (+ bias on 0e))"""
    n = x.shape[0]
    Din, Dout = spec.in_layout.dim, spec.out_layout.dim
    out = (torch.empty if spec.out_covered else torch.zeros)((n, Dout), device=x.device, dtype=torch.float32)
    descs = []
    for (l, in_off, K, out_off, N, w_off) in spec.pairs:
        d = 2 * l + 1
        b = bias if (spec.has_bias(l, out_off) and bias is not None) else None
        descs.append(_desc(0, (x, in_off), rows(d, Din, K), (weight, w_off), N, (out, out_off), rows(d, Dout, N), b,
                           n * d, N, K))
    return out, descs


def _lin_fwd(x, weight, bias, spec):
    out, descs = _lin_fwd_descs(x, weight, bias, spec)
    _gemm_group(descs, _stream())
    return out


def _lin_dgrad_descs(dy, weight, spec):
    n = dy.shape[0]
    Din, Dout = spec.in_layout.dim, spec.out_layout.dim
    dx = (torch.empty if spec.in_covered else torch.zeros)((n, Din), device=dy.device, dtype=torch.float32)
    descs = []
    for (l, in_off, K, out_off, N, w_off) in spec.pairs:
        d = 2 * l + 1
        descs.append(_desc(1, (dy, out_off), rows(d, Dout, N), (weight, w_off), N, (dx, in_off), rows(d, Din, K), None,
                           n * d, K, N))
    return dx, descs


def _lin_dgrad(dy, weight, spec):
    dx, descs = _lin_dgrad_descs(dy, weight, spec)
    _gemm_group(descs, _stream())
    return dx


def _lin_wgrad_descs(x, dy, spec, dw, db=None):
    n = x.shape[0]
    Din, Dout = spec.in_layout.dim, spec.out_layout.dim
    descs = []
    for (l, in_off, K, out_off, N, w_off) in spec.pairs:
        d = 2 * l + 1
        # kind 2: C[K,N] += sum_rows x[row, 0:K]^T dy[row, 0:N]; "rc" describes the dy rows, ldb = ldc
        descs.append(_desc(2, (x, in_off), rows(d, Din, K), (dy, out_off), N, (dw, w_off), rows(d, Dout, N),
                           db if spec.has_bias(l, out_off) else None, K, N, n * d))
    return descs


# Weight gradients of the node-row linears are not on backward's dependency chain: dW = x^T dy needs nothing that comes later
# and nothing later needs dW.  Each of these launches costs 16-24 us whatever it computes (2 304 rows; profiles/r04), so during
# loss.backward() they are QUEUED and launched TOGETHER when the autograd engine finishes the pass (engine callback): 27 launches
# of 3 problems become 4 launches of <= 24.  On by default (ops.set_deferred_weight_gradients(False) / EQF_DEFER_WGRAD=0 turn it
# off); it only ever applies where it is provably safe:
#
#   * backward() returns a zero-initialised tensor as the gradient of a LEAF parameter; AccumulateGrad makes it (or a copy of it)
#     the parameter's .grad, and the queued launch accumulates into whatever tensor the parameter's .grad IS when the pass ends;
#   * only passes that ACCUMULATE into .grad qualify (`loss.backward()`): under torch.autograd.grad(loss, params) the engine
#     captures gradients instead and sums several contributions of one parameter out of place, which would lose a queued one --
#     `_accumulates_into_grad` asks the engine itself (the parameter's AccumulateGrad node will run in this graph task);
#   * a create_graph pass (grad mode on inside backward) computes everything at once;
#   * the queue belongs to ONE graph task: a nested / re-entrant backward (torch.utils.checkpoint, a Function that calls
#     backward inside its backward) or a pass on another thread has its own queue and its own callback.  The queue object is owned
#     by the engine callback's closure only (the module keeps a weak reference), so a pass that dies inside backward takes its
#     entries with it: they can neither be launched into recycled memory nor block a later pass;
#   * consumers that read gradients DURING backward (FlatGradAllReduce's tail hook, any post-accumulate-grad hook) call
#     `flush_deferred_weight_gradients()` first: what is queued so far is launched (stream-ordered before whatever the hook
#     enqueues next), the rest of the pass queues anew.

_defer_wgrad = [os.environ.get("EQF_DEFER_WGRAD", "1") != "0"]
_defer_stats = {"queued": 0, "flushes": 0}


class _TaskQueue:
    """queued weight-gradient problems of one graph task"""
    __slots__ = ("entries", "__weakref__")

    def __init__(self):
        self.entries = []


_task_queues = weakref.WeakValueDictionary()  # graph task id -> _TaskQueue (kept alive by that task's engine callback only)


def note_create_graph():
    """called by the create_graph branches of the operators' backward (kept for callers; deferral keys on the grad mode itself)"""


def set_deferred_weight_gradients(on):
    prev = _defer_wgrad[0]
    _defer_wgrad[0] = bool(on)
    return prev


def deferred_weight_gradient_stats(reset=False):
    """{"queued": problems queued, "flushes": grouped launches} since the last reset: lets a caller (bench.py, the tests) state
    whether the deferred path actually ran"""
    out = dict(_defer_stats)
    if reset:
        _defer_stats["queued"] = _defer_stats["flushes"] = 0
    return out


def _alias(t):
    """(storage, offset, numel) of a tensor: keeps the MEMORY alive without holding the tensor object (AccumulateGrad only adopts
    a gradient tensor nobody else references)"""
    return (t.untyped_storage(), t.storage_offset(), t.numel(), t.device)


def _from_alias(a):
    stor, off, n, dev = a
    return torch.empty(0, dtype=torch.float32, device=dev).set_(stor, off, (n,))


def _launch_queue(q, early=False):
    """early: a flush from inside the pass (gradient hooks).  Only the entries whose parameter has ALREADY been accumulated go out
    then: while .grad is still None the zero tensor backward() returned may have been summed OUT of place with another
    contribution of the same parameter (a weight used several times: the engine's input buffer), and a launch into the dead
    zero tensor would lose the gradient; those entries wait for the end of the pass (round-5 advisor finding)."""
    entries, q.entries = q.entries, []
    if early:
        ready = [e for e in entries if e[0].grad is not None and (not e[5] or e[1].grad is not None)]
        q.entries = [e for e in entries if not (e[0].grad is not None and (not e[5] or e[1].grad is not None))]
        entries = ready
    descs = []
    for (w, b, x, dy, spec, fused_b, aw, ab) in entries:
        # where the gradient lives now: the zero tensor backward() returned -- adopted as .grad by AccumulateGrad, or still on
        # its way to it (an early flush: AccumulateGrad then adopts or copies the FILLED tensor, stream-ordered) -- unless
        # AccumulateGrad already made a copy of it (then .grad is another tensor and receives the launch)
        tw = _from_alias(aw)
        if w.grad is not None and w.grad.data_ptr() != tw.data_ptr():
            tw = w.grad.view(-1)
        tb = None
        if fused_b:
            tb = _from_alias(ab)
            if b.grad is not None and b.grad.data_ptr() != tb.data_ptr():
                tb = b.grad.view(-1)
        if not tw.is_contiguous() or tw.dtype != torch.float32 or (tb is not None and not tb.is_contiguous()):
            raise RuntimeError("deferred weight gradient: .grad must be a contiguous fp32 tensor")
        descs += _lin_wgrad_descs(x, dy, spec, tw, tb)
    if descs:
        _defer_stats["flushes"] += 1
        _gemm_group(descs, _stream())


def flush_deferred_weight_gradients():
    """Launch what the CURRENT backward pass has queued so far (no-op outside a pass or with an empty queue).  For code that
    consumes gradients during backward: gradient hooks, bucketed reducers."""
    try:
        task = torch._C._current_graph_task_id()
    except Exception:
        return
    q = _task_queues.get(task)
    if q is not None and q.entries:
        _launch_queue(q, early=True)


def _hooked(p):
    """Somebody reads this parameter's gradient DURING backward -- a tensor hook, a post-accumulate-grad hook, or (not visible
    from Python) a hook on its AccumulateGrad node, which is what torch DistributedDataParallel's reducer installs: at that
    moment a deferred gradient is still the zero tensor.  FlatGradAllReduce marks its parameters (`_eqf_flushes`): its hook
    flushes the queue before it reads.  Any other hook, and any initialised process group without that mark (stock DDP: wrong
    gradients at world size 1 already, round-5 advisor finding), switches deferral off for the parameter."""
    if getattr(p, "_eqf_flushes", False):
        return False
    if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
        return True
    dist = torch.distributed
    return bool(dist.is_available() and dist.is_initialized())


def _accumulates_into_grad(p):
    """True iff this graph task will run the parameter's AccumulateGrad node, i.e. the pass is a .backward() that stores into
    p.grad (torch.autograd.grad captures the gradient at that node instead: the engine refuses the question for such a leaf)"""
    try:
        return bool(torch._C._will_engine_execute_node(torch.autograd.graph.get_gradient_edge(p).node))
    except RuntimeError:
        return False


def _can_defer(*params):
    """plain first-order .backward(), leaf parameters without an existing .grad (an existing one is added to OUT of place or in
    place depending on the engine's mood: those gradients are computed at once)"""
    return (_defer_wgrad[0] and not torch.is_grad_enabled()
            and all(p is None or (p.is_leaf and p.requires_grad and p.grad is None and not _hooked(p)
                                  and _accumulates_into_grad(p))
                    for p in params))


def _defer_lin_wgrad(w, b, x, dy, spec, fused_b, dw, db):
    task = torch._C._current_graph_task_id()
    q = _task_queues.get(task)
    if q is None:
        q = _TaskQueue()
        _task_queues[task] = q
        # the closure is the ONLY strong reference to the queue: it lives exactly as long as the graph task does
        torch.autograd.Variable._execution_engine.queue_callback(lambda q=q: _launch_queue(q))
    q.entries.append((w, b, x, dy, spec, fused_b, _alias(dw), _alias(db) if fused_b else None))
    _defer_stats["queued"] += 1


def _lin_wgrad(x, dy, spec, dw, db=None):
    """dw (flat, zero-initialised by the caller) += x^T dy per degree; db (zero-initialised, optional) += the column
    sums of the scalar block of dy, accumulated by the same launch while dy is staged."""
    _gemm_group(_lin_wgrad_descs(x, dy, spec, dw, db), _stream())
    return dw


class _LinDgrad(Function):
    """dx = dy W^T as a differentiable op (used only when the backward runs with create_graph=True)."""

    @staticmethod
    def forward(ctx, dy, weight, spec):
        dy, weight = _c(dy), _c(weight)
        _chk(dy, weight)
        ctx.save_for_backward(dy, weight)
        ctx.spec = spec
        return _lin_dgrad(dy, weight, spec)

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        dy, weight = ctx.saved_tensors
        c = _c(c)
        _chk(c)
        g_dy = _lin_fwd(c, weight, None, ctx.spec) if ctx.needs_input_grad[0] else None
        g_w = None
        if ctx.needs_input_grad[1]:
            g_w = _lin_wgrad(c, dy, ctx.spec, _zeros_like(weight))
        return g_dy, g_w, None


class _LinWgrad(Function):
    """dW = x^T dy as a differentiable op (create_graph only)."""

    @staticmethod
    def forward(ctx, x, dy, spec):
        x, dy = _c(x), _c(dy)
        _chk(x, dy)
        ctx.save_for_backward(x, dy)
        ctx.spec = spec
        return _lin_wgrad(x, dy, spec, _zeros(spec.weight_numel, device=x.device, dtype=torch.float32))

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        x, dy = ctx.saved_tensors
        c = _c(c)
        _chk(c)
        g_x = _lin_dgrad(dy, c, ctx.spec) if ctx.needs_input_grad[0] else None
        g_dy = _lin_fwd(x, c, None, ctx.spec) if ctx.needs_input_grad[1] else None
        return g_x, g_dy, None


class _IrrepsLinear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, spec):
        x = _c(x)
        weight = _c(weight)
        _chk(x, weight, bias)
        Din = spec.in_layout.dim
        assert x.shape[1] == Din and weight.numel() == spec.weight_numel
        out = _lin_fwd(x, weight, bias, spec)
        ctx.save_for_backward(x, weight)
        ctx.spec = spec
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if (bias is not None and bias.is_leaf) else None  # (identity only: for the deferred gradients)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        spec = ctx.spec
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if torch.is_grad_enabled():  # create_graph: every piece is itself differentiable
            note_create_graph()
            dx = _LinDgrad.apply(dy, weight, spec) if ctx.needs_input_grad[0] else None
            if not _want_param_grads():
                return dx, None, None, None
            dw = _LinWgrad.apply(x, dy, spec) if ctx.needs_input_grad[1] else None
            db = None
            if want_b:
                j = spec.out_layout.seg_index(0)
                o = spec.out_layout.offsets[j]
                db = dy[:, o:o + spec.bias_dim].sum(0)
            return dx, dw, db, None
        dy = _c(dy)
        _chk(dy)
        n = x.shape[0]
        Dout = spec.out_layout.dim
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _lin_dgrad(dy, weight, spec)
        if not _want_param_grads():  # force evaluation: d E / d pos only
            return dx, None, None, None
        if ctx.needs_input_grad[1] or want_b:
            dw_, db_ = _zeros2(weight.numel(), spec.bias_dim if want_b else 0, x.device)
        fused_b = want_b and ctx.needs_input_grad[1] and any(spec.has_bias(l, o) and N == spec.bias_dim
                                                              for (l, _, _, o, N, _) in spec.pairs)
        if ctx.needs_input_grad[1]:
            bp = ctx.bias_param if fused_b else None
            if _can_defer(weight, bp) and (not fused_b or bp is not None):
                _defer_lin_wgrad(weight, bp, x, dy, spec, fused_b, dw_, db_)  # (zeros now) filled when backward ends
                dw = dw_
            else:
                dw = _lin_wgrad(x, dy, spec, dw_, db_ if fused_b else None)
        if want_b:
            db = db_
            if not fused_b:
                j = spec.out_layout.seg_index(0)
                call("eqf_colsum", _p(dy, spec.out_layout.offsets[j]), rows(1, Dout, 0), n, spec.bias_dim, _p(db), st)
        return dx, dw, db, None


def irreps_linear(x, weight, bias, spec):
    return _IrrepsLinear.apply(x, weight, bias, spec)


class _IrrepsLinearPair(Function):
    """Two per-degree linears of the SAME input (GraphAttention's merge_src / merge_dst, nets/graph_attention_transformer.py:
    485-486) with their GEMMs side by side in one launch: forward 1 launch instead of 2, data gradients 1 + one add instead of
    2 + autograd's add, weight gradients 1 instead of 2 -- these node-row launches cost ~12 us each whatever they compute
    (tools/gemm_shapes.py).  Under create_graph the backward is the two linears' own differentiable pieces."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, spec1, spec2):
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        _chk(x, w1, b1, w2, b2)
        assert x.shape[1] == spec1.in_layout.dim == spec2.in_layout.dim
        y1, d1 = _lin_fwd_descs(x, w1, b1, spec1)
        y2, d2 = _lin_fwd_descs(x, w2, b2, spec2)
        _gemm_group(d1 + d2, _stream())
        ctx.save_for_backward(x, w1, w2)
        ctx.specs = (spec1, spec2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.bias_params = tuple(b if (b is not None and b.is_leaf) else None for b in (b1, b2))
        return y1, y2

    @staticmethod
    def backward(ctx, dy1, dy2):
        x, w1, w2 = ctx.saved_tensors
        spec1, spec2 = ctx.specs
        need = ctx.needs_input_grad  # x, w1, b1, w2, b2
        ws, specs, dys, hb = (w1, w2), (spec1, spec2), [dy1, dy2], ctx.has_bias
        for i in (0, 1):
            if dys[i] is None:
                dys[i] = _zeros((x.shape[0], specs[i].out_layout.dim), device=x.device, dtype=torch.float32)
        if torch.is_grad_enabled():  # create_graph: every piece is itself differentiable
            note_create_graph()
            dx = None
            if need[0]:
                dx = _LinDgrad.apply(dys[0], w1, spec1) + _LinDgrad.apply(dys[1], w2, spec2)
            if not _want_param_grads():
                return dx, None, None, None, None, None, None
            out = [dx]
            for i in (0, 1):
                out.append(_LinWgrad.apply(x, dys[i], specs[i]) if need[1 + 2 * i] else None)
                db = None
                if hb[i] and need[2 + 2 * i]:
                    o = specs[i].out_layout.offsets[specs[i].out_layout.seg_index(0)]
                    db = dys[i][:, o:o + specs[i].bias_dim].sum(0)
                out.append(db)
            return tuple(out) + (None, None)
        dys = [_c(d) for d in dys]
        _chk(*dys)
        st = _stream()
        dx = None
        if need[0]:
            dxa, da = _lin_dgrad_descs(dys[0], w1, spec1)
            dxb, db_ = _lin_dgrad_descs(dys[1], w2, spec2)
            _gemm_group(da + db_, st)
            dx = dxa.add_(dxb)
        if not _want_param_grads():
            return dx, None, None, None, None, None, None
        grads = [None, None, None, None]  # dw1, db1, dw2, db2
        descs, late = [], []
        for i in (0, 1):
            want_b = hb[i] and need[2 + 2 * i]
            if not (need[1 + 2 * i] or want_b):
                continue
            dw_, dbv = _zeros2(ws[i].numel(), specs[i].bias_dim if want_b else 0, x.device)
            fused_b = want_b and need[1 + 2 * i] and any(specs[i].has_bias(l, o) and N == specs[i].bias_dim
                                                         for (l, _, _, o, N, _) in specs[i].pairs)
            if need[1 + 2 * i]:
                bp = ctx.bias_params[i] if fused_b else None
                if _can_defer(ws[i], bp) and (not fused_b or bp is not None):
                    _defer_lin_wgrad(ws[i], bp, x, dys[i], specs[i], fused_b, dw_, dbv)
                else:
                    descs += _lin_wgrad_descs(x, dys[i], specs[i], dw_, dbv if fused_b else None)
                grads[2 * i] = dw_
            if want_b:
                grads[2 * i + 1] = dbv
                if not fused_b:
                    late.append((i, dbv))
        if descs:
            _gemm_group(descs, st)
        for i, dbv in late:
            o = specs[i].out_layout.offsets[specs[i].out_layout.seg_index(0)]
            call("eqf_colsum", _p(dys[i], o), rows(1, specs[i].out_layout.dim, 0), x.shape[0], specs[i].bias_dim, _p(dbv), st)
        return dx, grads[0], grads[1], grads[2], grads[3], None, None


def irreps_linear_pair(x, w1, b1, spec1, w2, b2, spec2):
    return _IrrepsLinearPair.apply(x, w1, b1, w2, b2, spec1, spec2)


def _dense_fwd(x, weight, bias):
    M, K = x.shape
    N = weight.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    # (x may be a column block of a wider row-major tensor: row stride x.stride(0), read in place)
    _gemm_group([_desc(1, (x, 0), rows(1, x.stride(0), 0), (weight, 0), K, (y, 0), rows(1, N, 0), bias, M, N, K)], _stream())
    return y


def _dense_dgrad(dy, weight):
    M, N = dy.shape
    K = weight.shape[1]
    dx = torch.empty((M, K), device=dy.device, dtype=torch.float32)
    _gemm_group([_desc(0, (dy, 0), rows(1, N, 0), (weight, 0), K, (dx, 0), rows(1, K, 0), None, M, K, N)], _stream())
    return dx


def _dense_wgrad(x, dy, dw, db=None):
    """dw [N, K] (zero-initialised) += dy^T x; db [N] (zero-initialised, optional) += column sums of dy (same launch)"""
    M, K = x.shape
    N = dy.shape[1]
    # (kind 3: `rc` carries the row stride of x -- possibly a column block of a wider tensor -- and ldb the leading dimension of dw)
    _gemm_group([_desc(3, (dy, 0), rows(1, N, 0), (x, 0), K, (dw, 0), rows(1, x.stride(0), 0), db, N, K, M)], _stream())
    return dw


class _DenseDgrad(Function):
    @staticmethod
    def forward(ctx, dy, weight):
        dy, weight = _c(dy), _c(weight)
        _chk(dy, weight)
        ctx.save_for_backward(dy, weight)
        return _dense_dgrad(dy, weight)

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        dy, weight = ctx.saved_tensors
        c = _c(c)
        _chk(c)
        g_dy = _dense_fwd(c, weight, None) if ctx.needs_input_grad[0] else None
        g_w = _dense_wgrad(c, dy, _zeros_like(weight)) if ctx.needs_input_grad[1] else None
        return g_dy, g_w


class _DenseWgrad(Function):
    @staticmethod
    def forward(ctx, x, dy):
        x, dy = _c(x), _c(dy)
        _chk(x, dy)
        ctx.save_for_backward(x, dy)
        return _dense_wgrad(x, dy, _zeros((dy.shape[1], x.shape[1]), device=x.device, dtype=torch.float32))

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        x, dy = ctx.saved_tensors
        c = _c(c)
        _chk(c)
        g_x = _dense_dgrad(dy, c) if ctx.needs_input_grad[0] else None
        g_dy = _dense_fwd(x, c, None) if ctx.needs_input_grad[1] else None
        return g_x, g_dy


class _DenseLinear(Function):
    """torch.nn.Linear semantics (y = x W^T + b) on the exact-fp32 MFMA path."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if not (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.stride(0) % 4 == 0
                and x.storage_offset() % 4 == 0):  # (a column block of a wider tensor is read in place: the radial bank's hidden rows)
            x = _c(x)
        weight = _c(weight)
        _chk(x if x.is_contiguous() else x[:1, :1], weight, bias)  # (device / dtype of a row-strided x through its corner)
        y = _dense_fwd(x, weight, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if torch.is_grad_enabled():  # create_graph
            dx = _DenseDgrad.apply(dy, weight) if ctx.needs_input_grad[0] else None
            if not _want_param_grads():
                return dx, None, None
            dw = _DenseWgrad.apply(x, dy) if ctx.needs_input_grad[1] else None
            db = dy.sum(0) if want_b else None
            return dx, dw, db
        dy = _c(dy)
        _chk(dy)
        M, K = x.shape
        N = weight.shape[0]
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dense_dgrad(dy, weight)
        if not _want_param_grads():  # force evaluation
            return dx, None, None
        if ctx.needs_input_grad[1] or want_b:
            dw_, db_ = _zeros2(weight.numel(), N if want_b else 0, x.device)
        if ctx.needs_input_grad[1]:
            dw = _dense_wgrad(x, dy, dw_.view_as(weight), db_ if want_b else None)
        if want_b:
            db = db_
            if not ctx.needs_input_grad[1]:
                call("eqf_colsum", _p(dy), rows(1, N, 0), M, N, _p(db), st)
        return dx, dw, db


def dense_linear(x, weight, bias=None):
    return _DenseLinear.apply(x, weight, bias)


def split_columns(x, G):
    """G equal column blocks of a row-major [rows, G C] tensor, as views (their consumers read them in place; the backward is one
    concatenation of the G gradients)"""
    return x.split(x.shape[1] // G, dim=1)


# ------------------------------------------------------------------------------------------------- activations
class _Gate(Function):
    @staticmethod
    def forward(ctx, x, S, gated_layout, c_silu, c_sig):
        x = _c(x)
        _chk(x)
        n = x.shape[0]
        G = sum(m for m, _ in gated_layout.segs)
        Dout = S + gated_layout.dim
        assert x.shape[1] == S + G + gated_layout.dim
        y = torch.empty((n, Dout), device=x.device, dtype=torch.float32)
        call("eqf_gate_fwd", _p(x), _p(y), n, S, gated_layout.c_ref, c_silu, c_sig, _stream())
        ctx.save_for_backward(x)
        ctx.args = (S, gated_layout, c_silu, c_sig)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        S, gated_layout, c_silu, c_sig = ctx.args
        if torch.is_grad_enabled():  # create_graph
            return _GateBwd.apply(x, dy, S, gated_layout, c_silu, c_sig), None, None, None, None
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        call("eqf_gate_bwd", _p(x), _p(dy), _p(dx), x.shape[0], S, gated_layout.c_ref, c_silu, c_sig, _stream())
        return dx, None, None, None, None


class _GateBwd(Function):
    @staticmethod
    def forward(ctx, x, dy, S, gated_layout, c_silu, c_sig):
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        call("eqf_gate_bwd", _p(x), _p(dy), _p(dx), x.shape[0], S, gated_layout.c_ref, c_silu, c_sig, _stream())
        ctx.save_for_backward(x, dy)
        ctx.args = (S, gated_layout, c_silu, c_sig)
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        x, dy = ctx.saved_tensors
        S, gated_layout, c_silu, c_sig = ctx.args
        c = _c(c)
        _chk(c)
        g_x, g_dy = torch.empty_like(x), torch.empty_like(dy)
        call("eqf_gate_bwd2", _p(x), _p(dy), _p(c), _p(g_x), _p(g_dy), x.shape[0], S, gated_layout.c_ref, c_silu, c_sig,
             _stream())
        return g_x, g_dy, None, None, None, None


def gate(x, S, gated_layout, c_silu, c_sig):
    return _Gate.apply(x, S, gated_layout, c_silu, c_sig)


class _ScaledSilu(Function):
    @staticmethod
    def forward(ctx, x, c):
        x = _c(x)
        _chk(x)
        y = torch.empty_like(x)
        call("eqf_silu_fwd", _p(x), _p(y), x.numel(), c, _stream())
        ctx.save_for_backward(x)
        ctx.c = c
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():  # create_graph
            return _ScaledSiluBwd.apply(x, dy, ctx.c), None
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        call("eqf_silu_bwd", _p(x), _p(dy), _p(dx), x.numel(), ctx.c, _stream())
        return dx, None


class _ScaledSiluBwd(Function):
    @staticmethod
    def forward(ctx, x, dy, c0):
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        call("eqf_silu_bwd", _p(x), _p(dy), _p(dx), x.numel(), c0, _stream())
        ctx.save_for_backward(x, dy)
        ctx.c0 = c0
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        x, dy = ctx.saved_tensors
        c = _c(c)
        _chk(c)
        g_x, g_dy = torch.empty_like(x), torch.empty_like(x)
        call("eqf_silu_bwd2", _p(x), _p(dy), _p(c), _p(g_x), _p(g_dy), x.numel(), ctx.c0, _stream())
        return g_x, g_dy, None


def scaled_silu(x, c):
    return _ScaledSilu.apply(x, c)


class _LnSilu(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, groups=1):
        x = _c(x)
        _chk(x, gamma, beta)
        y = torch.empty_like(x)
        call("eqf_lnsilu_group_fwd", _p(x), _p(gamma), _p(beta), _p(y), x.shape[0], x.shape[1] // groups, groups, eps,
             _stream())
        ctx.save_for_backward(x, gamma, beta)
        ctx.eps, ctx.groups = eps, groups
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        G = ctx.groups
        if torch.is_grad_enabled():  # create_graph
            dx, dg, db = _LnSiluBwd.apply(x, gamma, beta, dy, ctx.eps, G)
            return dx, _guard_opt(dg, dy, "LayerNorm weight gradient"), _guard_opt(db, dy, "LayerNorm bias gradient"), \
                None, None
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        dg, db = _zeros2(gamma.numel(), beta.numel(), x.device)
        call("eqf_lnsilu_group_bwd", _p(x), _p(gamma), _p(beta), _p(dy), _p(dx), _p(dg), _p(db), x.shape[0],
             x.shape[1] // G, G, ctx.eps, _stream())
        return dx, dg, db, None, None


class _LnSiluBwd(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, dy, eps, groups=1):
        dy = _c(dy)
        _chk(dy)
        dx = torch.empty_like(x)
        dg, db = _zeros2(gamma.numel(), beta.numel(), x.device)
        call("eqf_lnsilu_group_bwd", _p(x), _p(gamma), _p(beta), _p(dy), _p(dx), _p(dg), _p(db), x.shape[0],
             x.shape[1] // groups, groups, eps, _stream())
        ctx.save_for_backward(x, gamma, beta, dy)
        ctx.eps, ctx.groups = eps, groups
        if not _want_param_grads():
            return dx, None, None
        ctx.mark_non_differentiable(dg, db)
        return dx, dg, db

    @staticmethod
    @once_differentiable
    def backward(ctx, c, _cg=None, _cb=None):
        x, gamma, beta, dy = ctx.saved_tensors
        c = _c(c)
        _chk(c)
        g_x, g_dy = torch.empty_like(x), torch.empty_like(x)
        g_g, g_b = _zeros2(gamma.numel(), beta.numel(), x.device)
        G = ctx.groups
        call("eqf_lnsilu_group_bwd2", _p(x), _p(gamma), _p(beta), _p(dy), _p(c), _p(g_x), _p(g_g), _p(g_b), _p(g_dy),
             x.shape[0], x.shape[1] // G, G, ctx.eps, _stream())
        return g_x, g_g, g_b, g_dy, None, None


def ln_silu(x, gamma, beta, eps=1e-5, groups=1):
    """silu(LayerNorm(x) * gamma + beta) on rows of C = x.shape[1] / groups channels; with groups > 1 the row holds
    `groups` independent feature vectors with their own gamma / beta ([groups * C])."""
    return _LnSilu.apply(x, gamma, beta, eps, int(groups))


def _glin_views(dys, wide, Ns, rows_n, dev):
    """(per-group dy tensors, their column offsets, their row stride or None) of a grouped linear's output gradient"""
    G = len(Ns)
    if wide:
        dy = _c(dys[0])
        _chk(dy)
        return [dy] * G, [sum(Ns[:g]) for g in range(G)], sum(Ns)
    dyt = [(_c(d) if d is not None else _zeros((rows_n, Ns[g]), dev)) for g, d in enumerate(dys)]
    _chk(*dyt)
    return dyt, [0] * G, None


def _glin_fwd(x, K, wide, Ws, bs):
    """y_g = x[:, g K:(g+1) K] W_g^T (+ b_g) for all g in one launch; one [rows, sum N] tensor if `wide`, else a tuple"""
    G = len(Ws)
    rows_n, ldx = x.shape
    Ns = [int(W.shape[0]) for W in Ws]
    if wide:
        out = torch.empty((rows_n, sum(Ns)), device=x.device, dtype=torch.float32)
        outs, ldo, offs = [out] * G, sum(Ns), [sum(Ns[:g]) for g in range(G)]
    else:
        outs = [torch.empty((rows_n, n), device=x.device, dtype=torch.float32) for n in Ns]
        ldo, offs = None, [0] * G
    descs = [_desc(1, (x, g * K), rows(1, ldx, 0), (Ws[g], 0), K, (outs[g], offs[g]),
                   rows(1, ldo if wide else Ns[g], 0), bs[g], rows_n, Ns[g], K) for g in range(G)]
    _gemm_group(descs, _stream())
    return out if wide else tuple(outs)


def _glin_dgrad(dyt, offs, ldd, Ws, K, rows_n):
    """dx[:, g K:(g+1) K] = dy_g W_g"""
    G = len(Ws)
    Ns = [int(W.shape[0]) for W in Ws]
    dx = torch.empty((rows_n, G * K), device=dyt[0].device, dtype=torch.float32)
    _gemm_group([_desc(0, (dyt[g], offs[g]), rows(1, ldd if ldd is not None else Ns[g], 0), (Ws[g], 0), K, (dx, g * K),
                       rows(1, G * K, 0), None, rows_n, K, Ns[g]) for g in range(G)], _stream())
    return dx


def _glin_wgrad(dyt, offs, ldd, x, K, Ns, has_b):
    """dW_g[N_g, K] = dy_g^T x_g, db_g = column sums of dy_g (same launch); one zero-filled flat buffer behind all of them"""
    G = len(Ns)
    rows_n, ldx = x.shape
    sizes = [n * K for n in Ns] + [n if hb else 0 for n, hb in zip(Ns, has_b)]
    flat = _zeros(sum(sizes), x.device)
    o, dWs, dbs = 0, [], []
    for g in range(G):
        dWs.append(flat[o:o + Ns[g] * K].view(Ns[g], K))
        o += Ns[g] * K
    for g in range(G):
        dbs.append(flat[o:o + Ns[g]] if has_b[g] else None)
        o += Ns[g] if has_b[g] else 0
    _gemm_group([_desc(3, (dyt[g], offs[g]), rows(1, ldd if ldd is not None else Ns[g], 0), (x, g * K), K, (dWs[g], 0),
                       rows(1, ldx, 0), dbs[g], Ns[g], K, rows_n) for g in range(G)], _stream())
    return dWs, dbs


class _GroupedDgrad(Function):
    """dx = [dy_g W_g]_g as a differentiable op (create_graph only).  args = dys (one wide tensor or G tensors) + the G weights."""

    @staticmethod
    def forward(ctx, K, wide, G, *args):
        ndy = 1 if wide else G
        dys, Ws = args[:ndy], [_c(W) for W in args[ndy:]]
        Ns = [int(W.shape[0]) for W in Ws]
        rows_n = next(d.shape[0] for d in dys if d is not None)
        dyt, offs, ldd = _glin_views(dys, wide, Ns, rows_n, Ws[0].device)
        ctx.save_for_backward(*dyt[:ndy], *Ws)
        ctx.meta = (K, wide, G, Ns, rows_n)
        return _glin_dgrad(dyt, offs, ldd, Ws, K, rows_n)

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        K, wide, G, Ns, rows_n = ctx.meta
        ndy = 1 if wide else G
        saved = ctx.saved_tensors
        dys, Ws = saved[:ndy], list(saved[ndy:])
        c = _c(c)
        _chk(c)
        g_dys = _glin_fwd(c, K, wide, Ws, [None] * G)  # d <c, dx> / d dy_g = c_g W_g^T
        g_dys = (g_dys,) if wide else tuple(g_dys)
        dyt, offs, ldd = _glin_views(dys, wide, Ns, rows_n, c.device)
        g_Ws, _ = _glin_wgrad(dyt, offs, ldd, c, K, Ns, [False] * G)  # d <c, dx> / d W_g = dy_g^T c_g
        return (None, None, None) + g_dys + tuple(g_Ws)


class _GroupedWgrad(Function):
    """(dW_0..dW_{G-1}, db_g of the groups that have a bias) as a differentiable op (create_graph only).  args = x + dys."""

    @staticmethod
    def forward(ctx, K, wide, Ns, has_b, x, *dys):
        x = _c(x)
        _chk(x)
        dyt, offs, ldd = _glin_views(dys, wide, list(Ns), x.shape[0], x.device)
        ndy = 1 if wide else len(Ns)
        ctx.save_for_backward(x, *dyt[:ndy])
        ctx.meta = (K, wide, list(Ns), list(has_b))
        dWs, dbs = _glin_wgrad(dyt, offs, ldd, x, K, list(Ns), list(has_b))
        return tuple(dWs) + tuple(b for b in dbs if b is not None)

    @staticmethod
    @once_differentiable
    def backward(ctx, *cs):
        K, wide, Ns, has_b = ctx.meta
        G = len(Ns)
        x, *dys = ctx.saved_tensors
        dev = x.device
        cWs = [(_c(cs[g]) if cs[g] is not None else _zeros((Ns[g], K), dev)) for g in range(G)]
        it = iter(cs[G:])
        cbs = [(next(it) if hb else None) for hb in has_b]
        cbs = [(_c(b) if b is not None else None) for b in cbs]
        dyt, offs, ldd = _glin_views(dys, wide, Ns, x.shape[0], dev)
        g_x = _glin_dgrad(dyt, offs, ldd, cWs, K, x.shape[0]) if ctx.needs_input_grad[4] else None  # dy_g cW_g
        g_dys = _glin_fwd(x, K, wide, cWs, cbs)  # x_g cW_g^T + cb_g
        g_dys = (g_dys,) if wide else tuple(g_dys)
        return (None, None, None, None, g_x) + g_dys


class _GroupedLinear(Function):
    """G independent nn.Linear layers on the column blocks of one wide input: y_g = x[:, g K:(g+1) K] W_g^T + b_g, all G
    GEMMs in ONE launch (eqf_gemm_group), forward and both gradients.  `wide`: the outputs form one [rows, sum N_g]
    tensor (returned as such), otherwise G separate [rows, N_g] tensors.  params = (W_0..W_{G-1}, b_0..b_{G-1}).
    Under create_graph the backward is made of differentiable grouped pieces (round 5: the radial bank no longer steps aside
    when forces are taken, MD17 / DeNS training)."""

    @staticmethod
    def forward(ctx, x, K, wide, *params):
        G = len(params) // 2
        Ws, bs = params[:G], params[G:]
        x = _c(x)
        _chk(x, *Ws, *bs)
        rows_n, ldx = x.shape
        Ns = [int(W.shape[0]) for W in Ws]
        assert ldx == G * K and all(W.shape[1] == K and W.is_contiguous() for W in Ws)
        out = _glin_fwd(x, K, wide, list(Ws), list(bs))
        ctx.save_for_backward(x, *Ws)
        ctx.meta = (G, K, wide, Ns, [b is not None for b in bs])
        return out

    @staticmethod
    def backward(ctx, *dys):
        x, *Ws = ctx.saved_tensors
        G, K, wide, Ns, has_b = ctx.meta
        rows_n, ldx = x.shape
        dev = x.device
        if torch.is_grad_enabled():  # create_graph: every piece is itself differentiable
            dlist = [dys[0]] if wide else [(d if d is not None else _zeros((rows_n, Ns[g]), dev)) for g, d in enumerate(dys)]
            dx = _GroupedDgrad.apply(K, wide, G, *dlist, *Ws) if ctx.needs_input_grad[0] else None
            if not _want_param_grads():
                return (dx, None, None) + (None,) * (2 * G)
            outs = _GroupedWgrad.apply(K, wide, tuple(Ns), tuple(has_b), x, *dlist)
            it = iter(outs[G:])
            return (dx, None, None) + tuple(outs[:G]) + tuple((next(it) if hb else None) for hb in has_b)
        dyt, offs, ldd = _glin_views(dys, wide, Ns, rows_n, dev)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _glin_dgrad(dyt, offs, ldd, list(Ws), K, rows_n)
        if not _want_param_grads():
            return (dx, None, None) + (None,) * (2 * G)
        # kind 3: dW_g[N_g, K] += dy_g^T x_g, db_g += column sums of dy_g (same launch)
        dWs, dbs = _glin_wgrad(dyt, offs, ldd, x, K, Ns, has_b)
        return (dx, None, None) + tuple(dWs) + tuple(dbs)


def grouped_linear(x, K, weights, biases, wide):
    return _GroupedLinear.apply(x, int(K), bool(wide), *weights, *biases)


class _FoldWeight(Function):
    """W'[row, :] = w[w_of_row[row]] * W[row, :] on a flat per-degree weight (shared depth-wise weights folded into the
    linear after the tensor product); multilinear, so the create_graph backward is itself made of fold products."""

    @staticmethod
    def forward(ctx, W, w, row_start, w_of_row):
        W, w = _c(W), _c(w)
        _chk(W, w, row_start, w_of_row)
        out = torch.empty_like(W)
        call("eqf_fold_weight_fwd", _p(W), _p(w), _p(row_start), _p(w_of_row), _p(out), w_of_row.numel(), _stream())
        ctx.save_for_backward(W, w, row_start, w_of_row)
        return out

    @staticmethod
    def backward(ctx, g):
        W, w, row_start, w_of_row = ctx.saved_tensors
        if torch.is_grad_enabled():  # create_graph: dW = fold(g, w); dw via the tensor identity (rare path, ATen)
            dW = _FoldWeight.apply(g, w, row_start, w_of_row)
            row_of = torch.repeat_interleave(torch.arange(w_of_row.numel(), device=W.device),
                                             (row_start[1:] - row_start[:-1]).long())
            dw = torch.zeros_like(w).index_add(0, w_of_row.long()[row_of], g * W)
            return dW, dw, None, None
        g = _c(g)
        _chk(g)
        dW = torch.empty_like(W)
        dw = torch.zeros_like(w) if w.numel() != w_of_row.numel() else torch.empty_like(w)
        call("eqf_fold_weight_bwd", _p(W), _p(w), _p(row_start), _p(w_of_row), _p(g), _p(dW), _p(dw), w_of_row.numel(),
             _stream())
        return dW, dw, None, None


def fold_weight(W, w, row_start, w_of_row):
    return _FoldWeight.apply(W, w, row_start, w_of_row)


# ------------------------------------------------------------------------------------------------- embedding
class _Embed(Function):
    @staticmethod
    def forward(ctx, types, W, b, D):
        _chk(types, W, b)
        n = types.shape[0]
        C = W.shape[1]
        y = torch.empty((n, D), device=W.device, dtype=torch.float32)
        call("eqf_embed_fwd", _p(types), _p(W), _p(b), _p(y), n, C, D, _stream())
        ctx.save_for_backward(types)
        ctx.shape = (W.shape, D, b is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (types,) = ctx.saved_tensors
        wshape, D, has_b = ctx.shape
        dy = _c(dy)
        _chk(dy)
        dW = _zeros(wshape, device=dy.device, dtype=torch.float32)
        db = _zeros(wshape[1], device=dy.device, dtype=torch.float32) if has_b else None
        call("eqf_embed_bwd", _p(types), _p(dy), _p(dW), _p(db), types.shape[0], wshape[1], D, _stream())
        return None, dW, db, None


def embed(types, W, b, D):
    """W: [num_types, C] (row lookup), b: [C]; output rows [D] with columns >= C zero."""
    return _Embed.apply(types, _c(W), b, D)


# ------------------------------------------------------------------------------------------------- graph ops
def _gather_add_fwd(a, b, graph):
    D = a.shape[1]
    msg = torch.empty((graph.E, D), device=a.device, dtype=torch.float32)
    call("eqf_gather_add_fwd", _p(a), _p(b), _p(graph.src), _p(graph.dst), _p(msg), graph.E, D, _stream())
    return msg


def _gather_add_bwd(dmsg, g, n, want_a, want_b):
    D = dmsg.shape[1]
    st = _stream()
    da = db = None
    if want_a:
        da = torch.empty((n, D), device=dmsg.device, dtype=torch.float32)
        call("eqf_segment_sum", _p(dmsg), _p(g.src_ptr), _p(g.src_perm), _p(da), n, D, 1.0, 0, st)
    if want_b:
        db = torch.empty((n, D), device=dmsg.device, dtype=torch.float32)
        call("eqf_segment_sum", _p(dmsg), _p(g.row_ptr), None, _p(db), n, D, 1.0, 0, st)
    return da, db


class _GatherAddBwd(Function):
    """(da, db) = adjoint of msg = a[src] + b[dst], as a differentiable (linear) op; create_graph only."""

    @staticmethod
    def forward(ctx, dmsg, graph, n, has_b):
        dmsg = _c(dmsg)
        _chk(dmsg)
        ctx.graph, ctx.has_b = graph, has_b
        da, db = _gather_add_bwd(dmsg, graph, n, True, has_b)
        if has_b:
            return da, db
        return da

    @staticmethod
    @once_differentiable
    def backward(ctx, ca, cb=None):
        ca = _c(ca) if ca is not None else None
        cb = _c(cb) if cb is not None else None
        if ca is None and cb is None:
            return None, None, None, None
        if ca is None:
            ca = _zeros_like(cb)
        _chk(ca, cb)
        return _gather_add_fwd(ca, cb, ctx.graph), None, None, None


class _GatherAdd(Function):
    @staticmethod
    def forward(ctx, a, b, graph):
        a = _c(a)
        b = _c(b) if b is not None else None
        _chk(a, b)
        msg = _gather_add_fwd(a, b, graph)
        ctx.graph = graph
        ctx.has_b = b is not None
        ctx.n = a.shape[0]
        return msg

    @staticmethod
    def backward(ctx, dmsg):
        g = ctx.graph
        if torch.is_grad_enabled():  # create_graph
            if ctx.has_b:
                da, db = _GatherAddBwd.apply(dmsg, g, ctx.n, True)
                return da, db, None
            return _GatherAddBwd.apply(dmsg, g, ctx.n, False), None, None
        dmsg = _c(dmsg)
        _chk(dmsg)
        da, db = _gather_add_bwd(dmsg, g, ctx.n, ctx.needs_input_grad[0], ctx.has_b and ctx.needs_input_grad[1])
        return da, db, None


def gather_add(a, b, graph):
    """msg[e] = a[src[e]] + b[dst[e]]  (b may be None)."""
    return _GatherAdd.apply(a, b, graph)


class _SegmentBcast(Function):
    """dx[q] = scale * dout[seg_of[q]] as a differentiable (linear) op; create_graph only."""

    @staticmethod
    def forward(ctx, dout, seg_of, ptr, n, scale):
        dout = _c(dout)
        _chk(dout)
        D = dout.shape[1]
        dx = torch.empty((n, D), device=dout.device, dtype=torch.float32)
        call("eqf_segment_bcast", _p(dout), _p(seg_of), _p(dx), n, D, scale, _stream())
        ctx.save_for_backward(ptr)
        ctx.args = (dout.shape[0], D, scale)
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        (ptr,) = ctx.saved_tensors
        nseg, D, scale = ctx.args
        c = _c(c)
        _chk(c)
        out = torch.empty((nseg, D), device=c.device, dtype=torch.float32)
        call("eqf_segment_sum", _p(c), _p(ptr), None, _p(out), nseg, D, scale, 0, _stream())
        return out, None, None, None, None


class _SegmentSum(Function):
    @staticmethod
    def forward(ctx, x, ptr, seg_of, nseg, scale):
        x = _c(x)
        _chk(x, ptr, seg_of)
        D = x.shape[1]
        out = torch.empty((nseg, D), device=x.device, dtype=torch.float32)
        call("eqf_segment_sum", _p(x), _p(ptr), None, _p(out), nseg, D, scale, 0, _stream())
        ctx.save_for_backward(seg_of, ptr)
        ctx.args = (x.shape[0], D, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        seg_of, ptr = ctx.saved_tensors
        n, D, scale = ctx.args
        if torch.is_grad_enabled():  # create_graph
            return _SegmentBcast.apply(dout, seg_of, ptr, n, scale), None, None, None, None
        dout = _c(dout)
        _chk(dout)
        dx = torch.empty((n, D), device=dout.device, dtype=torch.float32)
        call("eqf_segment_bcast", _p(dout), _p(seg_of), _p(dx), n, D, scale, _stream())
        return dx, None, None, None, None


def segment_sum(x, ptr, seg_of, nseg, scale=1.0):
    """out[s] = scale * sum of the rows of segment s (rows of a segment are contiguous; ptr = CSR offsets)."""
    return _SegmentSum.apply(x, ptr, seg_of, nseg, float(scale))


class _SegmentScale(Function):
    @staticmethod
    def forward(ctx, x, s, seg_of):
        x = _c(x)
        _chk(x, s, seg_of)
        out = torch.empty_like(x)
        call("eqf_segment_scale", _p(x), _p(s), _p(seg_of), _p(out), x.shape[0], x.shape[1], _stream())
        ctx.save_for_backward(s, seg_of)
        return out

    @staticmethod
    def backward(ctx, dout):
        s, seg_of = ctx.saved_tensors
        return _SegmentScale.apply(dout, s, seg_of), None, None  # linear: its own backward at every order


def segment_scale(x, s, seg_of):
    """out[q] = s[seg_of[q]] * x[q] (s: one non-differentiable factor per segment)."""
    return _SegmentScale.apply(x, s, seg_of)


class _EdgeGeom(Function):
    @staticmethod
    def forward(ctx, pos, offsets, graph, lmax):
        pos_in = pos
        pos = _c(pos)
        _chk(pos, offsets)
        E = graph.E
        vec = torch.empty((E, 3), device=pos.device, dtype=torch.float32)
        length = torch.empty((E,), device=pos.device, dtype=torch.float32)
        sh = torch.empty((E, (lmax + 1) ** 2), device=pos.device, dtype=torch.float32)
        call("eqf_edge_geom_fwd", _p(pos), _p(graph.src), _p(graph.dst), _p(offsets), E, lmax, _p(vec), _p(length),
             _p(sh), _stream())
        ctx.save_for_backward(vec, pos_in)
        ctx.graph, ctx.lmax, ctx.n = graph, lmax, pos.shape[0]
        ctx.offsets = offsets
        ctx.mark_non_differentiable(vec)
        return vec, length, sh

    @staticmethod
    def backward(ctx, _dvec, dlen, dsh):
        vec, pos_in = ctx.saved_tensors
        g = ctx.graph
        if torch.is_grad_enabled():  # create_graph
            return _EdgeGeomBwd.apply(pos_in, vec, dlen, dsh, g, ctx.lmax), None, None, None
        dlen = _c(dlen) if dlen is not None else None
        dsh = _c(dsh) if dsh is not None else None
        _chk(dlen, dsh)
        return _edge_geom_dpos(vec, dsh, dlen, g, ctx.lmax, ctx.n), None, None, None


def _edge_geom_dpos(vec, dsh, dlen, g, lmax, n):
    st = _stream()
    dvec = torch.empty_like(vec)
    call("eqf_edge_geom_bwd", _p(vec), _p(dsh), _p(dlen), g.E, lmax, _p(dvec), st)
    # d pos[n] = sum_{src(e)=n} dvec[e] - sum_{dst(e)=n} dvec[e]   (segmented, no atomics)
    dpos = torch.empty((n, 3), device=vec.device, dtype=torch.float32)
    call("eqf_segment_sum", _p(dvec), _p(g.src_ptr), _p(g.src_perm), _p(dpos), n, 3, 1.0, 0, st)
    call("eqf_segment_sum", _p(dvec), _p(g.row_ptr), None, _p(dpos), n, 3, -1.0, 1, st)
    return dpos


class _EdgeGeomBwd(Function):
    """d_pos from (d_len, d_sh) as a differentiable op.  `vec` is the saved edge vector pos[src] - pos[dst] (+ offsets);
    its dependence on `pos` is accounted for here (the backward returns the gradient wrt pos)."""

    @staticmethod
    def forward(ctx, pos, vec, dlen, dsh, graph, lmax):
        dlen = _c(dlen) if dlen is not None else None
        dsh = _c(dsh) if dsh is not None else None
        _chk(dlen, dsh)
        ctx.save_for_backward(vec, dlen, dsh)
        ctx.graph, ctx.lmax, ctx.n = graph, lmax, pos.shape[0]
        return _edge_geom_dpos(vec, dsh, dlen, graph, lmax, pos.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, c_pos):
        vec, dlen, dsh = ctx.saved_tensors
        g, lmax, n = ctx.graph, ctx.lmax, ctx.n
        c_pos = _c(c_pos)
        _chk(c_pos)
        st = _stream()
        E = g.E
        # cotangent of d_vec: c_pos[src] - c_pos[dst]  (the edge-vector kernel applied to c_pos, lmax = 0)
        c_vec = torch.empty((E, 3), device=vec.device, dtype=torch.float32)
        scratch = torch.empty((2 * E,), device=vec.device, dtype=torch.float32)
        call("eqf_edge_geom_fwd", _p(c_pos), _p(g.src), _p(g.dst), None, E, 0, _p(c_vec), _p(scratch), _p(scratch, E), st)
        g_vec = torch.empty_like(vec)
        g_dsh = torch.empty_like(dsh) if dsh is not None else None
        g_dlen = torch.empty_like(dlen) if dlen is not None else None
        call("eqf_edge_geom_bwd2", _p(vec), _p(dsh), _p(dlen), _p(c_vec), E, lmax, _p(g_vec), _p(g_dsh), _p(g_dlen), st)
        g_pos = torch.empty((n, 3), device=vec.device, dtype=torch.float32)
        call("eqf_segment_sum", _p(g_vec), _p(g.src_ptr), _p(g.src_perm), _p(g_pos), n, 3, 1.0, 0, st)
        call("eqf_segment_sum", _p(g_vec), _p(g.row_ptr), None, _p(g_pos), n, 3, -1.0, 1, st)
        return g_pos, None, g_dlen, g_dsh, None, None


def edge_geometry(pos, offsets, graph, lmax):
    """(edge_vec [non-differentiable copy], edge_length, edge_sh)."""
    return _EdgeGeom.apply(pos, offsets, graph, lmax)


def vec_sh(vec, keep, lmax, norm_scale):
    """[N, (lmax+1)^2] spherical harmonics of the rows of `vec`, times |v| * norm_scale, rows with keep == False zeroed
    (no gradient: the vectors are input data)."""
    vec = _c(vec.detach().to(torch.float32))
    _chk(vec)
    if keep is not None:
        keep = keep.to(torch.uint8).contiguous()
        if not keep.is_cuda:
            raise HipOnlyError("mask on %s" % keep.device)
    out = torch.empty((vec.shape[0], (lmax + 1) ** 2), device=vec.device, dtype=torch.float32)
    call("eqf_vec_sh", _p(vec), ctypes.c_void_p(keep.data_ptr()) if keep is not None and keep.numel() else None,
         vec.shape[0], lmax, float(norm_scale), _p(out), _stream())
    return out


class _RbfGaussian(Function):
    @staticmethod
    def forward(ctx, length, mean, std, weight, bias, cutoff):
        _chk(length, mean, std, weight, bias)
        E, R = length.shape[0], mean.numel()
        out = torch.empty((E, R), device=length.device, dtype=torch.float32)
        call("eqf_rbf_gaussian_fwd", _p(length), E, R, _p(mean), _p(std), _p(weight), _p(bias), cutoff, _p(out),
             _stream())
        ctx.save_for_backward(length, mean, std, weight, bias)
        ctx.cutoff = cutoff
        return out

    @staticmethod
    def backward(ctx, dout):
        length, mean, std, weight, bias = ctx.saved_tensors
        if torch.is_grad_enabled() and ctx.needs_input_grad[0]:  # create_graph (forces of a gaussian-basis MD17 model)
            outs = _RbfGaussianBwd.apply(length, mean, std, weight, bias, dout, ctx.cutoff)
            return (outs[0],) + tuple(_guard_opt(t, dout, "radial-basis parameter gradient") for t in outs[1:]) + (None,)
        dout = _c(dout)
        _chk(dout)
        return _rbf_gaussian_bwd(length, mean, std, weight, bias, dout, ctx.cutoff, ctx.needs_input_grad[0]) + (None,)


def _rbf_gaussian_bwd(length, mean, std, weight, bias, dout, cutoff, want_len):
    E, R = length.shape[0], mean.numel()
    dm, ds = _zeros_like(mean), _zeros_like(std)
    dw, db = _zeros_like(weight), _zeros_like(bias)
    dlen = torch.empty_like(length) if want_len else None
    call("eqf_rbf_gaussian_bwd", _p(length), _p(dout), E, R, _p(mean), _p(std), _p(weight), _p(bias), cutoff,
         _p(dm), _p(ds), _p(dw), _p(db), _p(dlen), _stream())
    return dlen, dm, ds, dw, db


class _RbfGaussianBwd(Function):
    """(d_len, first-order parameter gradients) of the Gaussian basis as a differentiable op: only d_len carries a
    cotangent in the force graph; its backward is eqf_rbf_gaussian_bwd2."""

    @staticmethod
    def forward(ctx, length, mean, std, weight, bias, dout, cutoff):
        dout = _c(dout)
        _chk(dout)
        dlen, dm, ds, dw, db = _rbf_gaussian_bwd(length, mean, std, weight, bias, dout, cutoff, True)
        ctx.save_for_backward(length, mean, std, weight, bias, dout)
        ctx.cutoff = cutoff
        if not _want_param_grads():
            return dlen, None, None, None, None
        ctx.mark_non_differentiable(dm, ds, dw, db)
        return dlen, dm, ds, dw, db

    @staticmethod
    @once_differentiable
    def backward(ctx, c_len, *_unused):
        length, mean, std, weight, bias, dout = ctx.saved_tensors
        c_len = _c(c_len)
        _chk(c_len)
        E, R = length.shape[0], mean.numel()
        g_len, g_dout = torch.empty_like(length), torch.empty_like(dout)
        g_m, g_s = _zeros_like(mean), _zeros_like(std)
        g_w, g_b = _zeros_like(weight), _zeros_like(bias)
        call("eqf_rbf_gaussian_bwd2", _p(length), _p(dout), _p(c_len), E, R, _p(mean), _p(std), _p(weight), _p(bias),
             ctx.cutoff, _p(g_len), _p(g_dout), _p(g_m), _p(g_s), _p(g_w), _p(g_b), _stream())
        return g_len, g_m, g_s, g_w, g_b, g_dout, None


def rbf_gaussian(length, mean, std, weight, bias, cutoff):
    return _RbfGaussian.apply(length, mean, std, weight, bias, float(cutoff))


class _RbfExpNorm(Function):
    @staticmethod
    def forward(ctx, length, means, betas, alpha, cutoff):
        _chk(length, means, betas)
        E, R = length.shape[0], means.numel()
        out = torch.empty((E, R), device=length.device, dtype=torch.float32)
        call("eqf_rbf_expnorm_fwd", _p(length), E, R, _p(means), _p(betas), alpha, cutoff, _p(out), _stream())
        ctx.save_for_backward(length, means, betas)
        ctx.args = (alpha, cutoff)
        return out

    @staticmethod
    def backward(ctx, dout):
        length, means, betas = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        if torch.is_grad_enabled():  # create_graph
            return _RbfExpNormBwd.apply(length, dout, means, betas, ctx.args[0], ctx.args[1]), None, None, None, None
        dout = _c(dout)
        _chk(dout)
        dlen = torch.empty_like(length)
        call("eqf_rbf_expnorm_bwd", _p(length), _p(dout), length.shape[0], means.numel(), _p(means), _p(betas),
             ctx.args[0], ctx.args[1], _p(dlen), _stream())
        return dlen, None, None, None, None


class _RbfExpNormBwd(Function):
    @staticmethod
    def forward(ctx, length, dout, means, betas, alpha, cutoff):
        dout = _c(dout)
        _chk(dout)
        dlen = torch.empty_like(length)
        call("eqf_rbf_expnorm_bwd", _p(length), _p(dout), length.shape[0], means.numel(), _p(means), _p(betas), alpha,
             cutoff, _p(dlen), _stream())
        ctx.save_for_backward(length, dout, means, betas)
        ctx.args = (alpha, cutoff)
        return dlen

    @staticmethod
    @once_differentiable
    def backward(ctx, c_len):
        length, dout, means, betas = ctx.saved_tensors
        c_len = _c(c_len)
        _chk(c_len)
        g_len, g_dout = torch.empty_like(length), torch.empty_like(dout)
        call("eqf_rbf_expnorm_bwd2", _p(length), _p(dout), _p(c_len), length.shape[0], means.numel(), _p(means),
             _p(betas), ctx.args[0], ctx.args[1], _p(g_len), _p(g_dout), _stream())
        return g_len, g_dout, None, None, None, None


def rbf_expnorm(length, means, betas, alpha, cutoff):
    return _RbfExpNorm.apply(length, means, betas, float(alpha), float(cutoff))


class _RbfBesselBwd(Function):
    @staticmethod
    def forward(ctx, length, freq, dout, cutoff):
        dout = _c(dout)
        _chk(dout)
        dlen = torch.empty_like(length)
        dfreq = _zeros_like(freq)
        call("eqf_rbf_bessel_bwd", _p(length), _p(dout), length.shape[0], freq.numel(), _p(freq), cutoff, _p(dfreq),
             _p(dlen), _stream())
        ctx.save_for_backward(length, freq, dout)
        ctx.cutoff = cutoff
        if not _want_param_grads():
            return dlen, None
        ctx.mark_non_differentiable(dfreq)
        return dlen, dfreq

    @staticmethod
    @once_differentiable
    def backward(ctx, c_len, _cf=None):
        length, freq, dout = ctx.saved_tensors
        c_len = _c(c_len)
        _chk(c_len)
        g_len, g_dout = torch.empty_like(length), torch.empty_like(dout)
        g_freq = _zeros_like(freq)
        call("eqf_rbf_bessel_bwd2", _p(length), _p(dout), _p(c_len), length.shape[0], freq.numel(), _p(freq), ctx.cutoff,
             _p(g_len), _p(g_dout), _p(g_freq), _stream())
        return g_len, g_freq, g_dout, None


class _RbfBessel(Function):
    @staticmethod
    def forward(ctx, length, freq, cutoff):
        _chk(length, freq)
        E, R = length.shape[0], freq.numel()
        out = torch.empty((E, R), device=length.device, dtype=torch.float32)
        call("eqf_rbf_bessel_fwd", _p(length), E, R, _p(freq), cutoff, _p(out), _stream())
        ctx.save_for_backward(length, freq)
        ctx.cutoff = cutoff
        return out

    @staticmethod
    def backward(ctx, dout):
        length, freq = ctx.saved_tensors
        if torch.is_grad_enabled() and ctx.needs_input_grad[0]:  # create_graph
            dlen, dfreq = _RbfBesselBwd.apply(length, freq, dout, ctx.cutoff)
            return dlen, _guard_opt(dfreq, dout, "Bessel frequency gradient"), None
        dout = _c(dout)
        _chk(dout)
        dlen = torch.empty_like(length) if ctx.needs_input_grad[0] else None
        dfreq = _zeros_like(freq) if (ctx.needs_input_grad[1] and _want_param_grads()) else None
        call("eqf_rbf_bessel_bwd", _p(length), _p(dout), length.shape[0], freq.numel(), _p(freq), ctx.cutoff, _p(dfreq),
             _p(dlen), _stream())
        return dlen, dfreq, None


def rbf_bessel(length, freq, cutoff):
    """Spherical Bessel basis with polynomial envelope (ocpmodels RadialBasis, 'spherical_bessel')."""
    return _RbfBessel.apply(length, _c(freq), float(cutoff))


# ------------------------------------------------------------------------------------------------- DTP
def _coupling_fwd(sh, table):
    E = sh.shape[0]
    M = torch.empty((E, table.m_numel), device=sh.device, dtype=torch.float32)
    call("eqf_dtp_coupling_fwd", _p(sh), _p(table.cg(sh.device)), table.c_ref, _p(M), E, _stream())
    return M


def _coupling_bwd(dM, table, shape):
    dsh = torch.empty(shape, device=dM.device, dtype=torch.float32)
    call("eqf_dtp_coupling_bwd", _p(dM), _p(table.cg(dM.device)), table.c_ref, _p(dsh), shape[0], _stream())
    return dsh


class _CouplingBwd(Function):
    """d_sh from d_coupling as a differentiable (linear) op; create_graph only."""

    @staticmethod
    def forward(ctx, dM, table, shape):
        dM = _c(dM)
        _chk(dM)
        ctx.table = table
        return _coupling_bwd(dM, table, shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, c):
        c = _c(c)
        _chk(c)
        return _coupling_fwd(c, ctx.table), None, None


class _Coupling(Function):
    @staticmethod
    def forward(ctx, sh, table):
        sh = _c(sh)
        _chk(sh)
        M = _coupling_fwd(sh, table)
        ctx.table, ctx.shape = table, sh.shape
        return M

    @staticmethod
    def backward(ctx, dM):
        if torch.is_grad_enabled():  # create_graph
            return _CouplingBwd.apply(dM, ctx.table, ctx.shape), None
        dM = _c(dM)
        _chk(dM)
        return _coupling_bwd(dM, ctx.table, ctx.shape), None


def dtp_coupling(sh, table):
    return _Coupling.apply(sh, table)


class _Dtp(Function):
    """Un-fused depth-wise tensor product (materialises the [E, out_dim] result)."""

    @staticmethod
    def forward(ctx, x, coupling, w, table):
        x, coupling = _c(x), _c(coupling)
        w = _c(w) if w is not None else None
        _chk(x, coupling, w)
        E = x.shape[0]
        out = torch.empty((E, table.layout_out.dim), device=x.device, dtype=torch.float32)
        call("eqf_dtp_fwd", _p(x), _p(coupling), _p(w), table.c_ref, _p(out), E, _stream())
        ctx.save_for_backward(x, coupling, w)
        ctx.table = table
        return out

    @staticmethod
    def backward(ctx, dout):
        x, coupling, w = ctx.saved_tensors
        if torch.is_grad_enabled():  # create_graph (forces of the E(3) models, which run the un-fused product)
            dx, dM, dw = _DtpBwd.apply(x, coupling, w, dout, ctx.table)
            return dx, dM, dw, None
        dout = _c(dout)
        _chk(dout)
        dx, dM, dw = _dtp_bwd(x, coupling, w, dout, ctx.table, ctx.needs_input_grad[1],
                              w is not None and ctx.needs_input_grad[2])
        return dx, dM, dw, None


def _dtp_fwd(x, coupling, w, table):
    out = torch.empty((x.shape[0], table.layout_out.dim), device=x.device, dtype=torch.float32)
    call("eqf_dtp_fwd", _p(x), _p(coupling), _p(w), table.c_ref, _p(out), x.shape[0], _stream())
    return out


def _dtp_bwd(x, coupling, w, dout, table, want_M=True, want_w=True, want_x=True):
    dx = (torch.empty_like(x) if table.in_covered else _zeros_like(x))
    dw = torch.empty_like(w) if (w is not None and want_w) else None
    dM = torch.empty_like(coupling) if want_M else None
    call("eqf_dtp_bwd", _p(x), _p(coupling), _p(w), table.c_ref, _p(dout), _p(dx), _p(dw), _p(dM), x.shape[0], _stream())
    return dx, dM, dw


class _DtpBwd(Function):
    """(dx, dM, dw) of the un-fused depth-wise tensor product as a differentiable op of (x, M, w, dout).  The product
    T(x, M, w) is trilinear, so with cotangents (cx, cM, cw) of the three outputs
        Phi = <dout, T(cx, M, w)> + <dout, T(x, cM, w)> + <dout, T(x, M, cw)>
    and every term of its gradient is T itself or one of its first-order backward maps with one argument swapped."""

    @staticmethod
    def forward(ctx, x, coupling, w, dout, table):
        dout = _c(dout)
        _chk(dout)
        ctx.save_for_backward(x, coupling, w, dout)
        ctx.table = table
        dx, dM, dw = _dtp_bwd(x, coupling, w, dout, table)
        return dx, dM, dw

    @staticmethod
    @once_differentiable
    def backward(ctx, cx, cM, cw):
        x, M, w, dout = ctx.saved_tensors
        t = ctx.table
        g_x = g_M = g_w = g_d = None

        def acc(a, b):
            return b if a is None else (a if b is None else a + b)
        if cx is not None:
            cx = _c(cx)
            _chk(cx)
            g_d = acc(g_d, _dtp_fwd(cx, M, w, t))
            _, m_, w_ = _dtp_bwd(cx, M, w, dout, t)
            g_M, g_w = acc(g_M, m_), acc(g_w, w_)
        if cM is not None:
            cM = _c(cM)
            _chk(cM)
            g_d = acc(g_d, _dtp_fwd(x, cM, w, t))
            x_, _, w_ = _dtp_bwd(x, cM, w, dout, t, want_M=False)
            g_x, g_w = acc(g_x, x_), acc(g_w, w_)
        if cw is not None and w is not None:
            cw = _c(cw)
            _chk(cw)
            g_d = acc(g_d, _dtp_fwd(x, M, cw, t))
            x_, m_, _ = _dtp_bwd(x, M, cw, dout, t, want_w=False)
            g_x, g_M = acc(g_x, x_), acc(g_M, m_)
        return g_x, g_M, g_w, g_d, None


def dtp(x, coupling, w, table):
    return _Dtp.apply(x, coupling, w, table)


class DtpLinearSpec:
    """DTP output (degree l3, K(l3) channels) -> per-degree linear to `out_layout` (N(l3) channels).
    Flat weight = [K(l3), N(l3)] blocks in ascending degree (e3nn LinearRS order on the simplified DTP irreps)."""

    def __init__(self, table, out_layout):
        self.table, self.out_layout = table, out_layout
        if table.has_odd or out_layout.has_odd:
            raise NotImplementedError("the DTP-generating GEMMs index their tables by degree: SE(3) irreps only")
        self.blocks = []  # (l3, K, N, w_off, mid_off, out_off)
        w_off = 0
        for (K, l3), mid_off in zip(table.layout_out.segs, table.layout_out.offsets):
            j = out_layout.seg_index(l3)
            if j is None:
                continue
            N = out_layout.segs[j][0]
            self.blocks.append((l3, K, N, w_off, mid_off, out_layout.offsets[j]))
            w_off += K * N
        self.weight_numel = w_off
        if len(self.blocks) != len(out_layout.segs):
            raise NotImplementedError("every output degree of a fused DTP-linear must be fed by the DTP")
        self.bias_dim = out_layout.mul_of(0)


class _DtpLinear(Function):
    """out = Linear(DTP(x, sh, w)) with the DTP result generated inside the GEMM (never stored)."""

    @staticmethod
    def forward(ctx, x, coupling, w, weight, bias, spec):
        x, coupling, weight = _c(x), _c(coupling), _c(weight)
        w = _c(w) if w is not None else None
        _chk(x, coupling, w, weight, bias)
        E = x.shape[0]
        assert weight.numel() == spec.weight_numel
        out = torch.empty((E, spec.out_layout.dim), device=x.device, dtype=torch.float32)
        Wl = (ctypes.c_void_p * 8)()
        for (l3, K, N, w_off, _, _) in spec.blocks:
            Wl[l3] = weight.data_ptr() + 4 * w_off
        call("eqf_dtp_linear_fwd", _p(x), _p(coupling), _p(w), spec.table.c_ref, Wl, _p(bias), _p(out),
             spec.out_layout.c_ref, E, _stream())
        ctx.save_for_backward(x, coupling, w, weight)
        ctx.spec = spec
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, coupling, w, weight = ctx.saved_tensors
        spec = ctx.spec
        table = spec.table
        dout = _c(dout)
        _chk(dout)
        E = x.shape[0]
        st = _stream()
        Dout, Dmid = spec.out_layout.dim, table.layout_out.dim
        dx = dM = dw = dweight = dbias = None
        need_mid = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or (w is not None and ctx.needs_input_grad[2])
        if need_mid:
            covered = len(spec.blocks) == len(table.layout_out.segs)
            dmid = (torch.empty if covered else torch.zeros)((E, Dmid), device=x.device, dtype=torch.float32)
            for (l3, K, N, w_off, mid_off, out_off) in spec.blocks:
                d = 2 * l3 + 1
                call("eqf_gemm_nt", _p(dout, out_off), rows(d, Dout, N), _p(weight, w_off), N, _p(dmid, mid_off),
                     rows(d, Dmid, K), None, E * d, K, N, 0, st)
            dx = torch.empty_like(x) if table.in_covered else _zeros_like(x)
            dw = torch.empty_like(w) if (w is not None and ctx.needs_input_grad[2]) else None
            dM = torch.empty_like(coupling) if ctx.needs_input_grad[1] else None
            call("eqf_dtp_bwd", _p(x), _p(coupling), _p(w), table.c_ref, _p(dmid), _p(dx), _p(dw), _p(dM), E, st)
            del dmid
        want_b = ctx.has_bias and ctx.needs_input_grad[4]
        if ctx.needs_input_grad[3] or want_b:
            dweight_, dbias_ = _zeros2(weight.numel(), spec.bias_dim if want_b else 0, x.device)
        if ctx.needs_input_grad[3]:
            dweight = dweight_
            dWl = (ctypes.c_void_p * 8)()
            for (l3, K, N, w_off, _, _) in spec.blocks:
                dWl[l3] = dweight.data_ptr() + 4 * w_off
            call("eqf_dtp_linear_wgrad", _p(x), _p(coupling), _p(w), table.c_ref, _p(dout), spec.out_layout.c_ref, dWl,
                 E, st)
        if want_b:
            dbias = dbias_
            j = spec.out_layout.seg_index(0)
            call("eqf_colsum", _p(dout, spec.out_layout.offsets[j]), rows(1, Dout, 0), E, spec.bias_dim, _p(dbias), st)
        return dx, dM, dw, dweight, dbias, None


def dtp_linear(x, coupling, w, weight, bias, spec):
    return _DtpLinear.apply(x, coupling, w, weight, bias, spec)


# ------------------------------------------------------------------------------------------------- fused SeparableFCTP
class SfcSpec:
    """Fused DTP -> per-degree linear(s) (eqf_sfc_*): `out_layout` = irreps of the main consumer (one [K(l), N1(l)]
    weight per degree), `n2` = width of an optional second scalar consumer fed by the degree-0 DTP output (its weight
    is concatenated to the degree-0 matrix: [K(0), N1(0)+n2])."""

    def __init__(self, table, out_layout, n2=0):
        self.table, self.out_layout, self.n2 = table, out_layout, int(n2)
        self.degs = []  # (l3, K, N1, Ncat)
        self.w_offs = []  # offset of the [K, N1] block of each degree in the flat main weight
        if table.has_odd or out_layout.has_odd:  # E(3) irreps: the un-fused tensor product + linear serve them
            self.supported = False
            return
        ok = table.fusable
        for (N1, l3) in out_layout.segs:
            i = table.layout_out.seg_index(l3)
            if i is None:
                raise NotImplementedError("output degree %d is not produced by the tensor product" % l3)
            K = table.layout_out.segs[i][0]
            ncat = N1 + (self.n2 if l3 == 0 else 0)
            ok = ok and ncat % 32 == 0 and N1 % 32 == 0 and l3 <= 3
            self.degs.append((l3, K, N1, ncat))
            self.w_offs.append(sum(k * n for (_, k, n, _) in self.degs[:-1]))
        if self.n2 and out_layout.seg_index(0) is None:
            ok = False
        # LDS footprint of the forward workgroup (A tile + weight tile + coupling tile of 64 edges), see sfc.hip
        for (l3, _, _, _) in self.degs:
            m_len = sum((2 * p["l1"] + 1) * (2 * l3 + 1) for p in table.paths if p["l3"] == l3)
            rows = 64 * (2 * l3 + 1)
            a_floats = max(32 * (rows + 1), rows * 36)  # [k][row] (fp32 MFMA step) / [row][k] (split-precision step)
            ok = ok and 4 * (a_floats + 32 * 68 + 64 * m_len) <= 160 * 1024
        self.supported = ok and len(self.degs) <= 4
        self.weight_numel = sum(k * n for (_, k, n, _) in self.degs)
        k0 = [k for (l3, k, _, _) in self.degs if l3 == 0]
        self.weight2_numel = (k0[0] * self.n2) if (self.n2 and k0) else 0
        self.bias_dim = out_layout.mul_of(0) + self.n2
        used = {l3 for l3, _, _, _ in self.degs}
        self.in_covered = {p["in_off"] for p in table.paths if p["l3"] in used} == set(table.layout_in.offsets)
        # split-precision kernels (csrc/sfcx.hip): each of the three launches has its own table limits (input slabs, work
        # items, LDS); the planners themselves are asked once per mode (eqf_sfcx_supported, host only) and a launch they
        # reject is served by the exact-fp32 kernel of the same shape
        self._x_mask = {}
        self._packed_numel = {}

    def x_mask(self, mode, E=None):
        """bit 0 forward, bit 1 data gradient, bit 2 weight gradient of csrc/sfcx.hip can serve this operator in `mode`.
        E: rows of the launch -- the split-precision kernels address every per-edge tensor with 32-bit element offsets
        (`fits32` in csrc/sfcx_common.h); a launch whose widest tensor reaches 2^31 elements is served by the exact-fp32
        kernels instead (the planners' verdict is cached for a nominal E, so the size limit is re-checked here)."""
        if E is not None and int(E) * self._widest_row() >= (1 << 31):
            return 0
        m = self._x_mask.get(mode)
        if m is None:
            m = 0
            if self.supported:
                m = lib.load().eqf_sfcx_supported(self.table.c_ref, self.out_layout.c_ref, self.n2, mode)
                if m < 0:
                    raise lib.HipLibraryError("eqf_sfcx_supported failed with code %d" % m)
            self._x_mask[mode] = m
        return m

    def _widest_row(self):
        # x (+ the gate scalars of a folded gate: at most as many again), w, coupling, out1, out2 rows
        return max(2 * self.table.layout_in.dim, self.table.weight_numel, self.table.m_numel, self.out_layout.dim, self.n2, 1)

    @property
    def x_ok(self):
        return self.supported and self.x_mask(0) != 0

    def packed_numel(self, mode):
        n = self._packed_numel.get(mode)
        if n is None:
            n = lib.load().eqf_sfcx_packed_numel(self.table.c_ref, self.out_layout.c_ref, self.n2, mode)
            if n <= 0:
                raise lib.HipLibraryError("eqf_sfcx_packed_numel failed with code %d" % n)
            self._packed_numel[mode] = n
        return n


def _ptr_array(pairs):
    arr = (ctypes.c_void_p * 8)()
    for l3, addr in pairs:
        arr[l3] = addr
    return arr


class _Guard(Function):
    """Identity whose backward raises: marks a first-order result that must not be differentiated again."""

    @staticmethod
    def forward(ctx, t, dep, what):
        ctx.what = what
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("second-order differentiation through the %s is not implemented" % ctx.what)


def _guard(t, dep, what):
    return _Guard.apply(t, dep, what) if dep.requires_grad else t


_side_streams = {}
# weight gradient of the fused SeparableFCTP on a side stream beside its data gradient: measured +1 % on the QM9 step (14.00 vs
# 14.15 ms, profiles/r03) -- the data-gradient kernel holds the whole register file of its SIMDs, so little co-resides -- and
# it blurs the per-kernel HIP-event durations bench.py reports.  Off by default.
_overlap_wgrad = [False]


def _side_stream(dev):
    st = _side_streams.get(dev)
    if st is None:
        st = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return st


def _sfc_mode(spec, E=None):
    """mode code of the split-precision kernels for this operator (and launch size), None = exact-fp32 kernels"""
    m = _MATRIX_MODES[_matrix_mode[0]]
    return m if (m is not None and spec.supported and spec.x_mask(m, E) != 0) else None


def _sfc_Wl(weight, spec):
    return _ptr_array((l3, weight.data_ptr() + 4 * off) for (l3, _, _, _), off in zip(spec.degs, spec.w_offs))


def _sfc_pack(weight, weight2, spec, mode):
    """bf16 planes of the per-degree weights in MFMA fragment order (forward and data-gradient orientation)"""
    packed = torch.empty(spec.packed_numel(mode), device=weight.device, dtype=torch.bfloat16)
    call("eqf_sfcx_pack", _sfc_Wl(weight, spec), _p(weight2), spec.table.c_ref, spec.out_layout.c_ref, spec.n2, mode,
         ctypes.c_void_p(packed.data_ptr()), _stream())
    return packed


def _sfc_fwd(x, coupling, w, weight, bias, weight2, bias2, spec, mode=None, packed=None):
    E = x.shape[0]
    out1 = torch.empty((E, spec.out_layout.dim), device=x.device, dtype=torch.float32)
    out2 = torch.empty((E, spec.n2), device=x.device, dtype=torch.float32) if spec.n2 else None
    if mode is None or not (spec.x_mask(mode) & 1):
        call("eqf_sfc_fwd", _p(x), _p(coupling), _p(w), spec.table.c_ref, _sfc_Wl(weight, spec), _p(bias), _p(weight2),
             _p(bias2), _p(out1), spec.out_layout.c_ref, _p(out2), spec.n2, E, _stream())
    else:
        if packed is None:
            packed = _sfc_pack(weight, weight2, spec, mode)
        call("eqf_sfcx_fwd", _p(x), _p(coupling), _p(w), spec.table.c_ref, ctypes.c_void_p(packed.data_ptr()), _p(bias),
             _p(bias2), _p(out1), spec.out_layout.c_ref, _p(out2), spec.n2, E, mode, _stream())
    return out1, out2


def _sfc_bwd_data(x, coupling, w, weight, weight2, d1, d2, spec, want_dM, mode=None, packed=None, dM_out=None):
    """dM_out: an existing d_coupling tensor the launch ADDS into (the kernels accumulate it with atomics anyway)"""
    E = x.shape[0]
    dx = (torch.empty_like if spec.in_covered else _zeros_like)(x)
    dw = torch.empty_like(w) if w is not None else None
    dM = (dM_out if dM_out is not None else _zeros_like(coupling)) if want_dM else None
    if mode is None or not (spec.x_mask(mode) & 2):
        call("eqf_sfc_bwd_data", _p(x), _p(coupling), _p(w), spec.table.c_ref, _sfc_Wl(weight, spec), _p(weight2), _p(d1),
             spec.out_layout.c_ref, _p(d2), spec.n2, _p(dx), _p(dw), _p(dM), E, _stream())
    else:
        if packed is None:
            packed = _sfc_pack(weight, weight2, spec, mode)
        call("eqf_sfcx_bwd_data", _p(x), _p(coupling), _p(w), spec.table.c_ref, ctypes.c_void_p(packed.data_ptr()), _p(d1),
             spec.out_layout.c_ref, _p(d2), spec.n2, _p(dx), _p(dw), _p(dM), E, mode, _stream())
    return dx, dM, dw


def _sfc_bwd_weight(x, coupling, w, d1, d2, spec, dweight, dweight2, mode=None, dbias=None, dbias2=None):
    """dweight (flat) / dweight2, zero-initialised by the caller, are accumulated into.  dbias / dbias2 (zero-initialised):
    the bias gradients of the degree-0 linears, taken along by the split-precision launch.  Returns True if they were (the
    exact-fp32 kernel does not: the caller then launches eqf_colsum)."""
    dWl = _sfc_Wl(dweight, spec)
    if mode is None or not (spec.x_mask(mode) & 4):
        call("eqf_sfc_bwd_weight", _p(x), _p(coupling), _p(w), spec.table.c_ref, _p(d1), spec.out_layout.c_ref, _p(d2),
             spec.n2, dWl, _p(dweight2), x.shape[0], _stream())
        return False
    if (dbias is not None or dbias2 is not None) and 0 in [l3 for l3, _, _, _ in spec.degs]:
        call("eqf_sfcx_bwd_weight_bias", _p(x), _p(coupling), _p(w), spec.table.c_ref, _p(d1), spec.out_layout.c_ref, _p(d2),
             spec.n2, dWl, _p(dweight2), _p(dbias), _p(dbias2), x.shape[0], mode, _stream())
        return True
    call("eqf_sfcx_bwd_weight", _p(x), _p(coupling), _p(w), spec.table.c_ref, _p(d1), spec.out_layout.c_ref, _p(d2),
         spec.n2, dWl, _p(dweight2), x.shape[0], mode, _stream())
    return False


class _SepFctpBwdData(Function):
    """(dx, d_coupling, dw) of the fused SeparableFCTP as a differentiable op (create_graph only).  Phi(x, M, w, W, g) =
    <g, F(x, M, w, W)> is multilinear, so the pairing with cotangents (cx, cM, cw) is
    Psi = Phi(cx, M, w, W, g) + Phi(x, cM, w, W, g) + Phi(x, M, cw, W, g) and every gradient of Psi is one of the
    first-order kernels (forward, data-gradient, weight-gradient) evaluated with one argument substituted."""

    @staticmethod
    def forward(ctx, x, coupling, w, weight, weight2, d1, d2, spec, mode, packed=None):
        # packed: the bf16 planes of (weight, weight2) the forward of the operator made -- the same weights, so the data
        # gradient here and the nine launches of the double backward reuse them (round 5: 39 -> 13 sfcx_pack launches per
        # MD17 step, which is bound by its launch count)
        x, coupling, weight, d1 = _c(x), _c(coupling), _c(weight), _c(d1)
        w = _c(w) if w is not None else None
        weight2 = _c(weight2) if weight2 is not None else None
        d2 = _c(d2) if d2 is not None else None
        _chk(x, coupling, w, weight, weight2, d1, d2)
        ctx.save_for_backward(x, coupling, w, weight, weight2, d1, d2)
        ctx.spec = spec
        # the arithmetic of the forward this is the gradient of (passed in by _SepFctp.backward), not the global of the moment
        ctx.mode = mode
        ctx.packed = packed if mode is not None else None
        dx, dM, dw = _sfc_bwd_data(x, coupling, w, weight, weight2, d1, d2, spec, True, ctx.mode, ctx.packed)
        if w is None:
            return dx, dM
        return dx, dM, dw

    @staticmethod
    @once_differentiable
    def backward(ctx, cx, cM, cw=None):
        x, M, w, weight, weight2, d1, d2 = ctx.saved_tensors
        spec, mode = ctx.spec, ctx.mode
        packed = ctx.packed
        if packed is None and mode is not None:
            packed = _sfc_pack(weight, weight2, spec, mode)
        need = ctx.needs_input_grad  # x, M, w, weight, weight2, d1, d2
        g_x = g_M = g_w = g_W = g_W2 = g_d1 = g_d2 = None

        def acc(a, b):
            if b is None:
                return a
            return b if a is None else a + b

        if need[3] or (weight2 is not None and need[4]):
            g_W = _zeros_like(weight)
            g_W2 = _zeros_like(weight2) if weight2 is not None else None
        subs = []
        if cx is not None:
            subs.append((_c(cx), M, w, "x"))
        if cM is not None:
            subs.append((x, _c(cM), w, "M"))
        if cw is not None and w is not None:
            subs.append((x, M, _c(cw), "w"))
        for (xs, Ms, ws, which) in subs:
            _chk(xs, Ms, ws)
            if need[5] or need[6]:
                o1, o2 = _sfc_fwd(xs, Ms, ws, weight, None, weight2, None, spec, mode, packed)
                g_d1, g_d2 = acc(g_d1, o1), acc(g_d2, o2)
            if need[0] or need[1] or need[2]:
                want_M = which != "M" and need[1]
                if want_M and g_M is None:
                    g_M = _zeros_like(M)  # the kernels ADD their d_coupling: the launches accumulate in place, no torch add
                dx_, _, dw_ = _sfc_bwd_data(xs, Ms, ws, weight, weight2, d1, d2, spec, want_M, mode, packed,
                                            dM_out=g_M if want_M else None)
                if which != "x" and need[0]:
                    g_x = acc(g_x, dx_)
                if which != "w" and w is not None and need[2]:
                    g_w = acc(g_w, dw_)
            if g_W is not None:
                _sfc_bwd_weight(xs, Ms, ws, d1, d2, spec, g_W, g_W2, mode)
        return g_x, g_M, g_w, g_W, g_W2, g_d1, g_d2, None, None, None


class _SepFctp(Function):
    """weight: flat [sum_l K(l) N1(l)] (e3nn LinearRS layout: one [K(l), N1(l)] block per degree, ascending);
    weight2: flat [K(0) n2] or None.  The kernels read the blocks in place (no slicing / concatenation on the host)."""

    @staticmethod
    def forward(ctx, x, coupling, w, weight, bias, weight2, bias2, spec):
        x, coupling, weight = _c(x), _c(coupling), _c(weight)
        w = _c(w) if w is not None else None
        weight2 = _c(weight2) if weight2 is not None else None
        _chk(x, coupling, w, weight, bias, weight2, bias2)
        assert weight.numel() == spec.weight_numel and (weight2 is None) == (spec.n2 == 0)
        assert weight2 is None or weight2.numel() == spec.weight2_numel
        ctx.mode = mode = _sfc_mode(spec, x.shape[0])
        # the planes are saved for the data gradient (a raw attribute: the tensor is not part of the autograd graph; like
        # any saved tensor they describe the weights AT FORWARD TIME -- an in-place weight update between forward and backward,
        # which FlatAdamW's raw-pointer writes would not even trip autograd's version check on, is not supported)
        ctx.packed = _sfc_pack(weight, weight2, spec, mode) if mode is not None else None
        out1, out2 = _sfc_fwd(x, coupling, w, weight, bias, weight2, bias2, spec, mode, ctx.packed)
        ctx.save_for_backward(x, coupling, w, weight, weight2)
        ctx.spec = spec
        ctx.has_bias = (bias is not None, bias2 is not None)
        if out2 is None:
            return out1
        return out1, out2

    @staticmethod
    def backward(ctx, d1, d2=None):
        x, coupling, w, weight, weight2 = ctx.saved_tensors
        spec = ctx.spec
        E = x.shape[0]
        st = _stream()
        dev = x.device
        if torch.is_grad_enabled():  # create_graph: differentiable data-gradient (forces); see _SepFctpBwdData
            note_create_graph()
            need = ctx.needs_input_grad
            if d1 is None:
                d1 = _zeros((E, spec.out_layout.dim), device=dev, dtype=torch.float32)
            if spec.n2 and d2 is None:
                d2 = _zeros((E, spec.n2), device=dev, dtype=torch.float32)
            outs = _SepFctpBwdData.apply(x, coupling, w, weight, weight2, d1, d2 if spec.n2 else None, spec, ctx.mode,
                                         ctx.packed)
            dx, dM = outs[0], outs[1]
            dw = outs[2] if w is not None else None
            # needs_input_grad is static (the parameters always "need" a gradient), so the weight / bias gradients
            # are produced here as well -- first-order kernels, guarded: differentiating THROUGH them is not implemented
            gW = gb = gW2 = gb2 = None
            if not _want_param_grads():
                return dx, dM, dw, None, None, None, None, None
            if need[3] or need[5]:
                with torch.no_grad():
                    gW_ = _zeros_like(weight)
                    gW2_ = _zeros_like(weight2) if weight2 is not None else None
                    _sfc_bwd_weight(x, coupling, w, _c(d1), _c(d2) if spec.n2 else None, spec, gW_, gW2_, ctx.mode)
                gW = _guard(gW_, d1, "weight gradient of the fused SeparableFCTP")
                gW2 = _guard(gW2_, d1, "weight gradient of the fused SeparableFCTP") if gW2_ is not None else None
            if ctx.has_bias[0] and need[4]:
                j = spec.out_layout.seg_index(0)
                o = spec.out_layout.offsets[j]
                gb = d1[:, o:o + spec.out_layout.mul_of(0)].sum(0)
            if ctx.has_bias[1] and need[6]:
                gb2 = d2.sum(0)
            return dx, dM, dw, gW, gb, gW2, gb2, None
        if d1 is None:
            d1 = _zeros((E, spec.out_layout.dim), device=dev, dtype=torch.float32)
        if spec.n2 and d2 is None:
            d2 = _zeros((E, spec.n2), device=dev, dtype=torch.float32)
        d1 = _c(d1)
        d2 = _c(d2) if spec.n2 else None
        _chk(d1, d2)
        need = ctx.needs_input_grad
        dx = dM = dw = dweight = dbias = dweight2 = dbias2 = None
        want_b = ctx.has_bias[0] and need[4]
        want_b2 = ctx.has_bias[1] and need[6]
        want_w = _want_param_grads() and (need[3] or need[5] or want_b or want_b2)
        flat = None
        if want_w:
            n1_0 = spec.out_layout.mul_of(0)
            sizes = [spec.weight_numel, spec.weight2_numel, n1_0 if want_b else 0, spec.n2 if want_b2 else 0]
            flat = _zeros(sum(sizes), device=dev, dtype=torch.float32)  # ONE fill for every accumulated gradient
            o1, o2, o3 = sizes[0], sizes[0] + sizes[1], sizes[0] + sizes[1] + sizes[2]
            dweight = flat[:o1]
            dweight2 = flat[o1:o2] if spec.n2 else None
        # The weight gradient and the data gradient read the same tensors and write disjoint ones: the weight gradient goes
        # out on a side stream and runs beside the data gradient (both kernels leave most of a SIMD's issue slots idle --
        # their waves wait on memory > 50 % of the time, profiles/r03 -- so the two overlap instead of queueing).
        side = _side_stream(dev) if (want_w and (need[3] or need[5]) and _overlap_wgrad[0]) else None
        if side is not None:
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                _sfc_bwd_weight(x, coupling, w, d1, d2, spec, dweight, dweight2, ctx.mode)
        if need[0] or need[1] or (w is not None and need[2]):
            dx, dM, dw = _sfc_bwd_data(x, coupling, w, weight, weight2, d1, d2, spec, need[1], ctx.mode, ctx.packed)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        if not _want_param_grads():  # force evaluation
            return dx, dM, dw, None, None, None, None, None
        if want_w:
            dbias = flat[o2:o3] if want_b else None
            dbias2 = flat[o3:] if want_b2 else None
            bias_done = False
            if side is None and (need[3] or need[5]):
                # (the bias gradients ride along in the weight-gradient launch: its degree-0 items stream those columns anyway)
                bias_done = _sfc_bwd_weight(x, coupling, w, d1, d2, spec, dweight, dweight2, ctx.mode, dbias, dbias2)
            if want_b and not bias_done:
                j = spec.out_layout.seg_index(0)
                call("eqf_colsum", _p(d1, spec.out_layout.offsets[j]), rows(1, spec.out_layout.dim, 0), E, n1_0,
                     _p(dbias), st)
            if want_b2 and not bias_done:
                call("eqf_colsum", _p(d2), rows(1, spec.n2, 0), E, spec.n2, _p(dbias2), st)
        return dx, dM, dw, dweight, dbias, dweight2, dbias2, None


class _SepFctpGated(Function):
    """Gate -> fused SeparableFCTP with the gate folded into the three split-precision kernels (eqf_sfcx_*_gated): `x_raw` is
    the gate's INPUT [scalars (S) | gates (G) | gated segments]; the gated rows [E, S + dim(gated)] are never written and
    the data gradient returns d x_raw (the gate's backward included).  GraphAttention: value = sep_value(sep_act.gate(...))
    [ref: nets/graph_attention_transformer.py:494-496; Gate: nets/fast_activation.py:132-148].  No second consumer.
    Under create_graph the backward is composed from the separate differentiable operators (gate, _SepFctpBwdData)."""

    @staticmethod
    def forward(ctx, x_raw, coupling, w, weight, bias, spec, gate):
        x_raw, coupling, weight = _c(x_raw), _c(coupling), _c(weight)
        w = _c(w) if w is not None else None
        _chk(x_raw, coupling, w, weight, bias)
        S, gated_layout, c_silu, c_sig = gate
        G = sum(m for m, _ in gated_layout.segs)
        assert spec.n2 == 0 and x_raw.shape[1] == S + G + gated_layout.dim and weight.numel() == spec.weight_numel
        mode = _sfc_mode(spec, x_raw.shape[0])
        assert mode is not None and spec.x_mask(mode, x_raw.shape[0]) == 7
        ctx.mode, ctx.gate = mode, gate
        ctx.gin = lib.EqfGateIn(int(S), int(G), float(c_silu), float(c_sig))
        ctx.packed = _sfc_pack(weight, None, spec, mode)
        E = x_raw.shape[0]
        out = torch.empty((E, spec.out_layout.dim), device=x_raw.device, dtype=torch.float32)
        call("eqf_sfcx_fwd_gated", _p(x_raw), ctypes.byref(ctx.gin), _p(coupling), _p(w), spec.table.c_ref,
             ctypes.c_void_p(ctx.packed.data_ptr()), _p(bias), _p(out), spec.out_layout.c_ref, E, mode, _stream())
        ctx.save_for_backward(x_raw, coupling, w, weight)
        ctx.spec = spec
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, d1):
        x_raw, coupling, w, weight = ctx.saved_tensors
        spec, mode = ctx.spec, ctx.mode
        S, gated_layout, c_silu, c_sig = ctx.gate
        need = ctx.needs_input_grad  # x_raw, coupling, w, weight, bias
        E = x_raw.shape[0]
        dev = x_raw.device
        if torch.is_grad_enabled():  # create_graph: the separate differentiable operators, on the re-materialised gate output
            note_create_graph()
            xg = gate(x_raw, S, gated_layout, c_silu, c_sig)
            outs = _SepFctpBwdData.apply(xg, coupling, w, weight, None, d1, None, spec, mode, ctx.packed)
            dxg, dM = outs[0], outs[1]
            dw = outs[2] if w is not None else None
            dx_raw = _GateBwd.apply(x_raw, dxg, S, gated_layout, c_silu, c_sig) if need[0] else None
            gW = gb = None
            if not _want_param_grads():
                return dx_raw, dM, dw, None, None, None, None
            if need[3]:
                with torch.no_grad():
                    gW_ = _zeros_like(weight)
                    _sfc_bwd_weight(xg.detach(), coupling, w, _c(d1), None, spec, gW_, None, mode)
                gW = _guard(gW_, d1, "weight gradient of the fused SeparableFCTP")
            if ctx.has_bias and need[4]:
                o = spec.out_layout.offsets[spec.out_layout.seg_index(0)]
                gb = d1[:, o:o + spec.out_layout.mul_of(0)].sum(0)
            return dx_raw, dM, dw, gW, gb, None, None
        d1 = _c(d1)
        _chk(d1)
        st = _stream()
        dx_raw = dM = dw = dweight = dbias = None
        if need[0] or need[1] or (w is not None and need[2]):
            dx_raw = torch.empty_like(x_raw)
            dw = torch.empty_like(w) if w is not None else None
            dM = _zeros_like(coupling) if need[1] else None
            call("eqf_sfcx_bwd_data_gated", _p(x_raw), ctypes.byref(ctx.gin), _p(coupling), _p(w), spec.table.c_ref,
                 ctypes.c_void_p(ctx.packed.data_ptr()), _p(d1), spec.out_layout.c_ref, _p(dx_raw), _p(dw), _p(dM), E, mode, st)
        if not _want_param_grads():
            return dx_raw, dM, dw, None, None, None, None
        want_b = ctx.has_bias and need[4]
        if need[3] or want_b:
            n1_0 = spec.out_layout.mul_of(0)
            flat = _zeros(spec.weight_numel + (n1_0 if want_b else 0), device=dev, dtype=torch.float32)
            dweight = flat[:spec.weight_numel]
            dbias = flat[spec.weight_numel:] if want_b else None
            if need[3]:
                call("eqf_sfcx_bwd_weight_gated_bias", _p(x_raw), ctypes.byref(ctx.gin), _p(coupling), _p(w), spec.table.c_ref,
                     _p(d1), spec.out_layout.c_ref, _sfc_Wl(dweight, spec), _p(dbias), E, mode, st)
            elif want_b:
                o = spec.out_layout.offsets[spec.out_layout.seg_index(0)]
                call("eqf_colsum", _p(d1, o), rows(1, spec.out_layout.dim, 0), E, n1_0, _p(dbias), st)
        return dx_raw, dM, dw, (dweight if need[3] else None), dbias, None, None


_fuse_gate = [os.environ.get("EQF_NO_GATE_FUSION", "") == ""]  # A/B switch: False = separate gate kernels in front of sep_value


def sep_fctp_gated_ok(spec, x_raw_dim, S, gated_layout, E=None):
    """the gate can be folded into this operator's kernels: split-precision kernels for all three launches, no second consumer,
    degrees <= 2, scalar segment = the first S channels"""
    if not _fuse_gate[0] or spec.n2 != 0 or not spec.supported:
        return False
    mode = _sfc_mode(spec, E)
    if mode is None or spec.x_mask(mode, E) != 7:
        return False
    if max([p["l1"] for p in spec.table.paths] + [l3 for l3, _, _, _ in spec.degs]) > 2:
        return False
    li = spec.table.layout_in
    return li.segs[0] == (S, 0) and S % 32 == 0 and li.dim == S + gated_layout.dim


def sep_fctp_gated(x_raw, coupling, w, weight, bias, spec, gate):
    """gate = (S, gated_layout, c_silu, c_sig)"""
    return _SepFctpGated.apply(x_raw, coupling, w, weight, bias, spec, gate)


def sep_fctp(x, coupling, w, weight, bias, spec, weight2=None, bias2=None):
    """Fused DTP -> linear(s).  weight: flat LinearRS weight of the main consumer; weight2 / bias2: the second scalar
    consumer (spec.n2 > 0).  Returns out1 (and out2 when spec.n2 > 0)."""
    return _SepFctp.apply(x, coupling, w, weight, bias, weight2, bias2, spec)


# ------------------------------------------------------------------------------------------------- attention
class _AlphaLogits(Function):
    @staticmethod
    def forward(ctx, a, alpha_dot, H, Kh, c):
        a, alpha_dot = _c(a), _c(alpha_dot)
        _chk(a, alpha_dot)
        E = a.shape[0]
        logit = torch.empty((E, H), device=a.device, dtype=torch.float32)
        call("eqf_alpha_fwd", _p(a), _p(alpha_dot), _p(logit), E, H, Kh, c, _stream())
        ctx.save_for_backward(a, alpha_dot)
        ctx.args = (H, Kh, c)
        return logit

    @staticmethod
    def backward(ctx, dlogit):
        a, alpha_dot = ctx.saved_tensors
        H, Kh, c = ctx.args
        if torch.is_grad_enabled():  # create_graph
            da, dd = _AlphaLogitsBwd.apply(a, alpha_dot, dlogit, H, Kh, c)
            return da, _guard_opt(dd, dlogit, "alpha_dot gradient"), None, None, None
        dlogit = _c(dlogit)
        _chk(dlogit)
        da = torch.empty_like(a)
        dd = _zeros_like(alpha_dot)
        call("eqf_alpha_bwd", _p(a), _p(alpha_dot), _p(dlogit), _p(da), _p(dd), a.shape[0], H, Kh, c, _stream())
        return da, dd, None, None, None


class _AlphaLogitsBwd(Function):
    @staticmethod
    def forward(ctx, a, alpha_dot, dlogit, H, Kh, c):
        dlogit = _c(dlogit)
        _chk(dlogit)
        da = torch.empty_like(a)
        dd = _zeros_like(alpha_dot)
        call("eqf_alpha_bwd", _p(a), _p(alpha_dot), _p(dlogit), _p(da), _p(dd), a.shape[0], H, Kh, c, _stream())
        ctx.save_for_backward(a, alpha_dot, dlogit)
        ctx.args = (H, Kh, c)
        if not _want_param_grads():
            return da, None
        ctx.mark_non_differentiable(dd)
        return da, dd

    @staticmethod
    @once_differentiable
    def backward(ctx, ca, _cd=None):
        a, alpha_dot, dlogit = ctx.saved_tensors
        H, Kh, c = ctx.args
        ca = _c(ca)
        _chk(ca)
        g_a, g_dl = torch.empty_like(a), torch.empty_like(dlogit)
        g_ad = _zeros_like(alpha_dot)
        call("eqf_alpha_bwd2", _p(a), _p(alpha_dot), _p(dlogit), _p(ca), _p(g_a), _p(g_ad), _p(g_dl), a.shape[0], H, Kh, c,
             _stream())
        return g_a, g_ad, g_dl, None, None, None


def alpha_logits(a, alpha_dot, H, Kh, c):
    return _AlphaLogits.apply(a, alpha_dot, H, Kh, float(c))


# Dropout under HIP-graph capture (equiformer_amd/capture.py): the by-value seed of a captured launch is frozen, so while a step
# is being captured the attention kernels take seed + *offset with `offset` a device word the captured step advances between
# replays (eqf_attn_aggregate_*_dseed).  None outside a capture.
_seed_offset = [None]


@contextlib.contextmanager
def dropout_seed_offset(t):
    """t: one-element int64 CUDA tensor (or None)."""
    prev = _seed_offset[0]
    _seed_offset[0] = t
    try:
        yield
    finally:
        _seed_offset[0] = prev


class _AttnAggregate(Function):
    @staticmethod
    def forward(ctx, logit, value, graph, H, layout, drop_p, seed):
        logit, value = _c(logit), _c(value)
        _chk(logit, value)
        N = graph.N
        alpha = torch.empty_like(logit)
        out = torch.empty((N, layout.dim), device=value.device, dtype=torch.float32)
        off = _seed_offset[0] if drop_p > 0.0 else None
        if off is None:
            call("eqf_attn_aggregate_fwd", _p(logit), _p(value), _p(graph.row_ptr), _p(alpha), _p(out), N, H, layout.c_ref,
                 drop_p, seed, _stream())
        else:
            call("eqf_attn_aggregate_fwd_dseed", _p(logit), _p(value), _p(graph.row_ptr), _p(alpha), _p(out), N, H,
                 layout.c_ref, drop_p, seed, _p(off), _stream())
        ctx.save_for_backward(alpha, value, logit)
        ctx.args = (graph, H, layout, drop_p, seed)
        ctx.seed_off = off
        return out

    @staticmethod
    def backward(ctx, dout):
        alpha, value, logit = ctx.saved_tensors
        graph, H, layout, drop_p, seed = ctx.args
        if torch.is_grad_enabled():  # create_graph (the dropout mask is replayed from the seed)
            if ctx.seed_off is not None:
                raise NotImplementedError("second-order backward of a captured step with attention dropout")
            dlogit, dvalue = _AttnAggregateBwd.apply(logit, value, dout, alpha, graph, H, layout, drop_p, seed)
            return dlogit, dvalue, None, None, None, None, None
        dout = _c(dout)
        _chk(dout)
        dvalue = torch.empty_like(value)
        dlogit = torch.empty_like(alpha)
        if ctx.seed_off is None:
            call("eqf_attn_aggregate_bwd", _p(alpha), _p(value), _p(graph.row_ptr), _p(dout), _p(dvalue), _p(dlogit),
                 graph.N, H, layout.c_ref, drop_p, seed, _stream())
        else:
            call("eqf_attn_aggregate_bwd_dseed", _p(alpha), _p(value), _p(graph.row_ptr), _p(dout), _p(dvalue), _p(dlogit),
                 graph.N, H, layout.c_ref, drop_p, seed, _p(ctx.seed_off), _stream())
        return dlogit, dvalue, None, None, None, None, None


class _AttnAggregateBwd(Function):
    """(d_logit, d_value) of the per-destination softmax + aggregation as a differentiable op of (logit, value, d_out);
    `alpha` is the softmax saved by the forward (its dependence on `logit` is accounted for in eqf_attn_aggregate_bwd2)."""

    @staticmethod
    def forward(ctx, logit, value, dout, alpha, graph, H, layout, drop_p, seed):
        dout = _c(dout)
        _chk(dout)
        dvalue = torch.empty_like(value)
        dlogit = torch.empty_like(alpha)
        call("eqf_attn_aggregate_bwd", _p(alpha), _p(value), _p(graph.row_ptr), _p(dout), _p(dvalue), _p(dlogit),
             graph.N, H, layout.c_ref, drop_p, seed, _stream())
        ctx.save_for_backward(alpha, value, dout)
        ctx.args = (graph, H, layout, drop_p, seed)
        return dlogit, dvalue

    @staticmethod
    @once_differentiable
    def backward(ctx, cl, cv):
        alpha, value, dout = ctx.saved_tensors
        graph, H, layout, drop_p, seed = ctx.args
        cl = _c(cl) if cl is not None else None
        cv = _c(cv) if cv is not None else None
        _chk(cl, cv)
        g_logit, g_value, g_dout = torch.empty_like(alpha), torch.empty_like(value), torch.empty_like(dout)
        call("eqf_attn_aggregate_bwd2", _p(alpha), _p(value), _p(graph.row_ptr), _p(dout), _p(cv), _p(cl), _p(g_logit),
             _p(g_value), _p(g_dout), graph.N, H, layout.c_ref, drop_p, seed, _stream())
        return g_logit, g_value, g_dout, None, None, None, None, None, None


def attn_aggregate(logit, value, graph, H, layout, drop_p=0.0, seed=0):
    return _AttnAggregate.apply(logit, value, graph, H, layout, float(drop_p), int(seed))


# ------------------------------------------------------------------------------------------------- dot-product attention
class _KvSplit(Function):
    """kv [E, 2D] -> (k, v) [E, D] each (rows follow the H-head irreps `layout`); linear, backward = _KvMerge."""

    @staticmethod
    def forward(ctx, kv, H, layout):
        kv = _c(kv)
        _chk(kv)
        E = kv.shape[0]
        assert kv.shape[1] == 2 * layout.dim
        k = torch.empty((E, layout.dim), device=kv.device, dtype=torch.float32)
        v = torch.empty_like(k)
        call("eqf_kv_split", _p(kv), _p(k), _p(v), E, H, layout.c_ref, _stream())
        ctx.args = (H, layout)
        return k, v

    @staticmethod
    def backward(ctx, dk, dv):
        H, layout = ctx.args
        return _KvMerge.apply(dk, dv, H, layout), None, None


class _KvMerge(Function):
    @staticmethod
    def forward(ctx, k, v, H, layout):
        ref = k if k is not None else v
        k = _c(k) if k is not None else None
        v = _c(v) if v is not None else None
        _chk(k, v)
        kv = torch.empty((ref.shape[0], 2 * layout.dim), device=ref.device, dtype=torch.float32)
        call("eqf_kv_merge", _p(k), _p(v), _p(kv), ref.shape[0], H, layout.c_ref, _stream())
        ctx.args = (H, layout)
        return kv

    @staticmethod
    def backward(ctx, dkv):
        H, layout = ctx.args
        k, v = _KvSplit.apply(dkv, H, layout)
        return k, v, None, None


def kv_split(kv, H, layout):
    return _KvSplit.apply(kv, H, layout)


def _dp_logits_fwd(q, k, graph, H, layout):
    logit = torch.empty((k.shape[0], H), device=k.device, dtype=torch.float32)
    call("eqf_dp_logits_fwd", _p(q), _p(k), _p(graph.dst), _p(logit), k.shape[0], H, layout.c_ref, _stream())
    return logit


def _dp_logits_bwd(q, k, dlogit, graph, H, layout, want_q, want_k):
    dq = torch.empty((graph.N, layout.dim), device=dlogit.device, dtype=torch.float32) if want_q else None
    dk = torch.empty((dlogit.shape[0], layout.dim), device=dlogit.device, dtype=torch.float32) if want_k else None
    call("eqf_dp_logits_bwd", _p(q) if want_k else None, _p(k) if want_q else None, _p(dlogit), _p(graph.row_ptr), _p(dq),
         _p(dk), graph.N, H, layout.c_ref, _stream())
    return dq, dk


class _DpLogits(Function):
    """logit[e,h] = <scaled q[dst[e]], k[e]> over the channels of head h [ref: nets/dp_attention_transformer.py:131-146]."""

    @staticmethod
    def forward(ctx, q, k, graph, H, layout):
        q, k = _c(q), _c(k)
        _chk(q, k)
        ctx.save_for_backward(q, k)
        ctx.args = (graph, H, layout)
        return _dp_logits_fwd(q, k, graph, H, layout)

    @staticmethod
    def backward(ctx, dlogit):
        q, k = ctx.saved_tensors
        graph, H, layout = ctx.args
        if torch.is_grad_enabled():  # create_graph
            dq, dk = _DpLogitsBwd.apply(q, k, dlogit, graph, H, layout)
            return dq, dk, None, None, None
        dlogit = _c(dlogit)
        _chk(dlogit)
        dq, dk = _dp_logits_bwd(q, k, dlogit, graph, H, layout, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dq, dk, None, None, None


class _DpLogitsBwd(Function):
    """(dq, dk) as a differentiable op of (q, k, dlogit).  The logits are bilinear, so with cotangents (cq, ck):
    g_dlogit = L(cq, k) + L(q, ck),  g_q = dq-map(dlogit, ck),  g_k = dk-map(dlogit, cq) -- the first-order kernels."""

    @staticmethod
    def forward(ctx, q, k, dlogit, graph, H, layout):
        dlogit = _c(dlogit)
        _chk(dlogit)
        ctx.save_for_backward(q, k, dlogit)
        ctx.args = (graph, H, layout)
        return _dp_logits_bwd(q, k, dlogit, graph, H, layout, True, True)

    @staticmethod
    @once_differentiable
    def backward(ctx, cq, ck):
        q, k, dlogit = ctx.saved_tensors
        graph, H, layout = ctx.args
        cq = _c(cq) if cq is not None else None
        ck = _c(ck) if ck is not None else None
        _chk(cq, ck)
        g_q = g_k = g_dl = None
        if cq is not None:
            g_dl = _dp_logits_fwd(cq, k, graph, H, layout)
            g_k = _dp_logits_bwd(cq, None, dlogit, graph, H, layout, False, True)[1]
        if ck is not None:
            t = _dp_logits_fwd(q, ck, graph, H, layout)
            g_dl = t if g_dl is None else g_dl + t
            g_q = _dp_logits_bwd(None, ck, dlogit, graph, H, layout, True, False)[0]
        return g_q, g_k, g_dl, None, None, None


def dp_logits(q, k, graph, H, layout):
    return _DpLogits.apply(q, k, graph, H, layout)
