"""Data parallelism over molecules: one process per GPU, one flat fp32 gradient buffer, two collectives per step.

The reference wraps the model in DistributedDataParallel over NCCL (main_qm9.py:178-179; oc20/trainer/
base_trainer_v2.py:376-384) with torch's default 25 MB buckets and shards the data set with a DistributedSampler
(main_qm9.py:204-210) or, for OC20, a sampler that balances the shards by atom count
(oc20/trainer/base_trainer_oc20.py:238-256).  The whole Equiformer gradient is 14-36 MB and xGMI is point-to-point
(7 links x ~153 GB/s per GPU), so a ring of many bucketed collectives is latency bound.  Here:

  * every parameter's gradient lives in ONE flat buffer (`FlatGradAllReduce.flat`, shared with the fused optimizer),
    laid out in forward order and cut once into a HEAD part (embeddings + first blocks) and a TAIL part (last blocks +
    output head);
  * the TAIL part -- whose gradients are complete first, backward runs the layers in reverse -- is all-reduced
    asynchronously from the gradient hook of whichever tail parameter is completed last, in the middle of backward, so
    that its RCCL collective overlaps the rest of backward; the HEAD part follows when backward is done (`reduce()`);
    parameters of late layers whose gradient nevertheless arrives at the very end (the radial MLPs evaluated side by
    side at the start of the forward) are laid out in the head part (`module.late_gradient_parameters()`);
  * molecules never interact (edges stay inside a molecule), so no other collective exists on the data path;
  * `shard_balanced` splits a batch over the ranks with near-equal sums of a per-molecule cost (edge count): the only
    scaling hazard of this path is the imbalance of the variable-size graphs.

Backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests.
"""
import torch
import torch.distributed as dist


FLAT_ALIGN = 64  # floats (256 bytes): every parameter's slice of a flat buffer starts on such a boundary


def flat_offsets(sizes, align=FLAT_ALIGN):
    """(offsets, total) of tensors of `sizes` elements laid out one after the other, each start rounded up to `align` elements:
    parameters that alias a flat buffer stay 16-byte aligned (the kernels' 16-byte loads of weights, the packed-plane kernels)
    whatever odd-sized tensors precede them.  The padding elements are zero in every buffer and stay zero (AdamW of 0 with
    gradient 0).  Shared by FlatGradAllReduce and FlatAdamW so that the two layouts coincide."""
    offs, off = [], 0
    for k in sizes:
        offs.append(off)
        off += (int(k) + align - 1) // align * align
    return offs, off


class FlatGradAllReduce:
    """Owns a flat buffer aliased by every parameter's .grad after `reduce()`; averages it across the ranks.

    overlap: fraction of the gradient bytes (counted from the END of the forward order) that is all-reduced from a
    hook during backward; 0 disables the hook (one collective in `reduce()`)."""

    def __init__(self, module, process_group=None, overlap=0.5, always_reduce=False):
        # always_reduce: issue the collectives even in a one-rank group (they are the identity there).  Used to exercise the
        # RCCL path -- ReduceOp.AVG, the asynchronous tail launched from the backward hook -- on a single-GPU box
        # (tests/test_gpu_parallel.py::test_one_rank_rccl_flat_allreduce); a real one-rank run has no reason to set it.
        self._always = bool(always_reduce)
        params = [p for p in module.parameters() if p.requires_grad]
        # Layout = forward order, except that parameters whose gradient is only complete at the END of backward although
        # they belong to late layers are moved to the front (the head bucket): a model says which ones through
        # `late_gradient_parameters()` -- for the Equiformer trunk the radial MLPs of all blocks, which are evaluated side
        # by side at the start of the forward (radial bank), so their backward node is among the last to run.
        late = {id(p) for p in getattr(module, "late_gradient_parameters", lambda: [])()}
        self.params = [p for p in params if id(p) in late] + [p for p in params if id(p) not in late]
        self.group = process_group
        self.offsets, n = flat_offsets([p.numel() for p in self.params])
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = [self.flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self.offsets)]
        # cut: parameters [k, end) form the tail bucket.  Every tail parameter reports its gradient through a
        # post-accumulate hook; the collective is launched by whichever arrives LAST, in whatever order autograd runs the
        # nodes.  (One backward() per reduce(): with several micro-batch backwards the first one would launch it.)
        self.split = len(self.params)
        for p in self.params:
            # this reducer's hook flushes the deferred weight gradients before it reads (ops._hooked): deferral stays on for
            # its parameters although a process group / gradient hooks exist
            p._eqf_flushes = True
        self._handles = []
        self._marks = []
        self._arrived = set()
        self._pending = None
        self._tail_done = False
        self._accumulate = False
        if overlap > 0 and len(self.params) > 1:
            want = n * (1.0 - overlap)
            k = next((i for i, o in enumerate(self.offsets) if o >= want), len(self.params))
            k = min(max(k, 1), len(self.params) - 1)
            self.split = k
            self._handles = [p.register_post_accumulate_grad_hook(self._on_tail_grad) for p in self.params[k:]]

    # ---------------------------------------------------------------------------------------------------------------
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _active(self):
        """collectives are issued: more than one rank, or a one-rank group with always_reduce"""
        if self.world_size() > 1:
            return True
        return self._always and dist.is_available() and dist.is_initialized()

    def broadcast_parameters(self, src=0):
        """Replicas start identical (what DDP's constructor does)."""
        if not self._active():
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)

    # ---- optional timing of what reduce() adds to a step: the head collective + whatever of the tail collective backward
    # did not hide (bench.py `allreduce_ms`).  Device tensors: HIP events on the current stream; CPU tensors: the host clock.
    timing = False

    def _mark(self):
        if self.flat.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def reduce_ms(self, reset=True):
        """mean milliseconds per reduce() since the last reset (synchronises the device), or None without samples"""
        if not getattr(self, "_marks", None):
            return None
        if self.flat.is_cuda:
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self._marks]
        else:
            ms = [1e3 * (b - a) for a, b in self._marks]
        if reset:
            self._marks = []
        return sum(ms) / len(ms)

    def _pack(self, lo, hi):
        """gradients of parameters [lo, hi) -> their slices of the flat buffer (multi-tensor copy: a handful of launches)"""
        have = [(p, v) for p, v in zip(self.params[lo:hi], self.views[lo:hi])
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for p, v in zip(self.params[lo:hi], self.views[lo:hi]) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])

    def _all_reduce(self, buf, async_op):
        backend = dist.get_backend(self.group)
        if backend == "nccl":  # RCCL averages in the collective itself
            return dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op), False
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op), True

    def no_sync(self):
        """Context manager for gradient accumulation: backward() passes inside it only accumulate into .grad (the hook is
        off); the LAST micro-batch runs outside it, followed by reduce().  [ref: DistributedDataParallel.no_sync]"""
        outer = self

        class _NoSync:
            def __enter__(self):
                outer._accumulate = True

            def __exit__(self, *exc):
                outer._accumulate = False
                return False
        return _NoSync()

    def _on_tail_grad(self, param):
        """Runs inside backward each time the gradient of a tail-bucket parameter has been accumulated."""
        if not self._active() or self._accumulate:
            return
        if self._tail_done:
            # the tail collective of this step is already in flight: a second backward() would accumulate into gradients
            # that are being (or have been) averaged, silently diverging the ranks
            raise RuntimeError("FlatGradAllReduce: backward() ran again before reduce(); wrap all but the last "
                               "micro-batch in no_sync()")
        self._arrived.add(id(param))
        if len(self._arrived) < len(self.params) - self.split:
            return
        # the hook ships gradients as soon as autograd has accumulated them: weight gradients that are only QUEUED at this point
        # (equiformer_amd.ops: deferred grouped launches of the node-row linears) are launched first -- stream-ordered before the
        # pack and the collective; what the rest of backward queues goes out when the pass ends, before reduce()
        from . import ops as _ops
        _ops.flush_deferred_weight_gradients()
        self._pack(self.split, len(self.params))
        tail = self.flat[self.offsets[self.split]:]
        self._pending = self._all_reduce(tail, async_op=True)
        self._tail_done = True

    def reduce(self):
        """Call after backward(): finishes the average of the flat gradient and re-points each .grad at its slice of the
        buffer (no copy back).  The collective SEQUENCE is the same on every rank whatever happened in backward: always
        all_reduce(tail) then all_reduce(head) when the buffer is split.  If the hook did not launch the tail on this rank
        (a tail parameter received no gradient here: an unused branch, a batch without edges), the tail is packed and
        launched now -- a rank that fell back to ONE collective over the whole buffer would pair its collective with
        another rank's tail collective (mismatched sizes: hang or corruption)."""
        ws = self.world_size()
        active = self._active()
        split = self.split < len(self.params)
        if not self._tail_done:
            self._pack(self.split, len(self.params))
            if active and split:
                self._pending = self._all_reduce(self.flat[self.offsets[self.split]:], async_op=True)
        self._pack(0, self.split)
        if active:
            mark = self._mark() if self.timing else None
            head = self.flat[:self.offsets[self.split]] if split else self.flat
            _, need_div_head = self._all_reduce(head, async_op=False)
            if need_div_head:
                head.div_(ws)
            if self._pending is not None:
                work, need_div = self._pending
                work.wait()
                if need_div:
                    self.flat[self.offsets[self.split]:].div_(ws)
            if mark is not None:
                self._marks.append((mark, self._mark()))
        self._pending, self._tail_done = None, False
        self._arrived.clear()
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat


def shard_molecules(num_molecules, rank, world_size):
    """Contiguous, balanced split of molecule indices (DistributedSampler semantics without shuffling)."""
    per = num_molecules // world_size
    rem = num_molecules % world_size
    start = rank * per + min(rank, rem)
    return range(start, start + per + (1 if rank < rem else 0))


def shard_balanced(costs, world_size):
    """Split molecules over `world_size` ranks with near-equal total cost (cost = edge count of the molecule: the edge
    pass is >95 % of the work).  Greedy longest-processing-time assignment with equal molecule counts per rank (+-1),
    so every rank also keeps the same per-step batch size.  Returns a list of index lists (ascending inside a rank).
    [ref: the role of BalancedBatchSampler, oc20/trainer/base_trainer_oc20.py:238-256]"""
    costs = [float(c) for c in costs]
    n = len(costs)
    cap = [n // world_size + (1 if r < n % world_size else 0) for r in range(world_size)]
    load = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for i in sorted(range(n), key=lambda i: -costs[i]):
        r = min((r for r in range(world_size) if len(out[r]) < cap[r]), key=lambda r: load[r])
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(ix) for ix in out]


def molecule_edge_counts(pos, batch, r, max_num_neighbors=1000):
    """Directed edges per molecule of the radius graph (device tensor, int64): the cost `shard_balanced` wants."""
    from .graph import EdgeGraph
    g = EdgeGraph.from_radius(pos, batch, r, max_num_neighbors)
    deg = (g.row_ptr[1:] - g.row_ptr[:-1]).to(torch.int64)
    return torch.zeros(g.num_graphs, dtype=torch.int64, device=pos.device).index_add_(0, g.batch.to(torch.int64), deg)
