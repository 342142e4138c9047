"""Data parallelism over molecules: one process per GPU, one flat fp32 gradient buffer, ONE all-reduce per step.

The reference wraps the model in DistributedDataParallel over NCCL (main_qm9.py:178-179; oc20/trainer/
base_trainer_v2.py:376-384) with torch's default 25 MB buckets.  The whole Equiformer gradient is 14-36 MB and xGMI
is point-to-point (7 links x ~153 GB/s per GPU), so a ring of several bucketed collectives is latency bound; one
flat buffer = a single RCCL all-reduce per step (backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).
Molecules never interact (edges stay inside a molecule), so no other collective exists on the data path.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Owns a flat buffer aliased by every parameter's .grad; `reduce()` averages it across ranks in one collective."""

    def __init__(self, module, process_group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.group = process_group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            self.views.append(v)
            off += p.numel()

    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, src=0):
        """Replicas start identical (what DDP's constructor does)."""
        if self.world_size() == 1:
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)

    def reduce(self):
        """Pack every .grad into the flat buffer (multi-tensor copy: a handful of launches, not one per parameter),
        all-reduce(mean) it in ONE collective, and re-point each .grad at its slice of the buffer (no copy back).
        Call after backward()."""
        have = [(p, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        missing = [v for p, v in zip(self.params, self.views) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
        ws = self.world_size()
        if ws > 1:
            backend = dist.get_backend(self.group)
            if backend == "nccl":  # RCCL averages in the collective itself
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.div_(ws)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat


def shard_molecules(num_molecules, rank, world_size):
    """Contiguous, balanced split of molecule indices (DistributedSampler semantics without shuffling)."""
    per = num_molecules // world_size
    rem = num_molecules % world_size
    start = rank * per + min(rank, rem)
    return range(start, start + per + (1 if rank < rem else 0))
