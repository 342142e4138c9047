"""Data-parallel path on CPU: two `gloo` processes, molecules sharded with `shard_molecules`, ONE flat-gradient
all-reduce per step (equiformer_amd.parallel.FlatGradAllReduce).  The model here is the CPU oracle (the HIP product
has no CPU path); what is under test is the sharding + reduction logic that bench.py runs over RCCL with N GPUs:
the averaged gradient of the two shards must equal the gradient of the whole batch, and replicas must stay identical."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_model():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    from weights import fill_deterministic
    from oracle import nets as onets
    m = onets.GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **mg.SMALL_L2)
    return fill_deterministic(m.train(), 21)


def _loss(model, d, idx):
    """sum-reduced L1 over the molecules `idx` of batch d (each rank later divides by its own count)."""
    n = d["pos"].shape[0] // d["y"].shape[0]
    sel = torch.cat([torch.arange(i * n, (i + 1) * n) for i in idx])
    batch = torch.repeat_interleave(torch.arange(len(idx)), n)
    y = model(None, d["pos"][sel], batch, d["z"][sel]).squeeze(-1)
    return (y - d["y"][list(idx)]).abs().mean()


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from equiformer_amd.parallel import FlatGradAllReduce, shard_molecules
    from equiformer_amd.synthetic import qm9_like_batch
    model = _small_model()
    if rank == 1:  # replicas start different on purpose: broadcast_parameters must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    red = FlatGradAllReduce(model)
    red.broadcast_parameters()
    d = qm9_like_batch(4, 10, side=5.0, seed=5)
    idx = shard_molecules(4, rank, world)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    opt.zero_grad()
    _loss(model, d, idx).backward()
    overlapped = bool(red._tail_done)  # the tail bucket's collective was launched from the hook, during backward
    flat = red.reduce().clone()
    opt.step()
    checksum = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().sum()
    torch.save({"flat": flat, "checksum": checksum, "idx": list(idx), "overlapped": overlapped, "split": red.split,
                "nparams": len(red.params)}, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_flat_allreduce_matches_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert r0["idx"] == [0, 1] and r1["idx"] == [2, 3]
    assert r0["overlapped"] and r1["overlapped"], "the tail bucket must be reduced from the backward hook"
    assert 0 < r0["split"] < r0["nparams"]
    assert torch.equal(r0["flat"], r1["flat"]), "ranks disagree on the reduced gradient"
    assert abs(r0["checksum"].item() - r1["checksum"].item()) == 0.0, "replicas diverged after the step"
    # single-process reference: mean over the 4 molecules == mean of the two per-shard means (equal shard sizes)
    sys.path.insert(0, ROOT)
    from equiformer_amd.synthetic import qm9_like_batch
    model = _small_model()
    d = qm9_like_batch(4, 10, side=5.0, seed=5)
    _loss(model, d, range(4)).backward()
    full = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in model.parameters() if p.requires_grad])
    got = _unpad(r0["flat"], [p.numel() for p in model.parameters() if p.requires_grad])
    err = ((full - got).abs().max() / full.abs().max()).item()
    assert err < 1e-5, err


def _unpad(flat, sizes):
    """the parameters' slices of a flat buffer (each starts 256-byte aligned: equiformer_amd.parallel.flat_offsets), concatenated"""
    from equiformer_amd.parallel import flat_offsets
    offs, n = flat_offsets(sizes)
    assert flat.numel() == n
    pad = torch.ones(n, dtype=torch.bool)
    for o, k in zip(offs, sizes):
        pad[o:o + k] = False
    assert (flat[pad] == 0).all(), "padding elements of the flat gradient must stay zero"
    return torch.cat([flat[o:o + k] for o, k in zip(offs, sizes)])


class _Toy(torch.nn.Module):
    """Late layer (`z`, registered last) whose parameter is used FIRST in the forward: its gradient arrives last."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.a, self.b, self.c = (torch.nn.Linear(6, 6) for _ in range(3))
        self.z = torch.nn.Linear(6, 6)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        self.declare_late = True

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(torch.tanh(self.z(x))))))).sum()

    def late_gradient_parameters(self):
        return list(self.z.parameters()) if self.declare_late else []


def _toy_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from equiformer_amd.parallel import FlatGradAllReduce
    res = {}
    for declare in (True, False):
        model = _Toy()
        model.declare_late = declare
        red = FlatGradAllReduce(model, overlap=0.5)
        order = []
        hooks = [p.register_post_accumulate_grad_hook(lambda p, n=n: order.append((n, bool(red._tail_done))))
                 for n, p in model.named_parameters()]
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 + rank))
        model(x).backward()
        launched_after = next((i for i, (_, done) in enumerate(order) if done), None)  # hooks run in registration order
        overlapped = bool(red._tail_done)
        red.reduce()
        names = [n for p in red.params for n, q in model.named_parameters() if q is p]
        res[declare] = dict(names=names, overlapped=overlapped, launched_after=launched_after, arrivals=[n for n, _ in order],
                            grads={n: p.grad.clone() for n, p in model.named_parameters()})
        for h in hooks:
            h.remove()
    torch.save(res, os.path.join(out, "toy%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_tail_bucket_is_launched_by_its_last_gradient_whatever_the_order(tmp_path):
    """A late layer whose gradient arrives last: declared through late_gradient_parameters() it is laid out in the head
    bucket and the tail's collective starts in the middle of backward; undeclared, the collective still starts from a
    hook (when the straggler arrives) instead of falling back to the synchronous path.  Both average correctly."""
    world = 2
    mp.spawn(_toy_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "toy0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "toy1.pt"))
    for declare in (True, False):
        a, b = r0[declare], r1[declare]
        assert a["overlapped"] and b["overlapped"]
        assert a["arrivals"][-2:] == ["z.weight", "z.bias"] or set(a["arrivals"][-2:]) == {"z.weight", "z.bias"}
        for n in a["grads"]:
            assert torch.equal(a["grads"][n], b["grads"][n]), n
        # reference: mean of the two ranks' local gradients
        ref = {}
        for rank in range(world):
            m = _Toy()
            m(torch.randn(5, 6, generator=torch.Generator().manual_seed(10 + rank))).backward()
            for n, p in m.named_parameters():
                ref[n] = ref.get(n, 0) + p.grad / world
        for n in ref:
            assert torch.allclose(a["grads"][n], ref[n], rtol=1e-6, atol=1e-7), n
    assert r0[True]["names"][:2] == ["z.weight", "z.bias"] and r0[False]["names"][-2:] == ["z.weight", "z.bias"]
    n = len(r0[True]["arrivals"])
    assert r0[True]["launched_after"] < n - 2 <= r0[False]["launched_after"], (r0[True]["launched_after"], r0[False]["launched_after"])


class _Branchy(torch.nn.Module):
    """`c` (in the tail bucket) is skipped when skip_c is set: that rank's hook never completes the tail."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(4)
        self.a, self.b, self.c = (torch.nn.Linear(6, 6) for _ in range(3))
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def forward(self, x, skip_c=False):
        h = torch.tanh(self.b(torch.tanh(self.a(x))))
        return (h if skip_c else self.c(h)).sum()


def _asym_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from equiformer_amd.parallel import FlatGradAllReduce
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(20 + rank))
    res = {}
    # (1) rank 1 does not use a tail parameter: rank 0 launches the tail from its hook, rank 1 only in reduce()
    model = _Branchy()
    red = FlatGradAllReduce(model, overlap=0.5)
    model(x, skip_c=(rank == 1)).backward()
    res["hook_fired"] = bool(red._tail_done)
    red.reduce()
    res["asym"] = {n: p.grad.clone() for n, p in model.named_parameters()}
    # (2) gradient accumulation: two micro-batches, the first under no_sync()
    model = _Branchy()
    red = FlatGradAllReduce(model, overlap=0.5)
    with red.no_sync():
        model(x).backward()
        assert not red._tail_done
    model(2 * x).backward()
    assert red._tail_done
    red.reduce()
    res["accum"] = {n: p.grad.clone() for n, p in model.named_parameters()}
    # (3) a second backward() without no_sync() after the tail went out must fail loudly, not corrupt the average
    model(x).backward()
    try:
        model(x).backward()
        res["raised"] = False
    except RuntimeError as e:
        res["raised"] = "no_sync" in str(e)
    red.reduce()  # leaves both ranks in step: the failed backward stopped inside the hook on both
    torch.save(res, os.path.join(out, "asym%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_sequence_is_rank_independent_and_accumulation_is_guarded(tmp_path):
    """Round-2 advisor findings: (medium) a rank whose hook did not fire used to issue ONE whole-buffer all-reduce against
    the other rank's tail + head pair (mismatched sizes); (low) a second backward() before reduce() was silently wrong."""
    world = 2
    mp.spawn(_asym_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "asym0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "asym1.pt"))
    assert r0["hook_fired"] and not r1["hook_fired"]
    assert r0["raised"] and r1["raised"]
    xs = [torch.randn(5, 6, generator=torch.Generator().manual_seed(20 + rank)) for rank in range(world)]
    ref_asym, ref_acc = {}, {}
    for rank in range(world):
        m = _Branchy()
        m(xs[rank], skip_c=(rank == 1)).backward()
        for n, p in m.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            ref_asym[n] = ref_asym.get(n, 0) + g / world
        m = _Branchy()
        m(xs[rank]).backward()
        m(2 * xs[rank]).backward()
        for n, p in m.named_parameters():
            ref_acc[n] = ref_acc.get(n, 0) + p.grad / world
    for key, ref in (("asym", ref_asym), ("accum", ref_acc)):
        for n in ref:
            assert torch.equal(r0[key][n], r1[key][n]), (key, n)
            assert torch.allclose(r0[key][n], ref[n], rtol=1e-6, atol=1e-7), (key, n)


def test_shard_molecules_partitions_every_molecule_once():
    from equiformer_amd.parallel import shard_molecules
    for n in (0, 1, 7, 128, 1000):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_molecules(n, r, w)]
            assert got == list(range(n))
            sizes = [len(shard_molecules(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_shard_balanced_equalises_edge_counts():
    from equiformer_amd.parallel import shard_balanced
    g = torch.Generator().manual_seed(0)
    costs = (torch.randint(50, 2800, (257,), generator=g)).tolist()  # OC20-like spread of edge counts
    for w in (1, 2, 4, 8):
        shards = shard_balanced(costs, w)
        assert sorted(i for s in shards for i in s) == list(range(len(costs)))
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs), (w, loads)
        # contiguous sharding of the same list is (much) worse once the costs are sorted, the adversarial order
        srt = sorted(costs)
        per = len(srt) // w
        naive = [sum(srt[r * per:(r + 1) * per]) for r in range(w)]
        bal = [sum(srt[i] for i in s) for s in shard_balanced(srt, w)]
        assert max(bal) - min(bal) <= max(naive) - min(naive)


def _loss_sum(model, d, idx, denom):
    """L1 summed over the molecules `idx`, divided by `denom` (the per-rank share of the global batch): the AVERAGE over ranks
    of these losses is the global mean whatever the shard sizes."""
    n = d["pos"].shape[0] // d["y"].shape[0]
    sel = torch.cat([torch.arange(i * n, (i + 1) * n) for i in idx])
    batch = torch.repeat_interleave(torch.arange(len(idx)), n)
    y = model(None, d["pos"][sel], batch, d["z"][sel]).squeeze(-1)
    return (y - d["y"][list(idx)]).abs().sum() / denom


def _edge_costs(d, nmol, r=5.0):
    n = d["pos"].shape[0] // nmol
    p = d["pos"].view(nmol, n, 3)
    dist2 = ((p[:, :, None] - p[:, None]) ** 2).sum(-1)
    return [int(((dist2[i] < r * r).sum() - n).item()) for i in range(nmol)]


def _worker8(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from equiformer_amd.parallel import FlatGradAllReduce, shard_balanced
    from equiformer_amd.synthetic import qm9_like_batch
    model = _small_model()
    red = FlatGradAllReduce(model)
    red.broadcast_parameters()
    nmol = 19  # 19 molecules over 8 ranks: shards of 3 and 2
    d = qm9_like_batch(nmol, 8, side=4.5, seed=11)
    shards = shard_balanced(_edge_costs(d, nmol), world)
    idx = shards[rank]
    if rank == 3:  # this rank's tail hook never fires: reduce() has to issue the tail collective itself, in the same order
        for h in red._handles:
            h.remove()
    model.zero_grad()
    _loss_sum(model, d, idx, nmol / world).backward()
    hooked = bool(red._tail_done)
    flat = red.reduce().clone()
    torch.save({"flat": flat, "idx": list(idx), "hooked": hooked, "world": dist.get_world_size()},
               os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gloo_uneven_balanced_shards_match_full_batch(tmp_path):
    """nothing had exercised more than two ranks.  Eight gloo ranks, 19 molecules dealt by `shard_balanced` on
    their edge counts (shards of 3 and 2 molecules), one rank whose backward hook does not fire: every rank ends with the SAME
    flat gradient, equal to the single-process gradient of the whole batch; the collective sequence (tail, then head) cannot
    depend on the hook -- a mismatch would hang or corrupt, here it would time out."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world)]
    assert all(r["world"] == world for r in rs)
    assert sorted(i for r in rs for i in r["idx"]) == list(range(19))
    assert sorted(len(r["idx"]) for r in rs) == [2] * 5 + [3] * 3
    assert not rs[3]["hooked"] and all(r["hooked"] for k, r in enumerate(rs) if k != 3)
    for r in rs[1:]:
        assert torch.equal(rs[0]["flat"], r["flat"]), "ranks disagree on the reduced gradient"
    sys.path.insert(0, ROOT)
    from equiformer_amd.synthetic import qm9_like_batch
    model = _small_model()
    d = qm9_like_batch(19, 8, side=4.5, seed=11)
    _loss_sum(model, d, range(19), 19.0).backward()
    full = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in model.parameters() if p.requires_grad])
    got = _unpad(rs[0]["flat"], [p.numel() for p in model.parameters() if p.requires_grad])
    err = ((full - got).abs().max() / full.abs().max()).item()
    assert err < 1e-5, err
