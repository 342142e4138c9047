"""The N > 1 path with the HIP model: two ranks (gloo, both on cuda:0 -- the GPU boxes of the test tier have one GPU)
run the sharded train step through FlatGradAllReduce (tail bucket reduced from the backward hook) + FlatAdamW; the
averaged gradient must equal the full-batch gradient of a single process and the replicas must stay identical.
[ref: DDP + DistributedSampler, main_qm9.py:178-179,204-210]"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unpad(flat, sizes):
    """the parameters' slices of a flat gradient buffer (256-byte aligned starts, equiformer_amd.parallel.flat_offsets)"""
    from equiformer_amd.parallel import flat_offsets
    offs, n = flat_offsets(sizes)
    assert flat.numel() == n
    return torch.cat([flat[o:o + k] for o, k in zip(offs, sizes)])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    from weights import fill_deterministic
    from equiformer_amd.nets.graph_attention_transformer import GraphAttentionTransformer
    m = GraphAttentionTransformer(irreps_in="5x0e", max_radius=5.0, number_of_basis=32, **dict(mg.SMALL_L2, alpha_drop=0.0))
    return fill_deterministic(m, 21).to(dev).train()


def _loss(model, d, idx, dev):
    n = d["pos"].shape[0] // d["y"].shape[0]
    sel = torch.cat([torch.arange(i * n, (i + 1) * n) for i in idx])
    batch = torch.repeat_interleave(torch.arange(len(idx)), n)
    y = model(None, d["pos"][sel].to(dev), batch.to(dev), d["z"][sel].to(dev)).squeeze(-1)
    return (y - d["y"][list(idx)].to(dev)).abs().mean()


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from equiformer_amd.optim import FlatAdamW, add_weight_decay
    from equiformer_amd.parallel import FlatGradAllReduce, shard_molecules
    from equiformer_amd.synthetic import qm9_like_batch
    model = _model(dev)
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.25)
    red = FlatGradAllReduce(model)
    red.broadcast_parameters()
    opt = FlatAdamW(add_weight_decay(model, 5e-3, model.no_weight_decay()), lr=1e-3, reducer=red)
    d = qm9_like_batch(4, 10, side=5.0, seed=5)
    idx = shard_molecules(4, rank, world)
    from equiformer_amd import ops
    opt.zero_grad(set_to_none=True)
    ops.deferred_weight_gradient_stats(reset=True)
    _loss(model, d, idx, dev).backward()
    deferred = ops.deferred_weight_gradient_stats()  # the grouped weight gradients coexist with the reducer's hook (round 5)
    overlapped = bool(red._tail_done)
    flat = red.reduce().clone().cpu()
    opt.step()
    torch.cuda.synchronize()
    checksum = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().sum().cpu()
    torch.save({"flat": flat, "checksum": checksum, "overlapped": overlapped, "deferred": deferred},
               os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_model_flat_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert r0["overlapped"] and r1["overlapped"]
    for r in (r0, r1):  # the N > 1 step is the N = 1 step: node-row weight gradients queued, flushed at the hook and at the end
        assert r["deferred"]["queued"] > 0 and r["deferred"]["flushes"] >= 2, r["deferred"]
    assert torch.equal(r0["flat"], r1["flat"]), "ranks disagree on the reduced gradient"
    assert r0["checksum"].item() == r1["checksum"].item(), "replicas diverged after the optimizer step"
    sys.path.insert(0, ROOT)
    from equiformer_amd.synthetic import qm9_like_batch
    dev = torch.device("cuda:0")
    model = _model(dev)
    d = qm9_like_batch(4, 10, side=5.0, seed=5)
    _loss(model, d, range(4), dev).backward()
    # the reducer lays the radial-bank parameters (late gradients) out first, the rest in forward order
    late = {id(p) for p in model.late_gradient_parameters()}
    assert late, "the trunk declares its radial bank"
    params = [p for p in model.parameters() if p.requires_grad]
    ordered = [p for p in params if id(p) in late] + [p for p in params if id(p) not in late]
    full = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ordered]).cpu()
    err = ((full - _unpad(r0["flat"], [p.numel() for p in ordered])).abs().max() / full.abs().max()).item()
    assert err < 2e-5, err


def _rccl_worker(rank, world, port, out):
    """ONE rank, backend "nccl" (= RCCL on ROCm) on cuda:0: the collectives are the identity, but they are real RCCL calls --
    ReduceOp.AVG, the asynchronous tail launched from the backward hook, broadcast, barrier."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from equiformer_amd.optim import FlatAdamW, add_weight_decay
    from equiformer_amd.parallel import FlatGradAllReduce
    from equiformer_amd.synthetic import qm9_like_batch
    res = {"backend": dist.get_backend()}
    model = _model(dev)
    red = FlatGradAllReduce(model, always_reduce=True)
    red.broadcast_parameters()
    # the AVG branch on its own
    buf = torch.arange(1000, dtype=torch.float32, device=dev) * 0.5
    work, need_div = red._all_reduce(buf, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    res["avg_identity"] = bool(torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.float32) * 0.5)) and not need_div
    opt = FlatAdamW(add_weight_decay(model, 5e-3, model.no_weight_decay()), lr=1e-3, reducer=red)
    d = qm9_like_batch(4, 10, side=5.0, seed=5)
    from equiformer_amd import ops
    opt.zero_grad(set_to_none=True)
    ops.deferred_weight_gradient_stats(reset=True)
    _loss(model, d, range(4), dev).backward()
    res["deferred"] = ops.deferred_weight_gradient_stats()
    res["overlapped"] = bool(red._tail_done)      # the tail collective was launched from the hook, inside backward
    res["pending_is_work"] = red._pending is not None and hasattr(red._pending[0], "wait")
    res["flat"] = red.reduce().clone().cpu()
    opt.step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.save(res, os.path.join(out, "rccl.pt"))
    dist.destroy_process_group()


def test_one_rank_rccl_flat_allreduce(tmp_path):
    """RCCL evidence that fits one GPU (the test tier's boxes have one): FlatGradAllReduce's "nccl" branch -- ReduceOp.AVG inside
    the collective, asynchronous tail from the backward hook, head in reduce() -- runs in a one-rank RCCL group and leaves the
    gradient equal to the single-process one.  [ref: init_process_group('nccl') utils.py:46-69, DDP main_qm9.py:178-179]"""
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(tmp_path, "rccl.pt"))
    assert r["backend"] == "nccl"
    assert r["avg_identity"], "ReduceOp.AVG over one rank must be the identity and need no division"
    assert r["overlapped"] and r["pending_is_work"], "the tail collective was not launched from the backward hook"
    assert r["deferred"]["queued"] > 0 and r["deferred"]["flushes"] >= 2, r["deferred"]
    sys.path.insert(0, ROOT)
    from equiformer_amd.synthetic import qm9_like_batch
    dev = torch.device("cuda:0")
    model = _model(dev)
    d = qm9_like_batch(4, 10, side=5.0, seed=5)
    _loss(model, d, range(4), dev).backward()
    late = {id(p) for p in model.late_gradient_parameters()}
    params = [p for p in model.parameters() if p.requires_grad]
    ordered = [p for p in params if id(p) in late] + [p for p in params if id(p) not in late]
    full = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ordered]).cpu()
    err = ((full - _unpad(r["flat"], [p.numel() for p in ordered])).abs().max() / full.abs().max()).item()
    assert err < 2e-5, err


def test_molecule_edge_counts_and_balanced_shards():
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.parallel import molecule_edge_counts, shard_balanced
    from equiformer_amd.synthetic import qm9_like_batch
    dev = torch.device("cuda:0")
    parts = [qm9_like_batch(1, n, side=4.0 + 0.2 * n, seed=n) for n in (6, 18, 9, 14, 18, 7, 11, 16)]
    pos = torch.cat([p["pos"] for p in parts]).to(dev)
    batch = torch.cat([torch.full((p["pos"].shape[0],), i) for i, p in enumerate(parts)]).to(dev)
    counts = molecule_edge_counts(pos, batch, 5.0)
    g = EdgeGraph.from_radius(pos, batch, 5.0)
    assert int(counts.sum()) == g.E and counts.shape[0] == 8
    shards = shard_balanced(counts.tolist(), 2)
    loads = [sum(int(counts[i]) for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= int(counts.max())


def test_bench_gpus_2_runs_by_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it re-runs itself as two ranks (VERDICT r4 weak #8a: the assert on
    WORLD_SIZE killed the driver-shaped command).  Both ranks share cuda:0 over gloo here (one-GPU test boxes); the line says
    n_gpus 2 and that the grouped weight gradients ran beside the reducer."""
    import json
    import subprocess
    env = dict(os.environ, EQF_BENCH_BACKEND="gloo", EQF_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--repeats", "1", "--no-cpu-baseline", "--no-sub-records"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 32
    dw = d["config"]["deferred_weight_gradients"]
    assert dw["queued"] > 0 and dw["flushes"] >= 2 * dw["steps"], dw


def test_bench_gpus_8_runs_by_itself_on_one_gpu(tmp_path):
    """`--gpus 8` must not fail on first contact with an 8-GPU node.  Eight ranks share cuda:0 over gloo here
    (tiny batch): the line carries the rank count the process group saw, every rank's shard size and what reduce() added per step."""
    import json
    import subprocess
    env = dict(os.environ, EQF_BENCH_BACKEND="gloo", EQF_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--repeats", "1", "--no-cpu-baseline", "--no-sub-records", "--prewarm-s", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 32
    pr = d["config"]["per_rank"]
    assert len(pr["edges_per_gpu"]) == 8 and all(e > 0 for e in pr["edges_per_gpu"]) and len(set(pr["edges_per_gpu"])) > 1
    assert d["allreduce_ms"] > 0 and all(v > 0 for v in pr["allreduce_ms"])
    assert d["config"]["hip_graph"] is False  # data-parallel steps launch eagerly (the reducer's collectives are not captured)
