#!/usr/bin/env python
"""Headline benchmark: molecules/sec of a full QM9 train step (graph_attention_transformer_nonlinear_l2, L_max=2,
6 blocks, fp32) on N MI355X, data-parallel over molecules, plus the roofline of the dominant kernel and the CPU
oracle timed on the host cores.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = graph construction from positions + forward + L1 loss + backward + (N>1: ONE flat-gradient RCCL
all-reduce) + AdamW update, on a synthetic QM9-shaped batch already resident in HBM (128 molecules x 18 atoms per
GPU, ~200 directed edges per molecule at r = 5 A).  Weak scaling: per-GPU batch fixed.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL = "graph_attention_transformer_nonlinear_l2"
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32)
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="molecules per GPU (reference script: 128)")
    ap.add_argument("--atoms", type=int, default=18)
    ap.add_argument("--side", type=float, default=6.5, help="cube edge of the synthetic molecules (6.5 -> ~200 edges)")
    ap.add_argument("--dominant", default="",
                    help="kernel-name substring timed with HIP events for the roofline line (default: the dominant SeparableFCTP kernel, picked by one untimed step after the warm-up)")
    ap.add_argument("--cpu-molecules", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--no-cpu-full-batch", dest="cpu_full_batch", action="store_false",
                    help="skip the like-for-like CPU run at the bench batch (3 steps of ~15 s)")
    ap.add_argument("--cpu-threads", type=int, default=8,
                    help="threads of the CPU-oracle leg (BASELINE.md section 2: torch.set_num_threads(8)); capped by the host's cores")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="internal: run only the CPU-oracle leg and print its JSON object (the default run starts this as a "
                         "background process beside the GPU sub-records)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions of K steps each: 1 = only the contract's region; 3 (default) adds two repeats for the spread")
    ap.add_argument("--no-sub-records", dest="sub_records", action="store_false",
                    help="skip the compact records of the other BASELINE configurations (bf16 mode, MD17 L2 / L3, OC20)")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="A/B switch: weight gradients of the fused SeparableFCTP on a side stream beside the data gradient")
    ap.add_argument("--workload", default="qm9", choices=["qm9", "md17_l2", "md17_l3", "oc20"],
                    help="qm9 (default, the headline: BASELINE configs #1/#2) | md17_l2 / md17_l3 (configs #3/#4: aspirin "
                         "energy + forces, force-loss train step through the second-order backward, 8 / 5 frames per GPU as "
                         "the reference scripts) | oc20 (config #5: IS2RE l1_256_nonlinear, 16 structures per GPU, periodic "
                         "graph built on the device)")
    ap.add_argument("--prewarm-s", type=float, default=1.5,
                    help="seconds of untimed steps BEFORE the W warm-up steps (setup, like building the model): a box that comes "
                         "out of idle clocks ramps for a second or two -- profiles/r05/r05_drv_bench_default_cold_box.json "
                         "has a headline region at 11 138 molecules/s followed, 3 s later in the same process, by a bf16 "
                         "sub-record at 13 749.  0 disables; the timed region is still exactly K steps after W warm-up steps")
    ap.add_argument("--no-hip-graph", dest="hip_graph", action="store_false",
                    help="launch the QM9 step eagerly (~280 launches per step) instead of replaying its HIP graph "
                         "(equiformer_amd/capture.py; one GPU only: data-parallel steps are always eager)")
    ap.add_argument("--diag-static-graph", action="store_true",
                    help="DIAGNOSTIC, not a valid measurement: build the radius graph once outside the step (no host "
                         "synchronisation inside the step) -- shows how much of the step is the graph's sync bubble")
    ap.add_argument("--matrix-mode", default="split", choices=["split", "bf16", "fp32", "split6"],
                    help="arithmetic of the fused SeparableFCTP matrix steps (equiformer_amd.ops.set_matrix_mode): split = "
                         "fp32 operands as bf16 planes on the bf16 matrix cores (fp32-class results, the headline); bf16 = "
                         "plain bf16 operands (BASELINE config #2, the reference's AMP default); fp32 = exact-fp32 MFMA")
    return ap.parse_args()


def make_optimizer(model, lr=5e-4, weight_decay=5e-3, reducer=None):
    """AdamW with the reference's name-based no-weight-decay groups (optim_factory.py:27-42,126-127), as the fused
    flat-buffer HIP optimizer of equiformer_amd/optim.py (bit-for-bit torch.optim.AdamW semantics, tests/test_optim.py)."""
    from equiformer_amd.optim import FlatAdamW, add_weight_decay
    return FlatAdamW(add_weight_decay(model, weight_decay, model.no_weight_decay()), lr=lr, reducer=reducer)


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference, plain torch fp32) doing the same train step on the host cores
    (SURVEY.md section 8d): at the bench's own batch (128 molecules: like for like, two steps -- one step is ~35 s) and
    at the reference's CPU plumbing batch (8 molecules: 3 warm-up + 10 timed steps); medians.  `value` is the
    like-for-like figure."""
    import statistics
    from equiformer_amd.synthetic import qm9_like_batch
    from oracle import nets as onets
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = onets.graph_attention_transformer_nonlinear_l2("5x0e", 5.0).train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=5e-3)

    def run(molecules, warm, steps, budget_s):
        d = qm9_like_batch(molecules, args.atoms, side=args.side, seed=0)

        def step():
            opt.zero_grad()
            loss = (model(None, d["pos"], d["batch"], d["z"]).squeeze() - d["y"]).abs().mean()
            loss.backward()
            opt.step()

        for _ in range(warm):
            step()
        ts, t_begin = [], time.perf_counter()
        while len(ts) < steps and (not ts or time.perf_counter() - t_begin < budget_s):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), len(ts)

    small_dt, small_n = run(args.cpu_molecules, 3, args.cpu_steps, 30.0)
    out = {"unit": "molecules/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
           "what": "oracle: the CPU restatement of the reference (oracle/nets.py, plain torch fp32 ops), not the reference's own "
                   "code -- its dependencies (e3nn, torch_scatter, torch_cluster, PyG) are absent from this image",
           "batch_%d" % args.cpu_molecules: {"value": args.cpu_molecules / small_dt, "s_per_step": small_dt,
                                             "timed_steps": small_n, "warmup_steps": 3}}
    big = args.batch if args.cpu_full_batch else 0
    if big:
        # one warm-up step + three timed ones (a step takes 35-60 s on 8 threads; the leg runs as a background process beside the
        # GPU sub-records of the default run, so it costs the bench no wall time)
        big_dt, big_n = run(big, 1, 3, 200.0)
        out["batch_%d" % big] = {"value": big / big_dt, "s_per_step": big_dt, "timed_steps": big_n, "warmup_steps": 1}
        out["value"] = big / big_dt
        out["sample"] = ("median of %d train steps of %d molecules x %d atoms after 1 warm-up step (the bench batch; oracle = CPU "
                         "restatement of the reference, torch fp32, %d threads of %d host cores), %.2f s/step; batch %d: median of %d "
                         "steps after 3 warm-up steps, %.3f s/step"
                         % (big_n, big, args.atoms, cores, os.cpu_count() or 0, big_dt, args.cpu_molecules, small_n, small_dt))
    else:
        out["value"] = args.cpu_molecules / small_dt
        out["sample"] = ("median of %d train steps of %d molecules x %d atoms (oracle, torch fp32 CPU, %d threads), "
                         "%.3f s/step" % (small_n, args.cpu_molecules, args.atoms, cores, small_dt))
    return out



# ------------------------------------------------------------------------------------------------------------ workloads
WORKLOADS = {
    "qm9": dict(model=MODEL, unit="molecules/s", metric="molecules/sec (train step) QM9 L_max=2, 6 blocks"),
    "md17_l2": dict(model="graph_attention_transformer_nonlinear_exp_l2_md17", unit="frames/s", frames=8, wf=80.0,
                    metric="frames/sec (energy + force train step) MD17 aspirin L_max=2, 6 blocks"),
    "md17_l3": dict(model="graph_attention_transformer_nonlinear_exp_l3_md17", unit="frames/s", frames=5, wf=100.0,
                    metric="frames/sec (energy + force train step) MD17 aspirin L_max=3, 6 blocks"),
    "oc20": dict(model="oc20_l1_256_nonlinear", unit="structures/s", structures=16, atoms=78,
                 metric="structures/sec (train step) OC20 IS2RE l1_256_nonlinear, 6 blocks"),
}


def _oc20_batch(B, Na, seed):
    """slab + adsorbate shaped synthetic structures: orthorhombic 11 x 11 x 30 A cell, atoms in the lower 45 %"""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(seed)
    cell = torch.diag(torch.tensor([11.0, 11.0, 30.0]))[None].repeat(B, 1, 1)
    pos = (torch.rand(B * Na, 3, generator=g) * torch.tensor([1.0, 1.0, 0.45])) @ cell[0]
    return SimpleNamespace(pos=pos, batch=torch.arange(B).repeat_interleave(Na), cell=cell,
                           atomic_numbers=torch.randint(1, 84, (B * Na,), generator=g),
                           tags=torch.randint(0, 3, (B * Na,), generator=g), natoms=torch.full((B,), Na)), torch.randn(B, generator=g)


def build_workload(args, dev, rank, world):
    """-> (model, optimizer-ready step pieces): dict(model, reducer, step(opt) -> loss, units per step, graph size, workload text)"""
    from equiformer_amd import nets
    from equiformer_amd.graph import EdgeGraph
    from equiformer_amd.parallel import FlatGradAllReduce
    from equiformer_amd.synthetic import md17_aspirin_batch, qm9_like_batch
    W = WORKLOADS[args.workload]
    torch.manual_seed(0)
    if args.workload == "qm9":
        model = nets.model_entrypoint(W["model"])(irreps_in="5x0e", radius=5.0, num_basis=128).to(dev).train()
        d = {k: v.to(dev) for k, v in qm9_like_batch(args.batch, args.atoms, side=args.side, seed=1000 + rank).items()}

        g = EdgeGraph.from_radius(d["pos"], d["batch"], 5.0)
        static_graph = g if getattr(args, "diag_static_graph", False) else None

        def fwd_loss(graph=None):
            pred = model(f_in=None, pos=d["pos"], batch=d["batch"], node_atom=d["z"], graph=graph or static_graph)
            return (pred.squeeze() - d["y"]).abs().mean()  # L1Loss (main_qm9.py:188-189)

        def build_graph(into):  # the radius graph of the step's batch, rebuilt EVERY step (outside the captured part)
            return EdgeGraph.from_radius(d["pos"], d["batch"], 5.0, num_graphs=args.batch, into=into)
        units = args.batch
        text = ("QM9 %s train step (radius graph + fwd + L1 + bwd + AdamW), %d molecules/GPU x %d atoms, r=5.0, "
                "num_basis=128, alpha_drop=0.2" % (W["model"], args.batch, args.atoms))
        opt_kw = dict(lr=5e-4, weight_decay=5e-3)
    elif args.workload.startswith("md17"):
        frames = W["frames"]
        model = nets.model_entrypoint(W["model"])(irreps_in="64x0e", radius=5.0, num_basis=32).to(dev).train()
        d = {k: v.to(dev) for k, v in md17_aspirin_batch(frames, seed=1000 + rank).items()}
        gen = torch.Generator().manual_seed(7 + rank)
        ty = torch.randn(frames, 1, generator=gen).to(dev)
        tf = torch.randn(frames * 21, 3, generator=gen).to(dev)

        def fwd_loss(graph=None):  # L2MAE force loss + L1 energy loss with the scripts' weights (se_l2 / se_l3 target@aspirin.sh)
            E, F = model(node_atom=d["z"], pos=d["pos"], batch=d["batch"], graph=graph)
            return (E - ty).abs().mean() + W["wf"] * (F - tf).norm(dim=1).mean()

        def build_graph(into):
            return EdgeGraph.from_radius(d["pos"], d["batch"], 5.0, num_graphs=frames, into=into)
        g = EdgeGraph.from_radius(d["pos"], d["batch"], 5.0)
        units = frames
        text = ("MD17 aspirin %s force-loss train step (radius graph + fwd + forces by create_graph backward + loss + second-"
                "order bwd + AdamW), %d frames/GPU x 21 atoms, r=5.0, num_basis=32" % (W["model"], frames))
        opt_kw = dict(lr=5e-4, weight_decay=1e-6)
    else:
        B, Na = W["structures"], W["atoms"]
        model = nets.model_entrypoint(W["model"])().to(dev).train()
        data, y = _oc20_batch(B, Na, 1000 + rank)
        for k, v in vars(data).items():
            setattr(data, k, v.to(dev))
        y = y.to(dev)

        def fwd_loss():
            return (model(data).squeeze() - y).abs().mean()
        g, _, _ = EdgeGraph.from_radius_pbc(data.pos, data.cell, data.batch, 5.0, 500)
        units = B
        text = ("OC20 IS2RE %s train step (periodic radius graph on the device + fwd + L1 + bwd + AdamW), %d structures/GPU x "
                "%d atoms, r=5.0, max_neighbors=500, alpha_drop=0.2" % (W["model"], B, Na))
        opt_kw = dict(lr=2e-4, weight_decay=1e-3)
    reducer = FlatGradAllReduce(model)
    reducer.broadcast_parameters()
    opt = make_optimizer(model, reducer=reducer if world > 1 else None, **opt_kw)

    captured = None
    if (args.workload in ("qm9", "md17_l2", "md17_l3") and world == 1 and getattr(args, "hip_graph", True)
            and not getattr(args, "diag_static_graph", False)):
        # one GPU: forward + loss + backward + AdamW replayed as ONE HIP graph per step (equiformer_amd/capture.py); the radius
        # graph is still rebuilt from the positions every step, outside the graph (its edge count is read back on the host)
        from equiformer_amd.capture import CapturedTrainStep
        captured = CapturedTrainStep(opt, fwd_loss)

    def step_eager():
        opt.zero_grad(set_to_none=True)
        loss = fwd_loss()
        loss.backward()
        if world > 1:
            reducer.reduce()
        opt.step()
        return loss

    def step_graph():
        return captured.step(build_graph)
    reducer.timing = world > 1
    return dict(step=step_graph if captured is not None else step_eager, step_eager=step_eager, units=units, nodes=g.N,
                edges=g.E, text=text, model_name=W["model"], captured=captured, reducer=reducer)


def cpu_baseline_other(args):
    """MD17 / OC20: the oracle's train step on the host cores at the reference script's batch (8 / 5 frames; OC20: a
    bounded sample of 2 of the 16 structures -- the dense CPU neighbour search and 26 k periodic edges per 16 structures
    make one full step minutes long)."""
    import statistics
    from equiformer_amd.synthetic import md17_aspirin_batch
    from oracle import nets as onets, pbc
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    W = WORKLOADS[args.workload]
    if args.workload.startswith("md17"):
        frames = W["frames"]
        model = onets.model_entrypoint(W["model"])("64x0e", 5.0, num_basis=32).train()
        d = md17_aspirin_batch(frames, seed=0)
        ty, tf = torch.randn(frames, 1), torch.randn(frames * 21, 3)
        opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-6)

        def step():
            opt.zero_grad()
            E, F = model(d["z"], d["pos"], d["batch"])
            ((E - ty).abs().mean() + W["wf"] * (F - tf).norm(dim=1).mean()).backward()
            opt.step()
        units, what = frames, "%d aspirin frames (the reference script's batch)" % frames
    else:
        B = 2
        model = onets.oc20_l1_256_nonlinear().train()
        data, y = _oc20_batch(B, W["atoms"], 0)
        ei, coff, nb = pbc.radius_graph_pbc(data.pos, data.cell, [W["atoms"]] * B, 5.0, 500)
        _, _, off = pbc.get_pbc_distances(data.pos, ei, data.cell, coff, nb)
        opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=1e-3)

        def step():
            opt.zero_grad()
            e = model(data.atomic_numbers, data.tags, data.pos, data.batch, edge_index=ei, offsets=off)
            (e.squeeze() - y).abs().mean().backward()
            opt.step()
        units, what = B, "%d of the 16 structures x %d atoms, %d periodic edges (graph built once, outside the timing)" % (
            B, W["atoms"], ei.shape[1])
    warm = 0 if args.workload.startswith("md17") else 1  # one MD17 step (second-order backward) is ~25 s on 8 cores
    for _ in range(warm):
        step()
    ts, t_begin = [], time.perf_counter()
    while len(ts) < 3 and (not ts or time.perf_counter() - t_begin < 20.0):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    return {"value": units / dt, "unit": W["unit"], "cores": cores, "kind": "port", "s_per_step": dt, "timed_steps": len(ts),
            "warmup_steps": warm,
            "sample": "median of %d train steps of %s; oracle = CPU restatement of the reference, torch fp32, "
                      "%d threads" % (len(ts), what, cores)}

ARITHMETIC = {
    "fp32": "fp32 storage and accumulation everywhere; every contraction on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 / "
            "16x16x4_f32: bit-equal to an fmaf chain)",
    "split": "fp32 storage and accumulation everywhere; the matrix steps of the fused SeparableFCTP (85 % of the flops) and of the "
             "per-degree / radial linears multiply fp32 operands split into bf16 planes (activations 2, weights 3; 5 products) on "
             "v_mfma_f32_32x32x16_bf16: fp32-class results (QM9 model vs fp64 oracle: energies / gradients inside 1e-4 with two "
             "orders of margin; tests/test_gpu_sfcx.py, tests/test_gpu_fullsize.py)",
    "split6": "as split with 3 + 3 planes, 6 products",
    "bf16": "BASELINE config #2: fp32 storage / accumulation, fp32 layer norm, softmax, radial basis (as the reference pins "
            "them under AMP); every matrix step (fused SeparableFCTP, per-degree linears, radial MLP) takes plain bf16 operands on "
            "v_mfma_f32_32x32x16_bf16 with fp32 accumulation (tolerances stated in tests/test_gpu_sfcx.py / test_gpu_fullsize.py)",
}
# Peaks (MI355X_MICROARCH.md chip table).  A kernel is priced against the pipe it ISSUES on: the split modes run `P` bf16 plane
# products per algorithmic product on the dense-bf16 matrix pipe, so that pipe can deliver 2 500 / P TFLOP/s of algorithmic work.
PEAK_BF16_MFMA_TFLOPS = 2500.0
PRODUCTS = {"split": 5, "split6": 6, "bf16": 1}
CLOCK_GHZ, N_SIMD = 2.4, 1024
MFMA_FLOPS_32x32x16 = 2.0 * 32 * 32 * 16   # per v_mfma_f32_32x32x16_bf16
MFMA_CYCLES_32x32x16 = 32.0                # 8 passes of 4 cycles


def kernel_peak(name, mode):
    """(peak TFLOP/s of the pipe the kernel's matrix instructions issue on, plane products per algorithmic product)"""
    on_bf16_pipe = name.startswith(("sfcx", "gemmx")) and mode in PRODUCTS
    if on_bf16_pipe:
        return PEAK_BF16_MFMA_TFLOPS / PRODUCTS[mode], PRODUCTS[mode]
    return PEAK_F32_MFMA_TFLOPS, 0


def _pmc_record():
    """profiles/pmc_dominant.json if (and only if) it was measured on the build that is running now"""
    pmc = os.path.join(ROOT, "profiles", "pmc_dominant.json")
    if not os.path.exists(pmc):
        return None, "no PMC measurement for this build (profiles/pmc_dominant.json)"
    try:
        from equiformer_amd.build import source_hash
        rec = json.load(open(pmc))
        if rec.get("build") == source_hash():
            return rec, "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE (+ SQ_INSTS_MFMA) over bench.py, build " + rec["build"]
        return None, "profiles/pmc_dominant.json belongs to build %s, running %s" % (rec.get("build"), source_hash())
    except Exception as exc:  # a malformed file must not take the bench line down
        return None, "profiles/pmc_dominant.json unreadable: %r" % (exc,)


def roofline_of(prof, dt_s, mode, with_pmc=True):
    """Roofline object of the matrix-core kernel with the largest total time in `prof` (HIP-event records of a timed region of
    dt_s seconds).  peak = the pipe the kernel issues on; frac_of_fp32_peak kept as the secondary figure."""
    rec = None
    for name, r in prof.items():
        if r["flops"] > 0 and (rec is None or r["total_ms"] > rec[1]["total_ms"]):
            rec = (name, r)
    if rec is None:
        return None
    name, r = rec
    avg_ms = r["total_ms"] / r["launches"]
    tflops = r["flops"] / r["total_ms"] / 1e9
    peak, products = kernel_peak(name, mode)
    alg_bytes = r["bytes"] / r["launches"]
    out = {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak,
           "peak_is": ("dense bf16 MFMA peak %.0f / %d plane products per algorithmic product" % (PEAK_BF16_MFMA_TFLOPS, products))
                      if products else "exact-fp32 MFMA peak",
           "frac_of_fp32_peak": tflops / PEAK_F32_MFMA_TFLOPS,
           "kernel": name, "launches": r["launches"], "avg_launch_ms": avg_ms,
           "flops_per_launch": r["flops"] / r["launches"], "algorithmic_bytes_per_launch": alg_bytes,
           "share_of_step": r["total_ms"] / (1e3 * dt_s)}
    if products:
        # matrix-pipe occupancy: MFMA instructions x 32 cycles over 1024 SIMDs at 2.4 GHz, against the launch duration
        n_mfma = r["flops"] / r["launches"] * products / MFMA_FLOPS_32x32x16
        out["mfma_busy"] = n_mfma * MFMA_CYCLES_32x32x16 / N_SIMD / (CLOCK_GHZ * 1e9) / (avg_ms * 1e-3)
        out["mfma_busy_source"] = "analytic: flops_per_launch x %d / 32768 instructions x 32 cycles / 1024 SIMDs / 2.4 GHz" % products
    out["traffic"], out["traffic_over_algorithmic"] = None, None
    if with_pmc:
        pmc, note = _pmc_record()
        out["traffic_source"] = note
        if pmc is not None and name in pmc:
            out["traffic"] = pmc[name].get("hbm_bytes_per_launch")
            if out["traffic"]:
                out["traffic_over_algorithmic"] = out["traffic"] / alg_bytes
                # the other bound of the same kernel: counter bytes over the launch duration against the 8 TB/s HBM peak
                out["hbm_frac"] = out["traffic"] / (avg_ms * 1e-3) / (PEAK_HBM_GBPS * 1e9)
            if products and pmc[name].get("mfma_insts_per_launch"):
                out["mfma_busy"] = (pmc[name]["mfma_insts_per_launch"] * MFMA_CYCLES_32x32x16 / N_SIMD / (CLOCK_GHZ * 1e9)
                                    / (avg_ms * 1e-3))
                out["mfma_busy_source"] = "SQ_INSTS_MFMA (PMC pass) x 32 cycles / 1024 SIMDs / 2.4 GHz"
    return out


def measure(args, dev, rank, world, workload, mode, steps, warmup, regions=("sfcx",)):
    """Build `workload`, run `warmup` untimed steps, then one timed region of EXACTLY `steps` steps per entry of `regions`
    (barrier + synchronize on both sides, max over ranks).  An entry is the HIP-event filter of that region: a kernel-name
    substring, "" for every matrix-core kernel, None for no events.  -> (workload dict, [(seconds, event records)], final loss)"""
    from equiformer_amd import lib, ops
    prev = ops.set_matrix_mode(mode)
    a2 = argparse.Namespace(**vars(args))
    a2.workload = workload
    wl = build_workload(a2, dev, rank, world)
    step = wl["step"]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    prewarm = float(getattr(args, "prewarm_s", 0.0) or 0.0)
    if prewarm > 0:
        t_end = time.perf_counter() + prewarm
        while True:
            step()
            torch.cuda.synchronize()
            go = time.perf_counter() < t_end
            if world > 1:  # every rank runs the SAME number of steps (a step holds collectives): stop when the first rank is done
                flag = torch.tensor([1.0 if go else 0.0], device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                go = flag.item() > 0.0
            if not go:
                break
    for _ in range(warmup):
        step()
    regions = [r for r in regions if r != "graph" or wl["captured"] is not None]
    step_eager = wl["step_eager"]
    if "auto" in regions:
        # one more untimed step with events on the SeparableFCTP kernels picks the dominant one; the timed region then carries
        # events on THAT kernel only (every event pair is a dependency between consecutive launches: events on all 39
        # SeparableFCTP launches of a QM9 step cost ~2 % of the step, `spread` [0] vs [2] of round 4)
        lib.prof_enable("sfc")  # sfcx_* (split / bf16 modes) and sfc_* (fp32 mode)
        step_eager()
        torch.cuda.synchronize()
        first = lib.prof_report()
        lib.prof_enable(None)
        regions[regions.index("auto")] = max(first, key=lambda k: first[k]["total_ms"]) if first else ""
    out = []
    loss = None
    for flt in regions:
        # "graph": the step replayed as one HIP graph (no per-launch events possible: the library's entry points are not called);
        # every other region launches eagerly, with HIP events on the kernels whose name contains `flt` (None: no events)
        run = step if flt == "graph" else step_eager
        lib.prof_enable(None if flt == "graph" else flt)
        ops.deferred_weight_gradient_stats(reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = run()
        barrier()
        dt = time.perf_counter() - t0
        prof = lib.prof_report()
        lib.prof_enable(None)
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            # what the data-parallel step adds, per rank: reduce() = head collective + the part of the tail collective that
            # backward did not hide; and the shard sizes the ranks actually had
            ar = wl["reducer"].reduce_ms()
            info = torch.tensor([float(wl["edges"]), float(wl["nodes"]), -1.0 if ar is None else ar], device=dev, dtype=torch.float64)
            allinfo = [torch.zeros_like(info) for _ in range(world)]
            torch.distributed.all_gather(allinfo, info)
            wl["per_rank"] = {"edges_per_gpu": [int(v[0].item()) for v in allinfo], "nodes_per_gpu": [int(v[1].item()) for v in allinfo],
                              "allreduce_ms": [round(v[2].item(), 4) for v in allinfo],
                              "n_ranks_seen": torch.distributed.get_world_size()}
        out.append((t.item(), prof, flt))
        wl["deferred_weight_gradients"] = dict(ops.deferred_weight_gradient_stats(), steps=steps)
    loss = float(loss.item())
    ops.set_matrix_mode(prev)
    return wl, out, loss


def sub_record(args, dev, workload, mode, steps=10, warmup=3):
    """Compact record of another BASELINE configuration measured in the same process (one GPU, no CPU leg)."""
    t0 = time.perf_counter()
    try:
        a2 = argparse.Namespace(**vars(args))
        a2.batch, a2.atoms, a2.side = 128, 18, 6.5
        # HIP events on the dominant SeparableFCTP kernel only, as in the headline region: the MD17 steps are launch-bound, and an
        # event pair around each of their ~500 matrix-core launches cost the round-4 sub-records 10-15 % (standalone 462 frames/s
        # vs 399 in the sub-record, profiles/r05)
        # (QM9 records: the value from the HIP-graph region, the dominant kernel from the eager region behind it, as in the headline)
        wl, regs, loss = measure(a2, dev, 0, 1, workload, mode, steps, warmup, regions=("graph", "auto"))
        dt = regs[0][0]
        dt_rf, prof, _ = regs[-1]
        rec = {"workload": workload, "matrix_mode": mode, "dtype": "bf16" if mode == "bf16" else "f32",
               "model": wl["model_name"], "value": wl["units"] * steps / dt, "unit": WORKLOADS[workload]["unit"],
               "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup, "units_per_step": wl["units"],
               "nodes": wl["nodes"], "edges": wl["edges"], "final_loss": loss, "what": wl["text"],
               "hip_graph": regs[0][2] == "graph"}
        if len(regs) > 1:
            rec["ms_per_step_eager"] = 1e3 * dt_rf / steps
        rf = roofline_of(prof, dt_rf, mode, with_pmc=False)
        if rf:
            rec["dominant"] = {k: rf[k] for k in ("kernel", "launches", "avg_launch_ms", "achieved", "peak", "frac", "unit",
                                                  "share_of_step") if k in rf}
            if "mfma_busy" in rf:
                rec["dominant"]["mfma_busy"] = rf["mfma_busy"]
        rec["wall_s"] = time.perf_counter() - t0
        return rec
    except Exception as exc:  # a sub-record must not take the headline line down
        return {"workload": workload, "matrix_mode": mode, "error": repr(exc)[:300]}
    finally:
        torch.cuda.empty_cache()


def _self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-run this command as N ranks under torch.distributed.run (one process per
    GPU, 127.0.0.1 rendezvous on a free port); rank 0 of the child job prints the JSON line on this process's stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL / cross-process HIP memory)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.cpu_baseline_only:  # the background leg of the default run: no GPU involved
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_spawn(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # test hooks (not used by the driver): EQF_BENCH_DEVICE pins every rank to one GPU and EQF_BENCH_BACKEND=gloo lets
    # the N > 1 code path run on a single-GPU box
    dev_index = int(os.environ.get("EQF_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("EQF_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)  # RCCL / xGMI
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from equiformer_amd import lib, ops
    lib.load()
    ops._overlap_wgrad[0] = args.overlap_wgrad
    cpu_proc = None
    # (a loss.backward() training loop: the node-row weight gradients go out in a few grouped launches when backward ends --
    # the library default, equiformer_amd/ops.py; FlatGradAllReduce's tail hook flushes what is queued before its collective,
    # so the N > 1 step is the same step.  `config.deferred_weight_gradients` records what the timed region did.)

    # Region 1 is THE timed region of the contract (W warm-up steps, then exactly K steps): HIP events on the launches of the
    # dominant kernel only (picked among the SeparableFCTP kernels by one profiled, untimed step after the warm-up; `--dominant`
    # overrides).  Regions 2 and 3 repeat the same K steps for the
    # run-to-run spread: 2 with events on every matrix-core kernel (figures of the other kernels), 3 with no events at all.
    # One GPU, QM9: region 1 replays the step as ONE HIP graph (equiformer_amd/capture.py) -- that is `value`; the per-kernel
    # figures of `roofline` then come from region 2, the same K steps launched eagerly with events on the dominant kernel (a
    # replayed launch cannot carry host-side events; a kernel's duration does not depend on how it was launched, and the
    # rocprofv3 kernel trace of this command, profiles/r06, sees the replayed launches).
    first = args.dominant if args.dominant else "auto"
    regions = (first, "", None) if args.repeats >= 3 else ((first, "") if args.repeats == 2 else (first,))
    regions = ("graph",) + regions  # (dropped by measure() where the step is not captured)
    wl, regs, loss = measure(args, dev, rank, world, args.workload, args.matrix_mode, args.steps, args.warmup, regions)
    graphed = regs[0][2] == "graph"
    dt = regs[0][0]
    irf = 1 if (graphed and len(regs) > 1) else 0  # the region whose events feed `roofline`
    dt_rf, prof, flt0 = regs[irf]
    n_nodes, n_edges = wl["nodes"], wl["edges"]

    if rank == 0:
        out = {
            "metric": WORKLOADS[args.workload]["metric"],
            "value": wl["units"] * world * args.steps / dt,
            "unit": WORKLOADS[args.workload]["unit"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if args.matrix_mode == "bf16" else "f32",
            "data": "synthetic",
            "config": {
                "workload": wl["text"],
                "global_batch": wl["units"] * world, "nodes_per_gpu": n_nodes, "edges_per_gpu": n_edges,
                "edges_per_unit": n_edges / wl["units"], "parallelism": "dp%d" % world,
                "final_loss": loss,
                "matrix_mode": args.matrix_mode,
                "prewarm_s": args.prewarm_s,
                "arithmetic": ARITHMETIC[args.matrix_mode],
                # node-row weight gradients queued during backward and launched in groups (library default; with N > 1 the
                # reducer's hook flushes them before its collective): problems queued / grouped launches in the last region
                "deferred_weight_gradients": wl.get("deferred_weight_gradients"),
            },
        }
        if world > 1 and wl.get("per_rank"):
            pr = wl["per_rank"]
            out["n_ranks_seen"] = pr["n_ranks_seen"]
            out["allreduce_ms"] = max(pr["allreduce_ms"])  # per step: head collective + exposed wait on the tail, slowest rank
            out["config"]["per_rank"] = pr
        vals = [wl["units"] * world * args.steps / d for d, _, _ in regs]
        what = {"graph": "the step replayed as one HIP graph", "": "eager launches, events on every matrix-core launch",
                None: "eager launches, no events"}
        out["spread"] = {"values": vals, "min": min(vals), "max": max(vals),
                         "regions": [what.get(f, "eager launches, HIP events on the launches of `%s`" % f) for _, _, f in regs],
                         "note": "the same %d steps timed %d times back to back in this process; [0] = `value`" % (args.steps, len(vals))}
        out["config"]["hip_graph"] = bool(graphed)
        if graphed and wl.get("captured") is not None:
            out["config"]["hip_graph_replays"] = wl["captured"].replays
            out["config"]["hip_graph_eager_steps"] = wl["captured"].eager_steps
        rf = roofline_of(prof, dt_rf, args.matrix_mode)
        if rf is not None and graphed:
            rf["measured_in"] = ("region [%d] of `spread`: the same %d steps launched eagerly with HIP events on this kernel's launches "
                                 "(%.3f ms/step there); `value` is region [0], where the step is one graph launch" % (irf, args.steps, 1e3 * dt_rf / args.steps))
        allprof = regs[irf + 1][1] if len(regs) > irf + 1 else prof
        if rf is not None:
            out["roofline"] = rf
            # the other matrix-core kernels of the step, same accounting (second region; not part of the contract)
            others = []
            for n2, r2 in sorted(allprof.items(), key=lambda kv: -kv[1]["total_ms"]):
                if n2 == rf["kernel"] or r2["flops"] <= 0:
                    continue
                pk, _ = kernel_peak(n2, args.matrix_mode)
                tf = r2["flops"] / r2["total_ms"] / 1e9
                others.append({"kernel": n2, "launches": r2["launches"], "avg_launch_ms": r2["total_ms"] / r2["launches"],
                               "achieved": tf, "peak": pk, "frac": tf / pk,
                               "share_of_step": r2["total_ms"] / (1e3 * regs[irf + 1][0] if len(regs) > irf + 1 else 1e3 * dt_rf)})
            out["roofline"]["others"] = others[:8]
        # the two kernels BASELINE.json's north_star asks to be stated against the gfx950 peaks (not part of the contract
        # line): HBM GB/s of the per-destination softmax + scatter (algorithmic bytes 1 940 E + 1 920 N per call,
        # SURVEY 8d) and MFMA rate of the radial MLP's widest layer (E x 64 -> 960)
        extra = {}
        sc = allprof.get("attn_fwd")
        if sc:
            byts = 1940.0 * n_edges + 1920.0 * n_nodes
            gbps = byts * sc["launches"] / sc["total_ms"] / 1e6
            extra["scatter"] = {"kernel": "attn_fwd (segment softmax + aggregation)", "bound": "hbm", "achieved": gbps,
                                "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                                "avg_launch_ms": sc["total_ms"] / sc["launches"], "algorithmic_bytes_per_launch": byts}
            pmc, note = _pmc_record()
            if pmc is not None and "attn_fwd" in pmc:  # rocprofv3 FETCH_SIZE / WRITE_SIZE of this build (tools/gpu_profile.sh)
                tb = pmc["attn_fwd"]["hbm_bytes_per_launch"]
                extra["scatter"].update({"traffic": tb, "traffic_over_algorithmic": tb / byts,
                                         "counter_gbps": tb * sc["launches"] / sc["total_ms"] / 1e6,
                                         "counter_frac": tb * sc["launches"] / sc["total_ms"] / 1e6 / PEAK_HBM_GBPS,
                                         "traffic_source": note})
        # the widest E-row GEMM with a memory A operand and an [N, K] weight = the radial MLP's last layer
        rad = [(n2, r2) for n2, r2 in allprof.items()
               if (n2.startswith("gemm_rows_") and n2.endswith("_mem_nk")) or n2 in ("gemmx_group_nk_edge", "gemm_group_nk")]
        if rad:
            n2, r2 = max(rad, key=lambda kv: kv[1]["flops"] / kv[1]["launches"])
            tf = r2["flops"] / r2["total_ms"] / 1e9
            pk, _ = kernel_peak(n2, args.matrix_mode)
            extra["radial_mlp"] = {"kernel": n2 + " (edge-row nn.Linear launches of the radial bank: 128 -> G x 64, G x (64 -> 64), "
                                             "G x (64 -> 960); the last is 94 % of their flops)", "bound": "mfma", "achieved": tf,
                                   "peak": pk, "unit": "TFLOP/s", "frac": tf / pk, "frac_of_fp32_peak": tf / PEAK_F32_MFMA_TFLOPS,
                                   "avg_launch_ms": r2["total_ms"] / r2["launches"]}
            pmc, note = _pmc_record()
            if pmc is not None and "gemmx_rows_wide" in pmc:
                # the last layer's own kernel (E x 64 -> 7 x 960): SQ_INSTS_MFMA per launch over its kernel-trace duration
                w = pmc["gemmx_rows_wide"]
                extra["radial_mlp"]["widest_layer_kernel"] = {
                    "kernel": "gemmx_rows_wide_kernel", "mfma_insts_per_launch": w.get("mfma_insts_per_launch"),
                    "avg_us_kernel_trace": w.get("avg_us_kernel_trace"), "mfma_busy": w.get("mfma_busy"),
                    "hbm_bytes_per_launch": w.get("hbm_bytes_per_launch"), "traffic_source": note}
        # whole step against the HBM roofline of SURVEY 8d: 3.0 MB algorithmic bytes per molecule-step at E = 200
        if args.workload == "qm9":
            b_alg = 6 * (172800.0 + 1112.0 * n_edges / args.batch) + 0.29e6 + 0.33e6
            extra["step_hbm_roofline"] = {"algorithmic_bytes_per_molecule": b_alg,
                                          "molecules_per_s_at_peak": PEAK_HBM_GBPS * 1e9 / b_alg * world,
                                          "frac": out["value"] / (PEAK_HBM_GBPS * 1e9 / b_alg * world)}
        out["north_star_kernels"] = extra
        print("[bench] gpu part done: %.1f %s, %.2f ms/step" % (out["value"], out["unit"], out["ms_per_step"]),
              file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline and args.workload == "qm9":
            # the CPU-oracle leg as a background process (its own interpreter, --cpu-threads threads): it runs beside the GPU
            # sub-records below, the headline figures above are already measured
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--batch", str(args.batch), "--atoms", str(args.atoms),
                   "--side", str(args.side), "--cpu-molecules", str(args.cpu_molecules), "--cpu-steps", str(args.cpu_steps),
                   "--cpu-threads", str(args.cpu_threads)] + ([] if args.cpu_full_batch else ["--no-cpu-full-batch"])
            cpu_proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        # the other BASELINE configurations, measured in this same run (one GPU; the headline workload only)
        if world == 1 and args.sub_records and args.workload == "qm9" and args.matrix_mode == "split":
            del wl, regs
            torch.cuda.empty_cache()
            subs = []
            # (qm9 / fp32: exact-fp32 MFMA in every matrix step -- the like-for-like figure against the reference's --no-amp)
            for wname, mode in (("qm9", "bf16"), ("qm9", "fp32"), ("oc20", "split"), ("md17_l2", "split"), ("md17_l3", "split")):
                subs.append(sub_record(args, dev, wname, mode))
                print("[bench] sub-record %s/%s: %s" % (wname, mode, {k: subs[-1].get(k) for k in ("value", "ms_per_step", "error")}),
                      file=sys.stderr, flush=True)
            out["configs"] = subs
        if cpu_proc is not None:
            try:
                txt, _ = cpu_proc.communicate(timeout=600)
                out["cpu_baseline"] = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])
            except Exception as exc:  # the leg must not take the GPU line down
                cpu_proc.kill()
                out["cpu_baseline"] = {"error": repr(exc)[:200]}
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_other(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
