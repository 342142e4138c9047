#!/usr/bin/env python
"""Development aid: which host-side ops issue the device-to-device copies / fills / small ATen kernels of one QM9 train
step (torch profiler, parent-op chains)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from equiformer_amd import lib, nets  # noqa: E402
from equiformer_amd.synthetic import qm9_like_batch  # noqa: E402

dev = torch.device("cuda:0")
lib.load()
torch.manual_seed(0)
model = nets.model_entrypoint(bench.MODEL)(irreps_in="5x0e", radius=5.0, num_basis=128).to(dev).train()
opt = bench.make_optimizer(model)
d = {k: v.to(dev) for k, v in qm9_like_batch(128, 18, side=6.5, seed=1000).items()}


def step():
    opt.zero_grad(set_to_none=True)
    pred = model(f_in=None, pos=d["pos"], batch=d["batch"], node_atom=d["z"])
    loss = (pred.squeeze() - d["y"]).abs().mean()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()

agg = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    kids = [k.name for k in ev.kernels] if hasattr(ev, "kernels") else []
    if not kids:
        continue
    chain, p = [], ev
    while p is not None and len(chain) < 5:
        chain.append(p.name)
        p = p.cpu_parent
    for k in kids:
        short = k.split("<")[0][:48]
        if "eqf" in short or "sfc_" in short or "gemm_" in short:
            continue
        agg[(short, " <- ".join(chain))] += 1
for (k, chain), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print("%4d  %-48s %s" % (n, k, chain[:150]))
