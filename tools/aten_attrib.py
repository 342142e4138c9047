#!/usr/bin/env python
"""Development aid: which Python lines launch the ATen (non-HIP-library) kernels of one QM9 train step."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::cat",
         "aten::index", "aten::sort", "aten::argsort", "aten::bincount", "aten::cumsum", "aten::div", "aten::sub",
         "aten::neg", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::item", "aten::_local_scalar_dense")
agg = collections.Counter()
for ev in prof.events():
    if ev.name in WATCH and ev.device_type == torch.autograd.DeviceType.CPU:
        frames = [f for f in (ev.stack or []) if "equiformer_amd" in f or "bench.py" in f or "aten_attrib" in f]
        where = frames[0].strip() if frames else "(autograd engine / no python frame)"
        agg[(ev.name, where)] += 1
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print("%4d  %-28s %s" % (n, name, where[-110:]))
