#!/usr/bin/env python
"""Where the HOST time of a train step goes (cProfile over a few steps of a bench workload):
   python tools/host_profile.py md17_l2 [steps]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl_name = sys.argv[1] if len(sys.argv) > 1 else "md17_l2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
args = argparse.Namespace(batch=int(os.environ.get("EQF_HP_BATCH", "128")), atoms=18, side=6.5, workload=wl_name)
dev = torch.device("cuda:0")
from equiformer_amd import lib  # noqa: E402
lib.load()
wl = bench.build_workload(args, dev, 0, 1)
step = wl["step"]
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("%s: %.2f ms/step wall, host issue time %.2f ms/step (the host %s the GPU)" % (
    wl_name, 1e3 * t_all / steps, 1e3 * t_issue / steps, "is behind" if t_issue > 0.9 * t_all else "runs ahead of"))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
