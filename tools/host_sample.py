#!/usr/bin/env python
"""Sampling profile of ALL Python threads of a train step (the autograd engine runs the Python backward functions on its own
thread, which cProfile does not see):   python tools/host_sample.py md17_l2 [steps]
Every ~0.5 ms the innermost frames of every thread are recorded; prints the share of samples per (file:function) -- inclusive
over the 6 innermost frames -- and per innermost frame."""
import argparse
import collections
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl_name = sys.argv[1] if len(sys.argv) > 1 else "md17_l2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
args = argparse.Namespace(batch=128, atoms=18, side=6.5, workload=wl_name)
dev = torch.device("cuda:0")
from equiformer_amd import lib  # noqa: E402
lib.load()
wl = bench.build_workload(args, dev, 0, 1)
step = wl["step"]
for _ in range(5):
    step()
torch.cuda.synchronize()
inner, incl = collections.Counter(), collections.Counter()
stop = [False]
me = threading.get_ident()
nsamp = [0]


def sampler():
    sid = threading.get_ident()
    while not stop[0]:
        for tid, fr in sys._current_frames().items():
            if tid == sid:
                continue
            seen, depth, f = set(), 0, fr
            while f is not None and depth < 8:
                key = "%s:%s" % (os.path.basename(f.f_code.co_filename), f.f_code.co_name)
                if depth == 0:
                    inner[key] += 1
                if key not in seen:
                    incl[key] += 1
                    seen.add(key)
                f = f.f_back
                depth += 1
        nsamp[0] += 1
        time.sleep(0.0005)


th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop[0] = True
th.join()
print("%s: %.2f ms/step under sampling, %d samples" % (wl_name, 1e3 * dt / steps, nsamp[0]))
print("--- innermost frame (share of samples; two threads -> up to 200 %)")
for k, v in inner.most_common(25):
    print("%6.1f %%  %s" % (100.0 * v / nsamp[0], k))
print("--- inclusive over the 8 innermost frames")
for k, v in incl.most_common(40):
    print("%6.1f %%  %s" % (100.0 * v / nsamp[0], k))
