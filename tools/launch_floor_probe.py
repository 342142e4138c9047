#!/usr/bin/env python
"""What does ONE launch of the grouped node-row GEMM cost when it has almost nothing to do?  Dependent back-to-back launches of
   (a) one 64 x 64 x 32 problem, (b) the real 480 -> 480 per-degree linear at 2 304 nodes, (c) a trivial torch kernel,
   (d) the layer norm at 2 304 rows -- us per launch on the GPU (HIP events around 200 launches)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import ops  # noqa: E402
from equiformer_amd.layout import RowLayout  # noqa: E402

dev = torch.device("cuda:0")
st = ops._stream


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


x = torch.randn(64, 32, device=dev)
w = torch.randn(32, 64, device=dev)
y = torch.empty(64, 64, device=dev)
tiny = [ops._desc(0, (x, 0), ops.rows(1, 32, 0), (w, 0), 64, (y, 0), ops.rows(1, 64, 0), None, 64, 64, 32)]
print("one 64 x 64 x 32 problem              %6.2f us / launch" % timeit(lambda: ops._gemm_group(tiny, st())))
irr = "128x0e+64x1e+32x2e"
spec = ops.LinearSpec(RowLayout(irr), RowLayout(irr))
if spec is not None:
    xn = torch.randn(2304, 480, device=dev)
    wn = torch.randn(spec.weight_numel, device=dev)
    print("480 -> 480 per-degree linear, 2304 rows %6.2f us / launch" % timeit(lambda: ops._lin_fwd(xn, wn, None, spec)))
    for mode in ("fp32", "bf16"):
        with ops.matrix_mode(mode):
            print("  the same in matrix mode %-5s          %6.2f us / launch" % (mode, timeit(lambda: ops._lin_fwd(xn, wn, None, spec))))
z = torch.zeros(256, device=dev)
print("torch add_ on 256 floats               %6.2f us / launch" % timeit(lambda: z.add_(1.0)))
