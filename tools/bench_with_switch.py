#!/usr/bin/env python
"""Development aid: bench.py with a development switch of the SeparableFCTP kernels set first, e.g.
   python tools/bench_with_switch.py 64 --steps 20 --warmup 5 --no-cpu-baseline     (64 = split-precision forward step)"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()  # the HIP runtime must be up (torch's device context) before the library is touched
torch.zeros(1, device="cuda:%d" % int(os.environ.get("LOCAL_RANK", 0)))
from equiformer_amd import lib  # noqa: E402

mask = int(sys.argv[1])
lib.load().eqf_sfc_debug_exp(mask)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
