#!/usr/bin/env python
"""Static instruction mix of the fused SeparableFCTP kernels (whole kernel bodies, all template branches) from the
gfx950 ISA that hipcc emits -- the VALU : MFMA ratio that DESIGN.md 3.1 argues about.  CPU-only:
    python tools/isa_mix.py > profiles/<tag>_sfc_isa_mix.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "equiformer_amd", "csrc", "sfc.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "sfc.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "--cuda-device-only", "-S",
                           src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()

kernels = re.findall(r"^(_ZN[^\n:]*sfc_[a-z]+_kernel[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M)


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return None


print("%-44s %7s %7s %7s %7s %7s %8s %9s" % ("kernel", "mfma", "valu", "lds", "vmem", "salu", "waitcnt", "valu/mfma"))
for name, body in kernels:
    c = collections.Counter()
    kinds = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith((";", ".", "_")) or line.endswith(":"):
            continue
        op = line.split()[0]
        k = cls(op)
        if k:
            c[k] += 1
            if k == "mfma":
                kinds[op] += 1
    short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    short = re.sub(r"EvNS_.*", "", short).replace("ILi", "<").replace("ELb", ",").replace("E", "")
    print("%-44s %7d %7d %7d %7d %7d %8d %9.1f   %s"
          % (short[:44], c["mfma"], c["valu"], c["lds"], c["vmem"], c["salu"], c["waitcnt"],
             c["valu"] / max(c["mfma"], 1), dict(kinds)))
