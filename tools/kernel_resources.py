#!/usr/bin/env python
"""Registers, scratch, LDS and occupancy of every kernel of one HIP source, as hipcc reports them for gfx950 (CPU only):
   python tools/kernel_resources.py sfcx.hip [filter] > profiles/<tag>_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd import build as B  # noqa: E402

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(B.CSRC, src), "-o", "/dev/null", "--cuda-device-only",
                                                           "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("%-72s %6s %6s %8s %5s" % ("kernel", "VGPR", "AGPR", "scratch", "occ"))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", n))
    if flt in n:
        print("%-72s %6s %6s %8s %5s" % (n[:72], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
