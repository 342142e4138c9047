#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes over tools/bench_sfc.py: mean counter value per launch of every sfc / sfcx kernel.
   python tools/pmc_sfc.py <dir with pass*/ **/*counter_collection.csv>"""
import collections
import csv
import glob
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sfc" not in k or "pack" in k:
            continue
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:30]
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(vals):
    print("%-30s %s" % (name, "  ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(vals[name].items()))))
