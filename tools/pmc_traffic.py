#!/usr/bin/env python
"""Turn rocprofv3 PMC passes over tools/bench_sfc.py (one pass with FETCH_SIZE, one with WRITE_SIZE) into
profiles/pmc_dominant.json: HBM bytes per launch of the fused SeparableFCTP kernels.

    python tools/pmc_traffic.py <dir with pmc_fetch/ and pmc_write/> [out.json]

Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.
bench_sfc.py launches the sep_act shape first and the sep_value shape second, so the first half of a kernel's rows
belongs to sep_act; bench.py's train step runs 6 sep_act-, 6 sep_value- and 1 embedding-shaped launches, which the
per-launch mean below weights as 6 : 7."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_dominant.json"
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/pmc_*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sfc_" not in k:
            continue
        k = k.split("sfc_")[1].split("(")[0]
        vals[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))


def halves(rows):
    rows = [v for _, v in sorted(rows)]
    h = len(rows) // 2
    a, b = rows[:h], rows[h:]
    return sum(a) / max(len(a), 1), sum(b) / max(len(b), 1)


def kernel_bytes(k):
    fa, fb = halves(vals[k]["FETCH_SIZE"])
    wa, wb = halves(vals[k]["WRITE_SIZE"])
    return {"sep_act": {"fetch": 2 * 1024 * fa, "write": 1024 * wa}, "sep_value": {"fetch": 2 * 1024 * fb, "write": 1024 * wb}}


res = {}
names = {"sfc_fwd": [k for k in vals if k.startswith("fwd_kernel")],
         "sfc_bwd_data": [k for k in vals if k.startswith("bwd_kernel")],
         "sfc_wgrad": [k for k in vals if k.startswith("wgrad_kernel")]}
for prof_name, kernels in names.items():
    tot = {"sep_act": 0.0, "sep_value": 0.0}
    detail = {}
    for k in kernels:
        kb = kernel_bytes(k)
        detail[k] = kb
        for shape in tot:
            tot[shape] += kb[shape]["fetch"] + kb[shape]["write"]
    res[prof_name] = {"hbm_bytes_per_launch": (6 * tot["sep_act"] + 7 * tot["sep_value"]) / 13.0,
                      "per_shape_bytes": tot, "detail": detail,
                      "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes; "
                              "mean over a train step's 6 sep_act- and 7 sep_value-shaped launches"}
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    print("%-14s %.1f MB / launch" % (k, v["hbm_bytes_per_launch"] / 1e6), v.get("per_shape_bytes", ""))
