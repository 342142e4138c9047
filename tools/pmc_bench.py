#!/usr/bin/env python
"""HBM traffic of the matrix-core kernels of bench.py's train step from two rocprofv3 PMC passes over bench.py itself:

    rocprofv3 --pmc FETCH_SIZE -d <dir>/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d <dir>/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc SQ_INSTS_MFMA -d <dir>/pmc_mfma -- python bench.py ...   (optional: matrix instructions per launch -> mfma_busy)
    python tools/pmc_bench.py <dir> [profiles/pmc_dominant.json]

Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as is.  The result is
the MEAN over every launch of the kernel in the run (a train step launches 6 sep_act-, 6 sep_value- and 1
embedding-shaped SeparableFCTP kernels of each kind), keyed by the names bench.py's HIP-event timers use, and stamped
with the hash of the HIP sources it was measured on (bench.py reports traffic only for a matching build)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from equiformer_amd.build import source_hash  # noqa: E402

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_dominant.json")
PROF = {"sfc_fwd_kernel": "sfc_fwd", "sfc_bwd_kernel": "sfc_bwd_data", "sfc_wgrad_kernel": "sfc_wgrad",
        "sfcx_fwd_kernel": "sfcx_fwd", "sfcy_fwd_kernel": "sfcx_fwd",  # (the multi-wave forward of round 6 reports under the same timer)
        "sfcx_bwd_kernel": "sfcx_bwd_data", "sfcx_wgrad_kernel": "sfcx_wgrad", "sfcw_wgrad_kernel": "sfcx_wgrad",
        # the two kernels BASELINE.json's north_star names: per-destination softmax + scatter, widest radial-MLP layer
        "attn_fwd_half_kernel": "attn_fwd", "attn_fwd_kernel": "attn_fwd", "gemmx_rows_wide_kernel": "gemmx_rows_wide"}
# average durations of the same kernels from the kernel trace of the SAME build (tools/gpu_profile.sh writes kernel_stats.csv
# next to the PMC directories): lets bench.py state MFMA utilisation of the radial kernel from its counter alone
avg_us = {}
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        for needle, name in PROF.items():
            if needle in r["kernel"]:
                a = avg_us.setdefault(name, [0.0, 0])
                a[0] += float(r["total_us"])
                a[1] += int(r["calls"])
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for needle, name in PROF.items():
            if needle in k:
                vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"build": source_hash(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over bench.py",
       "note": "bytes per launch = 2 x FETCH_SIZE KiB (gfx950 wide-read correction) + WRITE_SIZE KiB, mean over all launches"}
for name, c in vals.items():
    f, w = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
    if not f or not w:
        continue
    fetch = 2 * 1024 * sum(f) / len(f)
    write = 1024 * sum(w) / len(w)
    res[name] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "launches_seen": len(f)}
    m = c.get("SQ_INSTS_MFMA", [])
    if m:
        res[name]["mfma_insts_per_launch"] = sum(m) / len(m)
    if name in avg_us and avg_us[name][1]:
        res[name]["avg_us_kernel_trace"] = avg_us[name][0] / avg_us[name][1]
        if m:  # 32 cycles per v_mfma_f32_32x32x16_bf16, 1 024 SIMDs at 2.4 GHz
            res[name]["mfma_busy"] = res[name]["mfma_insts_per_launch"] * 32.0 / 1024 / 2.4e9 / (res[name]["avg_us_kernel_trace"] * 1e-6)
    print("%-14s %.1f MB / launch (fetch %.1f + write %.1f, %d launches)" % (name, (fetch + write) / 1e6, fetch / 1e6,
                                                                          write / 1e6, len(f)))
json.dump(res, open(out, "w"), indent=1)
