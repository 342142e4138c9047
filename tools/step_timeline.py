#!/usr/bin/env python
"""Timeline of ONE train step from a rocprofv3 (rocpd / sqlite) kernel trace: every launch of the last complete step in
order, with its duration, the idle gap before it and its grid -- the view that shows which small kernels and which gaps make
up the part of the step that is not the three SeparableFCTP kernels.
   python tools/step_timeline.py <results.db> [--marker adamw_kernel] > timeline.txt
A step is delimited by two consecutive launches of the marker kernel (the optimizer runs once per step)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    db = sys.argv[1]
    marker = sys.argv[sys.argv.index("--marker") + 1] if "--marker" in sys.argv else "adamw_kernel"
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    extra = [x for x in ("grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "lds_size", "stream_id", "queue_id") if x in cols]
    rows = c.execute("select %s, start, end%s from kernels order by start" % (namecol, "".join(", " + e for e in extra))).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        print("marker kernel %r seen %d times: no complete step" % (marker, len(marks)))
        return
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = rows[lo:hi]
    t0 = rows[lo - 1][2]
    print("# one step: %d launches, wall %.1f us (end of the previous optimizer kernel -> end of this one)" % (len(step), (step[-1][2] - t0) / 1e3))
    print("# %5s %9s %8s %7s  %-70s %s" % ("idx", "start_us", "dur_us", "gap_us", "kernel", " ".join(extra)))
    prev_end = t0
    busy = gaps = 0.0
    for i, r in enumerate(step):
        n, s, e = r[0], r[1], r[2]
        gap = (s - prev_end) / 1e3
        print("  %5d %9.1f %8.2f %7.2f  %-70s %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, gap, short(n), " ".join(str(v) for v in r[3:])))
        busy += (e - s) / 1e3
        gaps += max(gap, 0.0)
        prev_end = max(prev_end, e)
    print("# kernel time %.1f us, idle gaps %.1f us" % (busy, gaps))


if __name__ == "__main__":
    main()
