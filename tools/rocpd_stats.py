#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd / sqlite) kernel trace: per-kernel launches, total / average duration, share.
   python tools/rocpd_stats.py <results.db> [--csv out.csv] [--top N]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:110]


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, start, end from kernels" % namecol).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    lines = ["%-112s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        lines.append("%-112s %8d %12.1f %10.2f %6.2f" % (k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
    lines.append("TOTAL kernel time %.1f us over %d launches" % (tot, len(rows)))
    print("\n".join(lines))
    if "--csv" in sys.argv:
        with open(sys.argv[sys.argv.index("--csv") + 1], "w") as f:
            f.write("kernel,calls,total_us,avg_us,pct\n")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write('"%s",%d,%.1f,%.3f,%.3f\n' % (k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot))


if __name__ == "__main__":
    main()
