#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table.
Usage: python tools/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    rows = c.execute(f"select s.{name_col}, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for n, dt in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n or "?")
        n = n.split("(")[0][:90]
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dt
        a[2] = min(a[2], dt)
        a[3] = max(a[3], dt)
    tot = sum(a[1] for a in agg.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{n}",{a[0]},{int(a[1])},{a[1] / a[0]:.0f},{100 * a[1] / tot:.2f},{int(a[2])},{int(a[3])}')
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
