#!/usr/bin/env python
"""Compact per-kernel summary of rocprofv3 --pmc counter_collection CSVs (one or more passes).

Usage: python tools/pmc_summary.py OUT.csv PASS1_counter_collection.csv [PASS2 ...]
Keeps only the kernels of libgoliath_hip.so (anonymous-namespace kernels named *_kernel), averages every counter
over the dispatches of a kernel, and writes one row per kernel with one column per counter (+ dispatch count, grid,
VGPRs).  The raw CSVs (hundreds of MB of torch kernel names) never leave the GPU box.
"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main(out, *paths):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for p in paths:
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if not re.search(r"_kernel(<|$)", k) or "at::" in k or "rccl" in k.lower():
                    continue
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
    counters = sorted({c for k in vals for c in vals[k]})
    with open(out, "w") as o:
        o.write("kernel,dispatches,grid,workgroup,vgpr,sgpr,lds," + ",".join(counters) + "\n")
        for k in sorted(vals):
            n = max(len(v) for v in vals[k].values())
            row = [f'"{k}"', str(n), *meta[k]]
            for c in counters:
                v = vals[k].get(c)
                row.append("" if not v else f"{sum(v) / len(v):.1f}")
            o.write(",".join(row) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
