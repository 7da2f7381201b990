#!/usr/bin/env python
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `bench.py --micro 1`) into profiles/traffic.json:
HBM-side bytes per 8-view launch of every C-ABI call = 2 * FETCH_SIZE + WRITE_SIZE (FETCH_SIZE is doubled on gfx950,
MI355X_MICROARCH.md; calibrated on the streaming L1-loss kernel whose reads are known exactly, and -- round 5,
tools/probe/gather_probe.hip, profiles/r05_gather_calibration.txt -- on random 16- / 64- / 128-byte gathers: each fetches
ONE 128-byte line, tallied at 64 B, so the factor 2 holds for the gather kernels and a 64-byte record costs 128 B of traffic).
Usage: python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out_prefix"""
import collections
import csv
import json
import sys

CALLS = {  # kernel-name prefix -> ABI call
    "shade_fwd_kernel": "gol_shade_fwd", "shade_bwd_kernel": "gol_shade_bwd", "project_fwd_kernel": "gol_project_fwd",
    "project_bwd_kernel": "gol_project_bwd", "count_lds_kernel": "gol_bin_sort", "count_kernel": "gol_bin_sort",
    "scan_kernel": "gol_bin_sort", "scatter_lds_kernel": "gol_bin_sort", "scatter_kernel": "gol_bin_sort",
    "sort_kernel": "gol_bin_sort", "sort_big_kernel": "gol_bin_sort", "raster_fwd_kernel": "gol_rasterize_fwd", "raster_bwd_kernel": "gol_rasterize_bwd",
    "l1_kernel<false>": "gol_l1_fwd", "l1_kernel<true>": "gol_l1_bwd",
}


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[n].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}  # mean per dispatch


def main(fetch_csv, write_csv, prefix):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    rows, calls = [], collections.defaultdict(float)
    for k in sorted(set(f) | set(w)):
        call = next((c for p, c in CALLS.items() if k.startswith(p)), None)
        if call is None:
            continue
        fk, wk = f.get(k, 0.0) * 1024.0, w.get(k, 0.0) * 1024.0   # the counters are in KiB
        rows.append((k, fk / 1e6, 2 * fk / 1e6, wk / 1e6))
        calls[call] += 2 * fk + wk
    with open(prefix + "_hbm_traffic_pmc.csv", "w") as o:
        o.write("kernel,fetch_MB_raw,fetch_MB_x2,write_MB\n")
        for r in rows:
            o.write('"%s",%.2f,%.2f,%.2f\n' % r)
    calls["_note"] = ("bytes per launch (8 views, bench.py --micro 1) = 2*FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc "
                      "passes (%s_hbm_traffic_pmc.csv); FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated on the "
                      "streaming l1 kernels: raw FETCH = half of the bytes read); counts L2->fabric requests, i.e. "
                      "includes Infinity-Cache hits and every memory-side float atomic" % prefix.split("/")[-1])
    json.dump(calls, open("profiles/traffic.json", "w"), indent=1)
    for r in rows:
        print("%-40s fetch x2 %9.1f MB  write %9.1f MB" % (r[0][:40], r[2], r[3]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
