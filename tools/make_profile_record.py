#!/usr/bin/env python
"""Turn one run of tools/gpu_profile.sh (gpurun_out/TAG/) into the tracked evidence under profiles/:
    profiles/TAG_bench.json, TAG_bench_micro1.json, TAG_kernel_stats.csv, TAG_pmc_traffic.csv, TAG_pmc_sq.csv
    profiles/traffic.json   HBM-side bytes per 8-view launch of every C-ABI call = 2 * FETCH_SIZE + WRITE_SIZE
                            (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md; separate --pmc passes)
    profiles/valu.json      SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES ... per 8-view launch of every kernel
both stamped with the digest of the kernel sources they were measured on (goliath_amd.build.source_digest());
bench.py uses them only while that digest still matches the tree.
Usage: python tools/make_profile_record.py TAG"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLS = {  # kernel-name prefix -> ABI call
    "shade_fwd_kernel": "gol_shade_fwd", "shade_bwd_kernel": "gol_shade_bwd", "sum_views_kernel": "gol_shade_bwd", "project_fwd_kernel": "gol_project_fwd",
    "project_bwd_kernel": "gol_project_bwd", "count_lds_kernel": "gol_bin_sort", "count_kernel": "gol_bin_sort",
    "scan_kernel": "gol_bin_sort", "scatter_lds_kernel": "gol_bin_sort", "scatter_kernel": "gol_bin_sort",
    "sort_kernel": "gol_bin_sort", "sort_mid_kernel": "gol_bin_sort", "sort_queue_kernel": "gol_bin_sort", "sort_big_kernel": "gol_bin_sort", "bin_": "gol_bin_sort",
    "raster_fwd_kernel": "gol_rasterize_fwd", "l1_sum_kernel": "gol_rasterize_fwd", "splat_pack_kernel": "gol_splat_pack",
    "raster_bwd_kernel": "gol_rasterize_bwd", "l1_kernel<false>": "gol_l1_fwd", "l1_kernel<true>": "gol_l1_bwd",
    "tail_conv_fwd_kernel": "gol_tail_conv_fwd", "tail_bias_fwd_kernel": "gol_tail_conv_fwd", "tail_conv_bwd": "gol_tail_conv_bwd",
    "tail_bias_bwd_kernel": "gol_tail_conv_bwd", "ssim_fwd_kernel": "gol_ssim_fwd", "ssim_bwd_kernel": "gol_ssim_bwd",
    "envmap_pack_kernel": "gol_envmap_pack",
}


def call_of(kernel):
    c = next((c for p, c in CALLS.items() if kernel.startswith(p)), None)
    # the shading kernels with the projection fused in belong to gol_shade_project_fwd / bwd: PROJ is the THIRD template
    # argument of shade_fwd_kernel<ENV, RAND, PROJ, VEC4> (round 6) and the LAST of shade_bwd_kernel<ENV, RAND, VEC4, PROJ>
    if c in ("gol_shade_fwd", "gol_shade_bwd") and kernel.startswith("shade_") and "<" in kernel:
        args = [a.strip() for a in kernel[kernel.index("<") + 1:kernel.rindex(">")].split(",")]
        proj = args[2] if c == "gol_shade_fwd" and len(args) == 4 else args[-1]
        if proj == "true":
            c = c.replace("gol_shade_", "gol_shade_project_")
    return c


def main(tag, traffic_name="traffic.json", only_traffic=False):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    sha = open(os.path.join(src, "csrc_sha16.txt")).read().strip()
    head = open(os.path.join(src, "head.txt")).read().strip() if os.path.exists(os.path.join(src, "head.txt")) else ""
    for f in ("bench.json", "bench_micro1.json", "kernel_stats.csv", "pmc_traffic.csv", "pmc_sq.csv"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
    stamp = {"csrc_sha16": sha, "commit": head, "source": f"profiles/{tag}_pmc_*.csv",
             "command": "rocprofv3 --pmc <counters> -- python bench.py --micro 1 --no-graph --no-cpu-baseline --no-secondary --steps 3 "
                        "--warmup 1 (8 views per launch; one pass per counter group; mean over the dispatches)"}
    tpath = os.path.join(src, "pmc_traffic.csv")
    if os.path.exists(tpath):
        traffic = {}
        for r in csv.DictReader(open(tpath)):
            c = call_of(r["kernel"])
            if c and r["FETCH_SIZE"] and r["WRITE_SIZE"]:
                traffic[c] = traffic.get(c, 0.0) + 1024.0 * (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"]))
        traffic["_stamp"] = dict(stamp, note="bytes per 8-view launch = 2*FETCH_SIZE + WRITE_SIZE (KiB counters); counts "
                                              "L2->fabric requests, i.e. includes Infinity-Cache hits and memory-side atomics.  "
                                              "The doubling is calibrated for gathers too (profiles/r05_gather_calibration.txt: "
                                              "random 16 / 64 / 128-byte gathers all fetch ONE 128-byte line that the counter "
                                              "tallies at 64 B -- so a 64-byte record costs 128 B of fabric traffic)")
        json.dump(traffic, open(os.path.join(dst, traffic_name), "w"), indent=1)
    spath = os.path.join(src, "pmc_sq.csv")
    if os.path.exists(spath) and not only_traffic:
        valu = {}
        for r in csv.DictReader(open(spath)):
            valu[r["kernel"]] = {k: float(v) for k, v in r.items()
                                 if k.startswith(("SQ_", "GRBM_")) and v not in ("", None)}
            valu[r["kernel"]]["abi_call"] = call_of(r["kernel"])
        valu["_stamp"] = dict(stamp, note="per 8-view launch; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count "
                                          "quad-cycles summed over all SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs")
        json.dump(valu, open(os.path.join(dst, "valu.json"), "w"), indent=1)
    print("recorded", tag, "csrc", sha)


def secondary(tag):
    """gpurun_out/TAG/secondary_{workload}_pmc_sq.csv (tools/secondary_pmc.sh) -> profiles/valu_secondary.json:
    {workload name: {kernel: counters per launch}}, stamped with the kernel-source digest."""
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    sha = open(os.path.join(src, "csrc_sha16.txt")).read().strip()
    names = {"mvp": "mvp_config5", "urhand": "urhand_config4_uvlight", "sg": "sgutils_native"}
    out = {}
    for w, cfg_name in names.items():
        path = os.path.join(src, f"secondary_{w}_pmc_sq.csv")
        if not os.path.exists(path):
            continue
        shutil.copy(path, os.path.join(dst, f"{tag}_{w}_pmc_sq.csv"))
        rec = {}
        for r in csv.DictReader(open(path)):
            rec[r["kernel"]] = {k: float(v) for k, v in r.items() if k.startswith(("SQ_", "GRBM_")) and v not in ("", None)}
        out[cfg_name] = rec
        b = os.path.join(src, f"bench_{w}.json")
        if os.path.exists(b):
            shutil.copy(b, os.path.join(dst, f"{tag}_bench_{w}.json"))
    out["_stamp"] = {"csrc_sha16": sha, "source": f"profiles/{tag}_*_pmc_sq.csv",
                     "command": "rocprofv3 --pmc <counters> -- python bench.py --workload <w> --no-cpu-baseline --steps 2 "
                                "--warmup 1 (two passes per workload; mean over the dispatches of a kernel)"}
    json.dump(out, open(os.path.join(dst, "valu_secondary.json"), "w"), indent=1)
    print("recorded secondary", tag, "csrc", sha, list(out))


def e2e(tag):
    """gpurun_out/TAG/e2e_pmc_{traffic,sq}.csv + e2e_kernel_stats.csv (tools/e2e_pmc.sh: the hot path at the reference-native
    1,048,576 Gaussians, 8 views per launch) -> profiles/traffic_e2e.json (bytes per launch of every C-ABI call) and
    profiles/valu_e2e.json (instruction / cycle counters per launch of every kernel), stamped with the kernel-source digest."""
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    sha = open(os.path.join(src, "csrc_sha16.txt")).read().strip()
    for f in ("e2e_kernel_stats.csv", "e2e_pmc_traffic.csv", "e2e_pmc_sq.csv", "bench_e2e.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
    stamp = {"csrc_sha16": sha, "source": f"profiles/{tag}_e2e_pmc_*.csv",
             "command": "rocprofv3 --pmc <counters> -- python bench.py --workload e2e --no-cpu-baseline --steps 3 --warmup 2 "
                        "(1,048,576 Gaussians, 8 views per launch; one pass per counter group; mean over the dispatches)"}
    traffic = {}
    for r in csv.DictReader(open(os.path.join(src, "e2e_pmc_traffic.csv"))):
        c = call_of(r["kernel"])
        if c and r["FETCH_SIZE"] and r["WRITE_SIZE"]:
            traffic[c] = traffic.get(c, 0.0) + 1024.0 * (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"]))
    traffic["_stamp"] = dict(stamp, note="bytes per 8-view launch = 2*FETCH_SIZE + WRITE_SIZE (KiB counters).  The doubling is "
                                         "calibrated for this repo's gather patterns too (profiles/r05_gather_calibration.txt: "
                                         "16-, 64- and 128-byte random gathers all fetch one 128-byte line tallied at 64 B)")
    json.dump(traffic, open(os.path.join(dst, "traffic_e2e.json"), "w"), indent=1)
    valu = {}
    for r in csv.DictReader(open(os.path.join(src, "e2e_pmc_sq.csv"))):
        valu[r["kernel"]] = {k: float(v) for k, v in r.items() if k.startswith(("SQ_", "GRBM_")) and v not in ("", None)}
        valu[r["kernel"]]["abi_call"] = call_of(r["kernel"])
    valu["_stamp"] = dict(stamp, note="per 8-view launch at 1,048,576 Gaussians; quad-cycle counters summed over all SIMDs")
    json.dump(valu, open(os.path.join(dst, "valu_e2e.json"), "w"), indent=1)
    print("recorded e2e", tag, "csrc", sha, sorted(k for k in traffic if not k.startswith("_")))


if __name__ == "__main__":
    if sys.argv[1] == "--secondary":
        secondary(sys.argv[2])
    elif sys.argv[1] == "--e2e":
        e2e(sys.argv[2])
    elif sys.argv[1] == "--coherent-smooth":   # FETCH / WRITE of the micro1 command with --coherent-uv --smooth-normals
        main(sys.argv[2], traffic_name="traffic_coherent_smooth.json", only_traffic=True)
    else:
        main(sys.argv[1])
