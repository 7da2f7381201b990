#!/usr/bin/env python
"""Expected 1/2/4/8-GPU curve of BASELINE config 3 from what ONE GPU can measure (no 8-GPU node was
available to the builder in any round, so this is the prediction the driver's SCALE run is to be compared with).

Run ON THE GPU BOX:  python tools/scaling_model.py [--out gpurun_out/r06_scaling_model.json]

Measured here: the compute half -- bench.py's step at 8 / 4 / 2 / 1 views per GPU (what a rank of N = 1 / 2 / 4 / 8 renders
under config 3's strong scaling: the config-2 batch of 8 views sharded 8 / N per GPU, rgca.py:119-138), median of 5
windows each.  Modelled: the one exchange of the step, parallel.GradSync's reduce-scatter + all-gather of the 60 M-float
decoder gradient set + the albedo map (bench.py --gpus N pays exactly this message), over MI355X's point-to-point xGMI
(MI355X_MICROARCH.md / the round prompt: 7 links x ~153 GB/s per GPU, fully connected 8-GPU node):
  direct   every rank exchanges its 1/N slices with its N-1 peers over N-1 DISTINCT links at once:
           t = 2 phases x (G / N) / B_link          (what a fully connected topology allows; RCCL's multi-ring / direct paths)
  ring     one ring over one link per hop:           t = 2 x (N-1)/N x G / B_link                  (the pessimistic bound)
  + a fixed launch / synchronisation cost per collective (2 per bucket; 30 us each assumed).
Step predictions:  serial  T(N) = T_compute(8/N) + t_exchange(N)   (bench.py's N > 1 headline: exchange inside the step)
                   overlap T(N) = max(T_compute(8/N), t_exchange(N)) (bench.py's `overlapped_exchange`: beside the next step)
views/s = 8 / T(N); efficiency vs N x the 1-GPU rate.  Weak scaling (8 views per rank, bench.py --weak): T = T_compute(8) + t."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_LINK = 153e9          # bytes/s per xGMI link and direction (round prompt / MI355X guide)
T_COLL = 30e-6          # fixed cost per collective (assumption; RCCL launch + sync)
GRAD_FLOATS = 60_000_000 + 250_000 * 3


def measure(views, steps, warmup):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--views", str(views), "--no-secondary", "--no-cpu-baseline",
           "--steps", str(steps), "--warmup", str(warmup)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise RuntimeError(f"bench.py --views {views} failed:\n{r.stderr[-2000:]}")
    d = json.loads(line[0])
    return {"views_per_gpu": views, "ms_per_step_median": d["windows"]["ms_per_step_median"],
            "ms_per_step_min": d["windows"]["ms_per_step_min"], "ms_per_step_max": d["windows"]["ms_per_step_max"],
            "views_per_s": d["value"], "micro_batches": d["config"].get("micro_batches")}


def exchange_ms(n, kind):
    if n == 1:
        return 0.0
    g = 4.0 * GRAD_FLOATS
    buckets = -(-int(g) // (256 << 20))
    t = 2.0 * (g / n) / B_LINK if kind == "direct" else 2.0 * (n - 1) / n * g / B_LINK
    return 1e3 * (t + 2 * buckets * T_COLL)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_scaling_model.json"))
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--from-json", default=None, help="reuse the measured half of an earlier record (no GPU needed)")
    a = ap.parse_args()
    if a.from_json:
        comp = json.load(open(a.from_json))["measured_compute"]
    else:
        comp = [measure(v, a.steps, a.warmup) for v in (8, 4, 2, 1)]
    t_c = {c["views_per_gpu"]: c["ms_per_step_median"] for c in comp}
    base = 8.0 / (t_c[8] * 1e-3)
    rows = []
    for n in (1, 2, 4, 8):
        v = 8 // n
        row = {"n_gpus": n, "views_per_gpu": v, "compute_ms": t_c[v],
               "per_gpu_rate_vs_8_views": (v / t_c[v]) / (8 / t_c[8])}
        for kind in ("direct", "ring"):
            x = exchange_ms(n, kind)
            for mode, t in (("serial", t_c[v] + x), ("overlap", max(t_c[v], x))):
                row[f"{kind}_{mode}"] = {"exchange_ms": x, "step_ms": t, "views_per_s": 8.0 / (t * 1e-3),
                                         "efficiency": (8.0 / (t * 1e-3)) / (n * base)}
            row[f"weak_{kind}_serial_views_per_s"] = 8.0 * n / ((t_c[8] + x) * 1e-3)
        rows.append(row)
    rec = {"what": "expected BASELINE config-3 curve from one-GPU measurements + the xGMI link model (tools/scaling_model.py)",
           "measured_compute": comp, "exchange_model": {"bytes_per_step": 4 * GRAD_FLOATS, "link_GBs": B_LINK / 1e9,
                                                        "fixed_ms_per_collective": 1e3 * T_COLL,
                                                        "buckets": "parallel.GradSync default 256 MiB: ONE bucket of 243 MB -> one "
                                                                   "reduce-scatter + one all-gather per step"},
           "predicted": rows}
    try:
        sys.path.insert(0, ROOT)
        from goliath_amd import build

        rec["_stamp"] = {"csrc_sha16": build.source_digest()}
    except Exception:
        pass
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print("| N | views/GPU | compute ms | exchange ms (direct / ring) | views/s serial (direct / ring) | eff. | views/s overlapped (direct) | eff. |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %d | %.3f | %.2f / %.2f | %.0f / %.0f | %.2f / %.2f | %.0f | %.2f |" % (
            r["n_gpus"], r["views_per_gpu"], r["compute_ms"], r["direct_serial"]["exchange_ms"], r["ring_serial"]["exchange_ms"],
            r["direct_serial"]["views_per_s"], r["ring_serial"]["views_per_s"], r["direct_serial"]["efficiency"],
            r["ring_serial"]["efficiency"], r["direct_overlap"]["views_per_s"], r["direct_overlap"]["efficiency"]))


if __name__ == "__main__":
    main()
