#!/usr/bin/env python
"""A/B matrix of bench.py runs on the GPU box:  python tools/ab_matrix.py TAG "ENV1=a ENV2=b" "ENV1=c" ... -- [bench args]
Every environment combination x every --views in VIEWS (env, default 1,8) is run as
`bench.py --micro 1 --no-graph --no-cpu-baseline --no-secondary --views V` (per-call event times, one launch per ABI call)
and, with GRAPH=1, as the default graph-replay command too.  Prints one table; lines also go to gpurun_out/TAG.jsonl."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    rest = sys.argv[2:]
    extra = []
    if "--" in rest:
        i = rest.index("--")
        rest, extra = rest[:i], rest[i + 1:]
    combos = rest or [""]
    views = [int(v) for v in os.environ.get("VIEWS", "1,8").split(",")]
    graph = os.environ.get("GRAPH", "0") == "1"
    out = open(os.path.join(ROOT, "gpurun_out", tag + ".jsonl"), "a")
    for combo in combos:
        env = dict(os.environ)
        for kv in combo.split():
            k, v = kv.split("=", 1)
            env[k] = v
        for v in views:
            modes = [("eager1", ["--micro", "1", "--no-graph"])] + ([("graph", [])] if graph else [])
            for mode, margs in modes:
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--views", str(v), "--no-cpu-baseline",
                       "--no-secondary", "--steps", "20", "--warmup", "3"] + margs + extra
                r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if not lines:
                    print(f"[{combo}] views={v} {mode}: FAILED rc={r.returncode}\n{r.stderr[-1500:]}")
                    continue
                d = json.loads(lines[0])
                k = d.get("kernels_ms_per_call", {})
                short = {n.replace("gol_", "").replace("rasterize", "rast").replace("project", "proj"): round(1e3 * x)
                         for n, x in k.items()}
                print(f"[{combo or 'base'}] views={v} {mode}: {d['value']:.0f} views/s  {d['ms_per_step']:.3f} ms  us/call {short}",
                      flush=True)
                out.write(json.dumps({"combo": combo, "views": v, "mode": mode, "value": d["value"],
                                      "ms_per_step": d["ms_per_step"], "kernels_ms_per_call": k}) + "\n")
                out.flush()


if __name__ == "__main__":
    main()
