#!/usr/bin/env python
"""Which multi-stream dependency graphs survive HIP-graph capture on this ROCm?  (round 6: hipStreamEndCapture segfaults on
the 4-stream pipeline of bench.py --prio.)  Each pattern runs in its own process:  python tools/capture_topology_probe.py"""
import subprocess
import sys

PATTERNS = ["two_side_join_fork", "four_side_fork_join", "hub_chain_one", "hub_chain_two_interleaved",
            "hub_chain_two_interleaved_prio", "direct_side_to_side", "hub_chain_two_sequential_joins",
            "rejoin_same_side_twice"]


def run(pattern):
    import torch

    dev = torch.device("cuda")
    x = [torch.ones(1 << 20, device=dev) for _ in range(8)]
    main = torch.cuda.current_stream()
    prio = -1 if pattern.endswith("prio") else 0
    hA, hB = torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def op(s, i):
        with torch.cuda.stream(s):
            x[i].mul_(1.0001)

    def body():
        if pattern == "two_side_join_fork":
            for s in (sA, sB):
                s.wait_stream(main)
            op(sA, 0); op(sB, 1)
            for s in (sA, sB):
                main.wait_stream(s)
            for s in (sA, sB):
                s.wait_stream(main)
            op(sA, 0); op(sB, 1)
            for s in (sA, sB):
                main.wait_stream(s)
        elif pattern == "four_side_fork_join":
            for s in (sA, sB, hA, hB):
                s.wait_stream(main)
            op(sA, 0); op(sB, 1); op(hA, 2); op(hB, 3)
            for s in (sA, sB, hA, hB):
                main.wait_stream(s)
        elif pattern == "hub_chain_one":
            hA.wait_stream(main); op(hA, 0)
            main.wait_stream(hA); sA.wait_stream(main); op(sA, 0)
            main.wait_stream(sA); hA.wait_stream(main); op(hA, 0)
            main.wait_stream(hA)
        elif pattern.startswith("hub_chain_two_interleaved"):
            hA.wait_stream(main); hB.wait_stream(main)
            op(hA, 0); main.wait_stream(hA); sA.wait_stream(main); op(sA, 0)
            op(hB, 1); main.wait_stream(hB); sB.wait_stream(main); op(sB, 1)
            main.wait_stream(sA); hA.wait_stream(main); op(hA, 0)
            main.wait_stream(sB); hB.wait_stream(main); op(hB, 1)
            main.wait_stream(hA); main.wait_stream(hB); main.wait_stream(sA); main.wait_stream(sB)
        elif pattern == "hub_chain_two_sequential_joins":
            # all shading first, ONE join, all renders, ONE join, all shading backwards, join
            for s in (hA, hB):
                s.wait_stream(main)
            op(hA, 0); op(hB, 1)
            for s in (hA, hB):
                main.wait_stream(s)
            for s in (sA, sB):
                s.wait_stream(main)
            op(sA, 0); op(sB, 1)
            for s in (sA, sB):
                main.wait_stream(s)
            for s in (hA, hB):
                s.wait_stream(main)
            op(hA, 0); op(hB, 1)
            for s in (hA, hB):
                main.wait_stream(s)
        elif pattern == "direct_side_to_side":
            hA.wait_stream(main); op(hA, 0)
            sA.wait_stream(hA); op(sA, 0)
            hA.wait_stream(sA); op(hA, 0)
            main.wait_stream(hA)
        elif pattern == "rejoin_same_side_twice":
            sA.wait_stream(main); op(sA, 0); main.wait_stream(sA)
            x[4].mul_(1.0001)
            sA.wait_stream(main); op(sA, 0); main.wait_stream(sA)

    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    g.replay()
    torch.cuda.synchronize()
    print("OK", pattern, float(x[0][0]))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for p in PATTERNS:
            r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, p], capture_output=True, text=True, timeout=300)
            ok = [l for l in r.stdout.splitlines() if l.startswith("OK")]
            print(f"{p:40s} rc={r.returncode:4d} {ok[0] if ok else (r.stderr.strip().splitlines() or ['?'])[0][:120]}", flush=True)
