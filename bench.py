#!/usr/bin/env python
"""bench.py -- relit views/sec (forward + backward) of the RGCA render hot path on MI355X.

Metric (BASELINE.json): "relit views/sec (fwd+bwd) at 2048x1334, 250k Gaussians; 1/2/4/8 GPU".
Workload = BASELINE config 2: RGCA head, 250,000 Gaussians, 8 views x 2048x1334 per
GPU, single env-map relight.  One "step" = one batch of 8 views (issued as 2 micro-batches of 4 views on 2
HIP streams, --micro) through the whole hot path with the decoder outputs already resident in HBM
(SURVEY 8d mode A):
    fused shading tail (SH diffuse + activations + env-map specular) + EWA projection of the Gaussians it produces
                                                                          gol_shade_project_fwd
    tile binning + per-tile depth sort                                    gol_bin_sort
    colour + depth tile raster, L1 loss vs a fixed random target image    gol_rasterize_fwd (loss fused into its epilogue)
    raster backward -> per-Gaussian gradient records                      gol_rasterize_bwd
    projection backward + shading backward                                gol_shade_project_bwd
(--unfused-projection: gol_shade_fwd / gol_project_fwd ... gol_project_bwd / gol_shade_bwd, rounds 1-3.)
Multi-GPU (--gpus N): BASELINE config 3 -- the 8 views of the batch sharded 8 / N per GPU (strong scaling), the 60 M-float
gradient exchange of the decoder's parameter set inside the step; --weak keeps 8 views per rank.

Prints ONE JSON line (rank 0).  Per-ABI-call durations come from HIP events recorded on the launch
stream inside the timed region; `roofline` describes the longest one.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# VALU issue peak from the guide (MI355X_MICROARCH.md: v_fma_f32 = 2 cycles per wave64 instruction on a SIMD-32; 256 CUs x 4
# SIMDs x 2.4 GHz / 2 = 1229 G wave-instructions/s).  `frac` is quoted against THIS.  Beside it, as `probe_ceiling`: what
# tools/probe/valu_probe.hip sustains for the cheapest class on the bench box class (2.94 cycles -> 836 G/s,
# profiles/r02c_valu_probe.txt; packed fp32 7.6, transcendentals / permlane swaps 8.2 cycles).
VALU_PEAK_GINST = 1024 * 2.4 / 2.0
VALU_PROBE_GINST = 1024 * 2.4 / 2.94
VALU_BOUND_CALLS = {"gol_rasterize_fwd": "raster_fwd_kernel", "gol_rasterize_bwd": "raster_bwd_kernel",
                    "gol_render_fwd": "raster_fwd_kernel", "gol_render_bwd": "raster_bwd_kernel"}

CFG = dict(workload="rgca_config2_envrelight", gaussians=250_000, slab=500, height=2048, width=1334,
           views_per_gpu=8, focal=3000.0, cam_radius_mm=700.0, n_mips=4, seed=1234)



# The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner through C
# stdio when its first communicator comes up, flushed at exit, i.e. AFTER the JSON line): once this process knows it is a
# worker (not the self-spawning launcher), file descriptor 1 is pointed at stderr and the line goes to the saved descriptor.
_REAL_STDOUT = None


def _claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


LINE_LIMIT = 8000   # bytes: the driver reads a bounded tail of stdout (BENCH_r05: a 22.5 KB line came back `parsed: null`)
DETAIL_PATH = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _clean(x, sig=None):
    """JSON-strict copy: NaN / inf -> null (json.dumps(allow_nan=False) then cannot fail), floats optionally rounded to
    `sig` significant digits."""
    if isinstance(x, dict):
        return {str(k): _clean(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v, sig) for v in x]
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        return float(f"{x:.{sig}g}") if sig else x
    if torch.is_tensor(x):
        return _clean(x.tolist(), sig)
    return _clean(float(x), sig) if hasattr(x, "__float__") else str(x)


def _roof_brief(r):
    """roofline of the one-line record: the contract's keys + the issue-roofline figures of a VALU-bound kernel."""
    if not r:
        return None
    out = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic") if k in r}
    v = r.get("valu")
    if v:
        out["valu"] = {k: v.get(k) for k in ("issued_frac", "algorithmic_frac", "issued_over_algorithmic", "busy_frac_pmc",
                                             "peak_G_wave_inst_s") if v.get(k) is not None}
    if r.get("limiter"):
        out["limiter"] = r["limiter"]
    return out


_CONFIG_KEYS = ("workload", "gaussians", "image", "views_per_gpu", "views_per_step", "micro_batches", "world_size",
                "dist_backend", "rccl_world_size", "parallelism", "intersections_per_view", "grad_exchange_bytes_per_step",
                "grad_exchange_ms_alone", "ranks_share_gpu", "prims", "uv", "lights", "views", "fused_tail", "loss")


def _sec_brief(s):
    """a secondary workload in the one-line record: value, step time and the roofline fraction of its longest call."""
    if not isinstance(s, dict) or "error" in s:
        return s
    r = s.get("roofline") or {}
    out = {"value": s.get("value"), "unit": s.get("unit"), "ms_per_step": s.get("ms_per_step"),
           "roofline": {"kernel": r.get("kernel"), "bound": r.get("bound"), "frac": r.get("frac")}}
    if r.get("limiter") == "valu_issue" and (r.get("valu") or {}).get("issued_frac") is not None:
        # a VALU-bound kernel: its HBM fraction alone says little -- the issue fraction beside it
        out["roofline"].update(limiter="valu_issue", valu_issued_frac=r["valu"]["issued_frac"])
    w = s.get("windows") or {}
    if w.get("ms_per_step_median") is not None:
        out["ms_per_step_median"] = w["ms_per_step_median"]
    for k in ("loss_first_step", "loss_last_step", "hot_path_ms_per_step"):
        if k in s:
            out[k] = s[k]
    return out


def compact_line(res):
    """The ONE stdout line (< LINE_LIMIT bytes, strict JSON): the contract's fields, the per-call times, the roofline of the
    dominant kernel, the streaming calls' HBM fractions, the CPU baseline and one short entry per secondary workload.
    Everything else (secondary configs and per-call tables, the parity record, texts) goes to DETAIL_PATH and stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: res[k] for k in keep if k in res}
    cfg = res.get("config") or {}
    out["config"] = {k: cfg[k] for k in _CONFIG_KEYS if k in cfg}
    if cfg.get("launch"):
        out["config"]["launch"] = "hip_graph_replay" if cfg["launch"].startswith("hip_graph_replay") else "eager"
    if "grad_exchange" in cfg and cfg["grad_exchange"]:
        out["config"]["grad_exchange"] = "serial" if "serial" in cfg["grad_exchange"] else "overlapped"
    for k in ("stub_gradients_averaged", "note"):
        if k in res:
            out[k] = res[k]
    w = res.get("windows")
    if w:
        out["windows"] = {k: w[k] for k in ("n", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max") if k in w}
    if res.get("kernels_ms_per_call"):
        out["kernels_ms_per_call"] = res["kernels_ms_per_call"]
    if "roofline" in res:
        out["roofline"] = _roof_brief(res["roofline"])
    if res.get("hbm_per_call"):
        out["hbm_per_call"] = res["hbm_per_call"]
    for k in ("overlapped_exchange", "weak"):
        if k in res:
            out[k] = {kk: vv for kk, vv in res[k].items() if kk in ("value", "unit", "ms_per_step", "scaling", "views_per_gpu")}
    for k in ("loss_first_step", "loss_last_step", "hot_path_ms_per_step"):
        if k in res:
            out[k] = res[k]
    if "cpu_baseline" in res:
        out["cpu_baseline"] = dict(res["cpu_baseline"])
    if "secondary" in res:
        out["secondary"] = {k: _sec_brief(v) for k, v in res["secondary"].items()}
    out["detail"] = "gpurun_out/bench_detail.json (+ stderr)"
    out = _clean(out, sig=5)
    # belt and braces: shed the optional blocks, least important first, until the line fits
    for drop in ("hbm_per_call", "secondary", "windows", "kernels_ms_per_call"):
        if len(json.dumps(out, allow_nan=False, separators=(",", ":"))) < LINE_LIMIT:
            break
        out.pop(drop, None)
    return out


def emit(obj):
    """Full record -> gpurun_out/bench_detail.json and stderr; the compact one-line record -> stdout."""
    full = _clean(obj)
    try:
        os.makedirs(os.path.dirname(DETAIL_PATH), exist_ok=True)
        with open(DETAIL_PATH, "w") as f:
            json.dump(full, f, allow_nan=False, indent=1)
    except OSError as e:
        print(f"bench.py: could not write {DETAIL_PATH}: {e}", file=sys.stderr)
    print("bench.py detail: " + json.dumps(full, allow_nan=False), file=sys.stderr, flush=True)
    text = json.dumps(compact_line(obj), allow_nan=False, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    line = (text + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, line)


def make_inputs(cfg, device, rank=0):
    """SURVEY.md 8d config-2 synthetic inputs (decoder-output surrogates + cameras + env map)."""
    g = torch.Generator().manual_seed(cfg["seed"] + 1000 * rank)
    B, S = cfg["views_per_gpu"], cfg["slab"]
    N = S * S
    f_vn = 0.3 * torch.randn(B, 125, S, S, generator=g)
    # Gaussian parameter channels so that the activations reproduce the 8d statistics
    d = torch.randn(N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    pos = d * torch.rand(N, 1, generator=g) ** (1 / 3) * torch.tensor([90.0, 120.0, 100.0])
    if cfg.get("coherent_uv"):
        pos = _uv_coherent_order(pos, S)
    postex = pos.t().reshape(1, 3, S, S).expand(B, -1, -1, -1).contiguous()
    tn = F.normalize(pos, dim=-1).t().reshape(1, 3, S, S).expand(B, -1, -1, -1).contiguous()
    scale = torch.exp(math.log(0.3) + (math.log(3.0) - math.log(0.3)) * torch.rand(B, 3, S, S, generator=g))
    f_vn[:, 113 + 7:113 + 10] = torch.log(torch.expm1(scale))           # softplus^-1
    f_vn[:, 113 + 3:113 + 7] = torch.randn(B, 4, S, S, generator=g)      # quaternion
    f_vn[:, 113 + 10] = 1.5 * torch.randn(B, S, S, generator=g)          # opacity logit
    f_vn[:, 113 + 11] = -1.0 + 0.5 * torch.randn(B, S, S, generator=g)   # roughness: sigma ~ 0.04
    f_vc = 0.3 * torch.randn(B, 4, S, S, generator=g)
    if cfg.get("smooth_normals"):
        # --smooth-normals: what a convolutional decoder emits -- offsets that vary slowly over the slab (white noise on a
        # S/25 grid, bilinearly upsampled, same 0.3 standard deviation) instead of SURVEY 8d's per-texel white noise
        low = torch.randn(B, 4, max(S // 25, 2), max(S // 25, 2), generator=g)
        f_vc = F.interpolate(low, size=(S, S), mode="bilinear", align_corners=False)
        f_vc = 0.3 * f_vc / f_vc.std()
    albedo = 0.2 + 0.6 * torch.rand(1, N, 3, generator=g)
    light_sh = 0.3 * torch.randn(B, 3, 81, generator=g) / (1 + torch.arange(81.0)) ** 0.5
    light_sh[:, :, 0] = 1.5
    if cfg.get("env_per_view"):
        # rounds 1-4's workload: B DIFFERENT pyramids (harder than config 2: nothing of a map is shared between views)
        mips = [torch.exp(0.5 * torch.randn(B, 3, 512 >> i, 1024 >> i, generator=g)) * 0.5 for i in range(cfg["n_mips"])]
    else:
        # BASELINE config 2 "single env-map relight" / SURVEY 8d: ONE synthetic HDR pyramid (log-normal pixels) for every view
        # of the step on every rank, rotated per view by `lightrot` -- what EnvSpinDecorator feeds (light_decorator.py:96-100,
        # 112-118: one registered pyramid expanded over the batch; the lookup direction is rotated, not the map)
        ge = torch.Generator().manual_seed(cfg["seed"] + 4242)
        mips = [torch.exp(0.5 * torch.randn(1, 3, 512 >> i, 1024 >> i, generator=ge)) * 0.5 for i in range(cfg["n_mips"])]
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = cfg["focal"]
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = cfg["width"] / 2.0, cfg["height"] / 2.0, 1.0
    Rt, campos, rots = [], [], []
    for b in range(B):
        ang = 2 * math.pi * (b + 8 * rank) / 64.0 - 0.35
        eye = torch.tensor([cfg["cam_radius_mm"] * math.sin(ang), 0.0, -cfg["cam_radius_mm"] * math.cos(ang)])
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), fwd)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        R = torch.stack([right, down, fwd])
        Rt.append(torch.cat([R, (-R @ eye)[:, None]], 1))
        campos.append(eye)
        a = 2 * math.pi * b / 256.0
        rots.append(torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]))
    target = torch.rand(B, 3, cfg["height"], cfg["width"], generator=g)
    t = dict(f_vn=f_vn, f_vc=f_vc, postex=postex, tn=tn, albedo=albedo, light_sh=light_sh, K=K,
             Rt=torch.stack(Rt), campos=torch.stack(campos), lightrot=torch.stack(rots), target=target)
    t = {k: v.to(device).contiguous() for k, v in t.items()}
    t["mips"] = [m.to(device).contiguous() for m in mips]
    for k in ("f_vn", "f_vc", "postex", "tn", "albedo"):
        t[k].requires_grad_(True)
    return t


def _morton2(x, y):
    """Interleave the low 16 bits of two int64 tensors (x -> even bits, y -> odd bits)."""
    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return spread(x) | (spread(y) << 1)


def _uv_coherent_order(pos, S):
    """--coherent-uv: the same point set, laid out over the S x S slab the way a UV atlas lays out a surface -- texels that
    are neighbours in the slab hold points that are neighbours in direction (the default layout is a random permutation:
    every slab neighbour is an unrelated point, the worst case for the env-map gathers of the shading kernel).  Points are
    sorted along the Z-order curve of their direction's (azimuth, polar) cell and dealt out along the Z-order curve of the
    slab."""
    n = F.normalize(pos, dim=-1)
    az = (torch.atan2(n[:, 0], n[:, 2]) / (2 * math.pi) + 0.5).clamp(0, 1 - 1e-7)
    po = (torch.acos(n[:, 1].clamp(-1, 1)) / math.pi).clamp(0, 1 - 1e-7)
    order = torch.argsort(_morton2((az * 65536).long(), (po * 65536).long()))
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    slots = torch.argsort(_morton2(xx.reshape(-1).long(), yy.reshape(-1).long()))   # slab texels in Z-order
    out = torch.empty_like(pos)
    out[slots] = pos[order]
    return out


def _join_streams(streams, hub):
    """Every stream waits for what all the others have been given so far: joined on `hub`, forked again (GPU-side event
    waits, nothing blocks the host; the join / fork form is what a HIP graph capture digests -- mutual event waits between
    the side streams crash hipStreamEndCapture on ROCm 7.0)."""
    for s in streams:
        hub.wait_stream(s)
    for s in streams:
        s.wait_stream(hub)


def step(t, cfg, world):
    """One step = all views of this GPU's batch, as `len(t["micro"])` micro-batches on separate HIP streams (the views
    are independent; two half-batches in flight keep the chip busy while one of them is in a small-grid kernel such as
    the tile scan or the tail of the per-tile sort).
    cfg["align_micro_batches"]: all shading calls are issued first and the streams joined ONCE before the renders.  The
    first-issued shading kernel gets the chip's workgroup slots first and ends ~0.1 ms before the other; unjoined, its
    stream reaches the raster kernels (43 k small workgroups) that much earlier and the other stream's binning crawls
    behind them (profiles/r04_two_stream_timeline.txt: 0.6 ms for a 4 us kernel)."""
    from goliath_amd import losses, render_gs, shade

    main = torch.cuda.current_stream()
    micro, streams = t["micro"], t["streams"]
    align = cfg.get("align_micro_batches", False) and len(set(streams)) > 1
    for stream in streams:
        stream.wait_stream(main)

    def render_and_backward(mb, p):
        # loss == (rgb - target).abs().mean() (loss/__init__.py:411), evaluated in the raster epilogue and
        # back-propagated by the raster backward itself (losses.l1_image is the stand-alone form of the same op)
        loss = render_gs.render_batch(mb["K"], mb["Rt"], p, cfg["height"], cfg["width"], l1_target=mb["target"])[3]
        loss.backward()
        return loss

    preds, loss = [], None
    for mb, stream in zip(micro, streams):
        with torch.cuda.stream(stream):
            for k in ("f_vn", "f_vc", "postex", "tn", "albedo"):
                mb[k].grad = None
            # the cameras go into the shading call: its kernel projects the Gaussians it produces, the render starts at the
            # tile count and hands its gradient records back to the shading backward (what AutoEncoder.forward does,
            # goliath_amd/rgca.py); --unfused-projection: gol_project_fwd / bwd as kernels of their own (rounds 1-3)
            vs = render_gs.view_set(mb["K"], mb["Rt"], cfg["height"], cfg["width"]) if cfg.get("fused_projection", True) else None
            preds.append(shade.shading_tail(mb["f_vn"], mb["f_vc"], mb["postex"], mb["tn"], mb["albedo"], mb["light_sh"],
                                            mb["campos"], preconv_envmap=mb["mips"], lightrot=mb["lightrot"], views=vs))
            if not align:
                loss = render_and_backward(mb, preds[-1])
    if align:
        _join_streams(streams, main)
        for mb, stream, p in zip(micro, streams, preds):
            with torch.cuda.stream(stream):
                loss = render_and_backward(mb, p)
    for stream in streams:
        main.wait_stream(stream)
    # the path's only parameter (albedo, rgca.py:462-464) is shared by all views: sum the micro-batch grads
    grads = [mb["albedo"].grad for mb in t["micro"]]
    t["_albedo_grad"] = grads[0] if len(grads) == 1 else (
        torch.add(grads[0], grads[1]) if len(grads) == 2 else torch.stack(grads).sum(0))
    t["albedo"].grad = t["_albedo_grad"]
    return loss


def _split(n, parts):
    return [n // parts + (1 if i < n % parts else 0) for i in range(parts)] if n > 0 else []


def run_step(t, cfg, world, graph=None):
    """One timed step: the compute part (eager, or one replay of its captured HIP graph) + the gradient exchange.
    Exchange modes (N > 1): "overlap" (default) -- step k's reduce-scatter + all-gather is launched asynchronously after
    step k's compute and waited for right before step k+1 launches its own, so it runs beside step k+1's kernels (the
    hot-path-only step has no decoder backward of its own to hide it behind); "serial" -- launched and waited for
    inside the step (--serial-exchange)."""
    if t.get("stub"):
        t["albedo"].grad = t["albedo"].detach() * float(t["rank"] + 1)   # launcher self-test: no kernels
    elif graph is None:
        step(t, cfg, world)
    else:
        graph.replay()
        t["albedo"].grad = t["_albedo_grad"]  # the tensor the captured step writes (GradSync re-points .grad to its bucket)
    sync = t.get("_sync")
    if sync is not None:
        if t.get("overlap_exchange"):
            sync.wait()          # the previous step's exchange
            sync.launch_all()    # this step's: in flight during the next step's compute
        else:
            sync.sync()          # reduce-scatter + all-gather over RCCL, inside the step


def make_step_inputs(cfg, device, rank, n_micro):
    """The GPU's batch as n_micro micro-batches (leaf tensors per micro-batch, one shared albedo)."""
    B = cfg["views_per_gpu"]
    assert B % n_micro == 0
    micro = [make_inputs(dict(cfg, views_per_gpu=B // n_micro), device, rank * n_micro + i) for i in range(n_micro)]
    albedo = micro[0]["albedo"].detach().clone().requires_grad_(True)
    for mb in micro:
        mb["albedo"] = albedo.detach().clone().requires_grad_(True)  # per-stream alias of the shared parameter
        if not cfg.get("env_per_view"):
            mb["mips"] = micro[0]["mips"]                            # ONE pyramid in HBM for the whole step
    return {"micro": micro, "albedo": albedo, "streams": [torch.cuda.Stream(device=device) for _ in range(n_micro)]}


# algorithmic HBM bytes per view of each ABI call: what the call has to move given the data layout
# -- every input read once, every output written once, a list entry = its 4-byte id + the 64-byte record it gathers.
def algorithmic_bytes(name, N, I, P, n_mips_bytes):
    return {
        "gol_shade_fwd": (129 * 4 + 24 + 12) * N + 148 * N,
        "gol_shade_bwd": (16 * 4 + 24 + 12 + 12) * N + 56 * N + (129 * 4 + 24 + 12) * N,
        # with the projection fused in: + its outputs (xy 8, depth 4, radius 4, conic 12, compensation 4, effective opacity 4,
        # raster record 64) / + the gradient record 64, radius 4, conic 12, compensation 4 instead of the 56 B of upstream
        # gradients of colour / opacity / position / quaternion / scale
        "gol_shade_project_fwd": (129 * 4 + 24 + 12) * N + 148 * N + 100 * N,
        "gol_shade_project_bwd": (16 * 4 + 24 + 12 + 12) * N + 84 * N + (129 * 4 + 24 + 12) * N,
        # in: means 12, scales 12, quats 16, opacity 4, colour 12; out: xy 8, depth 4, radius 4, conic 12, compensation 4,
        # effective opacity 4, raster record 64
        "gol_project_fwd": 56 * N + 100 * N,
        # in: means / scales / quats / opacity 44, radius 4, conic 12, compensation 4, the Gaussian's gradient record 64;
        # out: gradients of means / scales / quats / opacity 44 + the dense colour gradient 12
        "gol_project_bwd": 128 * N + 56 * N,
        # count pass 28 N (xy, radius, conic, opacity), scatter pass 2 x 28 N + depth 4 N; keys 8 I written + 8 I read, ids 4 I
        "gol_bin_sort": 88 * N + 8 * I + 8 * I + 4 * I,
        # per entry: id 4 + record 64; per pixel: final_T, final_idx, rgb, alpha, depth_norm written (28 B) + fused L1:
        # target read 12, sign byte written
        "gol_rasterize_fwd": 68 * I + 28 * P + 13 * P,
        # per entry: id + record; per pixel: final_T, final_idx, sign byte; per Gaussian: 10 gradient floats accumulated
        "gol_rasterize_bwd": 68 * I + 9 * P + 40 * N,
        "gol_l1_fwd": 24 * P,           # rendered + target image
        "gol_l1_bwd": 24 * P + 12 * P,  # ... and the image gradient
    }[name]


def _stamped(name):
    """profiles/<name> (PMC-derived, per 8-view launch) if it was measured on the kernel sources of this tree."""
    from goliath_amd import build

    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    data = json.load(open(path))
    if data.get("_stamp", {}).get("csrc_sha16") != build.source_digest():
        return None  # stale: the kernels changed since the counters were taken
    return data


# minimal VALU lane-operations per TAKEN (pixel, Gaussian) pair (the arithmetic the compositing
# recurrence and its derivative need per pixel, nothing shared, no bookkeeping, perfect reductions):
#   forward  17: sigma 5, exp 1, alpha 2, cut 1, alpha T 1, T update 1, stop test 1, colour + depth FMAs 4, live mask 1
#   backward 35: sigma 5, exp 1, alpha 2, cut 1, 1 / (1 - alpha) 2, T 1, fac 1, <colour, v_out> 3, v_alpha 3, q 1, colour
#                gradients 3, gop 1, moments of gop 5, + one add per pixel for each of the 9 sums 9 (a perfect tree), mask 1
ALG_OPS_PER_PAIR = {"raster_fwd_kernel": 17, "raster_bwd_kernel": 35}


def make_roofline(call, ms, views_per_launch, N, I, P, mip_bytes, pairs_taken=None, records=""):
    """Roofline of the dominant ABI call.  Time: HIP events around the call, measured live.  Algorithmic bytes: DESIGN.md
    section 4.  HBM traffic and instruction counts: rocprofv3 PMC passes recorded under profiles/ (tools/gpu_profile.sh +
    tools/make_profile_record.py), used only while their source digest matches this tree (else null).  records: "" = the
    250k-Gaussian records (traffic.json / valu.json), "_e2e" = the ones taken at 1,048,576 Gaussians (tools/e2e_pmc.sh)."""
    alg = views_per_launch * algorithmic_bytes(call, N, I, P, mip_bytes)
    gbs = alg / (ms * 1e-3) / 1e9
    traffic = _stamped(f"traffic{records}.json")
    traffic = None if traffic is None or call not in traffic else traffic[call] * views_per_launch / 8.0
    roof = {"bound": "hbm", "kernel": call, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic, "algorithmic_bytes_per_launch": alg}
    if call not in VALU_BOUND_CALLS:
        return roof
    # The contract's roofline is algorithmic bytes / live time against the HBM peak (above).  The rasterizer moves ~1/10 of
    # what HBM could deliver in its run time (its records are served from L2: traffic < algorithmic bytes) and keeps the
    # vector ALUs busy instead, so the figure that says how good the kernel is sits beside it under `valu`: instruction
    # issue against the guide's peak, and how much of the issued work is arithmetic the compositing needs.
    roof["limiter"] = "valu_issue"
    valu = _stamped(f"valu{records}.json")
    v = _valu_roofline(valu, VALU_BOUND_CALLS[call], ms, views_per_launch / 8.0)
    vv = {"issued_G_wave_inst_s": v["achieved"], "peak_G_wave_inst_s": v["peak"], "issued_frac": v["frac"],
          "peak_source": v["peak_source"], "probe_ceiling": v["probe_ceiling"],
          "frac_of_probe_ceiling": v["frac_of_probe_ceiling"], "busy_frac_pmc": v["valu_busy_frac_pmc"],
          "useful_lane_frac": v["useful_lane_frac"], "valu_instructions_per_launch": v["valu_instructions_per_launch"],
          "evidence": v["evidence"]}
    if pairs_taken is not None:
        # issued vs algorithmic: algorithmic_inst = taken (pixel, Gaussian) pairs of the launch x minimal lane-operations per
        # pair / 64 lanes (wave-instructions)
        ai = pairs_taken * views_per_launch * ALG_OPS_PER_PAIR[VALU_BOUND_CALLS[call]] / 64.0
        insts = vv["valu_instructions_per_launch"]
        vv.update(algorithmic_inst=ai, taken_pixel_gaussian_pairs_per_view=pairs_taken,
                  lane_ops_per_taken_pair=ALG_OPS_PER_PAIR[VALU_BOUND_CALLS[call]],
                  issued_over_algorithmic=None if not insts else insts / ai,
                  algorithmic_frac=ai / (ms * 1e-3) / 1e9 / VALU_PEAK_GINST)
    roof["valu"] = vv
    return roof


def _valu_roofline(record, kernel_prefix, ms, scale=1.0):
    """Instruction-issue roofline of a VALU-bound kernel from a stamped PMC record (profiles/valu*.json; counters per
    launch, `scale` = this launch's share of the recorded launch).  achieved = SQ_INSTS_VALU / live duration;
    peak = the guide's v_fma_f32 rate; useful_lane_frac = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): the mean
    fraction of the 64 lanes that an issued VALU instruction had enabled (exec mask), i.e. issue utilisation x this =
    lane-level utilisation."""
    insts = busy = lanes = None
    if record is not None:
        for k, v in record.items():
            if k.startswith(kernel_prefix) and isinstance(v, dict) and "SQ_INSTS_VALU" in v:
                insts = v["SQ_INSTS_VALU"] * scale
                if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:  # quad-cycles over 1024 SIMDs / XCD-summed cycles
                    busy = 4.0 * v["SQ_ACTIVE_INST_VALU"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
                if v.get("SQ_THREAD_CYCLES_VALU") and v.get("SQ_ACTIVE_INST_VALU"):
                    lanes = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"])
                break
    ach = None if insts is None else insts / (ms * 1e-3) / 1e9
    return {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
            "frac": None if ach is None else ach / VALU_PEAK_GINST,
            "peak_source": "MI355X_MICROARCH.md: v_fma_f32 2 cycles / wave64 / SIMD-32 -> 1024 SIMDs x 2.4 GHz / 2",
            "probe_ceiling": VALU_PROBE_GINST, "frac_of_probe_ceiling": None if ach is None else ach / VALU_PROBE_GINST,
            "valu_busy_frac_pmc": busy, "useful_lane_frac": lanes, "valu_instructions_per_launch": insts,
            "evidence": "profiles/valu*.json, profiles/traffic.json (rocprofv3 --pmc, same source digest)" if insts else
                        "no PMC record for this source digest under profiles/: instruction count unavailable"}


def cpu_baseline(cfg):
    """Times the CPU oracle (a PORT: there is no reference CPU path for gsplat/sgutils) on ONE view of
    the same workload, forward + backward, on the host cores."""
    from oracle import chain, cref

    threads = min(32, os.cpu_count() or 1)  # more threads only add contention on the atomics
    torch.set_num_threads(threads)
    cref.set_threads(threads)
    n_views, dt = 4, 0.0
    for v in range(n_views):
        t = make_inputs(dict(cfg, views_per_gpu=1), "cpu", rank=v)
        t0 = time.perf_counter()
        chain.cpu_view(t, cfg["height"], cfg["width"])
        dt += time.perf_counter() - t0
    return {"value": n_views / dt, "unit": "views/s", "cores": threads, "kind": "port",
            "sample": f"{n_views} full views (250k Gaussians, 2048x1334, env relight) fwd+bwd through oracle/chain.py "
                      f"in {dt:.1f} s: C + OpenMP restatement of gsplat 0.1.11 project/bin/raster (no reference CPU path "
                      f"exists for it) and oracle/shade_ref.py, the torch restatement of rgca.py:505-588 pinned to the "
                      f"reference's own PrimDecoder.forward (/root/reference is not on the bench box)"}


MVP_CFG = dict(workload="mvp_config5", prims=4096, tdim=(8, 16, 16), height=2048, width=1334, views_per_gpu=1,
               seed=1112)


def mvp_inputs(cfg, device, rank=0):
    """16^3 perturbed lattice of boxes filling the unit cube, softplus(1.5 N(0,1))
    template with alpha - 3.5, primscale 16 (box half-extent 1/16), stepsize 1/64, pinhole camera."""
    g = torch.Generator().manual_seed(cfg["seed"] + rank)
    B, K = cfg["views_per_gpu"], cfg["prims"]
    k3 = round(K ** (1 / 3))
    lin = (torch.arange(k3) + 0.5) / k3 * 2 - 1
    centres = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    primpos = (centres[None] + 0.02 * torch.randn(B, K, 3, generator=g)).contiguous()
    q = F.normalize(torch.randn(B, K, 4, generator=g) * 0.1 + torch.tensor([1.0, 0, 0, 0]), dim=-1)
    w, x, y, z = q.unbind(-1)
    primrot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                           1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                           1 - 2 * (x * x + y * y)], -1).reshape(B, K, 3, 3).contiguous()
    primscale = torch.full((B, K, 3), float(k3))
    TD, TH, TW = cfg["tdim"]
    raw = 1.5 * torch.randn(B, K, TD, TH, TW, 4, generator=g)
    raw[..., 3] -= 3.5
    template = F.softplus(raw)
    H, W = cfg["height"], cfg["width"]
    t = dict(primpos=primpos, primrot=primrot, primscale=primscale, template=template,
             viewpos=torch.tensor([[0.3, -0.2, -2.6]] * B), viewrot=torch.eye(3)[None].repeat(B, 1, 1),
             focal=torch.full((B, 2), 2600.0), princpt=torch.tensor([[W / 2.0, H / 2.0]] * B),
             target=torch.rand(B, H, W, 4, generator=g))
    t = {k: v.to(device).contiguous() for k, v in t.items()}
    for k in ("primpos", "primrot", "primscale", "template"):
        t[k].requires_grad_(True)
    return t


def _median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def _per_call_ms(timing):
    """MEDIAN duration per ABI call over the instrumented passes (one late allocation or a clock ramp inside a bracket
    used to ride in the mean: VERDICT r5 weak 9)."""
    per = {}
    for name, e0, e1 in timing:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    return {k: _median(v) for k, v in per.items()}


def _window_stats(ms_list):
    """median / min / max of the per-window step times (ms per step)."""
    if not ms_list:
        return None
    v = sorted(ms_list)
    n = len(v)
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"n": n, "ms_per_step_median": med, "ms_per_step_min": v[0], "ms_per_step_max": v[-1],
            "spread_frac": (v[-1] - v[0]) / med if med > 0 else None}


def _window_plan(steps, n_windows=5):
    """Split K steps into up to n_windows consecutive windows (sizes differ by at most one)."""
    w = max(1, min(n_windows, steps))
    return _split(steps, w)


def _time_steps(step, args):
    """warmup + timed steps of a secondary workload; returns (last output, seconds, mean ms per ABI call, window stats).
    The K timed steps are ONE region (sync on both sides); HIP events recorded between windows of it give the spread."""
    from goliath_amd import _lib

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    _lib.TIMING = []
    plan = _window_plan(args.steps)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(plan) + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for wi, n in enumerate(plan):
        for _ in range(n):
            out = step()
        marks[wi + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing, _lib.TIMING = _lib.TIMING, None
    win = _window_stats([marks[i].elapsed_time(marks[i + 1]) / n for i, n in enumerate(plan)])
    return out, dt, _per_call_ms(timing), win


# secondary workloads: which calls are VALU-bound (kernel-name prefix in profiles/valu_secondary.json, produced by
# tools/secondary_pmc.sh + tools/make_profile_record.py --secondary and stamped with the kernel-source digest)
SECONDARY_VALU = {"gol_mvp_march_fwd": "march_fwd_kernel", "gol_mvp_march_bwd": "march_bwd_kernel",
                  "gol_mvp_shadow_march": "march_fwd_kernel", "gol_mesh_raster": "mesh_raster_kernel",
                  "gol_sg_eval_fwd": "sg_fwd_kernel", "gol_sg_eval_bwd": "sg_bwd_kernel",
                  "gol_uvlight_phong_fwd": "phong_kernel", "gol_uvlight_phong_bwd": "phong_kernel",
                  "gol_uvlight_ggx_fwd": "ggx_kernel", "gol_uvlight_ggx_bwd": "ggx_kernel"}


def _secondary_line(metric, unit, units_per_step, args, dt, ms, alg, config, windows=None):
    """JSON line of a secondary workload.  Roofline of the longest call: the instruction-issue roofline (guide peak, PMC
    instruction count of the stamped record for THIS workload) when the call is VALU-bound and a record of this tree
    exists, the HBM roofline (algorithmic bytes / live time) otherwise -- with the other one beside it."""
    dom = max(ms, key=ms.get)
    ach = alg.get(dom, 0) / (ms[dom] * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None}
    rec = _stamped("valu_secondary.json")
    rec = None if rec is None else rec.get(config["workload"])
    if dom in SECONDARY_VALU:
        v = _valu_roofline(rec, SECONDARY_VALU[dom], ms[dom])
        roof["limiter"] = "valu_issue"
        roof["valu"] = {"issued_G_wave_inst_s": v["achieved"], "peak_G_wave_inst_s": v["peak"], "issued_frac": v["frac"],
                        "busy_frac_pmc": v["valu_busy_frac_pmc"], "useful_lane_frac": v["useful_lane_frac"],
                        "valu_instructions_per_launch": v["valu_instructions_per_launch"], "evidence": v["evidence"]}
    return {
        "metric": metric, "value": units_per_step * args.steps / dt, "unit": unit, "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config, "kernels_ms_per_call": ms,
        "algorithmic_GBs_per_call": {k: alg[k] / (ms[k] * 1e-3) / 1e9 for k in ms if k in alg},
        "roofline": roof, "windows": windows}


URHAND_CFG = dict(workload="urhand_config4_uvlight", uv=1024, lights=32, frames_per_gpu=1, seed=4)


def urhand_main(args, emit_line=True):
    """Secondary workload (BASELINE config 4): URHand UV light loops, S=1024, L=32 point lights on a
    1100 mm sphere, B=1: per-light mesh depth render (the drtk call of shadowmap.py:39-50; synthetic closed mesh of 5120
    faces, no dataset) + shadow-map PCF (shadowmap.py:30-96) + Phong features (urhand.py:419-445) + GGX shading
    (:508-567), fwd+bwd."""
    from goliath_amd import meshraster, shadowmap, uvlight

    cfg = URHAND_CFG
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(cfg["seed"])
    B, S, L = cfg["frames_per_gpu"], cfg["uv"], cfg["lights"]
    # texel positions: a UV ATLAS of the closed surface (latitude / longitude chart of the sphere, like a hand's uv map:
    # neighbouring texels are neighbouring surface points), slightly jittered
    th_ = math.pi * (torch.arange(S) + 0.5) / S
    ph_ = 2 * math.pi * (torch.arange(S) + 0.5) / S
    d = torch.stack([torch.sin(th_)[:, None] * torch.cos(ph_)[None], torch.cos(th_)[:, None].expand(S, S),
                     torch.sin(th_)[:, None] * torch.sin(ph_)[None]])[None].repeat(B, 1, 1, 1)
    d = F.normalize(d + 0.002 * torch.randn(B, 3, S, S, generator=g), dim=1)
    t = dict(p_uv=d * 90.0, nml=F.normalize(d + 0.2 * torch.randn(B, 3, S, S, generator=g), dim=1),
             cam=torch.tensor([[0.0, 0.0, -700.0]] * B),
             lpos=1100.0 * F.normalize(torch.randn(B, L, 3, generator=g), dim=-1),
             lint=torch.rand(B, L, 1, generator=g),
             rough=0.3 + 0.5 * torch.rand(B, 1, S, S, generator=g), tex=torch.rand(B, 3, S, S, generator=g),
             u1=torch.randn(B, 1, S, S, generator=g), u2=torch.randn(B, 3, 1, S, S, generator=g),
             u3=torch.randn(B, 4, S, S, generator=g), u4=torch.randn(B, 3, S, S, generator=g))
    # light cameras looking at the origin, the reference's [R | light_pos] convention (urhand.py:415)
    z = F.normalize(-t["lpos"].reshape(-1, 3), dim=-1)
    x = F.normalize(torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]).expand_as(z), z), dim=-1)
    t["lrt"] = torch.cat([torch.stack([x, torch.linalg.cross(z, x), z], 1), t["lpos"].reshape(-1, 3, 1)], 2)
    # synthetic hand stand-in: a bumpy closed genus-0 mesh (icosphere, 2562 vertices / 5120 faces) of ~90 mm radius
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import icosphere

    verts, faces = icosphere(4, radius=90.0)
    verts = verts * (1.0 + 0.08 * torch.sin(0.05 * verts[:, :1] + 0.07 * verts[:, 1:2]))
    t["verts"] = verts[None].repeat(B, 1, 1)
    t = {k: v.to(dev).contiguous() for k, v in t.items()}
    faces = faces.to(dev)
    Kl = torch.eye(3, device=dev)[None].repeat(B * L, 1, 1)            # shadowmap.py:21-26: focal 1000, 1024^2 light cameras
    Kl[:, 0, 0] = Kl[:, 1, 1] = 1000.0
    Kl[:, 0, 2] = Kl[:, 1, 2] = 512.0
    # ordinary [R | -R c] extrinsics of a camera AT the light, for the mesh render and the texel projection alike (the
    # reference hands [R | light_pos] to both, urhand.py:415; same arithmetic, but here the mesh is in view)
    cam_rt = torch.cat([t["lrt"][:, :, :3], -(t["lrt"][:, :, :3] @ t["lpos"].reshape(-1, 3, 1))], 2)
    for k in ("p_uv", "nml", "rough", "tex"):
        t[k].requires_grad_(True)

    def step():
        for k in ("p_uv", "nml", "rough", "tex"):
            t[k].grad = None
        with torch.no_grad():  # urhand.py:404: the shadow map is evaluated without gradients
            v_pix = meshraster.transform(t["verts"].repeat_interleave(L, 0), Kl, cam_rt)
            depth = meshraster.rasterize(v_pix, faces, 1024, 1024, with_bary=False)[1]
        shadow = shadowmap.shadow_pcf(depth, cam_rt, t["p_uv"], t["nml"], exp_scale=8.0).view(B, L, 1, S, S)
        diff, spec = uvlight.phong_features(t["p_uv"], t["nml"], t["cam"], t["lpos"], t["lint"], shadow)
        feat, rgb = uvlight.ggx_features(t["p_uv"], t["nml"], t["cam"], t["lpos"], t["lint"], t["rough"], t["tex"],
                                         shadow)
        torch.autograd.backward([diff, spec, feat, rgb], [t["u1"], t["u2"], t["u3"], t["u4"]])
        return rgb

    _, dt, ms, win = _time_steps(step, args)
    T = B * S * S
    sh = 4 * L * T  # the shadow map is the dominant stream: one float per texel per light
    alg = {"gol_mesh_raster": B * L * (8 * 1024 * 1024 + 64 * 5120),  # index + depth images out, face records
           "gol_shadow_pcf": 24 * T + sh + 9 * 4 * L * T,  # texels + shadow out + 9 nearest depth taps per light
           "gol_uvlight_phong_fwd": sh + 24 * T + 16 * T, "gol_uvlight_phong_bwd": sh + 24 * T + 16 * T + 24 * T,
           "gol_uvlight_ggx_fwd": sh + 40 * T + 28 * T, "gol_uvlight_ggx_bwd": sh + 40 * T + 28 * T + 40 * T}
    res = _secondary_line("URHand UV light-loop frames/sec (Phong + GGX, fwd+bwd), 1024x1024 texels x 32 lights",
                    "frames/s", B, args, dt, ms, alg,
                    {"workload": cfg["workload"], "uv": [S, S], "lights": L, "frames_per_gpu": B,
                     "shadow_depth_render": "32 x 1024^2 mesh z-buffer renders of a 5120-face closed mesh per frame"}, windows=win)
    if emit_line:
        emit(res)
    return res


SG_CFG = dict(workload="sgutils_native", gaussians=1_048_576, views=8, lights=8, seed=7)


def sg_main(args, emit_line=True):
    """Secondary workload: sgutils evaluate_gaussian fwd+bwd at the reference-native N=1,048,576 Gaussians,
    B=8 views, 8 point lights per view (sgutils.py:17-98; SURVEY 8d sgutils bytes)."""
    from goliath_amd import sg

    cfg = SG_CFG
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(cfg["seed"])
    B, N, L = cfg["views"], cfg["gaussians"], cfg["lights"]
    t = dict(dirs=F.normalize(torch.randn(B, N, 3, generator=g), dim=-1), sig=0.02 + 0.3 * torch.rand(B, N, generator=g),
             lval=torch.rand(B, L, 3, generator=g), lpts=1100.0 * F.normalize(torch.randn(B, L, 3, generator=g), dim=-1),
             pts=100.0 * torch.randn(B, N, 3, generator=g), nl=torch.full((B,), L, dtype=torch.int32),
             up=torch.randn(B, N, 3, generator=g))
    t = {k: v.to(dev).contiguous() for k, v in t.items()}
    t["dirs"].requires_grad_(True)
    t["sig"].requires_grad_(True)

    def step():
        t["dirs"].grad = t["sig"].grad = None
        out = sg.evaluate_gaussian(t["dirs"], t["sig"], t["lval"], t["lpts"], t["pts"], t["nl"])
        out.backward(t["up"])
        return out

    _, dt, ms, win = _time_steps(step, args)
    alg = {"gol_sg_eval_fwd": 40 * B * N, "gol_sg_eval_bwd": 56 * B * N}
    res = _secondary_line("sgutils evaluate_gaussian Gaussians/sec (fwd+bwd), 8 lights", "Gaussians/s", B * N, args, dt, ms, alg,
                    {"workload": cfg["workload"], "gaussians": N, "views": B, "lights": L}, windows=win)
    if emit_line:
        emit(res)
    return res


def mvp_main(args, emit_line=True):
    """Secondary workload (BASELINE config 5): MVP ray march fwd+bwd, 1 view of 2048x1334 per step."""
    from goliath_amd import _lib, mvp

    cfg = MVP_CFG
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    t = mvp_inputs(cfg, dev)
    H, W = cfg["height"], cfg["width"]

    def step():
        for k in ("primpos", "primrot", "primscale", "template"):
            t[k].grad = None
        rp, rd, tm = mvp.compute_raydirs(t["viewpos"], t["viewrot"], t["focal"], t["princpt"], (W, H), 1.0)
        out = mvp.mvpraymarch(rp, rd, 1.0 / 64, tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], None,
                              fadescale=8.0, fadeexp=8.0)
        (out - t["target"]).abs().mean().backward()
        return out

    out, dt, ms, win = _time_steps(step, args)
    P = H * W
    tpl_bytes = t["template"].numel() * 4
    alg = {"gol_mvp_march_fwd": 32 * P + 28 * P + tpl_bytes, "gol_mvp_march_bwd": 32 * P + 28 * P + 3 * tpl_bytes}
    res = _secondary_line("MVP ray-march views/sec (fwd+bwd) at 2048x1334, 4096 primitives", "views/s", 1, args, dt, ms, alg,
                    {"workload": cfg["workload"], "prims": cfg["prims"], "template": list(cfg["tdim"]),
                     "image": [H, W], "stepsize": 1.0 / 64, "mean_alpha": float(out[..., 3].mean())}, windows=win)
    if emit_line:
        emit(res)
    return res


E2E_CFG = dict(workload="rgca_e2e_native_slab1024", slab=1024, height=2048, width=1334, views_per_gpu=8, focal=3000.0,
               cam_radius_mm=700.0, n_mips=4, seed=4321)


def e2e_main(args, emit_line=True):
    """SURVEY 8d mode B at the reference-native size: 1024^2 = 1,048,576 Gaussians, 8 views of 2048x1334 per
    step, random-init decoder of the reference architecture (goliath_amd.decoder) -> shading tail -> render ->
    L1 + SSIM losses (rgca_example.yml:43-52) -> backward -> Adam step (torch.optim.Adam, lr 5e-4).  Geometry (postex / tn)
    is synthetic input: the mesh -> uv rasteriser is outside the path.  The two last transposed-conv layers run
    light-contracted on gol_tail_conv_*  unless --unfused-tail is given."""
    from goliath_amd import decoder, losses, parallel, render_gs, shade, splat

    cfg = dict(E2E_CFG, views_per_gpu=args.views)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(cfg["seed"])  # same initial weights on every rank
    dec = decoder.PrimDecoderConvs(base=cfg["slab"] // 128).to(dev)
    B, S, H, W = cfg["views_per_gpu"], cfg["slab"], cfg["height"], cfg["width"]
    N = S * S
    small = make_inputs(dict(CFG, slab=8, views_per_gpu=B), "cpu", rank)  # cameras, lights, env map
    g = torch.Generator(device=dev).manual_seed(cfg["seed"] + 1000 * rank)
    d = F.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1)
    pos = d * torch.rand(N, 1, device=dev, generator=g) ** (1 / 3) * torch.tensor([90.0, 120.0, 100.0], device=dev)
    t = {k: small[k].detach().to(dev) for k in ("light_sh", "K", "Rt", "campos", "lightrot")}
    t["mips"] = [m.to(dev) for m in small["mips"]]
    t["postex"] = pos.t().reshape(1, 3, S, S).expand(B, -1, -1, -1).contiguous()
    t["tn"] = F.normalize(pos, dim=-1).t().reshape(1, 3, S, S).expand(B, -1, -1, -1).contiguous()
    t["embs"] = torch.randn(B, 256, device=dev, generator=g)
    albedo = torch.nn.Parameter(0.2 + 0.6 * torch.rand(1, N, 3, device=dev, generator=g))
    params = list(dec.parameters()) + [albedo]
    with torch.no_grad():
        # target = the render of a nearby latent code, so the fit stays near the synthetic scene's statistics (fitting
        # uniform noise makes the optimiser inflate the Gaussians: 7x the intersections within 40 steps)
        f_vn0, f_vc0 = dec(t["embs"] + 0.3 * torch.randn(B, 256, device=dev, generator=g), t["campos"])
        p0 = shade.shading_tail(f_vn0, f_vc0, t["postex"], t["tn"], albedo, t["light_sh"], t["campos"],
                                preconv_envmap=t["mips"], lightrot=t["lightrot"])
        rgb0 = render_gs.render_batch(t["K"], t["Rt"], p0, H, W)[0]
        t["target"] = rgb0.clamp(0.0, 1.0)
        del f_vn0, f_vc0, p0, rgb0
    opt = torch.optim.Adam(params, lr=5e-4, fused=True)  # one multi-tensor kernel: same math as the default
    sync = parallel.GradSync(params)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def step():
        if world > 1:
            sync.zero_grad()   # gradients are views of the persistent communication buckets (parallel.GradSync)
        else:
            opt.zero_grad(set_to_none=True)
        ev[0].record()
        # (the cameras go into the shading call, as in AutoEncoder.forward: projection inside the shading kernels)
        vs = None if args.unfused_projection else render_gs.view_set(t["K"], t["Rt"], H, W)
        if args.fused_tail:
            from goliath_amd import tail

            x_vn, x_vc = dec.trunk(t["embs"], t["campos"])
            ev[1].record()
            preds = tail.fused_tail(dec.vnocond_mod[-1], dec.vcond_mod[-1], x_vn, x_vc, t["postex"], t["tn"], albedo, t["light_sh"], t["campos"],
                                    preconv_envmap=t["mips"], lightrot=t["lightrot"], views=vs)
        else:
            f_vn, f_vc = dec(t["embs"], t["campos"])
            ev[1].record()
            preds = shade.shading_tail(f_vn, f_vc, t["postex"], t["tn"], albedo, t["light_sh"], t["campos"],
                                       preconv_envmap=t["mips"], lightrot=t["lightrot"], views=vs)
        rgb, alpha, depth, l1 = render_gs.render_batch(t["K"], t["Rt"], preds, H, W, l1_target=t["target"])
        loss = 10.0 * l1                                         # rgca_example.yml:43-47  rgb_l1 weight 1e1
        if not args.no_ssim:
            loss = loss + 0.2 * (1.0 - losses.ssim_image(rgb, t["target"]))  # :48-52  rgb_ssim weight 2e-1
        loss.backward()
        ev[2].record()
        if world > 1:
            sync.sync()
        opt.step()
        ev[3].record()
        return loss

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    from goliath_amd import _lib

    loss_first = None
    for _ in range(args.warmup):
        l = step()
        loss_first = l.detach() if loss_first is None else loss_first
    barrier()
    _lib.TIMING = []
    seg = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l = step()
        if args.segments:  # per-segment event times need a sync per step: off in the headline number
            torch.cuda.synchronize()
            for i in range(3):
                seg[i] += ev[i].elapsed_time(ev[i + 1])
    barrier()
    dt = time.perf_counter() - t0
    timing, _lib.TIMING = _lib.TIMING, None
    if world > 1:
        import torch.distributed as dist

        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    per = {}
    for name, e0, e1 in timing:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    ms = {k: _median(v) for k, v in per.items()}
    if rank == 0:
        res = {"metric": "relit views/sec end-to-end (decode + shade + render + loss + backward + Adam) at 2048x1334, "
                         "1,048,576 Gaussians", "value": B * world * args.steps / dt, "unit": "views/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": cfg["workload"], "gaussians": N, "image": [H, W], "views_per_gpu": B,
                          "fused_tail": bool(args.fused_tail), "loss": "10*l1" if args.no_ssim else "10*l1 + 0.2*(1-ssim)",
                          "trainable_params": sum(p.numel() for p in params),
                          "parallelism": f"view-parallel x{world}", "rccl_world_size": world,
                          "grad_sync": "bucketed reduce-scatter + all-gather, launched from backward hooks"},
               "kernels_ms_per_call": ms,
               # the C-ABI calls of a step: per-call MEDIAN x calls per step (the mean rode on the first steps' outliers)
               "hot_path_ms_per_step": sum(_median(v) * len(v) / args.steps for v in per.values()),
               # sanity signal of the whole gradient chain: the training loss on the fixed batch, first vs last step
               "loss_first_step": float(loss_first) if loss_first is not None else None,
               "loss_last_step": float(l.detach()),
               # forwards whose intersection capacity overflowed and were repaired inside render_views (never an exception)
               "capacity_overflow_reruns": splat.PLANNER.reruns}
        if args.segments:
            res["segments_ms"] = {"decoder_fwd" if not args.fused_tail else "decoder_trunk_fwd": seg[0] / args.steps,
                                  "tail_render_loss_and_all_backward": seg[1] / args.steps,
                                  "grad_sync_and_adam": seg[2] / args.steps}
        # roofline of the longest call of the hot path at this size: algorithmic bytes with the list
        # entries of the FINAL state of the fit (one extra forward) over the live event time
        hot = {k: v for k, v in ms.items() if k in ("gol_shade_fwd", "gol_shade_bwd", "gol_project_fwd", "gol_project_bwd",
                                                    "gol_shade_project_fwd", "gol_shade_project_bwd",
                                                    "gol_bin_sort", "gol_rasterize_fwd", "gol_rasterize_bwd")}
        if hot:
            with torch.no_grad():
                f_vn1, f_vc1 = dec(t["embs"], t["campos"])
                p1 = shade.shading_tail(f_vn1, f_vc1, t["postex"], t["tn"], albedo, t["light_sh"], t["campos"],
                                        preconv_envmap=t["mips"], lightrot=t["lightrot"])
                intr = torch.stack([t["K"][:, 0, 0], t["K"][:, 1, 1], t["K"][:, 0, 2], t["K"][:, 1, 2]], -1)
                o1 = splat.render_views(p1["primpos"], p1["primscale"], p1["primqvec"], p1["opacity"], p1["color"],
                                        t["Rt"], intr, H, W)
                bins = o1["tile_bins"]
                I = float((bins[..., 1] - bins[..., 0]).sum(1).float().mean())
                pairs_taken = float(splat.raster_pair_counts(o1).double().mean(0)[1])
                res["config"]["tiles_with_more_than_2048_entries"] = float(((bins[..., 1] - bins[..., 0]) > 2048).float().sum(1).mean())
                del f_vn1, f_vc1, p1, o1, bins
            res["config"]["intersections_per_view"] = I
            res["config"]["pixel_gaussian_pairs_taken_per_view"] = pairs_taken
            dom = max(hot, key=hot.get)
            # PMC records taken AT THIS SIZE (tools/e2e_pmc.sh -> profiles/traffic_e2e.json / valu_e2e.json; the scene of the
            # record is the fit's first steps, the time is the live one of this run)
            res["roofline"] = make_roofline(dom, hot[dom], B, N, I, H * W, 0, pairs_taken, records="_e2e")
            # every hot-path call at this size: live time, PMC traffic of the recorded launch, and for the binning / raster
            # kernels the issue statistics that say what they wait for
            tr, va = _stamped("traffic_e2e.json"), _stamped("valu_e2e.json")
            if tr is not None:
                res["hbm_traffic_per_call_MB"] = {k: round(tr[k] / 1e6, 1) for k in hot if k in tr}
            if va is not None:
                ks = {}
                for kern, c in va.items():
                    if kern.startswith("_") or c.get("abi_call") not in hot or not c.get("SQ_BUSY_CYCLES"):
                        continue
                    ks[kern] = {"abi_call": c["abi_call"], "valu_inst": c.get("SQ_INSTS_VALU"),
                                # quad-cycles over 1024 SIMDs / XCD-summed active cycles (as in _valu_roofline)
                                "valu_busy": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 3)
                                if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE") else None,
                                "wait_inst_any_of_wave_cycles": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
                                if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES") else None}
                res["kernel_issue_stats_at_this_size"] = ks
        if emit_line:
            emit(res)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return res if rank == 0 else None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="N > 1 without torchrun: this script re-executes itself as N ranks (one per GPU) under "
                         "torch.distributed.run; under torchrun WORLD_SIZE must equal N.  Never a silent N = 1")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--views", type=int, default=None,
                    help="views per GPU and step.  Default: BASELINE config 2's batch of 8 views, sharded 8 / N per GPU for "
                         "N > 1 (config 3: strong scaling); given explicitly (or with --weak) every rank renders that many")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: every rank renders its own 8 views (weak scaling, 8 N views per step) instead of a share "
                         "of the config-2 batch")
    ap.add_argument("--micro", type=int, default=2, help="micro-batches (HIP streams) per step")
    ap.add_argument("--coherent-uv", action="store_true",
                    help="lay the synthetic Gaussians out over the slab like a UV atlas (slab neighbours = spatial neighbours) "
                         "instead of the default random permutation; the point set is the same")
    ap.add_argument("--smooth-normals", action="store_true",
                    help="decoder-like low-frequency normal offsets (f_vcond smooth over the slab) instead of SURVEY 8d's "
                         "white noise; with --coherent-uv this is the realistic case for the env-map gathers")
    ap.add_argument("--env-per-view", action="store_true",
                    help="every view gets its OWN env-map pyramid (rounds 1-4's workload; config 2 is ONE pyramid shared by "
                         "all views and rotated per view, the default since round 5)")
    ap.add_argument("--grad-floats", type=int, default=-1,
                    help="size of the gradient set exchanged per step besides the albedo map (default: 60 M fp32 = the "
                         "config-3 decoder parameter set when N > 1, 0 when N = 1)")
    ap.add_argument("--serial-exchange", action="store_true",
                    help="N > 1: the gradient exchange is launched and waited for inside the step (the default since "
                         "round 4: what a synchronous training loop can do)")
    ap.add_argument("--overlap-exchange", action="store_true",
                    help="N > 1: time ONLY the variant that overlaps step k's exchange with step k+1's compute (by default "
                         "it is measured after the serial headline and reported under `overlapped_exchange`)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional test mode for a 1-GPU box: the N ranks share the visible GPU(s) and exchange "
                         "gradients over gloo with device tensors (RCCL refuses two ranks on one device); the line says so "
                         "and is not a scaling measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group (RCCL) even for one rank and route the gradient exchange through it: "
                         "HIP-graph capture / replay next to a live RCCL communicator on a 1-GPU box")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test on CPU (gloo): no kernels, no measurement -- checks the rank plumbing and "
                         "the gradient exchange of the N-rank command line")
    ap.add_argument("--no-graph", action="store_true",
                    help="time eager launches instead of replays of the step captured as one HIP graph.  Per-call HIP events "
                         "cannot be taken inside a graph: they come from an eager pass right after the timed replays")
    ap.add_argument("--workload", choices=["rgca", "mvp", "urhand", "sg", "e2e"], default="rgca",
                    help="rgca = the BASELINE metric (default); mvp = secondary BASELINE config 5 line")
    ap.add_argument("--no-align", action="store_true",
                    help="rgca: do not join the micro-batch streams after the shading calls (see step())")
    ap.add_argument("--unfused-projection", action="store_true",
                    help="rgca: run the EWA projection and its backward as kernels of their own (gol_project_fwd / bwd inside "
                         "gol_render_fwd / bwd, rounds 1-3) instead of inside the shading kernels (gol_shade_project_fwd / bwd)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="rgca, N = 1: skip the secondary workloads (one camera per GPU, e2e at 1 M Gaussians, MVP config 5, "
                         "URHand config 4, the coherent / smooth env-map case) that the default line carries under `secondary`")
    ap.add_argument("--fused-tail", action="store_true", default=True,
                    help="e2e: light-contracted last decoder layers on gol_tail_conv_* (default)")
    ap.add_argument("--unfused-tail", dest="fused_tail", action="store_false",
                    help="e2e: last decoder layers as PyTorch/MIOpen transposed convs + bias adds (the reference's structure)")
    ap.add_argument("--no-ssim", action="store_true", help="e2e: L1 loss only (default: 10*L1 + 0.2*(1-SSIM))")
    ap.add_argument("--segments", action="store_true", help="e2e: also report per-segment times (adds a sync per step)")
    return ap.parse_args(argv)


CONFIG2_VIEWS = CFG["views_per_gpu"]   # BASELINE configs[1] / [2]: ONE batch of 8 views (config 3 = the same batch, 8 / N per GPU)


def plan_views(args, world):
    """(views per GPU, scaling, workload name) of the rgca line.  N = 1: config 2, 8 views.  N > 1: BASELINE config 3 as
    written -- the config-2 batch sharded 8 / N views per GPU, one camera per GPU at N = 8 ("strong": the total work is
    fixed).  --weak (or an explicit --views) keeps that many views on EVERY rank ("weak")."""
    if args.views is not None:
        return args.views, "weak", CFG["workload"]
    if world > 1 and not args.weak:
        return max(1, CONFIG2_VIEWS // world), "strong", "rgca_config3_viewparallel"
    return CONFIG2_VIEWS, "weak", CFG["workload"]


def run_rgca(args, D, views, scaling, workload, overlap, micro=None, extras=True):
    """Warm up, capture and time the rgca step for `views` views per rank; returns the result dict (rank 0; None on the
    other ranks).  overlap: exchange mode for N > 1 (see run_step).  extras: intersection counts, per-call HBM table."""
    cfg = dict(CFG, workload=workload, views_per_gpu=views, coherent_uv=bool(args.coherent_uv),
               smooth_normals=bool(args.smooth_normals), fused_projection=not args.unfused_projection,
               env_per_view=bool(getattr(args, "env_per_view", False)))
    # with the projection fused in, micro-batches of >= 4 views are joined once after their shading kernels (see step();
    # measured: 8 views 2803 -> 2855 views/s; 2 + 2 and 1 + 1 views are faster left alone, and so is the unfused path)
    cfg["align_micro_batches"] = (cfg["fused_projection"] and views // max(1, min(args.micro if micro is None else micro, views)) >= 4
                                  and not args.no_align)
    micro = args.micro if micro is None else micro
    micro = max(1, min(micro, views))
    while views % micro:  # micro-batches must divide the views of a rank
        micro -= 1
    world, rank, dev = D.world, D.rank, D.device
    on_gpu = dev.type == "cuda"
    from goliath_amd import parallel

    if args.stub:
        t = {"stub": True, "rank": rank, "albedo": torch.nn.Parameter(torch.arange(1000.0) / 1000.0), "micro": []}
    else:
        from goliath_amd import _lib, splat

        t = make_step_inputs(cfg, dev, rank, micro)

    # Gradient exchange of the step.  Mode A has a single trainable tensor on the path (the albedo map, 3 MB); BASELINE
    # config 3 specifies the exchange of the 512^2-slab RGCA decoder parameter set (~60 M fp32) every step, so
    # for N > 1 a gradient set of that size travels with it (its values are irrelevant to the timing; in training it is
    # produced by the decoder backward, which is not part of mode A) -- a SCALE line must pay the real message size.
    exchanging = D.backend != "none"
    grad_floats = args.grad_floats if args.grad_floats >= 0 else (60_000_000 if world > 1 and not args.stub else 0)
    dec = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in _split(grad_floats, 4)]
    for p in dec:
        p.grad = torch.zeros_like(p)
    t["_sync"] = parallel.GradSync([t["albedo"]] + dec, single_rank_collectives=args.force_dist) if exchanging else None
    t["overlap_exchange"] = exchanging and overlap
    B, N = cfg["views_per_gpu"], cfg["gaussians"]
    P = cfg["height"] * cfg["width"]

    def drain():
        if t["_sync"] is not None and t["overlap_exchange"]:
            t["_sync"].wait()   # the last step's exchange belongs to the timed region

    def mark():
        if not on_gpu:
            return time.perf_counter()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    for _ in range(args.warmup):
        run_step(t, cfg, world)
    drain()
    D.barrier()
    graph = None
    if not args.no_graph and on_gpu:
        # The step is ~60 launches per micro-batch; issued from Python they cost about as much host time as the GPU
        # needs to execute them.  Capture the whole compute step (both streams, forward + backward) in ONE HIP graph
        # and replay it: the timed loop is then launch-overhead-free.  Capacities are frozen at their calibrated
        # values during capture; overflow is checked after the replays.
        splat.PLANNER.poll(block=True)
        splat.PLANNER.frozen = True
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: other threads (the RCCL watchdog of a multi-GPU run) keep issuing HIP calls meanwhile
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                step(t, cfg, world)
            run_step(t, cfg, world, graph)  # one untimed replay
            drain()
        except Exception as e:  # capture is an optimisation: fall back to eager issue
            print(f"bench.py: HIP-graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            graph = None
            splat.PLANNER.frozen = False
            splat.PLANNER.frozen_log.clear()
            torch.cuda.synchronize()
            run_step(t, cfg, world)
            drain()
        D.barrier()
    # the K timed steps are ONE region (barrier + synchronize on both sides); events recorded between up to five windows
    # of it give the spread without adding a sync
    plan = _window_plan(args.steps)
    t0 = time.perf_counter()
    marks = [mark()]
    for n in plan:
        for _ in range(n):
            run_step(t, cfg, world, graph)
        marks.append(mark())
    drain()
    D.barrier()
    dt = time.perf_counter() - t0
    win_ms = [(marks[i].elapsed_time(marks[i + 1]) if on_gpu else 1e3 * (marks[i + 1] - marks[i])) / n
              for i, n in enumerate(plan)]
    exchange_ms = None
    if exchanging and args.steps > 0:
        # the exchange alone, serial, right after the timed region (same buffers): what it costs when nothing hides it
        D.barrier()
        te = time.perf_counter()
        for _ in range(args.steps):
            t["_sync"].launch_all()
            t["_sync"].wait()
        D.barrier()
        exchange_ms = 1e3 * D.max_over_ranks(time.perf_counter() - te) / args.steps
    timing = []
    if on_gpu and not args.stub:
        if graph is not None:
            splat.PLANNER.check_frozen()  # raises if a replayed render overflowed its intersection capacity
            splat.PLANNER.frozen = False
        # The timed region runs UN-instrumented (graph replays, or the eager composite calls an unmodified training loop
        # issues).  The per-call durations come from an eager, instrumented pass of the same K steps right after it (same
        # buffers; HIP events cannot bracket nodes inside a captured graph, and the per-stage calls the events need are not
        # what the product path issues).  The micro-batches go on ONE stream here, so a call's events bracket that kernel
        # sequence alone and the durations agree with rocprofv3's per-kernel trace; with two streams in flight they would
        # include time-sharing.
        streams = t["streams"]
        if os.environ.get("GOL_TIMING_STREAMS", "0") != "1":   # (=1, diagnostics: keep the streams, durations include time-sharing)
            t["streams"] = [torch.cuda.current_stream()] * len(t["streams"])
        sync_keep, t["_sync"] = t.get("_sync"), None   # compute only
        # one untimed eager step first: the eager path's own warm-up after the graph replays (workspace / planner state of the
        # un-captured calls; a one-off host-side allocation inside an event bracket once put 0.5 ms on a 0.34 ms average)
        run_step(t, cfg, world)
        torch.cuda.synchronize()
        _lib.TIMING = []
        for _ in range(max(args.steps, 5)):   # >= 5 instrumented passes; the per-call figure is their median
            run_step(t, cfg, world)
        t["_sync"] = sync_keep
        D.barrier()
        t["streams"] = streams
        timing, _lib.TIMING = _lib.TIMING, None
        splat.PLANNER.poll(block=True)  # raises if any step overflowed its intersection capacity
    dt = D.max_over_ranks(dt)
    kernels_ms = _per_call_ms(timing)
    if os.environ.get("GOL_TIMING_STREAMS", "0") == "1" and timing and rank == 0:
        # diagnostics: the last step's calls on a common clock (start, end in us since the step's first call)
        per_step = len(timing) // max(args.steps, 5)
        last = timing[-per_step:]
        base = last[0][1]
        for name, e0, e1 in sorted(last, key=lambda x: base.elapsed_time(x[1])):
            print("  %8.1f %8.1f  %s" % (1e3 * base.elapsed_time(e0), 1e3 * base.elapsed_time(e1), name), file=sys.stderr)
    stub_ok = None
    if args.stub:
        # every rank's "gradient" was albedo * (rank + 1): the average is albedo * (world + 1) / 2
        want = t["albedo"].detach() * (world + 1) / 2.0
        stub_ok = bool(torch.allclose(t["albedo"].grad, want, atol=1e-6)) if exchanging else None
    if rank != 0:
        return None
    views_total = B * world * args.steps
    par = {"parallelism": f"view-parallel x{world}", "rccl_world_size": world if D.backend == "nccl" else 0,
           "world_size": world, "dist_backend": D.backend, "views_per_gpu": B, "views_per_step": B * world,
           "ranks_share_gpu": bool(D.shared_gpu),
           "grad_exchange_bytes_per_step": 4 * (grad_floats + t["albedo"].numel()) if exchanging else 0,
           "grad_exchange": None if not exchanging else (
               "reduce-scatter + all-gather of step k overlapped with the compute of step k+1 (waited before the next "
               "launch; the last one inside the timed region)" if t["overlap_exchange"] else
               "reduce-scatter + all-gather inside the step (serial): launched after the step's backward and waited for "
               "before the next step starts"),
           "grad_exchange_ms_alone": exchange_ms}
    base = {"value": views_total / dt, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None}
    if args.stub:
        return dict(base, metric="STUB launcher self-test (no kernels ran; not a measurement)", unit="stub steps x views/s",
                    dtype="f32", data="stub", config=dict(par, workload="stub"), stub_gradients_averaged=stub_ok)
    res = dict(base, metric="relit views/sec (fwd+bwd) at 2048x1334, 250k Gaussians", dtype="f32", data="synthetic")
    config = dict({"workload": cfg["workload"], "gaussians": N, "image": [cfg["height"], cfg["width"]],
                   "micro_batches": micro,
                   "launch": ("eager composite calls" if graph is None else "hip_graph_replay") +
                             " (kernels_ms_per_call / roofline: eager instrumented pass after the timed region)",
                   "relight": "envmap_4mips, " + ("one pyramid per view (rounds 1-4)" if cfg["env_per_view"] else
                                                  "ONE pyramid shared by all views, per-view lightrot (config 2)"),
                   "slab_layout": "uv-coherent" if args.coherent_uv else "random permutation",
                   "normal_offsets": "smooth (decoder-like)" if args.smooth_normals else "white noise (SURVEY 8d)"}, **par)
    res["config"] = config
    res["windows"] = _window_stats(win_ms)
    res["kernels_ms_per_call"] = kernels_ms
    views_per_launch = B // micro  # every ABI call processes one micro-batch
    I = 1.97e6   # stored list entries per view of the default scene (measured below when `extras`)
    pairs_taken = None
    if extras:
        # measured intersection count (determines the raster/sort work)
        with torch.no_grad():
            from goliath_amd import render_gs, shade

            m0 = t["micro"][0]
            preds = shade.shading_tail(m0["f_vn"], m0["f_vc"], m0["postex"], m0["tn"], m0["albedo"], m0["light_sh"],
                                       m0["campos"], preconv_envmap=m0["mips"], lightrot=m0["lightrot"])
            intr = torch.stack([m0["K"][:, 0, 0], m0["K"][:, 1, 1], m0["K"][:, 0, 2], m0["K"][:, 1, 2]], -1)
            out = splat.render_views(preds["primpos"], preds["primscale"], preds["primqvec"], preds["opacity"],
                                     preds["color"], m0["Rt"], intr, cfg["height"], cfg["width"])
            # list entries actually stored (n_isect is the RESERVED slot count since round 3: the tight tile boxes)
            bins = out["tile_bins"]
            I = float((bins[..., 1] - bins[..., 0]).sum(1).float().mean())
            config["intersections_per_view"] = I
            config["list_slots_reserved_per_view"] = float(out["n_isect"].float().mean())
            config["mean_alpha"] = float(out["alpha"].mean())
            pc = splat.raster_pair_counts(out).double().mean(0)
            config["pixel_gaussian_pairs_tested_per_view"], pairs_taken = float(pc[0]), float(pc[1])
            config["pixel_gaussian_pairs_taken_per_view"] = pairs_taken
            del preds, out, bins
    if kernels_ms:
        dom = max(kernels_ms, key=kernels_ms.get)
        mip_bytes = sum(m[0].numel() * 4 for m in t["micro"][0]["mips"])
        res["roofline"] = make_roofline(dom, kernels_ms[dom], views_per_launch, N, I, P, mip_bytes, pairs_taken)
        # every streaming call beside it: algorithmic GB/s, fraction of the 8 TB/s spec, PMC traffic / algorithmic bytes
        per = {}
        # PMC traffic of the layout that was run (the default random-permutation slab, or the coherent / smooth one)
        tr = _stamped("traffic_coherent_smooth.json" if (args.coherent_uv and args.smooth_normals) else "traffic.json")
        for k, ms in kernels_ms.items():
            try:
                alg = views_per_launch * algorithmic_bytes(k, N, I, P, mip_bytes)
            except KeyError:
                continue
            gbs = alg / (ms * 1e-3) / 1e9
            per[k] = {"GBs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                      "traffic_over_algorithmic": None if tr is None or k not in tr else
                      round(tr[k] * views_per_launch / 8.0 / alg, 3)}
        res["hbm_per_call"] = per
    if D.shared_gpu:
        res["note"] = ("ranks share one GPU and exchange over gloo: a functional run of the N-rank path, NOT a scaling "
                       "measurement")
    return res


def _release():
    import gc

    try:   # the packed env-map levels of the workload that just ended (the cache keeps its source tensors alive)
        from goliath_amd import shade

        shade.invalidate_envmap_cache()
    except Exception:
        pass
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def _brief(res, keys=("value", "unit", "ms_per_step", "scaling", "kernels_ms_per_call", "roofline", "hbm_per_call", "windows")):
    """A secondary entry of the main line: the measurement without the boiler-plate fields."""
    out = {k: res[k] for k in keys if k in res}
    out["metric"] = res.get("metric")
    out["config"] = res.get("config")
    for k in ("loss_first_step", "loss_last_step", "hot_path_ms_per_step", "steps"):
        if k in res:
            out[k] = res[k]
    return out


def secondary_workloads(args, D):
    """The other BASELINE configurations, measured by the SAME default command right after the headline (N = 1) and
    attached to its line under `secondary` -- each with its own ms_per_step, per-call times and roofline:
      views1    BASELINE config 3's per-GPU share at N = 8: ONE camera per GPU and step (no second micro-batch to hide a
                view's small-grid phases behind)
      e2e       SURVEY 8d mode B at the reference-native 1,048,576 Gaussians: decoder -> fused tail -> render -> L1 + SSIM
                -> backward -> Adam, >= 60 steps so that the loss is past Adam's first overshoot
      mvp       BASELINE config 5 (4096 primitives, 2048x1334)
      urhand    BASELINE config 4 (1024^2 texels x 32 lights, incl. the 32 shadow depth renders)
      env_distinct_maps     the headline step with a DIFFERENT env-map pyramid per view (rounds 1-4's headline workload;
                harder than config 2, whose single pyramid stays cache-resident)
      env_coherent_smooth   the headline step on a UV-coherent slab with decoder-like smooth normal offsets: the realistic
                case for the env-map gathers of shade_fwd (the default layout is the synthetic worst case)"""
    import copy

    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as e:  # a secondary workload must never take the headline down with it
            print(f"bench.py: secondary workload {name} failed: {type(e).__name__}: {e}", file=sys.stderr)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        _release()

    guarded("views1", lambda: _brief(run_rgca(args, D, 1, "weak", "rgca_config3_one_view_per_gpu", False, micro=1)))
    a2 = copy.copy(args)
    a2.coherent_uv = a2.smooth_normals = True
    guarded("env_coherent_smooth", lambda: _brief(run_rgca(a2, D, CONFIG2_VIEWS, "weak", "rgca_config2_coherent_smooth",
                                                           False)))
    a4 = copy.copy(args)
    a4.env_per_view = True
    guarded("env_distinct_maps", lambda: _brief(run_rgca(a4, D, CONFIG2_VIEWS, "weak", "rgca_8_distinct_envmaps", False,
                                                         extras=False)))
    guarded("mvp", lambda: _brief(mvp_main(args, emit_line=False)))
    guarded("urhand", lambda: _brief(urhand_main(args, emit_line=False)))
    a3 = copy.copy(args)
    a3.steps, a3.views = max(args.steps, 60), CONFIG2_VIEWS
    guarded("e2e", lambda: _brief(e2e_main(a3, emit_line=False)))
    return out


def main():
    args = parse_args()
    from goliath_amd import launch

    # N > 1 asked for and not started by torchrun: become the launcher of N ranks (the driver's own command line)
    rc = launch.maybe_spawn(args.gpus)
    if rc is not None:
        sys.exit(rc)
    _claim_stdout()  # a worker from here on: stdout carries the one JSON line and nothing else
    if args.workload != "rgca":
        if args.views is None:
            args.views = CONFIG2_VIEWS
        return {"mvp": mvp_main, "urhand": urhand_main, "sg": sg_main, "e2e": e2e_main}[args.workload](args)
    try:
        D = launch.init(args.gpus, share_gpu=args.share_gpu, cpu=args.stub, force_group=args.force_dist)
    except launch.LaunchError as e:
        raise SystemExit(f"bench.py: {e}")
    world = D.world
    views, scaling, workload = plan_views(args, world)
    multi = D.backend != "none"
    # N > 1: the headline pays the exchange INSIDE the step (what a synchronous training loop can do: the optimizer needs
    # the averaged gradients before the next forward); the overlapped variant (step k's exchange beside step k+1's
    # compute) is measured right after it and reported beside it, never as `value`
    overlap_only = bool(args.overlap_exchange) and not args.serial_exchange
    res = run_rgca(args, D, views, scaling, workload, overlap=overlap_only)
    if args.stub:
        if D.rank == 0:
            emit(res)
        D.shutdown()
        if D.rank == 0 and res.get("stub_gradients_averaged") is False:
            sys.exit(3)
        return
    _release()
    if multi and not overlap_only and not args.serial_exchange:
        ov = run_rgca(args, D, views, scaling, workload, overlap=True, extras=False)
        if D.rank == 0:
            res["overlapped_exchange"] = {k: ov[k] for k in ("value", "unit", "ms_per_step", "windows")}
            res["overlapped_exchange"]["grad_exchange"] = ov["config"]["grad_exchange"]
        _release()
    if world > 1 and scaling == "strong" and not D.shared_gpu:
        # the previous rounds' line, for continuity: every rank renders its own 8 views (8 N views per step)
        wk = run_rgca(args, D, CONFIG2_VIEWS, "weak", CFG["workload"], overlap=overlap_only, extras=False)
        if D.rank == 0:
            res["weak"] = {k: wk[k] for k in ("value", "unit", "ms_per_step", "scaling", "windows")}
            res["weak"]["views_per_gpu"] = CONFIG2_VIEWS
        _release()
    if D.rank == 0:
        if world == 1 and not args.no_secondary and args.views is None and not (args.coherent_uv or args.smooth_normals):
            res["secondary"] = secondary_workloads(args, D)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(dict(CFG))
        if world == 1:
            # parity of THIS workload's chain against the CPU oracle (tests/test_gpu_fullsize.py at config-2 size, same
            # kernels by digest) -- quoted ONCE, in the detail record only; null when profiles/fullsize_parity.json was taken
            # on other kernel sources
            par_rec = _stamped("fullsize_parity.json")
            res["chain_parity_vs_oracle"] = None if par_rec is None else {
                "outputs": par_rec.get("outputs"),
                "leaf_gradients": {k: {kk: v.get(kk) for kk in ("rel_l2_all_gaussians", "rel_l2_without_flagged",
                                                                "flagged_fraction", "W_fraction", "W_unexplained")}
                                   for k, v in par_rec.get("grads", {}).items()}}
        emit(res)
    D.shutdown()


if __name__ == "__main__":
    main()
