import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- parity ledger (VERDICT r3 "next" item 5) ---------------------------------------------------------------------------
# Every rel-L2 a `-m gpu` test measures (tests/scenes.py:rel_l2 -- what nearly every parity assertion is written with) is
# recorded with the test that took it, the source line that judges it (which carries the bar) and the SURVEY section-8 row
# the test file belongs to; the session writes gpurun_out/parity_ledger.json (merged back by gpurun; the copy under
# profiles/ is the tracked one).  Tests with their own reports (exact-math twin, full-size chain, MVP crops) are merged in
# from the JSON files they leave in gpurun_out/.
_ROWS = {
    "test_gpu_shade": "S / E  shading tail + env-map specular (shade.hip)", "test_gpu_sg": "R6  evaluate_gaussian (sg.hip)",
    "test_gpu_splat": "R1-R5  project / bin+sort / raster fwd+bwd vs oracle/gsplat_oracle.c (unpinned)",
    "test_gpu_edges": "R0-R5  edge cases of the render path", "test_gpu_exact_math": "R3-R5  raster boundary, identical inputs",
    "test_gpu_fullsize": "R0-R5 + S  the bench step at config 2 / config 1 size",
    "test_gpu_model_forward": "SELF-CONSISTENCY (no oracle): fused vs layer-by-layer host paths of the model glue",
    "test_gpu_rgca_dropin": "SELF-CONSISTENCY (no oracle): drop-in plumbing on a stand-in (parity lives in test_gpu_rgca_model_golden)",
    "test_gpu_rgca_model_golden": "R0 / (b)  AutoEncoder.forward / render / PrimDecoder.forward / env-relight driver vs the "
                                  "reference's own code (tests/golden/rgca_model_golden.npz)",
    "test_gpu_mvp": "M1-M4  mvpraymarch / aabb / raydirs (mvp.hip)", "test_gpu_mvp_fullsize": "M2  config-5 size crops",
    "test_gpu_uvlight": "U  URHand light loops (uvlight.hip)", "test_gpu_shadow": "U  shadow-map PCF (shadow.hip)",
    "test_gpu_urhand_model": "(+) URHand / teacher model-level bindings", "test_gpu_meshraster": "f4  mesh depth render (unpinned)",
    "test_gpu_tail": "f1  light-contracted decoder tail (tail.hip)", "test_gpu_ssim": "f3  SSIM (ssim.hip)",
    "test_gpu_losses": "f3  masked L1 (imgloss.hip)", "test_imgtail": "f3  image tail: CalV5 + LearnableBlur (imgtail.hip)",
    "test_gpu_fused_projection": "S -> R2  projection fused into the shading kernels vs the separate kernels",
    "test_gpu_e2e_descent": "8d mode B  end-to-end descent", "test_gpu_multirank": "8e  N-rank path on one GPU",
}
_LEDGER = []


def _install_ledger():
    import linecache

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes

    plain = scenes.rel_l2

    def rel_l2(a, b):
        v = plain(a, b)
        f = sys._getframe(1)
        fn = os.path.basename(f.f_code.co_filename)
        _LEDGER.append({"file": fn, "line": f.f_lineno, "test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0],
                        "source": linecache.getline(f.f_code.co_filename, f.f_lineno).strip()[:160], "rel_l2": v})
        return v

    scenes.rel_l2 = rel_l2


_install_ledger()


def pytest_sessionfinish(session, exitstatus):
    import json

    import torch

    out_dir = os.path.join(ROOT, "gpurun_out")
    if not _LEDGER or not torch.cuda.is_available() or not os.path.isdir(out_dir):
        return
    rows = {}
    for e in _LEDGER:
        key = f"{e['file']}:{e['line']}"
        r = rows.setdefault(key, {"row": _ROWS.get(e["file"][:-3], "?"), "judged_by": e["source"], "tests": [], "n": 0,
                                  "rel_l2_max": 0.0})
        r["n"] += 1
        # NaN must not disappear: max(x, nan) == x in Python, i.e. a NaN rel-L2 would be ledgered as perfect parity
        if e["rel_l2"] != e["rel_l2"] or r["rel_l2_max"] != r["rel_l2_max"]:
            r["rel_l2_max"], r["nan"] = float("nan"), True
        else:
            r["rel_l2_max"] = max(r["rel_l2_max"], e["rel_l2"])
        t = e["test"].split("::", 1)[-1]
        if t not in r["tests"] and len(r["tests"]) < 6:
            r["tests"].append(t)
    merged = {}
    for name in ("exact_math_parity.json", "fullsize_parity.json", "chain_parity_config1.json", "mvp_parity.json",
                 "rgca_model_parity.json"):
        path = os.path.join(out_dir, name)
        if os.path.exists(path) and os.path.getmtime(path) >= session.config._ledger_t0:
            merged[name] = json.load(open(path))
    by_row = {}
    for key, r in sorted(rows.items()):
        by_row.setdefault(r["row"], {})[key] = {k: r[k] for k in ("rel_l2_max", "n", "judged_by", "tests", "nan") if k in r}
    try:
        from goliath_amd import build

        digest = build.source_digest()
    except Exception:
        digest = None
    json.dump({"what": "rel-L2 measured by every parity assertion of this `pytest -m gpu` session (max over its calls), by "
                       "SURVEY section-8 row; `judged_by` is the asserting source line (it carries the bar)",
               "csrc_sha16": digest, "exit_status": int(exitstatus), "measurements": len(_LEDGER), "rows": by_row,
               "reports_of_tests_with_their_own_protocol": merged},
              open(os.path.join(out_dir, "parity_ledger.json"), "w"), indent=1)


def pytest_sessionstart(session):
    import time

    session.config._ledger_t0 = time.time() - 1.0
