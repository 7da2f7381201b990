"""CPU: the C-ABI library builds, loads, and exports every symbol include/goliath_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "goliath_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gol_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from goliath_amd import build, _lib

    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in goliath_hip.h but not exported"
    # the python binding knows the same set
    assert set(names) == set(_lib.exported_symbols())
    assert b"gfx950" in ctypes.cast(lib.gol_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "goliath_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_missing_gpu_fails_loudly():
    import pytest
    import torch

    from goliath_amd import sg, splat, _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError):
        sg.evaluate_gaussian(x, torch.ones(1, 4), torch.ones(1, 1, 3), torch.ones(1, 1, 3), x,
                             torch.ones(1, dtype=torch.int32))
    with pytest.raises(_lib.GoliathHipError):
        splat.render_views(torch.zeros(1, 4, 3), torch.ones(1, 4, 3), torch.ones(1, 4, 4), torch.ones(1, 4),
                           torch.ones(1, 4, 3), torch.eye(4)[:3][None], torch.ones(1, 4), 32, 32)
