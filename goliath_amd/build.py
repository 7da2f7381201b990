"""Build libgoliath_hip.so (the C-ABI library of include/goliath_hip.h) for gfx950 with hipcc.

In-tree build: goliath_amd/csrc/*.hip -> goliath_amd/csrc/_obj/*.o -> goliath_amd/lib/libgoliath_hip.so
so the shared object travels with the source tree (no JIT cache).  hipcc cross-compiles without
a GPU.  Usage:  python -m goliath_amd.build [--force] [--verbose]
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgoliath_hip.so")

# Build variants.  "" = the product library.  "exact" = a TEST-ONLY twin (libgoliath_hip_exact.so, -DGOL_EXACT_MATH): the
# rasterizer evaluates sigma in the oracle's operation order without contraction, exp through double precision and the
# transmittance recurrence as T (1 - alpha) -- no fast-math threshold flips -- so tests/test_gpu_exact_math.py can show that
# the gradient residuals of the fast build are flips and nothing else.  Loaded only when GOLIATH_HIP_LIB points at it.
VARIANTS = {"": [], "exact": ["-DGOL_EXACT_MATH=1"]}


def lib_path(variant=""):
    return os.path.join(LIBDIR, f"libgoliath_hip{'_' + variant if variant else ''}.so")


HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
    "-ffp-contract=fast", "-Wall", "-Wno-unused-function",
]


# per-file extra flags (none at present).  Measured and rejected: compiling raster.hip without packed fp32 ops
# ("-Xclang -target-feature -Xclang -packed-fp32-ops") -- the isolated probe prices v_pk_fma_f32 at 2.6x a plain FMA
# (profiles/r02c_valu_probe.txt), but inside the raster loops both builds take the same time (0.741 / 1.303 ms vs
# 0.739 / 1.302 ms per 8 views): in situ a packed op costs two plain ones, so only the operation COUNT matters.
EXTRA_FLAGS = {}


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "goliath_hip.h"))
    return hs


def source_digest():
    """sha256 (first 16 hex digits) over the kernel sources + the ABI header: identifies the code a profile under
    profiles/ was measured on (the .git directory does not travel to the GPU box, file contents do)."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(_sources() + _headers()):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose, variant=""):
    obj = os.path.join(OBJ + ("_" + variant if variant else ""), os.path.basename(src)[:-4] + ".o")
    cmd = [HIPCC, *FLAGS, *VARIANTS[variant], *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force=False, verbose=False, variant=""):
    """Compile every HIP source for gfx950 and link the C-ABI shared library.  Returns its path."""
    objdir = OBJ + ("_" + variant if variant else "")
    lib = lib_path(variant)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs, hdrs = _sources(), _headers()
    todo = [s for s in srcs
            if force or _stale(os.path.join(objdir, os.path.basename(s)[:-4] + ".o"), [s, *hdrs])]
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda s: _compile(s, verbose, variant), todo))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or todo or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    for v in (["", "exact"] if "--all" in sys.argv else ["exact"] if "--exact" in sys.argv else [""]):
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, variant=v))
