"""CPU: the N-rank command line of bench.py (`--gpus N`) -- it must launch N ranks itself or fail loudly, never report a
one-rank run as N GPUs (the driver's SCALE runs use exactly this command).  The kernels need a GPU, so the
launcher is driven here with `--stub` (no kernels, gloo, CPU tensors): rank plumbing + the gradient exchange."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ if env is None else env, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        if env is None:
            e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def _json_line(stdout):
    # the contract: ONE JSON line on stdout and NOTHING else -- a driver may take the last line (RCCL's version banner,
    # written through C stdio and flushed at exit, used to follow the JSON: the workers now point fd 1 at stderr)
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_exchanges_gradients():
    r = _run(["--gpus", "2", "--stub", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2
    assert line["config"]["dist_backend"] == "gloo" and line["data"] == "stub"
    assert line["stub_gradients_averaged"] is True
    # the default N > 1 line pays the exchange inside the step (a synchronous training loop cannot overlap it with the
    # next step's compute: ADVICE r3)
    assert "serial" in line["config"]["grad_exchange"]
    assert line["config"]["grad_exchange_bytes_per_step"] == 4000
    # BASELINE config 3 as written: the config-2 batch of 8 views sharded 8 / N per GPU, total work fixed
    assert line["config"]["views_per_gpu"] == 4 and line["config"]["views_per_step"] == 8
    assert line["scaling"] == "strong"


def test_gpus_2_weak_keeps_8_views_per_rank():
    r = _run(["--gpus", "2", "--stub", "--steps", "2", "--warmup", "1", "--weak"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["config"]["views_per_gpu"] == 8 and line["config"]["views_per_step"] == 16 and line["scaling"] == "weak"


def test_gpus_2_overlapped_exchange_on_request():
    r = _run(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1", "--overlap-exchange"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert "overlapped" in line["config"]["grad_exchange"] and line["stub_gradients_averaged"] is True


def test_view_plan_of_the_scaling_runs():
    """--gpus 1 / 2 / 4 / 8 = BASELINE configs[1] / [2]: 8, 4, 2, 1 views per GPU, 8 views per step throughout."""
    sys.path.insert(0, ROOT)
    import bench

    for n, per in ((1, 8), (2, 4), (4, 2), (8, 1)):
        a = bench.parse_args(["--gpus", str(n)])
        views, scaling, workload = bench.plan_views(a, n)
        assert views == per and views * n == 8
        assert scaling == ("weak" if n == 1 else "strong")
        assert workload == ("rgca_config2_envrelight" if n == 1 else "rgca_config3_viewparallel")
    a = bench.parse_args(["--gpus", "8", "--weak"])
    assert bench.plan_views(a, 8) == (8, "weak", "rgca_config2_envrelight")
    a = bench.parse_args(["--gpus", "4", "--views", "3"])
    assert bench.plan_views(a, 4)[:2] == (3, "weak")


def test_gpus_2_serial_exchange():
    r = _run(["--gpus", "2", "--stub", "--steps", "2", "--warmup", "1", "--serial-exchange"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and "serial" in line["config"]["grad_exchange"] and line["stub_gradients_averaged"] is True


def test_gpus_2_without_gpus_fails_loudly():
    """No GPU here: the real workload must refuse (non-zero exit, a reason), not fall back to one rank or to the CPU."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout
    assert "no GPU visible" in r.stderr or "needs" in r.stderr, r.stderr[-2000:]


def test_library_chatter_on_stdout_does_not_reach_the_json_stream():
    """A worker's fd 1 is stderr: whatever a library writes to C stdout (RCCL's banner) lands there, not before or after
    the line.  PYTHONSTARTUP-free simulation: the stub run with a site hook that writes to fd 1 at interpreter exit."""
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "sitecustomize.py"), "w") as f:
            f.write("import atexit, os\natexit.register(lambda: os.write(1, b'RCCL version : banner\\n'))\n")
        e = dict(os.environ, PYTHONPATH=d + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = _run(["--stub", "--steps", "2", "--warmup", "1"], env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["data"] == "stub"
    assert "RCCL version : banner" in r.stderr


def test_world_size_mismatch_is_an_error():
    """Started by a launcher with a different world size than --gpus says: refuse to mislabel the line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = _run(["--gpus", "4", "--stub", "--steps", "1", "--warmup", "0"], env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr, r.stderr[-2000:]


def test_spawn_command_is_the_drivers_command_line():
    sys.path.insert(0, ROOT)
    from goliath_amd import launch

    cmd = launch.spawn_command(8, ["bench.py", "--gpus", "8"], port=1234)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                       "--master-port", "1234", "bench.py", "--gpus", "8"]
    assert launch.maybe_spawn(1) is None


def test_one_line_record_is_short_strict_json():
    """the driver could not parse a 22.5 KB line.  The stdout line must stay under 8 KB, be strict JSON
    (no NaN / Infinity tokens) and carry value / roofline / cpu_baseline; everything else goes to the detail record.  Fed
    with the largest full records under profiles/ (round 5's 22.5 KB line, and round 6's detail record once it exists)."""
    import glob
    import math

    sys.path.insert(0, ROOT)
    import bench

    fulls = [os.path.join(ROOT, "profiles", "r05z_bench.json")] + sorted(glob.glob(os.path.join(ROOT, "profiles", "r06*_bench_detail.json")))
    for path in fulls:
        full = json.load(open(path))
        full["config"]["poison"] = float("nan")              # must not reach the line as a NaN token
        full["kernels_ms_per_call"]["gol_bin_sort"] = float("inf")
        text = json.dumps(bench.compact_line(full), allow_nan=False, separators=(",", ":"))
        assert len(text) < 8192, (path, len(text))
        line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, (path, k)
        assert math.isfinite(line["value"]) and line["config"]["workload"] == "rgca_config2_envrelight"
        assert set(line["config"]) <= set(bench._CONFIG_KEYS) | {"grad_exchange", "launch"}       # workload keys only
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in line["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in line["cpu_baseline"], k
        for name, sec in line.get("secondary", {}).items():
            assert set(sec) <= {"value", "unit", "ms_per_step", "ms_per_step_median", "roofline", "loss_first_step",
                                "loss_last_step", "hot_path_ms_per_step", "error"}, (name, sorted(sec))


def test_emit_writes_the_detail_record_and_one_line(tmp_path, capfd):
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r05z_bench.json")))
    old = bench.DETAIL_PATH
    bench.DETAIL_PATH = str(tmp_path / "d" / "bench_detail.json")
    try:
        bench.emit(full)
    finally:
        bench.DETAIL_PATH = old
    out, err = capfd.readouterr()
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 8192
    assert json.loads(lines[0])["value"] == float(f"{full['value']:.5g}")
    assert json.load(open(tmp_path / "d" / "bench_detail.json"))["secondary"]["mvp"]["config"]["prims"] == 4096
    assert "bench.py detail:" in err


def test_scaling_model_reproduces_its_record(tmp_path):
    """tools/scaling_model.py on the committed one-GPU measurements (no GPU needed): the expected config-3 curve of DESIGN
    section 7 -- compute at 8 / 4 / 2 / 1 views per GPU + the xGMI exchange model -- comes out as recorded."""
    rec = os.path.join(ROOT, "profiles", "r06_scaling_model.json")
    out = str(tmp_path / "m.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_model.py"), "--from-json", rec, "--out", out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    want, got = json.load(open(rec)), json.load(open(out))
    assert [p["n_gpus"] for p in got["predicted"]] == [1, 2, 4, 8]
    for a, b in zip(want["predicted"], got["predicted"]):
        assert abs(a["direct_serial"]["views_per_s"] - b["direct_serial"]["views_per_s"]) < 1e-6 * a["direct_serial"]["views_per_s"]
    p8 = got["predicted"][3]
    # one view per GPU at N = 8: the exchange over 7 links (direct) is shorter than the compute, over one ring it is not
    assert p8["direct_serial"]["exchange_ms"] < p8["compute_ms"] < p8["ring_serial"]["exchange_ms"]
