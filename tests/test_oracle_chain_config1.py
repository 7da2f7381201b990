"""CPU: BASELINE config 1 -- "rgca_example.yml, 10k random-init Gaussians, 1 camera, 512x512, CPU PyTorch path (no GPU,
plumbing)": the whole per-view chain through the CPU oracles (oracle/chain.py: shading tail -> project -> bin/sort ->
colour+depth raster -> L1 -> all backward passes), seeds fixed; shapes, keys, value ranges and finite gradients
(SURVEY 8d config 1).  The GPU counterpart (HIP vs this chain at the same size) is
tests/test_gpu_fullsize.py::test_bench_step_matches_oracle_chain[config1]."""
import torch


def test_config1_oracle_chain_plumbing():
    import bench
    from oracle import chain

    torch.manual_seed(0)
    cfg = dict(bench.CFG, views_per_gpu=1, slab=100, gaussians=10_000, height=512, width=512, focal=1150.0)
    t = bench.make_inputs(cfg, "cpu")
    out = chain.cpu_view(t, 512, 512)
    assert out["rgb"].shape == (3, 512, 512) and out["alpha"].shape == (512, 512) and out["depth_norm"].shape == (512, 512)
    assert torch.isfinite(out["rgb"]).all() and float(out["alpha"].min()) >= 0.0 and float(out["alpha"].max()) <= 1.0
    assert float(out["alpha"].max()) > 0.9 and 0.05 < float((out["alpha"] > 0.5).float().mean()) < 0.9
    assert out["n_isect"] > 20_000 and int(out["last_id"].max()) < 10_000
    d = out["depth_norm"][out["alpha"] > 0.5]
    assert 500.0 < float(d.min()) and float(d.max()) < 900.0       # the head sits 700 mm from the camera
    for k in ("f_vn", "f_vc", "postex", "tn", "albedo"):
        g = t[k].grad
        assert g is not None and g.shape == t[k].shape and torch.isfinite(g).all() and float(g.abs().max()) > 0.0, k
    assert set(out["stage_grads"]) == {"color", "opacity", "primpos", "primscale", "primqvec"}
