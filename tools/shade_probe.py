import sys, torch, time
sys.path.insert(0, "/root/repo")
import bench
from goliath_amd import shade, _lib
cfg = bench.CFG
t = bench.make_inputs(cfg, torch.device("cuda"))
B = cfg["views_per_gpu"]
li = torch.ones(B, 1, 3).cuda(); lp = torch.tensor([[[0., 0., 1100.]]] * B).cuda(); nl = torch.ones(B, dtype=torch.int32).cuda()
def run(mode):
    kw = dict(preconv_envmap=t["mips"], lightrot=t["lightrot"]) if mode == "env" else dict(light_intensity=li, headrel_light_pos=lp, n_lights=nl)
    with torch.no_grad():
        for _ in range(3):
            shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"], **kw)
        _lib.TIMING = []
        for _ in range(10):
            shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"], **kw)
        torch.cuda.synchronize()
        tm = [e0.elapsed_time(e1) for n, e0, e1 in _lib.TIMING if n == "gol_shade_fwd"]; _lib.TIMING = None
    ms = sum(tm) / len(tm)
    print(mode, "shade_fwd ms", round(ms, 4), "TB/s", round(B * 250000 * 700 / ms / 1e9, 2))
run("env"); run("sg1")
# pure copy ceiling for reference
x = torch.empty(1400 * 1000 * 1000 // 8, device="cuda"); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
print("copy TB/s (r+w)", round(2 * x.numel() * 4 * 10 / e0.elapsed_time(e1) / 1e9, 2))
