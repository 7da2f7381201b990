#!/usr/bin/env python
"""Time the tail-conv kernels (gol_tail_conv_fwd / _bwd pieces) one by one at the native size.
Usage: python tools/tail_probe.py [B] [h]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from goliath_amd import _lib
from goliath_amd._lib import c_int, fptr, stream_ptr


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    w = h
    dev = "cuda"
    for CH, E, nd, wB in ((15, 3, 113, B), (18, 6, 113, B), (4, 0, 0, 1)):
        x = torch.randn(B, 16, h, w, device=dev)
        weff = torch.randn(wB, 16, CH, 4, 4, device=dev)
        wt = weff.permute(0, 2, 3, 4, 1).contiguous()
        lc = torch.randn(nd, B, E, device=dev) if E else None
        bias = torch.randn(nd + CH - E, 2 * h, 2 * w, device=dev)
        out = torch.empty(B, CH, 2 * h, 2 * w, device=dev)
        g = torch.randn_like(out)
        gx, gw, gb = torch.empty_like(x), torch.zeros_like(weff), torch.empty_like(bias)
        scratch = torch.empty(8 * 2 * 768 * 4096 // 8, device=dev)
        dims = (c_int(B), c_int(16), c_int(h), c_int(w), c_int(CH), c_int(E), c_int(nd), c_int(wB))
        fwd = lambda: _lib.call("gol_tail_conv_fwd", *dims, fptr(x), fptr(weff), fptr(lc), fptr(bias), fptr(out), stream_ptr())
        bwd = lambda a, b2, c: (lambda: _lib.call("gol_tail_conv_bwd", *dims, fptr(x), fptr(wt), fptr(lc), fptr(g), fptr(a),
                                                  fptr(b2), fptr(c), fptr(scratch if b2 is not None else None), stream_ptr()))
        N = 4 * h * w
        print(f"CH={CH} E={E} B={B} N={N}: fwd {timed(fwd):.3f} ms | bwd_x {timed(bwd(gx, None, None)):.3f} | "
              f"bwd_w {timed(bwd(None, gw, None)):.3f} | bwd_bias {timed(bwd(None, None, gb)):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
