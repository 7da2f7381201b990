"""oracle/mesh_ref.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).

Plain numpy z-buffer rasterizer with the conventions of goliath_amd/csrc/meshraster.hip (sample at pixel centres, all
edge functions >= 0, nearest positive depth, ties to the lower face index, perspective-correct depth / barycentrics).
It checks the HIP kernel's bookkeeping (tile binning, compaction order, z-test); it is NOT a restatement of the
reference: the index / depth / barycentric images come from the third-party drtk in the reference
(/root/reference/ca_code/utils/render_drtk.py:44-46), whose source is absent -- PARITY UNPINNED for this stand-in.
"""
import numpy as np


def rasterize(v_pix, vi, H, W):
    """v_pix[B,V,3] float, vi[F,3] int -> index[B,H,W] int32, depth[B,H,W], bary[B,3,H,W] (float64 arithmetic)."""
    v_pix = np.asarray(v_pix, dtype=np.float64)
    vi = np.asarray(vi, dtype=np.int64)
    B = v_pix.shape[0]
    index = -np.ones((B, H, W), np.int32)
    best_iz = np.zeros((B, H, W))
    bary = np.zeros((B, 3, H, W))
    for b in range(B):
        for f, (i0, i1, i2) in enumerate(vi):
            (ax, ay, az), (bx, by, bz), (cx, cy, cz) = v_pix[b, i0], v_pix[b, i1], v_pix[b, i2]
            area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
            if not (az > 0 and bz > 0 and cz > 0) or area == 0:
                continue
            j0, j1 = max(0, int(np.ceil(min(ax, bx, cx) - 0.5))), min(W - 1, int(np.floor(max(ax, bx, cx) - 0.5)))
            k0, k1 = max(0, int(np.ceil(min(ay, by, cy) - 0.5))), min(H - 1, int(np.floor(max(ay, by, cy) - 0.5)))
            if j0 > j1 or k0 > k1:
                continue
            px, py = np.meshgrid(np.arange(j0, j1 + 1) + 0.5, np.arange(k0, k1 + 1) + 0.5)
            b0 = ((by - cy) * px + (cx - bx) * py + (bx * cy - cx * by)) / area
            b1 = ((cy - ay) * px + (ax - cx) * py + (cx * ay - ax * cy)) / area
            b2 = ((ay - by) * px + (bx - ax) * py + (ax * by - bx * ay)) / area
            w0, w1, w2 = b0 / az, b1 / bz, b2 / cz
            iz = w0 + w1 + w2
            sl = (b, slice(k0, k1 + 1), slice(j0, j1 + 1))
            win = (b0 >= 0) & (b1 >= 0) & (b2 >= 0) & (iz > best_iz[sl])
            best_iz[sl] = np.where(win, iz, best_iz[sl])
            index[sl] = np.where(win, f, index[sl])
            for c, w in enumerate((w0, w1, w2)):
                s = (b, c, slice(k0, k1 + 1), slice(j0, j1 + 1))
                bary[s] = np.where(win, w / np.where(iz != 0, iz, 1.0), bary[s])
    depth = np.where(index >= 0, 1.0 / np.where(best_iz != 0, best_iz, 1.0), 0.0)
    return index, depth, bary
