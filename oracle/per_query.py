"""Per-query numpy restatement of the CiaoSR head  --  TEST INFRASTRUCTURE ONLY.

Explicit index arithmetic of SURVEY Appendix A (what the HIP kernels implement), as opposed to
`ciaosr_oracle.py`, which mirrors the reference's tensor ops.  Follows:
  nearest gather  ciaosr_net.py:145-146,176-183  (F.grid_sample nearest, align_corners=False)
  shift / clamp   ciaosr_net.py:159-173
  rel / scale     ciaosr_net.py:185-193
  local attention ciaosr_net.py:211-216, decode :220-222, residual :107-108
Pinned against fixtures produced by the reference itself (tests/test_oracle_pin.py).
Only tests may import this file.
"""
import numpy as np

f32 = np.float32


def nearest_index(c, n):
    """idx = round_half_even((c + 1) * (n / 2) - 0.5), three separate fp32 roundings (A.2)."""
    c = np.asarray(c, dtype=f32)
    u = (c + f32(1.0)) * f32(n / 2.0) - f32(0.5)
    return np.rint(u.astype(f32)).astype(np.int64)


def shifted_coord(c, cell0, n, sign):
    """coord of the key sample along one axis: clamp(c + (sign * r + 1e-6)), r = (1 - cell0)/(n - 1)."""
    c = np.asarray(c, dtype=f32)
    if sign == 0:
        out = c.copy()
    else:
        t = f32(n - 1) / (f32(1.0) - f32(cell0))
        r = f32(1.0) / t                      # (2*|v|-1)/t with |v| = 1
        d = f32(f32(sign) * r) + f32(1e-6)
        out = (c + d).astype(f32)
    return np.minimum(np.maximum(out, f32(-1 + 1e-6)), f32(1 - 1e-6)).astype(f32)


def pixel_centre(k, n):
    """make_coord value of LR index k on an axis of n: fp32(-1 + 1/n) + fp32(2/n) * fp32(k)."""
    return (f32(-1.0 + 1.0 / n) + f32(2.0 / n) * np.asarray(k, dtype=f32)).astype(f32)


def unfold_at(feat, y, x):
    """U(y,x)[c*9 + ki*3 + kj] = F[c, y+ki-1, x+kj-1], zero outside (F.unfold(.,3,padding=1))."""
    C, H, W = feat.shape
    out = np.zeros((C, 3, 3), dtype=f32)
    for ki in range(3):
        for kj in range(3):
            yy, xx = y + ki - 1, x + kj - 1
            if 0 <= yy < H and 0 <= xx < W:
                out[:, ki, kj] = feat[:, yy, xx]
    return out.reshape(-1)


def mlp(x, params, prefix):
    ids = sorted({int(k.split('.')[2]) for k in params if k.startswith(prefix + '.layers.')})
    h = np.asarray(x, dtype=f32)
    for n, i in enumerate(ids):
        h = h @ np.asarray(params[f'{prefix}.layers.{i}.weight'], dtype=f32).T + \
            np.asarray(params[f'{prefix}.layers.{i}.bias'], dtype=f32)
        if n + 1 < len(ids):
            h = np.maximum(h, 0)
    return h.astype(f32)


def bilinear_border(img, cy, cx):
    """F.grid_sample bilinear, padding_mode='border', align_corners=False at (cy,cx) (A.4)."""
    C, H, W = img.shape

    def axis(c, n):
        u = (f32(c) + f32(1.0)) * f32(n / 2.0) - f32(0.5)
        u = min(max(f32(u), f32(0.0)), f32(n - 1))
        i0 = int(np.floor(u))
        f = f32(u) - f32(i0)
        return i0, min(i0 + 1, n - 1), f

    y0, y1, fy = axis(cy, H)
    x0, x1, fx = axis(cx, W)
    # ATen order: nw*(x1-x)(y1-y) + ne*(x-x0)(y1-y) + sw*(x1-x)(y-y0) + se*(x-x0)(y-y0)
    wy0, wx0 = f32(1.0) - fy, f32(1.0) - fx
    return (img[:, y0, x0] * (wx0 * wy0) + img[:, y0, x1] * (fx * wy0) +
            img[:, y1, x0] * (wx0 * fy) + img[:, y1, x1] * (fx * fy)).astype(f32)


def head_query(feat, nonlocal_map, x_lr, coord, cell, cell0, params, local_size=2, softmax_scale=1.0):
    """One query.  feat [C,H,W], nonlocal_map [Cn,H,W] or None, x_lr [3,H,W] normalised LR image,
    coord/cell (y,x) of the query, cell0 = cell of query 0 of the chunk (ciaosr_net.py:162-163)."""
    C, H, W = feat.shape
    cy, cx = f32(coord[0]), f32(coord[1])
    iy, ix = int(nearest_index(cy, H)), int(nearest_index(cx, W))
    q = unfold_at(feat, iy, ix)
    if local_size == 1:
        shifts = [(0, 0)]
    else:
        st = 4 - local_size
        shifts = [(a, b) for a in range(-1, 2, st) for b in range(-1, 2, st)]
    logits, pvs, idx = [], [], []
    for (sy, sx) in shifts:
        ky_c = shifted_coord(cy, cell0[0], H, np.sign(sy))
        kx_c = shifted_coord(cx, cell0[1], W, np.sign(sx))
        ky, kx = int(nearest_index(ky_c, H)), int(nearest_index(kx_c, W))
        idx.append((ky, kx))
        key = unfold_at(feat, ky, kx)
        val = key if nonlocal_map is None else np.concatenate([key, nonlocal_map[:, ky, kx]])
        rel = np.array([(cy - pixel_centre(ky, H)) * f32(H), (cx - pixel_centre(kx, W)) * f32(W)], dtype=f32)
        sc = np.array([f32(cell[0]) * f32(H), f32(cell[1]) * f32(W)], dtype=f32)
        wk = mlp(np.concatenate([key, rel, sc]), params, 'imnet_k')
        wv = mlp(np.concatenate([val, rel, sc]), params, 'imnet_v')
        logits.append(np.sum(q * (key * wk), dtype=f32))
        pvs.append(val * wv)
    lg = np.array(logits, dtype=f32) / f32(softmax_scale)
    e = np.exp(lg - lg.max())
    a = (e / e.sum()).astype(f32)
    z = sum(a[j] * pvs[j] for j in range(len(shifts))).astype(f32)
    rgb = mlp(z, params, 'imnet_q') + bilinear_border(x_lr, cy, cx)
    return rgb.astype(f32), dict(q_idx=(iy, ix), k_idx=idx, logits=lg, attn=a, z=z)
