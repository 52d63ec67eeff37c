"""Tile planner for NON-integer (or > 4) scales -- SURVEY section 8(f)4, an opt-in extension.

The reference tiles only integer scales <= 4 (`clip_test` slices the HR canvas with `hi*sf`, ciaosr.py:218-258, and the
configs set `tile=None` above x4), so an arbitrary-scale request on a 2K input falls back to the whole-image path whose
non-local attention is quadratic in the LR area.  This planner keeps the reference's LR tiling (same tile origins,
same uniform-weight blending of overlaps) and generalises the HR side:

* an HR pixel belongs to a tile when its centre, mapped to LR pixel units, lies inside the tile:
  (i + 0.5) * h / Ht in [y0, y0 + th)   -- for an integer scale this is exactly the reference's rectangle
  [y0*sf, (y0+th)*sf);
* the tile-local query coordinate is the affine image of the global grid-centre coordinate,
  c_local = ((c_global + 1) * h / 2 - y0) * 2 / th - 1, and the tile-local cell is (2 / Ht) * (h / th): for an integer
  scale these equal make_coord / the cell of the tile-local HR grid (ciaosr.py:240-243) up to fp32 rounding; for a
  tile that covers the whole axis they ARE the global values (bitwise), so a one-tile plan reproduces the
  whole-image path exactly.

Results differ from the reference's whole-image path because features and non-local attention become tile-local --
that is the point, and the reason this is opt-in (`test_cfg.tile_any_scale = True`).
"""
import numpy as np
import torch

from .coords import make_coord


def tile_starts(n, tile, overlap):
    stride = tile - overlap
    return list(range(0, n - tile, stride)) + [n - tile]


def _ceil_div(a, b):
    return -((-a) // b)


def hr_span(y0, th, n_lr, n_hr):
    """HR index range [i0, i1) whose pixel centres (i + 0.5) * n_lr / n_hr fall into [y0, y0 + th); exact integers."""
    i0 = max(0, _ceil_div(2 * y0 * n_hr - n_lr, 2 * n_lr))
    i1 = min(n_hr, _ceil_div(2 * (y0 + th) * n_hr - n_lr, 2 * n_lr))
    return i0, i1


def axis_local(y0, th, n_lr, n_hr):
    """(i0, i1, local coords fp32 [i1-i0], local cell fp32 scalar) of one axis of one tile."""
    i0, i1 = hr_span(y0, th, n_lr, n_hr)
    g = make_coord((n_hr,), flatten=True)[:, 0]                     # the global fp32 sequence (mmedit make_coord)
    if y0 == 0 and th == n_lr:
        return i0, i1, g[i0:i1].clone(), np.float32(2.0 / n_hr)
    c = ((g[i0:i1].double() + 1.0) * (n_lr / 2.0) - y0) * (2.0 / th) - 1.0
    return i0, i1, c.float(), np.float32((2.0 / n_hr) * (n_lr / th))


def plan(h, w, ht, wt, tile, overlap):
    """Row-major (h outer, w inner: the reference's blend order) list of
    dict(y0, x0, th, tw, i0, i1, j0, j1, coord [Q,2] fp32 (y,x), cell [Q,2] fp32) for an LR image h x w and an HR
    target ht x wt."""
    tile = min(tile, h, w)
    overlap = min(overlap, tile - 1)
    out = []
    ys = [axis_local(y0, tile, h, ht) for y0 in tile_starts(h, tile, overlap)]
    xs = [axis_local(x0, tile, w, wt) for x0 in tile_starts(w, tile, overlap)]
    for y0, (i0, i1, cy, celly) in zip(tile_starts(h, tile, overlap), ys):
        for x0, (j0, j1, cx, cellx) in zip(tile_starts(w, tile, overlap), xs):
            grid = torch.stack(torch.meshgrid(cy, cx, indexing='ij'), dim=-1).view(-1, 2)
            cell = torch.empty_like(grid)
            cell[:, 0] = float(celly)
            cell[:, 1] = float(cellx)
            out.append(dict(y0=y0, x0=x0, th=tile, tw=tile, i0=i0, i1=i1, j0=j0, j1=j1, coord=grid, cell=cell))
    return out
