"""Tile-sharded inference of one large image across the GPUs of a node.

Units = the LR tiles of `CiaoSR.clip_test` (ciaosr.py:233-254): every tile runs encoder + cs_attn +
head on its own crop with no cross-tile data, so the tile list is partitioned over ranks
(row-major tile index t -> rank t % R) with no collective on the data path.  The single exchange
step is the gather of output tiles to rank 0 over RCCL (torch.distributed backend 'nccl' on ROCm;
'gloo' in the CPU tests), after which rank 0 blends in the reference order (h outer, w inner) so the
result is bitwise equal to the 1-GPU run.  xGMI is point-to-point: every peer sends its own tiles
straight to rank 0 over its direct link (all_gather of equal-sized, padded slabs), no ring needed.
"""
import torch
import torch.distributed as dist

from .restorer import tile_grid


def partition(n_tiles, world):
    """tile index -> owning rank, and per-rank tile lists (round-robin keeps neighbours apart so
    each rank's share of the expensive border tiles is even)."""
    return [[t for t in range(n_tiles) if t % world == r] for r in range(world)]


def sharded_clip_test(img_shape, tile, overlap, sf, tile_fn, blend_fn, finalize_fn, rank, world, group=None,
                      device=None, gather_to_all=False):
    """Generic driver (device-agnostic so it can be exercised with gloo on CPU).

    tile_fn(hi, wi, tile)      -> [B, th*tw, 3] tensor (prediction of the LR crop), th = tw = tile*sf
    blend_fn(E, Wt, out, y0, x0, th, tw)  accumulates one tile (reference order)
    finalize_fn(E, Wt)         -> [B, H*W, 3]
    Returns the blended [B, h*sf*w*sf, 3] prediction on rank 0 (None elsewhere unless gather_to_all).
    """
    b, c, h, w = img_shape
    tile, origins = tile_grid(h, w, tile, overlap)
    th = tw = round(tile * sf)
    mine = partition(len(origins), world)[rank]
    n_max = (len(origins) + world - 1) // world
    slab = torch.zeros(n_max, b, th * tw, 3, dtype=torch.float32, device=device)
    for slot, t in enumerate(mine):
        hi, wi = origins[t]
        slab[slot] = tile_fn(hi, wi, tile)
    if world > 1:
        if slab.is_cuda and dist.get_backend(group) == 'gloo':
            # debugging / single-GPU rehearsal of the N-rank path: gloo gathers host copies
            host = slab.cpu()
            hparts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(hparts, host, group=group)
            parts = [h.to(slab.device) for h in hparts]
        else:
            parts = [torch.empty_like(slab) for _ in range(world)]
            dist.all_gather(parts, slab, group=group)      # the one exchange step (RCCL over xGMI)
    else:
        parts = [slab]
    if rank != 0 and not gather_to_all:
        return None
    E = torch.zeros(b, c, round(h * sf), round(w * sf), dtype=torch.float32, device=device)
    Wt = torch.zeros_like(E)
    for t, (hi, wi) in enumerate(origins):                 # reference blend order (ciaosr.py:233-234)
        out = parts[t % world][t // world]
        blend_fn(E, Wt, out, round(hi * sf), round(wi * sf), th, tw)
    return finalize_fn(E, Wt)


def clip_test_distributed(restorer, x_norm, rank=None, world=None, group=None, gather_to_all=False):
    """Tile-sharded counterpart of CiaoSR.clip_test on the GPUs of one node."""
    from . import hip_ops
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    cfg = restorer.test_cfg
    sf = cfg.get('scale')

    def tile_fn(hi, wi, tile):
        out, _ = restorer.run_tile(x_norm, hi, wi, tile, sf)
        return out

    def blend_fn(E, Wt, out, y0, x0, th, tw):
        for bi in range(E.shape[0]):
            hip_ops.tile_blend(E[bi], Wt[bi], out[bi].contiguous(), y0, x0, th, tw)

    def finalize_fn(E, Wt):
        return torch.stack([hip_ops.tile_finalize(E[bi], Wt[bi]) for bi in range(E.shape[0])])

    return sharded_clip_test(tuple(x_norm.shape), cfg.get('tile'), cfg.get('tile_overlap'), sf, tile_fn, blend_fn,
                             finalize_fn, rank, world, group, x_norm.device, gather_to_all)


# ---------------------------------------------------------------------------------------------------------------
# Single-tile configurations (C1, C2, C5): shard the QUERY RANGE of the one tile (second half; the north
# star's "RCCL broadcast of encoder features / gather of output tiles").  Rank 0 runs the encoder once and broadcasts the
# feature map (1.2 MB at 48x48x64, 18.9 MB at 192x192) -- point-to-point over each peer's own xGMI link -- every rank
# evaluates the head on a contiguous slice of the queries (cs_attn and the per-LR-pixel tables are recomputed per rank:
# they are LR-sized, the head is HR-sized), and the slices are gathered on rank 0.  Slice boundaries are multiples of
# `eval_bsize` whenever the cell is not uniform, because the reference's shift radius uses the first cell of each
# eval_bsize chunk (ciaosr_net.py:162-165, :238-246); with a uniform cell (every forward_test call) any split is exact.
# ---------------------------------------------------------------------------------------------------------------
def query_slices(n_query, world, chunk=None, uniform_cell=True):
    """Contiguous [q0, q1) per rank; chunk-aligned when the cell varies."""
    unit = 1 if (uniform_cell or not chunk) else int(chunk)
    n_units = (n_query + unit - 1) // unit
    per, extra = divmod(n_units, world)
    out, u0 = [], 0
    for r in range(world):
        u1 = u0 + per + (1 if r < extra else 0)
        out.append((min(u0 * unit, n_query), min(u1 * unit, n_query)))
        u0 = u1
    return out


def query_sharded_predict(feature_fn, predict_fn, coord, cell, rank, world, chunk=None, group=None, gather_to_all=False):
    """Generic driver (device-agnostic: exercised with gloo on CPU).

    feature_fn()                          -> encoder feature map (called on rank 0 only)
    predict_fn(feature, coord, cell)      -> [B, q, 3] for a slice of queries
    coord, cell [B, Q, 2].  Returns [B, Q, 3] on rank 0 (None elsewhere unless gather_to_all).
    """
    b, n_query = coord.shape[0], coord.shape[1]
    uniform = bool((cell == cell[:, :1]).all())
    slices = query_slices(n_query, world, chunk, uniform)
    feature = feature_fn() if rank == 0 else None
    if world > 1:
        meta = [tuple(feature.shape)] if rank == 0 else [None]
        dist.broadcast_object_list(meta, src=0, group=group)
        if rank != 0:
            feature = torch.empty(meta[0], dtype=torch.float32, device=coord.device)
        via_host = feature.is_cuda and dist.get_backend(group) == 'gloo'      # single-GPU rehearsal of the N-rank path
        buf = feature.cpu() if via_host else feature.contiguous()
        dist.broadcast(buf, src=0, group=group)                                # encoder features -> every rank
        feature = buf.to(coord.device) if via_host else buf
    q0, q1 = slices[rank]
    n_max = max(s[1] - s[0] for s in slices)
    slab = torch.zeros(b, n_max, 3, dtype=torch.float32, device=coord.device)
    if q1 > q0:
        slab[:, :q1 - q0] = predict_fn(feature, coord[:, q0:q1].contiguous(), cell[:, q0:q1].contiguous())
    if world > 1:
        if slab.is_cuda and dist.get_backend(group) == 'gloo':
            host = slab.cpu()
            hparts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(hparts, host, group=group)
            parts = [h.to(slab.device) for h in hparts]
        else:
            parts = [torch.empty_like(slab) for _ in range(world)]
            dist.all_gather(parts, slab, group=group)
    else:
        parts = [slab]
    if rank != 0 and not gather_to_all:
        return None
    return torch.cat([parts[r][:, :s[1] - s[0]] for r, s in enumerate(slices)], dim=1)


def predict_query_sharded(restorer, x_norm, coord, cell, rank=None, world=None, group=None, gather_to_all=False):
    """Whole-image (single tile) counterpart of CiaoSR generator(x, coord, cell, test_mode=True) over the GPUs of a node."""
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    gen = restorer.generator

    def feature_fn():
        return gen.gen_feature(x_norm)[0]

    def predict_fn(feature, c, cl):
        return gen._predict([feature], c, cl, gen.eval_bsize, x_norm)

    return query_sharded_predict(feature_fn, predict_fn, coord, cell, rank, world, gen.eval_bsize, group, gather_to_all)
