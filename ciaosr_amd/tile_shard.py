"""Tile-sharded inference of one large image across the GPUs of a node.

Units = the LR tiles of `CiaoSR.clip_test` (ciaosr.py:233-254): every tile runs encoder + cs_attn +
head on its own crop with no cross-tile data, so the tile list is partitioned over ranks
(row-major tile index t -> rank t % R) with no collective on the data path.  The one exchange is
the delivery of output tiles to rank 0 (RCCL point-to-point over xGMI; torch.distributed backend
'nccl' on ROCm, 'gloo' in the CPU tests): a peer `isend`s each tile the moment its kernels are
queued -- the copy runs on RCCL's own stream over that peer's direct xGMI link to rank 0 while the
peer's next tile computes -- and rank 0 blends the tiles of round k-1 (tiles (k-1)R+1 .. kR-1, which
have had a whole tile time to arrive) after launching its own tile kR, always in the reference order
(h outer, w inner), so the result is bitwise equal to the 1-GPU run.  Nothing is all-gathered: a peer
holds only its own not-yet-delivered tiles, rank 0 a ring of two rounds of receive buffers
(2 tiles per peer), and exchange + blend hide under compute except for the last round.
"""
import torch
import torch.distributed as dist

from .restorer import tile_grid


def partition(n_tiles, world):
    """tile index -> owning rank, and per-rank tile lists (round-robin keeps neighbours apart so
    each rank's share of the expensive border tiles is even)."""
    return [[t for t in range(n_tiles) if t % world == r] for r in range(world)]


class _Mover:
    """isend / irecv of one tile, hiding the single-GPU rehearsal case (gloo backend with CUDA tensors: staged through
    the host).  Keeps the tensors alive until their transfer has completed."""

    def __init__(self, group, device):
        self.group = group
        self.via_host = device is not None and torch.device(device).type == 'cuda' and dist.get_backend(group) == 'gloo'
        self.pending = []          # (work, tensors kept alive)

    def send(self, t, dst):
        buf = t.cpu() if self.via_host else t.contiguous()
        self.pending.append((dist.isend(buf, dst=dst, group=self.group), buf))
        # drop what has been delivered: a peer never holds more than its in-flight tiles
        while self.pending and self.pending[0][0].is_completed():
            self.pending.pop(0)

    def recv(self, like_shape, src, device):
        buf = torch.empty(like_shape, dtype=torch.float32, device='cpu' if self.via_host else device)
        return dist.irecv(buf, src=src, group=self.group), buf

    def take(self, handle, device):
        work, buf = handle
        work.wait()                # NCCL: the current stream waits for the copy (no host block); gloo: host waits
        return buf.to(device) if self.via_host else buf

    def drain(self):
        for work, _ in self.pending:
            work.wait()
        self.pending = []


def sharded_clip_test(img_shape, tile, overlap, sf, tile_fn, blend_fn, finalize_fn, rank, world, group=None,
                      device=None, gather_to_all=False, mark=None):
    """Generic driver (device-agnostic so it can be exercised with gloo on CPU).

    mark(name): optional probe, called at 'last_own_tile' (this rank's last tile has been queued) and, on rank 0,
    at 'finalized' (every tile blended and normalised) -- bench.py records stream events there to report the exposed tail.

    tile_fn(hi, wi, tile)      -> [B, th*tw, 3] tensor (prediction of the LR crop), th = tw = tile*sf
    blend_fn(E, Wt, out, y0, x0, th, tw)  accumulates one tile (reference order)
    finalize_fn(E, Wt)         -> [B, H*W, 3]
    Returns the blended [B, h*sf*w*sf, 3] prediction on rank 0 (None elsewhere unless gather_to_all).
    """
    b, c, h, w = img_shape
    tile, origins = tile_grid(h, w, tile, overlap)
    th = tw = round(tile * sf)
    n = len(origins)
    shape = (b, th * tw, 3)
    mover = _Mover(group, device) if world > 1 else None

    if rank != 0:
        # peer: compute tile t = rank, rank + R, ... and hand each to RCCL as soon as it is queued
        for t in range(rank, n, world):
            hi, wi = origins[t]
            mover.send(tile_fn(hi, wi, tile), 0)
        if mark is not None:
            mark('last_own_tile')
        mover.drain()
        result = None
    else:
        E = torch.zeros(b, c, round(h * sf), round(w * sf), dtype=torch.float32, device=device)
        Wt = torch.zeros_like(E)

        def blend(t, out):
            hi, wi = origins[t]
            blend_fn(E, Wt, out, round(hi * sf), round(wi * sf), th, tw)

        prev = []                  # receive handles of the previous round's peer tiles, in tile order
        for t0 in range(0, n, world):
            # receives of THIS round are posted before the own tile is launched, consumed one round later
            cur = [(t, mover.recv(shape, t % world, device)) for t in range(t0 + 1, min(t0 + world, n))]
            hi, wi = origins[t0]
            own = tile_fn(hi, wi, tile)
            if mark is not None and t0 + world >= n:
                mark('last_own_tile')
            for t, handle in prev:                         # tiles (t0 - R + 1 .. t0 - 1) precede t0 in the reference order
                blend(t, mover.take(handle, device))
            blend(t0, own)
            prev = cur
        for t, handle in prev:
            blend(t, mover.take(handle, device))
        result = finalize_fn(E, Wt)
        if mark is not None:
            mark('finalized')
    if gather_to_all and world > 1:
        meta = [tuple(result.shape)] if rank == 0 else [None]
        dist.broadcast_object_list(meta, src=0, group=group)
        if rank != 0:
            result = torch.empty(meta[0], dtype=torch.float32, device=device)
        if mover.via_host:
            hostbuf = result.cpu()
            dist.broadcast(hostbuf, src=0, group=group)
            result = hostbuf.to(device)
        else:
            dist.broadcast(result, src=0, group=group)
    return result


def clip_test_distributed(restorer, x_norm, rank=None, world=None, group=None, gather_to_all=False, options=None, stats=None):
    """Tile-sharded counterpart of CiaoSR.clip_test on the GPUs of one node.
    `stats` (optional dict) receives a stream event per probe point of `sharded_clip_test` (bench.py's exposed-tail figure)."""
    from . import hip_ops
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    cfg = restorer.test_cfg
    sf = cfg.get('scale')
    opt = restorer.options(options)

    def tile_fn(hi, wi, tile):
        out, _ = restorer.run_tile(x_norm, hi, wi, tile, sf, opt)
        return out

    # Tile batching (restorer.clip_test's `test_cfg.tile_batch`): this rank's next B tiles go through the encoder in ONE call (shared
    # dense-layer launches, encoder_hip.forward_hwc_batch: every feature map bitwise the single-tile one); the head then runs tile by
    # tile in the driver's order, so sends, receives and the blend order are unchanged.
    gen = restorer.generator
    enc = getattr(gen, '_encoder_hip', None)
    n_batch = min(int(cfg.get('tile_batch', 8) or 1), 16)
    if (n_batch > 1 and x_norm.is_cuda and x_norm.shape[0] == 1 and enc is not None and hasattr(enc, 'forward_hwc_batch')
            and enc.supported() and getattr(gen, '_head', None) is not None):
        tile_sz, origins = tile_grid(x_norm.shape[-2], x_norm.shape[-1], cfg.get('tile'), cfg.get('tile_overlap'))
        mine = origins[rank::world]
        pos = {o: i for i, o in enumerate(mine)}
        cache = {}

        def tile_fn(hi, wi, tile):                            # noqa: F811 - the batched variant replaces the one above
            if (hi, wi) not in cache:
                grp = mine[pos[(hi, wi)]:pos[(hi, wi)] + n_batch]
                patches = torch.cat([x_norm[..., h0:h0 + tile, w0:w0 + tile] for (h0, w0) in grp], 0).contiguous().float()
                feats = enc.forward_hwc_batch(patches, opt)
                for j, o in enumerate(grp):
                    cache[o] = (patches[j], feats[j])
            patch, feat = cache.pop((hi, wi))
            th, tw = round(patch.shape[-2] * sf), round(patch.shape[-1] * sf)
            coord, cell = hip_ops.make_coord_cell(th, tw, patch.device)
            out = gen._head.forward(None, patch, coord, cell, gen.eval_bsize, feature_hwc=feat, options=opt)
            return out.unsqueeze(0)

    def blend_fn(E, Wt, out, y0, x0, th, tw):
        for bi in range(E.shape[0]):
            hip_ops.tile_blend(E[bi], Wt[bi], out[bi].contiguous(), y0, x0, th, tw)

    def finalize_fn(E, Wt):
        return torch.stack([hip_ops.tile_finalize(E[bi], Wt[bi]) for bi in range(E.shape[0])])

    mark = None
    if stats is not None and x_norm.is_cuda:
        def mark(name):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(x_norm.device))
            stats[name] = ev

    return sharded_clip_test(tuple(x_norm.shape), cfg.get('tile'), cfg.get('tile_overlap'), sf, tile_fn, blend_fn,
                             finalize_fn, rank, world, group, x_norm.device, gather_to_all, mark)


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
    slab = torch.zeros(b, max(q1 - q0, 1), 3, dtype=torch.float32, device=coord.device)
    if q1 > q0:
        slab[:, :q1 - q0] = predict_fn(feature, coord[:, q0:q1].contiguous(), cell[:, q0:q1].contiguous())
    if world == 1:
        return slab[:, :q1 - q0]
    # RGB slices -> rank 0, point to point (each peer over its own xGMI link); only rank 0 allocates the full output
    mover = _Mover(group, coord.device)
    if rank != 0:
        if q1 > q0:
            mover.send(slab[:, :q1 - q0].contiguous(), 0)
        mover.drain()
        out = None
    else:
        handles = [(r, mover.recv((b, s[1] - s[0], 3), r, coord.device)) for r, s in enumerate(slices) if r > 0 and s[1] > s[0]]
        out = torch.empty(b, n_query, 3, dtype=torch.float32, device=coord.device)
        out[:, q0:q1] = slab[:, :q1 - q0]
        for r, handle in handles:
            out[:, slices[r][0]:slices[r][1]] = mover.take(handle, coord.device)
    if gather_to_all:
        if rank != 0:
            out = torch.empty(b, n_query, 3, dtype=torch.float32, device=coord.device)
        if mover.via_host:
            hostbuf = out.cpu()
            dist.broadcast(hostbuf, src=0, group=group)
            out = hostbuf.to(coord.device)
        else:
            dist.broadcast(out, src=0, group=group)
    return out


def predict_query_sharded(restorer, x_norm, coord, cell, rank=None, world=None, group=None, gather_to_all=False, options=None):
    """Whole-image (single tile) counterpart of CiaoSR generator(x, coord, cell, test_mode=True) over the GPUs of a node."""
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    gen = restorer.generator
    opt = restorer.options(options)

    def feature_fn():
        return gen.gen_feature(x_norm, opt)[0]

    def predict_fn(feature, c, cl):
        return gen._predict([feature], c, cl, gen.eval_bsize, x_norm, opt)

    return query_sharded_predict(feature_fn, predict_fn, coord, cell, rank, world, gen.eval_bsize, group, gather_to_all)
