"""Tile-sharded inference of one large image across the GPUs of a node.

Units = the LR tiles of `CiaoSR.clip_test` (ciaosr.py:233-254): every tile runs encoder + cs_attn +
head on its own crop with no cross-tile data, so the tile list is partitioned over ranks (`tile_owners`:
row-major tile index t -> rank t % R, except that rank 0 -- which also owns every blend and the finalize --
sits out the ragged last round) with no collective on the data path.  The one exchange is the delivery of
output tiles to rank 0 (RCCL point-to-point over xGMI; torch.distributed backend 'nccl' on ROCm, 'gloo' in
the CPU tests and the one-GPU rehearsals):

* a peer hands each tile to RCCL the moment its kernels are queued -- the send runs on RCCL's own stream
  over that peer's direct xGMI link to rank 0 while the peer's next tile computes;
* rank 0 posts the receives of a round as ONE grouped operation (one RCCL kernel for the R - 1 peers) AFTER
  its own tile of that round has been queued: RCCL orders the receive kernel behind the work already on the
  compute stream, so it starts when rank 0's tile ends -- which is when the peers' tiles of the same round
  end -- instead of spinning on rank 0's CUs for a whole tile time; it then runs under rank 0's NEXT tile;
* rank 0 blends a round's tiles one round later (they have had a whole tile time to arrive), always in the
  reference order (h outer, w inner), so the result is bitwise equal to the 1-GPU run.

Nothing is all-gathered: a peer holds only its own not-yet-delivered tiles; rank 0 holds the receive buffers
posted since its previous own tile -- two rounds (2 (R - 1) tiles, 99 MB at R = 8) while it joins every round,
and up to the peer tiles between two of its own tiles when `rank0_share` < 1 (share 0: every peer tile of the
image, n_tiles x 7 MB, posted at the start) -- and exchange + blend hide under compute except for the last
round.  Both sides use the grouped form (`dist.batch_isend_irecv`), i.e. the process group's ONE communicator
(created eagerly by `init_process_group(device_id=...)`): no per-pair communicator is built lazily inside the
timed region.

`loopback=True` (world size 1, an initialised 1-rank group) runs the SAME exchange code on one GPU: every second
tile is treated as a peer's, computed here, and delivered to this rank by a grouped isend + irecv to itself -- the
RCCL branches (`_Mover` with device tensors, the stream-ordered `wait`) execute without a second GPU
(tests/test_rccl_single_gpu.py).
"""
import math
import os
import sys
import threading
import time
import weakref

import torch
import torch.distributed as dist

from .restorer import tile_grid


def rccl_env_defaults(env=None):
    """Environment defaults for the tile exchange, applied with setdefault.  NCCL_*: read by RCCL when the communicator is
    created -- one point-to-point channel: a 7-MB tile per peer and round needs ~0.3 ms of one channel, and every further channel
    is one more workgroup of the receive kernel resident on rank 0's CUs beside its own tile.  HSA_ENABLE_IPC_MODE_LEGACY=0
    (dmabuf IPC, this driver's only mode): read by ROCr when HSA initialises, i.e. at the FIRST HIP call of the process -- so this
    function must run before anything touches the GPU (`torch.cuda.set_device`, a tensor on the device; counting devices does
    not): bench.py, tools/test.py and the test children call it first thing in main().  Returns the p2p channel count in force
    (reported by bench.py)."""
    env = os.environ if env is None else env
    env.setdefault('NCCL_MIN_P2P_NCHANNELS', '1')
    env.setdefault('NCCL_MAX_P2P_NCHANNELS', '1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return int(env['NCCL_MAX_P2P_NCHANNELS'])


class StepDeadline:
    """Bounded wait for the N-rank paths: `beat()` at the start of every step; a rank whose step does not finish within `seconds`
    writes one line to stderr and leaves with exit code 3 (`os._exit`: no exec, no unwinding through a hung collective), so the
    launcher (torch.distributed.run) tears the other ranks down instead of the job hanging on a lost peer."""

    def __init__(self, seconds, what='step', rank=0):
        self.seconds, self.what, self.rank = float(seconds), what, rank
        self.t = time.monotonic()
        self._stop = threading.Event()
        self._thread = None
        if self.seconds > 0:
            self._thread = threading.Thread(target=self._run, name='ciaosr-step-deadline', daemon=True)
            self._thread.start()

    def beat(self):
        self.t = time.monotonic()

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(min(1.0, self.seconds / 4)):
            late = time.monotonic() - self.t
            if late > self.seconds:
                sys.stderr.write(f'[ciaosr] rank {self.rank}: {self.what} not finished after {late:.0f} s '
                                 f'(deadline {self.seconds:.0f} s): leaving with exit code 3\n')
                sys.stderr.flush()
                os._exit(3)


def tile_owners(n_tiles, world, rank0_share=1.0):
    """owner[t] of every tile.  Rounds of `world` consecutive tiles, rank r taking the r-th tile of its round (t % R when nothing is
    skipped: neighbours stay apart, every rank's share of the border tiles is even).  Rank 0 also blends all n tiles and
    finalizes the image, so it gets the lighter share wherever one exists: it sits out the ragged last round (117 tiles on 8
    ranks: 14 / 15 x 5 / 14 x 2 instead of 15 x 5 / 14 x 3 with rank 0 among the fifteens), and with `rank0_share` s < 1 it takes
    part in a fraction s of the full rounds only (evenly spaced; s = 0: rank 0 only blends).  The blend order never changes."""
    if world <= 1:
        return [0] * n_tiles
    s = min(max(float(rank0_share), 0.0), 1.0)
    owners, k = [], 0
    while len(owners) < n_tiles:
        left = n_tiles - len(owners)
        takes0 = math.ceil((k + 1) * s - 1e-9) > math.ceil(k * s - 1e-9)
        if left < world and k > 0:
            takes0 = False
        ranks = ([0] if takes0 else []) + list(range(1, world))
        owners += ranks[:left]
        k += 1
    return owners


def partition(n_tiles, world, rank0_share=1.0):
    """Per-rank tile lists of `tile_owners`."""
    owners = tile_owners(n_tiles, world, rank0_share)
    return [[t for t in range(n_tiles) if owners[t] == r] for r in range(world)]


_warm_groups = {}          # id(group) -> (weak reference to the group object, device type the communicator was counted on)


def ensure_communicator(group=None, device=None):
    """The group's communicator must exist before the first grouped point-to-point call that only SOME ranks take part in
    (torch.distributed.batch_isend_irecv: 'if this is the first collective call in the group, all ranks must participate'): one
    all-reduce of ones, once per group OBJECT, by every rank.  Returns the rank count as the communicator counts it.
    The cache holds weak references and is re-validated against the live object: after destroy_process_group() + a second
    init_process_group() in the same interpreter a new group may reuse the id of a dead one, and skipping the all-reduce for it
    would leave the first partial grouped call to create the communicator lazily."""
    if not dist.is_initialized():
        _warm_groups.clear()
        return 1
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    on_gpu = device is not None and torch.device(device).type == 'cuda' and dist.get_backend(group) != 'gloo'
    kind = 'cuda' if on_gpu else 'cpu'
    hit = _warm_groups.get(id(pg))
    if hit is not None and hit[0]() is pg and hit[1] == kind:
        return dist.get_world_size(group)
    for k in [k for k, v in _warm_groups.items() if v[0]() is None]:
        del _warm_groups[k]
    ones = torch.ones(1, dtype=torch.int32, device=device if on_gpu else 'cpu')
    dist.all_reduce(ones, group=group)
    _warm_groups[id(pg)] = (weakref.ref(pg), kind)
    return int(ones.item())


class _Mover:
    """Grouped isend / irecv of tiles, hiding the single-GPU rehearsal case (gloo backend with CUDA tensors: staged through
    the host).  Keeps the tensors alive until their transfer has completed."""

    def __init__(self, group, device):
        self.group = group
        self.via_host = device is not None and torch.device(device).type == 'cuda' and dist.get_backend(group) == 'gloo'
        self.pending = []          # (works, tensors kept alive)

    def _send_ops(self, t, dst):
        buf = t.cpu() if self.via_host else t.contiguous()
        return [dist.P2POp(dist.isend, buf, dst, self.group)], buf

    def _recv_ops(self, like_shape, srcs, device):
        bufs = [torch.empty(like_shape, dtype=torch.float32, device='cpu' if self.via_host else device) for _ in srcs]
        return [dist.P2POp(dist.irecv, b, s, self.group) for b, s in zip(bufs, srcs)], bufs

    def _reap(self):
        # drop what has been delivered: a peer never holds more than its in-flight tiles
        while self.pending and all(w.is_completed() for w in self.pending[0][0]):
            self.pending.pop(0)

    def send(self, t, dst):
        ops, buf = self._send_ops(t, dst)
        works = dist.batch_isend_irecv(ops)
        self.pending.append((works, buf))
        self._reap()

    def recv_many(self, like_shape, srcs, device):
        """One grouped receive of len(srcs) equally shaped tiles (RCCL: ONE kernel on the communicator's stream, ordered behind
        what the compute stream holds at this moment).  Returns a batch handle for `take`."""
        ops, bufs = self._recv_ops(like_shape, srcs, device)
        works = dist.batch_isend_irecv(ops)
        return dict(works=works, bufs=bufs, waited=False)

    def loop_many(self, tiles, me, device):
        """Loopback (one rank acting as its own peers): the sends of `tiles` to this rank and their receives as ONE grouped call
        -- a send to oneself only completes inside the group that also holds its receive.  Same handle as `recv_many`.
        gloo has no pair to itself: there (CPU tests of the driver logic only) the hand-over is a host copy."""
        if dist.get_backend(self.group) == 'gloo':
            return dict(works=[], bufs=[t.detach().cpu().clone() if self.via_host else t.detach().clone() for t in tiles], waited=False)
        sends, kept = [], []
        for t in tiles:
            ops, buf = self._send_ops(t, me)
            sends += ops
            kept.append(buf)
        recvs, bufs = self._recv_ops(tuple(tiles[0].shape), [me] * len(tiles), device)
        works = dist.batch_isend_irecv(sends + recvs)
        self.pending.append((works, kept))
        self._reap()
        return dict(works=works, bufs=bufs, waited=False)

    def take(self, batch, i, device):
        if not batch['waited']:
            for w in batch['works']:
                w.wait()           # NCCL: the current stream waits for the grouped copy (no host block); gloo: host waits
            batch['waited'] = True
        buf, batch['bufs'][i] = batch['bufs'][i], None
        return buf.to(device) if self.via_host else buf

    def drain(self):
        for works, _ in self.pending:
            for w in works:
                w.wait()
        self.pending = []


def rccl_self_probe(device, tile_shape=(1, 768 * 768, 3), busy_ms=40.0):
    """What one GPU can show about the exchange: inside an initialised 1-rank 'nccl' (= RCCL) group, count the communicator's
    ranks with an all-reduce on the device, then run the hand-off `sharded_clip_test` relies on -- a tile-sized tensor produced by
    work still QUEUED on the compute stream goes through ONE grouped isend + irecv to this rank, `wait()` orders the current
    stream behind the copy without blocking the host, and a consumer kernel reads the receive buffer -- with no host
    synchronisation anywhere between producer and consumer.  Returns a dict (bitwise flags, host / GPU times); the caller asserts.
    Used by tests/rccl_child.py and by `bench.py`'s N = 1 probe child."""
    dev = torch.device(device)
    ranks = ensure_communicator(None, dev)
    n = 1
    for d in tile_shape:
        n *= int(d)
    g = torch.Generator(device='cpu').manual_seed(17)
    a = (torch.randn(2048, 2048, generator=g) / 45.0).to(dev)
    x0 = torch.randn(2048, 2048, generator=g).to(dev)
    reps = -(-n // x0.numel())

    def producer(rounds):
        x = x0
        for _ in range(rounds):
            x = torch.tanh(a @ x)                      # a chain of dependent kernels: the tile below exists only when it has run
        return x.flatten().repeat(reps)[:n].view(tile_shape) + 0.5

    producer(2)
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    producer(8)
    ev[1].record()
    torch.cuda.synchronize(dev)
    rounds = max(8, int(8 * busy_ms / max(ev[0].elapsed_time(ev[1]), 1e-3)))
    mover = _Mover(None, dev)
    out = {}
    for label in ('cold', 'warm'):                     # 'cold': the first grouped p2p of the communicator (connection set-up)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_host = time.perf_counter()
        ev[0].record()
        tile = producer(rounds)
        ev[1].record()
        batch = mover.loop_many([tile], dist.get_rank(), dev)
        got = mover.take(batch, 0, dev)               # wait(): stream-ordered on RCCL, no host block
        consumed = got * 2.0 + 1.0                    # consumer kernel on the current stream
        ev[2].record()
        t_host = (time.perf_counter() - t_host) * 1e3
        torch.cuda.synchronize(dev)
        mover.drain()
        out[label] = dict(host_enqueue_ms=round(t_host, 3), producer_gpu_ms=round(ev[0].elapsed_time(ev[1]), 3),
                          exchange_and_consumer_gpu_ms=round(ev[1].elapsed_time(ev[2]), 3),
                          tile_bitwise=bool(torch.equal(got, tile)), consumer_bitwise=bool(torch.equal(consumed, tile * 2.0 + 1.0)),
                          tile_nonconstant=bool(float(tile.std()) > 1e-3))
    b = torch.arange(1024, dtype=torch.float32, device=dev)
    dist.broadcast(b, src=0)
    return dict(ranks=ranks, backend=dist.get_backend(), tile_mb=round(n * 4 / 1e6, 2), p2p_channels=os.environ.get('NCCL_MAX_P2P_NCHANNELS'),
                ipc_mode_legacy=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), broadcast_ok=bool(b[-1].item() == 1023.0), **out)


def sharded_clip_test(img_shape, tile, overlap, sf, tile_fn, blend_fn, finalize_fn, rank, world, group=None,
                      device=None, gather_to_all=False, mark=None, rank0_share=1.0, loopback=False):
    """Generic driver (device-agnostic so it can be exercised with gloo on CPU).

    loopback (world == 1 inside an initialised 1-rank group): the schedule of a 2-rank run on ONE rank -- the tiles `tile_owners`
    gives to rank 1 are computed here and handed to this rank's own receive buffers by a grouped isend + irecv to itself
    (`_Mover.loop_many`), posted, waited for and blended exactly where a real peer's tiles are.  Same image, bitwise.

    mark(name): optional probe, called at 'start', at 'last_own_tile' (this rank's last tile has been queued) and, on rank 0,
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
    loopback = bool(loopback) and world == 1 and dist.is_initialized()
    owners = tile_owners(n, 2 if loopback else world, rank0_share)
    if world > 1 or loopback:
        ensure_communicator(group, device)
    mover = _Mover(group, device) if (world > 1 or loopback) else None
    if mark is not None:
        mark('start')

    if rank != 0:
        # peer: compute the own tiles in tile order and hand each to RCCL as soon as it is queued
        for t in range(n):
            if owners[t] == rank:
                hi, wi = origins[t]
                mover.send(tile_fn(hi, wi, tile), 0)
        if mark is not None:
            mark('last_own_tile')
        mover.drain()
        result = None
    else:
        E = torch.zeros(b, c, round(h * sf), round(w * sf), dtype=torch.float32, device=device)
        Wt = torch.zeros_like(E)
        handles = {}               # peer tile -> (batch, index in the batch)
        state = dict(posted=0, blended=0)

        def post(upto):
            """Grouped receive of every peer tile below `upto` that has none yet."""
            peer = [t for t in range(state['posted'], upto) if owners[t] != 0]
            state['posted'] = max(state['posted'], upto)
            if peer and loopback:
                batch = mover.loop_many([tile_fn(origins[t][0], origins[t][1], tile) for t in peer], rank, device)
            elif peer:
                batch = mover.recv_many(shape, [owners[t] for t in peer], device)
            for i, t in enumerate(peer):
                handles[t] = (batch, i)

        def blend(t, out):
            hi, wi = origins[t]
            blend_fn(E, Wt, out, round(hi * sf), round(wi * sf), th, tw)

        def blend_peers_below(upto):
            for t in range(state['blended'], upto):
                batch, i = handles.pop(t)
                blend(t, mover.take(batch, i, device))
            state['blended'] = max(state['blended'], upto)

        own = [t for t in range(n) if owners[t] == 0]
        if mark is not None and not own:
            mark('last_own_tile')
        for k, t0 in enumerate(own):
            hi, wi = origins[t0]
            out = tile_fn(hi, wi, tile)
            if mark is not None and k == len(own) - 1:
                mark('last_own_tile')
            # the peers compute the tiles up to rank 0's next own tile while t0 computes here: their receives are posted NOW,
            # behind t0 on the compute stream, and consumed one round later
            post(own[k + 1] if k + 1 < len(own) else n)
            blend_peers_below(t0)                          # tiles below t0 precede it in the reference order
            blend(t0, out)
            state['blended'] = t0 + 1
        post(n)
        blend_peers_below(n)
        result = finalize_fn(E, Wt)
        if mark is not None:
            mark('finalized')
        if loopback:
            mover.drain()
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


def clip_test_distributed(restorer, x_norm, rank=None, world=None, group=None, gather_to_all=False, options=None, stats=None,
                          rank0_share=None, loopback=False):
    """Tile-sharded counterpart of CiaoSR.clip_test on the GPUs of one node.
    `stats` (optional dict) receives a stream event per probe point of `sharded_clip_test` (bench.py's exposed-tail figure).
    `rank0_share` (default `test_cfg.rank0_share` or 1): see `tile_owners`; every rank must pass the same value.
    `loopback` (1-rank group only): see `sharded_clip_test`."""
    from . import hip_ops
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    cfg = restorer.test_cfg
    sf = cfg.get('scale')
    opt = restorer.options(options)
    if rank0_share is None:
        rank0_share = cfg.get('rank0_share', None)
    rank0_share = 1.0 if rank0_share is None else float(rank0_share)

    def tile_fn(hi, wi, tile):
        out, _ = restorer.run_tile(x_norm, hi, wi, tile, sf, opt)
        return out

    # Tile batching (restorer.clip_test's `test_cfg.tile_batch`): this rank's next B tiles go through the encoder in ONE call (shared
    # dense-layer launches, encoder_hip.forward_hwc_batch: every feature map bitwise the single-tile one); the head then runs tile by
    # tile in the driver's order, so sends, receives and the blend order are unchanged.
    gen = restorer.generator
    enc = getattr(gen, '_encoder_hip', None)
    n_batch = restorer.tile_batch(opt)
    if (n_batch > 1 and x_norm.is_cuda and x_norm.shape[0] == 1 and enc is not None and hasattr(enc, 'forward_hwc_batch')
            and enc.supported() and getattr(gen, '_head', None) is not None):
        tile_sz, origins = tile_grid(x_norm.shape[-2], x_norm.shape[-1], cfg.get('tile'), cfg.get('tile_overlap'))
        owners = [0] * len(origins) if (loopback and world == 1) else tile_owners(len(origins), world, rank0_share)
        mine = [o for o, r in zip(origins, owners) if r == rank]
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
                             finalize_fn, rank, world, group, x_norm.device, gather_to_all, mark, rank0_share, loopback)


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
    ensure_communicator(group, coord.device)
    mover = _Mover(group, coord.device)
    if rank != 0:
        if q1 > q0:
            mover.send(slab[:, :q1 - q0].contiguous(), 0)
        mover.drain()
        out = None
    else:
        handles = [(r, mover.recv_many((b, s[1] - s[0], 3), [r], coord.device)) for r, s in enumerate(slices) if r > 0 and s[1] > s[0]]
        out = torch.empty(b, n_query, 3, dtype=torch.float32, device=coord.device)
        out[:, q0:q1] = slab[:, :q1 - q0]
        for r, handle in handles:
            out[:, slices[r][0]:slices[r][1]] = mover.take(handle, 0, coord.device)
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
