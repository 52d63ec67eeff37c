"""N>1 path on CPU: world_size-2 gloo run of the tile-sharded clip_test must be bitwise equal to
the single-process run (same tile function = the oracle generator; only the driver is under test)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import load_golden, weights_from


def _cpu_blend(E, Wt, out, y0, x0, th, tw):
    b = E.shape[0]
    E[..., y0:y0 + th, x0:x0 + tw] += out.view(b, th, tw, 3).permute(0, 3, 1, 2)
    Wt[..., y0:y0 + th, x0:x0 + tw] += 1


def _cpu_finalize(E, Wt):
    return (E / Wt).view(E.shape[0], 3, -1).permute(0, 2, 1).contiguous()


def _tile_fn_factory(fx):
    from oracle import ciaosr_oracle as orc
    P = {k[len('generator.'):]: v for k, v in weights_from(fx).items()}
    lq = torch.from_numpy(fx['lq'])
    x = lq - torch.tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)

    def tile_fn(hi, wi, tile):
        patch = x[..., hi:hi + tile, wi:wi + tile]
        th, tw = tile * 2, tile * 2
        coord = orc.make_coord((th, tw)).unsqueeze(0)
        cell = orc.make_cell((th, tw)).unsqueeze(0)
        return orc.generator_forward(patch, coord, cell, P)
    return tile_fn, tuple(x.shape)


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from ciaosr_amd.tile_shard import sharded_clip_test
    fx = load_golden('tiling_small')
    tile_fn, shape = _tile_fn_factory(fx)
    out = sharded_clip_test(shape, 48, 16, 2, tile_fn, _cpu_blend, _cpu_finalize, rank, world)
    if rank == 0:
        ret['out'] = out
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_gloo_equals_single_process_bitwise():
    from ciaosr_amd.tile_shard import sharded_clip_test
    fx = load_golden('tiling_small')
    tile_fn, shape = _tile_fn_factory(fx)
    single = sharded_clip_test(shape, 48, 16, 2, tile_fn, _cpu_blend, _cpu_finalize, 0, 1)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29591, ret), nprocs=2, join=True)
    assert torch.equal(ret['out'], single)
    # and both equal the reference's own tiled output (after denorm/clamp)
    mean = torch.tensor((0.4488, 0.4371, 0.4040)).view(1, 1, 3)
    img = (single + mean).clamp(0, 1).view(1, 200, 264, 3).permute(0, 3, 1, 2)
    assert (img - torch.from_numpy(fx['out'])).abs().max() < 5e-5


def test_partition_covers_every_tile_once():
    from ciaosr_amd.tile_shard import partition
    for n, w in ((117, 8), (117, 2), (6, 4), (1, 8), (22, 8)):
        parts = partition(n, w)
        assert sorted(t for p in parts for t in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
