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


def test_query_slices_cover_the_range_and_respect_chunks():
    from ciaosr_amd.tile_shard import query_slices
    for n, w in ((36864, 8), (24964, 3), (5, 8), (30001, 2)):
        s = query_slices(n, w)
        assert s[0][0] == 0 and s[-1][1] == n and all(a[1] == b[0] for a, b in zip(s, s[1:]))
        assert max(q1 - q0 for q0, q1 in s) - min(q1 - q0 for q0, q1 in s) <= 1
    s = query_slices(36864, 4, chunk=30000, uniform_cell=False)       # varying cell: boundaries on eval_bsize chunks
    assert s == [(0, 30000), (30000, 36864), (36864, 36864), (36864, 36864)]


def _query_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from ciaosr_amd.tile_shard import query_sharded_predict
    feature_fn, predict_fn, coord, cell = _query_problem()
    out = query_sharded_predict(feature_fn, predict_fn, coord, cell, rank, world, chunk=200)
    if rank == 0:
        ret['out'] = out
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _query_problem():
    """Tiny head on the oracle: the feature map comes from rank 0 only (the others receive it by broadcast)."""
    from oracle import ciaosr_oracle as orc
    fx = load_golden('tiny_head_s2p7')
    P = {k: v for k, v in weights_from(fx).items()}
    feat = torch.from_numpy(fx['feature'])
    coord, cell = torch.from_numpy(fx['coord']), torch.from_numpy(fx['cell'])

    def predict_fn(feature, c, cl):
        return orc.query_rgb(feature, c, cl, P)

    return (lambda: feat.clone()), predict_fn, coord, cell


@pytest.mark.slow
def test_query_sharded_two_rank_gloo_equals_single_process():
    """Single-tile configs: the query range is sharded after a broadcast of the encoder features."""
    from ciaosr_amd.tile_shard import query_sharded_predict
    feature_fn, predict_fn, coord, cell = _query_problem()
    single = query_sharded_predict(feature_fn, predict_fn, coord, cell, 0, 1, chunk=200)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_query_worker, args=(2, 29593, ret), nprocs=2, join=True)
    # the torch-CPU stand-in head is not bitwise row-independent (GEMM blocking depends on the row count); the HIP head is
    # (bench.py asserts bitwise equality on the GPU)
    assert (ret['out'] - single).abs().max() < 1e-5
    fx = load_golden('tiny_head_s2p7')
    assert (single - torch.from_numpy(fx['out'])).abs().max() < 2e-5


def _collect_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tools.test import collect_results
    n = 7
    mine = [dict(eval_result=dict(PSNR=float(10 + i), SSIM=0.1 * i)) for i in range(n) if i % world == rank]
    out = collect_results(mine, n, rank, world)
    if rank == 0:
        ret['out'] = out
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_image_sharded_eval_results_are_collected_in_dataset_order():
    """tools/test.py with --launcher pytorch on a whole-image config shards the images over the ranks; the Eval-PSNR /
    Eval-SSIM it prints must equal the single-process numbers (every image, dataset order)."""
    from ciaosr_amd.dataset import SRFolderDataset
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_collect_worker, args=(2, 29595, ret), nprocs=2, join=True)
    single = [dict(eval_result=dict(PSNR=float(10 + i), SSIM=0.1 * i)) for i in range(7)]
    assert ret['out'] == single
    assert SRFolderDataset.evaluate(ret['out']) == SRFolderDataset.evaluate(single)
