"""One rank of the N-rank product path (started by tests/test_multirank_gpu.py through torch.distributed.run; not collected by
pytest).  Runs `tile_shard.clip_test_distributed` with the real HIP tile function (incl. its batched-encoder variant) and
`tile_shard.predict_query_sharded`, and asserts on rank 0 that each is BITWISE the single-process `CiaoSR.restore` of the same
input -- the replacement of the reference's multi-GPU test (tools/test.py:124-146 -> multi_gpu_test; the tiles of
mmedited/models/restorers/ciaosr.py:233-254 are the sharded units).  With fewer GPUs than ranks the ranks share a GPU and the
exchange goes through gloo (CIAOSR_DIST_BACKEND=gloo, host-staged copies); with one GPU per rank it runs over RCCL."""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from ciaosr_amd.tile_shard import rccl_env_defaults
    rccl_env_defaults()                  # before the first HIP call of the rank (HSA reads its IPC mode when it initialises)
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    n_dev = torch.cuda.device_count()
    backend = os.environ.get('CIAOSR_DIST_BACKEND') or ('nccl' if n_dev >= world else 'gloo')
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % max(n_dev, 1))
    torch.cuda.set_device(dev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)

    from bench import rdn_ciaosr
    from ciaosr_amd import _lib, hip_ops
    from ciaosr_amd.coords import make_cell, make_coord
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.tile_shard import clip_test_distributed, predict_query_sharded
    _lib.load()
    scale = 4
    model = rdn_ciaosr(dict(scale=scale, tile=192, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.0)
    model = model.to(dev)
    checked = []

    if os.environ.get('CIAOSR_CHILD_CASE') == 'c4':
        # C4 at its real shape: the 117 tiles of the LR 1356x2040 image over the ranks (8 ranks: 14 full rounds through the
        # two-round receive ring + a ragged round of 5 that rank 0 sits out), f16 (short) and one fp32 run
        from ciaosr_amd.tile_shard import StepDeadline, partition
        watchdog = StepDeadline(float(os.environ.get('CIAOSR_STEP_DEADLINE_S', '900')), what='C4 rehearsal step', rank=rank)
        lq = synthetic_pair(1356, 2040, scale)[0].to(dev)
        for precision in os.environ.get('CIAOSR_CHILD_PRECISIONS', 'f16,fp32').split(','):
            watchdog.beat()
            opt = hip_ops.Options(precision)
            x = model.normalize(lq)
            stats = {}
            pred = clip_test_distributed(model, x, rank, world, options=opt, stats=stats)
            if rank == 0:
                out = hip_ops.denorm_clamp(pred[0].contiguous(), 1356 * scale, 2040 * scale, model.rgb_mean, model.rgb_std)
                del pred
                watchdog.beat()
                ref = model.restore(lq, options=opt)[0]
                assert torch.equal(out, ref), (precision, (out - ref).abs().max().item())
                assert float(out.std()) > 1e-3 and 'finalized' in stats
                checked.append(f'c4/{precision}')
                del out, ref
            else:
                assert pred is None
            watchdog.beat()
            dist.barrier()
        torch.cuda.synchronize(dev)
        if rank == 0:
            print('MULTIRANK_OK', world, backend, ' '.join(checked), 'tiles_per_rank',
                  ','.join(str(len(p)) for p in partition(117, world)), flush=True)
        dist.barrier()
        watchdog.stop()
        dist.destroy_process_group()
        return

    # (1) C4 in small: the 6 tiles of a 339x510 LR image over the ranks, default tile batching (8) and batches of 2
    lq = synthetic_pair(339, 510, scale)[0].to(dev)
    for precision in ('fp32', 'f16'):
        opt = hip_ops.Options(precision)
        for tile_batch in (8, 2, 1):
            model.test_cfg['tile_batch'] = tile_batch
            x = model.normalize(lq)
            stats = {}
            pred = clip_test_distributed(model, x, rank, world, options=opt, stats=stats)
            assert 'last_own_tile' in stats
            if rank == 0:
                out = hip_ops.denorm_clamp(pred[0].contiguous(), 339 * scale, 510 * scale, model.rgb_mean, model.rgb_std)
                model.test_cfg['tile_batch'] = 8
                ref = model.restore(lq, options=opt)[0]
                assert torch.equal(out, ref), (precision, tile_batch, (out - ref).abs().max().item())
                assert float(out.std()) > 1e-3 and 'finalized' in stats
                checked.append(f'tiles/{precision}/batch{tile_batch}')
            else:
                assert pred is None
            dist.barrier()
    model.test_cfg['tile_batch'] = 8

    # gather_to_all: every rank ends up with rank 0's image
    x = model.normalize(lq)
    pred = clip_test_distributed(model, x, rank, world, gather_to_all=True)
    ref = model.clip_test(x, model.generator)
    assert torch.equal(pred, ref)
    checked.append('tiles/gather_to_all')
    dist.barrier()

    # (2) single-tile config (C2): encoder on rank 0, feature broadcast, query range sharded, RGB slices to rank 0
    lq2 = synthetic_pair(48, 48, scale)[0].to(dev)
    coord = make_coord((192, 192)).unsqueeze(0).to(dev)
    cell = make_cell((192, 192)).unsqueeze(0).to(dev)
    model.test_cfg['tile'] = None
    for precision in ('fp32', 'f16'):
        opt = hip_ops.Options(precision)
        x2 = model.normalize(lq2)
        pred = predict_query_sharded(model, x2, coord, cell, rank, world, options=opt)
        if rank == 0:
            out = hip_ops.denorm_clamp(pred[0].contiguous(), 192, 192, model.rgb_mean, model.rgb_std)
            ref = model.restore(lq2, coord, cell, options=opt)[0]
            assert torch.equal(out, ref), (precision, (out - ref).abs().max().item())
            checked.append(f'queries/{precision}')
        else:
            assert pred is None
        dist.barrier()

    torch.cuda.synchronize(dev)
    if rank == 0:
        print('MULTIRANK_OK', world, backend, ' '.join(checked), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
