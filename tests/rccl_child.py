"""ONE fresh process, ONE GPU, a 1-rank 'nccl' (= RCCL) process group (started by tests/test_rccl_single_gpu.py; not collected by
pytest).  Executes every RCCL-only branch of the tile exchange that needs no second GPU: communicator creation under
`rccl_env_defaults()`, `ensure_communicator` on the device, the grouped isend + irecv hand-off with device tensors behind queued
work (`rccl_self_probe`), and the PRODUCT path in loopback (`clip_test_distributed(..., loopback=True)`: every second tile of the
6-tile image travels through `_Mover` to this rank) -- bitwise `CiaoSR.restore`.  Reference being replaced: tools/test.py:82-86
(init_dist) and :124-146 (multi_gpu_test)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from ciaosr_amd.tile_shard import rccl_env_defaults
    channels = rccl_env_defaults()                   # before the first HIP call of the process (HSA reads its variable at init)
    import torch
    import torch.distributed as dist
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', init_method=f"tcp://127.0.0.1:{os.environ['CIAOSR_CHILD_PORT']}", rank=0, world_size=1, device_id=dev)

    from bench import rdn_ciaosr
    from ciaosr_amd import _lib, hip_ops
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.tile_shard import clip_test_distributed, rccl_self_probe, ensure_communicator, _warm_groups
    _lib.load()
    probe = rccl_self_probe(dev)
    assert probe['ranks'] == 1 and probe['backend'] == 'nccl' and probe['broadcast_ok'] and channels == 1, probe
    for label in ('cold', 'warm'):
        p = probe[label]
        assert p['tile_bitwise'] and p['consumer_bitwise'] and p['tile_nonconstant'], (label, p)
    # no host block: everything from the producer's first kernel to the consumer was enqueued while the producer still ran
    # (a wall-clock property: a loaded host can stall the enqueueing thread, so the colder of the two probes may miss it -- one of them
    # showing it is the evidence that nothing in the exchange code blocks the host)
    assert min(probe[k]['host_enqueue_ms'] / max(probe[k]['producer_gpu_ms'], 1e-9) for k in ('cold', 'warm')) < 0.9, probe
    assert ensure_communicator(None, dev) == 1 and len(_warm_groups) == 1

    scale = 4
    model = rdn_ciaosr(dict(scale=scale, tile=192, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.0)
    model = model.to(dev)
    lq = synthetic_pair(339, 510, scale)[0].to(dev)
    checked = []
    for precision, tile_batch, share in (('fp32', 8, 1.0), ('f16', 2, 1.0), ('f16', 8, 0.0)):
        opt = hip_ops.Options(precision)
        model.test_cfg['tile_batch'] = tile_batch
        stats = {}
        pred = clip_test_distributed(model, model.normalize(lq), 0, 1, options=opt, stats=stats, rank0_share=share, loopback=True)
        out = hip_ops.denorm_clamp(pred[0].contiguous(), 339 * scale, 510 * scale, model.rgb_mean, model.rgb_std)
        model.test_cfg['tile_batch'] = 8
        ref = model.restore(lq, options=opt)[0]
        assert torch.equal(out, ref), (precision, tile_batch, share, (out - ref).abs().max().item())
        assert float(out.std()) > 1e-3 and 'finalized' in stats and 'last_own_tile' in stats
        checked.append(f'loopback/{precision}/batch{tile_batch}/share{share:g}')
    torch.cuda.synchronize(dev)
    print('RCCL1_OK', json.dumps(dict(probe=probe, checked=checked)), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
