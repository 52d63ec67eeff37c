#!/usr/bin/env python
"""Test CLI with the reference's interface (tools/test.py:28-61):

    python tools/test.py CONFIG CHECKPOINT [--save-path DIR] [--out FILE] [--launcher none|pytorch] ...

Config = a python config file (this repo's configs/ or the reference's parsable ones); checkpoint = an
mmcv-style .pth ({'state_dict': ...} or a bare state_dict, keys `generator.*`).  Single process: every image
runs on cuda:0.  `--launcher pytorch` (started by torch.distributed.run): the TILES of each image are sharded
over the ranks (ciaosr_amd/tile_shard.py) instead of the reference's image-level sharding.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def self_launch(argv):
    """`tools/test.py CONFIG CKPT --launcher pytorch --gpus N` from a plain shell (no WORLD_SIZE): start the N ranks as ONE
    child `python -m torch.distributed.run ... tools/test.py <same args>` and exit with its return code -- what the reference's
    tools/dist_test.sh:8-10 does with torch.distributed.launch.  Runs before torch is imported, so this parent never
    initialises the GPU and nothing is exec'ed."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--launcher', default='none')
    ap.add_argument('--gpus', type=int, default=0)
    a = ap.parse_known_args(argv)[0]
    if a.launcher != 'pytorch' or 'WORLD_SIZE' in os.environ:
        return
    n = a.gpus
    if n <= 0:
        import torch                              # device_count() does not initialise the GPU on this image
        n = max(torch.cuda.device_count(), 1)
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('NCCL_MIN_P2P_NCHANNELS', '1')       # tile_shard.rccl_env_defaults (torch is not imported yet here)
    env.setdefault('NCCL_MAX_P2P_NCHANNELS', '1')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == '__main__':
    self_launch(sys.argv[1:])

import torch                                      # noqa: E402 - after self_launch on purpose


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='ciaosr_amd tester')
    p.add_argument('config', help='test config file path')
    p.add_argument('checkpoint', help='checkpoint file ("None" = cfg.test_checkpoint_path)')
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--out', help='output result pickle file')
    p.add_argument('--gpu-collect', action='store_true')
    p.add_argument('--save-path', default=None, type=str, help='path to store images')
    p.add_argument('--tmpdir')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--gpus', type=int, default=0, help='with --launcher pytorch from a plain shell: ranks to start (default: every visible GPU)')
    p.add_argument('--lq-folder', default=None, help='override cfg.data.test.lq_folder')
    p.add_argument('--gt-folder', default=None, help='override cfg.data.test.gt_folder')
    return p.parse_args(argv)


def collect_results(mine, n_items, rank, world, group=None):
    """Per-rank result lists of an `i % world == rank` image shard -> the full list in dataset order on rank 0
    (None elsewhere)."""
    import torch.distributed as dist
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0, group=group)
    if rank != 0:
        return None
    assert sum(len(p) for p in parts) == n_items, 'a rank lost or duplicated an image'
    return [parts[i % world][i // world] for i in range(n_items)]


def main(argv=None):
    args = parse_args(argv)
    import torch.distributed as dist
    import ciaosr_amd
    from ciaosr_amd.checkpoint import load_checkpoint
    from ciaosr_amd.config import Config
    from ciaosr_amd.dataset import SRFolderDataset
    from ciaosr_amd.tile_shard import StepDeadline, clip_test_distributed, rccl_env_defaults

    cfg = Config.fromfile(args.config)
    if args.checkpoint in (None, 'None'):
        args.checkpoint = cfg.get('test_checkpoint_path')
    distributed = args.launcher != 'none'
    rank, world = 0, 1
    if distributed:
        rccl_env_defaults()      # first: ROCr reads HSA_ENABLE_IPC_MODE_LEGACY at the first HIP call (set_device below), RCCL its NCCL_* at init
        rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', args.local_rank)) % max(torch.cuda.device_count(), 1))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # CIAOSR_DIST_BACKEND=gloo: rehearse the N-rank path on fewer GPUs than ranks (host-staged copies)
        import datetime
        backend = os.environ.get('CIAOSR_DIST_BACKEND') or cfg.get('dist_params', {}).get('backend', 'nccl')
        deadline_s = float(os.environ.get('CIAOSR_STEP_DEADLINE_S', '900'))          # per image; a lost peer ends the job, not hangs it
        kw = dict(device_id=torch.device('cuda', torch.cuda.current_device())) if backend == 'nccl' else {}
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=max(deadline_s, 60.0)), **kw)
        watchdog = StepDeadline(deadline_s, what='image', rank=rank)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    dev = torch.device('cuda', torch.cuda.current_device())

    tcfg = cfg.data['test']
    dataset = SRFolderDataset(args.lq_folder or tcfg['lq_folder'], args.gt_folder or tcfg['gt_folder'],
                              scale=tcfg.get('scale', 4), filename_tmpl=tcfg.get('filename_tmpl', '{}'))
    model = ciaosr_amd.build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    if args.checkpoint:
        load_checkpoint(model, args.checkpoint, map_location='cpu')
    model = model.to(dev).eval()

    results = []
    for i in range(len(dataset)):
        if distributed:
            watchdog.beat()
        d = dataset[i]
        lq, gt = d['lq'].unsqueeze(0).to(dev), d['gt'].unsqueeze(0).to(dev)
        coord, cell = d['coord'].unsqueeze(0).to(dev), d['cell'].unsqueeze(0).to(dev)
        save = args.save_path is not None
        if world > 1 and model.test_cfg.get('tile', None):
            # tile-sharded inference of this image; rank 0 evaluates
            x = model.normalize(lq)
            pred = clip_test_distributed(model, x, rank, world)
            if rank != 0:
                continue
            from ciaosr_amd import hip_ops, metrics
            h, w = round(lq.shape[-2] * model.test_cfg.scale), round(lq.shape[-1] * model.test_cfg.scale)
            out = hip_ops.denorm_clamp(pred[0].contiguous(), h, w, model.rgb_mean, model.rgb_std).unsqueeze(0)
            gt_img = gt.view(1, h, w, 3).permute(0, 3, 1, 2).contiguous()
            res = dict(eval_result=model.evaluate(out, gt_img))
            if save:
                from ciaosr_amd.imageio import imwrite
                name = os.path.splitext(os.path.basename(d['meta']['gt_path']))[0]
                imwrite(metrics.tensor2img(out), os.path.join(args.save_path, f'{name}.png'))
        else:
            if world > 1 and i % world != rank:
                continue
            res = model(lq=lq, gt=gt, test_mode=True, coord=coord, cell=cell, meta=[d['meta']], save_image=save,
                        save_path=args.save_path)
        results.append(res)
    if world > 1 and not model.test_cfg.get('tile', None):
        # whole-image configs shard the IMAGES (i % world == rank): collect every rank's results on rank 0 and
        # re-interleave them into dataset order before evaluating, like the reference's multi_gpu_test
        # (tools/test.py:82-146 -> mmedit.apis.multi_gpu_test collects before dataset.evaluate)
        results = collect_results(results, len(dataset), rank, world)
    if rank == 0 and results and 'eval_result' in results[0]:
        stats = SRFolderDataset.evaluate(results)
        print()
        for k, v in stats.items():
            print(f'Eval-{k}: {v}')
        if args.out:
            import pickle
            with open(args.out, 'wb') as f:
                pickle.dump(results, f)
    if distributed:
        watchdog.beat()
        dist.barrier()
        watchdog.stop()
        dist.destroy_process_group()
    return results


if __name__ == '__main__':
    main()
