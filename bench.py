#!/usr/bin/env python
"""bench.py -- HR Mpix/s of the CiaoSR LocalImplicitSR forward path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], "C2"): RDN-CiaoSR (c64b16) x4, one 48x48 LR image -> 192x192,
random-init weights (seeded, default-init scale), fp32 arithmetic, synthetic DIV2K-shaped input
resident in HBM.  A step = CiaoSR.forward_test body: normalise -> clip_test (1 tile) -> RDN encoder
-> cs_attn -> head -> de-normalise/clamp, device to device.
N > 1 (weak scaling): one LR image of 48 x 48N pixels = N tiles of the same 48x48 unit
(tile_overlap 0), tile t on rank t, outputs gathered to rank 0 over RCCL and blended in the
reference order; value = final HR pixels of the whole image / max-over-ranks time.

The JSON line also carries
  roofline     algorithmic FLOPs (or bytes) of the dominant kernel per launch / its average launch
               duration measured with HIP events inside the timed region, against the gfx950 peak
  cpu_baseline the CPU oracle (op-for-op port of the reference, per-chunk cs_attn recompute included)
               timed on this host's cores on the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0           # HBM3E spec
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA


def rdn_ciaosr(test_cfg):
    from ciaosr_amd import CiaoSR, LocalImplicitSRRDN
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[256, 256, 256, 256])
    gen = dict(type=LocalImplicitSRRDN,
               encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
                            upscale_factor=4, num_layers=8, channel_growth=64),
               imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000)
    return CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'),
                  rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1., 1., 1.), test_cfg=test_cfg).eval()


def kernel_work(tag, Q, HW, C=64, hidden=256, J=4, blocks=16, layers=8):
    """Algorithmic work of ALL launches of kernel `tag` in one step of this workload (one 48x48 tile):
    (amount, 'flop'|'byte').  FLOPs = 2 x MACs of the contraction as the reference writes it, minus the
    exact layer-1 hoist; bytes for the HBM-bound K4 = SURVEY 8(d)'s 22 064 B/query."""
    D, Dv, R = 9 * C, 10 * C, Q * J
    side = HW ** 0.5
    dense = sum(2.0 * HW * 9 * (C + C * l) * C for l in range(layers)) * blocks
    table = {
        # fused kernels: phi_k + phi_v layers 2..5 per (query, shift) row; phi_q all layers per query
        # (imnet_k's output layer is folded exactly into the 9-rows-per-LR-pixel logit table: 'head_logit_table')
        'head_kv_fused': (2.0 * R * (6 * hidden * hidden + hidden * Dv), 'flop'),
        'head_logit_table': (2.0 * 9 * HW * D * hidden, 'flop'),
        'head_logit_table_bf16': (2.0 * 9 * HW * D * hidden, 'flop16'),
        'head_decode_fused': (2.0 * Q * (Dv * hidden + 3 * hidden * hidden + 3 * hidden), 'flop'),
        'head_kv_fused_bf16': (2.0 * R * (6 * hidden * hidden + hidden * Dv), 'flop16'),
        'head_decode_fused_bf16': (2.0 * Q * (Dv * hidden + 3 * hidden * hidden + 3 * hidden), 'flop16'),
        'mlp_hidden': (6 * 2.0 * R * hidden * hidden, 'flop'),
        'mlp_out_k': (2.0 * R * hidden * D, 'flop'),
        'mlp_out_v': (2.0 * R * hidden * Dv, 'flop'),
        'mlp_in_q': (2.0 * Q * Dv * hidden, 'flop'),
        'mlp_hidden_q': (3 * 2.0 * Q * hidden * hidden, 'flop'),
        'head_table': (2.0 * HW * hidden * (D + Dv), 'flop'),
        'csa_scores': (2.0 * HW * (HW / 4) * 4.5 * C, 'flop'),
        # >= 64x64 maps run the composed fold+down tail (DESIGN 'cs_attn tail'): 16C value columns instead of 36C
        'csa_attn_v': (2.0 * HW * (HW / 4) * (16 if HW >= 4096 else 36) * C, 'flop'),
        'csa_scores_bf16': (2.0 * HW * (HW / 4) * 4.5 * C, 'flop16'),
        'csa_attn_v_bf16': (2.0 * HW * (HW / 4) * 25 * C, 'flop16'),
        'csa_attn_v_edge': (2.0 * (2 * side * 4 * C + C) * (HW / 4), 'flop'),
        'csa_down_partial': (2.0 * (side / 2 + 3) ** 2 * 9 * C * 9 * C, 'flop'),
        'csa_down': (2.0 * HW * 9 * C * C, 'flop'),
        'enc_conv3x3': (2 * 2.0 * HW * 9 * C * C, 'flop'),            # sfe2 + gff.1 (dense layers run in scatter form)
        'enc_dense_scatter': (dense, 'flop'),
        'enc_dense_bf16': (dense, 'flop16'),
        'enc_dense_gather': (dense, 'flop'),
        'enc_conv1x1': (blocks * 2.0 * HW * (C + C * layers) * C + 2.0 * HW * C * blocks * C, 'flop'),
        'local_attention': (Q * (4.0 * J * D + 4.0 * J * Dv + 4.0 * Dv + 16) + 2.0 * C * 4 * HW, 'byte'),
        'head_rows': (R * 4.0 * 2 * hidden * 2, 'byte'),
    }
    return table.get(tag)


def cpu_baseline(scale=4):
    """Oracle (port of the reference, configured like it) on this host's CPU cores, same workload."""
    from oracle import ciaosr_oracle as orc          # checker / baseline only
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    model = rdn_ciaosr(dict(scale=scale, tile=192, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.0)
    params = {k[len('generator.'):]: v.detach() for k, v in model.state_dict().items()}
    lq, _ = synthetic_pair(48, 48, scale)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))                   # small ops: more threads only add contention
    torch.set_num_threads(cores)
    times = []
    for i in range(4):                              # 1 warm-up + 3 timed, ~3 s each on 8 cores
        t0 = time.perf_counter()
        out = orc.forward_test(lq, None, None, params, scale=scale, tile=192, tile_overlap=32)
        times.append(time.perf_counter() - t0)
        if sum(times) > 45:
            break
    t = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    return dict(value=out.shape[-1] * out.shape[-2] / 1e6 / t, unit='Mpix/s', cores=torch.get_num_threads(),
                kind='port', sample=f'same workload (1 LR 48x48 -> 192x192 image), median of {len(times) - 1} '
                f'runs after 1 warm-up, {t * 1e3:.0f} ms/img, torch CPU fp32, reference-style per-chunk cs_attn')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help='fp32 (default): exact-fp32 MFMA everywhere; bf16: bf16 MFMA inputs in the fused head only')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c2q', 'c3tile', 'c3', 'c3s'],
                    help='c2 (default, BASELINE configs[1]): LR 48x48; c3tile: one 192x192 LR tile; c3: LR 1356x2040 (117 tiles); '
                         'c3s: LR 339x510 (6 tiles); c2q: the ONE 48x48 image of C2 with its query range sharded over the ranks after a '
                         'broadcast of the encoder features (strong scaling).  c3 / c3s with --gpus N > 1 shard the tiles of the ONE image over the ranks (C4, strong scaling)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world == 1:
        raise SystemExit('--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
    assert world == args.gpus, f'WORLD_SIZE {world} != --gpus {args.gpus}'
    n_dev = torch.cuda.device_count()
    backend = os.environ.get('CIAOSR_DIST_BACKEND', 'nccl')   # 'gloo': rehearse the N-rank path on one GPU
    if backend == 'nccl' and world > n_dev:
        raise SystemExit(f'{world} ranks but only {n_dev} GPUs visible (RCCL needs one GPU per rank)')
    local_dev = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ciaosr_amd import hip_ops, _lib
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.tile_shard import clip_test_distributed, predict_query_sharded
    from ciaosr_amd.coords import make_coord, make_cell
    _lib.load()
    hip_ops.set_precision(args.precision)

    scale, lr = 4, 48
    if args.workload == 'c3tile':
        assert world == 1, 'c3tile is a single-tile (single-GPU) measurement'
    weak = args.workload == 'c2'                      # c2: one 48x48 tile per rank; c3 / c3s: one image, tiles sharded
    test_cfg = dict(scale=scale, tile=192, tile_overlap=32) if (world == 1 or not weak) else dict(scale=scale, tile=lr, tile_overlap=0)
    model = rdn_ciaosr(test_cfg)
    seeded_init_(model, seed=0, gain=1.0)             # default-init scale for timing
    model = model.to(dev)
    lr_h, lr_w = {'c2': (lr, lr * world), 'c2q': (lr, lr), 'c3tile': (192, 192), 'c3': (1356, 2040), 'c3s': (339, 510)}[args.workload]
    lq, _ = synthetic_pair(lr_h, lr_w, scale)         # identical on every rank (CPU-generated)
    lq = lq.to(dev)
    out_pixels = (lr_h * scale) * (lr_w * scale)

    if args.workload == 'c2q':
        q_coord = make_coord((lr_h * scale, lr_w * scale)).unsqueeze(0).to(dev)
        q_cell = make_cell((lr_h * scale, lr_w * scale)).unsqueeze(0).to(dev)

    def step():
        if world == 1:
            return model.restore(lq)
        x = model.normalize(lq)
        if args.workload == 'c2q':
            pred = predict_query_sharded(model, x, q_coord, q_cell, rank, world)
            if rank == 0:
                return hip_ops.denorm_clamp(pred[0].contiguous(), lr_h * scale, lr_w * scale, model.rgb_mean, model.rgb_std)
            return None
        pred = clip_test_distributed(model, x, rank, world)
        if rank == 0:
            return hip_ops.denorm_clamp(pred[0].contiguous(), lr_h * scale, lr_w * scale, model.rgb_mean, model.rgb_std)
        return None

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if world > 1:
        # correctness of the sharded path before timing it: rank 0's blended image must be bitwise equal to the
        # single-process clip_test of the same image on this GPU
        out = step()
        if rank == 0:
            ref = model.restore(lq)[0]
            assert torch.equal(out, ref), f'tile-sharded output differs from 1-GPU output by {(out - ref).abs().max().item()}'


    # warm-up; the last warm-up step is fully profiled to find the dominant kernel
    for _ in range(max(args.warmup - 1, 0)):
        step()
    with hip_ops.profile():
        step()
        torch.cuda.synchronize(dev)
    prof_all = hip_ops.profile.results()
    dominant = max(prof_all, key=lambda k: prof_all[k]['total_ms']) if prof_all else None

    # timed region.  The dominant kernel carries HIP-event pairs inside it only when it is launched a few times
    # per step: a pair costs ~4 us on the GPU timeline, and the 128 convolution launches of a 48x48 RDN pass
    # would inflate a 5 ms step by ~1 ms.  Otherwise its duration is measured in a separate pass right after.
    lib = _lib.load()
    live = bool(dominant) and prof_all[dominant]['launches'] <= 8
    lib.ciaosr_prof_filter(dominant.encode() if dominant else None)
    lib.ciaosr_prof_reset()
    lib.ciaosr_prof_enable(1 if live else 0)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    lib.ciaosr_prof_enable(0)
    prof_steps = args.steps
    # the separate pass contains collectives when world > 1: every rank must take it if ANY rank needs it (ranks can
    # disagree on the dominant kernel, e.g. the query-sharded mode runs the encoder on rank 0 only)
    need_pass = int(bool(dominant) and not live)
    if world > 1:
        flag = torch.tensor([need_pass], device=dev if backend == 'nccl' else 'cpu', dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        need_pass = int(flag.item())
    if need_pass:
        live = False
        prof_steps = min(args.steps, 10)
        lib.ciaosr_prof_reset()
        lib.ciaosr_prof_enable(1)
        for _ in range(prof_steps):
            step()
        sync()
        lib.ciaosr_prof_enable(0)
    prof_dom = hip_ops.profile.results()
    lib.ciaosr_prof_filter(None)

    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        roof = None
        if dominant and dominant in prof_dom:
            tile_lr = lr if args.workload in ('c2', 'c2q') else 192
            n_tiles = {'c2': 1, 'c2q': 1, 'c3tile': 1, 'c3': 117, 'c3s': 6}[args.workload]
            if world > 1 and not weak:
                n_tiles = (n_tiles + world - 1) // world          # tiles t = 0, R, 2R, ... run on rank 0 (whose kernels are timed)
            Q, HW = (tile_lr * scale) ** 2, tile_lr * tile_lr
            work = kernel_work(dominant, Q, HW)
            if work:
                work = (work[0] * n_tiles, work[1])
            avg_ms = prof_dom[dominant]['avg_ms']
            step_ms = prof_dom[dominant]['total_ms'] / prof_steps      # all launches of the tag in one step
            if work:
                amount, kind = work
                if kind in ('flop', 'flop16'):
                    peak = PEAK_F32_MFMA_TFLOPS if kind == 'flop' else PEAK_BF16_MFMA_TFLOPS
                    ach = amount / (step_ms * 1e-3) / 1e12
                    roof = dict(bound='mfma', achieved=round(ach, 3), peak=peak, unit='TFLOP/s',
                                frac=round(ach / peak, 4), traffic=None)
                else:
                    ach = amount / (step_ms * 1e-3) / 1e9
                    roof = dict(bound='hbm', achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                                frac=round(ach / PEAK_HBM_GBS, 4), traffic=None)
                roof.update(kernel=dominant, timing='HIP events inside the timed region' if live else
                            f'HIP events in a separate pass of {prof_steps} steps (too many launches per step to bracket live)',
                            avg_launch_ms=round(avg_ms, 5),
                            launches=prof_dom[dominant]['launches'],
                            share_of_step=round(prof_all[dominant]['total_ms'] /
                                                max(sum(v['total_ms'] for v in prof_all.values()), 1e-9), 3))
                # the other kernels of the step against their own rooflines (one fully profiled step)
                others = {}
                for tag, pr in prof_all.items():
                    wk = kernel_work(tag, Q, HW)
                    if tag == dominant or not wk or pr['total_ms'] < 0.02:
                        continue
                    amt, knd = wk[0] * n_tiles, wk[1]
                    rate = amt / (pr['total_ms'] * 1e-3)
                    pk = {'flop': PEAK_F32_MFMA_TFLOPS * 1e12, 'flop16': PEAK_BF16_MFMA_TFLOPS * 1e12, 'byte': PEAK_HBM_GBS * 1e9}[knd]
                    others[tag] = dict(bound='hbm' if knd == 'byte' else 'mfma', ms_per_step=round(pr['total_ms'], 4),
                                       frac=round(rate / pk, 4))
                roof['other_kernels'] = others
                # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
                # WRITE_SIZE, tools/pmc_summary.py); offline evidence, null when the summary is absent
                tag2fn = {'enc_dense_scatter': 'ciaosr::dense_scatter_small_kernel', 'head_kv_fused': 'ciaosr::head_kv_fused_kernel',
                          'head_decode_fused': 'ciaosr::head_decode_fused_kernel'}
                pmc_path = os.path.join(REPO, 'profiles', 'r1_c2_pmc_hbm_traffic.json')
                if args.workload == 'c2' and os.path.exists(pmc_path) and dominant in tag2fn:
                    # the tag may be served by several instantiations of one kernel template: launch-weighted mean
                    hits = [v for k, v in json.load(open(pmc_path)).items() if tag2fn[dominant] in k]
                    pmc = None
                    if hits:
                        n_l = sum(h['launches'] for h in hits)
                        pmc = dict(hbm_bytes_per_launch=round(sum(h['hbm_bytes_per_launch'] * h['launches'] for h in hits) / max(n_l, 1)))
                    if pmc:
                        roof['traffic'] = pmc['hbm_bytes_per_launch']
                        roof['traffic_note'] = 'bytes per launch, rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE, profiles/r1_c2_pmc_hbm_traffic.json'
                        if dominant == 'enc_dense_scatter':   # input group + stacked weights + read-modify-write of the running sums (mean N = 288)
                            roof['algorithmic_bytes_per_launch'] = round(HW * 4.0 * (64 + 9 * 64 * 288 / HW + 2 * 288))
        line = {
            'metric': 'HR Mpix/s (RDN-CiaoSR x4, LocalImplicitSR forward_test)',
            'value': round(out_pixels / 1e6 / (elapsed / args.steps), 4),
            'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak' if weak else 'strong', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'fp32' else 'bf16 MFMA inputs (fp32 accumulate) in the head, the big-map dense layers and cs_attn contractions; small maps keep the f32 trunk',
            'data': 'synthetic',
            'config': {'workload': {'c2': 'C2: RDN-CiaoSR (c64b16) x4, LR 48x48 -> 192x192 per GPU, random-init weights, fp32',
                                    'c2q': 'C2 (query-sharded): RDN-CiaoSR (c64b16) x4, ONE LR 48x48 -> 192x192 image, random-init weights, fp32',
                                    'c3tile': 'C3 unit: RDN-CiaoSR x4, one 192x192 LR tile -> 768x768, random-init weights, fp32',
                                    'c3': 'C3: RDN-CiaoSR x4, LR 1356x2040 -> 5424x8160, 117 tiles of 192 (overlap 32), random-init weights, fp32',
                                    'c3s': 'C3 (DIV2K-val pairing): RDN-CiaoSR x4, LR 339x510 -> 1356x2040, 6 tiles of 192 (overlap 32), random-init weights, fp32'}[args.workload]
                                   + ('' if world == 1 else (f'; encoder on rank 0, RCCL broadcast of the feature map, query range sharded over {world} GPUs' if args.workload == 'c2q' else
                                                              f'; one {lr}x{lr * world} LR image, {world} tiles sharded one per GPU' if weak else
                                                              f'; tiles of the one image sharded over {world} GPUs (tile t -> rank t % {world})')
                                      + (', RCCL all_gather of the RGB slices' if args.workload == 'c2q' else ', RCCL all_gather + rank-0 blend')),
                       'lr': [lr_h, lr_w], 'scale': scale, 'queries_per_step': out_pixels,
                       'parallelism': (f'query-shard x{world}' if args.workload == 'c2q' else f'tile-shard x{world}')},
            'roofline': roof,
            'kernels_ms_per_step': {k: round(v['total_ms'], 4) for k, v in sorted(
                prof_all.items(), key=lambda kv: -kv[1]['total_ms'])},
        }
        if world == 1 and args.precision == 'fp32' and roof is not None:
            # the HBM-bound kernels of the staged route (north star: >= 40 % of the HBM roofline on the local-attention
            # kernel K4), measured on the same workload right after the timed region: one staged step, HIP events
            try:
                hip_ops.set_head_mode(1)
                step()
                with hip_ops.profile():
                    step()
                    torch.cuda.synchronize(dev)
                st = hip_ops.profile.results()
            finally:
                hip_ops.set_head_mode(0)
            tile_lr = lr if args.workload in ('c2', 'c2q') else 192
            n_tiles = {'c2': 1, 'c2q': 1, 'c3tile': 1, 'c3': 117, 'c3s': 6}[args.workload]
            hb = {}
            for tag in ('local_attention', 'head_rows'):
                wk = kernel_work(tag, (tile_lr * scale) ** 2, tile_lr * tile_lr)
                if tag in st and wk:
                    gbs = wk[0] * n_tiles / (st[tag]['total_ms'] * 1e-3) / 1e9
                    hb[tag] = dict(bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                                   frac=round(gbs / PEAK_HBM_GBS, 4), ms_per_step=round(st[tag]['total_ms'], 4))
            roof['staged_path_hbm_kernels'] = hb
        if world == 1 and not args.no_cpu_baseline and args.workload == 'c2':
            line['cpu_baseline'] = cpu_baseline(scale)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
