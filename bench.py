#!/usr/bin/env python
"""bench.py -- HR Mpix/s of the CiaoSR LocalImplicitSR forward path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: either launched by torch.distributed.run, one rank per GPU,
                                                          or from a plain shell -- bench.py then starts its N ranks itself)

Workload (default, every N): BASELINE.json's metric config "RDN-CiaoSR x4 on 2K LR input, tiled inference" (C3; C4 for
N > 1): RDN-CiaoSR (c64b16) x4, ONE synthetic DIV2K-shaped LR image of 1356x2040 -> 5424x8160, `clip_test` tiling
(tile 192, overlap 32 -> 117 tiles), random-init weights (seeded, default-init scale), fp32 arithmetic (the
reference's), input resident in HBM.  A step = the CiaoSR.forward_test body on that image: normalise -> per tile
[RDN trunk -> cs_attn -> head] -> overlap blend -> de-normalise/clamp, device to device.
N > 1 (C4, STRONG scaling): the 117 tiles of the same image are sharded over the ranks (tile t -> rank t % N), output
tiles travel to rank 0 over RCCL while later tiles compute, rank 0 blends in the reference order (bitwise equal to the
1-GPU image: asserted before timing); value = HR pixels of the image / max-over-ranks time.

The JSON line also carries
  roofline     algorithmic FLOPs (or bytes) of the dominant kernel per launch / its average launch
               duration measured with HIP events on the launch stream inside the timed region, against the gfx950 peak
  cpu_baseline the CPU oracle (op-for-op port of the reference, per-chunk cs_attn recompute included) timed on this
               host's cores on a bounded sample of one tile and extrapolated (rank 0, N = 1 only)
  extras       C2 (48x48 LR) and one-tile bf16-mode timings, and the staged K4 local-attention kernel against HBM peak.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def self_launch(argv):
    """`python bench.py --gpus N` from a plain shell with N > 1 (no WORLD_SIZE in the environment): start the N ranks here, the
    way the reference's tools/dist_test.sh:8-10 does (`python -m torch.distributed.launch --nproc_per_node=$GPUS ... test.py`):
    one fresh child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <same args>`; its stdout (rank 0's JSON line) is inherited, this process exits with its return code.  Runs BEFORE
    torch is imported: the parent never touches the GPU and nothing is exec'ed from an initialised process."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--gpus', type=int, default=1)
    n = ap.parse_known_args(argv)[0].gpus
    if n <= 1 or 'WORLD_SIZE' in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault('NCCL_MIN_P2P_NCHANNELS', '1')              # tile_shard.rccl_env_defaults (torch not imported yet here)
    env.setdefault('NCCL_MAX_P2P_NCHANNELS', '1')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def rccl_probe_child():
    """`python bench.py --rccl-probe-child PORT`: a fresh process that creates a 1-rank 'nccl' (= RCCL) group on cuda:0 and runs
    `tile_shard.rccl_self_probe` (ranks counted by an all-reduce over the communicator; a 7-MB tile through one grouped isend +
    irecv behind queued work, stream-ordered, bitwise).  Prints one JSON line.  Started by the N = 1 bench AFTER its timed region."""
    from ciaosr_amd.tile_shard import rccl_env_defaults
    rccl_env_defaults()                          # before the first HIP call (HSA reads HSA_ENABLE_IPC_MODE_LEGACY at init)
    import torch
    import torch.distributed as dist
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    port = int(sys.argv[sys.argv.index('--rccl-probe-child') + 1])
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
    from ciaosr_amd.tile_shard import rccl_self_probe
    res = rccl_self_probe(dev)
    torch.cuda.synchronize(dev)
    print('RCCL_PROBE ' + json.dumps(res), flush=True)
    dist.destroy_process_group()


def rccl_probe(timeout_s=240):
    """Run `rccl_probe_child` as a child process (bounded; its failure is reported, never raised)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'CIAOSR_DIST_BACKEND')}
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--rccl-probe-child', str(port)], capture_output=True, text=True,
                             timeout=timeout_s, env=env, cwd=REPO)
        lines = [l for l in out.stdout.splitlines() if l.startswith('RCCL_PROBE ')]
        if out.returncode != 0 or not lines:
            return dict(ok=False, error=f'rc {out.returncode}: {out.stderr[-400:]}')
        res = json.loads(lines[-1][len('RCCL_PROBE '):])
        res['ok'] = bool(res['ranks'] == 1 and all(res[k]['tile_bitwise'] and res[k]['consumer_bitwise'] for k in ('cold', 'warm')))
        return res
    except Exception as e:      # noqa: BLE001 - the probe must never take the bench line down
        return dict(ok=False, error=repr(e)[:400])


if __name__ == '__main__':
    if '--rccl-probe-child' in sys.argv:
        rccl_probe_child()
        sys.exit(0)
    self_launch(sys.argv[1:])

import torch                                    # noqa: E402 - after self_launch on purpose
import torch.distributed as dist                # noqa: E402

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
    """Algorithmic work of ALL launches of kernel `tag` on ONE tile of Q queries / HW LR pixels (the caller multiplies by the
    tiles of a step): (amount, 'flop'|'flop16'|'byte').  FLOPs = 2 x MACs of the contraction as the reference writes it, minus
    the exact layer-1 hoist; bytes for the HBM-bound K4 = SURVEY 8(d)'s 22 064 B/query.  Kernels that reach the
    same result with fewer multiplies (Winograd, box sum) are priced on the reference's form here; `executed_ratio` gives the
    flops their MFMAs actually execute, so that a fraction above 1 never stands alone."""
    if tag.endswith('_f16x3') or tag.endswith('_bf16x3'):      # both operands as 16-bit pairs: the same algorithmic work, three MFMAs per product
        tag = tag[:tag.rindex('_')] + '_bf16'
    if tag.endswith('_f16'):               # IEEE-half kernels: the bf16 kernels' work and peak (same MFMA rate)
        tag = tag[:-4] + '_bf16'
    if tag in ('head_kv_chain_bf16', 'head_kv_chain_pairs_bf16'):   # the weights-stationary form of the 16-bit kv kernel (round 5): same work
        tag = 'head_kv_fused_bf16'
    if tag in ('head_decode_chain_bf16', 'head_decode_chain_pairs_bf16'):   # ... and of imnet_q
        tag = 'head_decode_fused_bf16'
    D, Dv, R = 9 * C, 10 * C, Q * J
    side = HW ** 0.5
    dense = sum(2.0 * HW * 9 * (C + C * l) * C for l in range(layers)) * blocks
    table = {
        # fused kernels: phi_k + phi_v layers 2..5 per (query, shift) row; phi_q all layers per query
        # (imnet_k's output layer is folded exactly into the 9-rows-per-LR-pixel logit table: 'head_logit_table')
        'head_kv_fused': (2.0 * R * (6 * hidden * hidden + hidden * Dv), 'flop'),
        'head_logit_table': (2.0 * 9 * HW * D * hidden, 'flop'),           # GEMM form of the table
        'head_logit_table_w2': (2.0 * 9 * HW * D * hidden, 'flop'),        # nine Winograd F(2x2) convolutions of the product maps
        'head_logit_table_w4': (2.0 * 9 * HW * D * hidden, 'flop'),        # ... in F(4x4) form (the fp32 / f16x3 default)
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
        'head_table_bf16': (2.0 * HW * hidden * (D + Dv), 'flop16'),
        # fp32: computed as a 3x3 diagonal box sum of a K = C/2 per-pixel correlation (csa_scores_f32.hip): the ALGORITHMIC flops are the
        # reference's patch correlation (the kernel executes ~1/3.4 of them)
        'csa_scores': (2.0 * HW * (HW / 4) * 4.5 * C, 'flop'),
        # >= 64x64 maps run the composed fold+down tail (DESIGN 'cs_attn tail'): 16C value columns instead of 36C
        'csa_attn_v': (2.0 * HW * (HW / 4) * (16 if HW >= 4096 else 36) * C, 'flop'),
        'csa_scores_bf16': (2.0 * HW * (HW / 4) * 4.5 * C, 'flop16'),
        'csa_attn_v_bf16': (2.0 * HW * (HW / 4) * 16 * C, 'flop16'),        # main 16C columns; the 9C edge variants run as csa_attn_v_edge
        'csa_attn_v_edge': (2.0 * (2 * side * 4 * C + C) * (HW / 4), 'flop'),
        'csa_down_partial': (2.0 * (side / 2 + 3) ** 2 * 9 * C * 9 * C, 'flop'),
        'csa_down': (2.0 * HW * 9 * C * C, 'flop'),
        'enc_conv3x3': (2 * 2.0 * HW * 9 * C * C, 'flop'),            # sfe2 + gff.1 (dense layers run in scatter form)
        'enc_dense_scatter': (dense, 'flop'),
        'enc_dense_bf16': (dense, 'flop16'),
        'enc_dense_gather': (dense, 'flop'),
        # the same layers in Winograd F(2x2, 3x3) form: ALGORITHMIC flops are the direct convolution's (the kernel executes 1/2.25 of them,
        # so its fraction of the fp32 MFMA peak on algorithmic work can exceed 1)
        'enc_dense_wino': (dense, 'flop'),
        'enc_dense_wino4': (dense, 'flop'),           # F(4x4, 3x3): executes 1/4 of the direct convolution's flops
        'enc_conv1x1': (blocks * 2.0 * HW * (C + C * layers) * C + 2.0 * HW * C * blocks * C, 'flop'),
        # the blocks' local feature fusion in f16 mode: memory-bound (16-bit rows in, fp32 residual in, two fp32 + one 16-bit rows out)
        'enc_conv1x1_bf16': (blocks * HW * ((C + C * layers) * 2.0 + C * 4.0 + 2 * C * 4.0 + C * 2.0), 'byte'),
        'softmax_stats': (4.0 * HW * (HW / 4), 'byte'),       # one read of the fp32 logit matrix
        'local_attention': (Q * (4.0 * J * D + 4.0 * J * Dv + 4.0 * Dv + 16) + 2.0 * C * 4 * HW, 'byte'),
        'head_rows': (R * 4.0 * 2 * hidden * 2, 'byte'),
        # K1 as the reference assembles the MLP inputs (21 936 B / query at C = 64): write inp_k, inp_v, q rows; read coords + the feature rows once
        'gather_rows': (Q * (4.0 * J * (D + 4) + 4.0 * J * (Dv + 4) + 4.0 * D + 16) + 2.0 * C * 4 * HW, 'byte'),
    }
    return table.get(tag)


def executed_ratio(tag, HW, C=64, precision='fp32', bf16_single=False):
    if tag.startswith('head_kv_chain_pairs') or tag.startswith('head_decode_chain_pairs'):
        return 2.0
    if tag.startswith('head_kv_chain') or tag.startswith('head_decode_chain'):
        return 1.0
    return _executed_ratio(tag, HW, C, precision, bf16_single)


def _executed_ratio(tag, HW, C=64, precision='fp32', bf16_single=False):
    """MFMA flops a kernel EXECUTES / its algorithmic flops (`kernel_work`).  1 unless the kernel restructures the contraction:
    Winograd F(2x2, 3x3) issues 16 multiplies per 2x2 output tile and channel pair instead of 36 (dense layers, logit table:
    dense_wino_f32.hip); the fp32 correlation scores are a 3x3 diagonal box sum of a K = C/2 per-pixel correlation whose D blocks
    of (8+2)x(16+2) query-halo x (4+2)x(16+2) key-halo pixels are computed as 6 x 4 MFMA tiles of 32x32 per 8x16 x 4x16 item
    (csa_scores_f32.hip); hi + lo weight pairs issue two MFMAs per product (bf16 pairs, f16-pairs), activation pairs three."""
    if tag.endswith('_f16x3') or tag.endswith('_bf16x3'):
        return 3.0
    base = tag[:-5] if tag.endswith('_bf16') else (tag[:-4] if tag.endswith('_f16') else tag)
    half = base != tag
    if not half:
        if base in ('enc_dense_wino', 'head_logit_table_w2'):
            return 16.0 / 36.0
        if base in ('enc_dense_wino4', 'head_logit_table_w4'):
            return 36.0 / 144.0
        if base == 'csa_scores' and HW >= 4096:
            side = HW ** 0.5
            items = -(-side // 8) * -(-side // 16) * -(-(side / 2) // 4) * -(-(side / 2) // 16)
            return items * 2.0 * 192 * 128 * (C / 2) / (2.0 * HW * (HW / 4) * 4.5 * C)
        return 1.0
    if base in ('head_kv_fused', 'head_decode_fused', 'enc_dense'):
        if precision in ('f16-pairs', 'f16x3-fast') or (precision == 'bf16' and not bf16_single):
            return 2.0
    return 1.0


_TAG2FN_16 = {'head_kv_chain': 'head_kv_chain_kernel', 'head_decode_chain': 'head_decode_chain_kernel', 'head_kv_fused': 'head_kv_fused',
              'head_decode_fused': 'head_decode_fused', 'enc_dense': 'dense_h16'}


def offline_traffic(tag, precision):
    """(bytes per launch, source) of a 16-bit kernel from profiles/r6_ (else r5_) c3tile_<precision>_pmc_hbm_traffic.json (tools/pmc_summary.py over two
    rocprofv3 --pmc passes: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), or None."""
    prec = {'f16-pairs': 'f16_pairs', 'f16x3-fast': 'f16x3_fast', 'bf16-single': 'bf16_single'}.get(precision, precision)
    path = next((q for q in (os.path.join(REPO, 'profiles', f'r{r}_c3tile_{prec}_pmc_hbm_traffic.json') for r in (6, 5)) if os.path.exists(q)), '')
    fn = next((v for k, v in _TAG2FN_16.items() if tag.startswith(k)), None)
    if not fn or not os.path.exists(path):
        return None
    hits = [v for k, v in json.load(open(path)).items() if fn in k]
    if not hits:
        return None
    n_l = sum(h['launches'] for h in hits)
    return (round(sum(h['hbm_bytes_per_launch'] * h['launches'] for h in hits) / max(n_l, 1)),
            f'offline: rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE of the same kernel on one 192x192 tile, profiles/{os.path.basename(path)}')


def roofline_object(tag, total_ms, launches, Q, HW, n_tiles, precision='fp32', bf16_single=False, C=64):
    """The `roofline` entry of one kernel tag from its HIP-event time: algorithmic work / time against the gfx950 peak, with the
    executed-work twin (`executed_frac` <= 1 is the matrix pipe's share; `frac` may exceed 1 for Winograd / box-sum kernels)."""
    work = kernel_work(tag, Q, HW, C=C)
    if not work or total_ms <= 0:
        return None
    amount, kind = work[0] * n_tiles, work[1]
    if kind == 'byte':
        ach = amount / (total_ms * 1e-3) / 1e9
        return dict(bound='hbm', achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(ach / PEAK_HBM_GBS, 4),
                    traffic=None, algorithmic_bytes_per_launch=round(amount / max(launches, 1)))
    peak = PEAK_F32_MFMA_TFLOPS if kind == 'flop' else PEAK_BF16_MFMA_TFLOPS
    ach = amount / (total_ms * 1e-3) / 1e12
    ratio = executed_ratio(tag, HW, C=C, precision=precision, bf16_single=bf16_single)
    return dict(bound='mfma', achieved=round(ach, 3), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4), traffic=None,
                algorithmic_flop_per_launch=round(amount / max(launches, 1)),
                executed_flop_per_launch=round(amount * ratio / max(launches, 1)),
                executed_frac=round(ach * ratio / peak, 4))


def _avail_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def _cores():
    return max(1, min(_avail_cores(), 16))           # small ops: more threads only add contention


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _thread_counts():
    """Thread counts the CPU baseline is swept over: 16, 32, 64, 128 and every core this process may run on ('N = all physical
    cores'), capped at what the host has; the fastest is reported.  Each leg is bounded (see cpu_baseline_c3)."""
    a = _avail_cores()
    return sorted({min(a, n) for n in (16, 32, 64, 128)} | {a})


def cpu_baseline_c2(scale=4):
    """Oracle (port of the reference, configured like it) on this host's CPU cores: the whole C2 image."""
    from oracle import ciaosr_oracle as orc          # checker / baseline only
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    model = rdn_ciaosr(dict(scale=scale, tile=192, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.0)
    params = {k[len('generator.'):]: v.detach() for k, v in model.state_dict().items()}
    lq, _ = synthetic_pair(48, 48, scale)
    torch.set_num_threads(_cores())
    times = []
    for i in range(4):                              # 1 warm-up + 3 timed, ~3 s each on 8 cores
        t0 = time.perf_counter()
        out = orc.forward_test(lq, None, None, params, scale=scale, tile=192, tile_overlap=32)
        times.append(time.perf_counter() - t0)
        if sum(times) > 45:
            break
    t = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    return dict(value=round(out.shape[-1] * out.shape[-2] / 1e6 / t, 5), unit='Mpix/s', cores=torch.get_num_threads(), cpu_model=_cpu_model(),
                cores_available=_avail_cores(), kind='port', sample=f'same workload (1 LR 48x48 -> 192x192 image), median of {len(times) - 1} '
                f'runs after 1 warm-up, {t * 1e3:.0f} ms/img, torch CPU fp32, reference-style per-chunk cs_attn')


def cpu_baseline_c3(n_tiles, out_pixels, scale=4, tile=192, eval_bsize=30000):
    """SURVEY 8(d) CPU-baseline procedure for the tiled 2K workload: the oracle (op-for-op port of the reference) is
    timed on a BOUNDED part of ONE 192x192 tile -- the RDN trunk once, plus ONE eval_bsize chunk of query_rgb exactly
    as the reference runs it (materialised unfold + cs_attn recomputed inside the chunk, ciaosr_net.py:241-246 -> :135),
    plus the same chunk with the non-local map handed in (isolates cs_attn) -- and EXTRAPOLATED to the tile
    (ceil(Q / eval_bsize) chunks) and to the n_tiles of the image.  About 10-25 s of CPU work."""
    from oracle import ciaosr_oracle as orc          # checker / baseline only
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    model = rdn_ciaosr(dict(scale=scale, tile=tile, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.0)
    params = {k[len('generator.'):]: v.detach() for k, v in model.state_dict().items()}
    lq, _ = synthetic_pair(tile, tile, scale)
    x = lq - torch.tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)
    Q = (tile * scale) ** 2
    coord = orc.make_coord((tile * scale, tile * scale)).unsqueeze(0)[:, :eval_bsize].contiguous()
    cell = orc.make_cell((tile * scale, tile * scale)).unsqueeze(0)[:, :eval_bsize].contiguous()
    n_chunks = -(-Q // eval_bsize)
    last = (Q - (n_chunks - 1) * eval_bsize) / eval_bsize
    tried, best = {}, None
    t_sweep = time.perf_counter()
    slower_in_a_row = 0
    for n_thr in _thread_counts():                                          # the bounded sample at 16 / 32 / 64 / 128 / all threads: the fastest counts
        if best is not None and time.perf_counter() - t_sweep > 150.0:      # the whole baseline leg stays within a few minutes
            tried[str(n_thr)] = 'not run: sweep budget (150 s) spent'
            continue
        if slower_in_a_row >= 2:                                            # two larger thread counts in a row were already slower: more threads
            tried[str(n_thr)] = 'not run: the two thread counts before it were slower than the best'      # only add synchronisation
            continue
        torch.set_num_threads(n_thr)
        with torch.no_grad():
            orc.encoder_features(x[..., :48, :48], params)                     # warm-up of the thread pool / allocator
            t0 = time.perf_counter()
            # the trunk decides whether this thread count goes on (60-s cap, checked between its residual dense blocks)
            feat = orc.encoder_features(x, params, deadline=t0 + 60.0) if best is not None else orc.encoder_features(x, params)
            t_enc = time.perf_counter() - t0
            if feat is None or (best is not None and t_enc > 1.5 * best[2]):
                # a thread count at which the trunk alone is this much slower is not going to be the fastest: the rest of its sample
                # (minutes at 256 threads on a 2 x 64-core host: the port's small ops drown in synchronisation) is not run
                tried[str(n_thr)] = (f'not completed: RDN trunk ' + ('> 60 s (abandoned)' if feat is None else f'{t_enc:.1f} s') +
                                     f' against {best[2]:.1f} s at {best[1]} threads')
                slower_in_a_row += 1
                continue
            t0 = time.perf_counter()
            orc.query_rgb(feat, coord, cell, params)                           # one chunk as the reference runs it
            t_chunk = time.perf_counter() - t0
            nl = torch.zeros_like(feat)                                        # timing only: values do not matter
            t0 = time.perf_counter()
            orc.query_rgb(feat, coord, cell, params, nonlocal_map=nl)          # same chunk without the cs_attn recompute
            t_head = time.perf_counter() - t0
        t_csa = max(t_chunk - t_head, 0.0)
        ref_s = t_enc + (n_chunks - 1) * t_chunk + (t_csa + last * t_head)
        tried[str(n_thr)] = round(ref_s, 2)
        slower_in_a_row = 0 if (best is None or ref_s < best[0]) else slower_in_a_row + 1
        if best is None or ref_s < best[0]:
            best = (ref_s, n_thr, t_enc, t_chunk, t_head, t_csa)
    tile_ref, n_best, t_enc, t_chunk, t_head, t_csa = best
    torch.set_num_threads(n_best)
    tile_hoist = t_enc + t_csa + (Q / eval_bsize) * t_head
    return dict(value=round(out_pixels / 1e6 / (n_tiles * tile_ref), 6), unit='Mpix/s', cores=n_best, cpu_model=_cpu_model(),
                cores_available=_avail_cores(), seconds_per_tile_by_threads=tried,
                kind='port',
                sample=f'EXTRAPOLATED from a bounded sample of one {tile}x{tile} LR tile: RDN trunk {t_enc:.2f} s + one eval_bsize={eval_bsize} '
                       f'chunk of query_rgb incl. its cs_attn recompute {t_chunk:.2f} s (cs_attn {t_csa:.2f} s, head {t_head:.2f} s) '
                       f'-> x{n_chunks} chunks = {tile_ref:.1f} s/tile -> x{n_tiles} tiles = {n_tiles * tile_ref / 3600:.2f} h/img; '
                       f'torch CPU fp32, {torch.get_num_threads()} threads',
                seconds_per_tile=round(tile_ref, 2),
                cs_attn_hoisted=dict(value=round(out_pixels / 1e6 / (n_tiles * tile_hoist), 6), unit='Mpix/s',
                                     seconds_per_tile=round(tile_hoist, 2),
                                     note='same port with cs_attn computed once per tile instead of once per chunk'))


def live_pmc_traffic(kernel_substr, unit_workload, precision, timeout_s=300):
    """HBM bytes per launch of the kernel whose name contains `kernel_substr`, collected NOW by two child processes under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only), each running this script on one
    tile / one small image.  FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request for wide coalesced reads,
    MI355X_MICROARCH.md, HBM section); units KiB.  Returns (bytes_per_launch, launches) or None if rocprofv3 is unavailable
    or fails -- the caller then falls back to the committed offline summary."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    vals = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='ciaosr_pmc_', dir='/tmp')
        cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--',
               sys.executable, os.path.join(REPO, 'bench.py'), '--workload', unit_workload, '--precision', precision, '--steps', '1',
               '--warmup', '1', '--no-cpu-baseline', '--no-extras', '--no-live-pmc', '--no-rccl-probe']
        try:
            subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), timeout=timeout_s, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            got = []
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r['Counter_Name'] == counter and kernel_substr in r['Kernel_Name']:
                        got.append(float(r['Counter_Value']))
            if not got:
                return None
            vals[counter] = (sum(got) / len(got) * 1024.0, len(got))
        except Exception:      # noqa: BLE001 - profiler missing / refused / timed out: offline fallback
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return round(2.0 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]), vals['FETCH_SIZE'][1]


STAGED_PMC = os.path.join(REPO, 'profiles', 'r6_c3tile_staged_pmc_hbm_traffic.json')
STAGED_STATS = os.path.join(REPO, 'profiles', 'r6_c3tile_staged_kernel_stats.csv')


def staged_counter_bytes(kernel_substr, waves_per_query):
    """HBM bytes per query of a one-wave-per-unit staged kernel from the committed rocprofv3 --pmc passes of `bench.py --workload c3tile
    --head-route as-written` (tools/profile_round.sh -> tools/pmc_summary.py: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, divided by the
    launch's wavefronts), or None."""
    if not os.path.exists(STAGED_PMC):
        return None
    hits = [v for k, v in json.load(open(STAGED_PMC)).items() if kernel_substr in k and 'hbm_bytes_per_wave' in v]
    return hits[0]['hbm_bytes_per_wave'] * waves_per_query if hits else None


def staged_rocprof_ms(kernel_substr):
    """(average ns, calls) of a staged kernel in the committed rocprofv3 --kernel-trace --stats summary, or None."""
    import csv
    if not os.path.exists(STAGED_STATS):
        return None
    for r in csv.DictReader(open(STAGED_STATS)):
        if kernel_substr in r['Name']:
            return float(r['AverageNs']), int(r['Calls'])
    return None


def staged_hbm_rooflines(model, tl, Q, HW, tile_lr, scale, dev):
    """K1 / K4 of the two staged routes on one tile, HIP events on the launch stream: flat scalars (k4_*, k1_*) for the driver's record plus
    the per-kernel objects.  K4 = local_attention_kernel<4> (ciaosr_net.py:203-216): Q x 22 064 B; K1 = gather_rows_kernel
    (ciaosr_net.py:176-196): Q x 21 936 B, both SURVEY 8(d).  `head_rows` is the staged C route's hoisted K1 (layer-1 table rows + the
    4-column tail instead of the 580 / 644-wide inputs; priced on its nominal bytes -- two table rows read, two hidden rows written per
    (query, sample) -- of which the reads mostly hit on-die)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd._lib import HEAD_STAGED
    out, hb = {}, {}

    def obj(tag, total_ms, key_ms='ms_per_tile'):
        wk = kernel_work(tag, Q, HW)
        gbs = wk[0] / (total_ms * 1e-3) / 1e9
        return dict(bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                    algorithmic_bytes_per_query=round(wk[0] / Q), **{key_ms: round(total_ms, 4)})
    try:
        staged = hip_ops.Options('fp32', head_route=HEAD_STAGED)
        model.restore(tl, options=staged)
        with hip_ops.profile():
            model.restore(tl, options=staged)
            torch.cuda.synchronize(dev)
        st = hip_ops.profile.results()
        for tag in ('local_attention', 'head_rows'):
            if tag in st:
                hb[tag] = obj(tag, st[tag]['total_ms'])
                hb[tag]['launches'] = st[tag]['launches']
        # the reference's op order (as-written route): K1 as SURVEY 8(d) prices it, and K4 once more behind the un-hoisted MLPs
        gen = model.generator
        x = model.normalize(tl)
        feat = hip_ops.hwc_to_nchw(gen._encoder_hip.forward_hwc(x[0], None))
        cc, cl = hip_ops.make_coord_cell(tile_lr * scale, tile_lr * scale, dev)
        gen._head.forward_as_written(feat, x[0], cc, cl, chunk=gen.eval_bsize)
        with hip_ops.profile():
            gen._head.forward_as_written(feat, x[0], cc, cl, chunk=gen.eval_bsize)
            torch.cuda.synchronize(dev)
        aw = hip_ops.profile.results()
        if 'gather_rows' in aw:
            hb['gather_rows'] = obj('gather_rows', aw['gather_rows']['total_ms'])
            hb['gather_rows']['launches'] = aw['gather_rows']['launches']
        if 'local_attention' in aw:
            hb['local_attention_as_written_route'] = obj('local_attention', aw['local_attention']['total_ms'])
        del feat, x
    except Exception as e:                  # noqa: BLE001  (an extra: never the reason a bench line is lost)
        hb['error'] = repr(e)[:300]
    # K4 with 16-bit wk / wv / z (ciaosr_local_attention_bf16, SURVEY 8(d): 11 056 B per query): the kernel alone on one staged chunk's
    # worth of synthetic operands with the indices of a real chunk (neighbouring queries share LR pixels)
    try:
        Qc = 30000
        g_ = torch.Generator(device='cpu').manual_seed(5)
        U_ = torch.randn(HW, 10 * 64, generator=g_).to(dev)
        side_ = int(round(HW ** 0.5))
        cc_, cl_ = hip_ops.make_coord_cell(side_ * 4, side_ * 4, dev)
        qi_, ki_, _ = hip_ops.head_indices(cc_[:Qc].contiguous(), cl_[:Qc].contiguous(), side_, side_, want_rel=False)
        wk_ = torch.randn(Qc * 4, 576, generator=g_).to(dev).to(torch.bfloat16)
        wv_ = torch.randn(Qc * 4, 640, generator=g_).to(dev).to(torch.bfloat16)
        for _ in range(2):
            hip_ops.local_attention_16(U_, 64, 64, qi_, ki_, wk_, wv_)
        with hip_ops.profile():
            for _ in range(5):
                hip_ops.local_attention_16(U_, 64, 64, qi_, ki_, wk_, wv_)
            torch.cuda.synchronize(dev)
        t16 = hip_ops.profile.results()['local_attention_bf16']['avg_ms']
        wkf_, wvf_ = wk_.float(), wv_.float()
        hip_ops.local_attention(U_, 64, 64, qi_, ki_, wkf_, wvf_)
        with hip_ops.profile():
            for _ in range(5):
                hip_ops.local_attention(U_, 64, 64, qi_, ki_, wkf_, wvf_)
            torch.cuda.synchronize(dev)
        t32 = hip_ops.profile.results()['local_attention']['avg_ms']
        del wkf_, wvf_
        b16 = Qc * (2.0 * 4 * 576 + 2.0 * 4 * 640 + 2.0 * 640 + 16) + 2.0 * 64 * 4 * HW
        hb['local_attention_bf16'] = dict(bound='hbm', achieved=round(b16 / (t16 * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                                          frac=round(b16 / (t16 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), ms_per_chunk=round(t16, 4),
                                          queries=Qc, bytes_per_query=round(b16 / Qc), fp32_kernel_same_chunk_ms=round(t32, 4))
        del U_, wk_, wv_
    except Exception as e:                  # noqa: BLE001
        hb['local_attention_bf16'] = dict(error=repr(e)[:200])
    # flat scalars (the driver's record keeps scalars of `roofline`): live HIP-event figures + the committed rocprof / counter evidence
    k4, k1, k1h = hb.get('local_attention'), hb.get('gather_rows'), hb.get('head_rows')
    if k4:
        out.update(k4_hbm_frac=k4['frac'], k4_hbm_gbs=k4['achieved'], k4_ms_per_tile=k4['ms_per_tile'], k4_bytes_per_query=k4['algorithmic_bytes_per_query'])
    b = staged_counter_bytes('local_attention_kernel<4>', 1)
    if b:
        out.update(k4_counter_bytes_per_query=round(b, 1), k4_counter_over_algorithmic=round(b / kernel_work('local_attention', Q, HW)[0] * Q, 3))
    r = staged_rocprof_ms('local_attention_kernel<4>')
    if r:       # the committed --stats row: average launch of one eval_bsize chunk (the tile's 20 launches: 19 x 30 000 + 19 824 queries)
        out.update(k4_rocprof_avg_launch_ms=round(r[0] / 1e6, 4),
                   k4_rocprof_hbm_frac=round(kernel_work('local_attention', Q, HW)[0] / -(-Q // 30000) / (r[0] * 1e-9) / 1e9 / PEAK_HBM_GBS, 4))
    if k1:
        out.update(k1_hbm_frac=k1['frac'], k1_hbm_gbs=k1['achieved'], k1_ms_per_tile=k1['ms_per_tile'], k1_bytes_per_query=k1['algorithmic_bytes_per_query'])
    b = staged_counter_bytes('gather_rows_kernel', 4)
    if b:
        out.update(k1_counter_bytes_per_query=round(b, 1))
    if k1h:
        out.update(k1_hoisted_hbm_frac=k1h['frac'])
    if 'frac' in hb.get('local_attention_bf16', {}):
        out.update(k4_16bit_hbm_frac=hb['local_attention_bf16']['frac'])
    out['staged_path_hbm_kernels'] = hb
    return out


def scale_model(model, ms_1gpu, n_tiles, hh, ww, dev, link_gbs=48.0):
    """What the tile-sharded C4 run should show, from THIS run's single-GPU measurements (no multi-GPU hardware on this pool: a prediction
    the driver's eventual 8-GPU SCALE run can be compared with, not a measurement).  Per image rank 0 does everything only it can do --
    every tile's blend, tile_finalize, denorm_clamp (t_rank0_only, timed here on scratch buffers) -- plus its own tiles; a peer computes
    its tiles and hands each 768 x 768 x 3 fp32 output (7.08 MB) to rank 0 over its own xGMI link under the next tile's compute, so only the
    LAST tile's transfer is exposed (7.08 MB / `link_gbs`, one direction of one link; ~153 GB/s aggregate per link pair, 48 taken as a
    conservative RCCL point-to-point rate).  predicted = max over ranks; efficiency = t(1) / (N t(N))."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.tile_shard import partition
    E = torch.zeros(3, hh, ww, device=dev)
    Wt = torch.ones_like(E)
    tile_out = torch.zeros(768 * 768, 3, device=dev)

    def extra():
        for _ in range(n_tiles):
            hip_ops.tile_blend(E, Wt, tile_out, 0, 0, 768, 768)
        hip_ops.denorm_clamp(hip_ops.tile_finalize(E, Wt), hh, ww, model.rgb_mean, model.rgb_std)
    extra()
    t_extra = time_steps(extra, 2, dev)
    del E, Wt, tile_out
    t_tile = (ms_1gpu - t_extra) / n_tiles
    tile_mb = 768 * 768 * 3 * 4 / 1e6
    t_xfer = tile_mb / 1e3 / link_gbs * 1e3
    out = dict(note='PREDICTION from the 1-GPU measurements of this run (strong scaling of the one C3 image, tile t -> rank t mod N, rank 0 blends)',
               ms_per_tile_compute=round(t_tile, 3), rank0_only_ms_per_image=round(t_extra, 3), tile_mb=round(tile_mb, 2),
               exposed_transfer_ms=round(t_xfer, 3), link_gbs_assumed=link_gbs)
    for n in (2, 4, 8):
        parts = [len(p_) for p_ in partition(n_tiles, n, 1.0)]
        t_rank = [parts[0] * t_tile + t_extra + t_xfer] + [k * t_tile + t_xfer for k in parts[1:]]
        t_n = max(t_rank)
        out[f'n{n}'] = dict(tiles_per_rank=parts, predicted_ms_per_step=round(t_n, 1), predicted_mpix_s=round(hh * ww / 1e6 / (t_n * 1e-3), 2),
                            predicted_efficiency=round(ms_1gpu / (n * t_n), 4), critical_rank=int(max(range(n), key=lambda r: t_rank[r])))
    return out


def other_configs(dev):
    """C1 and C5 of BASELINE.json at their own sizes (random-init weights of the reference's config files / ctor arguments): ms per image
    and, from one profiled pass, the per-kernel times with a roofline object for the dominant kernel the work table knows and for the
    fused head's kv kernel."""
    import ciaosr_amd
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    out = {}
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[256, 256, 256, 256])
    cases = [
        ('c1', dict(type=ciaosr_amd.LocalImplicitSREDSR, encoder=dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16),
                    imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000), 2.0, 64, ('fp32',)),
        ('c5', dict(type=ciaosr_amd.LocalImplicitSRSWINIR, window_size=8,
                    encoder=dict(type='SwinIR', upscale=4, in_chans=3, img_size=48, window_size=8, img_range=1., depths=[6, 6, 6, 6, 6, 6],
                                 embed_dim=180, num_heads=[6, 6, 6, 6, 6, 6], mlp_ratio=2, upsampler='pixelshuffle', resi_connection='1conv'),
                    imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000), 3.3, 180, ('fp32', 'f16', 'bf16')),
    ]
    for name, gen, sc, C, precisions in cases:
        try:
            m = ciaosr_amd.CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss'), rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1., 1., 1.),
                                  test_cfg=dict(scale=sc)).eval()
            seeded_init_(m, seed=0, gain=1.0)
            m = m.to(dev)
            # C5 'bf16' runs as 'bf16x3' (bf16 hi + lo weights and activations in the head: LocalImplicitSRSWINIR.effective_options); labelled below
            lq = synthetic_pair(48, 48, 4)[0].to(dev)
            ht = wt = round(48 * sc)
            coord, cell = hip_ops.make_coord_cell(ht, wt, dev)
            coord, cell = coord.unsqueeze(0), cell.unsqueeze(0)
            Qc, HWc = ht * wt, 48 * 48
            for prec in precisions:
                o = hip_ops.Options(prec)
                eff = m.generator.effective_options(o)
                eff_name = eff.precision + ('x3' if eff.f16_pairs == 2 else '')
                label = f'{name}_{prec}' if eff_name == prec else f'{name}_{prec}_runs_as_{eff_name}'
                for _ in range(3):
                    m.restore(lq, coord, cell, options=o)
                # millisecond-scale, launch-bound steps timed from the host: the best of three groups of 20 (a group of 10 picked up host
                # hiccups of several ms on some boxes)
                t = min(time_steps(lambda: m.restore(lq, coord, cell, options=o), 20, dev) for _ in range(3))
                out[f'{label}_ms'] = round(t, 4)
                out[f'{label}_mpix_s'] = round(Qc / 1e6 / (t * 1e-3), 3)
                with hip_ops.profile():
                    m.restore(lq, coord, cell, options=o)
                    torch.cuda.synchronize(dev)
                pr = hip_ops.profile.results()
                out[f'{label}_kernels_ms'] = {k: round(v['total_ms'], 4) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]['total_ms'])[:8]}
                known = [k for k in sorted(pr, key=lambda k_: -pr[k_]['total_ms']) if kernel_work(k, Qc, HWc, C=C)]
                kv = [k for k in known if k.startswith('head_kv_')]
                for what, tags in (('roofline', known[:1]), ('head_kv_roofline', kv[:1])):
                    if tags:
                        ro = roofline_object(tags[0], pr[tags[0]]['total_ms'], pr[tags[0]]['launches'], Qc, HWc, 1, eff.precision, C=C)
                        if ro:
                            ro.update(kernel=tags[0], share_of_step=round(pr[tags[0]]['total_ms'] / max(sum(v['total_ms'] for v in pr.values()), 1e-9), 3))
                            out[f'{label}_{what}'] = ro
            del m
        except Exception as e:      # noqa: BLE001 - an extra must not take the headline down
            out[f'{name}_error'] = repr(e)[:300]
    return out


WORKLOADS = {
    # name: (LR h, LR w, tiles, description)
    'c3': (1356, 2040, 117, 'C3: RDN-CiaoSR (c64b16) x4, LR 1356x2040 -> 5424x8160 (DIV2K-shaped 2K LR), tiled inference: 117 tiles of '
                            '192x192 (overlap 32), random-init weights'),
    'c3s': (339, 510, 6, 'C3 (DIV2K-val pairing): RDN-CiaoSR x4, LR 339x510 -> 1356x2040, 6 tiles of 192 (overlap 32), random-init weights'),
    'c3tile': (192, 192, 1, 'C3 unit: RDN-CiaoSR x4, one 192x192 LR tile -> 768x768, random-init weights'),
    'c2': (48, 48, 1, 'C2: RDN-CiaoSR (c64b16) x4, LR 48x48 -> 192x192, random-init weights'),
    'c2q': (48, 48, 1, 'C2 (query-sharded): RDN-CiaoSR (c64b16) x4, ONE LR 48x48 -> 192x192 image, random-init weights'),
}


def time_steps(fn, n, dev):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-live-pmc', action='store_true', help='do not spawn the two rocprofv3 --pmc child passes that measure roofline.traffic')
    ap.add_argument('--no-extras', action='store_true', help='skip the extra measurements (C2, one bf16 tile, staged K4) after the timed region')
    ap.add_argument('--no-rccl-probe', action='store_true', help='N = 1: do not start the child that creates a 1-rank RCCL communicator on this GPU '
                    '(counts `rccl_ranks`, runs one grouped self send / receive of a 7-MB tile behind queued work) after the timed region')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16', 'bf16x3', 'f16', 'f16-pairs', 'f16x3', 'f16x3-fast'],
                    help='fp32 (default, the reference\'s arithmetic): exact-fp32 MFMA everywhere; bf16: bf16 MFMA inputs (weights as hi + lo '
                         'pairs), fp32 accumulation; f16: IEEE half MFMA inputs (one MFMA per product, saturating conversions), fp32 accumulation; '
                         'f16-pairs: half activations, every weight as a half hi + lo pair (two MFMAs per product); f16x3: the fp32-tolerance '
                         'fast mode -- head weights AND activations as half pairs (three MFMAs per product), fp32 trunk and tables')
    ap.add_argument('--tile-batch', type=int, default=0, help='developer: test_cfg.tile_batch (0 = the default: 7 with the fp32 trunk, else 8)')
    ap.add_argument('--encoder-ahead', action='store_true', help='(default since round 5; kept for old command lines)')
    ap.add_argument('--no-encoder-ahead', action='store_true', help='test_cfg.encoder_ahead = False: every kernel of the step on ONE stream (the product default runs the trunk '
                    'of the next tile batch on a side stream under the heads of the current one: bitwise the same image, ~1 %% faster)')
    ap.add_argument('--bf16-single', action='store_true',
                    help='with --precision bf16: weights as ONE bf16 (one MFMA per product; fails the 0.01 dB PSNR gate) instead of the default hi + lo pairs')
    ap.add_argument('--rank0-share', default='1.0',
                    help='N > 1, tile sharding: fraction of the full rounds rank 0 (which also blends every tile and finalizes the image) '
                         'takes a tile in (tile_shard.tile_owners; it always sits out the ragged last round); a number in [0, 1], or "auto": '
                         'rank 0 times its blend + finalize + de-normalise work against a tile and every rank takes its answer')
    ap.add_argument('--head-route', default='fused', choices=['fused', 'staged', 'as-written', 'as-written-bf16', 'as-written-f16'],
                    help='one-tile workloads only (c3tile / c2), for profiling the HBM-bound staged kernels under rocprofv3: staged = the per-layer '
                         'GEMM route of the C library (head_rows K1\', local_attention K4); as-written = the reference\'s op order through the staged '
                         'entry points (gather_rows K1 -> imnet_k / imnet_v -> local_attention K4 -> imnet_q), -bf16 / -f16: with the 16-bit MLP and '
                         'K4 entry points.  The headline route is `fused`')
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS),
                    help='c3 (default; the config BASELINE.json\'s metric is quoted on): LR 1356x2040, 117 tiles; c3s: LR 339x510 (6 tiles); '
                         'c3tile: one 192x192 LR tile; c2: LR 48x48 (BASELINE configs[1]); c2q: C2 with its query range sharded over the ranks '
                         'after a broadcast of the encoder features.  --gpus N > 1 (C4): the tiles of the ONE image are sharded over the ranks '
                         '(tile t -> rank t % N), strong scaling')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world == 1:
        raise SystemExit('--gpus N > 1 with WORLD_SIZE=1 in the environment: unset WORLD_SIZE (bench.py then starts its own ranks) '
                         'or launch with torch.distributed.run --nproc-per-node N')
    assert world == args.gpus, f'WORLD_SIZE {world} != --gpus {args.gpus}'
    if world > 1 and args.workload in ('c2', 'c3tile'):
        raise SystemExit(f'--workload {args.workload} is a single-tile, single-GPU measurement (use c3 / c3s for tile sharding, c2q for query sharding)')
    from ciaosr_amd.tile_shard import rccl_env_defaults
    p2p_channels_env = rccl_env_defaults()                     # BEFORE the first HIP call: ROCr reads HSA_ENABLE_IPC_MODE_LEGACY when it
    n_dev = torch.cuda.device_count()                          # initialises (counting devices does not initialise it, set_device does)
    backend = os.environ.get('CIAOSR_DIST_BACKEND', 'nccl')   # 'gloo': rehearse the N-rank path on one GPU
    if backend == 'nccl' and world > n_dev:
        raise SystemExit(f'{world} ranks but only {n_dev} GPUs visible (RCCL needs one GPU per rank)')
    local_dev = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    rccl_ranks = 1
    p2p_channels = None
    watchdog = None
    if world > 1:
        import datetime
        from ciaosr_amd.tile_shard import StepDeadline, ensure_communicator
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        p2p_channels = p2p_channels_env               # set before the communicator exists: RCCL reads its environment at init
        # bounded waits: a step (incl. the profiled / checked ones) past the deadline ends THIS rank with exit code 3, and the
        # launcher then ends the others; collectives time out on the same scale instead of the 10 / 30-minute defaults
        deadline_s = float(os.environ.get('CIAOSR_STEP_DEADLINE_S', '900'))
        pg_timeout = datetime.timedelta(seconds=max(deadline_s, 60.0))
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
        watchdog = StepDeadline(deadline_s, what='bench step', rank=rank)
        # ranks as the communicator itself counts them: an all-reduce of ones over RCCL (also: the communicator exists before the
        # first grouped point-to-point call)
        rccl_ranks = ensure_communicator(None, dev)
        if not (rccl_ranks == world == dist.get_world_size() == args.gpus):
            # one clear line, non-zero exit: a bench line whose value was produced by fewer ranks than --gpus must never be printed
            print(f'bench.py: FATAL: --gpus {args.gpus} but the {backend} communicator counts {rccl_ranks} rank(s) '
                  f'(WORLD_SIZE {world}, torch.distributed world {dist.get_world_size()}); no result line printed', file=sys.stderr, flush=True)
            sys.exit(4)

    from ciaosr_amd import hip_ops, _lib
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.tile_shard import clip_test_distributed, predict_query_sharded
    from ciaosr_amd.coords import make_coord, make_cell
    _lib.load()
    opt = hip_ops.Options(args.precision, bf16_single=int(args.bf16_single and args.precision == 'bf16'))
    if args.head_route != 'fused':
        if world > 1 or args.workload not in ('c3tile', 'c2'):
            raise SystemExit('--head-route staged / as-written* is a one-tile, one-GPU profiling workload (--workload c3tile or c2)')
        if args.head_route == 'staged':
            from ciaosr_amd._lib import HEAD_STAGED
            opt = opt.replace(head_route=HEAD_STAGED)

    scale = 4
    lr_h, lr_w, n_tiles_img, wl_desc = WORKLOADS[args.workload]
    model = rdn_ciaosr(dict(scale=scale, tile=192, tile_overlap=32, **({'tile_batch': args.tile_batch} if args.tile_batch else {}), ))
    seeded_init_(model, seed=0, gain=1.0)             # default-init scale for timing
    model = model.to(dev)
    lq, _ = synthetic_pair(lr_h, lr_w, scale)         # identical on every rank (CPU-generated)
    lq = lq.to(dev)
    out_pixels = (lr_h * scale) * (lr_w * scale)

    if args.workload == 'c2q':
        q_coord = make_coord((lr_h * scale, lr_w * scale)).unsqueeze(0).to(dev)
        q_cell = make_cell((lr_h * scale, lr_w * scale)).unsqueeze(0).to(dev)

    tail = {}                                         # stream events of the last step (N > 1): exposed-tail figure
    share = dict(value=1.0, source='argument')

    def rank0_share_auto():
        """Rank 0's own extra work per image (n blends of a tile + tile_finalize + denorm_clamp, on scratch buffers) against one tile's
        compute time -> the fraction of full rounds it should join so that its stream ends with the peers'; broadcast to every rank."""
        val = [1.0, 0.0, 0.0]
        if rank == 0:
            hh, ww = lr_h * scale, lr_w * scale
            E = torch.zeros(3, hh, ww, device=dev)
            Wt = torch.ones_like(E)
            tl = synthetic_pair(192, 192, scale)[0].to(dev)
            model.restore(tl, options=opt)
            t_tile = time_steps(lambda: model.restore(tl, options=opt), 2, dev)
            tile_out = torch.zeros(768 * 768, 3, device=dev)

            def extra():
                for _ in range(n_tiles_img):
                    hip_ops.tile_blend(E, Wt, tile_out, 0, 0, 768, 768)
                hip_ops.denorm_clamp(hip_ops.tile_finalize(E, Wt), hh, ww, model.rgb_mean, model.rgb_std)
            extra()
            t_extra = time_steps(extra, 1, dev)
            per_rank = t_tile * n_tiles_img / world
            val = [min(max(1.0 - t_extra / per_rank, 0.0), 1.0), t_tile, t_extra]
        box = [val]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def step_as_written():
        """One tile in the reference's op order through the staged entry points (PackedHead.forward_as_written): trunk -> cs_attn -> per
        eval_bsize chunk [K1 gather_rows -> imnet_k, imnet_v -> K4 local_attention -> imnet_q -> last Linear + bilinear residual]."""
        gen = model.generator
        half = {'as-written': None, 'as-written-bf16': 'bf16', 'as-written-f16': 'f16'}[args.head_route]
        x = model.normalize(lq)
        feat = hip_ops.hwc_to_nchw(gen._encoder_hip.forward_hwc(x[0], opt))
        cc, cl = hip_ops.make_coord_cell(lr_h * scale, lr_w * scale, dev)
        rgb = gen._head.forward_as_written(feat, x[0], cc, cl, chunk=gen.eval_bsize, options=opt, half=half)
        return hip_ops.denorm_clamp(rgb.contiguous(), lr_h * scale, lr_w * scale, model.rgb_mean, model.rgb_std)

    def step():
        if watchdog is not None:
            watchdog.beat()
        if args.head_route.startswith('as-written'):
            return step_as_written()
        if world == 1:
            return model.restore(lq, options=opt)
        x = model.normalize(lq)
        if args.workload == 'c2q':
            pred = predict_query_sharded(model, x, q_coord, q_cell, rank, world, options=opt)
        else:
            pred = clip_test_distributed(model, x, rank, world, options=opt, stats=tail, rank0_share=share['value'])
        if rank == 0:
            out = hip_ops.denorm_clamp(pred[0].contiguous(), lr_h * scale, lr_w * scale, model.rgb_mean, model.rgb_std)
            tail['ready'] = torch.cuda.Event(enable_timing=True)
            tail['ready'].record(torch.cuda.current_stream(dev))
            return out
        return None

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if world > 1:
        if args.workload != 'c2q':
            if args.rank0_share == 'auto':
                v = rank0_share_auto()
                share.update(value=float(v[0]), source=f'auto: tile {v[1]:.2f} ms, rank-0 blends + finalize + de-normalise {v[2]:.2f} ms per image')
            else:
                share.update(value=float(args.rank0_share))
        # correctness of the sharded path before timing it: rank 0's blended image must be bitwise equal to the
        # single-process clip_test of the same image on this GPU (this step also sets up every peer connection of the
        # communicator: nothing is connected lazily inside the timed region)
        out = step()
        if rank == 0:
            watchdog.beat()
            ref = model.restore(lq, options=opt)[0]
            assert torch.equal(out, ref), f'tile-sharded output differs from 1-GPU output by {(out - ref).abs().max().item()}'
            del ref
        del out

    # warm-up; the last warm-up step is fully profiled (per-kernel HIP events) to find the dominant kernel.  The weights are packed
    # first (Restorer.prepare: once per model, what the first call would do), so that the profiled step lists a steady-state step's
    # kernels and not the 6900 one-time pack_fragments launches of the Winograd weight sets
    # Per-kernel event timings need the kernels one after the other on one stream: the product default `test_cfg.encoder_ahead` (the next tile
    # batch's trunk on a side stream under the current batch's heads; bitwise the same image) is switched off for the two profiled passes
    # -- the fully profiled warm-up step here and the dominant kernel's pass behind the timed region -- and ON (unless --no-encoder-ahead)
    # for the timed region itself.
    t_prep = time.perf_counter()
    model.prepare(opt)
    torch.cuda.synchronize(dev)
    prepare_s = time.perf_counter() - t_prep
    ahead = not args.no_encoder_ahead
    model.test_cfg['encoder_ahead'] = ahead
    for _ in range(max(args.warmup - 1, 0)):
        step()
    model.test_cfg['encoder_ahead'] = False
    with hip_ops.profile():
        step()
        torch.cuda.synchronize(dev)
    model.test_cfg['encoder_ahead'] = ahead
    if args.warmup > 0:
        step()                                      # the side stream's scratch and events exist before the timed region
    prof_all = hip_ops.profile.results()
    dominant = max(prof_all, key=lambda k: prof_all[k]['total_ms']) if prof_all else None
    step_ms_est = sum(v['total_ms'] for v in prof_all.values()) or 1.0

    # timed region.  The dominant kernel carries HIP-event pairs inside it only when that is free: a pair costs ~4 us on
    # the GPU timeline, so it is bracketed live when its launches add < 0.3 % to the step (C3: ~1000 launches of 2.4 ms),
    # otherwise (C2: 128 dense-block launches of ~10 us) its duration is measured in a separate pass right after.
    lib = _lib.load()
    two_streams = world == 1 and ahead and n_tiles_img > model.tile_batch(opt)     # the timed region overlaps two streams: no live event timing
    live = bool(dominant) and prof_all[dominant]['launches'] * 0.004 < 0.003 * step_ms_est and not two_streams
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
        prof_steps = min(args.steps, 3 if two_streams else 10)
        model.test_cfg['encoder_ahead'] = False
        lib.ciaosr_prof_reset()
        lib.ciaosr_prof_enable(1)
        for _ in range(prof_steps):
            step()
        sync()
        lib.ciaosr_prof_enable(0)
        model.test_cfg['encoder_ahead'] = ahead
    prof_dom = hip_ops.profile.results()
    lib.ciaosr_prof_filter(None)

    rank_ms = None
    if world > 1:
        # every rank's own wall time of the timed region (load balance), then the contract's MAX over ranks
        tt = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        tile_lr = 48 if args.workload in ('c2', 'c2q') else 192
        n_tiles = n_tiles_img
        if world > 1 and args.workload != 'c2q':
            from ciaosr_amd.tile_shard import partition
            n_tiles = max(len(partition(n_tiles_img, world, share['value'])[0]), 1)      # rank 0's tiles (its kernels are timed)
        Q, HW = (tile_lr * scale) ** 2, tile_lr * tile_lr
        roof = None
        if dominant and dominant in prof_dom:
            avg_ms = prof_dom[dominant]['avg_ms']
            launches_per_step = prof_dom[dominant]['launches'] / prof_steps
            step_ms = prof_dom[dominant]['total_ms'] / prof_steps      # all launches of the tag in one step
            roof = roofline_object(dominant, step_ms, launches_per_step, Q, HW, n_tiles, args.precision, args.bf16_single)
            if roof:
                roof.update(kernel=dominant, timing='HIP events on the launch stream inside the timed region' if live else
                            (f'HIP events in a separate pass of {prof_steps} steps with test_cfg.encoder_ahead = False (the timed region overlaps the next '
                             f'tile batch\'s trunk with the heads on a second stream: events there would time two kernels at once)' if two_streams else
                             f'HIP events in a separate pass of {prof_steps} steps (too many short launches per step to bracket live)'),
                            avg_launch_ms=round(avg_ms, 5),
                            launches=prof_dom[dominant]['launches'],
                            share_of_step=round(prof_all[dominant]['total_ms'] /
                                                max(sum(v['total_ms'] for v in prof_all.values()), 1e-9), 3))
                # the other kernels of the step against their own rooflines (one fully profiled step); `frac` is priced on the
                # reference's form of the contraction, `executed_frac` on the flops the kernel's MFMAs execute
                others = {}
                chained_kv = any(t.startswith('head_kv_chain') for t in prof_all)
                for tag, pr in prof_all.items():
                    if chained_kv and tag.startswith('head_kv_fused'):
                        # the 128-row kernel launched behind the chained one, gated on its fallback flag: it returned at once (no work to price)
                        roof['gated_fallback_ms_per_step'] = round(pr['total_ms'], 4)
                        continue
                    ro = roofline_object(tag, pr['total_ms'], pr['launches'], Q, HW, n_tiles, args.precision, args.bf16_single)
                    if tag == dominant or not ro or pr['total_ms'] < 0.02:
                        continue
                    others[tag] = dict(bound=ro['bound'], ms_per_step=round(pr['total_ms'], 4), frac=ro['frac'])
                    if 'executed_frac' in ro:
                        others[tag]['executed_frac'] = ro['executed_frac']
                roof['other_kernels'] = others
                # HBM bytes per launch: OFFLINE evidence from the committed rocprofv3 --pmc passes of this round (separate
                # passes, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, tools/pmc_summary.py); null when the summary is absent
                tag2fn = {'enc_dense_scatter': 'dense_scatter_small_kernel', 'enc_rdb_fused': 'rdb_fused_kernel',
                          'head_kv_fused': 'head_kv_fused_kernel', 'head_fused': 'head_fused_kernel',
                          'head_decode_fused': 'head_decode_fused_kernel', 'head_kv_fused_bf16': 'head_kv_fused_h16_kernel',
                          'head_kv_fused_f16': 'head_kv_fused_h16_kernel', 'head_kv_chain_f16': 'head_kv_chain_kernel',
                          'head_kv_chain_pairs_f16': 'head_kv_chain_kernel', 'head_kv_chain_pairs_bf16': 'head_kv_chain_kernel', 'enc_dense_bf16': 'dense_h16_kernel', 'enc_dense_f16': 'dense_h16_kernel',
                          'enc_dense_gather': 'dense_f32_kernel', 'enc_dense_wino': 'dense_wino_f32_kernel', 'enc_dense_wino4': 'dense_wino4_f32_kernel'}
                unit = 'c2' if tile_lr == 48 else 'c3tile'
                pmc_path = next((q for q in (os.path.join(REPO, 'profiles', f'r{r}_{unit}_pmc_hbm_traffic.json') for r in (6, 5, 4, 3))
                                 if os.path.exists(q)), '')
                live_traffic = None
                if world == 1 and not args.no_live_pmc and dominant in tag2fn:
                    # LIVE: two rocprofv3 --pmc child passes of this script on one tile, now, on this box
                    live_traffic = live_pmc_traffic(tag2fn[dominant], unit, args.precision)
                if live_traffic:
                    roof['traffic'] = live_traffic[0]
                    roof['traffic_source'] = (f'live: bytes per launch of {tag2fn[dominant]} from two rocprofv3 --pmc child passes (FETCH_SIZE x2 '
                                              f'gfx950 correction + WRITE_SIZE, KiB units) of this script on one {tile_lr}x{tile_lr} tile, '
                                              f'{live_traffic[1]} launches')
                elif args.precision != 'fp32' and offline_traffic(dominant, args.precision):
                    roof['traffic'], roof['traffic_source'] = offline_traffic(dominant, args.precision)
                elif args.precision == 'fp32' and os.path.exists(pmc_path) and dominant in tag2fn:
                    hits = [v for k, v in json.load(open(pmc_path)).items() if tag2fn[dominant] in k]
                    if hits:
                        n_l = sum(h['launches'] for h in hits)
                        roof['traffic'] = round(sum(h['hbm_bytes_per_launch'] * h['launches'] for h in hits) / max(n_l, 1))
                        roof['traffic_source'] = (f'offline: bytes per launch from rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE of the same kernel on '
                                                  f'one {tile_lr}x{tile_lr} tile, profiles/{os.path.basename(pmc_path)} (not re-collected by this run)')
        line = {
            'metric': 'HR Mpix/s (RDN-CiaoSR x4, LocalImplicitSR forward_test)',
            'value': round(out_pixels / 1e6 / (elapsed / args.steps), 4),
            'unit': 'Mpix/s', 'n_gpus': world, 'rccl_ranks': rccl_ranks, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 4), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'fp32' else (
                'f16 (IEEE half) MFMA inputs, saturating conversions, fp32 accumulate, in the head, the dense layers and the cs_attn contractions'
                + ('; weights as half hi+lo pairs' if args.precision == 'f16-pairs' else '')
                + ('; head weights and activations as half hi+lo pairs (three MFMAs per product), fp32 trunk' if args.precision == 'f16x3' else '')
                + ('; head weights and activations as half hi+lo pairs (three MFMAs per product), trunk with half weight pairs' if args.precision == 'f16x3-fast' else '')
                if args.precision.startswith('f16') else
                ('bf16 MFMA inputs (fp32 accumulate): head weights and activations as bf16 hi+lo pairs (three MFMAs per product), fp32 trunk, bf16 cs_attn contractions'
                 if args.precision == 'bf16x3' else
                 'bf16 MFMA inputs (fp32 accumulate) in the head, the dense layers and the cs_attn contractions; weights as '
                 + ('single bf16' if args.bf16_single else 'bf16 hi+lo pairs'))),
            'data': 'synthetic',
            'config': {'workload': wl_desc + (', fp32' if args.precision == 'fp32' else f', {args.precision} mode')
                       + ('' if world == 1 else (f'; encoder on rank 0, RCCL broadcast of the feature map, query range sharded over {world} GPUs, '
                                                 'RCCL gather of the RGB slices' if args.workload == 'c2q' else
                                                 f' (C4); tiles of the one image sharded over {world} GPUs (tile t -> rank t % {world}; rank 0, which also '
                                                 'blends, sits out the ragged last round), grouped RCCL point-to-point delivery of the output tiles to '
                                                 'rank 0 under compute, rank-0 blend in the reference order')),
                       'lr': [lr_h, lr_w], 'scale': scale, 'tiles': n_tiles_img, 'queries_per_step': out_pixels,
                       'parallelism': (f'query-shard x{world}' if args.workload == 'c2q' else f'tile-shard x{world}')},
            'roofline': roof,
            'prepare_s': round(prepare_s, 3),        # model.prepare(): one-time weight packing (not in the timed region)
            'encoder_ahead': bool(two_streams),
            'kernels_ms_per_step': {k: round(v['total_ms'], 4) for k, v in sorted(
                prof_all.items(), key=lambda kv: -kv[1]['total_ms'])},
        }
        if world > 1:
            from ciaosr_amd.tile_shard import partition
            line['rank_ms_per_step'] = rank_ms
            line['p2p_channels'] = p2p_channels
            line['step_deadline_s'] = watchdog.seconds
            if args.workload != 'c2q':
                line['tiles_per_rank'] = [len(p) for p in partition(n_tiles_img, world, share['value'])]
                line['rank0_share'] = dict(share)
                if 'last_own_tile' in tail and 'ready' in tail:
                    # GPU-timeline time on rank 0 from "its last own tile's kernels are done" to "the image is ready": the last
                    # round's receives, the remaining blends, tile_finalize and denorm_clamp -- the part of the exchange that does
                    # NOT hide under compute (last timed step)
                    line['exposed_tail_ms'] = round(tail['last_own_tile'].elapsed_time(tail['ready']), 3)
        if world == 1 and not args.no_extras and roof is not None:
            extras = {}
            # (1) the HBM-bound kernels of the staged routes (north star: >= 40 % of the HBM roofline on the local-attention kernel K4;
            # K1 gather-rows and K4 are the HBM-bound stages) on ONE tile of this workload right after the timed region
            tl = synthetic_pair(tile_lr, tile_lr, scale)[0].to(dev)
            roof.update(staged_hbm_rooflines(model, tl, Q, HW, tile_lr, scale, dev))
            # (2) the other precision / the other single-tile config, for the record (not the headline); every 16-bit figure carries
            # the roofline object of ITS dominant kernel (one profiled pass of the same input)
            def mode_roofline(inp, o, precision, n_t):
                keep = model.test_cfg.get('encoder_ahead', True)
                model.test_cfg['encoder_ahead'] = False               # one stream: event timings of single kernels
                with hip_ops.profile():
                    model.restore(inp, options=o)
                    torch.cuda.synchronize(dev)
                model.test_cfg['encoder_ahead'] = keep
                pr = hip_ops.profile.results()
                if not pr:
                    return None
                dom = max(pr, key=lambda k_: pr[k_]['total_ms'])
                ro = roofline_object(dom, pr[dom]['total_ms'], pr[dom]['launches'], Q, HW, n_t, precision)
                if ro:      # HBM bytes per launch of that kernel: offline, from the round's committed --pmc passes of this precision at the C3 tile
                    tr = offline_traffic(dom, precision)
                    if tr:
                        ro['traffic'], ro['traffic_source'] = tr
                if ro:
                    ro.update(kernel=dom, avg_launch_ms=round(pr[dom]['avg_ms'], 5), launches=pr[dom]['launches'],
                              share_of_step=round(pr[dom]['total_ms'] / max(sum(v['total_ms'] for v in pr.values()), 1e-9), 3),
                              timing='HIP events on the launch stream, one profiled pass after the timed one')
                return ro

            if args.workload in ('c3', 'c3s', 'c3tile'):
                modes = [('bf16', hip_ops.Options('bf16'), 'bf16'), ('bf16_single', hip_ops.Options('bf16-single'), 'bf16-single'),
                         ('f16', hip_ops.Options('f16'), 'f16'), ('f16_pairs', hip_ops.Options('f16-pairs'), 'f16-pairs')]
                if 'f16x3' in hip_ops.PRECISIONS:
                    modes.append(('f16x3', hip_ops.Options('f16x3'), 'f16x3'))
                    modes.append(('f16x3_fast', hip_ops.Options('f16x3-fast'), 'f16x3-fast'))
                if 'bf16x3' in hip_ops.PRECISIONS:
                    modes.append(('bf16x3', hip_ops.Options('bf16x3'), 'bf16x3'))
                o16, oh = modes[0][1], modes[2][1]
                for nm, o, prec in modes:
                    model.restore(tl, options=o)
                    extras[f'c3_tile_{nm}_mode_ms'] = round(time_steps(lambda: model.restore(tl, options=o), 3, dev), 3)
                    if args.workload != 'c3':
                        extras[f'c3_tile_{nm}_mode_roofline'] = mode_roofline(tl, o, prec, 1)
                extras['c3_tile_fp32_ms'] = round(time_steps(lambda: model.restore(tl), 3, dev), 3)
                if args.workload == 'c3':        # the whole C3 image in the opt-in 16-bit modes (PSNR-gated extensions; not the headline)
                    for nm, o, prec in modes:
                        model.restore(lq, options=o)
                        t_ = time_steps(lambda: model.restore(lq, options=o), 1, dev)
                        extras[f'c3_{nm}_mode_ms'] = round(t_, 1)
                        extras[f'c3_{nm}_mode_mpix_s'] = round(out_pixels / 1e6 / (t_ * 1e-3), 2)
                        extras[f'c3_{nm}_mode_roofline'] = mode_roofline(lq, o, prec, n_tiles_img)
                    # opt-in test_cfg.encoder_ahead: the trunk of tile batch k + 1 on a side stream under the heads of batch k (bitwise the
                    # same image).  Not the headline: per-kernel event timings overlap while two streams share the chip.
                    # the one-stream form of the step (test_cfg.encoder_ahead = False), for the record
                    model.test_cfg['encoder_ahead'] = False
                    for nm, o in (('fp32', hip_ops.DEFAULT_OPTIONS), ('f16', oh)):
                        model.restore(lq, options=o)
                        t_ = time_steps(lambda: model.restore(lq, options=o), 1, dev)
                        extras[f'c3_{nm}_one_stream_ms'] = round(t_, 1)
                        extras[f'c3_{nm}_one_stream_mpix_s'] = round(out_pixels / 1e6 / (t_ * 1e-3), 2)
                    model.test_cfg['encoder_ahead'] = ahead
                # (3) the other BASELINE configs, every round: C1 (EDSR-CiaoSR x2, LR 48x48 -> 96x96) and C5 (SwinIR-CiaoSR x3.3, LR 48x48 ->
                # 158x158; its "bf16" = the precision the generator really runs, see `effective_precision`), each with the roofline object of its
                # dominant known kernel and of the fused head's kv kernel at C = 180
                extras.update(other_configs(dev))
                c2 = synthetic_pair(48, 48, scale)[0].to(dev)
                for _ in range(3):
                    model.restore(c2)
                t_c2 = time_steps(lambda: model.restore(c2), 20, dev)
                extras['c2_fp32_ms'] = round(t_c2, 4)
                extras['c2_fp32_mpix_s'] = round(192 * 192 / 1e6 / (t_c2 * 1e-3), 3)
                # opt-in tile streams (test_cfg.tile_streams = 2: consecutive tiles on two HIP streams, bitwise the same image),
                # on the 6-tile C3 variant; not used for the headline because it invalidates per-kernel event timing
                c3s = synthetic_pair(339, 510, scale)[0].to(dev)
                for ns in (1, 2):
                    model.test_cfg['tile_streams'] = ns
                    model.restore(c3s)
                    extras[f'c3s_6tiles_fp32_ms_tile_streams_{ns}'] = round(time_steps(lambda: model.restore(c3s), 2, dev), 2)
                model.test_cfg['tile_streams'] = 1
            line['extras'] = extras
        if world == 1 and not args.no_rccl_probe:
            # N = 1: `rccl_ranks` as a communicator counts them -- a fresh child creates a 1-rank RCCL group on this GPU and hands a tile to
            # itself through the exchange code (tile_shard.rccl_self_probe); None when RCCL could not be brought up
            probe = rccl_probe()
            line['rccl_probe'] = probe
            line['rccl_ranks'] = probe.get('ranks') if probe.get('ok') else None
        if world == 1 and not args.no_cpu_baseline:
            if args.workload in ('c2', 'c2q'):
                line['cpu_baseline'] = cpu_baseline_c2(scale)
            else:
                line['cpu_baseline'] = cpu_baseline_c3(n_tiles_img, out_pixels, scale)
        if world == 1 and args.workload == 'c3' and args.head_route == 'fused':
            line['scale_model'] = scale_model(model, ms, n_tiles_img, lr_h * scale, lr_w * scale, dev)
        if 'extras' in line and 'c3_f16_mode_mpix_s' in line['extras']:
            # the opt-in modes of the same workload at a glance (whole C3 image, one timed step each; PSNR-gated, see README): flat, up front
            ex = line['extras']
            line['modes_mpix_s'] = {'fp32 (headline)': line['value'],
                                    **{nm: ex[f'c3_{key}_mode_mpix_s'] for nm, key in (('f16x3 (fp32 tolerance)', 'f16x3'), ('bf16x3 (fp32 tolerance)', 'bf16x3'), ('f16x3-fast', 'f16x3_fast'), ('f16', 'f16'),
                                                                                          ('bf16-single', 'bf16_single'), ('f16-pairs', 'f16_pairs'), ('bf16 (weight pairs)', 'bf16'))
                                       if f'c3_{key}_mode_mpix_s' in ex}}
        # key order of the printed line: the contract's keys, the flat roofline scalars, the CPU baseline and the scaling model FIRST; the bulky
        # per-kernel tables and extras last (a consumer that keeps only the head or only the tail of a > 16-KB line still gets the headline)
        if line.get('roofline'):
            ro = line['roofline']
            line['roofline'] = {**{k: v for k, v in ro.items() if not isinstance(v, (dict, list))},
                                **{k: v for k, v in ro.items() if isinstance(v, (dict, list))}}
        tail_keys = ('kernels_ms_per_step', 'rccl_probe', 'extras')
        line = {**{k: v for k, v in line.items() if k not in tail_keys and k != 'roofline'},
                **({'roofline': line['roofline']} if 'roofline' in line else {}),
                **{k: line[k] for k in tail_keys if k in line}}
        print(json.dumps(line), flush=True)
    if world > 1:
        watchdog.beat()
        dist.barrier()
        watchdog.stop()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
