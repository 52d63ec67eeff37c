#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (build container only).

Imports the unmodified hot-path files from /root/reference through tools/ref_stubs.py, runs
them on seeded inputs/weights (CPU fp32) and stores inputs + expected outputs as small
fixtures.  The reference's Python never travels to the GPU box -- only these vectors do.
Re-run:  python tools/make_golden.py [--only NAME]
Weights are produced by ciaosr_amd.init_utils.seeded_init_ (name-keyed generators); each
fixture records the sha256 of the weights so RNG drift is detected instead of mis-compared.
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools import ref_stubs                                   # noqa: E402
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair, trained_like_  # noqa: E402
from ciaosr_amd.coords import make_coord, make_cell           # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
SQRT6 = math.sqrt(6.0)


def mlp_cfg(hidden, act=None):
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=list(hidden), **({'act': act} if act else {}))
    return mk(4, 3), mk(64, 64), mk(64, 64)


def edsr_generator(ref, mid=64, blocks=16, hidden=(256,) * 4, eval_bsize=30000, act=None, **kw):
    q, k, v = mlp_cfg(hidden, act)
    enc = dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=mid, num_blocks=blocks)
    return ref.LocalImplicitSREDSR(enc, q, k, v, eval_bsize=eval_bsize, **kw)


def rdn_generator(ref, hidden=(256,) * 4, eval_bsize=30000):
    q, k, v = mlp_cfg(hidden)
    enc = dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
               upscale_factor=4, num_layers=8, channel_growth=64)
    return ref.LocalImplicitSRRDN(enc, q, k, v, eval_bsize=eval_bsize)


def head_sd(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()
            if k.startswith('imnet_') or k.startswith('cs_attn')}


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = v
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **conv)
    print(f'  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)')


def coords_for(h, w, scale):
    ht, wt = round(h * scale), round(w * scale)
    return make_coord((ht, wt)).unsqueeze(0), make_cell((ht, wt)).unsqueeze(0), (ht, wt)


def randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


# ---------------------------------------------------------------------------------------------
def gen_tiny_head(ref):
    """C=8, hidden [32,32], LR 7x9 -> 19x24 (x2.7): odd H/W (reflect pad + crop), all weights stored."""
    m = edsr_generator(ref, mid=8, blocks=1, hidden=(32, 32), eval_bsize=200).eval()
    sha = seeded_init_(m, seed=11, gain=1.0, head_gain=SQRT6)
    feat = randn((1, 8, 7, 9), 21)
    coord, cell, (ht, wt) = coords_for(7, 9, 2.7)
    with torch.no_grad():
        nl = m.cs_attn(feat)
        out = m.batched_predict([feat], coord, cell)
    w = {('w.' + k): v for k, v in head_sd(m).items()}
    save('tiny_head_s2p7', feature=feat, coord=coord, cell=cell, out=out, nonlocal_map=nl,
         target=np.array([ht, wt]), sha=np.array(sha), eval_bsize=np.array(200), **w)
    return out


def gen_tiny_head_variants(ref):
    """local_size 3 / 1, softmax_scale 2, no non-local: branch coverage of query_rgb."""
    for tag, kw in (('ls3', dict(local_size=3)), ('ls1', dict(local_size=1)),
                    ('nonl0', dict(non_local_attn=False)), ('sm2', dict(softmax_scale=2)),
                    ('nounfold', dict(feat_unfold=False, non_local_attn=False))):
        m = edsr_generator(ref, mid=8, blocks=1, hidden=(32, 32), eval_bsize=None, **kw).eval()
        sha = seeded_init_(m, seed=12, gain=1.0, head_gain=SQRT6)
        feat = randn((1, 8, 6, 8), 22)
        coord, cell, (ht, wt) = coords_for(6, 8, 2.5)
        with torch.no_grad():
            out = m.query_rgb([feat], coord, cell)
        w = {('w.' + k): v for k, v in head_sd(m).items()}
        save(f'tiny_head_{tag}', feature=feat, coord=coord, cell=cell, out=out, sha=np.array(sha),
             target=np.array([ht, wt]), **w)


def gen_tiny_head_act(ref):
    """MLPRefiner(act='sin' | 'cos') (mlp_refiner.py:81-86) in all three implicit functions: C = 8, hidden [32,32],
    LR 6x8 -> x2.5.  The reference resolves type='MLPRefiner' through mmedit's registry; the stub maps it to the in-repo
    class, which is the one carrying the `act` argument."""
    for act in ('sin', 'cos'):
        m = edsr_generator(ref, mid=8, blocks=1, hidden=(32, 32), eval_bsize=None, act=act).eval()
        sha = seeded_init_(m, seed=13, gain=1.0, head_gain=SQRT6)
        feat = randn((1, 8, 6, 8), 23)
        coord, cell, (ht, wt) = coords_for(6, 8, 2.5)
        with torch.no_grad():
            out = m.query_rgb([feat], coord, cell)
        w = {('w.' + k): v for k, v in head_sd(m).items()}
        save(f'tiny_head_act_{act}', feature=feat, coord=coord, cell=cell, out=out, sha=np.array(sha),
             target=np.array([ht, wt]), **w)


def gen_head_c64(ref):
    """Full-size dims (576/580/644/640), LR 48x48 x4: Q = 36864."""
    m = edsr_generator(ref, mid=64, blocks=1).eval()
    sha = seeded_init_(m, seed=0, gain=1.0, head_gain=SQRT6)
    feat = randn((1, 64, 48, 48), 7)
    coord, cell, (ht, wt) = coords_for(48, 48, 4)
    with torch.no_grad():
        nl = m.cs_attn(feat)
        out = m.batched_predict([feat], coord, cell)
    print('  head_c64_x4: per-query std', out.std(dim=1).flatten().tolist())
    save('head_c64_x4', out=out[0], nonlocal_map=nl[0], sha=np.array(sha), feat_seed=np.array(7),
         weight_seed=np.array(0), target=np.array([ht, wt]))


def gen_k4_c64(ref):
    """Full-width K1 / K4 reference vectors: the model and feature map of `head_c64_x4` (weight seed 0, feature seed 7; nothing but the
    seeds is needed to rebuild them), 256 queries sampled from its 36 864; the REFERENCE's own intermediates of one query_rgb call on them,
    taken with module hooks (no reference code changed): inp_k / inp_v = the arguments of imnet_k / imnet_v (ciaosr_net.py:195-202; first 64
    queries only), wk / wv = their outputs (:202, :205), z = the argument of imnet_q (:211-221: softmax(q . (key * wk)) @ (value * wv))."""
    m = edsr_generator(ref, mid=64, blocks=1).eval()
    sha = seeded_init_(m, seed=0, gain=1.0, head_gain=SQRT6)
    feat = randn((1, 64, 48, 48), 7)
    coord, cell, (ht, wt) = coords_for(48, 48, 4)
    idx = torch.randperm(ht * wt, generator=torch.Generator().manual_seed(41))[:256].sort().values
    # queries on the image border included (clamped shifts): force the four corners and two edge mid-points in
    idx[:6] = torch.tensor([0, wt - 1, (ht - 1) * wt, ht * wt - 1, wt // 2, (ht // 2) * wt])
    idx = idx.sort().values
    cap = dict(inp_k=[], inp_v=[], wk=[], wv=[], z=[])
    def tap(inp_key, out_key):
        def hook(mod, a, o):                              # returns None: the module's output is left alone
            cap[inp_key].append(a[0].detach().clone())
            cap[out_key].append(o.detach().clone())
        return hook

    def tap_q(mod, a):
        cap['z'].append(a[0].detach().clone())
    hooks = [m.imnet_k.register_forward_hook(tap('inp_k', 'wk')), m.imnet_v.register_forward_hook(tap('inp_v', 'wv')),
             m.imnet_q.register_forward_pre_hook(tap_q)]
    with torch.no_grad():
        out = m.query_rgb([feat], coord[:, idx], cell[:, idx])
    for h in hooks:
        h.remove()
    assert len(cap['wk']) == 4 and len(cap['wv']) == 4 and len(cap['z']) == 1
    stack = lambda lst: torch.stack(lst, dim=1)           # four shifts -> [256, 4, width], (query, sample) rows like the C ABI's
    save('k4_c64_x4', idx=idx, wk=stack(cap['wk']), wv=stack(cap['wv']), z=cap['z'][0], out=out[0],
         inp_k=stack(cap['inp_k'])[:64], inp_v=stack(cap['inp_v'])[:64], sha=np.array(sha), feat_seed=np.array(7),
         weight_seed=np.array(0), target=np.array([ht, wt]))


def gen_head_c64_x3p3(ref):
    """Non-integer scale with rounding ties: LR 48x48 -> 158x158."""
    m = edsr_generator(ref, mid=64, blocks=1).eval()
    sha = seeded_init_(m, seed=0, gain=1.0, head_gain=SQRT6)
    feat = randn((1, 64, 48, 48), 7)
    coord, cell, (ht, wt) = coords_for(48, 48, 3.3)
    with torch.no_grad():
        out = m.batched_predict([feat], coord, cell)
    save('head_c64_x3p3', out=out[0], sha=np.array(sha), feat_seed=np.array(7), weight_seed=np.array(0),
         target=np.array([ht, wt]))


def gen_nearest_idx(ref):
    """Per-axis nearest-index tables obtained from F.grid_sample on an index-valued map, for the
    query coordinate and both shifted/clamped key coordinates (ciaosr_net.py:145,162-183)."""
    out = {}
    for (n, nt) in ((48, 158), (192, 634), (45, 148), (192, 768), (48, 192), (7, 19), (9, 24), (100, 231)):
        ramp = torch.arange(n, dtype=torch.float32).view(1, 1, n, 1).expand(1, 1, n, 3).contiguous()
        seq = make_coord((nt, 3))[:, 0].view(nt, 3)[:, 0]            # per-axis coordinate sequence
        cell0 = torch.tensor(2.0 / nt)

        def sample(c):
            grid = torch.stack([torch.zeros_like(c), c], -1).view(1, 1, -1, 2)   # (x=0, y=c)
            return F.grid_sample(ramp, grid, mode='nearest', align_corners=False).flatten().long()

        t = (n - 1) / (1 - cell0)
        r = 1 / t
        minus = (seq + (-1.0 * r + 1e-6)).clamp(-1 + 1e-6, 1 - 1e-6)
        plus = (seq + (1.0 * r + 1e-6)).clamp(-1 + 1e-6, 1 - 1e-6)
        out[f'q_{n}_{nt}'] = sample(seq).numpy().astype(np.int32)
        out[f'km_{n}_{nt}'] = sample(minus).numpy().astype(np.int32)
        out[f'kp_{n}_{nt}'] = sample(plus).numpy().astype(np.int32)
    save('nearest_idx', **out)


def gen_csattn(ref):
    for tag, (h, w) in (('48', (48, 48)), ('45x51', (45, 51))):
        att = ref.CrossScaleAttention(channel=64, scale=[2]).eval()
        sha = seeded_init_(att, seed=3, gain=1.5)
        x = randn((1, 64, h, w), 31)
        with torch.no_grad():
            y = att(x)
        print(f'  csattn_{tag}: out std {y.std():.4f}')
        save(f'csattn_c64_{tag}', out=y[0], sha=np.array(sha), in_seed=np.array(31), weight_seed=np.array(3),
             gain=np.array(1.5), shape=np.array([h, w]))
    att = ref.CrossScaleAttention(channel=8, scale=[2]).eval()
    seeded_init_(att, seed=4, gain=1.5)
    for tag, (h, w) in (('10x12', (10, 12)), ('9x11', (9, 11)), ('7x8', (7, 8))):
        x = randn((2, 8, h, w), 32)
        with torch.no_grad():
            y = att(x)
        w_ = {('w.cs_attn.' + k): v for k, v in att.state_dict().items()}
        save(f'csattn_c8_{tag}', x=x, out=y, **w_)


def gen_head_c180(ref):
    """SwinIR-width head (1620/1624/1804/1800): encoder replaced by a 180-channel EDSR stand-in
    because only the feature width matters to the head."""
    m = edsr_generator(ref, mid=180, blocks=1).eval()
    sha = seeded_init_(m, seed=5, gain=1.0, head_gain=SQRT6)
    feat = randn((1, 180, 24, 24), 8)
    coord, cell, (ht, wt) = coords_for(24, 24, 3.3)
    with torch.no_grad():
        out = m.batched_predict([feat], coord, cell)
    save('head_c180_x3p3', out=out[0], sha=np.array(sha), feat_seed=np.array(8), weight_seed=np.array(5),
         target=np.array([ht, wt]))


def gen_head_c180_stress(ref):
    """The SwinIR-width head (C = 180: 1620 / 1624 / 1804 / 1800) on a feature map with TRAINED-LIKE statistics -- what BASELINE config 5's
    16-bit modes must survive and the Gaussian `head_c180_x3p3` does not show: per-channel log-normal scales (std 0.6 .. 60, overall ~10),
    per-channel DC offsets, a step edge down the middle, magnitudes > 100; default-init-scale head weights (head_gain 1, as in the RDN stress
    fixtures).  LR 24x24 -> x3.3 through the reference's batched_predict; the feature map is stored (its exp() would not be host-stable)."""
    m = edsr_generator(ref, mid=180, blocks=1).eval()
    sha = seeded_init_(m, seed=6, gain=1.0, head_gain=1.0)
    g = torch.Generator().manual_seed(9)
    base = torch.randn((1, 180, 24, 24), generator=g)
    zc = torch.randn(180, generator=g)
    sc = torch.tensor([10.0 * math.exp(0.9 * z - 0.81) for z in zc.double().tolist()], dtype=torch.float64).float()
    off = torch.randn(180, generator=g) * 5.0
    feat = base * sc.view(1, -1, 1, 1) + off.view(1, -1, 1, 1)
    feat[..., :, 12:] += (torch.randn(180, generator=g) * 8.0).view(1, -1, 1, 1)           # step edge
    coord, cell, (ht, wt) = coords_for(24, 24, 3.3)
    with torch.no_grad():
        nl = m.cs_attn(feat)
        out = m.batched_predict([feat], coord, cell)
    print(f'  head_c180_stress: feature std {feat.std():.2f} max {feat.abs().max():.1f}; non-local std {nl.std():.2f} max {nl.abs().max():.1f}; '
          f'out std {out.std():.3f} range [{out.min():.2f}, {out.max():.2f}]')
    save('head_c180_stress_x3p3', out=out[0], feature=feat, sha=np.array(sha), weight_seed=np.array(6), target=np.array([ht, wt]))


def gen_csattn_big(ref):
    """Maps of >= 4096 LR pixels: the size class of every C3/C4 tile (192x192), where the build takes its
    composed fold+down route.  64x64 is the smallest such map; 67x70 adds reflect-pad (both axes odd -> 68x70),
    crop, and the row-0 / column-0 edge variants on a non-square map (arch_csnln.py:444-449,462-469,505-526)."""
    for tag, (h, w) in (('64x64', (64, 64)), ('67x70', (67, 70))):
        att = ref.CrossScaleAttention(channel=64, scale=[2]).eval()
        sha = seeded_init_(att, seed=3, gain=1.5)
        x = randn((1, 64, h, w), 33)
        with torch.no_grad():
            y = att(x)
        print(f'  csattn_{tag}: out std {y.std():.4f}')
        save(f'csattn_c64_{tag}', out=y[0], sha=np.array(sha), in_seed=np.array(33), weight_seed=np.array(3),
             gain=np.array(1.5), shape=np.array([h, w]))


def gen_csattn_scales(ref):
    """CrossScaleAttention with scale entries 3, 4 and the list [2, 3, 4] (arch_csnln.py:421-427, :436-528; unused by the configs):
    C = 8, sizes that need the reflect mod-pad for every scale."""
    for tag, scale in (('s3', [3]), ('s4', [4]), ('s234', [2, 3, 4])):
        att = ref.CrossScaleAttention(channel=8, scale=scale).eval()
        seeded_init_(att, seed=6, gain=1.5)
        x = randn((2, 8, 10, 13), 34)
        with torch.no_grad():
            y = att(x)
        w_ = {('w.cs_attn.' + k): v for k, v in att.state_dict().items()}
        save(f'csattn_c8_{tag}', x=x, out=y, scale=np.array(scale), **w_)
    # and inside the head: multi_scale = [2, 3] widens imnet_q / imnet_v by 2C (ciaosr_net.py:73-76)
    m = edsr_generator(ref, mid=8, blocks=1, hidden=(32, 32), eval_bsize=None, multi_scale=[2, 3]).eval()
    sha = seeded_init_(m, seed=14, gain=1.0, head_gain=SQRT6)
    feat = randn((1, 8, 7, 9), 24)
    coord, cell, (ht, wt) = coords_for(7, 9, 2.7)
    with torch.no_grad():
        out = m.query_rgb([feat], coord, cell)
    w = {('w.' + k): v for k, v in head_sd(m).items()}
    save('tiny_head_ms23', feature=feat, coord=coord, cell=cell, out=out, sha=np.array(sha), target=np.array([ht, wt]), **w)


def tile192_subset(out):
    """The stored part of a [1,3,768,768] output: every 4th pixel + an 8-pixel frame (tile borders are where
    the clamp, the zero-padded unfold and the cs_attn edge variants act)."""
    return dict(out_s4=out[..., ::4, ::4], out_top=out[..., :8, :], out_bot=out[..., -8:, :],
                out_left=out[..., :, :8], out_right=out[..., :, -8:])


def gen_e2e_tile192(ref):
    """One full C3 tile: 192x192 LR through the reference CiaoSR.forward_test (tile=192 -> clip_test with exactly
    one tile, ciaosr.py:224-254; 20 eval_bsize chunks each recomputing cs_attn, ciaosr_net.py:241-246 -> :135).
    ~7 min of CPU.  Encoder gain 1.5: the network term is O(0.2) next to the bilinear residual and only a few per cent
    of the pixels saturate at the clamp (gain 1.6 saturates 45 % of a 192x192 tile and would hide errors there).  Stores a subset of the 768x768 output, its PSNR/max over the FULL output and the
    reference's PSNR against the synthetic GT (metrics.py:211-226 formula via ciaosr_amd.metrics)."""
    from ciaosr_amd.metrics import psnr_tensors
    mean = (0.4488, 0.4371, 0.4040)
    q, k, v = mlp_cfg((256,) * 4)
    gen = dict(type=ref.LocalImplicitSRRDN,
               encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
                            upscale_factor=4, num_layers=8, channel_growth=64),
               imnet_q=q, imnet_k=k, imnet_v=v, feat_unfold=True, eval_bsize=30000)
    test_cfg = ref.ConfigDict(scale=4, tile=192, tile_overlap=32)
    model = ref.CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'),
                       rgb_mean=mean, rgb_std=(1., 1., 1.), test_cfg=test_cfg).eval()
    sha = seeded_init_(model, seed=0, gain=1.5, head_gain=SQRT6)
    lq, gt = synthetic_pair(192, 192, 4)
    coord, cell, (ht, wt) = coords_for(192, 192, 4)
    import time
    t0 = time.time()
    with torch.no_grad():
        res = model(lq=lq, gt=None, test_mode=True, coord=coord, cell=cell)
    out = res['output']
    psnr_ref = psnr_tensors(out, gt, crop_border=4)
    print(f'  e2e_rdn_x4_tile192: {time.time() - t0:.0f} s, out range [{out.min():.3f},{out.max():.3f}] std '
          f'{out.std():.3f} frac clamped {(out.eq(0) | out.eq(1)).float().mean():.3f} PSNR(ref,GT) {psnr_ref:.4f}')
    save('e2e_rdn_x4_tile192', sha=np.array(sha), weight_seed=np.array(0), gain=np.array(1.5), scale=np.array(4),
         psnr_ref_gt=np.array(psnr_ref, dtype=np.float64), out_mean=np.array(out.double().mean().item()),
         out_sq=np.array((out.double() ** 2).mean().item()), **tile192_subset(out))


def gen_e2e(ref):
    mean = (0.4488, 0.4371, 0.4040)
    for tag, kind, scale in (('e2e_edsr_x2_48', 'edsr', 2), ('e2e_rdn_x4_48', 'rdn', 4)):
        q, k, v = mlp_cfg((256,) * 4)
        if kind == 'edsr':
            gen = dict(type=ref.LocalImplicitSREDSR,
                       encoder=dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16),
                       imnet_q=q, imnet_k=k, imnet_v=v, feat_unfold=True, eval_bsize=30000)
        else:
            gen = dict(type=ref.LocalImplicitSRRDN,
                       encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
                                    upscale_factor=4, num_layers=8, channel_growth=64),
                       imnet_q=q, imnet_k=k, imnet_v=v, feat_unfold=True, eval_bsize=30000)
        test_cfg = ref.ConfigDict(scale=scale, tile=192, tile_overlap=32)
        model = ref.CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'),
                           rgb_mean=mean, rgb_std=(1., 1., 1.), test_cfg=test_cfg).eval()
        gain = 1.25 if kind == 'edsr' else 1.45     # RDN: 1.6 clamps 31 % of the pixels at 0 / 1 (errors hide there), 1.45: 0.3 %
        sha = seeded_init_(model, seed=0, gain=gain, head_gain=SQRT6)
        lq, gt = synthetic_pair(48, 48, scale)
        coord, cell, (ht, wt) = coords_for(48, 48, scale)
        with torch.no_grad():
            feat = model.generator.gen_feature((lq - torch.tensor(mean).view(1, 3, 1, 1)))[0]
            res = model(lq=lq, gt=None, test_mode=True, coord=coord, cell=cell)
        out = res['output']
        print(f'  {tag}: feature std {feat.std():.3f}  out range [{out.min():.3f},{out.max():.3f}] '
              f'std {out.std():.3f}  frac clamped {(out.eq(0) | out.eq(1)).float().mean():.3f}')
        names = {k2: list(v2.shape) for k2, v2 in model.state_dict().items()}
        with open(os.path.join(OUT, f'state_dict_names_{kind}.json'), 'w') as f:
            json.dump(names, f, indent=0)
        save(tag, lq=lq, out=out, sha=np.array(sha), weight_seed=np.array(0),
             gain=np.array(gain), scale=np.array(scale))


def stress_input(h, w, scale):
    """LR input with a DC offset and a step edge (the right half 0.45 darker): what a Winograd tile straddling an edge sees."""
    lq, gt = synthetic_pair(h, w, scale)
    lq = (0.35 * lq + 0.5).clamp(0, 1)
    lq[..., :, w // 2:] = (lq[..., :, w // 2:] - 0.45).clamp(0, 1)
    return lq


STRESS = dict(gain=2.3, head_gain=1.0, sigma=1.0)      # trunk features: std ~10, max ~110, per-channel std 0.1 .. 26 (printed below)


def gen_stress(ref):
    """Condition-stress vectors for the fp32 default route (Winograd F(4x4, 3x3) dense layers, Winograd logit table): RDN-CiaoSR x4 at LR
    48x48 and 64x64 through the REFERENCE's CiaoSR.forward_test with TRAINED-LIKE trunk statistics (init_utils.trained_like_: log-normal
    per-output-channel scales, 1 % of the weights x20, bias offsets; trunk gain 2.3 -> feature magnitudes O(10 .. 100)) on an input with a
    DC offset and a step edge.  Stores the output, the reference-side trunk features (fp32) and the same trunk evaluated in fp64."""
    mean = (0.4488, 0.4371, 0.4040)
    for size in (48, 64):
        q, k, v = mlp_cfg((256,) * 4)
        gen = dict(type=ref.LocalImplicitSRRDN,
                   encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
                                upscale_factor=4, num_layers=8, channel_growth=64),
                   imnet_q=q, imnet_k=k, imnet_v=v, feat_unfold=True, eval_bsize=30000)
        model = ref.CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'),
                           rgb_mean=mean, rgb_std=(1., 1., 1.), test_cfg=ref.ConfigDict(scale=4, tile=192, tile_overlap=32)).eval()
        seeded_init_(model, seed=0, gain=STRESS['gain'], head_gain=STRESS['head_gain'])
        sha = trained_like_(model, seed=0, sigma=STRESS['sigma'])
        lq = stress_input(size, size, 4)
        coord, cell, (ht, wt) = coords_for(size, size, 4)
        x = lq - torch.tensor(mean).view(1, 3, 1, 1)
        with torch.no_grad():
            feat = model.generator.gen_feature(x)[0]
            res = model(lq=lq, gt=None, test_mode=True, coord=coord, cell=cell)
            feat64 = model.generator.double().gen_feature(x.double())[0]
            model.generator.float()
        out = res['output']
        cs = feat.std(dim=(0, 2, 3))
        print(f'  stress_rdn_x4_{size}: feature std {feat.std():.2f} max {feat.abs().max():.1f} per-channel std {cs.min():.2f} .. {cs.max():.2f}; '
              f'fp32 vs fp64 trunk {(feat.double() - feat64).abs().max():.2e}; out range [{out.min():.3f},{out.max():.3f}] std {out.std():.3f} '
              f'frac clamped {(out.eq(0) | out.eq(1)).float().mean():.3f}')
        save(f'stress_rdn_x4_{size}', lq=lq, out=out, feat=feat, feat64=feat64.float(), sha=np.array(sha), weight_seed=np.array(0),
             gain=np.array(STRESS['gain']), head_gain=np.array(STRESS['head_gain']), sigma=np.array(STRESS['sigma']), scale=np.array(4))


def gen_tiling(ref):
    """clip_test index lists + E/W blending: LR 100x132, tile 48, overlap 16, x2, small EDSR."""
    q, k, v = mlp_cfg((64, 64))
    gen = dict(type=ref.LocalImplicitSREDSR,
               encoder=dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=16, num_blocks=2),
               imnet_q=q, imnet_k=k, imnet_v=v, feat_unfold=True, eval_bsize=30000)
    test_cfg = ref.ConfigDict(scale=2, tile=48, tile_overlap=16)
    model = ref.CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss'), rgb_mean=(0.4488, 0.4371, 0.4040),
                       rgb_std=(1., 1., 1.), test_cfg=test_cfg).eval()
    sha = seeded_init_(model, seed=9, gain=1.75, head_gain=SQRT6)      # 2.0 clamps 30 % of the pixels, 1.75: 0.6 %
    lq, gt = synthetic_pair(100, 132, 2)
    coord, cell, _ = coords_for(100, 132, 2)
    with torch.no_grad():
        res = model(lq=lq, gt=None, test_mode=True, coord=coord, cell=cell)
    out = res['output']
    print(f'  tiling_small: out range [{out.min():.3f},{out.max():.3f}] std {out.std():.3f} frac clamped {(out.eq(0) | out.eq(1)).float().mean():.3f}')
    w = {('w.' + k2): v2 for k2, v2 in model.state_dict().items()}
    save('tiling_small', lq=lq, out=out, sha=np.array(sha), **w)


def gen_swinir(ref):
    """SwinIR-CiaoSR (config C5 shape: embed 180, 6x6 blocks, window 8): parameter names, trunk features on an
    odd-sized input (reflect pad to the window multiple + crop, ciaosr_net.py:509-523) and the end-to-end
    whole-image path at x3.3 (non-integer scale: no tiling, ciaosr.py:158)."""
    import torch.nn as nn
    from mmedited.models.backbones.sr_backbones.swinir_net import SwinIR
    orig_cuda = nn.Module.cuda
    nn.Module.cuda = lambda self, device=None: self          # swinir_net.py:684,723,725 hard-code .cuda()
    try:
        q, k, v = mlp_cfg((256,) * 4)
        gen = dict(type=ref.LocalImplicitSRSWINIR, window_size=8,
                   encoder=dict(type=SwinIR, upscale=4, in_chans=3, img_size=48, window_size=8, img_range=1.,
                                depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, mlp_ratio=2,
                                upsampler='pixelshuffle', resi_connection='1conv'),
                   imnet_q=q, imnet_k=k, imnet_v=v, feat_unfold=True, eval_bsize=30000)
        model = ref.CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss'), rgb_mean=(0.4488, 0.4371, 0.4040),
                           rgb_std=(1., 1., 1.), test_cfg=ref.ConfigDict(scale=3.3)).eval()
    finally:
        nn.Module.cuda = orig_cuda
    gain = 0.6                                   # trunk gain 1.0 clamps 31 % of the output pixels at 0 / 1; 0.6: 2 %
    sha = seeded_init_(model, seed=2, gain=gain, head_gain=SQRT6)
    names = {k2: list(v2.shape) for k2, v2 in model.state_dict().items()}
    with open(os.path.join(OUT, 'state_dict_names_swinir.json'), 'w') as f:
        json.dump(names, f, indent=0)
    x = randn((1, 3, 20, 27), 91) * 0.3
    with torch.no_grad():
        feat = model.generator.gen_feature(x)[0]
    lq, gt = synthetic_pair(24, 24, 3.3)
    coord, cell, (ht, wt) = coords_for(24, 24, 3.3)
    with torch.no_grad():
        res = model(lq=lq, gt=None, test_mode=True, coord=coord, cell=cell)
    out = res['output']
    print(f'  swinir: feat std {feat.std():.3f}, out range [{out.min():.3f},{out.max():.3f}] std {out.std():.3f} '
          f'frac clamped {(out.eq(0) | out.eq(1)).float().mean():.3f}')
    save('swinir_c5', feat=feat[0], x_seed=np.array(91), lq=lq, out=out, sha=np.array(sha), weight_seed=np.array(2),
         gain=np.array(gain), target=np.array([ht, wt]))


def gen_swinir48(ref):
    """BASELINE config 5 at ITS OWN size: SwinIR-CiaoSR x3.3, LR 48x48 -> 158x158 (Q = 24 964, C = 180) through the reference's
    CiaoSR.forward_test (whole-image path, ciaosr.py:158-169), plus the reference trunk's feature map of the same input
    (ciaosr_net.py:475-525; every 2nd pixel stored).  48 = 6 windows of 8: the blocks' own attn_mask buffers are used
    (input_resolution == x_size, swinir_net.py:233-236), unlike the 24x24 fixture, which takes calculate_mask."""
    import torch.nn as nn
    from ciaosr_amd.metrics import psnr_tensors
    from mmedited.models.backbones.sr_backbones.swinir_net import SwinIR
    orig_cuda = nn.Module.cuda
    nn.Module.cuda = lambda self, device=None: self          # swinir_net.py:684,723,725 hard-code .cuda()
    try:
        q, k, v = mlp_cfg((256,) * 4)
        gen = dict(type=ref.LocalImplicitSRSWINIR, window_size=8,
                   encoder=dict(type=SwinIR, upscale=4, in_chans=3, img_size=48, window_size=8, img_range=1.,
                                depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, mlp_ratio=2,
                                upsampler='pixelshuffle', resi_connection='1conv'),
                   imnet_q=q, imnet_k=k, imnet_v=v, feat_unfold=True, eval_bsize=30000)
        model = ref.CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss'), rgb_mean=(0.4488, 0.4371, 0.4040),
                           rgb_std=(1., 1., 1.), test_cfg=ref.ConfigDict(scale=3.3)).eval()
    finally:
        nn.Module.cuda = orig_cuda
    gain = 0.55                                  # trunk gain 1.0 clamps 23 % of the output pixels at 0 / 1; 0.55: 4 %
    sha = seeded_init_(model, seed=2, gain=gain, head_gain=SQRT6)
    lq, gt = synthetic_pair(48, 48, 3.3)
    coord, cell, (ht, wt) = coords_for(48, 48, 3.3)
    import time
    t0 = time.time()
    with torch.no_grad():
        feat = model.generator.gen_feature(lq - torch.tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1))[0]
        res = model(lq=lq, gt=None, test_mode=True, coord=coord, cell=cell)
    out = res['output']
    psnr_ref = psnr_tensors(out, gt, crop_border=3)
    print(f'  swinir_c5_48: {time.time() - t0:.0f} s, feat std {feat.std():.3f}, out {tuple(out.shape)} range [{out.min():.3f},{out.max():.3f}] '
          f'std {out.std():.3f} frac clamped {(out.eq(0) | out.eq(1)).float().mean():.3f} PSNR(ref,GT) {psnr_ref:.4f}')
    save('swinir_c5_48', feat_s2=feat[0][:, ::2, ::2], lq=lq, out=out, sha=np.array(sha), weight_seed=np.array(2), gain=np.array(gain),
         target=np.array([ht, wt]), psnr_ref_gt=np.array(psnr_ref, dtype=np.float64))


ALL = dict(tiny_head=gen_tiny_head, tiny_variants=gen_tiny_head_variants, tiny_act=gen_tiny_head_act, csattn_scales=gen_csattn_scales, head_c64=gen_head_c64,
           head_c64_x3p3=gen_head_c64_x3p3, nearest_idx=gen_nearest_idx, csattn=gen_csattn,
           head_c180=gen_head_c180, k4_c64=gen_k4_c64, head_c180_stress=gen_head_c180_stress, e2e=gen_e2e, csattn_big=gen_csattn_big, e2e_tile192=gen_e2e_tile192, tiling=gen_tiling, swinir=gen_swinir, swinir48=gen_swinir48, stress=gen_stress)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref = ref_stubs.load_reference()
    for name, fn in ALL.items():
        if args.only and name not in args.only:
            continue
        print(f'[{name}]')
        fn(ref)
