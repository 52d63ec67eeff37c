"""CPU oracle for the CiaoSR LocalImplicitSR forward path  --  TEST INFRASTRUCTURE ONLY.

This file is a torch-CPU fp32 restatement of the reference algorithm.  It is the checker for
the HIP path: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  Nothing under `ciaosr_amd/` imports it and the product path never falls back to it.

Pinning (see tests/test_oracle_pin.py, tools/make_golden.py): every function below is checked
against outputs of the reference itself (the unmodified /root/reference Python imported in the
build container with stub mmcv/mmedit modules) stored as fixtures under tests/golden/.
Third-party arithmetic the reference pulls from mmedit 0.11.0 (README.md:57) and that is absent
from /root/reference -- make_coord, the RDN/EDSR encoders, tensor2img/psnr -- is restated from
the published definitions; for those boundaries parity is "unpinned"  and the
fixtures pin the head/restorer given this repo's encoder restatement.

All functions are functional: weights come in a flat dict with the reference's state_dict names
(`imnet_k.layers.0.weight`, `cs_attn.conv_match_1.0.weight`, ...).  The default mode follows the
reference op for op, including its redundancy (materialised unfold, cs_attn recomputed for every
eval_bsize chunk, ciaosr_net.py:241-246 -> :132-136), so it can be timed as the CPU baseline.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# coordinates (mmedit make_coord; call sites ciaosr_net.py:148, ciaosr.py:240)
# ----------------------------------------------------------------------------------------------
def make_coord(shape, flatten=True):
    axes = []
    for n in shape:
        r = 1.0 / n
        axes.append((-1.0 + r) + (2.0 * r) * torch.arange(n).float())
    g = torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=-1)
    return g.reshape(-1, 2) if flatten else g


def make_cell(target_hw):
    ht, wt = target_hw
    cell = torch.ones(ht * wt, 2)
    cell[:, 0] *= 2 / ht
    cell[:, 1] *= 2 / wt
    return cell


# ----------------------------------------------------------------------------------------------
# MLPRefiner (mlp_refiner.py:65-102): Linear/ReLU x len(hidden) + Linear
# ----------------------------------------------------------------------------------------------
def mlp_layer_ids(params, prefix):
    ids = sorted({int(k[len(prefix) + 8:].split('.')[0]) for k in params
                  if k.startswith(prefix + '.layers.') and k.endswith('.weight')})
    return ids


def mlp(x, params, prefix, act=None):
    """MLPRefiner.forward (mlp_refiner.py:74-102): Linear / act ... Linear; act = ReLU unless 'cos' / 'sin' (:81-86)."""
    ids = mlp_layer_ids(params, prefix)
    lead = x.shape[:-1]
    h = x.reshape(-1, x.shape[-1])
    fn = torch.cos if act == 'cos' else torch.sin if act == 'sin' else torch.relu
    for n, i in enumerate(ids):
        h = F.linear(h, params[f'{prefix}.layers.{i}.weight'], params[f'{prefix}.layers.{i}.bias'])
        if n + 1 < len(ids):
            h = fn(h)
    return h.reshape(*lead, -1)


# ----------------------------------------------------------------------------------------------
# CrossScaleAttention for scale list (arch_csnln.py:407-532; helpers :32-87)
# ----------------------------------------------------------------------------------------------
def _same_pad(t, k, s):
    """TF-style 'same' zero padding (arch_csnln.py:32-48)."""
    rows, cols = t.shape[-2:]
    o_r, o_c = (rows + s - 1) // s, (cols + s - 1) // s
    p_r = max(0, (o_r - 1) * s + k - rows)
    p_c = max(0, (o_c - 1) * s + k - cols)
    top, left = int(p_r / 2.), int(p_c / 2.)
    return F.pad(t, (left, p_c - left, top, p_r - top))


def _conv1x1_prelu(t, params, name):
    y = F.conv2d(t, params[f'{name}.0.weight'], params[f'{name}.0.bias'])
    return F.prelu(y, params[f'{name}.1.weight'])


def cross_scale_attention(x, params, prefix='cs_attn', scales=(2,), ksize=3, softmax_scale=10.0):
    B, C, H, W = x.shape
    floor = params.get(f'{prefix}.escape_NaN', torch.tensor([1e-4])).to(x)
    outs = []
    for s in scales:
        ph = (s - H % s) % s
        pw = (s - W % s) % s
        xp = F.pad(x, (0, pw, 0, ph), mode='reflect')                              # :444-449
        emb = _conv1x1_prelu(xp, params, f'{prefix}.conv_assembly')               # :452
        mat = _conv1x1_prelu(xp, params, f'{prefix}.conv_match_1')                # :453
        Hp, Wp = xp.shape[-2:]
        kk = s * ksize
        # value patches: kk x kk, stride s, 'same' padding  (:462-469)
        vpat = F.unfold(_same_pad(emb, kk, s), kk, stride=s)                       # [B, C*kk*kk, L]
        L = vpat.shape[-1]
        vpat = vpat.view(B, C, kk, kk, L).permute(0, 4, 1, 2, 3).contiguous()      # [B, L, C, kk, kk]
        # key patches from the downscaled map (:474-484)
        ref = F.interpolate(xp, scale_factor=1. / s, mode='bilinear')
        ref = _conv1x1_prelu(ref, params, f'{prefix}.conv_match_2')
        hr, wr = ref.shape[-2:]
        kpat = F.unfold(_same_pad(ref, ksize, 1), ksize, stride=1)
        kpat = kpat.view(B, ref.shape[1], ksize, ksize, -1).permute(0, 4, 1, 2, 3).contiguous()
        per_item = []
        for b in range(B):                                                        # :491
            wi = kpat[b]
            nrm = torch.sqrt((wi ** 2).sum(dim=(1, 2, 3), keepdim=True))
            wi_n = wi / torch.max(nrm, floor)                                     # :494-496
            score = F.conv2d(_same_pad(mat[b:b + 1], ksize, 1), wi_n, stride=1)   # [1, L, Hp, Wp]
            score = score.view(1, hr * wr, Hp, Wp)
            prob = F.softmax(score * softmax_scale, dim=1)                        # :505
            y = F.conv_transpose2d(prob, vpat[b], stride=s, padding=s)            # :511
            down = {2: 'down', 3: 'downx3', 4: 'downx4'}[s]
            y = F.conv2d(y, params[f'{prefix}.{down}.weight'], params[f'{prefix}.{down}.bias'],
                         stride=s, padding=1)                                     # :516
            per_item.append(y / 6.)                                               # :522
        y = torch.cat(per_item, 0)[:, :, :H, :W]                                  # :526
        outs.append(y)
    return torch.cat(outs, 1)


# ----------------------------------------------------------------------------------------------
# query_rgb (ciaosr_net.py:113-224), op for op
# ----------------------------------------------------------------------------------------------
def _nearest(fmap, coord):
    """F.grid_sample nearest at coord (y,x) -> [B, Q, Cmap]."""
    g = coord.flip(-1).unsqueeze(1)
    return F.grid_sample(fmap, g, mode='nearest', align_corners=False)[:, :, 0, :].permute(0, 2, 1)


def shift_list(local_size):
    if local_size == 1:
        return [(0, 0)]
    step = 4 - local_size
    return [(i, j) for i in range(-1, 2, step) for j in range(-1, 2, step)]


def query_rgb(feature, coord, cell, params, local_size=2, softmax_scale=1.0, feat_unfold=True,
              non_local=True, multi_scale=(2,), nonlocal_map=None, return_intermediates=False, act=None):
    B, C, H, W = feature.shape
    if feat_unfold:
        fq = F.unfold(feature, 3, padding=1).view(B, C * 9, H, W)
        fk = F.unfold(feature, 3, padding=1).view(B, C * 9, H, W)
        fv = F.unfold(feature, 3, padding=1).view(B, C * 9, H, W)
    else:
        fq = fk = fv = feature
    if non_local:
        nl = cross_scale_attention(feature, params, scales=multi_scale) if nonlocal_map is None else nonlocal_map
        fv = torch.cat([fv, nl], dim=1)
    query = _nearest(fq, coord).contiguous()                                       # [B,Q,D]
    fcoord = make_coord((H, W), flatten=False).permute(2, 0, 1).unsqueeze(0).expand(B, 2, H, W).to(coord)
    eps = 1e-6
    pk, pv, inter = [], [], {}
    for (vy, vx) in shift_list(local_size):     # reference names these (vx, vy): first moves axis 0
        ty = ((H - 1) / (1 - cell[:, 0, 0])).view(B, 1)
        tx = ((W - 1) / (1 - cell[:, 0, 1])).view(B, 1)
        c_ = coord.clone()
        if vy != 0:
            ry = (2 * abs(vy) - 1) / ty
            c_[:, :, 0] += vy / abs(vy) * ry + eps
        if vx != 0:
            rx = (2 * abs(vx) - 1) / tx
            c_[:, :, 1] += vx / abs(vx) * rx + eps
        c_.clamp_(-1 + 1e-6, 1 - 1e-6)
        key = _nearest(fk, c_).contiguous()
        val = _nearest(fv, c_).contiguous()
        ck = _nearest(fcoord, c_)
        rel = coord - ck
        rel[:, :, 0] *= H
        rel[:, :, 1] *= W
        sc = cell.clone()
        sc[:, :, 0] *= H
        sc[:, :, 1] *= W
        bs, q = coord.shape[:2]
        in_k = torch.cat([key, rel, sc], -1).view(bs * q, -1)
        in_v = torch.cat([val, rel, sc], -1).view(bs * q, -1)
        wk = mlp(in_k, params, 'imnet_k', act).view(bs, q, -1)
        wv = mlp(in_v, params, 'imnet_v', act).view(bs, q, -1)
        pk.append(key * wk)
        pv.append(val * wv)
    pk = torch.stack(pk, dim=-1)                 # [B,Q,D,J]
    pv = torch.stack(pv, dim=-2)                 # [B,Q,J,Dv]
    logit = query.unsqueeze(2) @ pk              # [B,Q,1,J]
    attn = (logit / softmax_scale).softmax(dim=-1)
    z = (attn @ pv).view(coord.shape[0] * coord.shape[1], -1)
    out = mlp(z, params, 'imnet_q', act).view(coord.shape[0], coord.shape[1], -1)
    if return_intermediates:
        inter.update(logit=logit[:, :, 0], attn=attn[:, :, 0], z=z.view(coord.shape[0], coord.shape[1], -1))
        return out, inter
    return out


def batched_predict(feature, coord, cell, params, eval_bsize=30000, hoist_nonlocal=False, **kw):
    """ciaosr_net.py:226-248.  hoist_nonlocal=True computes cs_attn once (result-identical)."""
    nl = None
    if hoist_nonlocal and kw.get('non_local', True):
        nl = cross_scale_attention(feature, params, scales=kw.get('multi_scale', (2,)))
    n, left, outs = coord.shape[1], 0, []
    with torch.no_grad():
        while left < n:
            right = min(left + eval_bsize, n)
            outs.append(query_rgb(feature, coord[:, left:right], cell[:, left:right], params,
                                  nonlocal_map=nl, **kw))
            left = right
    return torch.cat(outs, 1)


def bilinear_residual(x, coord):
    """ciaosr_net.py:107-108."""
    return F.grid_sample(x, coord.flip(-1).unsqueeze(1), mode='bilinear', padding_mode='border',
                         align_corners=False)[:, :, 0, :].permute(0, 2, 1)


# ----------------------------------------------------------------------------------------------
# encoders (restated public mmedit 0.11 definitions; names per ciaosr_net.py:314-318, :388-390)
# ----------------------------------------------------------------------------------------------
def _conv(x, params, name, pad):
    return F.conv2d(x, params[name + '.weight'], params[name + '.bias'], padding=pad)


def rdn_features(x, params, deadline=None):
    """`deadline` (time.perf_counter() value; bench.py's cpu_baseline thread sweep only): give up -- return None -- once it has passed,
    checked between residual dense blocks."""
    import time
    s1 = _conv(x, params, 'sfe1', 1)
    h = _conv(s1, params, 'sfe2', 1)
    nb = 1 + max(int(k.split('.')[1]) for k in params if k.startswith('rdbs.'))
    local = []
    for b in range(nb):
        if deadline is not None and time.perf_counter() > deadline:
            return None
        nl = 1 + max(int(k.split('.')[3]) for k in params if k.startswith(f'rdbs.{b}.layers.'))
        t = h
        for l in range(nl):
            t = torch.cat([t, torch.relu(_conv(t, params, f'rdbs.{b}.layers.{l}.conv', 1))], 1)
        h = h + _conv(t, params, f'rdbs.{b}.lff', 0)
        local.append(h)
    g = _conv(torch.cat(local, 1), params, 'gff.0', 0)
    return _conv(g, params, 'gff.1', 1) + s1


def edsr_features(x, params, res_scale=1.0):
    f = _conv(x, params, 'conv_first', 1)
    nb = 1 + max(int(k.split('.')[1]) for k in params if k.startswith('body.'))
    h = f
    for b in range(nb):
        h = h + _conv(torch.relu(_conv(h, params, f'body.{b}.conv1', 1)), params, f'body.{b}.conv2', 1) * res_scale
    return _conv(h, params, 'conv_after_body', 1) + f


def encoder_features(x, params, deadline=None):
    if 'sfe1.weight' in params:
        return rdn_features(x, params, deadline)
    if 'conv_first.weight' in params and 'body.0.conv1.weight' in params:
        return edsr_features(x, params)
    raise KeyError('unknown encoder parameter naming')


# ----------------------------------------------------------------------------------------------
# generator forward (ciaosr_net.py:88-110) and restorer (ciaosr.py:111-169, :218-258)
# ----------------------------------------------------------------------------------------------
def generator_forward(x, coord, cell, params, eval_bsize=30000, feature=None, hoist_nonlocal=False, **kw):
    feat = encoder_features(x, params) if feature is None else feature
    with torch.no_grad():
        if eval_bsize is None:
            pred = query_rgb(feat, coord, cell, params, **kw)
        else:
            pred = batched_predict(feat, coord, cell, params, eval_bsize, hoist_nonlocal, **kw)
        return pred + bilinear_residual(x, coord)


def tile_starts(n, tile, overlap):
    stride = tile - overlap
    return list(range(0, n - tile, stride)) + [n - tile]


def clip_test(x, params, scale, tile, tile_overlap, **kw):
    b, c, h, w = x.shape
    tile = min(tile, h, w)
    E = torch.zeros(b, c, h * scale, w * scale)
    Wt = torch.zeros_like(E)
    for hi in tile_starts(h, tile, tile_overlap):
        for wi in tile_starts(w, tile, tile_overlap):
            patch = x[..., hi:hi + tile, wi:wi + tile]
            th, tw = round(patch.shape[-2] * scale), round(patch.shape[-1] * scale)
            coord = make_coord((th, tw)).unsqueeze(0).expand(b, -1, 2)
            cell = make_cell((th, tw)).unsqueeze(0).expand(b, -1, 2)
            out = generator_forward(patch, coord, cell, params, **kw)
            out = out.view(b, th, tw, 3).permute(0, 3, 1, 2)
            E[..., hi * scale:(hi + tile) * scale, wi * scale:(wi + tile) * scale] += out
            Wt[..., hi * scale:(hi + tile) * scale, wi * scale:(wi + tile) * scale] += 1
    out = E / Wt
    return out.view(b, 3, -1).permute(0, 2, 1).contiguous()


def forward_test(lq, coord, cell, params, rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1., 1., 1.),
                 scale=None, tile=None, tile_overlap=None, **kw):
    """Returns the de-normalised, clamped image [B,3,round(h*s),round(w*s)] (ciaosr.py:142-169)."""
    mean = torch.tensor(rgb_mean).view(1, 3, 1, 1)
    std = torch.tensor(rgb_std).view(1, 3, 1, 1)
    x = (lq - mean) / std
    with torch.no_grad():
        if tile:
            pred = clip_test(x, params, scale, tile, tile_overlap, **kw)
        else:
            pred = generator_forward(x, coord, cell, params, **kw)
        pred = pred * std.view(1, 1, 3) + mean.view(1, 1, 3)
        pred = pred.clamp(0, 1)
    ih, iw = lq.shape[-2:]
    s = math.sqrt(pred.shape[1] / (ih * iw))
    return pred.view(lq.shape[0], round(ih * s), round(iw * s), 3).permute(0, 3, 1, 2).contiguous()
