"""Deterministic, construction-order-independent parameter initialisation.

SURVEY 8(d): random-init weights must not depend on module construction order (the
real mmedit encoders consume RNG for layers CiaoSR later deletes).  Every tensor in
`state_dict()` is drawn from its own CPU generator seeded by (seed, crc32(name)), so
any model exposing the reference's parameter names gets bit-identical weights.

  Linear/Conv weight <- randn * gain / sqrt(3 * fan_in)   (variance of torch's default
                         kaiming_uniform(a=sqrt(5)) when gain == 1)
  bias               <- randn * 0.01
  PReLU slope        <- 0.25;  LayerNorm <- (1, 0);  relative_position_bias_table <- randn*0.02
Timing runs use gain 1; parity runs use head_gain sqrt(6) on imnet_* (SURVEY fact 5b).
"""
import hashlib
import math
import zlib

import torch


def _gen(seed, name):
    g = torch.Generator(device='cpu')
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def seeded_tensor(name, shape, seed=0, gain=1.0, head_gain=None):
    """The fp32 tensor `seeded_init_` assigns to state_dict entry `name` of `shape`."""
    head_gain = gain if head_gain is None else head_gain
    shape = tuple(shape)
    g = _gen(seed, name)
    parts = name.split('.')
    leaf = parts[-1]
    if name.endswith('escape_NaN'):
        return torch.full(shape, 1e-4)
    if 'relative_position_bias_table' in name:
        return torch.randn(shape, generator=g) * 0.02
    if leaf == 'weight' and len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        is_head = any(p.startswith('imnet_') for p in parts)
        return torch.randn(shape, generator=g) * ((head_gain if is_head else gain) / math.sqrt(3.0 * fan_in))
    if leaf == 'weight' and len(shape) == 1:
        # PReLU slope (numel 1) or LayerNorm weight
        return torch.full(shape, 0.25) if shape[0] == 1 else torch.ones(shape)
    if leaf == 'bias':
        is_norm = len(parts) >= 2 and 'norm' in parts[-2]
        return torch.zeros(shape) if is_norm else torch.randn(shape, generator=g) * 0.01
    return torch.randn(shape, generator=g) * 0.02


def seeded_state_dict(shapes, seed=0, gain=1.0, head_gain=None):
    """name -> tensor for a {name: shape} mapping (same values as seeded_init_ on a module)."""
    return {k: seeded_tensor(k, v, seed, gain, head_gain) for k, v in shapes.items()}


@torch.no_grad()
def seeded_init_(module, seed=0, gain=1.0, head_gain=None):
    """Overwrite every floating parameter/buffer of `module` in place.  Returns sha256 of the fp32 bytes."""
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        if torch.is_floating_point(t):
            t.copy_(seeded_tensor(name, t.shape, seed, gain, head_gain).to(t.dtype))
    return state_dict_sha256(sd)


def state_dict_sha256(module_or_sd):
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    h = hashlib.sha256()
    for name in sorted(sd.keys()):
        t = sd[name]
        if not torch.is_floating_point(t):
            continue
        h.update(name.encode())
        h.update(t.detach().cpu().float().contiguous().numpy().tobytes())
    return h.hexdigest()


def synthetic_gt(ht, wt, seed=1234):
    """Synthetic HR image: smooth sinusoid field + band-limited texture, in [0,1]."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    i = torch.arange(ht, dtype=torch.float32).view(1, ht, 1)
    j = torch.arange(wt, dtype=torch.float32).view(1, 1, wt)
    c = torch.arange(3, dtype=torch.float32).view(3, 1, 1)
    base = 0.5 + 0.25 * torch.sin(2 * math.pi * (3 * i / ht + 5 * j / wt) + c)
    noise = torch.randn(1, 3, ht, wt, generator=g)
    tex = torch.nn.functional.avg_pool2d(noise, 5, stride=1, padding=2, count_include_pad=True)[0]
    return (base + 0.1 * tex).clamp(0, 1).unsqueeze(0)


def synthetic_pair(h, w, scale, seed=1234):
    """(lq [1,3,h,w], gt [1,3,round(h*s),round(w*s)]) with lq = antialiased bicubic of gt, CPU."""
    ht, wt = round(h * scale), round(w * scale)
    gt = synthetic_gt(ht, wt, seed)
    lq = torch.nn.functional.interpolate(gt, size=(h, w), mode='bicubic', antialias=True,
                                         align_corners=False).clamp(0, 1)
    return lq, gt


@torch.no_grad()
def trained_like_(module, seed=0, sigma=1.0, outlier_frac=0.01, outlier_scale=20.0, bias_std=0.2):
    """Re-shape the TRUNK weights of an already `seeded_init_`-ed model towards the statistics of a trained network -- what the
    Winograd forms of the dense layers are sensitive to and Gaussian weights do not show: every 4-D convolution weight outside the head
    (`imnet_*`, `cs_attn*`) gets a log-normal scale per OUTPUT channel (exp(sigma z), normalised to unit mean square), `outlier_frac` of
    its entries are multiplied by `outlier_scale`, and its bias a N(0, bias_std) offset: per-channel spread, a few large weights,
    DC-offset features.  Deterministic per (seed, parameter name), like `seeded_init_`.  Returns the sha256 of the resulting weights."""
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        parts = name.split('.')
        if not torch.is_floating_point(t) or any(p.startswith('imnet_') or p.startswith('cs_attn') for p in parts):
            continue
        g = _gen(seed + 7919, name)
        if parts[-1] == 'weight' and t.dim() == 4:
            z = torch.randn(t.shape[0], generator=g)
            # E[scale^2] = 1.  exp() through Python's libm on doubles, rounded to fp32 once: torch.exp's vectorised fp32 path differs in
            # the last bit between CPU generations (AVX2 / AVX-512), and the weights' sha256 must not depend on the host
            scale = torch.tensor([math.exp(sigma * zv - sigma * sigma) for zv in z.double().tolist()], dtype=torch.float64).float()
            mask = torch.rand(t.shape, generator=g) < outlier_frac
            w = t * scale.view(-1, 1, 1, 1)
            w = torch.where(mask, w * outlier_scale, w)
            # expected rms unchanged, so that the gain keeps its meaning -- by the ANALYTIC factor (a measured rms would depend on the
            # reduction order, i.e. on the host's thread count, and with it the last bit of every weight and the fixture's sha256)
            w = w * (1.0 / math.sqrt(1.0 + outlier_frac * (outlier_scale * outlier_scale - 1.0)))
            t.copy_(w)
        elif parts[-1] == 'bias' and t.dim() == 1 and len(parts) >= 2 and 'norm' not in parts[-2]:
            t.add_(torch.randn(t.shape, generator=g) * bias_std)
    return state_dict_sha256(sd)
