"""Import the reference's hot-path Python files UNMODIFIED from /root/reference (CPU only).

Only used in the build container by tools/make_golden.py and by the `-m "not gpu"` oracle
pin tests when /root/reference is present; never shipped to / used on the GPU box.

The reference depends on mmcv / mmedit 0.11.0 / torchvision / timm / thop, none of which is
installed here.  The stubs below carry no arithmetic of the hot path except the third-party
pieces SURVEY 8(c) lists as "parity unpinned": make_coord, RDN/EDSR encoders, tensor2img/psnr,
which resolve to this repo's own restatements of the public mmedit definitions.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('CIAOSR_REFERENCE', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, 'mmedited'))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition('.')
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


def install():
    """Install stub modules and make `mmedited.*` resolve to the reference tree."""
    if 'mmedited' in sys.modules and getattr(sys.modules['mmedited'], '_ciaosr_ref', False):
        return
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from ciaosr_amd import coords, metrics
    from ciaosr_amd.encoders import RDN, EDSR

    # drop this repo's shim package if it was imported first
    for k in [k for k in sys.modules if k == 'mmedited' or k.startswith('mmedited.')]:
        del sys.modules[k]
    pkg = types.ModuleType('mmedited')
    pkg.__path__ = [os.path.join(REF_ROOT, 'mmedited')]
    pkg._ciaosr_ref = True
    sys.modules['mmedited'] = pkg

    registry = {}

    def _build(cfg, **extra):
        cfg = dict(cfg)
        typ = cfg.pop('type')
        cls = registry[typ] if isinstance(typ, str) else typ
        cfg.update(extra)
        return cls(**cfg)

    def load_checkpoint(model, filename, map_location=None, strict=False, logger=None, **kw):
        ckpt = torch.load(filename, map_location=map_location or 'cpu')
        sd = ckpt.get('state_dict', ckpt)
        model.load_state_dict(sd, strict=strict)
        return ckpt

    def auto_fp16(apply_to=None, out_fp32=False):
        return lambda fn: fn

    class _Dict(dict):
        __getattr__ = dict.get

    _mod('mmcv', imwrite=lambda img, path, **kw: None, ConfigDict=_Dict)
    _mod('mmcv.runner', load_checkpoint=load_checkpoint, auto_fp16=auto_fp16)
    _mod('mmcv.cnn', constant_init=lambda m, val, bias=0: None)
    _mod('mmedit')
    _mod('mmedit.utils', get_root_logger=lambda *a, **k: None)
    _mod('mmedit.core', tensor2img=metrics.tensor2img, psnr=metrics.psnr, ssim=metrics.ssim)
    _mod('mmedit.models')
    _mod('mmedit.models.base', BaseModel=nn.Module)
    _mod('mmedit.models.builder', build_backbone=_build, build_component=_build, build_loss=_build)
    _mod('mmedit.datasets')
    _mod('mmedit.datasets.pipelines')
    _mod('mmedit.datasets.pipelines.utils', make_coord=coords.make_coord)
    _mod('torchvision')
    _mod('torchvision.models')
    _mod('torchvision.models.vgg')
    _mod('thop', profile=lambda *a, **k: (0, 0))
    _mod('timm')
    _mod('timm.models')

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    _mod('timm.models.layers', DropPath=DropPath, to_2tuple=to_2tuple,
         trunc_normal_=lambda t, std=1., **k: nn.init.trunc_normal_(t, std=std))

    from mmedited.models.components.refiners.mlp_refiner import MLPRefiner  # reference, in-repo copy

    class L1Loss(nn.Module):
        def __init__(self, loss_weight=1.0, reduction='mean', **kw):
            super().__init__()

    registry.update(RDN=RDN, EDSR=EDSR, MLPRefiner=MLPRefiner, L1Loss=L1Loss)
    return registry


def load_reference():
    """Returns a namespace with the reference classes of the hot path."""
    install()
    from mmedited.models.backbones.sr_backbones import ciaosr_net
    from mmedited.models.common.arch_csnln import CrossScaleAttention
    from mmedited.models.components.refiners.mlp_refiner import MLPRefiner
    from mmedited.models.restorers.ciaosr import CiaoSR
    import mmcv
    return types.SimpleNamespace(
        ciaosr_net=ciaosr_net, LocalImplicitSRNet=ciaosr_net.LocalImplicitSRNet,
        LocalImplicitSRRDN=ciaosr_net.LocalImplicitSRRDN, LocalImplicitSREDSR=ciaosr_net.LocalImplicitSREDSR,
        LocalImplicitSRSWINIR=ciaosr_net.LocalImplicitSRSWINIR,
        CrossScaleAttention=CrossScaleAttention, MLPRefiner=MLPRefiner, CiaoSR=CiaoSR,
        ConfigDict=mmcv.ConfigDict)
