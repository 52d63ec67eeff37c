"""Minimal counterpart of mmedit's builder/registry (mmedit.models.builder, external to the
reference): configs pass `type=<class>` for the restorer/generator and registry strings
('RDN', 'EDSR', 'MLPRefiner', 'L1Loss') for the rest (configs/001_*_rdn_*.py:12-44)."""
import torch.nn as nn

_REGISTRY = {}


def register(name=None):
    def deco(cls):
        _REGISTRY[name or cls.__name__] = cls
        return cls
    return deco


def build(cfg, **extra):
    if cfg is None:
        return None
    if isinstance(cfg, nn.Module):
        return cfg
    cfg = dict(cfg)
    typ = cfg.pop('type')
    if isinstance(typ, str):
        if typ not in _REGISTRY:
            raise KeyError(f'{typ} is not in the registry; known: {sorted(_REGISTRY)}')
        cls = _REGISTRY[typ]
    else:
        cls = typ
    cfg.update(extra)
    return cls(**cfg)


build_backbone = build_component = build_loss = build_model_from_cfg = build


def build_model(cfg, train_cfg=None, test_cfg=None):
    """tools/test.py:109 `build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)`."""
    return build(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


@register('L1Loss')
class L1Loss(nn.Module):
    """Placeholder for mmedit's L1Loss: the restorer builds it (basic_restorer.py:54) but the
    forward path never evaluates it (training is out of scope)."""

    def __init__(self, loss_weight=1.0, reduction='mean', sample_wise=False):
        super().__init__()
        self.loss_weight = loss_weight
        self.reduction = reduction

    def forward(self, pred, target, **kw):
        raise NotImplementedError('training losses are out of scope of the MI355X inference path')
