"""MLPRefiner: parameter container with the reference's names and a HIP forward.

Mirrors mmedited/models/components/refiners/mlp_refiner.py:65-102 (and the identical mmedit class
the configs' `type='MLPRefiner'` string resolves to): `layers` = Sequential(Linear, ReLU, ...,
Linear) so state_dict keys are `layers.{0,2,4,...}.{weight,bias}`.  `act='cos'/'sin'` (mlp_refiner.py:81-86; unused by
the configs) run through the staged per-layer route; the fused head kernels are ReLU-only.
"""
import torch
import torch.nn as nn

from . import _lib, hip_ops
from .registry import register


class Cos(nn.Module):
    def forward(self, input):
        return torch.cos(input)


class Sin(nn.Module):
    def forward(self, input):
        return torch.sin(input)


@register('MLPRefiner')
class MLPRefiner(nn.Module):
    def __init__(self, in_dim, out_dim, hidden_list=None, act=None):
        super().__init__()
        self.act = act if act in ('cos', 'sin') else None           # anything else is ReLU, like the reference
        layers, last = [], in_dim
        for hidden in (hidden_list or []):
            layers += [nn.Linear(last, hidden), Cos() if act == 'cos' else Sin() if act == 'sin' else nn.ReLU()]
            last = hidden
        layers.append(nn.Linear(last, out_dim))
        self.layers = nn.Sequential(*layers)
        self.in_dim, self.out_dim = in_dim, out_dim

    def act_code(self):
        return {'cos': _lib.ACT_COS, 'sin': _lib.ACT_SIN}.get(self.act, _lib.ACT_RELU)

    def linears(self):
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    @torch.no_grad()
    def forward(self, x):
        """x [..., in_dim] on the GPU -> [..., out_dim]: one exact-fp32 MFMA GEMM per Linear
        (bias + ReLU fused in the epilogue)."""
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1]).contiguous().float()
        hip_ops.require_gpu(h)
        lin = self.linears()
        for i, l in enumerate(lin):
            w = l.weight
            if w.shape[1] % 4:   # the GEMM wants 16-byte aligned rows
                pad = 4 - w.shape[1] % 4
                w = torch.nn.functional.pad(w, (0, pad))
                h = torch.nn.functional.pad(h, (0, pad))
            h = hip_ops.gemm(h.contiguous(), w.contiguous(), l.bias.contiguous(),
                             act=self.act_code() if i + 1 < len(lin) else _lib.ACT_NONE)
        return h.view(*lead, -1)

    def init_weights(self, pretrained=None, strict=True):
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=strict)
        elif pretrained is not None:
            raise TypeError(f'"pretrained" must be a str or None. But received {type(pretrained)}.')
