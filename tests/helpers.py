"""Shared helpers for the test-suite (fixtures loading, seeded weights by name)."""
import math
import os

import numpy as np
import torch

from ciaosr_amd.init_utils import seeded_state_dict, state_dict_sha256

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SQRT6 = math.sqrt(6.0)


def free_port():
    """A TCP port the OS hands out as free right now, for a torch.distributed rendezvous on 127.0.0.1 (fixed port numbers collide with other
    jobs on a shared box -- and with a previous test's listener still in TIME_WAIT)."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: z[k] for k in z.files}


def weights_from(fix, prefix='w.'):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in fix.items() if k.startswith(prefix)}


def mlp_shapes(prefix, in_dim, out_dim, hidden):
    shapes, last = {}, in_dim
    for n, h in enumerate(list(hidden) + [out_dim]):
        shapes[f'{prefix}.layers.{2 * n}.weight'] = (h, last)
        shapes[f'{prefix}.layers.{2 * n}.bias'] = (h,)
        last = h
    return shapes


def csattn_shapes(C, prefix='cs_attn.'):
    s = {}
    for nm, co in (('conv_match_1', C // 2), ('conv_match_2', C // 2), ('conv_assembly', C)):
        s[f'{prefix}{nm}.0.weight'] = (co, C, 1, 1)
        s[f'{prefix}{nm}.0.bias'] = (co,)
        s[f'{prefix}{nm}.1.weight'] = (1,)
    s[f'{prefix}down.weight'] = (C, C, 3, 3)
    s[f'{prefix}down.bias'] = (C,)
    s[f'{prefix}escape_NaN'] = (1,)
    return s


def head_shapes(C, hidden=(256,) * 4, non_local=True):
    D = 9 * C
    Dn = C if non_local else 0
    s = {}
    s.update(mlp_shapes('imnet_q', D + Dn, 3, hidden))
    s.update(mlp_shapes('imnet_k', D + 4, D, hidden))
    s.update(mlp_shapes('imnet_v', D + Dn + 4, D + Dn, hidden))
    if non_local:
        s.update(csattn_shapes(C))
    return s


def seeded_head(C, seed, hidden=(256,) * 4, gain=1.0, head_gain=SQRT6, non_local=True):
    return seeded_state_dict(head_shapes(C, hidden, non_local), seed, gain, head_gain)


def randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(int(seed)))
