"""Pin the CPU oracle against vectors produced by the reference itself (tools/make_golden.py).

Runs without a GPU.  Tolerances: the oracle re-orders nothing relative to the reference (same
ATen ops), so agreement is expected at fp32 round-off (<= 2e-5 abs on O(1) outputs).
"""
import numpy as np
import pytest
import torch

from oracle import ciaosr_oracle as orc
from oracle import per_query as pq
from tests.helpers import load_golden, weights_from, seeded_head, randn, csattn_shapes
from ciaosr_amd.init_utils import seeded_state_dict, state_dict_sha256
from ciaosr_amd.coords import make_coord, make_cell

TOL = 2e-5


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_tiny_head_matches_reference():
    fx = load_golden('tiny_head_s2p7')
    P = weights_from(fx)
    feat, coord, cell = _t(fx['feature']), _t(fx['coord']), _t(fx['cell'])
    nl = orc.cross_scale_attention(feat, P)
    assert torch.allclose(nl, _t(fx['nonlocal_map']), atol=TOL)
    out = orc.batched_predict(feat, coord, cell, P, eval_bsize=int(fx['eval_bsize']))
    assert (out - _t(fx['out'])).abs().max() < TOL
    out_h = orc.batched_predict(feat, coord, cell, P, eval_bsize=int(fx['eval_bsize']), hoist_nonlocal=True)
    assert (out_h - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('tag,kw', [('ls3', dict(local_size=3)), ('ls1', dict(local_size=1)),
                                    ('nonl0', dict(non_local=False)), ('sm2', dict(softmax_scale=2.0)),
                                    ('nounfold', dict(feat_unfold=False, non_local=False))])
def test_tiny_head_variants(tag, kw):
    fx = load_golden('tiny_head_' + tag)
    P = weights_from(fx)
    out = orc.query_rgb(_t(fx['feature']), _t(fx['coord']), _t(fx['cell']), P, **kw)
    assert (out - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('act', ['sin', 'cos'])
def test_tiny_head_sin_cos_activations(act):
    fx = load_golden('tiny_head_act_' + act)
    out = orc.query_rgb(_t(fx['feature']), _t(fx['coord']), _t(fx['cell']), weights_from(fx), act=act)
    assert (out - _t(fx['out'])).abs().max() < TOL


def test_per_query_restatement_matches_reference():
    """The explicit-index numpy form (what the HIP kernels follow) on every query of the tiny case."""
    fx = load_golden('tiny_head_s2p7')
    P = {k: v.numpy() for k, v in weights_from(fx).items()}
    feat = fx['feature'][0]
    nl = fx['nonlocal_map'][0]
    coord, cell, ref = fx['coord'][0], fx['cell'][0], fx['out'][0]
    bs = int(fx['eval_bsize'])
    zero_img = np.zeros((3,) + feat.shape[1:], dtype=np.float32)
    worst = 0.0
    for qi in range(0, coord.shape[0], 3):
        cell0 = cell[(qi // bs) * bs]
        rgb, _ = pq.head_query(feat, nl, zero_img, coord[qi], cell[qi], cell0, P)
        worst = max(worst, float(np.abs(rgb - ref[qi]).max()))
    assert worst < 5e-5, worst


def test_nearest_index_tables():
    """Index rule incl. exact rounding ties vs F.grid_sample-derived tables."""
    fx = load_golden('nearest_idx')
    for key in [k for k in fx if k.startswith('q_')]:
        _, n, nt = key.split('_')
        n, nt = int(n), int(nt)
        seq = make_coord((nt, 1))[:, 0].numpy()
        cell0 = np.float32(2.0 / nt)
        assert np.array_equal(pq.nearest_index(seq, n), fx[key]), key
        km = pq.nearest_index(pq.shifted_coord(seq, cell0, n, -1), n)
        kp = pq.nearest_index(pq.shifted_coord(seq, cell0, n, +1), n)
        assert np.array_equal(km, fx[f'km_{n}_{nt}']), key
        assert np.array_equal(kp, fx[f'kp_{n}_{nt}']), key


@pytest.mark.parametrize('tag', ['10x12', '9x11', '7x8'])
def test_csattn_small(tag):
    fx = load_golden('csattn_c8_' + tag)
    P = weights_from(fx)
    y = orc.cross_scale_attention(_t(fx['x']), P)
    assert (y - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('tag', ['s3', 's4', 's234'])
def test_csattn_other_scales(tag):
    """scale entries 3 / 4 and the list [2, 3, 4] (arch_csnln.py:421-427,:436-528)."""
    fx = load_golden('csattn_c8_' + tag)
    y = orc.cross_scale_attention(_t(fx['x']), weights_from(fx), scales=tuple(int(v) for v in fx['scale']))
    assert (y - _t(fx['out'])).abs().max() < TOL


def test_tiny_head_multi_scale():
    fx = load_golden('tiny_head_ms23')
    out = orc.query_rgb(_t(fx['feature']), _t(fx['coord']), _t(fx['cell']), weights_from(fx), multi_scale=(2, 3))
    assert (out - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('tag', ['48', '45x51', '64x64', '67x70'])
def test_csattn_c64(tag):
    fx = load_golden('csattn_c64_' + tag)
    h, w = [int(v) for v in fx['shape']]
    P = seeded_state_dict(csattn_shapes(64, prefix=''), int(fx['weight_seed']), float(fx['gain']))
    assert state_dict_sha256(P) == str(fx['sha'])
    P = {'cs_attn.' + k: v for k, v in P.items()}
    y = orc.cross_scale_attention(randn((1, 64, h, w), fx['in_seed']), P)
    assert (y[0] - _t(fx['out'])).abs().max() < TOL


@pytest.mark.slow
def test_head_c64_x4():
    fx = load_golden('head_c64_x4')
    P = seeded_head(64, int(fx['weight_seed']))
    feat = randn((1, 64, 48, 48), fx['feat_seed'])
    ht, wt = [int(v) for v in fx['target']]
    coord, cell = make_coord((ht, wt)).unsqueeze(0), make_cell((ht, wt)).unsqueeze(0)
    out = orc.batched_predict(feat, coord, cell, P, eval_bsize=30000, hoist_nonlocal=True)
    assert (out[0] - _t(fx['out'])).abs().max() < 5e-5


def test_head_c64_x3p3_sampled_queries():
    """Non-integer scale: per-query restatement on queries around the rounding ties."""
    fx = load_golden('head_c64_x3p3')
    P = {k: v.numpy() for k, v in seeded_head(64, int(fx['weight_seed'])).items()}
    feat = randn((1, 64, 48, 48), fx['feat_seed'])
    ht, wt = [int(v) for v in fx['target']]
    coord, cell = make_coord((ht, wt)).numpy(), make_cell((ht, wt)).numpy()
    nl = orc.cross_scale_attention(feat, {k: torch.from_numpy(v) for k, v in P.items()})[0].numpy()
    zero = np.zeros((3, 48, 48), np.float32)
    rng = np.random.default_rng(0)
    for qi in rng.choice(ht * wt, 24, replace=False):
        rgb, _ = pq.head_query(feat[0].numpy(), nl, zero, coord[qi], cell[qi], cell[(qi // 30000) * 30000], P)
        assert np.abs(rgb - fx['out'][qi]).max() < 1e-4


@pytest.mark.parametrize('tag,scale', [('e2e_edsr_x2_48', 2), ('e2e_rdn_x4_48', 4)])
@pytest.mark.slow
def test_e2e_restorer(tag, scale):
    import json, os
    from tests.helpers import GOLDEN
    fx = load_golden(tag)
    kind = 'edsr' if 'edsr' in tag else 'rdn'
    names = json.load(open(os.path.join(GOLDEN, f'state_dict_names_{kind}.json')))
    P = seeded_state_dict(names, int(fx['weight_seed']), float(fx['gain']), head_gain=6 ** 0.5)
    assert state_dict_sha256(P) == str(fx['sha'])
    P = {k[len('generator.'):]: v for k, v in P.items()}
    lq = _t(fx['lq'])
    out = orc.forward_test(lq, None, None, P, scale=scale, tile=192, tile_overlap=32, hoist_nonlocal=True)
    assert (out - _t(fx['out'])).abs().max() < 1e-4


@pytest.mark.slow
def test_e2e_full_c3_tile_sampled():
    """The full 192x192 C3 tile of the reference (20 eval_bsize chunks): the oracle's trunk + cs_attn (once; the
    reference recomputes it per chunk with identical results) + head on the stored every-4th-pixel queries."""
    import json, os
    from tests.helpers import GOLDEN
    from ciaosr_amd.init_utils import synthetic_pair
    fx = load_golden('e2e_rdn_x4_tile192')
    names = json.load(open(os.path.join(GOLDEN, 'state_dict_names_rdn.json')))
    P = seeded_state_dict(names, int(fx['weight_seed']), float(fx['gain']), head_gain=6 ** 0.5)
    assert state_dict_sha256(P) == str(fx['sha'])
    P = {k[len('generator.'):]: v for k, v in P.items()}
    lq, _ = synthetic_pair(192, 192, 4)
    mean = torch.tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)
    x = lq - mean
    coord = make_coord((768, 768)).view(768, 768, 2)[::4, ::4].reshape(1, -1, 2).contiguous()
    cell = make_cell((768, 768)).view(768, 768, 2)[::4, ::4].reshape(1, -1, 2).contiguous()
    with torch.no_grad():
        feat = orc.encoder_features(x, P)
        nl = orc.cross_scale_attention(feat, P)
        pred = orc.query_rgb(feat, coord, cell, P, nonlocal_map=nl) + orc.bilinear_residual(x, coord)
    out = (pred + mean.view(1, 1, 3)).clamp(0, 1).view(1, 192, 192, 3).permute(0, 3, 1, 2)
    assert (out - _t(fx['out_s4'])).abs().max() < 2e-4


def test_tiling_small():
    fx = load_golden('tiling_small')
    P = {k[len('generator.'):]: v for k, v in weights_from(fx).items()}
    out = orc.forward_test(_t(fx['lq']), None, None, P, scale=2, tile=48, tile_overlap=16)
    assert (out - _t(fx['out'])).abs().max() < 5e-5


def test_sensitivity_of_fixture():
    """A deliberately wrong head (uniform attention) must move the fixture output by >> 1e-3,
    i.e. the fixture can actually detect a broken kernel."""
    fx = load_golden('tiny_head_s2p7')
    P = weights_from(fx)
    out = orc.query_rgb(_t(fx['feature']), _t(fx['coord']), _t(fx['cell']), P, softmax_scale=1e9)
    assert (out - _t(fx['out'])).abs().max() > 5e-2


def test_full_width_intermediates_match_reference():
    """The oracle's z (local attention output = imnet_q's argument) and its RGB on the 256 sampled queries of `k4_c64_x4` against the
    reference's own intermediates (module hooks on the unmodified reference, tools/make_golden.py::gen_k4_c64): full model widths
    (C = 64: 576 / 640), the sqrt(6)-gain fixture's logits of std ~40."""
    fx = load_golden('k4_c64_x4')
    P = seeded_head(64, int(fx['weight_seed']))
    feat = randn((1, 64, 48, 48), fx['feat_seed'])
    ht, wt = [int(v) for v in fx['target']]
    idx = _t(fx['idx']).long()
    coord, cell = make_coord((ht, wt))[idx].unsqueeze(0), make_cell((ht, wt))[idx].unsqueeze(0)
    out, inter = orc.query_rgb(feat, coord, cell, P, return_intermediates=True)
    assert (inter['z'][0] - _t(fx['z'])).abs().max() < 5e-5 * max(1.0, float(np.abs(fx['z']).max()))
    assert (out[0] - _t(fx['out'])).abs().max() < 1e-4
