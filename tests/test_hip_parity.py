"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Tolerance for fp32: |delta| <= 1e-3 (BASELINE.json north_star); the exact-fp32 MFMA
path is expected to sit 1-2 orders of magnitude below that, so most checks use 1e-4.
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from tests.helpers import (GOLDEN, SQRT6, csattn_shapes, load_golden, randn, seeded_head, weights_from)

pytestmark = pytest.mark.gpu

TOL = 1e-4          # working tolerance of the fp32 path
NORTH_STAR_TOL = 1e-3
GT30_SEED = 30      # seed of the white noise that places GT' 30 dB from the reference output (realistic-quality PSNR gate)
RMS_16BIT = {'bf16': 1.35e-3, 'f16': 3e-4}     # rms |build - reference| bounds on the full C3 tile: measured 1.02e-3 (bf16 pairs: 8-bit activations) + 30 %


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('GPU tests need the MI355X (run them with: python -m pytest tests -m gpu)')
    from ciaosr_amd import _lib
    _lib.load()                       # fail loudly if the extension is missing
    return torch.device('cuda:0')


def _t(a):
    return torch.from_numpy(np.asarray(a))


# ------------------------------------------------------------------------------------------------
# dense contraction
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (257, 130, 76), (1000, 576, 256), (64, 3, 36), (5, 7, 4),
                                   (2304, 598, 288)])
@pytest.mark.parametrize('kn', [False, True])
def test_gemm_matches_torch(dev, M, N, K, kn):
    from ciaosr_amd import hip_ops, _lib
    a = randn((M, K), 1).to(dev)
    ld = (N + 3) // 4 * 4
    b_full = (randn((K, ld), 2) if kn else randn((N, K), 2)).to(dev)
    b = b_full[:, :N] if kn else b_full          # row-strided view: ldb = ld
    bias = randn((N,), 3).to(dev)
    bd = b.double().cpu()
    want = (a.double().cpu() @ (bd if kn else bd.t()) + bias.double().cpu()) * 0.5
    got = hip_ops.gemm(a, b, bias, act=_lib.ACT_NONE, alpha=0.5, b_is_kn=kn)
    assert (got.cpu().double() - want).abs().max() < 2e-5 * math.sqrt(K)
    got_relu = hip_ops.gemm(a, b, bias, act=_lib.ACT_RELU, alpha=0.5, b_is_kn=kn)
    assert (got_relu.cpu().double() - want.clamp(min=0)).abs().max() < 2e-5 * math.sqrt(K)


@pytest.mark.parametrize('M,N,K', [(300, 130, 598), (2304, 64, 625), (128, 256, 17), (64, 36, 1190)])
def test_gemm_kn_ragged_k_ignores_what_lies_behind_b(dev, M, N, K):
    """B in [k][n] form with K % 16 != 0 (cs_attn's attn.V with L = 598, 625, 1190): the k-tile of the [k][n] image travels in the
    buffer load's SCALAR offset, which the descriptor's range check does not include, so the rows k >= K of the last k-tile must be
    masked per lane -- B is a sub-view of a NaN-filled buffer here, and a NaN read there would survive the zeroed A columns."""
    from ciaosr_amd import hip_ops
    lda = (K + 3) // 4 * 4 + 4                         # A too is a view of NaN-filled memory: columns >= K of its rows are poison
    apool = torch.full((M + 8, lda), float('nan'), device=dev)
    apool[:M, :K] = randn((M, K), 21).to(dev)
    a = apool[:M, :K]
    ld = (N + 3) // 4 * 4
    pool = torch.full((K + 64, ld), float('nan'), device=dev)
    pool[:K, :N] = randn((K, N), 22).to(dev)
    b = pool[:K, :N]
    got = hip_ops.gemm(a, b, b_is_kn=True)
    want = a.double().cpu() @ b.double().cpu()
    assert torch.isfinite(got).all()
    assert (got.cpu().double() - want).abs().max() < 2e-5 * math.sqrt(K)


def test_gemm_asymmetric_transpose_detecting(dev):
    """A = I with an asymmetric B catches a row/col swap in the MFMA C-layout."""
    from ciaosr_amd import hip_ops
    n = 160
    a = torch.eye(n, device=dev)
    b = (torch.arange(n * n, dtype=torch.float32, device=dev).view(n, n) % 97) - 3 * torch.arange(n, device=dev).view(n, 1)
    got = hip_ops.gemm(a, b.contiguous())          # I @ b^T
    assert torch.equal(got, b.t().contiguous())
    got_kn = hip_ops.gemm(a, b.contiguous(), b_is_kn=True)
    assert torch.equal(got_kn, b)


def test_gemm_prelu_epilogue(dev):
    from ciaosr_amd import hip_ops, _lib
    a, b, bias = randn((300, 64), 4).to(dev), randn((32, 64), 5).to(dev), randn((32,), 6).to(dev)
    want = torch.nn.functional.prelu(a @ b.t() + bias, torch.tensor([0.25], device=dev))
    got = hip_ops.gemm(a, b, bias, act=_lib.ACT_PRELU, slope=0.25)
    assert (got - want).abs().max() < 1e-4


# ------------------------------------------------------------------------------------------------
# index math (exact)
# ------------------------------------------------------------------------------------------------
def test_nearest_indices_bit_exact(dev):
    """Nearest index of the query and of both shifted/clamped key coordinates must equal the tables
    F.grid_sample produced on the CPU, including the exact rounding ties at non-integer scales."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord
    fx = load_golden('nearest_idx')
    for key in [k for k in fx if k.startswith('q_')]:
        _, n, nt = key.split('_')
        n, nt = int(n), int(nt)
        # queries along the diagonal exercise both axes with the same per-axis sequence
        seq = make_coord((nt, 1))[:, 0]
        coord = torch.stack([seq, seq], -1).contiguous().to(dev)
        cell = torch.full((nt, 2), 2.0 / nt).to(dev)
        q_idx, k_idx, _ = hip_ops.head_indices(coord, cell, n, n, local_size=2)
        q_idx, k_idx = q_idx.cpu().numpy(), k_idx.cpu().numpy()
        tq, tm, tp = fx[key], fx[f'km_{n}_{nt}'], fx[f'kp_{n}_{nt}']
        assert np.array_equal(q_idx, tq * n + tq), key
        assert np.array_equal(k_idx[:, 0], tm * n + tm), key      # (-1,-1)
        assert np.array_equal(k_idx[:, 1], tm * n + tp), key      # (-1,+1)
        assert np.array_equal(k_idx[:, 2], tp * n + tm), key      # (+1,-1)
        assert np.array_equal(k_idx[:, 3], tp * n + tp), key      # (+1,+1)


@pytest.mark.parametrize('ht,wt', [(192, 192), (158, 131), (19, 24), (768, 5)])
def test_device_make_coord_bit_exact(dev, ht, wt):
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    coord, cell = hip_ops.make_coord_cell(ht, wt, dev)
    assert torch.equal(coord.cpu(), make_coord((ht, wt)))
    assert torch.equal(cell.cpu(), make_cell((ht, wt)))


def test_rel_offsets_match_oracle(dev):
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    from oracle import per_query as pq
    H, W, ht, wt = 13, 17, 43, 56
    coord, cell = make_coord((ht, wt)), make_cell((ht, wt))
    _, k_idx, rel = hip_ops.head_indices(coord.to(dev), cell.to(dev), H, W, local_size=3)
    k_idx, rel = k_idx.cpu().numpy(), rel.cpu().numpy()
    c, cl = coord.numpy(), cell.numpy()
    for qi in range(0, ht * wt, 37):
        j = 0
        for sy in (-1, 0, 1):
            for sx in (-1, 0, 1):
                ky = int(pq.nearest_index(pq.shifted_coord(c[qi, 0], cl[0, 0], H, sy), H))
                kx = int(pq.nearest_index(pq.shifted_coord(c[qi, 1], cl[0, 1], W, sx), W))
                assert k_idx[qi, j] == ky * W + kx
                ry = (np.float32(c[qi, 0]) - pq.pixel_centre(ky, H)) * np.float32(H)
                rx = (np.float32(c[qi, 1]) - pq.pixel_centre(kx, W)) * np.float32(W)
                assert rel[qi, j, 0] == ry and rel[qi, j, 1] == rx
                j += 1


# ------------------------------------------------------------------------------------------------
# CrossScaleAttention
# ------------------------------------------------------------------------------------------------
def _my_csattn(C, params, dev, prefix='cs_attn.'):
    from ciaosr_amd import CrossScaleAttention
    att = CrossScaleAttention(channel=C, scale=[2])
    att.load_state_dict({k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)})
    return att.to(dev)


@pytest.mark.parametrize('tag', ['10x12', '9x11', '7x8'])
def test_csattn_small_vs_golden(dev, tag):
    fx = load_golden('csattn_c8_' + tag)
    att = _my_csattn(8, weights_from(fx), dev)
    y = att(_t(fx['x']).to(dev)).cpu()
    assert (y - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('tag', ['48', '45x51'])
def test_csattn_c64_vs_golden(dev, tag):
    from ciaosr_amd.init_utils import seeded_state_dict
    fx = load_golden('csattn_c64_' + tag)
    h, w = [int(v) for v in fx['shape']]
    P = seeded_state_dict(csattn_shapes(64, prefix=''), int(fx['weight_seed']), float(fx['gain']))
    att = _my_csattn(64, P, dev, prefix='')
    y = att(randn((1, 64, h, w), fx['in_seed']).to(dev)).cpu()
    assert (y[0] - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('tag', ['s3', 's4', 's234'])
def test_csattn_other_scales_vs_golden(dev, tag):
    """scale entries 3 and 4 and the list [2, 3, 4] (arch_csnln.py:421-427,:436-528; unused by the configs): reflect mod-pad to
    the scale, (3s)x(3s) value patches, 1/s bilinear downscale, stride-s transposed convolution as a gather, downx3 / downx4,
    channel concatenation over the list."""
    from ciaosr_amd import CrossScaleAttention
    fx = load_golden('csattn_c8_' + tag)
    att = CrossScaleAttention(channel=8, scale=[int(v) for v in fx['scale']])
    att.load_state_dict({k[len('cs_attn.'):]: v for k, v in weights_from(fx).items()})
    y = att.to(dev)(_t(fx['x']).to(dev)).cpu()
    assert y.shape == _t(fx['out']).shape
    assert (y - _t(fx['out'])).abs().max() < TOL


def test_head_multi_scale_vs_golden(dev):
    """multi_scale=[2, 3]: two non-local maps in the value rows (ciaosr_net.py:73-76, :134-137), staged and as-written routes."""
    fx = load_golden('tiny_head_ms23')
    g = _my_generator(8, (32, 32), weights_from(fx), dev, eval_bsize=None, multi_scale=[2, 3])
    feat, coord, cell = _t(fx['feature']).to(dev), _t(fx['coord']).to(dev), _t(fx['cell']).to(dev)
    out = g.query_rgb([feat], coord, cell).cpu()
    assert (out - _t(fx['out'])).abs().max() < TOL
    written = g._head.forward_as_written(feat[0], None, coord[0], cell[0]).cpu()
    assert (written - _t(fx['out'])[0]).abs().max() < TOL


def _csattn_golden(tag, dev):
    from ciaosr_amd.init_utils import seeded_state_dict
    fx = load_golden('csattn_c64_' + tag)
    h, w = [int(v) for v in fx['shape']]
    P = seeded_state_dict(csattn_shapes(64, prefix=''), int(fx['weight_seed']), float(fx['gain']))
    att = _my_csattn(64, P, dev, prefix='')
    return att, randn((1, 64, h, w), fx['in_seed']).to(dev), _t(fx['out'])


@pytest.mark.parametrize('tag', ['64x64', '67x70'])
def test_csattn_big_map_composed_tail_vs_reference(dev, tag):
    """Maps of >= 4096 LR pixels -- the size class of every C3/C4 tile -- take the composed fold+down tail
    (csattn.hip: attn.V' with 16C columns + the row-0 / column-0 edge variants, csa_gather_out).  Reference vectors
    from arch_csnln.py:430-532 at 64x64 and at 67x70 (reflect-pad to 68x70, crop, non-square).  The uncomposed tail
    (fold -> down conv) is forced on the same input as a second route."""
    from ciaosr_amd import hip_ops
    att, x, want = _csattn_golden(tag, dev)
    with hip_ops.profile():
        y = att(x).cpu()
    prof = hip_ops.profile.results()
    assert 'csa_attn_v_edge' in prof and 'csa_down_partial' in prof, sorted(prof)     # the composed branch really ran
    err = (y[0] - want).abs().max().item()
    print(f'cs_attn {tag} composed tail: max|hip - reference| = {err:.3e} (out scale {want.abs().max().item():.3f})')
    assert err < TOL
    with hip_ops.profile():
        y2 = att(x, options=hip_ops.Options(csa_composed_min=-1)).cpu()
    assert 'csa_down' in hip_ops.profile.results() and 'csa_attn_v_edge' not in hip_ops.profile.results()
    assert (y2[0] - want).abs().max().item() < TOL
    # the correlation scores: default = 3x3 diagonal box sum of the per-pixel correlation (csa_scores_f32.hip, no patch rows);
    # csa_scores_gemm = 1 = the 288-wide patch-row GEMM.  Both against the reference, and against each other
    assert 'csa_key_norms' in prof and 'csa_patch_q' not in prof, sorted(prof)
    with hip_ops.profile():
        y3 = att(x, options=hip_ops.Options(csa_scores_gemm=1)).cpu()
    prof3 = hip_ops.profile.results()
    assert 'csa_patch_q' in prof3 and 'csa_key_norms' not in prof3, sorted(prof3)
    assert (y3[0] - want).abs().max().item() < TOL and (y3 - y).abs().max().item() < TOL


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'f16'])
def test_csattn_never_reads_scratch_it_did_not_write(dev, precision):
    """67x70 (L = 34 x 35 = 1190: ragged against every k-tile depth, reflect-padded rows): every scratch buffer of the call is
    filled with NaN bit patterns before the second run.  Same result => no kernel of cs_attn (incl. the softmax-staging attn.V
    GEMM, whose [k][n] operand sits in the middle of the workspace) reads a byte that this call did not write first."""
    from ciaosr_amd import hip_ops
    att, x, want = _csattn_golden('67x70', dev)
    opt = hip_ops.Options(precision)
    y0 = att(x, options=opt).clone()
    hip_ops.poison_workspaces()
    y1 = att(x, options=opt)
    assert torch.isfinite(y1).all() and torch.equal(y0, y1)
    if precision == 'fp32':
        assert (y1.cpu()[0] - want).abs().max().item() < TOL


@pytest.mark.parametrize('hw,channel', [((192, 192), 64), ((190, 187), 64), ((192, 192), 180)])
def test_csattn_attn_v_big_tile_kernel_is_bitwise_the_128_tile_kernel(dev, hw, channel):
    """attn.V at a C3 tile's size runs the 192 x 256 one-workgroup-per-CU kernel (gemm_big_f32.hip; 768 workgroup tiles, K = 9216 = 192 k-tiles
    in a three-buffer pipeline); `Options(csa_attn_tile128=1)` keeps the 128 x 128 kernel.  Same products in the same order: the two must agree
    BITWISE -- on the even size and on one that is reflect-padded (ragged row tiles: 190 x 187 -> 190 x 188 = 35 720 rows = 186.04 row tiles)
    and at the SwinIR head's width (C = 180: N = 2880 = 11.25 column tiles, the last one ragged) -- and again with every scratch byte poisoned
    first (the kernel's clamped last loads and stale-buffer reads must not reach the result)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.nonlocal_attn import CrossScaleAttention
    torch.manual_seed(5)
    att = CrossScaleAttention(channel=channel, scale=2).to(dev)
    x = (randn((1, channel) + hw, 91) * 0.5).to(dev)
    with hip_ops.profile():
        big = att(x).clone()
    assert 'csa_attn_v' in hip_ops.profile.results()
    small = att(x, options=hip_ops.Options(csa_attn_tile128=1)).clone()
    assert torch.isfinite(big).all() and torch.equal(big, small), (big - small).abs().max().item()
    hip_ops.poison_workspaces()
    assert torch.equal(att(x), big)


@pytest.mark.parametrize('tag', ['48', '45x51'])
def test_csattn_composed_tail_forced_on_small_goldens(dev, tag):
    """The composed branch forced (per-call option, no process state) on the < 4096-pixel reference vectors too:
    45x51 pads to 46x52, so the edge variants see both an even and a reflect-padded axis."""
    from ciaosr_amd import hip_ops
    att, x, want = _csattn_golden(tag, dev)
    with hip_ops.profile():
        y = att(x, options=hip_ops.Options(csa_composed_min=1)).cpu()
    assert 'csa_attn_v_edge' in hip_ops.profile.results()
    assert (y[0] - want).abs().max().item() < TOL


def test_csattn_logit_matrix_past_4gib_runs_as_row_blocks(dev):
    """256 x 256 LR pixels: the fp32 logit matrix [65536][16384] is 4.3 GB, more than one buffer descriptor spans.  The contractions over
    it run as row blocks (gemm_f32 / gemm_f32_softmax_a split M; the scores consumers base their descriptor at their own output rows).
    No reference vector at this size (the CPU path needs minutes): the composed tail (softmax in the attn.V staging, V' with 16C columns,
    edge variants) against the uncomposed one (in-place softmax, 36C-column V patches, fold, down convolution) -- two routes that share
    only the scores."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_state_dict
    P = seeded_state_dict(csattn_shapes(64, prefix=''), 11, 1.0)
    att = _my_csattn(64, P, dev, prefix='')
    x = randn((1, 64, 256, 256), 12).to(dev)
    with hip_ops.profile():
        y = att(x).cpu()
    assert 'csa_attn_v_edge' in hip_ops.profile.results()
    z = att(x, options=hip_ops.Options(csa_composed_min=-1)).cpu()
    d = (y - z).abs().max().item()
    print(f'256x256 cs_attn: composed vs uncomposed tail max |delta| {d:.2e} (scale {z.abs().max().item():.2f})')
    assert d < 2e-5 * max(1.0, z.abs().max().item()), d
    del x, y, z
    torch.cuda.empty_cache()


@pytest.mark.parametrize('precision', ['bf16', 'f16'])
@pytest.mark.parametrize('tag', ['64x64', '67x70'])
def test_csattn_bf16_mode_vs_reference(dev, tag, precision):
    """ciaosr_cs_attn_bf16 / _f16 (scores and P.V' on the 16-bit MFMA) against the REFERENCE's output, not against this
    build's own fp32 result.  IEEE half carries 3 more mantissa bits than bf16: its bound is 8x tighter."""
    import math
    from ciaosr_amd import hip_ops
    att, x, want = _csattn_golden(tag, dev)
    with hip_ops.profile():
        y = att(x, options=precision).cpu()
    prof = hip_ops.profile.results()
    assert f'csa_attn_v_{precision}' in prof and f'csa_scores_{precision}' in prof, sorted(prof)
    err = (y[0] - want).abs()
    scale = want.abs().max().item()
    psnr = 10 * math.log10(scale ** 2 / max((err ** 2).mean().item(), 1e-20))
    print(f'{precision} cs_attn {tag}: max|d| vs reference {err.max().item():.3e} (scale {scale:.3f}), PSNR {psnr:.1f} dB')
    # measured (round 3): bf16 max 5.7e-3 / 6.7e-3 of the scale, 62.7 / 61.9 dB; f16 6.3e-4 / 6.8e-4, 80.8 / 80.2 dB
    if precision == 'bf16':
        assert err.max().item() < 0.01 * scale and psnr > 58.0
    else:
        assert err.max().item() < 0.001 * scale and psnr > 76.0


# ------------------------------------------------------------------------------------------------
# head
# ------------------------------------------------------------------------------------------------
def _my_generator(C, hidden, params, dev, act=None, **kw):
    from ciaosr_amd import LocalImplicitSREDSR
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=list(hidden), **({'act': act} if act else {}))
    g = LocalImplicitSREDSR(dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=C, num_blocks=1),
                            mk(4, 3), mk(64, 64), mk(64, 64), **kw)
    missing, unexpected = g.load_state_dict(params, strict=False)
    assert not unexpected and all(not m.startswith(('imnet', 'cs_attn')) for m in missing), (missing, unexpected)
    return g.to(dev).eval()


def test_head_tiny_vs_golden(dev):
    fx = load_golden('tiny_head_s2p7')
    g = _my_generator(8, (32, 32), weights_from(fx), dev, eval_bsize=int(fx['eval_bsize']))
    feat, coord, cell = _t(fx['feature']).to(dev), _t(fx['coord']).to(dev), _t(fx['cell']).to(dev)
    nl = g.cs_attn(feat).cpu()
    assert (nl - _t(fx['nonlocal_map'])).abs().max() < TOL
    out = g.batched_predict([feat], coord, cell).cpu()
    assert (out - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('tag,kw', [('ls3', dict(local_size=3)), ('ls1', dict(local_size=1)),
                                    ('nonl0', dict(non_local_attn=False)), ('sm2', dict(softmax_scale=2)),
                                    ('nounfold', dict(feat_unfold=False, non_local_attn=False))])
def test_head_variants_vs_golden(dev, tag, kw):
    fx = load_golden('tiny_head_' + tag)
    g = _my_generator(8, (32, 32), weights_from(fx), dev, eval_bsize=None, **kw)
    out = g.query_rgb([_t(fx['feature']).to(dev)], _t(fx['coord']).to(dev), _t(fx['cell']).to(dev)).cpu()
    assert (out - _t(fx['out'])).abs().max() < TOL


@pytest.mark.parametrize('act', ['sin', 'cos'])
def test_head_sin_cos_activations_vs_golden(dev, act):
    """MLPRefiner(act='sin' | 'cos') (mlp_refiner.py:81-86; unused by the configs): the staged route (layer-1 hoist, per-layer
    GEMMs with sin / cos epilogues, K4, decode) and the as-written route against the reference's output."""
    from ciaosr_amd import hip_ops
    fx = load_golden('tiny_head_act_' + act)
    g = _my_generator(8, (32, 32), weights_from(fx), dev, act=act, eval_bsize=None)
    feat, coord, cell = _t(fx['feature']).to(dev), _t(fx['coord']).to(dev), _t(fx['cell']).to(dev)
    with hip_ops.profile():
        out = g.query_rgb([feat], coord, cell).cpu()
    assert 'local_attention' in hip_ops.profile.results()        # the fused kernels are ReLU-only
    assert (out - _t(fx['out'])).abs().max() < TOL
    written = g._head.forward_as_written(feat[0], None, coord[0], cell[0]).cpu()
    assert (written - _t(fx['out'])[0]).abs().max() < TOL


def test_head_sensitivity(dev):
    """Self-test of the fixture: breaking the local attention (uniform softmax) must be visible."""
    fx = load_golden('tiny_head_s2p7')
    g = _my_generator(8, (32, 32), weights_from(fx), dev, eval_bsize=200, softmax_scale=1e9)
    out = g.batched_predict([_t(fx['feature']).to(dev)], _t(fx['coord']).to(dev), _t(fx['cell']).to(dev)).cpu()
    assert (out - _t(fx['out'])).abs().max() > 5e-2


@pytest.mark.parametrize('name,C,hw', [('head_c64_x4', 64, 48), ('head_c64_x3p3', 64, 48), ('head_c180_x3p3', 180, 24)])
def test_head_full_width_vs_golden(dev, name, C, hw):
    from ciaosr_amd.coords import make_coord, make_cell
    fx = load_golden(name)
    g = _my_generator(C, (256,) * 4, seeded_head(C, int(fx['weight_seed'])), dev, eval_bsize=30000)
    feat = randn((1, C, hw, hw), fx['feat_seed']).to(dev)
    ht, wt = [int(v) for v in fx['target']]
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    out = g.batched_predict([feat], coord, cell).cpu()
    # These fixtures are deliberately ill-conditioned (sqrt(6)-gain MLPs on N(0,1) features: logits
    # of std ~40 feed the 4-way softmax); the reference's own fp32 result is 9.4e-5 away from an fp64
    # evaluation.  Bounds: 5e-4 against the fp32 reference, 2.5e-4 against fp64 on a query subset.
    err = (out[0] - _t(fx['out'])).abs().max().item()
    assert err < 5e-4, err
    from oracle import ciaosr_oracle as orc
    idx = torch.arange(0, ht * wt, 16)
    P64 = {k: v.double() for k, v in seeded_head(C, int(fx['weight_seed'])).items()}
    torch.set_default_dtype(torch.float64)
    try:
        want = orc.query_rgb(feat.cpu().double(), coord[:, idx].cpu().double(), cell[:, idx].cpu().double(), P64)
    finally:
        torch.set_default_dtype(torch.float32)
    err64 = (out[0, idx].double() - want[0]).abs().max().item()
    assert err64 < 2.5e-4, err64


@pytest.mark.parametrize('precision', ['bf16', 'f16'])
@pytest.mark.parametrize('name,C,hw', [('head_c64_x4', 64, 48), ('head_c180_x3p3', 180, 24)])
def test_head_bf16_mode_full_width_vs_reference(dev, name, C, hw, precision):
    """The 16-bit head kernels (bf16 with weight pairs; IEEE half) at both encoder widths of the configs -- C = 64 (RDN/EDSR: 576/580/644/640) and
    C = 180 (SwinIR, config C5: 1620/1624/1804/1800, ragged 8-column last chunk of the decode input layer) --
    against the REFERENCE's head output on the ill-conditioned sqrt(6)-gain fixtures (logit std ~40)."""
    import math
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    fx = load_golden(name)
    g = _my_generator(C, (256,) * 4, seeded_head(C, int(fx['weight_seed'])), dev, eval_bsize=30000)
    feat = randn((1, C, hw, hw), fx['feat_seed']).to(dev)
    ht, wt = [int(v) for v in fx['target']]
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    with hip_ops.profile():
        out = g.batched_predict([feat], coord, cell, options=precision).cpu()
    prof = hip_ops.profile.results()
    # the chained kernels where a logit table exists (kv; imnet_q follows at C = 64), else the 128-row ones
    assert f'head_kv_fused_{precision}' in prof and any(k.startswith('head_decode_') and k.endswith('_' + precision) for k in prof), sorted(prof)
    ref = _t(fx['out'])
    err = (out[0] - ref).abs()
    scale = ref.abs().max().item()
    psnr = 10 * math.log10(scale ** 2 / max((err ** 2).mean().item(), 1e-20))
    print(f'{precision} head C={C}: max|d| {err.max().item():.3e} (out scale {scale:.3f}), PSNR vs reference {psnr:.1f} dB')
    # measured (round 3): bf16 47.7 dB (C = 64) / 48.4 dB (C = 180), max 0.20 / 0.11 of the scale (isolated flips of the 4-way
    # attention on these logit-std-40 fixtures); f16 62.5 / 63.3 dB, max 0.029 / 0.023 of the scale
    assert psnr > (44.0 if precision == 'bf16' else 58.0) and err.max().item() < (0.3 if precision == 'bf16' else 0.05) * scale


@pytest.mark.parametrize('precision', ['fp32', 'f16', 'f16-pairs', 'f16x3', 'bf16'])
def test_head_c180_trained_like_features_vs_reference(dev, precision):
    """BASELINE config 5's head width (C = 180) on a feature map with trained-like statistics (head_c180_stress_x3p3: per-channel
    log-normal scales, DC offsets, a step edge, std ~12, magnitudes > 120; the reference's own batched_predict) -- logits of std ~2200 in
    front of the 4-way softmax, i.e. a nearly one-hot attention whose argmax the 16-bit modes must not flip.  fp32: |delta| <= 1e-3.
    16-bit modes: the north star's PSNR gate on the head's output against GT' = reference + white noise at 30 dB (the head output is what
    the restorer adds to the bilinear residual: an error here is an error of the image), <= 0.01 dB -- measured and asserted per mode."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    fx = load_golden('head_c180_stress_x3p3')
    g = _my_generator(180, (256,) * 4, seeded_head(180, int(fx['weight_seed']), head_gain=1.0), dev, eval_bsize=30000)
    feat = _t(fx['feature']).to(dev)
    ht, wt = [int(v) for v in fx['target']]
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    ref = _t(fx['out'])
    opt = hip_ops.Options(precision) if precision != 'bf16' else hip_ops.Options('bf16')
    out = g.batched_predict([feat], coord, cell, options=opt).cpu()[0]
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max().item()
    rms = (out - ref).double().pow(2).mean().sqrt().item()
    noise = torch.randn(ref.shape, generator=torch.Generator().manual_seed(GT30_SEED), dtype=torch.float64) * 10 ** (-30 / 20)
    gt30 = ref.double() + noise
    psnr30 = lambda a: -10 * math.log10((a.double() - gt30).pow(2).mean().item())
    d30 = abs(psnr30(out) - psnr30(ref))
    print(f'C=180 trained-like head, {precision}: max|d| {err:.3e} rms {rms:.3e} (out std {ref.std().item():.3f}), PSNR delta at 30 dB {d30:.5f} dB')
    if precision == 'fp32':
        assert err < NORTH_STAR_TOL, err
    else:
        assert d30 <= 0.01, (precision, d30)


@pytest.mark.parametrize('C,hw,target', [(64, (21, 30), (59, 83)), (180, (16, 19), (53, 61))])
def test_half_head_wide_and_narrow_workgroups_agree(dev, C, hw, target):
    """The IEEE-half head kernels come in two cuts: head_fused_h16.hip (128 rows per workgroup, two workgroups per CU; the default)
    and head_fused_wide.hip (256 rows, one per CU; head_route bit HEAD_WIDE_WG).  Same products, same rounding points, so the two
    agree to the fp32 summation order of the logit dot product -- on a ragged query count (last workgroup partly empty), with the
    logit table and with the MFMA output layer of imnet_k (HEAD_NO_LOGIT_TABLE), for f16 and f16-pairs; f16x3 (wide only) sits
    within 1e-4 of the fp32 kernels on the same inputs."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd._lib import HEAD_WIDE_WG, HEAD_NO_LOGIT_TABLE
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(C, (256,) * 4, seeded_head(C, 3, head_gain=2.0), dev, eval_bsize=30000)
    feat = randn((1, C) + hw, 11).to(dev)
    ht, wt = target                          # Q not a multiple of 64
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    x = (randn((1, 3) + hw, 12) * 0.3).to(dev)
    fp32 = g._predict([feat], coord, cell, 30000, x, hip_ops.Options('fp32')).cpu()
    scale = fp32.abs().max().item()
    for prec in ('f16', 'f16-pairs'):
        for route in (0, HEAD_NO_LOGIT_TABLE):
            narrow = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(prec, head_route=route)).cpu()
            wide = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(prec, head_route=route | HEAD_WIDE_WG)).cpu()
            d = (wide - narrow).abs().max().item()
            e = (wide - fp32).abs().max().item()
            print(f'C={C} {prec} route {route}: wide vs narrow {d:.2e}, wide vs fp32 {e:.2e} (scale {scale:.2f})')
            assert torch.isfinite(wide).all() and d < 2e-3 * scale and e < 0.05 * scale, (prec, route, d, e)
    for route in (0, HEAD_NO_LOGIT_TABLE):
        x3 = g._predict([feat], coord, cell, 30000, x, hip_ops.Options('f16x3', head_route=route)).cpu()
        e = (x3 - fp32).abs().max().item()
        print(f'C={C} f16x3 route {route}: vs fp32 {e:.2e}')
        assert e < 1e-4 * max(scale, 1.0), (route, e)


@pytest.mark.parametrize('C,hw,target', [(64, (21, 30), (59, 83)), (64, (48, 48), (192, 192)), (180, (12, 16), (40, 53))])
def test_chained_16bit_head_kernel_vs_the_128_row_kernels(dev, C, hw, target):
    """Round 5: the default 16-bit kv kernel is the weights-stationary, register-chained one (csrc/head_chain_h16.hip; tags
    head_kv_chain_{f16 | pairs_f16 | pairs_bf16 | bf16}); head_route bit HEAD_NO_CHAIN keeps the 128-row kernels (head_fused_h16.hip).
    Same rounding points except the layer-0 tail term (hi + lo pairs through the MFMA instead of fp32 FMAs) and the fp32 summation orders
    of the logit and of z: the two agree far inside the 16-bit modes' own distance to fp32.  Ragged sizes (the target grid does not
    divide into 16 x 4 blocks, Q not a multiple of 64), C = 180 (57 output units: an odd count, the padded tile), every weight form.
    The traversal hint must not change a bit: queries given as a make_coord grid (hinted, walked in 16 x 4 blocks) and as the same
    coordinates in a tensor of their own (index order) give the same output wherever both run the chained kernel."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd._lib import HEAD_NO_CHAIN, HEAD_NO_DECODE_CHAIN
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(C, (256,) * 4, seeded_head(C, 3, head_gain=2.0), dev, eval_bsize=30000)
    feat = randn((1, C) + hw, 11).to(dev)
    ht, wt = target
    own = (make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev))
    hc, hl = hip_ops.make_coord_cell(ht, wt, dev)
    hinted = (hc.unsqueeze(0), hl.unsqueeze(0))
    assert torch.equal(hinted[0], own[0]) and hip_ops.grid_width_of(hinted[0][0]) == wt and hip_ops.grid_width_of(own[0][0]) == 0
    x = (randn((1, 3) + hw, 12) * 0.3).to(dev)
    fp32 = g._predict([feat], own[0], own[1], 30000, x, hip_ops.Options('fp32')).cpu()
    scale = fp32.abs().max().item()
    for prec, kw, tag in (('f16', {}, 'head_kv_chain_f16'), ('f16-pairs', {}, 'head_kv_chain_pairs_f16'), ('bf16', {}, 'head_kv_chain_pairs_bf16'),
                          ('bf16', dict(bf16_single=1), 'head_kv_chain_bf16')):
        old = g._predict([feat], own[0], own[1], 30000, x, hip_ops.Options(prec, head_route=HEAD_NO_CHAIN, **kw)).cpu()
        e_old = (old - fp32).abs().max().item()
        # (imnet_q kept on the 128-row kernel here: the fallback comparison below is bitwise)
        with hip_ops.profile():
            new = g._predict([feat], hinted[0], hinted[1], 30000, x, hip_ops.Options(prec, head_route=HEAD_NO_DECODE_CHAIN, **kw)).cpu()
        prof = hip_ops.profile.results()
        assert tag in prof and prof[tag]['launches'] == 1, (tag, sorted(prof))
        with hip_ops.profile():
            unhinted = g._predict([feat], own[0], own[1], 30000, x, hip_ops.Options(prec, head_route=HEAD_NO_DECODE_CHAIN, **kw)).cpu()
        # imnet_q in the chained form too (the default where Dv is a multiple of 128: C = 64): same products, other fp32 summation orders, the
        # 3-row output layer as a hi + lo pair on the MFMA instead of fp32 FMAs -- as close to the 128-row kernel as the kv kernels are to
        # each other, and no further from fp32
        with hip_ops.profile():
            full = g._predict([feat], hinted[0], hinted[1], 30000, x, hip_ops.Options(prec, **kw)).cpu()
        qtag = tag.replace('head_kv_chain', 'head_decode_chain')
        if C == 64:
            assert qtag in hip_ops.profile.results() and not any(k.startswith('head_decode_fused') for k in hip_ops.profile.results())
            dq = (full - new).abs().max().item()
            e_full = (full - fp32).abs().max().item()
            print(f'    imnet_q chained vs 128-row on the same Z: {dq:.2e}; vs fp32 {e_full:.2e}')
            assert dq < (2e-2 if prec == 'bf16' else 2e-3) * scale, (prec, dq)      # other summation orders flip 16-bit roundings of activations
            assert e_full < 1.5 * e_old + 1e-4 * scale, (prec, e_full, e_old)
        else:
            assert qtag not in hip_ops.profile.results() and torch.equal(full, new)
        d, e_new, e_old = (new - old).abs().max().item(), (new - fp32).abs().max().item(), (old - fp32).abs().max().item()
        print(f'C={C} {target} {prec} {kw}: chained vs 128-row {d:.2e}; vs fp32: chained {e_new:.2e}, 128-row {e_old:.2e} (scale {scale:.2f})')
        assert torch.isfinite(new).all()
        assert d < (2e-2 if prec == 'bf16' else 2e-3) * scale, (prec, d)
        assert e_new < 1.5 * e_old + 1e-4 * scale, (prec, e_new, e_old)
        # index-order traversal: either the chained kernel again (bitwise the hinted result: rows do not depend on their row tile) or --
        # where 8 consecutive queries of a row leave a row tile's 4 x 4 key-pixel window -- the flagged fallback = the 128-row result
        assert torch.equal(unhinted, new) or torch.equal(unhinted, old), (prec, (unhinted - new).abs().max().item())


def test_chained_16bit_head_falls_back_when_a_row_tile_leaves_its_window(dev):
    """Queries in RANDOM order (a shuffled target grid: no traversal hint, eight consecutive queries scattered over the map) put the key
    pixels of a row tile outside its 4 x 4 gather window: the chained kernel raises the launch's flag and the 128-row kernel, launched
    behind it and gated on that flag, redoes the launch -- bitwise the HEAD_NO_CHAIN result, and finite.  (The same happens for target
    grids coarser than the LR map; those are too small to have a logit table and never reach the chained kernel.)"""
    from ciaosr_amd import hip_ops
    from ciaosr_amd._lib import HEAD_NO_CHAIN, HEAD_NO_DECODE_CHAIN
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(64, (256,) * 4, seeded_head(64, 3, head_gain=2.0), dev, eval_bsize=30000)
    feat = randn((1, 64, 24, 32), 11).to(dev)
    perm = torch.randperm(96 * 128, generator=torch.Generator().manual_seed(4))
    coord = make_coord((96, 128))[perm].unsqueeze(0).to(dev)
    cell = make_cell((96, 128))[perm].unsqueeze(0).to(dev)
    x = (randn((1, 3, 24, 32), 12) * 0.3).to(dev)
    for prec in ('f16', 'bf16'):
        old = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(prec, head_route=HEAD_NO_CHAIN))
        with hip_ops.profile():
            new = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(prec, head_route=HEAD_NO_DECODE_CHAIN))
        prof = hip_ops.profile.results()
        chain = [k for k in prof if k.startswith('head_kv_chain')]
        fused = [k for k in prof if k.startswith('head_kv_fused')]
        assert chain and fused and prof[fused[0]]['total_ms'] > 0.2 * prof[chain[0]]['total_ms'], (sorted(prof), 'the fallback did not do the work')
        assert torch.isfinite(new).all() and torch.equal(new, old), (prec, (new - old).abs().max().item())


def test_fused_and_staged_head_paths_agree(dev):
    """The fused kernels (head_kv_fused / head_decode_fused) against the staged per-layer path on the
    same inputs (both through ciaosr_head_forward_f32), including a ragged last workgroup."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(64, (256,) * 4, seeded_head(64, 3, head_gain=2.0), dev, eval_bsize=30000)
    feat = randn((1, 64, 21, 30), 11).to(dev)
    ht, wt = 59, 83                         # Q = 4897: not a multiple of 16 or 64
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    x = (randn((1, 3, 21, 30), 12) * 0.3).to(dev)
    from ciaosr_amd._lib import HEAD_STAGED, HEAD_NO_LOGIT_TABLE, HEAD_TABLE_GEMM
    staged = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(head_route=HEAD_STAGED)).cpu()
    with hip_ops.profile():                  # logit table as the 576-deep GEMM of (q*key) rows
        fused_gemm = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(head_route=HEAD_TABLE_GEMM)).cpu()
    assert 'head_qk_maps' not in hip_ops.profile.results()
    # fused kernels, imnet_k output layer on the MFMA per row
    fused_mfma = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(head_route=HEAD_NO_LOGIT_TABLE)).cpu()
    with hip_ops.profile():                  # default route: fused kernels + logit table (9 rows per LR pixel) as nine Winograd
        fused = g._predict([feat], coord, cell, 30000, x).cpu()     # convolutions of the product maps (C = 64; ragged 8x16 tiles here)
    prof = hip_ops.profile.results()
    assert 'head_kv_fused' in prof and 'head_logit_table_w4' in prof and 'head_qk_maps' in prof, 'fused kernels / logit table did not run'
    tol = 5e-5 * max(1.0, staged.abs().max().item())
    print('fused - staged', (fused - staged).abs().max().item(), 'gemm table - staged', (fused_gemm - staged).abs().max().item(), 'tol', tol)
    assert (fused - staged).abs().max() < tol and (fused_mfma - staged).abs().max() < tol and (fused_gemm - staged).abs().max() < tol


@pytest.mark.parametrize('hw', [(37, 53), (64, 64), (23, 24)])
def test_logit_table_winograd_route_matches_gemm_route(dev, hw):
    """The fp32 logit table built as nine Winograd F(2x2, 3x3) convolutions of the product maps Pi_o = F . shift_o(F)
    (head_ops.hip qk_maps + dense_wino_f32.hip wino_table_f32; default from 512 LR pixels at C = 64) against the 576-deep GEMM of
    (q * key) rows (`head_route` bit CIAOSR_HEAD_TABLE_GEMM): the same sums re-associated through the transform.  Ragged 8x16 tiles in
    both directions, a map that is exactly tiled, and one just above the threshold."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd._lib import HEAD_TABLE_GEMM
    from ciaosr_amd.coords import make_coord, make_cell
    h, w = hw
    g = _my_generator(64, (256,) * 4, seeded_head(64, 5, head_gain=2.0), dev, eval_bsize=30000)
    feat = randn((1, 64, h, w), 21).to(dev)
    ht, wt = 3 * h, 3 * w
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    x = (randn((1, 3, h, w), 22) * 0.3).to(dev)
    with hip_ops.profile():
        wino = g._predict([feat], coord, cell, 30000, x).cpu()
    assert 'head_qk_maps' in hip_ops.profile.results()
    with hip_ops.profile():
        gemm = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(head_route=HEAD_TABLE_GEMM)).cpu()
    assert 'head_qk_maps' not in hip_ops.profile.results()
    # the default is the F(4x4, 3x3) form (dense_wino4_f32.hip, table kernel); CIAOSR_HEAD_TABLE_WINO2 keeps F(2x2)
    from ciaosr_amd._lib import HEAD_TABLE_WINO2
    wino2 = g._predict([feat], coord, cell, 30000, x, hip_ops.Options(head_route=HEAD_TABLE_WINO2)).cpu()
    d, d2 = (wino - gemm).abs().max().item(), (wino2 - gemm).abs().max().item()
    print(f'{h}x{w}: Winograd table vs GEMM table max |delta| F(4x4) {d:.2e}, F(2x2) {d2:.2e} (output scale {gemm.abs().max().item():.2f})')
    assert d2 < 2e-5 * max(1.0, gemm.abs().max().item()), d2
    assert d < 5e-5 * max(1.0, gemm.abs().max().item()), d
    assert not torch.equal(wino, wino2)              # the two forms really are different kernels


def test_logit_table_winograd_route_at_its_largest_map(dev):
    """The Winograd logit-table route at the largest map it takes (256 x 256 LR pixels = the 65536-row chunk its product maps live in;
    one column more falls back to the GEMM route), x2 queries, against the GEMM route.  Head without the non-local branch (its fp32
    logit matrix would be 4.3 GB at this size, past the 4-GiB descriptors of the attn.V contraction: that path raises)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd._lib import HEAD_TABLE_GEMM
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(64, (256,) * 4, seeded_head(64, 6, head_gain=1.5, non_local=False), dev, eval_bsize=30000, non_local_attn=False)
    for h, w, expect_wino in ((256, 256, True), (256, 257, False)):
        feat = randn((1, 64, h, w), 31).to(dev)
        ht, wt = 2 * h, 2 * w
        coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
        with hip_ops.profile():
            a = g._predict([feat], coord, cell, 30000, None).cpu()
        assert ('head_qk_maps' in hip_ops.profile.results()) == expect_wino, (h, w, sorted(hip_ops.profile.results()))
        b = g._predict([feat], coord, cell, 30000, None, hip_ops.Options(head_route=HEAD_TABLE_GEMM)).cpu()
        d = (a - b).abs().max().item()
        print(f'{h}x{w}: default vs GEMM-table route max |delta| {d:.2e}')
        assert d < 2e-5 * max(1.0, b.abs().max().item()), d
        del feat, coord, cell, a, b
        torch.cuda.empty_cache()


def test_as_written_staged_route_vs_golden_and_fused(dev):
    """Third evaluation route: the reference's op order through the staged C entry points with no algebraic
    restructuring (ciaosr_gather_rows_f32 -> ciaosr_mlp_forward_f32 x2 -> ciaosr_local_attention_f32 ->
    ciaosr_mlp_forward_f32 -> ciaosr_decode_residual_f32) against the reference fixture, and against the
    fused path at full width on a ragged query count with eval_bsize chunks."""
    from ciaosr_amd.coords import make_coord, make_cell
    fx = load_golden('tiny_head_s2p7')
    g = _my_generator(8, (32, 32), weights_from(fx), dev, eval_bsize=int(fx['eval_bsize']))
    feat, coord, cell = _t(fx['feature']).to(dev), _t(fx['coord']).to(dev), _t(fx['cell']).to(dev)
    out = g._head.forward_as_written(feat[0], None, coord[0], cell[0], chunk=int(fx['eval_bsize'])).cpu()
    assert (out - _t(fx['out'])[0]).abs().max() < TOL

    g = _my_generator(64, (256,) * 4, seeded_head(64, 3, head_gain=2.0), dev, eval_bsize=2000)
    feat = randn((1, 64, 21, 30), 11).to(dev)
    ht, wt = 59, 83
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    x = (randn((1, 3, 21, 30), 12) * 0.3).to(dev)
    fused = g._predict([feat], coord, cell, 2000, x).cpu()[0]
    written = g._head.forward_as_written(feat[0], x[0], coord[0], cell[0], chunk=2000).cpu()
    assert (fused - written).abs().max() < 5e-5 * max(1.0, written.abs().max().item())


def test_k1_k4_entry_points_vs_reference_intermediates_full_width(dev):
    """The staged entry points at the model's real widths (C = 64: 576 / 580 / 644 / 640) against the REFERENCE's own intermediates of one
    query_rgb call (tests/golden/k4_c64_x4.npz, tools/make_golden.py::gen_k4_c64: module hooks on the unmodified reference; 256 queries
    sampled from the 36 864 of `head_c64_x4`, the four image corners among them):
      K1  ciaosr_gather_rows_f32   inp_k / inp_v == the arguments of imnet_k / imnet_v           (ciaosr_net.py:176-200)  bitwise
          ciaosr_mlp_forward_f32   on the reference's inp_k / inp_v -> wk / wv                    (:202, :205)
      K4  ciaosr_local_attention_f32 on the REFERENCE's wk / wv -> z == the argument of imnet_q   (:203-216)
          ... and the 16-bit K4 entry points on the same operands, within their element type's rounding of wk / wv / z
          the as-written route end to end on the 256 queries == the reference's query_rgb output (:113-224)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    from ciaosr_amd.head_hip import unfold_perm
    fx, fh = load_golden('k4_c64_x4'), load_golden('head_c64_x4')
    C, H, W = 64, 48, 48
    g = _my_generator(C, (256,) * 4, seeded_head(C, int(fx['weight_seed'])), dev, eval_bsize=30000)
    feat = randn((1, C, H, W), fx['feat_seed']).to(dev)
    ht, wt = [int(v) for v in fx['target']]
    idx = _t(fx['idx']).long()
    coord, cell = make_coord((ht, wt))[idx].contiguous().to(dev), make_cell((ht, wt))[idx].contiguous().to(dev)
    nl_ref = _t(fh['nonlocal_map']).to(dev)                       # the reference's cs_attn output: K1 / K4 are isolated from cs_attn's own error
    U = torch.cat([hip_ops.patch_rows(hip_ops.nchw_to_hwc(feat[0].contiguous()), 3, 1, 1, H, W),
                   hip_ops.nchw_to_hwc(nl_ref.contiguous()).view(H * W, C)], dim=1).contiguous()
    perm = unfold_perm(C, dev)                                    # device column d_dev <- reference column perm[d_dev]
    st = g._head.struct()
    nQ = idx.numel()
    # ---- K1
    q_rows, inp_k, inp_v, q_idx, k_idx = hip_ops.gather_rows(U, C, C, coord, cell, H, W, local_size=2)
    ref_k, ref_v = _t(fx['inp_k']).to(dev), _t(fx['inp_v']).to(dev)                 # [64, 4, 580], [64, 4, 644] in the reference's column order
    n1 = ref_k.shape[0]
    mine_k, mine_v = inp_k.view(nQ, 4, -1)[:n1], inp_v.view(nQ, 4, -1)[:n1]
    assert torch.equal(mine_k[..., :576], ref_k[..., :576][..., perm]), 'K1: gathered key rows differ from the reference'
    assert torch.equal(mine_v[..., :576], ref_v[..., :576][..., perm]) and torch.equal(mine_v[..., 576:640], ref_v[..., 576:640])
    d_tail = max((mine_k[..., 576:] - ref_k[..., 576:]).abs().max().item(), (mine_v[..., 640:] - ref_v[..., 640:]).abs().max().item())
    print(f'K1 vs reference: gathered columns bitwise; rel / scale columns max |delta| {d_tail:.2e}')
    assert d_tail <= 1e-6, d_tail
    # ---- imnet_k / imnet_v as staged GEMM chains on the reference's inputs (device column order on both sides)
    wk_ref = _t(fx['wk']).to(dev)[..., perm].reshape(nQ * 4, 576).contiguous()
    wv_ref = torch.cat([_t(fx['wv']).to(dev)[..., :576][..., perm], _t(fx['wv']).to(dev)[..., 576:]], dim=-1).reshape(nQ * 4, 640).contiguous()
    wk = hip_ops.mlp_forward(inp_k, st.k)
    wv = hip_ops.mlp_forward(inp_v, st.v)
    e_k = (wk.view(nQ, 4, -1)[:n1] - wk_ref.view(nQ, 4, -1)[:n1]).abs().max().item()
    e_v = (wv.view(nQ, 4, -1)[:n1] - wv_ref.view(nQ, 4, -1)[:n1]).abs().max().item()
    e_k_all, e_v_all = (wk - wk_ref).abs().max().item(), (wv - wv_ref).abs().max().item()
    print(f'imnet_k / imnet_v vs reference: max |delta| {e_k_all:.2e} / {e_v_all:.2e} (|wk| max {wk_ref.abs().max().item():.1f})')
    assert max(e_k, e_v, e_k_all, e_v_all) < 2e-5 * max(1.0, wk_ref.abs().max().item(), wv_ref.abs().max().item())
    # ---- K4 on the reference's wk / wv
    z_ref = torch.cat([_t(fx['z']).to(dev)[:, :576][:, perm], _t(fx['z']).to(dev)[:, 576:]], dim=-1)
    z = hip_ops.local_attention(U, C, C, q_idx, k_idx, wk_ref, wv_ref, softmax_scale=st.softmax_scale)
    e_z = (z - z_ref).abs().max().item()
    zs = max(1.0, z_ref.abs().max().item())
    print(f'K4 vs reference: max |delta z| {e_z:.2e} (|z| max {zs:.1f})')
    # logits of std ~40 in front of the softmax (sqrt(6)-gain fixture): an fp32 logit carries ~1e-5 of re-association noise
    assert e_z < 5e-5 * zs, e_z
    for half, td, tol in (('bf16', torch.bfloat16, 3e-2), ('f16', torch.float16, 4e-3)):
        z16 = hip_ops.local_attention_16(U, C, C, q_idx, k_idx, wk_ref.to(td), wv_ref.to(td), softmax_scale=st.softmax_scale).float()
        # the same arithmetic on the ROUNDED operands is the checker for the kernel; against the reference itself only the rounding shows
        z_same = hip_ops.local_attention(U, C, C, q_idx, k_idx, wk_ref.to(td).float(), wv_ref.to(td).float(), softmax_scale=st.softmax_scale)
        e_round = (z16 - z_same.to(td).float()).abs().max().item()
        e_ref = (z16 - z_ref).abs().max().item()
        print(f'K4 {half} operands: vs fp32 kernel on the rounded operands {e_round:.2e}, vs reference {e_ref:.2e}')
        assert e_round <= (2.0 ** -8 if half == 'bf16' else 2.0 ** -11) * zs * 1.01 and e_ref < tol * zs, (half, e_round, e_ref)
    # ---- the as-written route end to end (cs_attn computed here, not handed in) against the reference's query_rgb of these queries
    out = g._head.forward_as_written(feat[0], None, coord, cell, chunk=None).cpu()
    e_out = (out - _t(fx['out'])).abs().max().item()
    print(f'as-written route vs reference query_rgb: max |delta| {e_out:.2e}')
    assert e_out < 5e-4, e_out


@pytest.mark.parametrize('half', ['bf16', 'f16'])
def test_staged_route_through_the_16bit_mlp_and_k4_entry_points(dev, half):
    """the staged route with imnet_k / imnet_v on ciaosr_mlp_forward_bf16 / _f16 (every Linear on the 16-bit MFMA GEMM,
    single 16-bit weights, 16-bit activations between layers) and K4 on ciaosr_local_attention_bf16 / _f16 (16-bit wk / wv / z), against
    the fp32 staged route on the same head: within the element type's own distance (bf16-single is the mode that does NOT meet the PSNR
    gate of the product -- these entry points exist for the staged measurements, the product's 16-bit head is the fused route)."""
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(64, (256,) * 4, seeded_head(64, 3, head_gain=2.0), dev, eval_bsize=2000)
    feat = randn((1, 64, 21, 30), 11).to(dev)
    ht, wt = 37, 41
    coord, cell = make_coord((ht, wt)).to(dev), make_cell((ht, wt)).to(dev)
    x = (randn((1, 3, 21, 30), 12) * 0.3).to(dev)
    fp32 = g._head.forward_as_written(feat[0], x[0], coord, cell, chunk=700).cpu()
    got = g._head.forward_as_written(feat[0], x[0], coord, cell, chunk=700, half=half).cpu()
    scale = max(1.0, fp32.abs().max().item())
    d = (got - fp32).abs().max().item()
    print(f'staged {half}: max |delta| vs the fp32 staged route {d:.3e} (scale {scale:.2f})')
    assert torch.isfinite(got).all()
    assert d < (6e-2 if half == 'bf16' else 8e-3) * scale, (half, d)


def test_gather_rows_matches_reference_assembly(dev):
    """K1 alone: inp_k / inp_v / q rows against torch indexing of the same unfold rows (net:145,176-196)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    C, Cn, H, W = 8, 8, 7, 9
    U = randn((H * W, 9 * C + Cn), 50).to(dev)
    ht, wt = 19, 25
    coord, cell = make_coord((ht, wt)).to(dev), make_cell((ht, wt)).to(dev)
    q_rows, inp_k, inp_v, q_idx, k_idx = hip_ops.gather_rows(U, C, Cn, coord, cell, H, W, local_size=2)
    qi, ki, rel = hip_ops.head_indices(coord, cell, H, W, local_size=2)
    assert torch.equal(q_idx, qi) and torch.equal(k_idx, ki)
    kl = k_idx.long().view(-1)
    assert torch.equal(q_rows, U[q_idx.long(), :9 * C])
    assert torch.equal(inp_k[:, :9 * C], U[kl, :9 * C]) and torch.equal(inp_v[:, :9 * C + Cn], U[kl])
    assert torch.equal(inp_k[:, 9 * C:9 * C + 2], rel.view(-1, 2)) and torch.equal(inp_k[:, 9 * C:], inp_v[:, 9 * C + Cn:])
    scale = (cell * torch.tensor([H, W], dtype=torch.float32, device=dev)).repeat_interleave(4, 0)
    assert torch.equal(inp_k[:, 9 * C + 2:], scale)


@pytest.mark.parametrize('precision', ['bf16', 'f16'])
def test_e2e_test_cfg_precision_selects_the_16bit_kernels_vs_reference(dev, precision):
    """`test_cfg.precision` (+ `test_cfg.hip_options`) is what a config file uses to select a 16-bit mode: every 16-bit kernel
    must engage (trunk dense layers, cs_attn contractions, fused head), popping the key restores the fp32 default, and the result
    is held to the REFERENCE's 48x48 output (tests/golden/e2e_rdn_x4_48.npz) through the north star's PSNR gate."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    fx = load_golden('e2e_rdn_x4_48')
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32, hip_options=dict(dense_min_tiles=1, csa_composed_min=1)))
    seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6)
    model = model.to(dev)
    lq = _t(fx['lq']).to(dev)
    model.test_cfg['precision'] = precision
    try:
        assert model.options().precision == precision and model.options().dense_min_tiles == 1
        with hip_ops.profile():
            got = model.restore(lq).cpu()
        prof = hip_ops.profile.results()
    finally:
        model.test_cfg.pop('precision')
    for tag in ('enc_dense', 'csa_attn_v', 'csa_scores', 'head_kv_chain' + ('_pairs' if precision == 'bf16' else ''),
                'head_decode_chain' + ('_pairs' if precision == 'bf16' else '')):
        assert f'{tag}_{precision}' in prof, (tag, sorted(prof))
    assert model.options().precision == 'fp32'
    ref = _t(fx['out'])
    _, gt = synthetic_pair(48, 48, 4)
    d_psnr = abs(psnr_tensors(got, gt, crop_border=4) - psnr_tensors(ref, gt, crop_border=4))
    rms = (got - ref).double().pow(2).mean().sqrt().item()
    print(f'test_cfg.precision={precision}: rms|d| vs reference {rms:.3e}, max {(got - ref).abs().max().item():.3e}, PSNR delta vs GT {d_psnr:.5f} dB')
    assert d_psnr <= 0.01, d_psnr


def test_staged_local_attention_kernel(dev):
    """K4 alone: ciaosr_local_attention_f32 against a direct torch evaluation of net:211-216."""
    from ciaosr_amd import hip_ops
    C, Cn, H, W, Q, J = 16, 16, 9, 11, 700, 4
    U = randn((H * W, 9 * C + Cn), 40).to(dev)
    q_idx = torch.randint(0, H * W, (Q,), generator=torch.Generator().manual_seed(1)).int().to(dev)
    k_idx = torch.randint(0, H * W, (Q, J), generator=torch.Generator().manual_seed(2)).int().to(dev)
    wk, wv = randn((Q * J, 9 * C), 41).to(dev), randn((Q * J, 9 * C + Cn), 42).to(dev)
    z = hip_ops.local_attention(U, C, Cn, q_idx, k_idx, wk, wv, softmax_scale=1.5)
    qv = U[q_idx.long(), :9 * C]
    kv = U[k_idx.long().view(-1)]
    logit = (qv.unsqueeze(1) * (kv[:, :9 * C] * wk).view(Q, J, -1)).sum(-1)
    a = (logit / 1.5).softmax(-1)
    want = (a.unsqueeze(-1) * (kv * wv).view(Q, J, -1)).sum(1)
    assert (z - want).abs().max() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
def test_staged_local_attention_kernel_16bit_operands(dev, dtype):
    """ciaosr_local_attention_bf16 / _f16 (K4 with 16-bit wk / wv / z): exactly the fp32 kernel's arithmetic on the
    widened operands -- z equals the fp32 kernel's result on the SAME (rounded) wk / wv up to its own rounding to 16 bits -- and within
    the element type's precision of the torch evaluation of net:211-216 on the unrounded operands.  J = 4 and 9, a ragged Q."""
    from ciaosr_amd import hip_ops
    td = torch.bfloat16 if dtype == 'bf16' else torch.float16
    for J, C, Cn in ((4, 16, 16), (9, 16, 16), (4, 20, 20)):       # 9C = 144 / 160: 16-byte steps; 9C = 180: the 8-byte form
        H, W, Q = 9, 11, 701
        U = randn((H * W, 9 * C + Cn), 40).to(dev)
        q_idx = torch.randint(0, H * W, (Q,), generator=torch.Generator().manual_seed(1)).int().to(dev)
        q_idx[5] = -1                                              # a query outside the map: zero logits, uniform attention
        k_idx = torch.randint(0, H * W, (Q, J), generator=torch.Generator().manual_seed(2)).int().to(dev)
        wk, wv = randn((Q * J, 9 * C), 41).to(dev), randn((Q * J, 9 * C + Cn), 42).to(dev)
        wk16, wv16 = wk.to(td), wv.to(td)
        z16 = hip_ops.local_attention_16(U, C, Cn, q_idx, k_idx, wk16, wv16, softmax_scale=1.5)
        assert z16.dtype == td
        z32 = hip_ops.local_attention(U, C, Cn, q_idx, k_idx, wk16.float(), wv16.float(), softmax_scale=1.5)
        # (the two kernels are compiled separately: their fp32 sums may differ in the last bit, hence an ulp of the 16-bit type, not bitwise)
        eps = 2.0 ** -8 if dtype == 'bf16' else 2.0 ** -11
        d = (z16.float() - z32).abs().max().item()
        assert d <= eps * max(1.0, z32.abs().max().item()), (dtype, J, d)
        qv = U[q_idx.clamp(min=0).long(), :9 * C] * (q_idx >= 0).unsqueeze(1)
        kv = U[k_idx.long().view(-1)]
        logit = (qv.unsqueeze(1) * (kv[:, :9 * C] * wk).view(Q, J, -1)).sum(-1)
        a = (logit / 1.5).softmax(-1)
        want = (a.unsqueeze(-1) * (kv * wv).view(Q, J, -1)).sum(1)
        assert (z16.float() - want).abs().max() < 8 * eps * max(1.0, want.abs().max().item()), (dtype, J)


# ------------------------------------------------------------------------------------------------
# encoder trunks (implicit-GEMM convolutions)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('hw', [(37, 53), (48, 48), (70, 21), (16, 32)])
def test_rdn_trunk_halo_resident_fp32_dense_layers(dev, hw):
    """The big-map fp32 dense-layer kernel (dense_f32.hip, gather form, 12x12 tiles, K-sliced waves) forced onto small
    ragged maps, against the torch-CPU trunk and against the default scatter-form path."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_
    from oracle import ciaosr_oracle as orc
    model = _restorer('rdn', 4, dev, dict(scale=4))
    seeded_init_(model, seed=21, gain=1.6)
    params = {k[len('generator.'):]: v.detach().clone().cpu() for k, v in model.state_dict().items()}
    x = randn((1, 3) + hw, 77) * 0.3
    want = orc.encoder_features(x, params)
    gen = model.generator.to(dev)
    scatter = gen.gen_feature(x.to(dev))[0].cpu()
    with hip_ops.profile():
        got = gen.gen_feature(x.to(dev), hip_ops.Options(dense_min_tiles=1, dense_direct=1))[0].cpu()
    assert 'enc_dense_gather' in hip_ops.profile.results(), 'halo-resident fp32 dense kernel did not run'
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < 2e-4 * max(scale, 1.0), ((got - want).abs().max().item(), scale)
    assert (got - scatter).abs().max().item() < 2e-4 * max(scale, 1.0)
    # the same layers in Winograd F(2x2, 3x3) form (dense_wino_f32.hip; ragged 8x16 tiles here) ...
    with hip_ops.profile():
        wino = gen.gen_feature(x.to(dev), hip_ops.Options(dense_min_tiles=1, dense_direct=2))[0].cpu()
    assert 'enc_dense_wino' in hip_ops.profile.results(), 'Winograd F(2x2) fp32 dense kernel did not run'
    # ... and in F(4x4, 3x3) form, the default big-map route (dense_wino4_f32.hip; ragged 16x32 tiles here)
    with hip_ops.profile():
        wino4 = gen.gen_feature(x.to(dev), hip_ops.Options(dense_min_tiles=1))[0].cpu()
    assert 'enc_dense_wino4' in hip_ops.profile.results() and 'enc_dense_wino' not in hip_ops.profile.results()
    ew, e4, ed = (wino - want).abs().max().item(), (wino4 - want).abs().max().item(), (got - want).abs().max().item()
    print(f'rdn trunk {hw}: max|d| vs oracle: F(4x4) {e4:.3e}, F(2x2) {ew:.3e}, direct {ed:.3e} (scale {scale:.3f})')
    assert ew < 2e-4 * max(scale, 1.0), (ew, scale)
    assert e4 < 2e-4 * max(scale, 1.0), (e4, scale)
    # third implementation of the same layers: the generic tap-major convolution in scatter form (conv_f32.hip)
    generic = gen.gen_feature(x.to(dev), hip_ops.Options(scatter_small_max=-1))[0].cpu()
    assert (generic - want).abs().max().item() < 2e-4 * max(scale, 1.0)


def _rdn_trunk_bf16_emulation(x, P, nb, nl, single=False, f16=False):
    """torch-CPU RDN trunk with the 16-bit modes' rounding points: dense-layer inputs rounded to bf16, weights as the
    bf16 pair hi + lo (or hi alone when `single`) -- or inputs and weights rounded to IEEE half when `f16`, where the 1x1
    local feature fusion runs on the 16-bit rows too; products are then exact in fp32; fp32 accumulation, everything else fp32."""
    F = torch.nn.functional
    bf = (lambda t: t.half().float()) if f16 else (lambda t: t.bfloat16().float())
    bw = bf if (single or f16) else (lambda t: bf(t) + bf(t - bf(t)))
    sfe1 = F.conv2d(x, P['sfe1.weight'], P['sfe1.bias'], padding=1)
    cur = F.conv2d(sfe1, P['sfe2.weight'], P['sfe2.bias'], padding=1)
    outs = []
    for b in range(nb):
        feats = [cur]
        for l in range(nl):
            inp = torch.cat([bf(f) for f in feats], 1)
            feats.append(F.relu(F.conv2d(inp, bw(P[f'rdbs.{b}.layers.{l}.conv.weight']), P[f'rdbs.{b}.layers.{l}.conv.bias'], padding=1)))
        if f16:     # f16 mode: the 1x1 local feature fusion reads the same 16-bit rows, weights rounded to half; fp32 residual
            cur = cur + F.conv2d(torch.cat([bf(f) for f in feats], 1), bf(P[f'rdbs.{b}.lff.weight']), P[f'rdbs.{b}.lff.bias'])
        else:
            cur = cur + F.conv2d(torch.cat(feats, 1), P[f'rdbs.{b}.lff.weight'], P[f'rdbs.{b}.lff.bias'])
        outs.append(cur)
    g = F.conv2d(torch.cat(outs, 1), P['gff.0.weight'], P['gff.0.bias'])
    return F.conv2d(g, P['gff.1.weight'], P['gff.1.bias'], padding=1) + sfe1


@pytest.mark.parametrize('single', [0, 1, 'f16'])
@pytest.mark.parametrize('hw,blocks,layers,tol', [((37, 53), 1, 1, 1e-4), ((29, 40), 2, 3, 5e-4), ((48, 60), 16, 8, 6e-3)])
def test_rdn_trunk_bf16_dense_layers(dev, hw, blocks, layers, tol, single):
    """ciaosr_rdn_forward_bf16 / _f16 (dense layers on the 16-bit MFMA, dense_h16.hip; ragged 12x12 tiles) against a torch
    emulation with the same rounding points, and its distance from the fp32 trunk.  The two sides round nearly
    equal fp32 activations to bf16, and the rare value that lands on the other side of a rounding boundary (1 bf16
    ulp = 0.4 %) moves a few outputs by ~1e-4 and propagates with depth: the max bound loosens with depth while the
    mean error stays at summation-order level for the one-layer case."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_
    from oracle import ciaosr_oracle as orc
    model = _restorer('rdn', 4, dev, dict(scale=4), blocks=blocks, layers=layers)
    seeded_init_(model, seed=23, gain=1.6)
    params = {k[len('generator.'):]: v.detach().clone().cpu() for k, v in model.state_dict().items()}
    x = randn((1, 3) + hw, 78) * 0.3
    gen = model.generator.to(dev)
    nb, nl = len(gen.rdbs), len(gen.rdbs[0].layers)
    f16 = single == 'f16'
    want = _rdn_trunk_bf16_emulation(x, params, nb, nl, single=bool(single), f16=f16)
    f32 = orc.encoder_features(x, params)
    opt = hip_ops.Options('f16', dense_min_tiles=1) if f16 else hip_ops.Options('bf16', dense_min_tiles=1, bf16_single=single)
    with hip_ops.profile():
        got = gen.gen_feature(x.to(dev), opt)[0].cpu()
    assert ('enc_dense_f16' if f16 else 'enc_dense_bf16') in hip_ops.profile.results(), '16-bit dense kernel did not run'
    scale = want.abs().max().item()
    err, dist = (got - want).abs().max().item(), (got - f32).abs().max().item()
    print(f'{opt} trunk {hw}: max|d| vs emulation {err:.3e}, vs fp32 trunk {dist:.3e} (feature scale {scale:.3f})')
    assert err < tol * scale, (err, scale)
    if blocks == 1:
        assert (got - want).abs().mean().item() < 2e-6 * scale
    assert dist < 5e-2 * scale, (dist, scale)


@pytest.mark.parametrize('kind,hw', [('rdn', (48, 48)), ('edsr', (48, 48)), ('rdn', (37, 53)), ('edsr', (19, 70))])
def test_encoder_features_vs_oracle(dev, kind, hw):
    """HIP gen_feature (split-K implicit GEMM, concat-by-leading-dimension) vs the torch-CPU trunk."""
    from ciaosr_amd.init_utils import seeded_init_
    from oracle import ciaosr_oracle as orc
    model = _restorer(kind, 4, dev, dict(scale=4))
    seeded_init_(model, seed=21, gain=1.6 if kind == 'rdn' else 1.25)
    params = {k[len('generator.'):]: v.detach().clone() for k, v in model.state_dict().items()}
    x = randn((1, 3) + hw, 77) * 0.3
    want = orc.encoder_features(x, params)
    gen = model.generator.to(dev)
    assert gen._encoder_hip.supported()
    got = gen.gen_feature(x.to(dev))[0].cpu()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < 2e-5 * max(scale, 1.0) * 10, ((got - want).abs().max().item(), scale)
    # the PyTorch-ROCm evaluation of the same modules (tests/torch_trunks.py, a checker) must agree too
    from tests import torch_trunks
    with torch.no_grad():
        got_t = (torch_trunks.rdn_features if kind == 'rdn' else torch_trunks.edsr_features)(gen, x.to(dev)).cpu()
    assert (got_t - want).abs().max().item() < 2e-4 * max(scale, 1.0)


@pytest.mark.parametrize('kind,mid,hw', [('edsr', 16, (19, 24)), ('edsr', 40, (33, 20)), ('rdn', 16, (21, 30)), ('rdn', 48, (48, 48))])
def test_narrow_trunk_widths_run_on_the_hip_trunk(dev, kind, mid, hw):
    """Widths that are not multiples of 32 (the tiny fixtures' 8 / 16-channel trunks, mid_channels = 48, ...) run on the HIP trunk
    with zero-padded weights -- there is no PyTorch fallback -- and match the CPU oracle's trunk."""
    from ciaosr_amd import CiaoSR, LocalImplicitSREDSR, LocalImplicitSRRDN, hip_ops
    from ciaosr_amd.init_utils import seeded_init_
    from oracle import ciaosr_oracle as orc
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[32, 32])
    if kind == 'edsr':
        gen = dict(type=LocalImplicitSREDSR, encoder=dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=mid, num_blocks=3),
                   imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=None)
    else:
        gen = dict(type=LocalImplicitSRRDN, encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=mid, num_blocks=3,
                                                          upscale_factor=4, num_layers=4, channel_growth=mid),
                   imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=None)
    model = CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss'), test_cfg=dict(scale=2)).eval()
    seeded_init_(model, seed=5, gain=1.5)
    params = {k[len('generator.'):]: v.detach().clone() for k, v in model.state_dict().items()}
    x = randn((1, 3) + hw, 78) * 0.3
    want = orc.encoder_features(x, params)
    g = model.generator.to(dev)
    assert g._encoder_hip.supported() and g._encoder_hip.width() == (mid, (mid + 31) // 32 * 32)
    with hip_ops.profile():
        got = g.gen_feature(x.to(dev))[0].cpu()
    assert any(t.startswith('enc_') for t in hip_ops.profile.results()), 'HIP trunk did not run'
    assert got.shape == want.shape == (1, mid) + hw
    scale = max(want.abs().max().item(), 1.0)
    assert (got - want).abs().max().item() < 1e-4 * scale, ((got - want).abs().max().item(), scale)


# ------------------------------------------------------------------------------------------------
# restorer end to end
# ------------------------------------------------------------------------------------------------
def _restorer(kind, scale, dev, test_cfg, mid=64, blocks=16, hidden=(256,) * 4, layers=8):
    from ciaosr_amd import CiaoSR, LocalImplicitSREDSR, LocalImplicitSRRDN
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=list(hidden))
    if kind == 'edsr':
        gen = dict(type=LocalImplicitSREDSR,
                   encoder=dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=mid, num_blocks=blocks),
                   imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000)
    else:
        gen = dict(type=LocalImplicitSRRDN,
                   encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=blocks,
                                upscale_factor=4, num_layers=layers, channel_growth=64),
                   imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000)
    return CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'),
                  rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1., 1., 1.), test_cfg=test_cfg).eval()


@pytest.mark.parametrize('tag,kind,scale', [('e2e_edsr_x2_48', 'edsr', 2), ('e2e_rdn_x4_48', 'rdn', 4)])
def test_e2e_restorer_vs_golden(dev, tag, kind, scale):
    """CiaoSR.forward_test (normalise -> clip_test -> generator -> denorm/clamp) vs the reference's
    output; encoder through PyTorch-ROCm.  Also reports the PSNR delta of the north star."""
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    fx = load_golden(tag)
    model = _restorer(kind, scale, dev, dict(scale=scale, tile=192, tile_overlap=32))
    sha = seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6)
    assert sha == str(fx['sha'])
    names = json.load(open(os.path.join(GOLDEN, f'state_dict_names_{kind}.json')))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == names
    model = model.to(dev)
    lq = _t(fx['lq']).to(dev)
    out = model(lq=lq, gt=None, test_mode=True, coord=None, cell=None)['output']
    ref = _t(fx['out'])
    err = (out - ref).abs().max().item()
    assert err < NORTH_STAR_TOL, err
    _, gt = synthetic_pair(48, 48, scale)
    d_psnr = abs(psnr_tensors(out, gt, crop_border=scale) - psnr_tensors(ref, gt, crop_border=scale))
    assert d_psnr <= 0.01, d_psnr


@pytest.mark.parametrize('size', [48, 64])
def test_condition_stress_trained_like_trunk_vs_reference(dev, size):
    """The fp32 default route is not an fmaf chain any more (Winograd F(4x4, 3x3) dense layers with transform constants up to 8 and
    1/24, the logit table as Winograd convolutions of product maps), and every other fixture has Gaussian weights on a smooth image.
    This one has the statistics a TRAINED RDN lives in -- per-output-channel log-normal weight scales, 1 % of the weights x20, bias
    offsets, an input with a DC offset and a step edge, trunk features of std ~10 and magnitude ~100 (tools/make_golden.py gen_stress,
    produced by the unmodified reference's CiaoSR.forward_test) -- and is run with the Winograd routes FORCED on these small maps
    (dense_min_tiles = 1): RGB within the north-star 1e-3 of the reference on every trunk route, trunk features against the fp64
    evaluation reported per route and bounded relative to the feature scale."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_, trained_like_
    fx = load_golden(f'stress_rdn_x4_{size}')
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32))
    seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=float(fx['head_gain']))
    assert trained_like_(model, seed=int(fx['weight_seed']), sigma=float(fx['sigma'])) == str(fx['sha'])
    model = model.to(dev)
    lq = _t(fx['lq']).to(dev)
    ref, f64, f32ref = _t(fx['out']), _t(fx['feat64']), _t(fx['feat'])
    fscale = f64.abs().max().item()
    assert fscale > 50 and f64.std().item() > 5                       # the regime the fixture exists for
    routes = {'wino4 (default on big maps)': dict(dense_min_tiles=1, scatter_small_max=-1),
              'wino2': dict(dense_min_tiles=1, scatter_small_max=-1, dense_direct=2),
              'direct': dict(dense_min_tiles=1, scatter_small_max=-1, dense_direct=1),
              'small-map default (scatter form)': {}}
    print(f'stress {size}x{size}: feature scale {fscale:.1f}; reference fp32 trunk vs fp64 {(f32ref - f64).abs().max().item():.2e}')
    for name, kw in routes.items():
        opt = hip_ops.Options('fp32', **kw)
        with hip_ops.profile():
            feat = model.generator.gen_feature(model.normalize(lq), opt)[0].cpu()
        prof = hip_ops.profile.results()
        want = {'wino4 (default on big maps)': 'enc_dense_wino4', 'wino2': 'enc_dense_wino', 'direct': 'enc_dense_gather',
                'small-map default (scatter form)': 'enc_dense_scatter'}[name]
        assert want in prof, (name, sorted(prof))
        ferr = (feat - f64).abs().max().item()
        out = model.restore(lq, options=opt).cpu()
        err = (out - ref).abs().max().item()
        print(f'  {name:34s} trunk max|d| vs fp64 {ferr:.2e} ({ferr / fscale:.1e} of scale)   RGB max|d| vs reference {err:.2e}')
        assert ferr < 2e-5 * fscale, (name, ferr)
        assert err < NORTH_STAR_TOL, (name, err)
    # the logit table's Winograd form against its GEMM form on the same features
    from ciaosr_amd._lib import HEAD_TABLE_GEMM
    a = model.restore(lq, options=hip_ops.Options('fp32')).cpu()
    b = model.restore(lq, options=hip_ops.Options('fp32', head_route=HEAD_TABLE_GEMM)).cpu()
    print(f'  logit table: Winograd vs GEMM form, RGB max|d| {(a - b).abs().max().item():.2e}')
    assert (b - ref).abs().max().item() < NORTH_STAR_TOL


@pytest.mark.parametrize('precision', ['f16', 'bf16', 'bf16-single', 'f16-pairs'])
@pytest.mark.parametrize('hw', [(150, 170), (192, 192), (24, 1040)])
def test_dense_16bit_wide_tiles_vs_12x12_kernels_and_batch_invariance(dev, precision, hw):
    """The round-6 cut of the 16-bit dense layers (dense_h16_wide_kernel: 16x32-pixel tiles, one persistent 512-thread workgroup per CU,
    weights and halo patch through LDS by DMA, waves split pixels -- no K-slice reduction) against the 12x12-tile kernels it replaces on
    big maps (`dense_direct = 1` keeps them): the same 16-bit products summed in a different order, so the RDN trunk features agree to the
    element type's rounding of the layer outputs -- on a map that is a whole number of tiles, on a ragged one (150 x 170: partial
    tiles in both directions, halo outside the image on every side) and on a strip (24 x 1040: a second tile row that is half empty, 33
    tile columns with an 16-pixel remainder); against the fp32 trunk both sit at the same distance.  And a
    tile's result must not depend on the batch it is computed in: images [A, B, A, B, A] and a batch of seven through one batched call
    (persistent workgroups walking several items; the 8 x 32 tile shape of small launches against the 16 x 32 shape of full ones) are bitwise
    the single-image results."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    h, w = hw
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32), blocks=3)
    seeded_init_(model, seed=2, gain=1.5, head_gain=SQRT6)
    model = model.to(dev)
    enc = model.generator._encoder_hip
    xa = model.normalize(synthetic_pair(h, w, 4)[0].to(dev))[0]
    xb = torch.flip(xa, dims=(1, 2)).contiguous() * 0.8
    new, old = hip_ops.Options(precision, dense_min_tiles=1), hip_ops.Options(precision, dense_min_tiles=1, dense_direct=1)
    with hip_ops.profile():
        fa = enc.forward_hwc(xa, new)
    assert any(k.startswith('enc_dense_') and k.endswith(('_bf16', '_f16')) for k in hip_ops.profile.results()), sorted(hip_ops.profile.results())
    fo = enc.forward_hwc(xa, old)
    f32 = enc.forward_hwc(xa, hip_ops.Options('fp32'))
    scale = f32.abs().max().item()
    d_routes = (fa - fo).abs().max().item()
    d_new, d_old = (fa - f32).abs().max().item(), (fo - f32).abs().max().item()
    rms_new, rms_old = (fa - f32).pow(2).mean().sqrt().item(), (fo - f32).pow(2).mean().sqrt().item()
    print(f'dense {precision} {h}x{w}: wide vs 12x12 max|d| {d_routes:.3e}; vs fp32 trunk: wide max {d_new:.3e} rms {rms_new:.3e}, 12x12 max {d_old:.3e} rms {rms_old:.3e} (scale {scale:.2f})')
    assert torch.isfinite(fa).all() and not torch.equal(fa, fo), 'the two cuts sum in different orders: bitwise equality means one of them did not run'
    half = 'bf16' if precision.startswith('bf16') else 'f16'
    assert d_routes < (4e-2 if half == 'bf16' else 6e-3) * scale, d_routes
    assert rms_new < 1.3 * rms_old + 1e-6 * scale, (rms_new, rms_old)
    assert torch.equal(enc.forward_hwc(xa, new), fa), 'rerun differs'
    fb = enc.forward_hwc(xb, new)
    # five images: the launch still takes the 8 x 32-pixel tile shape of small launches (like the single image); seven images of 192 x 192
    # are 504 items of the 16 x 32 shape -- the other template instantiation, whose outputs must be bitwise the same (the accumulation order
    # of an output does not depend on the tile shape)
    for seq in ((0, 1, 0, 1, 0), (0, 1, 1, 0, 1, 0, 0)):
        want = [(fa, fb)[i] for i in seq]
        batch = enc.forward_hwc_batch(torch.stack([(xa, xb)[i] for i in seq]), new)
        for i in range(len(seq)):
            assert torch.equal(batch[i], want[i]), (len(seq), i, (batch[i] - want[i]).abs().max().item())


def _f16_storage_points_report(model, lq, dev):
    """Where the IEEE-half head STORES 16-bit values (unfold rows U incl. the non-local map, the hidden activations of the three MLPs, the
    attention output z), evaluated with the fp32 staged entry points on the same input: (max magnitude, count above the half range 65 504 =
    conversions that would saturate) per storage point."""
    from ciaosr_amd import hip_ops
    gen = model.generator
    x = model.normalize(lq)
    feat = hip_ops.hwc_to_nchw(gen._encoder_hip.forward_hwc(x[0], None))
    C, H, W = feat.shape
    st = gen._head.struct()
    U = hip_ops.patch_rows(hip_ops.nchw_to_hwc(feat), 3, 1, 1, H, W)
    nl = hip_ops.nchw_to_hwc(gen.cs_attn(feat.unsqueeze(0))[0].contiguous())
    U = torch.cat([U, nl.view(H * W, C)], dim=1).contiguous()
    cc, cl = hip_ops.make_coord_cell(H * 4, W * 4, dev)
    rep = {'U (features + non-local map)': U}
    q_rows, inp_k, inp_v, q_idx, k_idx = hip_ops.gather_rows(U, C, C, cc, cl, H, W, st.local_size)
    for nm, inp, m in (('imnet_k', inp_k, st.k), ('imnet_v', inp_v, st.v)):
        for l in range(1, m.n_layers):
            rep[f'{nm} hidden {l}'] = hip_ops.mlp_forward(inp, m, n_run=l)
    wk, wv = hip_ops.mlp_forward(inp_k, st.k), hip_ops.mlp_forward(inp_v, st.v)
    rep['wk'], rep['wv'] = wk, wv
    z = hip_ops.local_attention(U, C, C, q_idx, k_idx, wk, wv, softmax_scale=st.softmax_scale)
    rep['z (imnet_q input)'] = z
    for l in range(1, st.q.n_layers):
        rep[f'imnet_q hidden {l}'] = hip_ops.mlp_forward(z, st.q, n_run=l)
    return {k: (v.abs().max().item(), int((v.abs() > 65504.0).sum().item())) for k, v in rep.items()}


@pytest.mark.parametrize('precision', ['bf16', 'bf16-single', 'bf16x3', 'f16', 'f16-pairs', 'f16x3', 'f16x3-fast'])
@pytest.mark.parametrize('size', [48, 64])
def test_condition_stress_trained_like_16bit_modes_vs_reference(dev, size, precision):
    """The 16-bit modes on the statistics a TRAINED RDN lives in (stress_rdn_x4_{48,64}: trunk features of std ~10 and magnitude > 100,
    log-normal per-channel weight scales, 1 % of the weights x20, an input with a DC offset and a step edge; output of the unmodified
    reference's CiaoSR.forward_test) -- every other 16-bit gate uses Gaussian weights on a smooth image.  Both routes of the trunk: the product
    default on these small maps and the big-map 16-bit kernels forced (dense_min_tiles = 1, csa_composed_min = 1).  Gates, as on the full C3
    tile: |PSNR(build, GT) - PSNR(ref, GT)| <= 0.01 dB at the fixture's own level and against GT' = reference + white noise at exactly 30 dB;
    f16x3 additionally |delta| <= 1e-3 (the fp32 tolerance it is sold on; measured 7.6e-5 / 1.1e-4).  f16x3-fast does NOT hold that bound here
    (measured 2.7e-2 / 2.4e-2, rms 2.6e-4: half activations in the dense layers on features of magnitude > 100) -- it is documented as a
    PSNR-gated mode since round 6 and asserted as such (a loose isolated-flip bound on the maximum).  The f16 case also reports, from the fp32 staged
    entry points, the largest magnitude at every point where the half head stores 16 bits and how many values exceed the half range
    (= saturating conversions): none may."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_, trained_like_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    fx = load_golden(f'stress_rdn_x4_{size}')
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32))
    seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=float(fx['head_gain']))
    assert trained_like_(model, seed=int(fx['weight_seed']), sigma=float(fx['sigma'])) == str(fx['sha'])
    model = model.to(dev)
    lq = _t(fx['lq']).to(dev)
    ref = _t(fx['out'])
    _, gt = synthetic_pair(size, size, 4)
    noise = torch.randn(ref.shape, generator=torch.Generator().manual_seed(GT30_SEED), dtype=torch.float64) * 10 ** (-30 / 20)
    gt30 = ref.double() + noise
    psnr30 = lambda a: -10 * math.log10((a.double() - gt30).pow(2).mean().item())
    if precision == 'f16':
        rep = _f16_storage_points_report(model, lq, dev)
        print(f'stress {size}: 16-bit storage points of the half head (fp32 staged evaluation): ' +
              '; '.join(f'{k}: max {v[0]:.1f}, > 65504: {v[1]}' for k, v in rep.items()))
        assert all(v[1] == 0 for v in rep.values()), rep
    for route, kw in (('small-map default', {}), ('big-map 16-bit kernels forced', dict(dense_min_tiles=1, csa_composed_min=1))):
        opt = hip_ops.Options(precision, **kw)
        with hip_ops.profile():
            out = model.restore(lq, options=opt).cpu()
        prof = hip_ops.profile.results()
        if kw and precision not in ('f16x3', 'bf16x3'):
            assert any(k.startswith('enc_dense_') and k.endswith(('_bf16', '_f16')) for k in prof), sorted(prof)
        assert any(k.startswith(('head_kv_chain', 'head_kv_fused')) and not k == 'head_kv_fused' for k in prof), sorted(prof)
        assert torch.isfinite(out).all()
        err = (out - ref).abs().max().item()
        rms = (out - ref).double().pow(2).mean().sqrt().item()
        d_fix = abs(psnr_tensors(out, gt, crop_border=4) - psnr_tensors(ref, gt, crop_border=4))
        d_30 = abs(psnr30(out) - psnr30(ref))
        print(f'stress {size} {precision} [{route}]: max|d| {err:.3e} rms {rms:.3e}  PSNR delta at the fixture level {d_fix:.5f} dB, at 30 dB {d_30:.5f} dB')
        assert d_fix <= 0.01 and d_30 <= 0.01, (precision, route, d_fix, d_30)
        if precision in ('f16x3', 'bf16x3'):
            assert err < NORTH_STAR_TOL, (precision, route, err)
        else:
            assert err < 8e-2 and rms < 2.5e-3, (precision, route, err, rms)     # isolated attention flips; the rms is what the PSNR gate sees


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'f16', 'f16-pairs'])
def test_whole_path_never_reads_scratch_it_did_not_write(dev, precision):
    """RDN x4 on a ragged 45x51 LR image (nothing divides the tile sizes of any kernel): second run with every scratch buffer of
    the first filled with NaN bit patterns must be bitwise the first and finite -- trunk, cs_attn, tables, fused head, decode."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32), blocks=3)
    seeded_init_(model, seed=5, gain=1.5, head_gain=SQRT6)
    model = model.to(dev)
    lq = synthetic_pair(45, 51, 4, seed=77)[0].to(dev)
    opt = hip_ops.Options(precision)
    y0 = model.restore(lq, options=opt).clone()
    hip_ops.poison_workspaces()
    y1 = model.restore(lq, options=opt)
    assert torch.isfinite(y1).all() and torch.equal(y0, y1) and float(y1.std()) > 1e-3


@pytest.mark.parametrize('mode', ['whole', 'tiled', 'encoder_ahead', 'tile_streams'])
def test_graphed_restore_replays_bitwise_and_owns_its_scratch(dev, mode):
    """CiaoSR.graphed_restore: one hipGraph of the whole step; replay on a new input is bitwise the eager result.  The graph owns
    every scratch buffer its launches point into -- also those of the restorer's side streams (encoder_ahead / tile_streams): a LARGER
    eager call afterwards (which re-grows scratch on the same streams) must not disturb a later replay."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    cfg = dict(scale=4, tile=192, tile_overlap=32)
    if mode != 'whole':
        cfg.update(tile=32, tile_overlap=8)
    if mode == 'encoder_ahead':
        cfg.update(encoder_ahead=True, tile_batch=2)
    if mode == 'tile_streams':
        cfg.update(tile_streams=2)
    model = _restorer('rdn', 4, dev, cfg, blocks=2)
    seeded_init_(model, seed=6, gain=1.5, head_gain=SQRT6)
    model = model.to(dev)
    lq_a = synthetic_pair(40, 56, 4, seed=80)[0].to(dev)
    lq_b = synthetic_pair(40, 56, 4, seed=81)[0].to(dev)
    want_a, want_b = model.restore(lq_a).clone(), model.restore(lq_b).clone()
    run = model.graphed_restore(lq_a)
    assert torch.equal(run(), want_a)
    assert torch.equal(run(lq_b), want_b)
    # a larger eager call on the same restorer (same cached side streams) grows fresh scratch ...
    big = synthetic_pair(72, 100, 4, seed=82)[0].to(dev)
    model.restore(big)
    hip_ops.poison_workspaces()
    torch.cuda.synchronize()
    # ... and the graph still replays into its own
    assert torch.equal(run(lq_a), want_a) and torch.equal(run(lq_b), want_b)


def _tile192_checks(out, fx, tol):
    """Compare a [1,3,768,768] output with the stored subset of the reference's (tools/make_golden.py tile192_subset)."""
    errs = {
        's4': (out[..., ::4, ::4] - _t(fx['out_s4'])).abs().max().item(),
        'top': (out[..., :8, :] - _t(fx['out_top'])).abs().max().item(),
        'bot': (out[..., -8:, :] - _t(fx['out_bot'])).abs().max().item(),
        'left': (out[..., :, :8] - _t(fx['out_left'])).abs().max().item(),
        'right': (out[..., :, -8:] - _t(fx['out_right'])).abs().max().item()}
    return errs


@pytest.mark.parametrize('precision', ['fp32', 'fp32-wino2', 'fp32-direct', 'bf16', 'bf16-single', 'bf16x3', 'f16', 'f16-pairs', 'f16x3', 'f16x3-fast'])
def test_e2e_full_c3_tile_vs_reference(dev, precision):
    """One full C3 tile: 192x192 LR -> 768x768 through CiaoSR.forward_test (clip_test with one tile; RDN trunk on the
    halo-resident dense kernels, cs_attn on the composed tail, 589 824 queries = 20 reference eval_bsize chunks) against
    the reference's own output (tests/golden/e2e_rdn_x4_tile192.npz: every 4th pixel + an 8-pixel frame + the
    reference's PSNR against the synthetic GT).
      fp32: |delta| <= 1e-3 and |PSNR(build, GT) - PSNR(ref, GT)| <= 0.01 dB   (north star)
      bf16 mode (opt-in extension; bf16 MFMA inputs, every weight as a bf16 hi + lo pair): the same
      PSNR-delta-vs-GT gate, <= 0.01 dB (measured 0.00014 dB), plus a loose max bound (isolated attention flips).
      bf16-single (weights as one bf16, `Options('bf16-single')` = `Options('bf16', bf16_single=1)`): with round-to-nearest weights it
      did NOT meet the gate -- 0.042 dB: rounding the WEIGHTS to 8 bits is a fixed perturbation whose response on smooth RDN
      features is spatially coherent (62 % of it is a 0.42 % change of the network term's amplitude), so it does not average
      out over pixels the way activation rounding does (activation rounding alone: 0.0004 dB).  Round 6 packs the
      head's single-bf16 weights with error-feedback rounding along K and corrects the biases with the calibrated mean input
      of every rounded layer (head_hip.py::_build_single): 0.0038 dB here, one MFMA per product -- asserted INSIDE the gate.
      f16 mode (IEEE half MFMA inputs, ONE MFMA per product like bf16-single): 11 mantissa bits put the weight
      perturbation 8x lower, and the gate holds: <= 0.01 dB asserted (CPU emulation of the rounding points: slope
      -2.0e-4 against bf16-single's -4.2e-3).
      f16-pairs (`Options('f16-pairs')`: half activations, every dense-layer / head weight as a half hi + lo pair, two MFMAs per
      product; fp32 local feature fusion and layer-0 tables): PSNR delta 0.0000 dB at 15 dB and 0.0002 dB at 30 dB, rms 1.3e-4, but
      max |delta| = 1.04e-3 over the stored pixels -- 4 % ABOVE the north star's fp32 bound of 1e-3, so it is NOT an fp32-tolerance
      mode on this ill-conditioned (head gain sqrt 6) vector.  What is left is the 11-bit rounding of the ACTIVATIONS in front of the
      4-way softmax (logit std ~40); W5 through the exact-fp32 table GEMM changed nothing (1.14e-3).  Asserted: < 1.5e-3.
      f16x3 (`Options('f16x3')`, the fp32-tolerance fast mode): the head's weights AND activations as half pairs (three MFMAs per
      product, head_fused_wide.hip), fp32 trunk / tables, half cs_attn contractions: |delta| < 1e-3 on EVERY stored pixel, rms <= 5e-5,
      both PSNR gates (measured: max 2.7e-5, rms 2.1e-6).
      f16x3-fast: the same head on the f16-pairs trunk (half activations in the dense layers): max 4.0e-4, rms 4.6e-5 -- still inside
      both bounds, at 2/3 of the time."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    fx = load_golden('e2e_rdn_x4_tile192')
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32))
    sha = seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6)
    assert sha == str(fx['sha'])
    model = model.to(dev)
    lq, gt = synthetic_pair(192, 192, 4)
    opt = {'fp32': hip_ops.Options('fp32'), 'fp32-wino2': hip_ops.Options('fp32', dense_direct=2), 'fp32-direct': hip_ops.Options('fp32', dense_direct=1), 'bf16': hip_ops.Options('bf16'),
           'bf16-single': hip_ops.Options('bf16', bf16_single=1), 'f16': hip_ops.Options('f16'),
           'f16-pairs': hip_ops.Options('f16-pairs'), 'f16x3': hip_ops.Options('f16x3'), 'f16x3-fast': hip_ops.Options('f16x3-fast'),
           'bf16x3': hip_ops.Options('bf16x3')}[precision]
    with hip_ops.profile():
        out = model.restore(lq.to(dev), options=opt).cpu()
    prof = hip_ops.profile.results()
    if precision in ('f16x3', 'f16x3-fast', 'bf16x3'):
        h = 'bf16' if precision == 'bf16x3' else 'f16'
        for tag in ('enc_dense_f16' if precision == 'f16x3-fast' else 'enc_dense_wino4', f'csa_attn_v_{h}', f'csa_scores_{h}', 'head_logit_table_w4',
                    f'head_kv_fused_{h}x3', f'head_decode_fused_{h}x3'):
            assert tag in prof, (tag, sorted(prof))
    elif precision in ('fp32', 'fp32-wino2', 'fp32-direct'):
        for tag in ({'fp32': 'enc_dense_wino4', 'fp32-wino2': 'enc_dense_wino', 'fp32-direct': 'enc_dense_gather'}[precision], 'csa_attn_v_edge', 'head_logit_table_w4'):
            assert tag in prof, (tag, sorted(prof))
    else:
        sfx = '_f16' if precision.startswith('f16') else '_bf16'
        for tag in ('enc_dense', 'csa_attn_v', 'csa_scores', 'head_kv_fused', 'head_logit_table'):
            assert tag + sfx in prof, (tag + sfx, sorted(prof))
        # the kv kernel that did the work: the chained one (its gated fallback 'head_kv_fused' + sfx is launched behind it and returns at once)
        chain = {'bf16': 'head_kv_chain_pairs_bf16', 'bf16-single': 'head_kv_chain_bf16', 'f16': 'head_kv_chain_f16', 'f16-pairs': 'head_kv_chain_pairs_f16'}[precision]
        assert chain in prof and prof[chain]['total_ms'] > 20 * prof['head_kv_fused' + sfx]['total_ms'], (chain, prof.get(chain), prof['head_kv_fused' + sfx])
        # ... and imnet_q in the same form (C = 64: Dv = 640 is a whole number of Z line pairs)
        assert chain.replace('head_kv_chain', 'head_decode_chain') in prof and 'head_decode_fused' + sfx not in prof, sorted(prof)
    assert out.shape == (1, 3, 768, 768)
    errs = _tile192_checks(out, fx, None)
    psnr_build = psnr_tensors(out, gt, crop_border=4)
    d_psnr = abs(psnr_build - float(fx['psnr_ref_gt']))
    mean_err = abs(out.double().mean().item() - float(fx['out_mean']))
    # (a) rms of the error against the reference over the stored every-4th-pixel grid, and (b) the same gate at a REALISTIC quality
    # level: the synthetic GT sits 15 dB from the random-weight output, where 0.01 dB admits an rms error of 8e-3; a trained model
    # sits near 30 dB, where it admits 1.5e-3.  GT' = reference output + seeded white noise at exactly 30 dB.
    ref_s4 = _t(fx['out_s4'])
    got_s4 = out[..., ::4, ::4]
    rms = (got_s4 - ref_s4).double().pow(2).mean().sqrt().item()
    noise = torch.randn(ref_s4.shape, generator=torch.Generator().manual_seed(GT30_SEED), dtype=torch.float64) * 10 ** (-30 / 20)
    gt30 = ref_s4.double() + noise
    psnr30 = lambda a: -10 * math.log10((a.double() - gt30).pow(2).mean().item())
    d_psnr30 = abs(psnr30(got_s4) - psnr30(ref_s4))
    print(f'tile192 {precision}: max|d| {errs}, rms|d| {rms:.3e}, PSNR(build,GT) {psnr_build:.4f} vs ref {float(fx["psnr_ref_gt"]):.4f} '
          f'(delta {d_psnr:.5f} dB), at 30 dB: PSNR(ref,GT\') {psnr30(ref_s4):.3f}, delta {d_psnr30:.5f} dB, |mean delta| {mean_err:.2e}')
    if precision in ('fp32', 'fp32-wino2', 'fp32-direct'):   # default = Winograd F(4x4) dense layers; 'fp32-wino2' = F(2x2); 'fp32-direct' = the direct kernel
        assert d_psnr <= 0.01 and d_psnr30 <= 0.001, (d_psnr, d_psnr30)
        assert max(errs.values()) < NORTH_STAR_TOL, errs
        assert mean_err < 1e-5 and rms < 1e-5, (mean_err, rms)
    elif precision in ('f16x3', 'f16x3-fast', 'bf16x3'):
        # bf16x3 (round 6): bf16 hi + lo weights AND activations in the head (16 mantissa bits against the half pairs' 22): max 1.8e-4, rms 1.5e-5
        assert d_psnr <= 0.01 and d_psnr30 <= 0.01, (d_psnr, d_psnr30)
        assert max(errs.values()) < NORTH_STAR_TOL, errs          # the fp32 tolerance, on every stored pixel
        assert rms <= 5e-5 and mean_err < 1e-5, (rms, mean_err)
        if precision == 'f16x3':
            assert max(errs.values()) < 1e-4 and rms < 1e-5, (errs, rms)     # fp32 trunk: a decade inside
    elif precision == 'f16-pairs':
        assert d_psnr <= 0.01 and d_psnr30 <= 0.01, (d_psnr, d_psnr30)
        assert max(errs.values()) < 1.5e-3, errs      # measured 1.04e-3: just above the fp32 bound (see the docstring)
        assert rms < 1.7e-4, rms
    elif precision in ('bf16', 'f16'):
        assert d_psnr <= 0.01, d_psnr            # the north-star gate; measured 0.00014 dB (bf16 pairs)
        assert d_psnr30 <= 0.01, d_psnr30        # ... and at a trained model's quality level
        assert rms <= RMS_16BIT[precision], rms
        assert max(errs.values()) < 0.15, errs
    else:
        # bf16-single since round 6: error-feedback rounding along K + calibrated bias correction at pack time (head_hip.py::_build_single);
        # round-to-nearest single weights measured 0.042 dB here, the packed form 0.0038 dB (0.0083 dB at the 30-dB level: the bf16
        # ACTIVATIONS' rms error, the same as with weight pairs)
        assert d_psnr <= 0.01, d_psnr
        assert d_psnr30 <= 0.01, d_psnr30
        assert rms <= 2.0e-3 and max(errs.values()) < 0.2, (rms, errs)


@pytest.mark.parametrize('tag,kind,scale', [('e2e_rdn_x4_48', 'rdn', 4)])
def test_e2e_bf16_mode_psnr_delta_vs_gt_on_reference_golden(dev, tag, kind, scale):
    """The north-star gate for the bf16 mode on the 48x48 reference vector: |PSNR(bf16 build, GT) - PSNR(reference, GT)|
    <= 0.01 dB, with the default routing at this size (bf16 head, fp32 trunk and cs_attn) AND with every bf16 kernel
    forced to engage (per-call options: halo-resident dense layers from 1 tile, composed cs_attn tail from 1 pixel)."""
    from ciaosr_amd import _lib, hip_ops
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    fx = load_golden(tag)
    model = _restorer(kind, scale, dev, dict(scale=scale, tile=192, tile_overlap=32))
    seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6)
    model = model.to(dev)
    ref = _t(fx['out'])
    _, gt = synthetic_pair(48, 48, scale)
    for opt, tags, gate in ((hip_ops.Options('bf16'), ('head_kv_chain_pairs_bf16', 'head_decode_chain_pairs_bf16'), 0.01),
                            (hip_ops.Options('bf16', dense_min_tiles=1, csa_composed_min=1),
                             ('enc_dense_bf16', 'csa_attn_v_bf16', 'head_kv_chain_pairs_bf16', 'head_decode_chain_pairs_bf16'), 0.01),
                            (hip_ops.Options('bf16', head_route=_lib.HEAD_NO_CHAIN), ('head_kv_fused_bf16', 'head_decode_fused_bf16'), 0.01),
                            (hip_ops.Options('f16'), ('head_kv_chain_f16', 'head_decode_chain_f16'), 0.01),
                            (hip_ops.Options('f16', head_route=_lib.HEAD_NO_CHAIN), ('head_kv_fused_f16', 'head_decode_fused_f16'), 0.01),
                            (hip_ops.Options('f16', dense_min_tiles=1, csa_composed_min=1),
                             ('enc_dense_f16', 'csa_attn_v_f16', 'head_kv_chain_f16', 'head_decode_chain_f16'), 0.01)):
        with hip_ops.profile():
            out = model.restore(_t(fx['lq']).to(dev), options=opt).cpu()
        prof = hip_ops.profile.results()
        for t in tags:
            assert t in prof, (t, sorted(prof))
        d_psnr = abs(psnr_tensors(out, gt, crop_border=scale) - psnr_tensors(ref, gt, crop_border=scale))
        print(f'16-bit mode on {tag} {opt}: max|d| vs reference {(out - ref).abs().max().item():.3e}, PSNR delta vs GT {d_psnr:.5f} dB')
        assert d_psnr <= gate, (opt, d_psnr)
    # the fp32-tolerance fast mode holds the fp32 bound itself, default routing and with the half cs_attn contractions forced on
    for opt in (hip_ops.Options('f16x3'), hip_ops.Options('f16x3', csa_composed_min=1)):
        with hip_ops.profile():
            out = model.restore(_t(fx['lq']).to(dev), options=opt).cpu()
        assert 'head_kv_fused_f16x3' in hip_ops.profile.results() and 'head_decode_fused_f16x3' in hip_ops.profile.results()
        err, rms = (out - ref).abs().max().item(), (out - ref).double().pow(2).mean().sqrt().item()
        d_psnr = abs(psnr_tensors(out, gt, crop_border=scale) - psnr_tensors(ref, gt, crop_border=scale))
        print(f'f16x3 on {tag} {opt}: max|d| vs reference {err:.3e}, rms {rms:.3e}, PSNR delta vs GT {d_psnr:.5f} dB')
        assert err < NORTH_STAR_TOL and rms <= 5e-5 and d_psnr <= 0.01, (opt, err, rms, d_psnr)


def test_tile_streams_are_bitwise_the_single_stream_result(dev):
    """clip_test runs consecutive tiles on two HIP streams (own scratch per stream, blend on the caller's stream in the
    reference order): a 6-tile image (C3's DIV2K-val pairing, LR 339x510) must come out bitwise equal to the
    one-stream loop, in fp32, bf16 and f16 mode, and repeatedly (no race on the cached coordinates / packed weights)."""
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    lq, _ = synthetic_pair(339, 510, 4)
    lq = lq.to(dev)
    for precision in ('fp32', 'bf16', 'f16'):
        model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32, tile_streams=1, precision=precision))
        seeded_init_(model, seed=0, gain=1.5, head_gain=SQRT6)
        model = model.to(dev)
        one = model.restore(lq)
        model.test_cfg['tile_streams'] = 2
        two = model.restore(lq)
        assert torch.equal(one, two), (precision, (one - two).abs().max().item())
        assert torch.equal(model.restore(lq), one)
        model.test_cfg['tile_streams'] = 3
        assert torch.equal(model.restore(lq), one)
        # the trunk of tile batch k + 1 on a side stream under the heads of batch k (test_cfg.encoder_ahead), batches of 2 and 8
        model.test_cfg['tile_streams'] = 1
        model.test_cfg['encoder_ahead'] = True
        for nb in (2, 8):
            model.test_cfg['tile_batch'] = nb
            assert torch.equal(model.restore(lq), one), (precision, nb)
            assert torch.equal(model.restore(lq), one), (precision, nb)


def test_tile_batch_is_bitwise_the_one_tile_result(dev):
    """clip_test feeds `test_cfg.tile_batch` consecutive tiles through ONE encoder call (grid.y = image in the dense-layer
    kernels, row-wise 1x1 kernels over all rows; ciaosr_rdn_forward_batch_*): the 6-tile image must come out bitwise equal to the
    one-tile-at-a-time loop for batch 2, 4 (ragged last group 4 + 2) and 6 (the default 8 covers the whole image), in every precision."""
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    lq, _ = synthetic_pair(339, 510, 4)
    lq = lq.to(dev)
    for precision in ('fp32', 'bf16', 'f16'):
        model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32, tile_batch=1, precision=precision))
        seeded_init_(model, seed=0, gain=1.5, head_gain=SQRT6)
        model = model.to(dev)
        one = model.restore(lq)
        for nb in (2, 4, 6):
            model.test_cfg['tile_batch'] = nb
            got = model.restore(lq)
            assert torch.equal(one, got), (precision, nb, (one - got).abs().max().item())


@pytest.mark.parametrize('precision', ['fp32', 'f16'])
def test_encoder_batch_beyond_32bit_offsets_runs_as_sub_batches(dev, precision):
    """A tile batch whose block buffers exceed the 32-bit buffer offsets of the halo-resident dense kernels (default tile_batch = 8 with
    tiles of ~360 pixels and more: 8 x 368 x 368 x 1024 channels x 4 B = 4.4 GB) must run -- as sub-batches that fit -- and every
    feature map must stay bitwise the single-image result (round-2 regression: CIAOSR_ERR_ARG)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.init_utils import seeded_init_
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=368, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.0)
    enc = model.generator.to(dev)._encoder_hip
    opt = hip_ops.Options(precision)
    x = (randn((8, 3, 368, 368), 5) * 0.25).to(dev)
    with hip_ops.profile():
        feats = enc.forward_hwc_batch(x, opt)
    assert ('enc_dense_wino4' if precision == 'fp32' else 'enc_dense_f16') in hip_ops.profile.results()
    assert feats.shape == (8, 368, 368, 64) and bool(torch.isfinite(feats).all())
    for i in (0, 6, 7):
        assert torch.equal(feats[i], enc.forward_hwc(x[i], opt)), i
    del feats
    hip_ops.release_workspaces()
    torch.cuda.empty_cache()


def test_c3_full_image_tiled_restore_properties(dev):
    """C3 itself: the 1356x2040 LR image through restore() (117 tiles of 192, overlap 32 -> 5424x8160).  No CPU
    reference finishes at this size (11 h), so: (1) where a tile's interior is covered by that tile alone (the
    centre 128 LR pixels of a non-border tile; stride 160) the blended image must equal the stand-alone result of
    that tile BITWISE (blend = x/1); (2) an overlap band equals the mean of the two stand-alone tiles that cover it;
    (3) the image is finite and inside [0,1].  The stand-alone tile is the path pinned to the reference by
    test_e2e_full_c3_tile_vs_reference."""
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.restorer import tile_grid
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.5, head_gain=SQRT6)
    model = model.to(dev)
    lq, _ = synthetic_pair(1356, 2040, 4)
    lq = lq.to(dev)
    out = model.restore(lq)
    assert out.shape == (1, 3, 5424, 8160) and torch.isfinite(out).all()
    assert out.min().item() >= 0.0 and out.max().item() <= 1.0
    tile, origins = tile_grid(1356, 2040, 192, 32)
    assert len(origins) == 117
    x = model.normalize(lq)
    from ciaosr_amd import hip_ops

    def alone(hi, wi):
        pred, (th, tw) = model.run_tile(x, hi, wi, tile, 4)
        return hip_ops.denorm_clamp(pred[0].contiguous(), th, tw, model.rgb_mean, model.rgb_std)

    # (1) interior of tile (row 3, col 5): origin (480, 800)
    hi, wi = 480, 800
    assert (hi, wi) in origins
    a = alone(hi, wi)
    ys, xs = slice((hi + 32) * 4, (hi + 160) * 4), slice((wi + 32) * 4, (wi + 160) * 4)
    assert torch.equal(out[0, :, ys, xs], a[:, 32 * 4:160 * 4, 32 * 4:160 * 4])
    # (2) the 32-pixel band shared with the right-hand neighbour (origin wi + 160), rows interior to both
    b = alone(hi, wi + 160)
    band = out[0, :, ys, (wi + 160) * 4:(wi + 192) * 4]
    # blend happens on the un-clamped predictions; compare where neither tile saturates
    mean2 = (a[:, 32 * 4:160 * 4, 160 * 4:192 * 4] + b[:, 32 * 4:160 * 4, 0:32 * 4]) / 2
    ok = (a[:, 32 * 4:160 * 4, 160 * 4:192 * 4] > 0) & (a[:, 32 * 4:160 * 4, 160 * 4:192 * 4] < 1) & \
         (b[:, 32 * 4:160 * 4, 0:32 * 4] > 0) & (b[:, 32 * 4:160 * 4, 0:32 * 4] < 1)
    assert ok.float().mean().item() > 0.5
    assert ((band - mean2).abs() * ok).max().item() < 1e-6


def test_c3_full_image_16bit_modes_meet_the_psnr_gate(dev):
    """C3 itself in the two opt-in 16-bit modes: the whole 1356x2040 LR image (117 tiles), sqrt(6)-gain head and gain-1.5 trunk as in the
    full-tile reference vector.  The fp32 image is the pinned path (tile golden + the tiled-restore properties above); the bf16 (weight
    pairs) and f16 images must be within the north-star gate of it, measured exactly as `evaluate` does (uint8, Y channel, crop 4):
    |PSNR(mode, GT) - PSNR(fp32, GT)| <= 0.01 dB."""
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32))
    seeded_init_(model, seed=0, gain=1.5, head_gain=SQRT6)
    model = model.to(dev)
    lq, gt = synthetic_pair(1356, 2040, 4)
    lq = lq.to(dev)
    ref = psnr_tensors(model.restore(lq).cpu(), gt, crop_border=4)
    for mode in ('f16', 'bf16'):
        out = model.restore(lq, options=mode).cpu()
        assert torch.isfinite(out).all()
        d = abs(psnr_tensors(out, gt, crop_border=4) - ref)
        print(f'C3 full image {mode}: PSNR delta vs the fp32 image {d:.5f} dB (fp32 PSNR vs GT {ref:.4f})')
        assert d <= 0.01, (mode, d)


def test_tiling_vs_golden(dev):
    fx = load_golden('tiling_small')
    model = _restorer('edsr', 2, dev, dict(scale=2, tile=48, tile_overlap=16), mid=16, blocks=2, hidden=(64, 64))
    model.load_state_dict(weights_from(fx))
    model = model.to(dev)
    out = model(lq=_t(fx['lq']).to(dev), gt=None, test_mode=True)['output']
    assert (out - _t(fx['out'])).abs().max() < 2e-4


def test_whole_image_path_non_integer_scale_vs_oracle(dev):
    """No tiling (scale > 4 style path, ciaosr.py:158) at x3.3 against the oracle run here."""
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'f16'])
def test_swinir_e2e_vs_golden(dev, precision):
    """Config C5: SwinIR-CiaoSR x3.3, whole-image path (non-integer scale), HIP SwinIR trunk + C = 180 head through
    the fused kernels.  Reference output from tests/golden/swinir_c5.npz.  fp32: |delta| <= 1e-3; both modes: PSNR delta
    vs GT <= 0.01 dB (bf16 mode = bf16 head with weight pairs, f16 mode = IEEE-half head; the launch-bound SwinIR trunk stays fp32)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    from tests.test_host_logic import _swinir_ciaosr
    fx = load_golden('swinir_c5')
    model = _swinir_ciaosr(dict(scale=3.3))
    assert seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6) == str(fx['sha'])
    model = model.to(dev)
    ht, wt = [int(v) for v in fx['target']]
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    model.test_cfg['precision'] = precision
    if precision == 'bf16':
        # the 8-bit-activation bf16 forms miss the gate on this head: 'bf16' runs as 'bf16x3' (bf16 hi + lo weights AND activations) ...
        eff = model.generator.effective_options('bf16')
        assert eff.precision == 'bf16' and eff.f16_pairs == 2
        # ... unless the caller opts in to the rounds-4/5 substitution by the IEEE-half kernels
        model.test_cfg['allow_f16_substitute'] = True
        assert model.generator.effective_options('bf16').precision == 'f16'
        model.test_cfg.pop('allow_f16_substitute')
    with hip_ops.profile():
        out = model(lq=_t(fx['lq']).to(dev), gt=None, test_mode=True, coord=coord, cell=cell)['output']
    assert {'fp32': 'head_kv_fused', 'bf16': 'head_kv_fused_bf16x3', 'f16': 'head_kv_chain_f16'}[precision] in hip_ops.profile.results()
    assert not any(k.startswith('head_kv') and k.endswith(('_bf16', 'pairs_bf16')) for k in hip_ops.profile.results())      # no 8-bit-activation kernel
    ref = _t(fx['out'])
    err = (out - ref).abs().max().item()
    _, gt = synthetic_pair(24, 24, 3.3)
    d_psnr = abs(psnr_tensors(out, gt, crop_border=3) - psnr_tensors(ref, gt, crop_border=3))
    print(f'C5 {precision}: max|d| {err:.3e}, PSNR delta vs GT {d_psnr:.5f} dB')
    if precision == 'fp32':
        assert err < NORTH_STAR_TOL, err
    assert d_psnr <= 0.01, d_psnr


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16-f16-substitute', 'f16', 'f16x3'])
def test_swinir_c5_at_its_own_size_vs_reference(dev, precision):
    """BASELINE config 5 at ITS size: SwinIR-CiaoSR x3.3, LR 48x48 -> 158x158 (Q = 24 964, C = 180), against the reference's
    CiaoSR.forward_test output and the reference trunk's features (tests/golden/swinir_c5_48.npz).  48 = 6 windows of 8, so the
    shifted blocks use their own `attn_mask` buffers (swinir_net.py:233-236) -- the 24x24 fixture takes `calculate_mask`.
      fp32: trunk features <= 2e-4 * scale, output |delta| <= 1e-3, PSNR delta vs GT <= 0.01 dB
      f16 / f16x3: PSNR delta vs GT <= 0.01 dB at the fixture's own level AND against GT' = reference + 30 dB noise; f16x3 also
      the fp32 bound itself.  precision='bf16' (BASELINE's name for this config): the 8-bit-activation bf16 forms measured 0.060 dB at
      30 dB here, six times the gate, so the generator runs 'bf16' as 'bf16x3' -- bf16 hi + lo weights AND activations in the head, three
      MFMAs per product (round 6): measured max |delta| 9.3e-6, held to the fp32 bound like f16x3.  'bf16-f16-substitute': the rounds-4/5
      opt-in (`test_cfg.allow_f16_substitute`) that runs the IEEE-half kernels instead, with one warning."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd.metrics import psnr_tensors
    from tests.test_host_logic import _swinir_ciaosr
    fx = load_golden('swinir_c5_48')
    model = _swinir_ciaosr(dict(scale=3.3))
    assert seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6) == str(fx['sha'])
    model = model.to(dev)
    ht, wt = [int(v) for v in fx['target']]
    assert (ht, wt) == (158, 158)
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    lq = _t(fx['lq']).to(dev)
    if precision == 'fp32':
        feat = model.generator.gen_feature(model.normalize(lq))[0][0].cpu()
        want = _t(fx['feat_s2'])
        ferr = (feat[:, ::2, ::2] - want).abs().max().item()
        print(f'C5 48x48 trunk: max|d| vs reference features {ferr:.3e} (scale {want.abs().max().item():.3f})')
        assert ferr < 2e-4 * max(want.abs().max().item(), 1.0), ferr
    substitute = precision == 'bf16-f16-substitute'
    model.test_cfg['precision'] = 'bf16' if substitute else precision
    import warnings
    from ciaosr_amd.implicit_net import LocalImplicitSRSWINIR
    LocalImplicitSRSWINIR._warned_bf16 = False
    if substitute:
        model.test_cfg['allow_f16_substitute'] = True
    with hip_ops.profile(), warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        out = model(lq=lq, gt=None, test_mode=True, coord=coord, cell=cell)['output']
    prof = hip_ops.profile.results()
    assert 'swin_window_attention' in prof
    assert {'fp32': 'head_kv_fused', 'bf16': 'head_kv_fused_bf16x3', 'bf16-f16-substitute': 'head_kv_chain_f16', 'f16': 'head_kv_chain_f16',
            'f16x3': 'head_kv_fused_f16x3'}[precision] in prof
    assert not any(k.startswith('head_kv') and k.endswith(('_bf16', 'pairs_bf16')) for k in prof)       # never an 8-bit-activation kernel
    assert substitute == any('does not meet the 0.01 dB PSNR gate' in str(c.message) for c in caught)
    ref = _t(fx['out'])
    err = (out - ref).abs().max().item()
    rms = (out - ref).double().pow(2).mean().sqrt().item()
    _, gt = synthetic_pair(48, 48, 3.3)
    psnr_ref = psnr_tensors(ref, gt, crop_border=3)
    assert abs(psnr_ref - float(fx['psnr_ref_gt'])) < 1e-6
    d_psnr = abs(psnr_tensors(out, gt, crop_border=3) - psnr_ref)
    gt30 = ref.double() + torch.randn(ref.shape, generator=torch.Generator().manual_seed(GT30_SEED), dtype=torch.float64) * 10 ** (-30 / 20)
    psnr30 = lambda a: -10 * math.log10((a.double() - gt30).pow(2).mean().item())
    d_psnr30 = abs(psnr30(out) - psnr30(ref))
    print(f'C5 48x48 {precision}: max|d| {err:.3e}, rms {rms:.3e}, PSNR delta vs GT {d_psnr:.5f} dB, at 30 dB {d_psnr30:.5f} dB')
    if precision in ('fp32', 'f16x3', 'bf16'):
        assert err < NORTH_STAR_TOL, err
    if precision in ('f16x3', 'bf16'):
        assert rms <= 5e-5, rms
    assert d_psnr <= 0.01, d_psnr
    assert d_psnr30 <= 0.01, d_psnr30


def test_tools_test_cli_end_to_end(dev, tmp_path, capsys):
    """tools/test.py CONFIG CHECKPOINT on a folder of synthetic PNGs: Eval-PSNR equals the CPU oracle's
    pipeline (file decode -> restorer -> tensor2img -> Y-channel PSNR with crop_border) within 0.01 dB."""
    import sys
    from ciaosr_amd.imageio import imwrite
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd import metrics
    from ciaosr_amd.dataset import SRFolderDataset
    from oracle import ciaosr_oracle as orc
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), '..'))
    import tools.test as cli
    (tmp_path / 'lq').mkdir(); (tmp_path / 'gt').mkdir()
    for i, (h, w) in enumerate([(40, 56), (64, 48)]):
        lq, gt = synthetic_pair(h, w, 4, seed=100 + i)
        imwrite(metrics.tensor2img(lq), str(tmp_path / 'lq' / f'img{i}.png'))
        imwrite(metrics.tensor2img(gt), str(tmp_path / 'gt' / f'img{i}.png'))
    cfg_path = tmp_path / 'cfg.py'
    cfg_path.write_text(
        "from mmedited.models.restorers.ciaosr import CiaoSR\n"
        "from mmedited.models.backbones.sr_backbones.ciaosr_net import LocalImplicitSREDSR\n"
        "mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[256, 256, 256, 256])\n"
        "model = dict(type=CiaoSR, generator=dict(type=LocalImplicitSREDSR, encoder=dict(type='EDSR', in_channels=3,"
        " out_channels=3, mid_channels=64, num_blocks=4), imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64),"
        " feat_unfold=True, eval_bsize=30000), rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1., 1., 1.),"
        " pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'))\n"
        "test_cfg = dict(metrics=['PSNR', 'SSIM'], crop_border=4, scale=4, tile=32, tile_overlap=8, convert_to='y')\n"
        f"data = dict(test=dict(type='SRFolderDataset', lq_folder={str(tmp_path / 'lq')!r}, gt_folder={str(tmp_path / 'gt')!r},"
        " scale=4, filename_tmpl='{}'))\n"
        "dist_params = dict(backend='nccl')\n")
    import ciaosr_amd
    from ciaosr_amd.config import Config
    cfg = Config.fromfile(str(cfg_path))
    model = ciaosr_amd.build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    seeded_init_(model, seed=8, gain=1.25, head_gain=2.0)
    torch.save({'state_dict': model.state_dict()}, tmp_path / 'ck.pth')
    results = cli.main([str(cfg_path), str(tmp_path / 'ck.pth'), '--save-path', str(tmp_path / 'out')])
    printed = capsys.readouterr().out
    assert 'Eval-PSNR' in printed and 'Eval-SSIM' in printed
    assert os.path.exists(tmp_path / 'out' / 'img0.png') and os.path.exists(tmp_path / 'out' / 'img1.png')
    P = {k[len('generator.'):]: v.detach().cpu() for k, v in model.state_dict().items()}
    ds = SRFolderDataset(tmp_path / 'lq', tmp_path / 'gt', scale=4)
    for i in range(len(ds)):
        d = ds[i]
        want = orc.forward_test(d['lq'].unsqueeze(0), None, None, P, scale=4, tile=32, tile_overlap=8, hoist_nonlocal=True)
        h, w = want.shape[-2:]
        gt = d['gt'].view(1, h, w, 3).permute(0, 3, 1, 2)
        ref_psnr = metrics.psnr(metrics.tensor2img(want), metrics.tensor2img(gt), crop_border=4, convert_to='y')
        assert abs(results[i]['eval_result']['PSNR'] - ref_psnr) <= 0.01, (results[i]['eval_result'], ref_psnr)


@pytest.mark.parametrize('head_gain', [1.0, 2.0])
def test_bf16_head_mode_vs_fp32(dev, head_gain):
    """Precision mode 1 (bf16 MFMA inputs, fp32 accumulate) of the fused head against the exact-fp32 mode on the
    same inputs.  bf16 carries 8 mantissa bits, so this is a PSNR-class check, not the 1e-3 contract:
    >= 45 dB PSNR of the bf16 head output against the fp32 output (measured: 99.7 dB at default-init scale, 53.5 dB
    with gain-2 MLPs whose large logits let a bf16 rounding flip an attention weight now and then: max |d| 6e-2)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(64, (256,) * 4, seeded_head(64, 5, head_gain=head_gain), dev, eval_bsize=30000)
    feat = randn((1, 64, 24, 31), 13).to(dev)
    ht, wt = 67, 90
    coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
    x = (randn((1, 3, 24, 31), 14) * 0.3).to(dev)
    ref = g._predict([feat], coord, cell, 30000, x).cpu()
    with hip_ops.profile():
        got = g._predict([feat], coord, cell, 30000, x, 'bf16').cpu()
    assert 'head_kv_fused_bf16' in hip_ops.profile.results()
    err = (got - ref).abs()
    scale = ref.abs().max().item()
    mse = (err ** 2).mean().item()
    psnr = 10 * math.log10(max(scale, 1e-6) ** 2 / max(mse, 1e-20))
    print(f'bf16 head: max|d| {err.max().item():.3e} (out scale {scale:.3f}), PSNR vs fp32 {psnr:.1f} dB')
    assert err.max().item() < 0.15 * max(scale, 1.0) and psnr > 45.0


def test_f16_head_saturates_instead_of_overflowing(dev):
    """IEEE half tops out at 65504.  The f16 kernels clamp at every fp32 -> half conversion (v_med3_f32 in front of the
    convert), so features 2e4 times the usual scale -- whose layer-0 rows, hidden activations and q*key products all exceed
    the half range -- give a finite (saturated) output, not inf -> NaN; at the usual scale the same call is within half
    rounding of the fp32 path."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(64, (256,) * 4, seeded_head(64, 5, head_gain=1.0), dev, eval_bsize=30000)
    feat = randn((1, 64, 24, 31), 13).to(dev)
    coord, cell = make_coord((67, 90)).unsqueeze(0).to(dev), make_cell((67, 90)).unsqueeze(0).to(dev)
    with hip_ops.profile():
        big = g._predict([feat * 2.0e4], coord, cell, 30000, None, 'f16')
    assert 'head_kv_fused_f16' in hip_ops.profile.results()
    assert torch.isfinite(big).all()
    ref = g._predict([feat], coord, cell, 30000, None)
    got = g._predict([feat], coord, cell, 30000, None, 'f16')
    assert (got - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------------------------------------
# edge cases and size-independent properties
# ------------------------------------------------------------------------------------------------
def test_arbitrary_query_set_and_cells_vs_oracle(dev):
    """The head takes ANY coordinate list (not only a regular grid): random coordinates incl. the exact
    borders +-1, per-query random cells, Q = 1 and a Q that is not a multiple of any tile size."""
    from oracle import ciaosr_oracle as orc
    P = seeded_head(64, 17, head_gain=1.5)
    g = _my_generator(64, (256,) * 4, P, dev, eval_bsize=None)
    feat = randn((1, 64, 11, 14), 51)
    gen = torch.Generator().manual_seed(3)
    for Q in (1, 333):
        coord = torch.rand(1, Q, 2, generator=gen) * 2 - 1
        coord[0, 0] = torch.tensor([-1.0, 1.0])
        cell = (torch.rand(1, Q, 2, generator=gen) * 0.2 + 0.01)
        want = orc.query_rgb(feat, coord, cell, P)
        got = g.query_rgb([feat.to(dev)], coord.to(dev), cell.to(dev)).cpu()
        assert (got - want).abs().max() < 2e-4, (Q, (got - want).abs().max().item())


def test_batch_of_two_images(dev):
    """B = 2: batch items are independent (arch_csnln.py:491, ciaosr_net.py batching)."""
    from ciaosr_amd.coords import make_coord, make_cell
    P = seeded_head(64, 18, head_gain=1.5)
    g = _my_generator(64, (256,) * 4, P, dev, eval_bsize=30000)
    feat = randn((2, 64, 9, 12), 52).to(dev)
    coord = make_coord((20, 27)).unsqueeze(0).expand(2, -1, 2).contiguous().to(dev)
    cell = make_cell((20, 27)).unsqueeze(0).expand(2, -1, 2).contiguous().to(dev)
    both = g.batched_predict([feat], coord, cell)
    one = g.batched_predict([feat[1:2]], coord[1:2], cell[1:2])
    assert torch.equal(both[1], one[0])


def test_rerun_is_bitwise_deterministic(dev):
    """No atomics anywhere: the same call twice gives identical bits (needed for the N-GPU == 1-GPU guarantee)."""
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    model = _restorer('rdn', 4, dev, dict(scale=4, tile=32, tile_overlap=8))
    seeded_init_(model, 31, gain=1.4, head_gain=2.0)
    model = model.to(dev)
    lq, _ = synthetic_pair(40, 44, 4)
    a = model.restore(lq.to(dev))
    b = model.restore(lq.to(dev))
    assert torch.equal(a, b)


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'f16'])
def test_head_rerun_is_bitwise_deterministic_at_scale(dev, precision):
    """65 536 queries on a 64x64 map, six runs, sqrt(6)-gain weights (every rounding difference flips an attention weight
    somewhere): all outputs bitwise equal.  Round 2 found the bf16 decode kernel's last-Linear loop NON-deterministic when
    hipcc's SLP vectoriser packed it into v_pk_* instructions (channel 0 of ~100 of 65 536 queries changed from run to run;
    the library is built with -fno-slp-vectorize since)."""
    from ciaosr_amd.coords import make_coord, make_cell
    g = _my_generator(64, (256,) * 4, seeded_head(64, 0, head_gain=SQRT6), dev, eval_bsize=30000)
    feat = (randn((1, 64, 64, 64), 7) * 0.3).to(dev)
    coord, cell = make_coord((256, 256)).unsqueeze(0).to(dev), make_cell((256, 256)).unsqueeze(0).to(dev)
    x = (randn((1, 3, 64, 64), 14) * 0.3).to(dev)
    outs = [g._predict([feat], coord, cell, 30000, x, precision).clone() for _ in range(6)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o), int((outs[0] != o).any(-1).sum())


def test_linearity_of_the_decode_residual_at_full_tile_size(dev):
    """Size-independent property at BASELINE's full tile size (LR 192x192 -> 768x768, Q = 589 824): the output is
    head(feature) + bilinear(x); changing only x by dx changes the output by exactly bilinear(dx) (fp32 rounding),
    and all 589 824 queries are finite.  Exercises the 192 tile (workspace sizes, 32-bit offsets, chunking)."""
    from ciaosr_amd import hip_ops
    P = seeded_head(64, 19, head_gain=1.0)
    g = _my_generator(64, (256,) * 4, P, dev, eval_bsize=30000)
    feat = (randn((1, 64, 192, 192), 53) * 0.5).to(dev)
    coord, cell = hip_ops.make_coord_cell(768, 768, dev)
    coord, cell = coord.unsqueeze(0), cell.unsqueeze(0)
    x0 = torch.zeros(1, 3, 192, 192, device=dev)
    x1 = (randn((1, 3, 192, 192), 54) * 0.3).to(dev)
    y0 = g._predict([feat], coord, cell, 30000, x0)
    y1 = g._predict([feat], coord, cell, 30000, x1)
    assert torch.isfinite(y0).all() and y0.shape == (1, 589824, 3)
    bil = torch.nn.functional.grid_sample(x1, coord.flip(-1).unsqueeze(1), mode='bilinear', padding_mode='border',
                                          align_corners=False)[:, :, 0, :].permute(0, 2, 1)
    assert ((y1 - y0) - bil).abs().max() < 5e-5          # two fp32 roundings on O(1) values
    # and the staged path agrees with the fused path on a slice of this tile
    sl = slice(300000, 300000 + 4096)
    ys = g._predict([feat], coord[:, sl].contiguous(), cell[:, sl].contiguous(), 0, x0, hip_ops.Options(head_route=1))
    yf = g._predict([feat], coord[:, sl].contiguous(), cell[:, sl].contiguous(), 0, x0)
    assert (ys - yf).abs().max() < 5e-5 * max(1.0, yf.abs().max().item())


def test_whole_image_path_with_operands_past_4gib(dev):
    """Untiled RDN x4 on a 300 x 300 LR image (the configs' scale > 4 rule disables tiling, configs/001_...rdn...py:47-50): the cs_attn
    logit matrix is 8.1 GB in fp32 and 4.1 GB of half probabilities -- operands past one 4-GiB buffer descriptor, walked as row blocks.
    No reference vector at this size: the f16 mode against the fp32 path (PSNR > 50 dB, the distance the reference-pinned 192 x 192
    tile shows), every value finite."""
    import math
    from ciaosr_amd.coords import make_coord, make_cell
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    model = _restorer('rdn', 4, dev, dict(scale=4))
    seeded_init_(model, seed=3, gain=1.0, head_gain=SQRT6)
    model = model.to(dev)
    lq = synthetic_pair(300, 300, 4)[0].to(dev)
    coord, cell = make_coord((1200, 1200)).unsqueeze(0).to(dev), make_cell((1200, 1200)).unsqueeze(0).to(dev)
    a = model.restore(lq, coord, cell, options='fp32').cpu()
    b = model.restore(lq, coord, cell, options='f16').cpu()
    assert a.shape == (1, 3, 1200, 1200) and torch.isfinite(a).all() and torch.isfinite(b).all()
    psnr = -10 * math.log10(max((a - b).double().pow(2).mean().item(), 1e-20))
    print(f'300x300 whole image: PSNR(f16, fp32) {psnr:.1f} dB, means {a.mean().item():.6f} / {b.mean().item():.6f}')
    assert psnr > 50.0, psnr


def test_tiny_image_whole_path(dev):
    """Smallest sensible LR image (4x6, odd sizes inside cs_attn after halving) at a big scale (x12)."""
    from ciaosr_amd.coords import make_coord, make_cell
    from ciaosr_amd.init_utils import seeded_init_
    from oracle import ciaosr_oracle as orc
    model = _restorer('edsr', 12, dev, dict(scale=12), mid=64, blocks=2)
    seeded_init_(model, 41, gain=1.25, head_gain=2.0)
    params = {k[len('generator.'):]: v.detach().clone() for k, v in model.state_dict().items()}
    lq = torch.rand(1, 3, 4, 6, generator=torch.Generator().manual_seed(6))
    coord, cell = make_coord((48, 72)).unsqueeze(0), make_cell((48, 72)).unsqueeze(0)
    want = orc.forward_test(lq, coord, cell, params)
    got = model.to(dev).restore(lq.to(dev), coord.to(dev), cell.to(dev)).cpu()
    assert (got - want).abs().max() < 2e-4


def test_any_scale_tiled_restore_vs_oracle_composition(dev):
    """Opt-in tiling at a non-integer scale (tile_plan.py): EDSR x3.3 on a 45x51 LR image with 32-pixel
    tiles against the same plan evaluated tile by tile with the torch-CPU oracle and blended on the CPU; and a one-tile
    plan against the plain whole-image path (bitwise)."""
    from ciaosr_amd import tile_plan
    from ciaosr_amd.coords import make_coord, make_cell
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from oracle import ciaosr_oracle as orc
    h, w, ht, wt = 45, 51, 149, 168
    model = _restorer('edsr', 3.3, dev, dict(tile=32, tile_overlap=8, tile_any_scale=True), blocks=4)
    seeded_init_(model, seed=5, gain=1.4)
    params = {k[len('generator.'):]: v.detach().clone().cpu() for k, v in model.state_dict().items()}
    model = model.to(dev)
    lq, _ = synthetic_pair(h, w, 3)
    coord, cell = make_coord((ht, wt)).unsqueeze(0), make_cell((ht, wt)).unsqueeze(0)
    got = model.restore(lq.to(dev), coord.to(dev), cell.to(dev)).cpu()
    assert got.shape == (1, 3, ht, wt)
    mean = torch.tensor(model.rgb_mean).view(1, 3, 1, 1)
    x = lq - mean
    E, Wt = torch.zeros(1, 3, ht, wt), torch.zeros(1, 3, ht, wt)
    for t in tile_plan.plan(h, w, ht, wt, 32, 8):
        patch = x[..., t['y0']:t['y0'] + 32, t['x0']:t['x0'] + 32]
        out = orc.generator_forward(patch, t['coord'].unsqueeze(0), t['cell'].unsqueeze(0), params)
        nh, nw = t['i1'] - t['i0'], t['j1'] - t['j0']
        E[..., t['i0']:t['i1'], t['j0']:t['j1']] += out.view(1, nh, nw, 3).permute(0, 3, 1, 2)
        Wt[..., t['i0']:t['i1'], t['j0']:t['j1']] += 1
    want = (E / Wt + mean).clamp(0, 1)
    assert (got - want).abs().max().item() < 1e-4
    # one tile (square image, tile >= image) == whole-image path, bitwise
    lq2, _ = synthetic_pair(40, 40, 3)
    coord2, cell2 = make_coord((131, 131)).unsqueeze(0).to(dev), make_cell((131, 131)).unsqueeze(0).to(dev)
    model.test_cfg['tile'] = 192
    one = model.restore(lq2.to(dev), coord2, cell2).cpu()
    model.test_cfg.pop('tile'); model.test_cfg.pop('tile_any_scale')
    whole = model.restore(lq2.to(dev), coord2, cell2).cpu()
    assert torch.equal(one, whole)


@pytest.mark.parametrize('hw', [(48, 48), (45, 51), (8, 20)])
def test_swinir_trunk_hip_vs_torch(dev, hw):
    """ciaosr_swinir_forward_f32 (LayerNorm / window attention with folded shift, bias and mask / GELU MLP / 3x3 convs on
    padded channel maps) against the PyTorch trunk with the same weights: map sizes that need reflect padding, and a map
    with a single window row (the cyclic shift wraps inside one window there)."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.config import Config
    from ciaosr_amd import build_model
    from ciaosr_amd.init_utils import seeded_init_
    cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', '001_localimplicitsr_swinir_div2k_g1_c64b16_1000k_unfold_lec_mulwkv_res_nonlocal.py'))
    model = build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    seeded_init_(model, seed=9, gain=1.3)
    gen = model.generator.to(dev).eval()
    assert gen._encoder_hip.supported()
    x = (randn((1, 3) + hw, 91) * 0.3).to(dev)
    from tests.torch_trunks import swinir_features
    with torch.no_grad():
        want = swinir_features(gen, x)           # checker; pinned to the reference's features by tests/test_host_logic.py
    with hip_ops.profile():
        got = gen.gen_feature(x)[0]
    assert 'swin_window_attention' in hip_ops.profile.results(), 'HIP SwinIR trunk did not run'
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    print(f'swinir trunk {hw}: max|d| {err:.3e} (scale {scale:.3f})')
    assert err < 2e-4 * max(scale, 1.0), (err, scale)


def test_bench_line_contract(dev):
    """`python bench.py` prints ONE JSON line with the driver's contract fields, the roofline object of the dominant kernel
    and the CPU baseline (bounded sample of one tile, extrapolated).  Run on the 6-tile C3 variant (c3s: same code path
    as the default 117-tile C3, 20x cheaper); the default workload must be C3."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'bench.py')).read()
    assert "ap.add_argument('--workload', default='c3'" in src
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'c3s', '--steps', '2', '--warmup', '1'],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'rccl_ranks', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['unit'] == 'Mpix/s' and d['higher_is_better'] is True
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['config']['workload'].startswith('C3')
    assert abs(d['value'] - 1356 * 2040 / 1e6 / (d['ms_per_step'] * 1e-3)) < 1e-2 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel'):
        assert k in r, k
    assert r['bound'] in ('mfma', 'hbm') and 0 < r['frac'] <= 1.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['staged_path_hbm_kernels']['local_attention']['frac'] >= 0.40          # north star: >= 40 % of the HBM roofline on K4
    # ... as flat scalars of `roofline` (what the driver's record keeps), live HIP-event figures next to the committed rocprof / counter evidence
    assert r['k4_hbm_frac'] >= 0.40 and r['k4_bytes_per_query'] == 22064 and r['k1_hbm_frac'] >= 0.40 and r['k1_bytes_per_query'] == 21936, r
    assert r['k4_rocprof_hbm_frac'] >= 0.40 and r['k4_counter_over_algorithmic'] <= 1.3, r      # profiles/r6_c3tile_staged_*
    keys = list(d)
    assert keys.index('cpu_baseline') < keys.index('roofline') < keys.index('kernels_ms_per_step'), keys      # headline first, bulk last
    assert len(d['cpu_baseline']['seconds_per_tile_by_threads']) >= 2
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['value'] > 0 and d['value'] / c['value'] >= 50     # north star: >= 50x the CPU path
    assert 'EXTRAPOLATED' in c['sample'] and c['cs_attn_hoisted']['value'] > c['value']
