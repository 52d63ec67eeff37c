"""CPU-side tests: host logic, ABI surface, fail-loud behaviour (no GPU needed)."""
import ctypes
import json
import os
import re

import pytest
import torch

from tests.helpers import GOLDEN

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_restorer(test_cfg):
    from ciaosr_amd import CiaoSR, LocalImplicitSREDSR
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[32, 32])
    gen = dict(type=LocalImplicitSREDSR,
               encoder=dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=8, num_blocks=1),
               imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000)
    return CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss'), rgb_mean=(0.4488, 0.4371, 0.4040),
                  rgb_std=(1., 1., 1.), test_cfg=test_cfg).eval()


def test_library_loads_and_exports_every_declared_symbol():
    """Every function declared in include/ciaosr_hip.h is exported by the built .so, and the ctypes
    table in ciaosr_amd/_lib.py covers exactly that set (no compute calls here)."""
    from ciaosr_amd import _lib
    header = open(os.path.join(REPO, 'include', 'ciaosr_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(ciaosr_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in the header but not exported'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().ciaosr_version() >= 100
    assert _lib.load().ciaosr_error_string(-4).decode().startswith('workspace')


def _c_class(decl):
    """Coarse class of a C parameter / return declaration: 'ptr', 'int', 'float', 'size_t', 'double*' ..."""
    decl = decl.strip()
    if '*' in decl:
        return 'ptr'
    base = decl.split()[:-1] if len(decl.split()) > 1 else decl.split()
    base = ' '.join(t for t in base if t != 'const')
    return {'int': 'int', 'float': 'float', 'size_t': 'size_t', 'void': 'void'}[base]


def _ctypes_class(t):
    if t is None:
        return 'void'
    if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, 'contents'):
        return 'ptr'
    return {ctypes.c_int: 'int', ctypes.c_float: 'float', ctypes.c_size_t: 'size_t'}[t]


def test_ctypes_signatures_and_struct_layouts_match_the_header():
    """Prototype by prototype: return class, arity and per-argument class (pointer / int / float / size_t) of the
    ctypes table equal the header's; struct by struct: field count, per-field class and array length equal the
    header's typedef, and sizeof equals what the built library reports (ciaosr_sizeof)."""
    from ciaosr_amd import _lib
    header = open(os.path.join(REPO, 'include', 'ciaosr_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    protos = re.findall(r'^\s*((?:const\s+)?[a-z_]+\s*\*?)\s*(ciaosr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', header, flags=re.M)
    assert len(protos) == len(_lib.SIGNATURES), (len(protos), len(_lib.SIGNATURES))
    for ret, name, args in protos:
        res, argtypes = _lib.SIGNATURES[name]
        params = [] if args.strip() in ('', 'void') else [a for a in args.split(',')]
        assert _c_class(ret + ' x') == _ctypes_class(res), name
        assert len(params) == len(argtypes), (name, len(params), len(argtypes))
        for i, (a, t) in enumerate(zip(params, argtypes)):
            assert _c_class(a) == _ctypes_class(t), (name, i, a.strip(), t)
    lib = _lib.load()
    structs = re.findall(r'typedef struct \w+ \{(.*?)\}\s*(ciaosr_\w+_t);', header, flags=re.S)
    assert {n for _, n in structs} == set(_lib.STRUCTS)
    for body, name in structs:
        st = _lib.STRUCTS[name]
        assert lib.ciaosr_sizeof(name.encode()) == ctypes.sizeof(st), name
        fields = []
        for stmt in [x.strip() for x in body.split(';') if x.strip()]:
            m = re.match(r'(.*?)([\w\s,\*\[\]]+)$', stmt, flags=re.S)
            first, *more = stmt.split(',')
            toks = first.rsplit(None, 1)
            base = toks[0]
            for d in [toks[1]] + [x.strip() for x in more]:
                arr = re.search(r'\[(\w+)\]', d)
                n = None if not arr else (8 if arr.group(1) == 'CIAOSR_MAX_LAYERS' else int(arr.group(1)))
                nm = re.sub(r'\[.*', '', d).replace('*', '').strip()
                is_ptr = '*' in base or '*' in d
                kind = 'ptr' if is_ptr else ('struct' if base.split()[-1].startswith('ciaosr_') else
                                             {'int': 'int', 'float': 'float'}[base.replace('const', '').strip()])
                fields.append((nm, kind, n))
        mine = []
        for fname, ftype in st._fields_:
            n = None
            if hasattr(ftype, '_length_'):
                n, ftype = ftype._length_, ftype._type_
            kind = ('struct' if issubclass(ftype, ctypes.Structure) else
                    'ptr' if (ftype in (ctypes.c_void_p,) or hasattr(ftype, 'contents')) else
                    {ctypes.c_int: 'int', ctypes.c_float: 'float'}[ftype])
            mine.append((kind, n))
        assert [(k, n) for _, k, n in fields] == mine, (name, fields, mine)


def test_product_path_rejects_cpu_tensors():
    """No CPU fallback: calling the generator on CPU tensors raises instead of computing."""
    from ciaosr_amd._lib import CiaoSRHipError
    model = _small_restorer(dict(scale=2))
    with pytest.raises(CiaoSRHipError):
        model.restore(torch.rand(1, 3, 8, 8), torch.zeros(1, 4, 2), torch.ones(1, 4, 2))


def test_product_code_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, 'ciaosr_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


@pytest.mark.parametrize('kind', ['rdn', 'edsr'])
def test_state_dict_names_match_reference(kind):
    """Parameter names/shapes/order equal what the reference's classes produce (checkpoint compat)."""
    from ciaosr_amd import CiaoSR, LocalImplicitSREDSR, LocalImplicitSRRDN
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[256] * 4)
    if kind == 'rdn':
        gen = dict(type=LocalImplicitSRRDN,
                   encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
                                upscale_factor=4, num_layers=8, channel_growth=64),
                   imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000)
    else:
        gen = dict(type=LocalImplicitSREDSR,
                   encoder=dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16),
                   imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000)
    m = CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'))
    names = json.load(open(os.path.join(GOLDEN, f'state_dict_names_{kind}.json')))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(mine) == list(names) and mine == names
    assert not hasattr(m.generator, 'encoder')       # re-parented then deleted (ciaosr_net.py:314-319)


def test_feat_unfold_false_wiring():
    """ciaosr_net.py:61-68: without the unfold the implicit functions see C (not 9C) features; the reference's own forward
    only works with non_local_attn=False then (the non-local concat lives on the unfold branch, :129-141)."""
    from ciaosr_amd import LocalImplicitSREDSR
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[32, 32])
    enc = dict(type='EDSR', in_channels=3, out_channels=3, mid_channels=8, num_blocks=1)
    g = LocalImplicitSREDSR(enc, mk(4, 3), mk(64, 64), mk(64, 64), feat_unfold=False, non_local_attn=False)
    assert g.imnet_q.in_dim == 8 and g.imnet_k.in_dim == 12 and g.imnet_k.out_dim == 8 and g.imnet_v.in_dim == 12
    with pytest.raises(ValueError):
        LocalImplicitSREDSR(enc, mk(4, 3), mk(64, 64), mk(64, 64), feat_unfold=False, non_local_attn=True)


def test_dims_wiring():
    """ciaosr_net.py:56-76: the config's in/out dims are overwritten from the encoder width."""
    m = _small_restorer(dict(scale=2)).generator
    assert m.imnet_q.layers[0].in_features == 8 * 9 + 8
    assert m.imnet_k.layers[0].in_features == 8 * 9 + 4 and m.imnet_k.layers[-1].out_features == 72
    assert m.imnet_v.layers[0].in_features == 8 * 9 + 8 + 4 and m.imnet_v.layers[-1].out_features == 80
    with pytest.raises(TypeError):
        m.init_weights(pretrained=3)


def test_reassigned_test_cfg_reaches_the_generator():
    """`restorer.test_cfg = {...}` (reassignment, what mmedit's apis do before a test run) re-binds the dict the generator reads its
    extensions from; in-place edits are seen too."""
    r = _small_restorer(dict(scale=2))
    g = r.generator
    assert g._test_cfg is r.test_cfg and r.test_cfg.scale == 2
    r.test_cfg['allow_f16_substitute'] = True
    assert g._test_cfg.get('allow_f16_substitute') is True
    r.test_cfg = dict(scale=3, tile=64)
    assert g._test_cfg is r.test_cfg and g._test_cfg.get('allow_f16_substitute', False) is False and r.test_cfg.tile == 64
    r.test_cfg = None
    assert g._test_cfg is None and r.test_cfg is None


def test_tile_grid_matches_reference_lists():
    from ciaosr_amd.restorer import tile_grid, tile_starts
    assert tile_starts(1356, 192, 32) == list(range(0, 1356 - 192, 160)) + [1356 - 192]
    tile, origins = tile_grid(1356, 2040, 192, 32)
    assert tile == 192 and len(origins) == 9 * 13 and origins[0] == (0, 0) and origins[-1] == (1164, 1848)
    tile, origins = tile_grid(48, 48, 192, 32)
    assert tile == 48 and origins == [(0, 0)]


def test_unfold_permutation_is_the_documented_one():
    from ciaosr_amd.head_hip import unfold_perm
    C = 5
    perm = unfold_perm(C, 'cpu')
    x = torch.arange(C * 9).view(C, 3, 3)          # reference index c*9 + ki*3 + kj
    dev_order = x.permute(1, 2, 0).reshape(-1)     # (ki,kj,c)
    assert torch.equal(perm, dev_order)


def _swinir_ciaosr(test_cfg):
    from ciaosr_amd import CiaoSR, LocalImplicitSRSWINIR
    from ciaosr_amd.encoders import SwinIR
    mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[256] * 4)
    gen = dict(type=LocalImplicitSRSWINIR, window_size=8,
               encoder=dict(type=SwinIR, upscale=4, in_chans=3, img_size=48, window_size=8, img_range=1.,
                            depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, mlp_ratio=2, upsampler='pixelshuffle',
                            resi_connection='1conv', compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
                            overlap_ratio=0.5),
               imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64), feat_unfold=True, eval_bsize=30000)
    return CiaoSR(generator=gen, pixel_loss=dict(type='L1Loss'), rgb_mean=(0.4488, 0.4371, 0.4040),
                  rgb_std=(1., 1., 1.), test_cfg=test_cfg).eval()


def test_swinir_names_and_trunk_match_reference():
    """SwinIR-CiaoSR: state_dict names/shapes/order equal the reference's; the PyTorch CHECKER of the trunk
    (tests/torch_trunks.py: reflect padding to the window multiple, 36 Swin blocks, crop) reproduces the reference's features
    on CPU -- it is what the HIP trunk is compared with on the GPU box.  The product itself has no CPU / PyTorch trunk."""
    import numpy as np
    from ciaosr_amd.init_utils import seeded_init_
    from tests.helpers import load_golden, randn
    fx = load_golden('swinir_c5')
    m = _swinir_ciaosr(dict(scale=3.3))
    names = json.load(open(os.path.join(GOLDEN, 'state_dict_names_swinir.json')))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(mine) == list(names) and mine == names
    assert seeded_init_(m, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=6 ** 0.5) == str(fx['sha'])
    x = randn((1, 3, 20, 27), fx['x_seed']) * 0.3
    from tests.torch_trunks import swinir_features
    from ciaosr_amd._lib import CiaoSRHipError
    with torch.no_grad():
        feat = swinir_features(m.generator, x)
    assert feat.shape == (1, 180, 20, 27)
    assert (feat[0] - torch.from_numpy(fx['feat'])).abs().max() < 2e-5
    with pytest.raises(CiaoSRHipError):
        m.generator.gen_feature(x)


@pytest.mark.parametrize('kind', ['rdn', 'edsr', 'swinir'])
def test_configs_build_through_the_shim_import_paths(kind):
    """configs/001_* import `mmedited.models...` exactly like the reference's configs and build via build_model."""
    import glob
    import ciaosr_amd
    from ciaosr_amd.config import Config
    path = glob.glob(os.path.join(REPO, 'configs', f'001_localimplicitsr_{kind}_*.py'))[0]
    cfg = Config.fromfile(path)
    model = ciaosr_amd.build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    n = sum(p.numel() for p in model.parameters())
    assert n == {'rdn': 23402566, 'edsr': 2649030, 'swinir': 14667218}[kind]
    assert model.test_cfg.tile == 192 and model.test_cfg.tile_overlap == 32 and model.generator.eval_bsize == 30000


def test_reference_rdn_config_loads_unmodified():
    """The reference's own RDN config file executes unmodified against the shim packages (its EDSR and SwinIR
    config files have an unbalanced parenthesis upstream and cannot be parsed by any loader)."""
    ref = '/root/reference/configs/001_localimplicitsr_rdn_div2k_g1_c64b16_1000k_unfold_lec_mulwkv_res_nonlocal.py'
    if not os.path.exists(ref):
        pytest.skip('reference tree not present (GPU box)')
    import ciaosr_amd
    from ciaosr_amd.config import Config
    cfg = Config.fromfile(ref)
    model = ciaosr_amd.build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    assert type(model).__name__ == 'CiaoSR' and type(model.generator).__name__ == 'LocalImplicitSRRDN'
    assert cfg.data.test.filename_tmpl == '{}' and cfg.dist_params.backend == 'nccl'


def test_dataset_pipeline_and_checkpoint_roundtrip(tmp_path):
    """SRFolderDataset (paired folders -> lq/gt/coord/cell) and mmcv-style checkpoint loading."""
    from ciaosr_amd.checkpoint import load_checkpoint
    from ciaosr_amd.dataset import SRFolderDataset
    from ciaosr_amd.imageio import imwrite
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair, state_dict_sha256
    from ciaosr_amd.metrics import tensor2img
    (tmp_path / 'lq').mkdir(); (tmp_path / 'gt').mkdir()
    lq, gt = synthetic_pair(12, 16, 2)
    imwrite(tensor2img(lq), str(tmp_path / 'lq' / 'a.png'))
    imwrite(tensor2img(gt), str(tmp_path / 'gt' / 'a.png'))
    ds = SRFolderDataset(tmp_path / 'lq', tmp_path / 'gt', scale=2)
    d = ds[0]
    assert d['lq'].shape == (3, 12, 16) and d['gt'].shape == (24 * 32, 3) and d['coord'].shape == (768, 2)
    assert (d['lq'] - (lq[0] * 255).round() / 255).abs().max() < 1e-6          # RGB order preserved through PNG
    assert torch.allclose(d['cell'][0], torch.tensor([2 / 24, 2 / 32]))
    m = _small_restorer(dict(scale=2))
    sha = seeded_init_(m, 5)
    torch.save({'state_dict': m.state_dict(), 'meta': {}}, tmp_path / 'ck.pth')
    m2 = _small_restorer(dict(scale=2))
    load_checkpoint(m2, str(tmp_path / 'ck.pth'), strict=True)
    assert state_dict_sha256(m2) == sha
    # a generator-only file (keys without the 'generator.' prefix) matches nothing: refuse instead of evaluating random init
    torch.save({'state_dict': m.generator.state_dict()}, tmp_path / 'gen_only.pth')
    with pytest.raises(RuntimeError, match='none of its'):
        load_checkpoint(_small_restorer(dict(scale=2)), str(tmp_path / 'gen_only.pth'))
    # a partially matching file loads non-strictly but says what is missing
    part = {k: v for k, v in m.state_dict().items() if 'imnet_q' not in k}
    part['generator.extra.weight'] = torch.zeros(1)
    torch.save(part, tmp_path / 'part.pth')
    with pytest.warns(UserWarning, match='missing key'):
        load_checkpoint(_small_restorer(dict(scale=2)), str(tmp_path / 'part.pth'))
    with pytest.raises(RuntimeError, match='checkpoint mismatch'):
        load_checkpoint(_small_restorer(dict(scale=2)), str(tmp_path / 'part.pth'), strict=True)


def test_any_scale_tile_plan_properties():
    """tile_plan.py (opt-in): integer scales reproduce clip_test's rectangles and tile-local grids, a
    one-tile plan IS the whole-image grid (bitwise), every HR pixel is covered, coordinates stay inside the tile."""
    import numpy as np
    from ciaosr_amd import tile_plan
    from ciaosr_amd.coords import make_coord, make_cell
    # integer scale: same rectangles as ciaosr.py:233-254 and the same local grids up to fp32 rounding
    h, w, sf, tile, ov = 50, 70, 4, 32, 8
    for t in tile_plan.plan(h, w, h * sf, w * sf, tile, ov):
        assert (t['i0'], t['i1'], t['j0'], t['j1']) == (t['y0'] * sf, (t['y0'] + tile) * sf, t['x0'] * sf, (t['x0'] + tile) * sf)
        ref = make_coord((tile * sf, tile * sf))
        assert (t['coord'] - ref).abs().max().item() < 3e-7
        assert (t['cell'] - make_cell((tile * sf, tile * sf))).abs().max().item() < 1e-9
    # one tile covering the image: the global grid and cell, bitwise
    (t,) = tile_plan.plan(24, 24, 79, 79, 192, 32)
    assert torch.equal(t['coord'], make_coord((79, 79))) and torch.equal(t['cell'], make_cell((79, 79)))
    # non-integer scale: full coverage, blend weights >= 1 everywhere, local coordinates inside (-1, 1)
    h, w, ht, wt = 45, 51, 149, 168                     # x3.3
    cover = np.zeros((ht, wt), np.int32)
    for t in tile_plan.plan(h, w, ht, wt, 32, 8):
        cover[t['i0']:t['i1'], t['j0']:t['j1']] += 1
        assert t['coord'].shape[0] == (t['i1'] - t['i0']) * (t['j1'] - t['j0'])
        assert t['coord'].abs().max().item() < 1.0
        # centre rule: the first/last HR row of the tile maps into the tile, its outer neighbours do not
        pos = lambda i, n_lr, n_hr: (i + 0.5) * n_lr / n_hr
        assert t['y0'] <= pos(t['i0'], h, ht) < t['y0'] + t['th'] and t['y0'] <= pos(t['i1'] - 1, h, ht) < t['y0'] + t['th']
        assert t['i0'] == 0 or pos(t['i0'] - 1, h, ht) < t['y0']
        assert t['i1'] == ht or pos(t['i1'], h, ht) >= t['y0'] + t['th']
    assert cover.min() >= 1 and cover.max() <= 4


def test_built_library_has_no_crossed_packed_fp32_instruction():
    """gfx950: a packed fp32 VALU instruction with ONE crossed source selection (what hipcc's SLP vectoriser emits) returns wrong values
    in lanes 48-63 while another wave of its SIMD issues 16-bit MFMAs (tools/ubench/pk_mfma_corun.hip) -- the root cause of the
    run-to-run differences the bitwise re-run test once caught.  The library is built with -fno-slp-vectorize; this disassembles what
    was actually built and holds it to zero such instructions."""
    import shutil
    from ciaosr_amd import _lib
    from tools import isa_scan
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built')
    if shutil.which(os.path.join(isa_scan.LLVM, 'llvm-objdump')) is None:
        pytest.skip('no llvm-objdump')
    hits, n_pk, n_obj = isa_scan.scan(_lib.LIB_PATH)
    assert n_obj >= 20, n_obj                      # every translation unit was found in the fat binary
    assert not hits, hits[:5]


def test_chained_head_blob_sizes_follow_the_documented_stream_layout():
    """`ciaosr_head_chain_bytes` is host arithmetic (no GPU): the blob of the weights-stationary 16-bit head = two 8-KB tail blocks + the
    phi_k / phi_v stream (24 + 24 hidden tiles + the output layer's 32-column units padded to an even count; 16 KB a tile, twice that as
    hi + lo pairs), followed by imnet_q's stream (input layer k-step major, 24 hidden tiles, the 3-row output layer as a hi + lo tile)
    where Dv is a multiple of 128.  C = 180 (Dv = 1800): 57 units -> 58 tiles and no imnet_q stream.  Shapes the kernels do not
    cover give 0 (the caller then leaves `chain16` NULL and the 128-row kernels run)."""
    from ciaosr_amd import _lib
    lib = _lib.load()
    dummy = ctypes.c_void_p(64)                                   # never dereferenced by the size function

    def head(C, hidden=(256, 256, 256, 256), local_size=2):
        w = _lib.HeadWeightsT()
        w.channels, w.nonlocal_channels, w.nonlocal_max_scale, w.local_size, w.no_unfold = C, C, 4, local_size, 0
        D, Dv = 9 * C, 10 * C
        for m, fan, out in ((w.q, Dv, 3), (w.k, D + 4, D), (w.v, Dv + 4, Dv)):
            m.n_layers, m.in_dim = len(hidden) + 1, fan
            for i, wd in enumerate(tuple(hidden) + (out,)):
                m.width[i], m.ld[i] = wd, 4 * ((([fan] + list(hidden))[i] + 3) // 4)
                m.weight[i], m.bias[i] = dummy, dummy
        return w

    KB = 1024
    w = head(64)
    kv, q = 16 * KB + (48 + 20) * 16 * KB, (640 // 16 // 2 + 24 + 2) * 16 * KB
    assert lib.ciaosr_head_chain_bytes(ctypes.byref(w), 0) == kv + q
    kv2, q2 = 16 * KB + (48 + 20) * 32 * KB, (640 // 16 + 48 + 2) * 16 * KB
    assert lib.ciaosr_head_chain_bytes(ctypes.byref(w), 1) == kv2 + q2
    w = head(180)
    assert lib.ciaosr_head_chain_bytes(ctypes.byref(w), 0) == 16 * KB + (48 + 58) * 16 * KB           # 57 units padded, no imnet_q stream
    assert lib.ciaosr_head_chain_bytes(ctypes.byref(head(64, hidden=(256, 256, 128, 256))), 0) == 0
    assert lib.ciaosr_head_chain_bytes(ctypes.byref(head(64, local_size=3)), 0) == 0


def r_swin_bf16_is_x3():
    """`precision='bf16'` on the SwinIR-CiaoSR generator resolves to 'bf16x3' (f16_pairs = 2), to 'f16' with the rounds-4/5 opt-in."""
    import warnings
    from ciaosr_amd import hip_ops
    m = _swinir_ciaosr(dict(scale=3.3))
    eff = m.generator.effective_options('bf16')
    ok = eff.precision == 'bf16' and eff.f16_pairs == 2 and eff.bf16_single == 0
    ok = ok and m.generator.effective_options(hip_ops.Options('bf16-single')).f16_pairs == 2
    ok = ok and m.generator.effective_options('f16').precision == 'f16' and m.generator.effective_options('bf16x3').f16_pairs == 2
    m.test_cfg['allow_f16_substitute'] = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ok = ok and m.generator.effective_options('bf16').precision == 'f16'
    return ok


def test_round6_host_logic_options_tile_batch_and_error_feedback_rounding():
    """Host-side pieces of round 6 that need no GPU: `Options('bf16-single')` is the bf16 entry with `bf16_single = 1`; the default
    tile batch is 7 wherever the dense layers run a 16x32-pixel-tile kernel (fp32 Winograd F(4x4) and, now, the 16-bit modes) and 8 with
    `dense_direct = 1`; error-feedback rounding to bf16 (`PackedHead._ef_round_bf16`) returns bf16 numbers whose rounding error has every
    prefix sum -- the row sum in particular -- within half an ulp of the largest element, where round-to-nearest accumulates a random walk."""
    from ciaosr_amd import hip_ops
    from ciaosr_amd.head_hip import PackedHead
    o = hip_ops.Options('bf16-single')
    assert o.precision == 'bf16' and o.bf16_single == 1 and o.suffix == 'bf16' and 'bf16-single' in hip_ops.PRECISIONS
    assert hip_ops.Options('bf16').bf16_single == 0
    x3 = hip_ops.Options('bf16x3')
    assert x3.precision == 'bf16' and x3.f16_pairs == 2 and x3.suffix == 'bf16' and 'bf16x3' in hip_ops.PRECISIONS
    assert r_swin_bf16_is_x3()
    r = _small_restorer(dict(scale=2))
    assert r.tile_batch() == 7 and r.tile_batch(hip_ops.Options('f16')) == 7 and r.tile_batch(hip_ops.Options('bf16-single')) == 7
    assert r.tile_batch(hip_ops.Options('f16', dense_direct=1)) == 8 and r.tile_batch(hip_ops.Options('fp32', dense_direct=1)) == 8
    assert r.tile_batch(hip_ops.Options('f16x3')) == 7                       # fp32 trunk
    r.test_cfg['tile_batch'] = 3
    assert r.tile_batch(hip_ops.Options('f16')) == 3
    g = torch.Generator().manual_seed(7)
    w = torch.randn(64, 576, generator=g) * 0.03
    q = PackedHead._ef_round_bf16(w)
    assert q.dtype == torch.float32 and torch.equal(q.to(torch.bfloat16).float(), q)          # bf16 numbers: the pack kernels' rounding is the identity
    rne = w.to(torch.bfloat16).float()
    import math
    ulp = 2.0 ** (math.floor(math.log2(w.abs().max().item())) - 7)           # spacing of bf16 numbers (8-bit significand) at the largest magnitude
    prefix_ef = (q.double() - w.double()).cumsum(1).abs().max().item()
    prefix_rne = (rne.double() - w.double()).cumsum(1).abs().max().item()
    assert prefix_ef <= 0.5 * ulp + 1e-12, (prefix_ef, ulp)
    assert prefix_rne > 4 * prefix_ef, (prefix_rne, prefix_ef)
    assert (q - w).abs().max().item() <= ulp                                 # an element moves by at most one spacing (rne: half)
