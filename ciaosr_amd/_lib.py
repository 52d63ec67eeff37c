"""ctypes binding of libciaosr_hip.so (the C ABI declared in include/ciaosr_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C ciaosr_amd/csrc`.  There is
no CPU fallback: if the shared object is missing, or an entry point returns an error code, the
caller gets an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CIAOSR_HIP_LIB: developer override (e.g. the cycle-stamp probe build, `make -C ciaosr_amd/csrc probe`)
LIB_PATH = os.environ.get('CIAOSR_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libciaosr_hip.so')
MAX_LAYERS = 8

ACT_NONE, ACT_RELU, ACT_PRELU, ACT_GELU, ACT_SIN, ACT_COS = 0, 1, 2, 3, 4, 5


class CiaoSRHipError(RuntimeError):
    pass


class MlpT(C.Structure):
    _fields_ = [('n_layers', C.c_int), ('act', C.c_int), ('in_dim', C.c_int), ('width', C.c_int * MAX_LAYERS),
                ('weight', C.c_void_p * MAX_LAYERS), ('ld', C.c_int * MAX_LAYERS),
                ('bias', C.c_void_p * MAX_LAYERS), ('frag', C.c_void_p * MAX_LAYERS),
                ('frag16', C.c_void_p * MAX_LAYERS), ('frag16_lo', C.c_void_p * MAX_LAYERS)]


class HeadWeightsT(C.Structure):
    _fields_ = [('channels', C.c_int), ('nonlocal_channels', C.c_int), ('nonlocal_max_scale', C.c_int), ('local_size', C.c_int),
                ('no_unfold', C.c_int),
                ('softmax_scale', C.c_float), ('q', MlpT), ('k', MlpT), ('v', MlpT), ('k_out_wino', C.c_void_p), ('k_out_wino4', C.c_void_p),
                ('chain16', C.c_void_p), ('chain16_pairs', C.c_void_p)]


class CsAttnWeightsT(C.Structure):
    _fields_ = [('channels', C.c_int), ('scale', C.c_int),
                ('w_match1', C.c_void_p), ('b_match1', C.c_void_p), ('slope_match1', C.c_float),
                ('w_match2', C.c_void_p), ('b_match2', C.c_void_p), ('slope_match2', C.c_float),
                ('w_assembly', C.c_void_p), ('b_assembly', C.c_void_p), ('slope_assembly', C.c_float),
                ('w_down', C.c_void_p), ('b_down', C.c_void_p), ('w_down_masked', C.c_void_p),
                ('escape_nan', C.c_float), ('softmax_scale', C.c_float)]


class OptionsT(C.Structure):
    """ciaosr_options_t: per-call route options (include/ciaosr_hip.h)."""
    _fields_ = [('head_route', C.c_int), ('csa_composed_min', C.c_int), ('dense_min_tiles', C.c_int),
                ('scatter_small_max', C.c_int), ('kv_rows', C.c_int), ('decode_rows', C.c_int), ('bf16_single', C.c_int),
                ('dense_direct', C.c_int), ('csa_scores_gemm', C.c_int), ('csa_attn_tile128', C.c_int), ('query_grid_w', C.c_int), ('f16_pairs', C.c_int)]


HEAD_STAGED, HEAD_NO_LOGIT_TABLE, HEAD_TABLE_GEMM, HEAD_WIDE_WG, HEAD_TABLE_WINO2, HEAD_NO_CHAIN, HEAD_NO_DECODE_CHAIN = 1, 2, 4, 8, 16, 32, 64


class ConvT(C.Structure):
    _fields_ = [('weight', C.c_void_p), ('bias', C.c_void_p), ('cin', C.c_int), ('cout', C.c_int), ('ksize', C.c_int),
                ('frag16', C.c_void_p), ('frag16_lo', C.c_void_p), ('frag', C.c_void_p), ('frag_wino', C.c_void_p),
                ('frag_wino4', C.c_void_p)]


class RdnWeightsT(C.Structure):
    _fields_ = [('mid_channels', C.c_int), ('growth', C.c_int), ('num_blocks', C.c_int), ('num_layers', C.c_int),
                ('sfe1', ConvT), ('sfe2', ConvT), ('gff0', ConvT), ('gff1', ConvT),
                ('dense', C.POINTER(ConvT)), ('lff', C.POINTER(ConvT)),
                ('scatter_weight', C.POINTER(C.c_void_p)), ('scatter_bias', C.c_void_p),
                ('scatter_frag', C.POINTER(C.c_void_p))]


class SwinBlockT(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('ln1_w', 'ln1_b', 'qkv_w', 'qkv_b', 'bias', 'proj_w', 'proj_b', 'ln2_w', 'ln2_b',
                                           'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b')] + [('shift', C.c_int), ('mask', C.c_void_p)]


class SwinirWeightsT(C.Structure):
    _fields_ = [('embed_dim', C.c_int), ('num_heads', C.c_int), ('window_size', C.c_int), ('hidden', C.c_int),
                ('num_groups', C.c_int), ('depth', C.c_int), ('conv_first', ConvT), ('conv_after_body', ConvT),
                ('pe_norm_w', C.c_void_p), ('pe_norm_b', C.c_void_p), ('norm_w', C.c_void_p), ('norm_b', C.c_void_p),
                ('blocks', C.POINTER(SwinBlockT)), ('group_conv', C.POINTER(ConvT))]


class EdsrWeightsT(C.Structure):
    _fields_ = [('mid_channels', C.c_int), ('num_blocks', C.c_int), ('res_scale', C.c_float),
                ('conv_first', ConvT), ('conv_after_body', ConvT),
                ('conv1', C.POINTER(ConvT)), ('conv2', C.POINTER(ConvT))]


_P, _I, _F, _S = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_O = C.POINTER(OptionsT)

# name -> (restype, argtypes); kept in sync with include/ciaosr_hip.h (tests/test_host_logic.py parses the header's
# prototypes and compares names, arity and argument classes; struct sizes are compared with ciaosr_sizeof())
SIGNATURES = {
    'ciaosr_version': (_I, []),
    'ciaosr_error_string': (C.c_char_p, [_I]),
    'ciaosr_sizeof': (_S, [C.c_char_p]),
    'ciaosr_prof_enable': (_I, [_I]),
    'ciaosr_prof_filter': (_I, [C.c_char_p]),
    'ciaosr_prof_reset': (_I, []),
    'ciaosr_prof_collect': (_I, []),
    'ciaosr_prof_get': (_I, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    'ciaosr_prof_names': (_I, [C.c_char_p, _I]),
    'ciaosr_nchw_to_hwc_f32': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'ciaosr_hwc_to_nchw_f32': (_I, [_P, _I, _P, _I, _I, _I, _P]),
    'ciaosr_gemm_f32': (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _F, _I, _F, _P]),
    'ciaosr_patch_rows_f32': (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _F, _P]),
    'ciaosr_cs_attn_workspace_bytes': (_S, [_I, _I, _I]),
    'ciaosr_cs_attn_workspace_bytes_scale': (_S, [_I, _I, _I, _I]),
    'ciaosr_cs_attn_f32': (_I, [_P, _I, _I, _I, C.POINTER(CsAttnWeightsT), _P, _I, _O, _P, _S, _P]),
    'ciaosr_cs_attn_bf16': (_I, [_P, _I, _I, _I, C.POINTER(CsAttnWeightsT), _P, _I, _O, _P, _S, _P]),
    'ciaosr_make_coord_cell_f32': (_I, [_P, _P, _I, _I, _P]),
    'ciaosr_fragment_floats': (_S, [_I, _I]),
    'ciaosr_pack_fragments_f32': (_I, [_P, _I, _I, _I, _P, _P]),
    'ciaosr_cs_attn_f16': (_I, [_P, _I, _I, _I, C.POINTER(CsAttnWeightsT), _P, _I, _O, _P, _S, _P]),
    'ciaosr_fragment_bf16_bytes': (_S, [_I, _I]),
    'ciaosr_fragment_f16_bytes': (_S, [_I, _I]),
    'ciaosr_pack_fragments_f16': (_I, [_P, _I, _I, _I, _P, _P]),
    'ciaosr_pack_fragments_bf16': (_I, [_P, _I, _I, _I, _P, _P]),
    'ciaosr_pack_fragments_bf16_lo': (_I, [_P, _I, _I, _I, _P, _P]),
    'ciaosr_pack_fragments_f16_lo': (_I, [_P, _I, _I, _I, _P, _P]),
    'ciaosr_pack_fragments_bf16_pair': (_I, [_P, _I, _I, _I, _P, _P, _P]),
    'ciaosr_pack_fragments_f16_pair': (_I, [_P, _I, _I, _I, _P, _P, _P]),
    'ciaosr_pack_conv3x3_f32': (_I, [_P, _S, _S, _S, _S, _I, _I, _P, _P, _P, _P]),
    'ciaosr_head_chain_bytes': (_S, [C.POINTER(HeadWeightsT), _I]),
    'ciaosr_pack_head_chain_bf16': (_I, [C.POINTER(HeadWeightsT), _I, _P, _P]),
    'ciaosr_pack_head_chain_f16': (_I, [C.POINTER(HeadWeightsT), _I, _P, _P]),
    'ciaosr_head_indices_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'ciaosr_local_attention_f32': (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _F, _P]),
    'ciaosr_local_attention_bf16': (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _F, _P]),
    'ciaosr_local_attention_f16': (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _F, _P]),
    'ciaosr_gather_rows_f32': (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P]),
    'ciaosr_mlp_workspace_bytes': (_S, [C.POINTER(MlpT), _I]),
    'ciaosr_mlp_forward_f32': (_I, [_P, _I, C.POINTER(MlpT), _I, _I, _P, _I, _P, _S, _P]),
    'ciaosr_mlp_workspace_bytes_16': (_S, [C.POINTER(MlpT), _I]),
    'ciaosr_mlp_forward_bf16': (_I, [_P, _I, C.POINTER(MlpT), _I, _P, _I, _P, _S, _P]),
    'ciaosr_mlp_forward_f16': (_I, [_P, _I, C.POINTER(MlpT), _I, _P, _I, _P, _S, _P]),
    'ciaosr_decode_residual_f32': (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P]),
    'ciaosr_head_workspace_bytes': (_S, [_I, _I, C.POINTER(HeadWeightsT), _I]),
    'ciaosr_head_forward_f32': (_I, [_P, _I, _I, C.POINTER(HeadWeightsT), C.POINTER(CsAttnWeightsT), _P, _P, _P,
                                     _I, _I, _P, _O, _P, _S, _P]),
    'ciaosr_head_forward_bf16': (_I, [_P, _I, _I, C.POINTER(HeadWeightsT), C.POINTER(CsAttnWeightsT), _P, _P, _P,
                                      _I, _I, _P, _O, _P, _S, _P]),
    'ciaosr_head_forward_f16': (_I, [_P, _I, _I, C.POINTER(HeadWeightsT), C.POINTER(CsAttnWeightsT), _P, _P, _P,
                                      _I, _I, _P, _O, _P, _S, _P]),
    'ciaosr_rdn_workspace_bytes': (_S, [_I, _I, C.POINTER(RdnWeightsT)]),
    'ciaosr_rdn_forward_f32': (_I, [_P, _I, _I, C.POINTER(RdnWeightsT), _P, _O, _P, _S, _P]),
    'ciaosr_rdn_forward_bf16': (_I, [_P, _I, _I, C.POINTER(RdnWeightsT), _P, _O, _P, _S, _P]),
    'ciaosr_rdn_forward_f16': (_I, [_P, _I, _I, C.POINTER(RdnWeightsT), _P, _O, _P, _S, _P]),
    'ciaosr_rdn_workspace_bytes_batch': (_S, [_I, _I, _I, C.POINTER(RdnWeightsT)]),
    'ciaosr_rdn_forward_batch_f32': (_I, [_P, _I, _I, _I, C.POINTER(RdnWeightsT), _P, _O, _P, _S, _P]),
    'ciaosr_rdn_forward_batch_bf16': (_I, [_P, _I, _I, _I, C.POINTER(RdnWeightsT), _P, _O, _P, _S, _P]),
    'ciaosr_rdn_forward_batch_f16': (_I, [_P, _I, _I, _I, C.POINTER(RdnWeightsT), _P, _O, _P, _S, _P]),
    'ciaosr_edsr_workspace_bytes': (_S, [_I, _I, C.POINTER(EdsrWeightsT)]),
    'ciaosr_edsr_forward_f32': (_I, [_P, _I, _I, C.POINTER(EdsrWeightsT), _P, _P, _S, _P]),
    'ciaosr_swinir_workspace_bytes': (_S, [_I, _I, C.POINTER(SwinirWeightsT)]),
    'ciaosr_swinir_forward_f32': (_I, [_P, _I, _I, C.POINTER(SwinirWeightsT), _P, _P, _S, _P]),
    'ciaosr_normalize_f32': (_I, [_P, _P, _I, _I, C.POINTER(_F), C.POINTER(_F), _P]),
    'ciaosr_denorm_clamp_f32': (_I, [_P, _P, _I, _I, C.POINTER(_F), C.POINTER(_F), _P]),
    'ciaosr_tile_blend_f32': (_I, [_P, _P, _I, _I, _P, _I, _I, _I, _I, _P]),
    'ciaosr_tile_finalize_f32': (_I, [_P, _P, _P, _I, _I, _P]),
}

# ctypes mirror of every ABI struct, by the header's typedef name (layout checked against ciaosr_sizeof at load time)
STRUCTS = {'ciaosr_options_t': OptionsT, 'ciaosr_csattn_weights_t': CsAttnWeightsT, 'ciaosr_mlp_t': MlpT,
           'ciaosr_head_weights_t': HeadWeightsT, 'ciaosr_conv_t': ConvT, 'ciaosr_rdn_weights_t': RdnWeightsT,
           'ciaosr_edsr_weights_t': EdsrWeightsT, 'ciaosr_swin_block_t': SwinBlockT,
           'ciaosr_swinir_weights_t': SwinirWeightsT}

_lib = None


def load():
    """Load the shared object (once).  Raises CiaoSRHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CiaoSRHipError(
            f'{LIB_PATH} not found: build the HIP extension first '
            f'(python -c "import __graft_entry__ as g; g.build()" or make -C ciaosr_amd/csrc). '
            f'There is no CPU fallback for the product path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, st in STRUCTS.items():
        want = lib.ciaosr_sizeof(name.encode())
        if want != C.sizeof(st):
            raise CiaoSRHipError(f'ABI drift: sizeof({name}) is {want} in {LIB_PATH} but {C.sizeof(st)} in the ctypes mirror')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ciaosr_error_string(rc).decode()
        raise CiaoSRHipError(f'{what} failed: {msg} (code {rc})')


def call(name, *args):
    check(getattr(load(), name)(*args), name)
