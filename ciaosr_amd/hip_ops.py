"""Tensor-level wrappers over the C ABI: device pointers from `tensor.data_ptr()`, the stream
from `torch.cuda.current_stream()`.  PyTorch is used for device memory and streams only."""
import ctypes as C

import torch

from . import _lib
from ._lib import CiaoSRHipError

_workspaces = {}


def require_gpu(*tensors, allow_row_stride=False):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.is_cuda:
            # every launch goes to torch's current stream of the CURRENT device: a tensor living on another GPU would be
            # dereferenced from the wrong device (run the call under `torch.cuda.device(t.device)`)
            if dev is None:
                dev = torch.cuda.current_device()
            if t.device.index != dev:
                raise CiaoSRHipError(f'tensor on cuda:{t.device.index} but the current device is cuda:{dev}: '
                                     f'run the call under torch.cuda.device({t.device.index})')
        if not t.is_cuda:
            raise CiaoSRHipError('the LocalImplicitSR path runs on the MI355X only: got a CPU tensor '
                                 '(no CPU fallback; move the model and inputs to cuda)')
        if t.dtype not in (torch.float32, torch.int32):
            raise CiaoSRHipError(f'expected float32/int32 tensor, got {t.dtype}')
        if not t.is_contiguous() and not (allow_row_stride and t.dim() == 2 and t.stride(1) == 1):
            raise CiaoSRHipError('expected a contiguous tensor')


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device, looked up per call so that a
    process driving several GPUs launches on the right one; the index is passed explicitly because the implicit
    lookup inside current_stream() costs ~0.2 ms per call on hosts with many cores)."""
    idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
    return C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


_coord_cache = {}


def make_coord_cell(ht, wt, device):
    """Device-side make_coord/make_cell of an ht x wt target grid, cached per shape."""
    key = (ht, wt, device.type, device.index)
    hit = _coord_cache.get(key)
    if hit is None:
        coord = torch.empty(ht * wt, 2, dtype=torch.float32, device=device)
        cell = torch.empty(ht * wt, 2, dtype=torch.float32, device=device)
        _lib.call('ciaosr_make_coord_cell_f32', ptr(coord), ptr(cell), ht, wt, stream_ptr())
        if len(_coord_cache) > 16:
            _coord_cache.clear()
            _grid_width.clear()
        hit = _coord_cache[key] = (coord, cell)
        _grid_width[(coord.data_ptr(), ht * wt)] = wt
    return hit


_grid_width = {}


def grid_width_of(coord):
    """Columns of the row-major target grid when `coord` [Q, 2] is (a view of) a tensor `make_coord_cell` produced, else 0: the
    traversal hint ciaosr_options_t.query_grid_w of the 16-bit fused head (results do not depend on it)."""
    return _grid_width.get((coord.data_ptr(), coord.shape[0]), 0)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def workspace(nbytes, device, slot='head'):
    """Grow-only scratch buffer per (slot, device, stream): two streams or two devices never share scratch."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (slot, device.type, idx, torch.cuda.current_stream(idx).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _workspaces[key] = None
        buf = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def take_workspaces(stream):
    """Remove and return the scratch buffers that were grown on `stream` (a torch.cuda.Stream): the caller becomes their owner
    (graphed_restore hands them to the captured graph's closure), and a later stream that happens to get the same raw handle
    starts with fresh scratch.  `release_workspaces()` = the same for every stream, dropping the buffers."""
    handle = stream.cuda_stream
    keys = [k for k in _workspaces if k[3] == handle and k[2] == stream.device_index]
    return {k: _workspaces.pop(k) for k in keys}


def poison_workspaces():
    """Test hook: fill every cached scratch buffer with 0xFF bytes (fp32 / bf16 / half NaN patterns).  A kernel that reads scratch
    it (or an earlier kernel of the call) did not write -- a ragged tile edge, a row past K -- then shows up as NaN in the result."""
    for buf in _workspaces.values():
        if buf is not None:
            buf.fill_(0xFF)


def release_workspaces():
    """Drop every cached scratch buffer (they are re-grown on demand)."""
    _workspaces.clear()


def gemm(a, b, bias=None, act=_lib.ACT_NONE, slope=0.0, alpha=1.0, b_is_kn=False, out=None):
    """out[M,N] = act((a[M,K] @ (b[N,K]^T | b[K,N]) + bias) * alpha) through ciaosr_gemm_f32."""
    require_gpu(a, b, bias, allow_row_stride=True)
    M, K = a.shape
    N = b.shape[1] if b_is_kn else b.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _lib.call('ciaosr_gemm_f32', ptr(a), a.stride(0), ptr(b), b.stride(0), int(b_is_kn), ptr(out), out.stride(0),
              ptr(bias), M, N, K, float(alpha), int(act), float(slope), stream_ptr())
    return out


def nchw_to_hwc(x):
    require_gpu(x)
    Cc, H, W = x.shape
    out = torch.empty(H, W, Cc, dtype=torch.float32, device=x.device)
    _lib.call('ciaosr_nchw_to_hwc_f32', ptr(x), ptr(out), Cc, H, W, Cc, stream_ptr())
    return out


def hwc_to_nchw(x):
    require_gpu(x)
    H, W, Cc = x.shape
    out = torch.empty(Cc, H, W, dtype=torch.float32, device=x.device)
    _lib.call('ciaosr_hwc_to_nchw_f32', ptr(x), Cc, ptr(out), Cc, H, W, stream_ptr())
    return out


def head_indices(coord, cell, H, W, local_size=2, chunk=0, want_rel=True):
    require_gpu(coord, cell)
    Q = coord.shape[0]
    J = {1: 1, 2: 4, 3: 9}[local_size]
    q_idx = torch.empty(Q, dtype=torch.int32, device=coord.device)
    k_idx = torch.empty(Q, J, dtype=torch.int32, device=coord.device)
    rel = torch.empty(Q, J, 2, dtype=torch.float32, device=coord.device) if want_rel else None
    _lib.call('ciaosr_head_indices_f32', ptr(coord), ptr(cell), Q, chunk, H, W, local_size, ptr(q_idx), ptr(k_idx),
              ptr(rel), stream_ptr())
    return q_idx, k_idx, rel


def patch_rows(src_hwc, ksize, stride, pad, OH, OW, normalize=False, floor=0.0):
    require_gpu(src_hwc)
    Hs, Ws, Cs = src_hwc.shape
    out = torch.empty(OH * OW, ksize * ksize * Cs, dtype=torch.float32, device=src_hwc.device)
    _lib.call('ciaosr_patch_rows_f32', ptr(src_hwc), Cs, Hs, Ws, Cs, ksize, stride, pad, OH, OW, ptr(out),
              out.stride(0), int(normalize), float(floor), stream_ptr())
    return out


def local_attention(unfold, C_, Cn, q_idx, k_idx, wk, wv, softmax_scale=1.0):
    require_gpu(unfold, q_idx, k_idx, wk, wv)
    Q, J = k_idx.shape
    z = torch.empty(Q, 9 * C_ + Cn, dtype=torch.float32, device=unfold.device)
    _lib.call('ciaosr_local_attention_f32', ptr(unfold), unfold.stride(0), C_, Cn, ptr(q_idx), ptr(k_idx), ptr(wk),
              wk.stride(0), ptr(wv), wv.stride(0), ptr(z), z.stride(0), Q, J, float(softmax_scale), stream_ptr())
    return z


def local_attention_16(unfold, C_, Cn, q_idx, k_idx, wk16, wv16, softmax_scale=1.0):
    """K4 with wk / wv / z as torch.bfloat16 or torch.float16 tensors (ciaosr_local_attention_bf16 / _f16): half the HBM bytes per query."""
    require_gpu(unfold, q_idx, k_idx)
    if not (wk16.is_cuda and wv16.is_cuda and wk16.dtype == wv16.dtype and wk16.dtype in (torch.bfloat16, torch.float16)):
        raise CiaoSRHipError(f'local_attention_16: wk / wv must be bfloat16 or float16 tensors on the GPU, got {wk16.dtype} / {wv16.dtype}')
    Q, J = k_idx.shape
    z = torch.empty(Q, 9 * C_ + Cn, dtype=wk16.dtype, device=unfold.device)
    _lib.call('ciaosr_local_attention_' + ('bf16' if wk16.dtype == torch.bfloat16 else 'f16'), ptr(unfold), unfold.stride(0), C_, Cn,
              ptr(q_idx), ptr(k_idx), ptr(wk16), wk16.stride(0), ptr(wv16), wv16.stride(0), ptr(z), z.stride(0), Q, J, float(softmax_scale),
              stream_ptr())
    return z


def _f3(vals):
    return (C.c_float * 3)(*[float(v) for v in vals])


def normalize(lq_chw, mean, std):
    require_gpu(lq_chw)
    out = torch.empty_like(lq_chw)
    _, H, W = lq_chw.shape
    _lib.call('ciaosr_normalize_f32', ptr(lq_chw), ptr(out), H, W, _f3(mean), _f3(std), stream_ptr())
    return out


def denorm_clamp(pred_q3, H, W, mean, std):
    require_gpu(pred_q3)
    out = torch.empty(3, H, W, dtype=torch.float32, device=pred_q3.device)
    _lib.call('ciaosr_denorm_clamp_f32', ptr(pred_q3), ptr(out), H, W, _f3(mean), _f3(std), stream_ptr())
    return out


def tile_blend(E, Wt, tile_q3, y0, x0, th, tw):
    require_gpu(E, Wt, tile_q3)
    _, Himg, Wimg = E.shape
    _lib.call('ciaosr_tile_blend_f32', ptr(E), ptr(Wt), Himg, Wimg, ptr(tile_q3), y0, x0, th, tw, stream_ptr())


def tile_finalize(E, Wt):
    require_gpu(E, Wt)
    _, Himg, Wimg = E.shape
    out = torch.empty(Himg * Wimg, 3, dtype=torch.float32, device=E.device)
    _lib.call('ciaosr_tile_finalize_f32', ptr(E), ptr(Wt), ptr(out), Himg, Wimg, stream_ptr())
    return out


PRECISIONS = ('fp32', 'bf16', 'bf16-single', 'bf16x3', 'f16', 'f16-pairs', 'f16x3', 'f16x3-fast')       # what Options(precision) / test_cfg.precision accept


class Options:
    """Per-call evaluation options, passed explicitly down the call chain (no process-global switches).

    precision  'fp32' (exact-fp32 MFMA, the contract precision), 'bf16' (bf16 MFMA inputs, fp32 accumulation; weights as
               hi + lo pairs unless bf16_single), 'f16' (IEEE half MFMA inputs, one MFMA per product, saturating at 65504)
               'f16-pairs' (= 'f16' with f16_pairs=1: half activations, every weight as a half hi + lo pair) or 'f16x3'
               (= 'f16' with f16_pairs=2, the fp32-tolerance fast mode: the head's weights AND activations as half pairs, three
               MFMAs per product, fp32 trunk and tables, half cs_attn contractions), 'bf16-single' (= 'bf16' with bf16_single=1: one
               bf16 weight per product, packed with error feedback + calibrated biases), 'bf16x3' (= 'bf16' with f16_pairs=2: the bf16
               counterpart of 'f16x3'): selects the _f32 / _bf16 / _f16 entry point.
    the rest   fields of ciaosr_options_t (include/ciaosr_hip.h): result-equivalent route choices; 0 = default.
    Immutable; `replace()` returns a modified copy."""
    _C_FIELDS = ('head_route', 'csa_composed_min', 'dense_min_tiles', 'scatter_small_max', 'kv_rows', 'decode_rows', 'bf16_single', 'dense_direct', 'csa_scores_gemm', 'csa_attn_tile128', 'query_grid_w', 'f16_pairs')
    __slots__ = ('precision',) + _C_FIELDS + ('_c',)

    def __init__(self, precision='fp32', **kw):
        if precision in ('bf16-single', 'bf16_single'):          # ONE bf16 weight per product, error-feedback rounding + calibrated biases (head_hip.py)
            precision = 'bf16'
            kw.setdefault('bf16_single', 1)
        if precision in ('bf16x3', 'bf16-x3'):                   # bf16 hi + lo weights AND activations in the head (three MFMAs per product, fp32 Z), fp32 trunk
            precision = 'bf16'                                   # and tables, bf16 cs_attn contractions: the bf16 counterpart of 'f16x3' (round 6)
            kw.setdefault('f16_pairs', 2)
        if precision in ('f16-pairs', 'f16_pairs', 'f16p'):      # the fp32-tolerance fast mode: half activations, half weight PAIRS
            precision = 'f16'
            kw.setdefault('f16_pairs', 1)
        if precision in ('f16x3', 'f16-x3'):                     # ... with the activations of the MLP chains as pairs too
            precision = 'f16'
            kw.setdefault('f16_pairs', 2)
        if precision == 'f16x3-fast':                            # ... and the trunk back on half weight pairs (fp32 trunk = 'f16x3'); PSNR-gated
                                                                 # only: 2.7e-2 max on trained-like trunk statistics (f16x3: 1.1e-4)
            precision = 'f16'
            kw.setdefault('f16_pairs', 3)
        object.__setattr__(self, 'precision', {'fp32': 'fp32', 'f32': 'fp32', 'bf16': 'bf16', 'f16': 'f16', 'fp16': 'f16', 'half': 'f16'}[precision])
        for f in self._C_FIELDS:
            object.__setattr__(self, f, int(kw.pop(f, 0)))
        if kw:
            raise TypeError(f'unknown option(s): {sorted(kw)}')
        st = None
        if any(getattr(self, f) for f in self._C_FIELDS):
            st = _lib.OptionsT()
            for f in self._C_FIELDS:
                setattr(st, f, getattr(self, f))
        object.__setattr__(self, '_c', st)

    def __setattr__(self, k, v):
        raise AttributeError('Options is immutable; use replace()')

    def replace(self, **kw):
        cur = {f: getattr(self, f) for f in ('precision',) + self._C_FIELDS}
        cur.update(kw)
        return Options(**cur)

    @property
    def bf16(self):
        return self.precision == 'bf16'

    @property
    def half(self):
        """None for the fp32 entries, else the 16-bit element type of the MFMA operands: 'bf16' | 'f16'."""
        return None if self.precision == 'fp32' else self.precision

    @property
    def suffix(self):
        """Entry-point suffix of the C ABI: 'f32' | 'bf16' | 'f16'."""
        return 'f32' if self.precision == 'fp32' else self.precision

    def c_arg(self):
        """ctypes argument for `const ciaosr_options_t* opt` (NULL when every field is default)."""
        return C.byref(self._c) if self._c is not None else None

    def __repr__(self):
        extra = ''.join(f', {f}={getattr(self, f)}' for f in self._C_FIELDS if getattr(self, f))
        return f'Options({self.precision!r}{extra})'


DEFAULT_OPTIONS = Options()


def as_options(options):
    """None -> defaults; 'fp32'/'bf16'/'f16' -> Options(precision); Options -> itself; dict -> Options(**dict)."""
    if options is None:
        return DEFAULT_OPTIONS
    if isinstance(options, Options):
        return options
    if isinstance(options, str):
        return Options(options)
    if isinstance(options, dict):
        return Options(**options)
    raise TypeError(f'options must be None, str, dict or hip_ops.Options, got {type(options)}')


def gather_rows(unfold, C_, Cn, coord, cell, H, W, local_size=2, chunk=0):
    """K1 as the reference assembles it: (q_rows [Q,9C], inp_k [Q*J,9C+4], inp_v [Q*J,9C+Cn+4], q_idx, k_idx)."""
    require_gpu(unfold, coord, cell)
    Q = coord.shape[0]
    J = {1: 1, 2: 4, 3: 9}[local_size]
    D, Dv = 9 * C_, 9 * C_ + Cn
    dev = unfold.device
    q_rows = torch.empty(Q, D, dtype=torch.float32, device=dev)
    inp_k = torch.empty(Q * J, D + 4, dtype=torch.float32, device=dev)
    inp_v = torch.empty(Q * J, Dv + 4, dtype=torch.float32, device=dev)
    q_idx = torch.empty(Q, dtype=torch.int32, device=dev)
    k_idx = torch.empty(Q, J, dtype=torch.int32, device=dev)
    _lib.call('ciaosr_gather_rows_f32', ptr(unfold), unfold.stride(0), C_, Cn, ptr(coord), ptr(cell), Q, int(chunk or 0),
              H, W, local_size, ptr(q_rows), D, ptr(inp_k), D + 4, ptr(inp_v), Dv + 4, ptr(q_idx), ptr(k_idx),
              stream_ptr())
    return q_rows, inp_k, inp_v, q_idx, k_idx


def mlp_forward(x, mlp_struct, n_run=0):
    """MLPRefiner.forward layer by layer (no hoist); n_run > 0 stops after that many layers (ReLU applied)."""
    require_gpu(x)
    rows = x.shape[0]
    n = n_run or mlp_struct.n_layers
    out = torch.empty(rows, mlp_struct.width[n - 1], dtype=torch.float32, device=x.device)
    nbytes = _lib.load().ciaosr_mlp_workspace_bytes(C.byref(mlp_struct), rows)
    ws = workspace(nbytes, x.device, slot='mlp')
    _lib.call('ciaosr_mlp_forward_f32', ptr(x), x.stride(0), C.byref(mlp_struct), int(n_run), rows, ptr(out),
              out.stride(0), ptr(ws), ws.numel(), stream_ptr())
    return out


def mlp_forward_16(x, mlp_struct, half='bf16'):
    """MLPRefiner.forward with every Linear on the 16-bit MFMA GEMM (ciaosr_mlp_forward_bf16 / _f16): fp32 in, fp32 out."""
    require_gpu(x)
    rows = x.shape[0]
    out = torch.empty(rows, mlp_struct.width[mlp_struct.n_layers - 1], dtype=torch.float32, device=x.device)
    nbytes = _lib.load().ciaosr_mlp_workspace_bytes_16(C.byref(mlp_struct), rows)
    ws = workspace(nbytes, x.device, slot='mlp16')
    _lib.call('ciaosr_mlp_forward_' + half, ptr(x), x.stride(0), C.byref(mlp_struct), rows, ptr(out), out.stride(0), ptr(ws), ws.numel(),
              stream_ptr())
    return out


def decode_residual(h, w_last, b_last, x_lr_chw, coord, H, W):
    require_gpu(h, w_last, b_last, x_lr_chw, coord)
    Q = h.shape[0]
    rgb = torch.empty(Q, 3, dtype=torch.float32, device=h.device)
    _lib.call('ciaosr_decode_residual_f32', ptr(h), h.stride(0), h.shape[1], ptr(w_last), w_last.stride(0), ptr(b_last),
              ptr(x_lr_chw), ptr(coord), Q, H, W, ptr(rgb), stream_ptr())
    return rgb


class profile:
    """Context manager around the library's per-kernel HIP-event timing."""

    def __init__(self, only=None):
        self.only = only

    def __enter__(self):
        lib = _lib.load()
        lib.ciaosr_prof_filter(self.only.encode() if self.only else None)
        lib.ciaosr_prof_reset()
        lib.ciaosr_prof_enable(1)
        return self

    def __exit__(self, *exc):
        lib = _lib.load()
        lib.ciaosr_prof_collect()
        lib.ciaosr_prof_enable(0)
        return False

    @staticmethod
    def results():
        lib = _lib.load()
        lib.ciaosr_prof_collect()
        buf = C.create_string_buffer(8192)
        lib.ciaosr_prof_names(buf, 8192)
        out = {}
        for name in [n for n in buf.value.decode().split(';') if n]:
            ms, n = C.c_double(0), C.c_long(0)
            lib.ciaosr_prof_get(name.encode(), C.byref(ms), C.byref(n))
            if n.value:
                out[name] = dict(total_ms=ms.value, launches=n.value, avg_ms=ms.value / n.value)
        return out
