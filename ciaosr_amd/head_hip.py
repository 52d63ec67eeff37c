"""Weight packing + launch of the HIP head (ciaosr_head_forward_f32).

Device channel order (include/ciaosr_hip.h): an unfold row is stored (ki,kj,c)-major instead of the
reference's F.unfold order c*9+ki*3+kj (ciaosr_net.py:132), so that one 3x3 tap is C contiguous
floats of the channels-last feature map.  The permutation is folded into the MLP weights once:
  imnet_k / imnet_v layer 0 : columns   [perm(9C) | (Cn) | rel_y rel_x scale_y scale_x]
  imnet_k / imnet_v last    : rows+bias [perm(9C) | (Cn)]
  imnet_q layer 0           : columns   [perm(9C) | (Cn)]
Dot products and element-wise products over the channel axis are invariant under it.
"""
import ctypes as C

import torch

from . import _lib, hip_ops


def unfold_perm(channels, device):
    """perm[d_dev] = d_ref with d_dev = t*C + c, d_ref = c*9 + t (t = ki*3+kj)."""
    return torch.arange(9 * channels, device=device).view(channels, 9).t().reshape(-1)


class PackedHead:
    """Packed (device-order, contiguous fp32) copies of a LocalImplicitSRNet head's weights."""

    def __init__(self, net):
        self.net = net
        self._key = None
        self._st = None
        self._keep = None
        self._st_half = {}        # per 16-bit element type: (copy of the struct with that type's fragment pairs + chain stream, kept tensors)
        self._grid_opts = {}      # (Options, grid width) -> Options carrying the traversal hint

    def _version_key(self):
        mods = [self.net.imnet_q, self.net.imnet_k, self.net.imnet_v]
        return tuple((p.data_ptr(), p._version) for m in mods for p in m.parameters())

    def _pack_mlp(self, mlp, col_perm=None, row_perm=None, frag_layers=()):
        st = _lib.MlpT()
        lin = mlp.linears()
        self._bias_tmp = {}
        if len(lin) > _lib.MAX_LAYERS:
            raise _lib.CiaoSRHipError(f'MLP deeper than {_lib.MAX_LAYERS} Linear layers')
        st.n_layers = len(lin)
        st.act = mlp.act_code() if hasattr(mlp, 'act_code') else _lib.ACT_RELU
        keep = []
        srcs = []      # (layer, weight) of the layers with MFMA fragments
        for i, l in enumerate(lin):
            w = l.weight.detach().float()
            b = l.bias.detach().float()
            if i == 0 and col_perm is not None:
                w = w[:, col_perm]
            if i == len(lin) - 1 and row_perm is not None:
                w, b = w[row_perm], b[row_perm]
            if w.shape[1] % 4:
                w = torch.nn.functional.pad(w, (0, 4 - w.shape[1] % 4))
            w, b = w.contiguous(), b.contiguous()
            hip_ops.require_gpu(w, b)
            keep += [w, b]
            self._last_wb = (w, b)
            self._bias_tmp[i] = b
            st.width[i] = w.shape[0]
            st.weight[i] = w.data_ptr()
            st.ld[i] = w.stride(0)
            st.bias[i] = b.data_ptr()
            st.frag[i] = None
            st.frag16[i] = None
            st.frag16_lo[i] = None
            if i in frag_layers and w.shape[1] % 8 == 0:
                # MFMA fragment order for the fused kernels, packed on the device by the library
                n, k = w.shape
                srcs.append((i, w))
                frag = torch.empty(_lib.load().ciaosr_fragment_floats(n, k), dtype=torch.float32, device=w.device)
                _lib.call('ciaosr_pack_fragments_f32', hip_ops.ptr(w), w.stride(0), n, k, hip_ops.ptr(frag),
                          hip_ops.stream_ptr())
                keep.append(frag)
                st.frag[i] = frag.data_ptr()
        st.in_dim = lin[0].weight.shape[1]
        return st, keep, srcs

    def struct(self, half=None, single=False):
        """The ciaosr_head_weights_t of the net.  half=None: fp32 fragments only (the _f32 entry).  half='bf16' | 'f16': the copy that also
        carries the 16-bit hi + lo fragment pairs of that element type and the weight stream of the chained kv kernel -- packed when the
        mode is first used.  single=True (half='bf16' only; `Options('bf16', bf16_single=1)` = 'bf16-single'): the copy for ONE bf16 weight
        per product, packed so that it meets the PSNR gate -- see `_build_single`."""
        key = self._version_key()
        if self._st is None or self._key != key:
            self._build()
            self._st_half = {}
        if half not in ('bf16', 'f16'):
            return self._st
        if single and half == 'bf16':
            if 'bf16-single' not in self._st_half:
                self._st_half['bf16-single'] = self._build_single()
            return self._st_half['bf16-single'][0]
        if half not in self._st_half:
            st = _lib.HeadWeightsT()
            C.memmove(C.byref(st), C.byref(self._st), C.sizeof(st))
            keep = []
            lib = _lib.load()
            for name in ('k', 'v', 'q'):
                m = getattr(st, name)
                for i, w in self._srcs[name]:
                    n, k = w.shape
                    f = torch.empty(getattr(lib, f'ciaosr_fragment_{half}_bytes')(n, k), dtype=torch.uint8, device=w.device)
                    lo = torch.empty_like(f)                 # h16(w - h16(w)): the lo half of the weight pair
                    _lib.call(f'ciaosr_pack_fragments_{half}_pair', hip_ops.ptr(w), w.stride(0), n, k, hip_ops.ptr(f), hip_ops.ptr(lo), hip_ops.stream_ptr())
                    keep += [f, lo]
                    m.frag16[i] = f.data_ptr()
                    m.frag16_lo[i] = lo.data_ptr()
            keep += self._pack_chain(st, half)
            self._st_half[half] = (st, keep)
        return self._st_half[half][0]

    # ---- one bf16 weight per product that meets the gate (round 6) ---------------------------------------------------------------------
    # Round-to-nearest single-bf16 weights fail the 0.01 dB gate on the full C3 tile (0.042 dB) although the rms error is small: the weight
    # rounding error dW is ONE fixed perturbation for the whole image, and its response to the common (mean) part of the post-ReLU
    # activations is the same for every query -- a coherent error that PSNR does not average out.  Two pack-time measures remove
    # that term (tools/bf16_single_lab.py, profiles/r6_bf16_single_rounding_lab*.txt: 0.042 -> 0.0038 dB on the C3 tile, 0.0014 / 0.0027 dB
    # on the trained-like stress vectors; 0.006-0.008 dB at the 30-dB level, where the bf16 ACTIVATIONS' rms error is what shows):
    #   * error-feedback rounding along K: the running rounding error of a row is carried into the next element, so every prefix sum of
    #     dW -- the row sum in particular -- stays within half an ulp (the response to the all-ones component of the activations vanishes);
    #   * calibrated bias correction b' = b + dW E[x]: the mean input vector E[x] of every rounded layer is measured ONCE per model, at
    #     pack time, on a fixed synthetic 48x48 image through the fp32 trunk and the fp32 staged head (input-independent, deterministic).
    # The kernels are the bf16_single ones; nothing changes at run time.
    @staticmethod
    def _ef_round_bf16(w):
        """Error-feedback round-to-bf16 of every row of w [N, K] (fp32, any device) along K; returns fp32 values that ARE bf16 numbers."""
        w64 = w.detach().double().cpu()
        out = torch.empty(w64.shape, dtype=torch.float32)
        e = torch.zeros(w64.shape[0], dtype=torch.float64)
        for k in range(w64.shape[1]):
            t = w64[:, k] + e
            q = t.float().to(torch.bfloat16).float()
            e = t - q.double()
            out[:, k] = q
        return out.to(w.device)

    @torch.no_grad()
    def _calibration_means(self):
        """Mean input vector (device channel order) of the Linear layers the bf16 head rounds: layers 1.. of imnet_k / imnet_v, layers 0..n-2
        of imnet_q, on the x4 grid of a fixed synthetic 48x48 LR image through this generator's own fp32 trunk, cs_attn and staged head."""
        from .init_utils import synthetic_pair
        net = self.net
        enc = getattr(net, '_encoder_hip', None)
        if enc is None or not enc.supported() or not getattr(net, 'feat_unfold', True):
            raise _lib.CiaoSRHipError("precision 'bf16-single' needs the HIP trunk and the unfold head for its pack-time calibration")
        st = self.struct()
        dev = net.imnet_q.layers[0].weight.device
        lq = synthetic_pair(48, 48, 4)[0].to(dev)
        x = (lq - torch.tensor((0.4488, 0.4371, 0.4040), device=dev).view(1, 3, 1, 1)).contiguous()
        feat_hwc = enc.forward_hwc(x[0], None)
        H, W, Cc = feat_hwc.shape
        Cn = st.nonlocal_channels
        U = hip_ops.patch_rows(feat_hwc, 3, 1, 1, H, W)
        if net.non_local_attn:
            nl = hip_ops.nchw_to_hwc(net.cs_attn(hip_ops.hwc_to_nchw(feat_hwc).unsqueeze(0))[0].contiguous())
            U = torch.cat([U, nl.view(H * W, Cn)], dim=1).contiguous()
        cc, cl = hip_ops.make_coord_cell(H * 4, W * 4, dev)
        q_rows, inp_k, inp_v, q_idx, k_idx = hip_ops.gather_rows(U, Cc, Cn, cc, cl, H, W, st.local_size)
        means = {}
        for nm, inp, m in (('k', inp_k, st.k), ('v', inp_v, st.v)):
            for l in range(1, m.n_layers):
                means[(nm, l)] = hip_ops.mlp_forward(inp, m, n_run=l).double().mean(0).float()
        wk, wv = hip_ops.mlp_forward(inp_k, st.k), hip_ops.mlp_forward(inp_v, st.v)
        z = hip_ops.local_attention(U, Cc, Cn, q_idx, k_idx, wk, wv, softmax_scale=st.softmax_scale)
        means[('q', 0)] = z.double().mean(0).float()
        for l in range(1, st.q.n_layers):
            means[('q', l)] = hip_ops.mlp_forward(z, st.q, n_run=l).double().mean(0).float()
        return means

    @torch.no_grad()
    def _build_single(self):
        """(struct, kept tensors) of the 'bf16-single' weight form: error-feedback-rounded weights + calibrated biases (comment above) of
        every layer the bf16 kernels take from 16-bit fragments; everything else points at the fp32 struct's tensors."""
        means = self._calibration_means()
        st = _lib.HeadWeightsT()
        C.memmove(C.byref(st), C.byref(self._st), C.sizeof(st))
        keep = []
        lib = _lib.load()
        for name in ('k', 'v', 'q'):
            m = getattr(st, name)
            for i, w in self._srcs[name]:
                n, k = w.shape
                wq = self._ef_round_bf16(w).contiguous()
                mu = means.get((name, i))
                b_old = self._bias_of[(name, i)]
                b_new = b_old.clone()
                if mu is not None:
                    d = (w.double() - wq.double())[:, :mu.numel()]
                    b_new += (d @ mu.double()).float()
                f = torch.empty(lib.ciaosr_fragment_bf16_bytes(n, k), dtype=torch.uint8, device=w.device)
                lo = torch.empty_like(f)                     # all zeros by construction (wq is a bf16 number): unused by the single kernels
                _lib.call('ciaosr_pack_fragments_bf16_pair', hip_ops.ptr(wq), wq.stride(0), n, k, hip_ops.ptr(f), hip_ops.ptr(lo), hip_ops.stream_ptr())
                keep += [wq, b_new, f, lo]
                m.weight[i] = wq.data_ptr()
                m.ld[i] = wq.stride(0)
                m.bias[i] = b_new.data_ptr()
                m.frag16[i] = f.data_ptr()
                m.frag16_lo[i] = lo.data_ptr()
        # imnet_q's 3-row output layer: the chained decode kernel takes it from the weight stream as 16-bit values too (single form: one bf16
        # per weight), so it gets the same treatment; the kernels with an fp32 tail then see bf16-valued fp32 weights + the corrected bias
        nl = st.q.n_layers - 1
        w_last, b_last = self._q_last
        wq = self._ef_round_bf16(w_last).contiguous()
        mu = means[('q', nl)]
        b_new = b_last.clone() + ((w_last.double() - wq.double())[:, :mu.numel()] @ mu.double()).float()
        keep += [wq, b_new]
        st.q.weight[nl] = wq.data_ptr()
        st.q.ld[nl] = wq.stride(0)
        st.q.bias[nl] = b_new.data_ptr()
        keep += self._pack_chain(st, 'bf16')
        return st, keep

    @staticmethod
    def _pack_chain(st, half):
        """Weight stream of the weights-stationary 16-bit head kernel (csrc/head_chain_h16.hip) for both weight forms (single 16-bit
        weights / hi + lo pairs), where the head has the shape it covers (hidden_list = [256] * 4); else the pointers stay NULL and
        the 128-row kernels run."""
        lib = _lib.load()
        keep = []
        st.chain16 = None
        st.chain16_pairs = None
        for pairs, field in ((0, 'chain16'), (1, 'chain16_pairs')):
            n = lib.ciaosr_head_chain_bytes(C.byref(st), pairs)
            if n == 0:
                break
            dev = torch.device('cuda', torch.cuda.current_device())
            blob = torch.empty(n, dtype=torch.uint8, device=dev)
            _lib.call('ciaosr_pack_head_chain_' + half, C.byref(st), pairs, hip_ops.ptr(blob), hip_ops.stream_ptr())
            keep.append(blob)
            setattr(st, field, blob.data_ptr())
        return keep

    def _build(self):
        key = self._version_key()
        net = self.net
        Cc = net.imnet_dim
        Cn = Cc * len(net.multi_scale) if net.non_local_attn else 0
        dev = net.imnet_q.layers[0].weight.device
        unfold = bool(getattr(net, 'feat_unfold', True))
        perm = unfold_perm(Cc, dev) if unfold else torch.arange(Cc, device=dev)       # feat_unfold=False: rows are the C features
        D = (9 if unfold else 1) * Cc
        tail_n = torch.arange(D, D + Cn, device=dev)
        k_cols = torch.cat([perm, torch.arange(D, D + 4, device=dev)])
        v_cols = torch.cat([perm, tail_n, torch.arange(D + Cn, D + Cn + 4, device=dev)])
        v_rows = torch.cat([perm, tail_n])
        st = _lib.HeadWeightsT()
        st.channels, st.nonlocal_channels = Cc, Cn
        st.local_size, st.softmax_scale = int(net.local_size), float(net.softmax_scale)
        st.nonlocal_max_scale = max(net.multi_scale) if (net.non_local_attn and net.multi_scale) else 0
        st.no_unfold = 0 if unfold else 1
        keep = []
        nk, nv, nq = len(net.imnet_k.linears()), len(net.imnet_v.linears()), len(net.imnet_q.linears())
        self._bias_of = {}
        st.k, kk, sk = self._pack_mlp(net.imnet_k, col_perm=k_cols, row_perm=perm, frag_layers=range(1, nk))
        self._bias_of.update({('k', i): b for i, b in self._bias_tmp.items()})
        st.k_out_wino = None
        st.k_out_wino4 = None
        st.chain16 = None
        st.chain16_pairs = None
        w5 = self._last_wb[0]                                   # imnet_k's output layer [9C][256], rows in device order (tap, c)
        if unfold and Cc == 64 and tuple(w5.shape) == (576, 256):
            # the logit table as nine 3x3 convolutions (head.hip): g[n][c][a][b] = W5[(3a+b) C + c][n] in Winograd F(2x2, 3x3) form,
            # U = G g G^T in fp64, rounded once; position p = 4 i + j as its own [256][64] matrix in MFMA fragment order
            # (both forms in one launch, transform in fp64 on the device: csrc/pack_ops.hip; element (n, a, b, c) of the convolution weight is
            # w5[(3 a + b) 64 + c][n])
            nfl = _lib.load().ciaosr_fragment_floats(256, 64)
            fw = torch.empty(16 * nfl, dtype=torch.float32, device=dev)
            fw4 = torch.empty(36 * nfl, dtype=torch.float32, device=dev)
            ld5 = w5.stride(0)
            _lib.call('ciaosr_pack_conv3x3_f32', hip_ops.ptr(w5), 1, 3 * 64 * ld5, 64 * ld5, ld5, 256, 64, None, hip_ops.ptr(fw), hip_ops.ptr(fw4),
                      hip_ops.stream_ptr())
            kk = kk + [fw, fw4]
            st.k_out_wino = fw.data_ptr()
            st.k_out_wino4 = fw4.data_ptr()
        st.v, kv, sv = self._pack_mlp(net.imnet_v, col_perm=v_cols, row_perm=v_rows, frag_layers=range(1, nv))
        self._bias_of.update({('v', i): b for i, b in self._bias_tmp.items()})
        st.q, kq, sq = self._pack_mlp(net.imnet_q, col_perm=v_rows, frag_layers=range(0, nq - 1))
        self._bias_of.update({('q', i): b for i, b in self._bias_tmp.items()})
        self._q_last = self._last_wb
        self._keep = kk + kv + kq
        self._srcs = {'k': sk, 'v': sv, 'q': sq}
        self._st, self._key = st, key

    @torch.no_grad()
    def forward(self, feature_chw, x_lr_chw, coord, cell, chunk, feature_hwc=None, options=None):
        """feature [C,H,W] (or channels-last [H,W,C] via feature_hwc), x_lr [3,H,W] or None,
        coord/cell [Q,2] -> rgb [Q,3] (all on the GPU).  `options`: hip_ops.Options (precision + route)."""
        net = self.net
        opt = hip_ops.as_options(options)
        if feature_hwc is not None:
            feature_hwc = feature_hwc.contiguous().float()
            feature_chw = feature_hwc.permute(2, 0, 1)      # shape bookkeeping only
        else:
            feature_chw = feature_chw.contiguous().float()
        coord = coord.contiguous().float()
        cell = cell.contiguous().float()
        if x_lr_chw is not None:
            x_lr_chw = x_lr_chw.contiguous().float()
        hip_ops.require_gpu(feature_hwc if feature_hwc is not None else feature_chw, x_lr_chw, coord, cell)
        Cc, H, W = feature_chw.shape
        Q = coord.shape[0]
        st = self.struct(opt.half, single=bool(opt.bf16_single))
        gw = hip_ops.grid_width_of(coord) if (opt.half and not opt.query_grid_w) else 0
        if gw:                        # traversal hint of the 16-bit chained head kernel: the queries are a make_coord grid
            key = (opt, gw)
            hinted = self._grid_opts.get(key)
            if hinted is None:
                if len(self._grid_opts) > 64:
                    self._grid_opts.clear()
                hinted = self._grid_opts[key] = opt.replace(query_grid_w=gw)
            opt = hinted
        cs = None
        if net.non_local_attn:
            cs, _ = net.cs_attn.packed()
        feat_hwc = feature_hwc if feature_hwc is not None else hip_ops.nchw_to_hwc(feature_chw)
        nbytes = _lib.load().ciaosr_head_workspace_bytes(H, W, C.byref(st), Q)
        ws = hip_ops.workspace(nbytes, coord.device)
        rgb = torch.empty(Q, 3, dtype=torch.float32, device=coord.device)
        _lib.call('ciaosr_head_forward_' + opt.suffix, hip_ops.ptr(feat_hwc), H, W,
                  C.byref(st), cs if cs is not None else None, hip_ops.ptr(x_lr_chw), hip_ops.ptr(coord),
                  hip_ops.ptr(cell), Q, int(chunk or 0), hip_ops.ptr(rgb), opt.c_arg(), hip_ops.ptr(ws), ws.numel(),
                  hip_ops.stream_ptr())
        return rgb

    @torch.no_grad()
    def forward_as_written(self, feature_chw, x_lr_chw, coord, cell, chunk=None, options=None, half=None):
        """The reference's op order, stage by stage through the staged C entry points, with NO algebraic
        restructuring (no layer-1 hoist, no logit table, no fusion): K1 gather rows (net:145-196) ->
        imnet_k / imnet_v on every (query, sample) row (net:202-206) -> K4 local attention (net:211-216) ->
        imnet_q hidden layers -> last Linear + bilinear residual (net:221, 107-108); one pass per
        eval_bsize chunk like batched_predict (net:238-246), cs_attn once.
        Used by the parity tests as a third evaluation route and to measure K1/K4 against the HBM roofline."""
        net = self.net
        if not getattr(net, 'feat_unfold', True):
            raise _lib.CiaoSRHipError('the staged K1 / K4 entry points take 3x3-unfold rows (D = 9C); feat_unfold=False runs through '
                                      'ciaosr_head_forward_f32 only')
        feature_chw = feature_chw.contiguous().float()
        coord, cell = coord.contiguous().float(), cell.contiguous().float()
        x_lr_chw = x_lr_chw.contiguous().float() if x_lr_chw is not None else None
        Cc, H, W = feature_chw.shape
        st = self.struct()
        Cn = st.nonlocal_channels
        feat_hwc = hip_ops.nchw_to_hwc(feature_chw)
        U = hip_ops.patch_rows(feat_hwc, 3, 1, 1, H, W)
        if net.non_local_attn:
            nl = hip_ops.nchw_to_hwc(net.cs_attn(feature_chw.unsqueeze(0), options=options)[0].contiguous())
            U = torch.cat([U, nl.view(H * W, Cn)], dim=1).contiguous()
        Q = coord.shape[0]
        step = int(chunk) if chunk else Q
        out = torch.empty(Q, 3, dtype=torch.float32, device=coord.device)
        nq = st.q.n_layers
        w_last, b_last = self._q_last
        for q0 in range(0, Q, step):
            q1 = min(Q, q0 + step)
            cq, cl = coord[q0:q1].contiguous(), cell[q0:q1].contiguous()
            q_rows, inp_k, inp_v, q_idx, k_idx = hip_ops.gather_rows(U, Cc, Cn, cq, cl, H, W, st.local_size)
            if half is None:
                wk = hip_ops.mlp_forward(inp_k, st.k)
                wv = hip_ops.mlp_forward(inp_v, st.v)
                z = hip_ops.local_attention(U, Cc, Cn, q_idx, k_idx, wk, wv, softmax_scale=st.softmax_scale)
            else:
                # the staged 16-bit entry points: imnet_k / imnet_v on the 16-bit GEMM, K4 on 16-bit wk / wv / z
                td = torch.bfloat16 if half == 'bf16' else torch.float16
                wk = hip_ops.mlp_forward_16(inp_k, st.k, half).to(td)
                wv = hip_ops.mlp_forward_16(inp_v, st.v, half).to(td)
                z = hip_ops.local_attention_16(U, Cc, Cn, q_idx, k_idx, wk, wv, softmax_scale=st.softmax_scale).float()
            h = hip_ops.mlp_forward(z, st.q, n_run=nq - 1)
            out[q0:q1] = hip_ops.decode_residual(h, w_last, b_last, x_lr_chw, cq, H, W)
        return out
