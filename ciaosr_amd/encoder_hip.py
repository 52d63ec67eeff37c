"""Weight packing + launch of the HIP encoder trunks (ciaosr_rdn_forward_f32 / ciaosr_edsr_forward_f32).

Conv weights [co][ci][a][b] are packed once to [co][(a*k+b)*ci' + ci] so that one tap of the
channels-last map is contiguous in K (implicit-GEMM convolution, csrc/conv_f32.hip)."""
import ctypes as C

import torch

from . import _lib, hip_ops


def _ceil32(c):
    return (c + 31) // 32 * 32


def _widen(w, b, widen):
    """Zero-pad a conv's channel GROUPS from width c to the next multiple of 32 (the implicit-GEMM kernels take 32-channel K
    steps): weight [co][g*c][k][k] -> [co'][g*c'][k][k], bias [co] -> [co'].  The extra output channels are exactly 0 (zero
    weights, zero bias; relu(0) = 0) and the extra input channels meet zero weights, so the first c channels of every group are
    the unpadded network's values bit for bit up to the order of the (unchanged) non-zero products."""
    if widen is None:
        return w, b
    c, cp = widen
    co, ci, kh, kw = w.shape
    if ci % c == 0 and ci >= c:
        g = ci // c
        w = torch.nn.functional.pad(w.view(co, g, c, kh, kw), (0, 0, 0, 0, 0, cp - c)).reshape(co, g * cp, kh, kw)
    if co == c:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cp - c))
        b = torch.nn.functional.pad(b, (0, cp - c))
    return w.contiguous(), b.contiguous()


def _pack_conv(conv, keep, pad_cin_to=None, frag16=False, frag=False, widen=None):
    w, b0 = _widen(conv.weight.detach().float(), conv.bias.detach().float(), widen)
    co, ci, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if pad_cin_to is not None and ci < pad_cin_to:
        w = torch.nn.functional.pad(w, (0, pad_cin_to - ci))
    w = w.reshape(co, -1).contiguous()
    b = b0.contiguous()
    hip_ops.require_gpu(w, b)
    keep += [w, b]
    st = _lib.ConvT()
    st.weight, st.bias, st.cin, st.cout, st.ksize = w.data_ptr(), b.data_ptr(), ci, co, kh
    st.frag16 = None
    st.frag16_lo = None
    st.frag = None
    st.frag_wino = None
    st.frag_wino4 = None
    if frag16:
        # every fp32 fragment form of the dense layer in ONE launch (csrc/pack_ops.hip): the direct [cout][9 cin] matrix for the halo-resident
        # kernel of big maps (dense_f32.hip) and, for the 64 -> 64-per-group layers, the Winograd F(2x2, 3x3) / F(4x4, 3x3) forms
        # U = G g G^T (dense_wino_f32.hip / dense_wino4_f32.hip; transform in fp64 on the device, rounded once).  The 16-bit fragments of
        # the bf16 / f16 trunk modes are packed when such a mode is first used (PackedEncoder.struct).
        lib = _lib.load()
        n_, k_ = w.shape
        f32 = torch.empty(lib.ciaosr_fragment_floats(n_, k_), dtype=torch.float32, device=w.device)
        wino = kh == 3 and co == 64 and ci % 64 == 0
        fw = fw4 = None
        if wino:
            nfl = lib.ciaosr_fragment_floats(co, ci)
            fw = torch.empty(16 * nfl, dtype=torch.float32, device=w.device)
            fw4 = torch.empty(36 * nfl, dtype=torch.float32, device=w.device)
        if kh == 3:
            _lib.call('ciaosr_pack_conv3x3_f32', hip_ops.ptr(w), 9 * ci, 3 * ci, ci, 1, co, ci, hip_ops.ptr(f32), hip_ops.ptr(fw), hip_ops.ptr(fw4),
                      hip_ops.stream_ptr())
        else:
            _lib.call('ciaosr_pack_fragments_f32', hip_ops.ptr(w), w.stride(0), n_, k_, hip_ops.ptr(f32), hip_ops.stream_ptr())
        keep.append(f32)
        st.frag = f32.data_ptr()
        if wino:
            keep += [fw, fw4]
            st.frag_wino, st.frag_wino4 = fw.data_ptr(), fw4.data_ptr()
    elif frag:
        # exact-fp32 MFMA fragments only: the one-launch small-map 3x3 kernel (conv_small_f32.hip)
        n_, k_ = w.shape
        f32 = torch.empty(_lib.load().ciaosr_fragment_floats(n_, k_), dtype=torch.float32, device=w.device)
        _lib.call('ciaosr_pack_fragments_f32', hip_ops.ptr(w), w.stride(0), n_, k_, hip_ops.ptr(f32), hip_ops.stream_ptr())
        keep.append(f32)
        st.frag = f32.data_ptr()
    return st


class PackedEncoder:
    """Packed weights of the re-parented RDN / EDSR trunk of a LocalImplicitSR generator."""

    def __init__(self, net, kind):
        self.net, self.kind = net, kind
        self._key = None
        self._st = None
        self._keep = None
        self._st_half = {}        # RDN: per 16-bit element type, the copy of the struct whose dense layers carry its fragment pairs

    def _params(self):
        n = self.net
        mods = [n.sfe1, n.sfe2, n.rdbs, n.gff] if self.kind == 'rdn' else [n.conv_first, n.body, n.conv_after_body]
        return [p for m in mods for p in m.parameters()]

    def struct(self, half=None):
        """The trunk's weight struct.  half='bf16' | 'f16' (RDN): the copy whose dense layers carry the 16-bit hi + lo fragment pairs of that
        element type, packed when the mode is first used (one launch pair per layer); None: the fp32 forms only."""
        key = tuple((p.data_ptr(), p._version) for p in self._params())
        if self._st is None or key != self._key:
            self._build(key)
            self._st_half = {}
        if half not in ('bf16', 'f16') or self.kind != 'rdn':
            return self._st
        if half not in self._st_half:
            base = self._st
            st = _lib.RdnWeightsT()
            C.memmove(C.byref(st), C.byref(base), C.sizeof(st))
            nd = base.num_blocks * base.num_layers
            dense = (_lib.ConvT * nd)()
            keep = [dense]
            lib = _lib.load()
            for i in range(nd):
                C.memmove(C.byref(dense[i]), C.byref(base.dense[i]), C.sizeof(_lib.ConvT))
                c = dense[i]
                n_, k_ = c.cout, c.ksize * c.ksize * c.cin
                f = torch.empty(getattr(lib, f'ciaosr_fragment_{half}_bytes')(n_, k_), dtype=torch.uint8, device=self._keep[0].device)
                lo = torch.empty_like(f)                     # h16(w - h16(w)): the lo half of the weight pair (bf16 default, Options(f16_pairs=1))
                _lib.call(f'ciaosr_pack_fragments_{half}_pair', c.weight, k_, n_, k_, hip_ops.ptr(f), hip_ops.ptr(lo), hip_ops.stream_ptr())
                keep += [f, lo]
                c.frag16 = f.data_ptr()
                c.frag16_lo = lo.data_ptr()
            st.dense = dense
            self._st_half[half] = (st, keep)
        return self._st_half[half][0]

    def width(self):
        """(c, c'): the trunk's channel width and the width the kernels run it at (next multiple of 32, zero-padded weights)."""
        n = self.net
        c = (n.sfe1 if self.kind == 'rdn' else n.conv_first).out_channels
        return c, _ceil32(c)

    def _build(self, key):
        n, keep = self.net, []
        c, cp = self.width()
        wd = (c, cp) if cp != c else None
        if self.kind == 'rdn':
            st = _lib.RdnWeightsT()
            nb, nl = len(n.rdbs), len(n.rdbs[0].layers)
            st.mid_channels = cp
            st.growth = cp
            st.num_blocks, st.num_layers = nb, nl
            st.sfe1 = _pack_conv(n.sfe1, keep, pad_cin_to=4, widen=wd)
            st.sfe2 = _pack_conv(n.sfe2, keep, frag=True, widen=wd)
            st.gff0 = _pack_conv(n.gff[0], keep, widen=wd)
            st.gff1 = _pack_conv(n.gff[1], keep, frag=True, widen=wd)
            dense = (_lib.ConvT * (nb * nl))()
            lff = (_lib.ConvT * nb)()
            for b in range(nb):
                for l in range(nl):
                    dense[b * nl + l] = _pack_conv(n.rdbs[b].layers[l].conv, keep, frag16=True, widen=wd)
                lff[b] = _pack_conv(n.rdbs[b].lff, keep, widen=wd)
            st.dense, st.lff = dense, lff
            keep += [dense, lff]
            st.scatter_weight, st.scatter_bias, st.scatter_frag = None, None, None
            if st.mid_channels == 64 and st.growth == 64:
                # scatter form of the dense blocks (include/ciaosr_hip.h): stack, for input group s, the weight
                # slices of every later layer -> one N = 64*(nl-s), K = 576 convolution per group
                ptrs = (C.c_void_p * (nb * nl))()
                fptrs = (C.c_void_p * (nb * nl))()
                lib = _lib.load()
                for b in range(nb):
                    convs = [_widen(n.rdbs[b].layers[l].conv.weight.detach().float(), n.rdbs[b].layers[l].conv.bias.detach().float(), wd)[0]
                             for l in range(nl)]                                                 # [64][64(l+1)][3][3]
                    # the nl stacked matrices of the block, row-stacked into one [64 (nl + ... + 1)][576] matrix: ONE fragment-pack launch per
                    # block (fragments are per 32-row tile, every group's row count is a multiple of 64: group s's fragments are a slice)
                    sl = [convs[l][:, 64 * s_:64 * s_ + 64].permute(0, 2, 3, 1).reshape(64, 576) for s_ in range(nl) for l in range(s_, nl)]
                    wblk = torch.cat(sl, 0).contiguous()
                    fblk = torch.empty(lib.ciaosr_fragment_floats(wblk.shape[0], 576), dtype=torch.float32, device=wblk.device)
                    _lib.call('ciaosr_pack_fragments_f32', hip_ops.ptr(wblk), 576, wblk.shape[0], 576, hip_ops.ptr(fblk), hip_ops.stream_ptr())
                    keep += [wblk, fblk]
                    row = 0
                    for s_ in range(nl):
                        ptrs[b * nl + s_] = wblk.data_ptr() + 4 * row * 576
                        fptrs[b * nl + s_] = fblk.data_ptr() + 4 * lib.ciaosr_fragment_floats(row, 576) if row else fblk.data_ptr()
                        row += 64 * (nl - s_)
                bias = torch.stack([torch.stack([_widen(n.rdbs[b].layers[l].conv.weight.detach().float(),
                                                        n.rdbs[b].layers[l].conv.bias.detach().float(), wd)[1] for l in range(nl)])
                                    for b in range(nb)]).contiguous()
                keep += [ptrs, fptrs, bias]
                st.scatter_frag = C.cast(fptrs, C.POINTER(C.c_void_p))
                st.scatter_weight = C.cast(ptrs, C.POINTER(C.c_void_p))
                st.scatter_bias = bias.data_ptr()
        else:
            st = _lib.EdsrWeightsT()
            nb = len(n.body)
            st.mid_channels = cp
            st.num_blocks = nb
            st.res_scale = float(n.body[0].res_scale) if nb else 1.0
            st.conv_first = _pack_conv(n.conv_first, keep, pad_cin_to=4, widen=wd)
            st.conv_after_body = _pack_conv(n.conv_after_body, keep, frag=True, widen=wd)
            c1 = (_lib.ConvT * max(nb, 1))()
            c2 = (_lib.ConvT * max(nb, 1))()
            for b in range(nb):
                c1[b] = _pack_conv(n.body[b].conv1, keep, frag=True, widen=wd)
                c2[b] = _pack_conv(n.body[b].conv2, keep, frag=True, widen=wd)
            st.conv1, st.conv2 = c1, c2
            keep += [c1, c2]
        self._st, self._keep, self._key = st, keep, key

    def supported(self):
        """Every width runs on the HIP trunk (narrow / odd widths zero-padded to a multiple of 32 at packing time).  Not covered:
        other than 3 input channels, and an RDN whose growth differs from its width (mmedit's RDN feeds rdbs[b > 0] with
        `channel_growth` channels and adds sfe1 (`mid_channels`) at the end, so it cannot even run with the two different)."""
        n = self.net
        if self.kind == 'rdn':
            return n.rdbs[0].layers[0].conv.out_channels == n.sfe1.out_channels and n.sfe1.in_channels == 3
        return n.conv_first.in_channels == 3

    def why_unsupported(self):
        n = self.net
        if self.kind == 'rdn':
            return (f'RDN trunk with in_channels={n.sfe1.in_channels}, mid_channels={n.sfe1.out_channels}, '
                    f'channel_growth={n.rdbs[0].layers[0].conv.out_channels}: the HIP trunk needs in_channels=3 and '
                    f'channel_growth == mid_channels')
        return f'EDSR trunk with in_channels={n.conv_first.in_channels}: the HIP trunk needs in_channels=3'

    def _narrow(self, out):
        c, cp = self.width()
        return out if c == cp else out[..., :c].contiguous()

    @torch.no_grad()
    def forward_hwc_batch(self, x_bchw, options=None):
        """x [B,3,H,W] normalised LR crops of one size (GPU) -> features [B,H,W,C] channels-last, the B images sharing the
        trunk's dense-layer launches (ciaosr_rdn_forward_batch_*; each image bitwise equal to a forward_hwc call)."""
        B = x_bchw.shape[0]
        if self.kind != 'rdn' or B == 1:
            return torch.stack([self.forward_hwc(x_bchw[i], options) for i in range(B)])
        opt = hip_ops.as_options(options)
        x_bchw = x_bchw.contiguous().float()
        hip_ops.require_gpu(x_bchw)
        _, _, H, W = x_bchw.shape
        st = self.struct(opt.half)
        nbytes = _lib.load().ciaosr_rdn_workspace_bytes_batch(B, H, W, C.byref(st))
        ws = hip_ops.workspace(nbytes, x_bchw.device, slot='encoder')
        out = torch.empty(B, H, W, st.mid_channels, dtype=torch.float32, device=x_bchw.device)
        _lib.call('ciaosr_rdn_forward_batch_' + opt.suffix, hip_ops.ptr(x_bchw), B, H, W, C.byref(st), hip_ops.ptr(out), opt.c_arg(),
                  hip_ops.ptr(ws), ws.numel(), hip_ops.stream_ptr())
        return self._narrow(out)

    @torch.no_grad()
    def forward_hwc(self, x_chw, options=None):
        """x [3,H,W] normalised LR (GPU) -> feature [H,W,C] channels-last.  `options`: hip_ops.Options."""
        opt = hip_ops.as_options(options)
        x_chw = x_chw.contiguous().float()
        hip_ops.require_gpu(x_chw)
        _, H, W = x_chw.shape
        st = self.struct(opt.half)
        lib = _lib.load()
        if self.kind == 'rdn':
            nbytes = lib.ciaosr_rdn_workspace_bytes(H, W, C.byref(st))
            fn = 'ciaosr_rdn_forward_' + opt.suffix
        else:
            nbytes = lib.ciaosr_edsr_workspace_bytes(H, W, C.byref(st))
            fn = 'ciaosr_edsr_forward_f32'
        ws = hip_ops.workspace(nbytes, x_chw.device, slot='encoder')
        out = torch.empty(H, W, st.mid_channels, dtype=torch.float32, device=x_chw.device)
        if self.kind == 'rdn':
            _lib.call(fn, hip_ops.ptr(x_chw), H, W, C.byref(st), hip_ops.ptr(out), opt.c_arg(), hip_ops.ptr(ws), ws.numel(),
                      hip_ops.stream_ptr())
        else:
            _lib.call(fn, hip_ops.ptr(x_chw), H, W, C.byref(st), hip_ops.ptr(out), hip_ops.ptr(ws), ws.numel(),
                      hip_ops.stream_ptr())
        return self._narrow(out)
