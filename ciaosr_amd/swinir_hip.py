"""Weight packing + launch of the HIP SwinIR trunk (ciaosr_swinir_forward_f32, csrc/swinir.hip):
LocalImplicitSRSWINIR.gen_feature (ciaosr_net.py:475-525) without PyTorch kernels."""
import ctypes as C

import torch

from . import _lib, hip_ops
from .encoder_hip import _pack_conv


def _ld(c):
    return (c + 63) // 64 * 64


def _pad_cols(w, ld):
    w = w.detach().float()
    if w.shape[1] < ld:
        w = torch.nn.functional.pad(w, (0, ld - w.shape[1]))
    return w.contiguous()


class PackedSwinIR:
    """Packed weights of the re-parented SwinIR trunk of a LocalImplicitSRSWINIR generator."""

    def __init__(self, net):
        self.net = net
        self._key = None
        self._st = None
        self._keep = None
        self._masks = {}

    def _params(self):
        n = self.net
        return [p for m in (n.conv_first, n.patch_embed, n.layers, n.norm, n.conv_after_body) for p in m.parameters()]

    def supported(self):
        n = self.net
        try:
            blk = n.layers[0].residual_group.blocks[0]
            c = n.conv_first.out_channels
            ws = n.window_size
            ok = isinstance(n.layers[0].conv, torch.nn.Conv2d) and isinstance(n.conv_after_body, torch.nn.Conv2d)
            ok = ok and n.conv_first.in_channels == 3 and c % 4 == 0 and blk.mlp.fc1.out_features % 4 == 0
            ok = ok and ws * ws <= 64 and c % blk.num_heads == 0 and c // blk.num_heads <= 32 and (c // blk.num_heads) % 2 == 0
            ok = ok and n.patch_embed.norm is not None
            depths = {len(l.residual_group.blocks) for l in n.layers}
            return ok and len(depths) == 1 and all(b.window_size == ws for l in n.layers for b in l.residual_group.blocks)
        except (AttributeError, IndexError):
            return False

    def why_unsupported(self):
        n = self.net
        return (f'SwinIR trunk (embed_dim={n.conv_first.out_channels}, window_size={n.window_size}): the HIP trunk needs 3 input '
                f'channels, resi_connection="1conv", patch_norm=True, window_size <= 8, head_dim even and <= 32, embed_dim and the '
                f'MLP width multiples of 4, and the same depth in every group')

    def struct(self, half=None):
        key = tuple((p.data_ptr(), p._version) for p in self._params())
        if self._st is not None and key == self._key:
            return self._st
        n, keep = self.net, []
        dev = n.conv_first.weight.device
        Cc = n.conv_first.out_channels
        blk0 = n.layers[0].residual_group.blocks[0]
        heads, ws, hid = blk0.num_heads, n.window_size, blk0.mlp.fc1.out_features
        ld, ldh, d = _ld(Cc), _ld(hid), Cc // heads
        st = _lib.SwinirWeightsT()
        st.embed_dim, st.num_heads, st.window_size, st.hidden = Cc, heads, ws, hid
        st.num_groups, st.depth = len(n.layers), len(n.layers[0].residual_group.blocks)
        st.conv_first = _pack_conv(n.conv_first, keep, pad_cin_to=4)
        st.conv_after_body = _pack_conv(n.conv_after_body, keep, pad_cin_to=ld, frag=True)
        st.conv_after_body.cin = ld

        def dptr(t):
            t = t.detach().float().contiguous().to(dev)
            keep.append(t)
            return t.data_ptr()

        st.pe_norm_w, st.pe_norm_b = dptr(n.patch_embed.norm.weight), dptr(n.patch_embed.norm.bias)
        st.norm_w, st.norm_b = dptr(n.norm.weight), dptr(n.norm.bias)
        blocks = (_lib.SwinBlockT * (st.num_groups * st.depth))()
        gconv = (_lib.ConvT * st.num_groups)()
        for g, layer in enumerate(n.layers):
            for l, b in enumerate(layer.residual_group.blocks):
                sb = blocks[g * st.depth + l]
                a = b.attn
                scale = torch.ones(3 * Cc, device=dev)
                scale[:Cc] = a.scale                                    # q = (x W_q^T + b_q) * scale  (swinir_net.py:125)
                sb.ln1_w, sb.ln1_b = dptr(b.norm1.weight), dptr(b.norm1.bias)
                sb.qkv_w = dptr(_pad_cols(a.qkv.weight.detach().float() * scale[:, None], ld))
                sb.qkv_b = dptr(a.qkv.bias.detach().float() * scale)
                nn_ = ws * ws
                bias = a.relative_position_bias_table[a.relative_position_index.view(-1)].view(nn_, nn_, heads)
                sb.bias = dptr(bias.permute(2, 0, 1))
                sb.proj_w, sb.proj_b = dptr(_pad_cols(a.proj.weight, ld)), dptr(a.proj.bias)
                sb.ln2_w, sb.ln2_b = dptr(b.norm2.weight), dptr(b.norm2.bias)
                sb.fc1_w, sb.fc1_b = dptr(_pad_cols(b.mlp.fc1.weight, ld)), dptr(b.mlp.fc1.bias)
                sb.fc2_w, sb.fc2_b = dptr(_pad_cols(b.mlp.fc2.weight, ldh)), dptr(b.mlp.fc2.bias)
                sb.shift = int(b.shift_size)
                sb.mask = None
            gconv[g] = _pack_conv(layer.conv, keep, pad_cin_to=ld, frag=True)
            gconv[g].cin = ld
        st.blocks, st.group_conv = blocks, gconv
        keep += [blocks, gconv]
        self._st, self._keep, self._key = st, keep, key
        return st

    def _mask(self, hp, wp, dev):
        """calculate_mask (swinir_net.py:192-213) for the padded map size, shift = window_size // 2."""
        k = (hp, wp, str(dev))
        if k not in self._masks:
            from .encoders.swinir import shift_mask
            ws = self.net.window_size
            self._masks[k] = shift_mask(hp, wp, ws, ws // 2).contiguous().to(dev)
        return self._masks[k]

    @torch.no_grad()
    def forward_hwc(self, x_chw, options=None):
        """x [3,H,W] normalised LR (GPU) -> feature [H,W,C] channels-last."""
        x_chw = x_chw.contiguous().float()
        hip_ops.require_gpu(x_chw)
        _, H, W = x_chw.shape
        st = self.struct()
        ws = st.window_size
        hp, wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
        # shifted blocks: the block's own attn_mask buffer when the map has its input_resolution, else calculate_mask(x_size)
        # -- exactly the reference's choice (swinir_net.py:233-236)
        self._mask_keep = []
        for g, layer in enumerate(self.net.layers):
            for l, b in enumerate(layer.residual_group.blocks):
                sb = st.blocks[g * st.depth + l]
                if b.shift_size > 0:
                    if tuple(b.input_resolution) == (hp, wp) and b.attn_mask is not None:
                        m = b.attn_mask.detach().float().contiguous()
                    else:
                        m = self._mask(hp, wp, x_chw.device)
                    hip_ops.require_gpu(m)
                    self._mask_keep.append(m)
                    sb.mask = m.data_ptr()
                else:
                    sb.mask = None
        nbytes = _lib.load().ciaosr_swinir_workspace_bytes(H, W, C.byref(st))
        wsb = hip_ops.workspace(nbytes, x_chw.device, slot='encoder')
        out = torch.empty(H, W, st.embed_dim, dtype=torch.float32, device=x_chw.device)
        _lib.call('ciaosr_swinir_forward_f32', hip_ops.ptr(x_chw), H, W, C.byref(st), hip_ops.ptr(out),
                  hip_ops.ptr(wsb), wsb.numel(), hip_ops.stream_ptr())
        return out
