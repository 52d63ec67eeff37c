"""CrossScaleAttention ("scale-aware non-local attention"): parameters with the reference's
names (arch_csnln.py:407-428) and a HIP forward (ciaosr_cs_attn_f32, arch_csnln.py:430-532).

state_dict: conv_match_1.0.{weight,bias}, conv_match_1.1.weight (PReLU), conv_match_2.*,
conv_assembly.*, (downx3.*, downx4.*,) down.{weight,bias}, buffer escape_NaN.  Scale entries 2, 3 and 4 and lists of them
(arch_csnln.py:421-427, :436-528); the configs use scale=[2] (ciaosr_net.py:44), the only one with the composed tail and a bf16 route.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, hip_ops


def _conv_prelu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 1, padding=0, bias=True), nn.PReLU())


class CrossScaleAttention(nn.Module):
    def __init__(self, channel=64, reduction=2, ksize=3, scale=2, stride=1, softmax_scale=10, average=True):
        super().__init__()
        scale = list(scale) if isinstance(scale, (list, tuple)) else [scale]
        if not scale or any(s_ not in (2, 3, 4) for s_ in scale) or ksize != 3 or stride != 1 or not average:
            raise NotImplementedError('CrossScaleAttention needs scale entries in {2, 3, 4}, ksize=3, stride=1, average=True '
                                      '(every CiaoSR config uses scale=[2])')
        if reduction != 2:
            raise NotImplementedError('reduction must be 2')
        self.channel, self.scale, self.softmax_scale = channel, scale, softmax_scale
        self.register_buffer('escape_NaN', torch.FloatTensor([1e-4]))
        self.conv_match_1 = _conv_prelu(channel, channel // reduction)
        self.conv_match_2 = _conv_prelu(channel, channel // reduction)
        self.conv_assembly = _conv_prelu(channel, channel)
        if 3 in scale:                                        # arch_csnln.py:421-427 (same creation order)
            self.downx3 = nn.Conv2d(channel, channel, ksize, 3, 1)
        if 4 in scale:
            self.downx4 = nn.Conv2d(channel, channel, ksize, 4, 1)
        self.down = nn.Conv2d(channel, channel, ksize, 2, 1)
        self._packed = None

    # -- weight packing -------------------------------------------------------------------------
    def _version_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed(self):
        key = self._version_key()
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        Cc = self.channel
        half = Cc // 2
        pad = (-half) % 4            # the kernels move channels as float4: zero-pad C/2 to a multiple of 4
        padw = lambda w: torch.nn.functional.pad(w.detach().reshape(half, Cc).float(), (0, 0, 0, pad)).contiguous()
        padb = lambda b: torch.nn.functional.pad(b.detach().float(), (0, pad)).contiguous()
        keep = dict(
            w1=padw(self.conv_match_1[0].weight), b1=padb(self.conv_match_1[0].bias),
            w2=padw(self.conv_match_2[0].weight), b2=padb(self.conv_match_2[0].bias),
            wa=self.conv_assembly[0].weight.detach().reshape(Cc, Cc).contiguous().float(),
            ba=self.conv_assembly[0].bias.detach().contiguous().float())
        downs = {2: self.down, 3: getattr(self, 'downx3', None), 4: getattr(self, 'downx4', None)}
        for s_ in self.scale:
            # down.weight [co][ci][a][b] -> [co][(a*3+b)*C + ci]
            keep[f'wd{s_}'] = downs[s_].weight.detach().permute(0, 2, 3, 1).reshape(Cc, 9 * Cc).contiguous().float()
            keep[f'bd{s_}'] = downs[s_].bias.detach().contiguous().float()
        # composed fold+down form (scale 2 only): `down` masked per tap subset, [9][C][9C] (include/ciaosr_hip.h)
        if 2 in self.scale:
            wdm = []
            subsets = ([0], [0, 1, 2], [1, 2])
            wd4 = self.down.weight.detach().float()                      # [co][ci][a][b]
            for R in subsets:
                for S in subsets:
                    m = torch.zeros_like(wd4)
                    for a in R:
                        for b in S:
                            m[:, :, a, b] = wd4[:, :, a, b]
                    wdm.append(m.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc))
            keep['wdm'] = torch.stack(wdm).contiguous()
        hip_ops.require_gpu(*keep.values())
        sts = (_lib.CsAttnWeightsT * len(self.scale))()           # one struct per scale entry, shared match / assembly weights
        for i, s_ in enumerate(self.scale):
            st = sts[i]
            st.channels, st.scale = Cc, int(s_)
            st.w_match1, st.b_match1 = keep['w1'].data_ptr(), keep['b1'].data_ptr()
            st.w_match2, st.b_match2 = keep['w2'].data_ptr(), keep['b2'].data_ptr()
            st.w_assembly, st.b_assembly = keep['wa'].data_ptr(), keep['ba'].data_ptr()
            st.w_down, st.b_down = keep[f'wd{s_}'].data_ptr(), keep[f'bd{s_}'].data_ptr()
            st.w_down_masked = keep['wdm'].data_ptr() if s_ == 2 else None
            st.slope_match1 = float(self.conv_match_1[1].weight.detach().float().cpu()[0])
            st.slope_match2 = float(self.conv_match_2[1].weight.detach().float().cpu()[0])
            st.slope_assembly = float(self.conv_assembly[1].weight.detach().float().cpu()[0])
            st.escape_nan = float(self.escape_NaN.detach().float().cpu()[0])
            st.softmax_scale = float(self.softmax_scale)
        st = sts
        self._packed = (key, (st, keep))
        return self._packed[1]

    @torch.no_grad()
    def forward(self, input, options=None):
        """[B,C,H,W] -> [B,C,H,W] like the reference module (batch items are independent,
        arch_csnln.py:491).  `options`: hip_ops.Options (precision, csa_composed_min); None = fp32 defaults."""
        opt = hip_ops.as_options(options)
        x = input.contiguous().float()
        hip_ops.require_gpu(x)
        B, Cc, H, W = x.shape
        sts, _keep = self.packed()
        ns = len(self.scale)
        nbytes = _lib.load().ciaosr_cs_attn_workspace_bytes_scale(H, W, Cc, max(self.scale))
        ws = hip_ops.workspace(nbytes, x.device)
        out = torch.empty(B, ns * Cc, H, W, dtype=torch.float32, device=x.device)       # torch.cat(res_y, dim=1), csa:528
        for b in range(B):
            f = hip_ops.nchw_to_hwc(x[b])
            o = torch.empty(H, W, ns * Cc, dtype=torch.float32, device=x.device)
            for i in range(ns):           # scale i writes channels [i*C, (i+1)*C) of the channels-last rows
                o_i = o.view(-1)[i * Cc:]
                _lib.call('ciaosr_cs_attn_' + opt.suffix, hip_ops.ptr(f), Cc, H, W, C.byref(sts[i]),
                          hip_ops.ptr(o_i), ns * Cc, opt.c_arg(), hip_ops.ptr(ws), ws.numel(), hip_ops.stream_ptr())
            out[b] = hip_ops.hwc_to_nchw(o)
        return out
