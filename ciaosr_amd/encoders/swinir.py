"""SwinIR deep-feature trunk (encoder of LocalImplicitSRSWINIR, config C5).

Restates mmedited/models/backbones/sr_backbones/swinir_net.py (SwinIR :619-805, RSTB :420-493,
BasicLayer :350-417, SwinTransformerBlock :165-298, WindowAttention :66-146, Mlp :15-34,
PatchEmbed/PatchUnEmbed :496-570) with the same parameter and buffer names, so reference checkpoints
load unchanged:  conv_first, patch_embed.norm, layers.N.residual_group.blocks.M.{norm1, attn.{qkv, proj,
relative_position_bias_table, relative_position_index}, attn_mask, norm2, mlp.fc1, mlp.fc2}, layers.N.conv,
norm, conv_after_body.  Differences from the reference file: no hard-coded `.cuda()` in the constructor
(:684,723,725), no timm dependency (DropPath is the identity at inference), and the image-reconstruction
tail (upsample / conv_last) is not built because the CiaoSR adapter never uses it (ciaosr_net.py:460-473).
This trunk runs through PyTorch-ROCm (encoders are the "next" row, not hand-written HIP yet).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing='ij')).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += wh - 1
        rel[:, :, 1] += ww - 1
        rel[:, :, 0] *= 2 * ww - 1
        self.register_buffer('relative_position_index', rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.softmax = nn.Softmax(dim=-1)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        qkv = self.qkv(x).reshape(B_, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        n = self.window_size[0] * self.window_size[1]
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
        attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(B_ // nW, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, N, N)
        attn = self.softmax(attn)
        return self.proj((attn @ v).transpose(1, 2).reshape(B_, N, C))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, 'shift_size must in 0-window_size'
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, _pair(self.window_size), num_heads, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.register_buffer('attn_mask', self.calculate_mask(self.input_resolution) if self.shift_size > 0 else None)

    def calculate_mask(self, x_size):
        H, W = x_size
        img_mask = torch.zeros((1, H, W, 1))
        sl = (slice(0, -self.window_size), slice(-self.window_size, -self.shift_size), slice(-self.shift_size, None))
        cnt = 0
        for h in sl:
            for w in sl:
                img_mask[:, h, w, :] = cnt
                cnt += 1
        mw = window_partition(img_mask, self.window_size).view(-1, self.window_size * self.window_size)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))

    def forward(self, x, x_size):
        H, W = x_size
        B, L, C = x.shape
        shortcut = x
        x = self.norm1(x).view(B, H, W, C)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
        xw = window_partition(x, self.window_size).view(-1, self.window_size * self.window_size, C)
        if self.input_resolution == tuple(x_size):
            aw = self.attn(xw, mask=self.attn_mask)
        else:
            aw = self.attn(xw, mask=self.calculate_mask(x_size).to(x.device) if self.shift_size > 0 else None)
        x = window_reverse(aw.view(-1, self.window_size, self.window_size, C), self.window_size, H, W)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        x = shortcut + x.view(B, H * W, C)
        return x + self.mlp(self.norm2(x))


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if (i % 2 == 0) else window_size // 2, mlp_ratio, qkv_bias, qk_scale)
            for i in range(depth)])

    def forward(self, x, x_size):
        for blk in self.blocks:
            x = blk(x, x_size)
        return x


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size, patch_size = _pair(img_size), _pair(patch_size)
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        x = x.flatten(2).transpose(1, 2)
        return self.norm(x) if self.norm is not None else x


class PatchUnEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.embed_dim = embed_dim

    def forward(self, x, x_size):
        B, HW, C = x.shape
        return x.transpose(1, 2).view(B, self.embed_dim, x_size[0], x_size[1])


class RSTB(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 img_size=224, patch_size=4, resi_connection='1conv'):
        super().__init__()
        self.residual_group = BasicLayer(dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale)
        if resi_connection == '1conv':
            self.conv = nn.Conv2d(dim, dim, 3, 1, 1)
        else:
            self.conv = nn.Sequential(nn.Conv2d(dim, dim // 4, 3, 1, 1), nn.LeakyReLU(0.2, inplace=True),
                                      nn.Conv2d(dim // 4, dim // 4, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True),
                                      nn.Conv2d(dim // 4, dim, 3, 1, 1))
        self.patch_embed = PatchEmbed(img_size, patch_size, 0, dim, None)
        self.patch_unembed = PatchUnEmbed(img_size, patch_size, 0, dim, None)

    def forward(self, x, x_size):
        return self.patch_embed(self.conv(self.patch_unembed(self.residual_group(x, x_size), x_size))) + x


class SwinIR(nn.Module):
    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=[6, 6, 6, 6], num_heads=[6, 6, 6, 6],
                 window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True, use_checkpoint=False,
                 upscale=2, img_range=1., upsampler='', resi_connection='1conv', **kwargs):
        super().__init__()
        if ape:
            raise NotImplementedError('ape=True is not used by any CiaoSR config')
        self.window_size, self.embed_dim, self.num_features = window_size, embed_dim, embed_dim
        self.num_layers, self.mlp_ratio, self.upscale, self.img_range = len(depths), mlp_ratio, upscale, img_range
        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.patch_embed = PatchEmbed(img_size, patch_size, embed_dim, embed_dim, norm_layer if patch_norm else None)
        self.patches_resolution = self.patch_embed.patches_resolution
        self.patch_unembed = PatchUnEmbed(img_size, patch_size, embed_dim, embed_dim, None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.layers = nn.ModuleList([
            RSTB(embed_dim, self.patches_resolution, depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias, qk_scale,
                 img_size, patch_size, resi_connection) for i in range(self.num_layers)])
        self.norm = norm_layer(self.num_features)
        if resi_connection == '1conv':
            self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        else:
            self.conv_after_body = nn.Sequential(
                nn.Conv2d(embed_dim, embed_dim // 4, 3, 1, 1), nn.LeakyReLU(0.2, inplace=True),
                nn.Conv2d(embed_dim // 4, embed_dim // 4, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True),
                nn.Conv2d(embed_dim // 4, embed_dim, 3, 1, 1))


def swinir_features(net, img):
    """LocalImplicitSRSWINIR.gen_feature (ciaosr_net.py:499-525): reflect-pad to a window multiple, trunk,
    crop.  `net` holds the re-parented SwinIR submodules."""
    ws = net.window_size
    _, _, h, w = img.shape
    ph = (ws - h % ws) % ws
    pw = (ws - w % ws) % ws
    x = net.conv_first(F.pad(img, (0, pw, 0, ph), 'reflect'))
    x_size = (x.shape[2], x.shape[3])
    t = net.pos_drop(net.patch_embed(x))
    for layer in net.layers:
        t = layer(t, x_size)
    res = net.conv_after_body(net.patch_unembed(net.norm(t), x_size)) + x
    return res[:, :, :x_size[0] - ph, :x_size[1] - pw]
