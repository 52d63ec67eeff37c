"""SwinIR deep-feature trunk of LocalImplicitSRSWINIR (config C5): PARAMETER CONTAINERS only.

The arithmetic runs in csrc/swinir.hip (ciaosr_amd/swinir_hip.py packs these parameters); nothing here has a
`forward`.  What this file fixes is the checkpoint contract of the reference's
mmedited/models/backbones/sr_backbones/swinir_net.py (SwinIR :619-805): the same constructor arguments and the same
parameter / buffer names, shapes and registration order, so reference checkpoints load unchanged --
conv_first, patch_embed.norm, layers.N.residual_group.blocks.M.{norm1, attn.{relative_position_bias_table,
relative_position_index, qkv, proj}, attn_mask, norm2, mlp.{fc1, fc2}}, layers.N.conv, norm, conv_after_body
(checked name for name against the reference class: tests/golden/state_dict_names_swinir.json).
Not built: the `.cuda()` calls of the reference constructor (:684,723,725), timm's DropPath (identity at
inference) and the image-reconstruction tail, which the CiaoSR adapter never uses (ciaosr_net.py:460-473).
A PyTorch evaluation of these containers exists as a test checker only (tests/torch_trunks.py).
"""
import torch
import torch.nn as nn


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def shift_mask(h, w, ws, shift):
    """Attention mask of a shifted-window block for an h x w token map (`calculate_mask`, swinir_net.py:192-213):
    [windows][ws*ws][ws*ws], 0 where two tokens of a window come from the same image region, -100 elsewhere."""
    region = torch.zeros(h, w)
    bands = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
    for i, hs in enumerate(bands):
        for j, wsl in enumerate(bands):
            region[hs, wsl] = 3 * i + j
    per_window = region.view(h // ws, ws, w // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = per_window.unsqueeze(1) - per_window.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def relative_position_index(wh, ww):
    """Index of each (token, token) pair of a wh x ww window into the (2wh-1)(2ww-1)-row bias table (swinir_net.py:91-103)."""
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing='ij')
    dy = ys.reshape(-1, 1) - ys.reshape(1, -1) + (wh - 1)
    dx = xs.reshape(-1, 1) - xs.reshape(1, -1) + (ww - 1)
    return dy * (2 * ww - 1) + dx


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        self.register_buffer('relative_position_index', relative_position_index(wh, ww))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:       # swinir_net.py:178-181
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, 'shift_size must in 0-window_size'
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, _pair(self.window_size), num_heads, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.register_buffer('attn_mask', shift_mask(*self.input_resolution, self.window_size, self.shift_size)
                             if self.shift_size > 0 else None)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if (i % 2 == 0) else window_size // 2, mlp_ratio, qkv_bias, qk_scale)
            for i in range(depth)])


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size, patch_size = _pair(img_size), _pair(patch_size)
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None


class PatchUnEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.embed_dim = embed_dim


def _tail_conv(dim, resi_connection):
    if resi_connection == '1conv':
        return nn.Conv2d(dim, dim, 3, 1, 1)
    return nn.Sequential(nn.Conv2d(dim, dim // 4, 3, 1, 1), nn.LeakyReLU(0.2, inplace=True),
                         nn.Conv2d(dim // 4, dim // 4, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True),
                         nn.Conv2d(dim // 4, dim, 3, 1, 1))


class RSTB(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 img_size=224, patch_size=4, resi_connection='1conv'):
        super().__init__()
        self.residual_group = BasicLayer(dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale)
        self.conv = _tail_conv(dim, resi_connection)
        self.patch_embed = PatchEmbed(img_size, patch_size, 0, dim, None)
        self.patch_unembed = PatchUnEmbed(img_size, patch_size, 0, dim, None)


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
        self.conv_after_body = _tail_conv(embed_dim, resi_connection)
