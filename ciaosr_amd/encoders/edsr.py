"""EDSR feature extractor (encoder for LocalImplicitSREDSR).

`type='EDSR'` resolves in mmedit 0.11.0 (configs/001_localimplicitsr_edsr_...py:16),
absent from /root/reference.  Restated from the public definition with the names the
adapter re-parents (ciaosr_net.py:388-390): conv_first, body.N.conv1/conv2,
conv_after_body.  mmedit's own mean shift / upsampler / conv_last are unused by
CiaoSR and not built.
"""
import torch.nn as nn


class ResidualBlockNoBN(nn.Module):
    def __init__(self, mid_channels=64, res_scale=1.0):
        super().__init__()
        self.res_scale = res_scale
        self.conv1 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1)
        self.conv2 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return x + self.conv2(self.relu(self.conv1(x))) * self.res_scale


class EDSR(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
                 upscale_factor=4, res_scale=1, rgb_mean=(0.4488, 0.4371, 0.4040),
                 rgb_std=(1.0, 1.0, 1.0)):
        super().__init__()
        self.mid_channels = mid_channels
        self.num_blocks = num_blocks
        self.conv_first = nn.Conv2d(in_channels, mid_channels, 3, 1, 1)
        self.body = nn.Sequential(*[ResidualBlockNoBN(mid_channels, res_scale)
                                    for _ in range(num_blocks)])
        self.conv_after_body = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1)

    def features(self, x):
        """== LocalImplicitSREDSR.gen_feature (ciaosr_net.py:393-408)."""
        x = self.conv_first(x)
        return self.conv_after_body(self.body(x)) + x
