"""RDN feature extractor (encoder for LocalImplicitSRRDN).

The reference resolves `type='RDN'` in mmedit 0.11.0's BACKBONES registry
(configs/001_localimplicitsr_rdn_...py:16); the class is not in /root/reference.
This is a restatement of the public mmedit definition with the parameter names
the CiaoSR adapter re-parents (ciaosr_net.py:314-318): sfe1, sfe2,
rdbs.N.layers.M.conv, rdbs.N.lff, gff.0, gff.1.  The x-scale upsampler and the
output conv of mmedit's RDN are never used by CiaoSR (the adapter deletes the
encoder after re-parenting) and are therefore not built.
"""
import torch
import torch.nn as nn


class DenseLayer(nn.Module):
    def __init__(self, in_channels, growth):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, growth, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return torch.cat([x, self.relu(self.conv(x))], 1)


class RDB(nn.Module):
    def __init__(self, in_channels, growth, num_layers):
        super().__init__()
        self.layers = nn.Sequential(*[
            DenseLayer(in_channels + growth * i, growth) for i in range(num_layers)])
        self.lff = nn.Conv2d(in_channels + growth * num_layers, growth, 1)

    def forward(self, x):
        return x + self.lff(self.layers(x))


class RDN(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, mid_channels=64, num_blocks=16,
                 upscale_factor=4, num_layers=8, channel_growth=64):
        super().__init__()
        self.mid_channels = mid_channels
        self.channel_growth = channel_growth
        self.num_blocks = num_blocks
        self.num_layers = num_layers
        self.sfe1 = nn.Conv2d(in_channels, mid_channels, 3, padding=1)
        self.sfe2 = nn.Conv2d(mid_channels, mid_channels, 3, padding=1)
        blocks = [RDB(mid_channels, channel_growth, num_layers)]
        blocks += [RDB(channel_growth, channel_growth, num_layers) for _ in range(num_blocks - 1)]
        self.rdbs = nn.ModuleList(blocks)
        self.gff = nn.Sequential(
            nn.Conv2d(channel_growth * num_blocks, mid_channels, 1),
            nn.Conv2d(mid_channels, mid_channels, 3, padding=1))

    def features(self, x):
        """Trunk without upsampler == LocalImplicitSRRDN.gen_feature (ciaosr_net.py:321-342)."""
        sfe1 = self.sfe1(x)
        x = self.sfe2(sfe1)
        local = []
        for blk in self.rdbs:
            x = blk(x)
            local.append(x)
        return self.gff(torch.cat(local, 1)) + sfe1
