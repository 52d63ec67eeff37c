# CiaoSR test config, RDN encoder -- model / test_cfg / test-data sections only (inference scope).
# Same keys and import paths as the reference config of the same name; training, optimiser and logging
# sections are out of scope and omitted.  Set val_scale / data_type as in the reference.
exp_name = '001_ciaosr_rdn_div2k'
val_scale = 4
data_type = 'Urban100'   # Set5, Set14, BSDS100, Urban100, Manga109

from mmedited.models.restorers.ciaosr import CiaoSR
from mmedited.models.backbones.sr_backbones.ciaosr_net import LocalImplicitSRRDN


def _mlp(in_dim, out_dim):
    return dict(type='MLPRefiner', in_dim=in_dim, out_dim=out_dim, hidden_list=[256, 256, 256, 256])


model = dict(
    type=CiaoSR,
    generator=dict(
        type=LocalImplicitSRRDN,
        encoder=dict(type='RDN', in_channels=3, out_channels=3, mid_channels=64, num_blocks=16, upscale_factor=4,
                     num_layers=8, channel_growth=64),
        imnet_q=_mlp(4, 3), imnet_k=_mlp(64, 64), imnet_v=_mlp(64, 64),   # in/out dims are rewired from the encoder width
        feat_unfold=True,
        eval_bsize=30000),
    rgb_mean=(0.4488, 0.4371, 0.4040),
    rgb_std=(1., 1., 1.),
    pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'))

train_cfg = None
if val_scale <= 4:   # tiled inference; larger tile is better
    test_cfg = dict(metrics=['PSNR', 'SSIM'], crop_border=val_scale, scale=val_scale, tile=192, tile_overlap=32,
                    convert_to='y')
else:                # x6, x8, x12: whole image
    test_cfg = dict(metrics=['PSNR', 'SSIM'], crop_border=val_scale, scale=val_scale, convert_to='y')

data_dir = 'data'
data = dict(
    test=dict(type='SRFolderDataset',
              lq_folder=f'{data_dir}/Classical/{data_type}/LRbicx{val_scale}',
              gt_folder=f'{data_dir}/Classical/{data_type}/GTmod12',
              scale=val_scale, filename_tmpl='{}'))

dist_params = dict(backend='nccl')
test_checkpoint_path = f'./work_dirs/{exp_name}/latest.pth'
