"""SRFolderDataset + the test pipeline of the reference configs (configs/001_*_rdn_*.py:100-120,143-149):
paired LQ/GT folders -> dict(lq, gt, coord, cell, meta).  The mmedit dataset classes are external; this is
the minimum tools/test.py needs.  GenerateCoordinateAndCell follows generate_assistant.py:56-96: the target
size is the GT size, gt is reshaped to [H*W, 3], coord = make_coord(target), cell = (2/H, 2/W)."""
import os

import torch

from .coords import make_coord, make_cell
from .imageio import imread_rgb01

IMG_EXT = ('.png', '.jpg', '.jpeg', '.bmp', '.tif', '.tiff')


class SRFolderDataset(torch.utils.data.Dataset):
    def __init__(self, lq_folder, gt_folder, pipeline=None, scale=4, test_mode=True, filename_tmpl='{}'):
        self.lq_folder, self.gt_folder, self.scale, self.filename_tmpl = str(lq_folder), str(gt_folder), scale, filename_tmpl
        names = sorted(f for f in os.listdir(self.gt_folder) if f.lower().endswith(IMG_EXT))
        self.pairs = []
        for n in names:
            stem, ext = os.path.splitext(n)
            lq = os.path.join(self.lq_folder, self.filename_tmpl.format(stem) + ext)
            if not os.path.exists(lq):
                raise FileNotFoundError(f'{lq} is not in lq_paths.')
            self.pairs.append((lq, os.path.join(self.gt_folder, n)))

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, i):
        lq_path, gt_path = self.pairs[i]
        lq, gt = imread_rgb01(lq_path), imread_rgb01(gt_path)
        ht, wt = gt.shape[-2:]
        return dict(lq=lq, gt=gt.contiguous().view(3, -1).permute(1, 0).contiguous(), coord=make_coord((ht, wt)),
                    cell=make_cell((ht, wt)), meta=dict(gt_path=gt_path, lq_path=lq_path))

    @staticmethod
    def evaluate(results):
        """Mean of every metric over the per-image eval_result dicts (mmedit BaseSRDataset.evaluate)."""
        keys = results[0]['eval_result'].keys()
        return {k: sum(r['eval_result'][k] for r in results) / len(results) for k in keys}
