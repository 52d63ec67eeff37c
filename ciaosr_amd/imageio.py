"""Image file I/O for the test pipeline (LoadImageFromFile / mmcv.imwrite counterparts; mmcv is external).
Files are read as RGB (`channel_order='rgb'`, configs/001_*_rdn_*.py:100-112); `tensor2img` output is BGR."""
import os

import numpy as np
import torch


def imread_rgb01(path):
    """-> float32 tensor [3,H,W] in [0,1] (LoadImageFromFile + RescaleToZeroOne + ImageToTensor)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert('RGB'), dtype=np.float32) / 255.0
    return torch.from_numpy(img).permute(2, 0, 1).contiguous()


def imwrite(img_bgr_u8, path):
    """mmcv.imwrite of a tensor2img result (HxWx3 BGR uint8)."""
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)) or '.', exist_ok=True)
    arr = np.asarray(img_bgr_u8)
    if arr.ndim == 3:
        arr = arr[:, :, ::-1]
    Image.fromarray(np.ascontiguousarray(arr.astype(np.uint8))).save(path)
