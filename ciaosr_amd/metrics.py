"""Image conversion + PSNR used by the restorer's `evaluate` (basic_restorer.py:101-124).

`tensor2img`, `psnr` are mmedit.core functions (external); the formula followed is the
in-repo copy mmedited/core/evaluation/metrics.py:181-226 and SURVEY Appendix A.6.
Y channel = mmcv.bgr2ycbcr(img/255, y_only=True)*255 on BGR uint8 images.
"""
import numpy as np


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1)):
    """[1,3,H,W] or [3,H,W] RGB tensor in [0,1] -> HxWx3 BGR uint8 (x255, round)."""
    t = tensor.detach().float().cpu()
    while t.dim() > 3 and t.shape[0] == 1:
        t = t[0]
    t = t.clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 3:
        img = t.numpy()[[2, 1, 0], :, :].transpose(1, 2, 0)
    elif t.dim() == 2:
        img = t.numpy()
    else:
        raise ValueError(f'tensor2img expects 2-4 dims, got {tuple(tensor.shape)}')
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def bgr2y(img01):
    """mmcv.bgr2ycbcr(y_only=True) for float32 images in [0,1]; returns Y in [16/255, 235/255]."""
    img01 = img01.astype(np.float32)
    y = np.dot(img01, np.array([24.966, 128.553, 65.481], dtype=np.float32)) + 16.0
    return (y / 255.0).astype(np.float32)


def psnr(img1, img2, crop_border=0, convert_to=None):
    assert img1.shape == img2.shape, f'Image shapes are different: {img1.shape}, {img2.shape}.'
    a, b = img1.astype(np.float32), img2.astype(np.float32)
    if isinstance(convert_to, str) and convert_to.lower() == 'y':
        a = bgr2y(a / 255.) * 255.
        b = bgr2y(b / 255.) * 255.
    elif convert_to is not None:
        raise ValueError('Wrong color model. Supported values are "Y" and None.')
    if crop_border != 0:
        a = a[crop_border:-crop_border, crop_border:-crop_border, None]
        b = b[crop_border:-crop_border, crop_border:-crop_border, None]
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 20. * np.log10(255. / np.sqrt(mse))


def psnr_tensors(pred, gt, crop_border=0, convert_to='y'):
    """PSNR between two [1,3,H,W] tensors in [0,1] exactly as `evaluate` does."""
    return psnr(tensor2img(pred), tensor2img(gt), crop_border, convert_to)


def ssim(img1, img2, crop_border=0, convert_to=None):
    """SSIM (metrics.py:264-330 semantics) with a pure-numpy 11x11 gaussian 'valid' filter."""
    a, b = img1.astype(np.float32), img2.astype(np.float32)
    if isinstance(convert_to, str) and convert_to.lower() == 'y':
        a = (bgr2y(a / 255.) * 255.)[..., None]
        b = (bgr2y(b / 255.) * 255.)[..., None]
    if crop_border != 0:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()

    def filt(x):  # separable 'valid' filtering
        x = np.apply_along_axis(lambda v: np.convolve(v, k, mode='valid'), 0, x)
        return np.apply_along_axis(lambda v: np.convolve(v, k, mode='valid'), 1, x)

    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for ch in range(a.shape[2]):
        x, y = a[..., ch].astype(np.float64), b[..., ch].astype(np.float64)
        mx, my = filt(x), filt(y)
        sxx, syy, sxy = filt(x * x) - mx * mx, filt(y * y) - my * my, filt(x * y) - mx * my
        m = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
        vals.append(m.mean())
    return float(np.mean(vals))
