"""TEST INFRASTRUCTURE (oracle): tf.image.ssim_multiscale with TF 1.15 defaults, restated in
PyTorch (sga.py:175: `tf.image.ssim_multiscale(x_tilde, x, 255)`).  PARITY UNPINNED: TF is not
installable here; this follows the published TF implementation (image_ops_impl.py: _fspecial_gauss,
_ssim_helper, _ssim_per_channel, ssim_multiscale): 5 scales, power factors (0.0448, 0.2856, 0.3001,
0.2363, 0.1333), 11x11 Gaussian (sigma 1.5, softmax-normalised), k1 = 0.01, k2 = 0.03, VALID
filtering, 2x2 average-pool downsampling with SYMMETRIC end-padding of odd sizes, relu on the
per-scale terms, product over scales, mean over channels."""
import torch
import torch.nn.functional as F

MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def fspecial_gauss(size=11, sigma=1.5, dtype=torch.float32):
    coords = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = coords ** 2 * (-0.5 / sigma ** 2)
    g = (g.reshape(1, -1) + g.reshape(-1, 1)).reshape(1, -1)
    return torch.softmax(g, dim=-1).reshape(size, size)


def _ssim_per_channel(img1, img2, max_val, size=11, sigma=1.5, k1=0.01, k2=0.03):
    """img*: [B,H,W,C]. Returns (ssim [B,C], cs [B,C])."""
    C = img1.shape[-1]
    kern = fspecial_gauss(size, sigma, img1.dtype).reshape(1, 1, size, size).repeat(C, 1, 1, 1)

    def reducer(x):
        return F.conv2d(x.permute(0, 3, 1, 2), kern, groups=C).permute(0, 2, 3, 1)

    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mean0, mean1 = reducer(img1), reducer(img2)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 ** 2 + mean1 ** 2
    luminance = (num0 + c1) / (den0 + c1)
    num1 = reducer(img1 * img2) * 2.0
    den1 = reducer(img1 ** 2 + img2 ** 2)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (luminance * cs).mean(dim=(1, 2)), cs.mean(dim=(1, 2))


def _downsample(x):
    B, H, W, C = x.shape
    t = x.permute(0, 3, 1, 2)
    if H % 2 or W % 2:      # SYMMETRIC pad at the end == replicate the last row/column
        t = F.pad(t, (0, W % 2, 0, H % 2), mode="replicate")
    return F.avg_pool2d(t, 2, 2).permute(0, 2, 3, 1)


def ssim_multiscale(img1, img2, max_val, power_factors=MSSSIM_WEIGHTS):
    """[B,H,W,C] x2 -> [B]; needs min(H,W) >= 11 * 2^4 = 176 (TF asserts the same)."""
    if min(img1.shape[1], img1.shape[2]) < 11 * 2 ** (len(power_factors) - 1):
        raise ValueError("image too small for 5-scale MS-SSIM with an 11x11 filter")
    mcs = []
    ssim = None
    for k in range(len(power_factors)):
        if k > 0:
            img1, img2 = _downsample(img1), _downsample(img2)
        ssim, cs = _ssim_per_channel(img1, img2, max_val)
        mcs.append(torch.relu(cs))
    mcs.pop()
    terms = torch.stack(mcs + [torch.relu(ssim)], dim=-1)                # [B,C,scales]
    pf = torch.tensor(power_factors, dtype=img1.dtype)
    return torch.prod(terms ** pf, dim=-1).mean(dim=-1)
