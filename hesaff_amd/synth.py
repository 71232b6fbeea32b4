"""Deterministic synthetic 8-bit grey images for tests and bench.py.

Multi-band filtered Gaussian noise: sum over (sigma, amplitude) of
amp * gaussian_filter(noise, sigma) / std, min-max stretched to 0..255.  Dense in blobs at
every scale (about 12-15 k Hessian keypoints per Mpx).
"""
import numpy as np

BANDS = ((1.5, 40.0), (3.0, 40.0), (6.0, 50.0), (12.0, 60.0), (24.0, 60.0))


def band_noise_image(height, width, seed=1234, bands=BANDS):
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(seed)
    acc = np.zeros((height, width), np.float32)
    for sigma, amp in bands:
        n = rng.standard_normal((height, width), dtype=np.float32)
        g = gaussian_filter(n, sigma)
        acc += (amp * g / g.std()).astype(np.float32)
    lo, hi = float(acc.min()), float(acc.max())
    return np.clip(np.rint((acc - lo) * (255.0 / (hi - lo))), 0, 255).astype(np.uint8)


def band_noise_batch_torch(n, height, width, seed=1234, device="cuda", bands=BANDS):
    """Same image family generated on the GPU (bench.py): returns uint8 [n, H, W].

    Not bit-identical to band_noise_image (different RNG and filter arithmetic); the
    uint8 bytes it returns are the common input of the GPU path and the CPU baseline.
    """
    import torch
    import torch.nn.functional as F

    g = torch.Generator(device=device)
    out = torch.empty((n, height, width), dtype=torch.uint8, device=device)
    for i in range(n):
        g.manual_seed(seed + i)
        acc = torch.zeros((1, 1, height, width), dtype=torch.float32, device=device)
        for sigma, amp in bands:
            x = torch.randn((1, 1, height, width), generator=g, device=device, dtype=torch.float32)
            r = int(4 * sigma + 0.5)
            t = torch.arange(-r, r + 1, device=device, dtype=torch.float32)
            k = torch.exp(-0.5 * (t / sigma) ** 2)
            k = k / k.sum()
            x = F.conv2d(F.pad(x, (r, r, 0, 0), mode="reflect"), k.view(1, 1, 1, -1))
            x = F.conv2d(F.pad(x, (0, 0, r, r), mode="reflect"), k.view(1, 1, -1, 1))
            acc += amp * x / x.std()
        lo, hi = acc.min(), acc.max()
        out[i] = torch.clamp(torch.round((acc[0, 0] - lo) * (255.0 / (hi - lo))), 0, 255).to(torch.uint8)
    return out
