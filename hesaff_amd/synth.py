"""Deterministic synthetic 8-bit grey images for tests and bench.py.

Multi-band filtered Gaussian noise: sum over (sigma, amplitude) of
amp * gaussian_filter(noise, sigma) / std, min-max stretched to 0..255.  Dense in blobs at
every scale (about 12-15 k Hessian keypoints per Mpx).
"""
import numpy as np

BANDS = ((1.5, 40.0), (3.0, 40.0), (6.0, 50.0), (12.0, 60.0), (24.0, 60.0))
# the same family one octave coarser and with a weak finest band: about 2.5 k descriptors per Mpx, the density of
# ordinary photographs (bench.py --density natural); BANDS is about five times denser
BANDS_NATURAL = ((3.0, 25.0), (6.0, 40.0), (12.0, 60.0), (24.0, 60.0), (48.0, 50.0))


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

    The bands are filtered in the frequency domain (one rfft2 per band, one irfft2 per image,
    periodic boundary) so that generating a few hundred 4K images takes seconds and uses no
    convolution library.  Not bit-identical to band_noise_image (different RNG, boundary and
    filter arithmetic); the uint8 bytes it returns are the common input of the GPU path and
    the CPU baseline.
    """
    import math
    import torch

    g = torch.Generator(device=device)
    out = torch.empty((n, height, width), dtype=torch.uint8, device=device)
    fy = torch.fft.fftfreq(height, device=device, dtype=torch.float32).view(height, 1)
    fx = torch.fft.rfftfreq(width, device=device, dtype=torch.float32).view(1, width // 2 + 1)
    f2 = fx * fx + fy * fy
    # weight of each rfft column in the full spectrum (Parseval): interior columns count twice
    wcol = torch.full((1, width // 2 + 1), 2.0, device=device)
    wcol[0, 0] = 1.0
    if width % 2 == 0:
        wcol[0, -1] = 1.0
    transfer = []
    for sigma, amp in bands:
        G = torch.exp(-2.0 * math.pi * math.pi * sigma * sigma * f2)
        var = float((G * G * wcol).sum()) / (height * width)   # variance of filtered unit white noise
        transfer.append(G * (amp / math.sqrt(var)))
    for i in range(n):
        g.manual_seed(seed + i)
        acc_hat = None
        for H in transfer:
            x = torch.randn((height, width), generator=g, device=device, dtype=torch.float32)
            t = torch.fft.rfft2(x) * H
            acc_hat = t if acc_hat is None else acc_hat + t
        acc = torch.fft.irfft2(acc_hat, s=(height, width))
        lo, hi = acc.min(), acc.max()
        out[i] = torch.clamp(torch.round((acc - lo) * (255.0 / (hi - lo))), 0, 255).to(torch.uint8)
    return out
