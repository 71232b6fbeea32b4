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


# ---- photographs (evidence on real image content) ----
# Two photographs ship with this image's scikit-learn (sklearn/datasets/images: china.jpg, flower.jpg, 640 x 427, JPEG).
# They are decoded by the in-tree JPEG reader (pixel-equal to libjpeg) and tiled into images of any size: every tile is one
# of the photographs under one of its four flips, the tiling is shifted per image index, so that a batch holds distinct
# images with the content statistics of photographs at their native resolution (4.3 k / 2.5 k descriptors per Mpx).
def sample_photo_paths():
    """Paths of the photographs, or [] when scikit-learn's sample images are not installed."""
    import importlib.util
    import os
    spec = importlib.util.find_spec("sklearn")
    if spec is None or not spec.submodule_search_locations:
        return []
    d = os.path.join(list(spec.submodule_search_locations)[0], "datasets", "images")
    out = [os.path.join(d, n) for n in ("china.jpg", "flower.jpg")]
    return out if all(os.path.exists(q) for q in out) else []


def load_sample_photos():
    """-> list of HxWx3 uint8 arrays (R, G, B as the in-tree reader delivers them), [] when unavailable."""
    from . import _binding
    return [_binding.read_image(q) for q in sample_photo_paths()]


def _mosaic_plan(height, width, index, n_photos, ph, pw):
    """Tile choices of image `index`: (photo, flip) per tile of a (ty, tx) grid and the offset (oy, ox) of the crop."""
    rng = np.random.default_rng(9000 + index)
    ty, tx = -(-height // ph) + 1, -(-width // pw) + 1
    tiles = [[(int(rng.integers(n_photos)), int(rng.integers(4))) for _ in range(tx)] for _ in range(ty)]
    return tiles, int(rng.integers(ph)), int(rng.integers(pw))


def _grey_u8(rgb):
    """round((r + g + b) / 3) as uint8: an 8-bit grey plane for the device-resident entry point."""
    return ((rgb.astype(np.uint16).sum(axis=2) * 2 + 3) // 6).astype(np.uint8)


def photo_mosaic(height, width, index=0, photos=None, grey=False):
    """A height x width image tiled from the sample photographs; `index` selects flips and the shift of the tiling.
    grey=False: HxWx3 uint8; grey=True: HxW uint8, the rounded channel mean."""
    if photos is None:
        photos = load_sample_photos()
    if not photos:
        raise RuntimeError("scikit-learn's sample photographs are not installed")
    ph, pw = min(p.shape[0] for p in photos), min(p.shape[1] for p in photos)
    src = [_grey_u8(p[:ph, :pw]) if grey else p[:ph, :pw] for p in photos]
    tiles, oy, ox = _mosaic_plan(height, width, index, len(photos), ph, pw)
    rows = []
    for trow in tiles:
        row = []
        for k, f in trow:
            t = src[k]
            if f & 1:
                t = t[:, ::-1]
            if f & 2:
                t = t[::-1]
            row.append(t)
        rows.append(np.concatenate(row, axis=1))
    big = np.concatenate(rows, axis=0)
    return np.ascontiguousarray(big[oy:oy + height, ox:ox + width])


def photo_mosaic_batch_torch(n, height, width, first_index=0, device="cuda", photos=None):
    """photo_mosaic(grey=True) for indices first_index .. first_index + n - 1, composed on the device: uint8 [n, H, W]."""
    import torch
    if photos is None:
        photos = load_sample_photos()
    if not photos:
        raise RuntimeError("scikit-learn's sample photographs are not installed")
    ph, pw = min(p.shape[0] for p in photos), min(p.shape[1] for p in photos)
    base = [torch.from_numpy(_grey_u8(p[:ph, :pw])).to(device) for p in photos]
    flips = [[b, b.flip(1), b.flip(0), b.flip(0).flip(1)] for b in base]
    out = torch.empty((n, height, width), dtype=torch.uint8, device=device)
    for i in range(n):
        tiles, oy, ox = _mosaic_plan(height, width, first_index + i, len(photos), ph, pw)
        big = torch.cat([torch.cat([flips[k][f] for k, f in trow], dim=1) for trow in tiles], dim=0)
        out[i] = big[oy:oy + height, ox:ox + width]
    return out
