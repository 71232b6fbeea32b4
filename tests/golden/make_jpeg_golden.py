"""JPEG fixtures: files encoded by Pillow's libjpeg-turbo and the pixels the same library decodes them to.

    python tests/golden/make_jpeg_golden.py        (needs Pillow; run in the development container)

The product's reader (hesaff_amd/csrc/jpeg_decode.cpp) must reproduce the .pgm / .ppm bytes from the .jpg files
(tests/test_host_side.py::test_read_jpeg_golden_pixels runs without Pillow).  cv::imread (hesaff.cpp:137) decodes with
libjpeg at its defaults (JDCT_ISLOW, fancy up-sampling), which is what Pillow's decoder uses too."""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def synth(h, w, color, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 7.0 + yy / 11.0), 127 + 100 * np.cos(xx / 5.0 - yy / 13.0),
                     127 + 90 * np.sin(xx / 3.0) * np.cos(yy / 4.0)], -1) + rng.normal(0, 25, (h, w, 3))
    a = np.clip(base, 0, 255).astype(np.uint8)
    return a if color else a[..., 0]


def write_pnm(path, a):
    with open(path, "wb") as f:
        if a.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        else:
            f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a).tobytes())


FIXTURES = [
    # name, (h, w), colour, Pillow save options
    ("jpeg_gray_q90", (61, 83), False, dict(quality=90)),
    ("jpeg_420_q85", (120, 211), True, dict(quality=85, subsampling=2)),
    ("jpeg_422_q70_rst", (64, 97), True, dict(quality=70, subsampling=1, restart_marker_blocks=3)),
    ("jpeg_prog_420_q80", (75, 130), True, dict(quality=80, subsampling=2, progressive=True)),
    ("jpeg_prog_gray_q60", (33, 49), False, dict(quality=60, progressive=True, optimize=True)),
]

if __name__ == "__main__":
    for k, (name, (h, w), color, opts) in enumerate(FIXTURES):
        buf = io.BytesIO()
        Image.fromarray(synth(h, w, color, 100 + k)).save(buf, "JPEG", **opts)
        jpg = os.path.join(HERE, name + ".jpg")
        open(jpg, "wb").write(buf.getvalue())
        ref = np.asarray(Image.open(jpg))
        write_pnm(os.path.join(HERE, name + (".ppm" if color else ".pgm")), ref)
        print(name, len(buf.getvalue()), "bytes", ref.shape)
