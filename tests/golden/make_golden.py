#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Run from the repo root:  python tests/golden/make_golden.py

WHAT THESE FIXTURES ARE: regression vectors produced by the repo's own CPU oracle
(oracle/hesaff_oracle.cpp) on small deterministic images.  They are NOT outputs of the
reference binary: perdoch/hesaff cannot be built in this image (all sources include
OpenCV's <cv.h>, which is neither installed nor vendored) and it ships no golden vectors
of its own.  The fixtures pin the oracle against accidental change and give the GPU tests
byte-exact files to reproduce.

The one fixture that ties the oracle to the COMPILED reference is probe_vga.pgm: the 640x480
input of SURVEY.md Appendix C.3 (md5 74f828b5...), for which the survey recorded the md5 of the
reference's own output file (e004ba88..., App. C.4) and its call counts (App. C.6).  The oracle
reproduces both (tests/test_oracle.py::test_survey_probe_*, scripts/check_survey_probe.py).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hesaff_amd.synth import band_noise_image  # noqa: E402
from tests import _oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = ((1.5, 40.0), (3.0, 40.0), (6.0, 50.0))
CASES = [("band_131x77", 77, 131, 7), ("band_96x96", 96, 96, 8), ("band_160x120", 120, 160, 9), ("tiny_20x15", 15, 20, 1),
         ("thin_12x40", 40, 12, 2)]


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def main():
    manifest = {}
    for name, h, w, seed in CASES:
        img = band_noise_image(h, w, seed, SMALL)
        write_pgm(os.path.join(HERE, name + ".pgm"), img)
        o = _oracle.OracleRun(_oracle.gray_from_u8(img))
        text = o.export_text()
        with open(os.path.join(HERE, name + ".hesaff.sift"), "wb") as f:
            f.write(text)
        hf, hi = o.hessian()
        U, ci = o.affine()
        np.savez_compressed(os.path.join(HERE, name + "_stages.npz"), hess_f=hf, hess_i=hi, aff_U=U, aff_i=ci,
                            key_src=o.key_sources())
        manifest[name] = {"width": w, "height": h, "seed": seed, "candidates": int(o.n_candidates),
                          "hessian": int(o.n_hessian), "descriptors": int(o.n_keys),
                          "pgm_md5": hashlib.md5(open(os.path.join(HERE, name + ".pgm"), "rb").read()).hexdigest(),
                          "sift_md5": hashlib.md5(text).hexdigest()}
    # bigger images: counts + hash only
    for name, h, w, seed in [("band_640x480", 480, 640, 1234)]:
        img = band_noise_image(h, w, seed)
        o = _oracle.OracleRun(_oracle.gray_from_u8(img))
        manifest[name] = {"width": w, "height": h, "seed": seed, "candidates": int(o.n_candidates), "hessian": int(o.n_hessian),
                          "descriptors": int(o.n_keys), "image_md5": hashlib.md5(img.tobytes()).hexdigest(),
                          "sift_md5": hashlib.md5(o.export_text()).hexdigest()}
    # SURVEY.md App. C.3 input (640x480, seed 1234): committed as data, regenerated and compared by the tests
    from scripts.check_survey_probe import PROBE, probe_image, pgm_bytes
    h, w, seed, bands = PROBE["vga"][:4]
    img = probe_image(h, w, seed, bands)
    with open(os.path.join(HERE, "probe_vga.pgm"), "wb") as f:
        f.write(pgm_bytes(img))
    o = _oracle.OracleRun(_oracle.gray_from_u8(img))
    manifest["probe_vga"] = {"width": w, "height": h, "seed": seed, "candidates": int(o.n_candidates), "hessian": int(o.n_hessian),
                             "descriptors": int(o.n_keys), "pgm_md5": hashlib.md5(pgm_bytes(img)).hexdigest(),
                             "sift_md5": hashlib.md5(o.export_text()).hexdigest(),
                             "survey_recorded": {"pgm_md5_prefix": "74f828b5", "sift_md5_prefix": "e004ba88",
                                                 "candidates": 5281, "hessian": 4763, "descriptors": 4183}}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
