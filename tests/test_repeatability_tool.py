"""tools/repeatability.py: the geometry of the evaluation on regions whose
behaviour under the homography is known in closed form."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import repeatability as rp   # noqa: E402


def _random_regions(n, w, h, rng):
    u = rng.uniform(40, w - 40, n); v = rng.uniform(40, h - 40, n)
    r1 = rng.uniform(6, 25, n); r2 = r1 * rng.uniform(0.5, 1.0, n); th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    a = c * c / r1 ** 2 + s * s / r2 ** 2; b = c * s * (1 / r1 ** 2 - 1 / r2 ** 2); cc = s * s / r1 ** 2 + c * c / r2 ** 2
    return np.c_[u, v, a, b, cc]


def test_overlap_error_of_circles_matches_closed_form():
    r = 30.0
    for d in (0.0, 10.0, 30.0, 45.0, 70.0):
        e1 = np.array([[100.0, 100.0, 1 / r ** 2, 0.0, 1 / r ** 2]]); e2 = e1.copy(); e2[0, 0] += d
        if d >= 2 * r:
            want = 1.0
        else:
            inter = 2 * r * r * np.arccos(d / (2 * r)) - d / 2 * np.sqrt(4 * r * r - d * d)
            want = 1 - inter / (2 * np.pi * r * r - inter)
        got = float(rp.overlap_error(e1, e2, grid=200)[0])
        assert abs(got - want) < 0.01, (d, got, want)


def test_regions_mapped_by_the_same_affinity_are_all_repeated():
    rng = np.random.default_rng(3)
    w, h = 800, 640
    reg1 = _random_regions(400, w, h, rng)
    A = np.array([[0.9, -0.25, 60.0], [0.2, 1.1, -30.0], [0.0, 0.0, 1.0]])
    reg2 = rp.map_regions(A, reg1)                       # exact for an affinity
    desc = rng.integers(0, 256, (400, 128), dtype=np.uint8)
    perm = rng.permutation(400)
    ev = rp.evaluate(reg1, desc, reg2[perm], desc[perm], A, (w, h), (w, h))
    assert ev["correspondences"] == min(ev["n1"], ev["n2"]) and ev["repeatability"] == 1.0
    assert ev["matches"] == ev["correspondences"]      # identical descriptors: every correspondence is also the NN
    # unrelated regions: (almost) nothing repeats
    other = _random_regions(400, w, h, np.random.default_rng(99))
    ev2 = rp.evaluate(reg1, desc, other, desc, A, (w, h), (w, h))
    assert ev2["repeatability"] < 0.1


def test_sift_file_round_trip(tmp_path):
    import hesaff_amd
    rng = np.random.default_rng(5)
    n = 50
    keys = np.zeros(n, hesaff_amd.KEYPOINT_DTYPE)
    keys["x"] = rng.uniform(0, 640, n); keys["y"] = rng.uniform(0, 480, n); keys["s"] = rng.uniform(2, 9, n)
    keys["a11"] = rng.uniform(0.6, 1.6, n); keys["a21"] = rng.uniform(-0.5, 0.5, n); keys["a22"] = 1.0 / keys["a11"]
    keys["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    mr = hesaff_amd.default_params().mrSize
    p = tmp_path / "a.hesaff.sift"
    hesaff_amd.write_sift(str(p), keys, mr)
    reg, desc = rp.read_sift(str(p))
    assert np.array_equal(desc, keys["desc"])
    e = hesaff_amd.ellipse(keys, mr)
    assert np.allclose(reg[:, 2:], e, rtol=2e-5) and np.allclose(reg[:, 0], keys["x"], rtol=2e-5)
