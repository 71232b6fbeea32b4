import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import hesaff_amd
from tests import _oracle
from tests.test_gpu_parity import _structured_image
p = hesaff_amd.default_params(); p.max_kpts_per_mpx = 400000
ctx = hesaff_amd.HesaffContext(p, device=0)
kinds = ["checker", "blobs", "lines", "saturated", "ramp"]
bad = 0; total = 0; nimg = 0
for seed in range(12):
    rng = np.random.default_rng(1000 + seed)
    imgs = []
    for i in range(40):
        h, w = int(rng.integers(13, 400)), int(rng.integers(13, 500))
        imgs.append(_structured_image(kinds[(i + seed) % 5], h, w, rng))
    res = ctx.detect_batch(imgs)
    for i, (img, (nh, keys)) in enumerate(zip(imgs, res)):
        o = _oracle.OracleRun(_oracle.gray_from_u8(img)); g, t, d = o.keys(); nimg += 1
        ok = nh == o.n_hessian and len(keys) == o.n_keys
        if ok and len(keys):
            ok = np.array_equal(keys["desc"], d) and all(np.array_equal(np.ascontiguousarray(keys[n]).view(np.uint32), np.ascontiguousarray(g[:, j], np.float32).view(np.uint32)) or np.allclose(keys[n], g[:, j], rtol=0, atol=0) for j, n in enumerate(["x","y","s","a11","a12","a21","a22","response"]))
        if not ok:
            bad += 1; print("MISMATCH seed", seed, "img", i, img.shape, kinds[(i+seed)%5], nh, o.n_hessian, len(keys), o.n_keys)
        total += len(keys)
print("images", nimg, "keypoints", total, "mismatching images", bad)
