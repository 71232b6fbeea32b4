#!/usr/bin/env python3
"""Fast mode (hesaff_params.fast = 2) against parity mode on the bench's image family or on photographs.

Both modes run the same batch through hesaff_detect_batch_device; the report gives, per SURVEY App. C.5's statistics:
keypoint counts, keypoints of the parity run that the fast run reproduces within 0.01 px (nearest neighbour in (x, y)
with equal scale to 1e-3 relative), the share of those with an identical 128-byte descriptor, the distribution of
|delta desc|, the relative difference of the shape matrices, and the speed-up of the whole step.

    python tools/fast_mode_report.py [--batch 16] [--width 3840 --height 2160] [--out profiles/r02_fast_mode.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def device_keys(torch, hesaff_amd, dkeys, total):
    buf = torch.empty(max(total, 1) * 164, dtype=torch.uint8, device="cuda")
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    if total:
        assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(dkeys), ctypes.c_size_t(total * 164), 3) == 0
    return np.frombuffer(buf.cpu().numpy().tobytes()[: total * 164], dtype=hesaff_amd.KEYPOINT_DTYPE).copy()


def compare(par, fast):
    """par, fast: KEYPOINT_DTYPE arrays of one image."""
    from scipy.spatial import cKDTree
    out = {"n_parity": int(len(par)), "n_fast": int(len(fast))}
    if len(par) == 0 or len(fast) == 0:
        return out
    # position AND scale: several keypoints may share a position across levels
    pf = np.stack([fast["x"], fast["y"], np.log(fast["s"]) * 10.0], 1)
    pp = np.stack([par["x"], par["y"], np.log(par["s"]) * 10.0], 1)
    d, j = cKDTree(pf).query(pp, k=1)
    ok = d < 0.01
    out["matched_within_0.01px"] = int(ok.sum())
    a, b = par[ok], fast[j[ok]]
    dd = np.abs(a["desc"].astype(np.int16) - b["desc"].astype(np.int16))
    same = (dd.max(axis=1) == 0)
    out["identical_descriptors"] = int(same.sum())
    out["max_abs_desc_delta"] = int(dd.max()) if dd.size else 0
    out["mean_abs_desc_delta"] = float(dd.mean()) if dd.size else 0.0
    out["rows_with_delta_le_1"] = int((dd.max(axis=1) <= 1).sum())
    out["rows_with_delta_le_4"] = int((dd.max(axis=1) <= 4).sum())
    nrm = np.linalg.norm(a["desc"].astype(np.float64) - b["desc"].astype(np.float64), axis=1)
    out["desc_l2_distance_p50_p99_max"] = [float(np.percentile(nrm, 50)), float(np.percentile(nrm, 99)), float(nrm.max())]
    na = np.linalg.norm(a["desc"].astype(np.float64), axis=1); nb = np.linalg.norm(b["desc"].astype(np.float64), axis=1)
    cos = (a["desc"].astype(np.float64) * b["desc"].astype(np.float64)).sum(axis=1) / np.maximum(na * nb, 1e-9)
    out["desc_cosine_p01_p10_p50"] = [float(np.percentile(cos, 1)), float(np.percentile(cos, 10)), float(np.percentile(cos, 50))]
    A = np.stack([a["a11"], a["a21"], a["a22"]], 1).astype(np.float64)
    Bm = np.stack([b["a11"], b["a21"], b["a22"]], 1).astype(np.float64)
    rel = np.abs(A - Bm).max(axis=1) / np.maximum(np.abs(A).max(axis=1), 1e-12)
    out["shape_rel_delta_p50_p99_max"] = [float(np.percentile(rel, 50)), float(np.percentile(rel, 99)), float(rel.max())]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--level", type=int, default=2, help="hesaff_params.fast of the fast run (2: pyramid-sampled large windows)")
    ap.add_argument("--photo", action="store_true", help="mosaics of scikit-learn's sample photographs instead of band noise")
    a = ap.parse_args()
    import torch
    import hesaff_amd
    from hesaff_amd.synth import band_noise_batch_torch, photo_mosaic_batch_torch
    if a.photo:
        imgs = photo_mosaic_batch_torch(a.batch, a.height, a.width, first_index=0, device="cuda")
    else:
        imgs = band_noise_batch_torch(a.batch, a.height, a.width, seed=1234, device="cuda")
    torch.cuda.synchronize()
    res = {}
    for mode in (0, 1):
        p = hesaff_amd.default_params()
        p.max_batch = a.batch
        p.fast = a.level if mode else 0
        with hesaff_amd.HesaffContext(p, device=0) as ctx:
            ctx.detect_batch_device(imgs.data_ptr(), a.batch, a.width, a.height)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                ch, cd, dk, total = ctx.detect_batch_device(imgs.data_ptr(), a.batch, a.width, a.height)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            res[mode] = (ch.copy(), cd.copy(), device_keys(torch, hesaff_amd, dk, total), dt)
    (ch0, cd0, k0, t0), (ch1, cd1, k1, t1) = res[0], res[1]
    s0 = np.concatenate([[0], np.cumsum(cd0)]); s1 = np.concatenate([[0], np.cumsum(cd1)])
    tot = {}
    for b in range(a.batch):
        c = compare(k0[s0[b]:s0[b + 1]], k1[s1[b]:s1[b + 1]])
        for k, v in c.items():
            if isinstance(v, list):
                tot.setdefault(k, []).append(v)
            elif k.startswith("max_"):
                tot[k] = max(tot.get(k, 0), v)
            elif k.startswith("mean_"):
                tot.setdefault(k, []).append(v)
            else:
                tot[k] = tot.get(k, 0) + v
    for k in list(tot):
        if isinstance(tot[k], list):
            arr = np.array(tot[k], dtype=np.float64)
            tot[k] = arr.mean(axis=0).tolist() if arr.ndim == 2 else float(arr.mean())
    m = max(tot.get("matched_within_0.01px", 0), 1)
    report = {
        "workload": "%d x %dx%d %s, default parameters, hesaff_detect_batch_device"
                    % (a.batch, a.width, a.height, "mosaics of two photographs" if a.photo else "band-noise images"),
        "fast_level": a.level,
        "hessian_keypoints_equal": bool(np.array_equal(ch0, ch1)),
        "parity_ms_per_step": t0 * 1e3, "fast_ms_per_step": t1 * 1e3, "speed_up": t0 / t1,
        "descriptors_parity": int(cd0.sum()), "descriptors_fast": int(cd1.sum()),
        "matched_pct_of_parity": 100.0 * tot.get("matched_within_0.01px", 0) / max(int(cd0.sum()), 1),
        "identical_descriptor_pct_of_matched": 100.0 * tot.get("identical_descriptors", 0) / m,
        "desc_delta_le_1_pct_of_matched": 100.0 * tot.get("rows_with_delta_le_1", 0) / m,
        "desc_delta_le_4_pct_of_matched": 100.0 * tot.get("rows_with_delta_le_4", 0) / m,
        "totals": tot,
        "what_differs": "fast = 2: the windows larger than the 41 x 41 patch are sampled from the scale-space level with the matching blur "
                        "instead of being warped and blurred (another algorithm for those keypoints); everything else - pyramid, extrema, "
                        "localisation, ordering, affine shapes, window geometry, the small windows and the descriptor arithmetic - runs on "
                        "the parity kernels",
    }
    print(json.dumps(report))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
