"""Repeatability and matching score of Hessian-Affine regions under a homography
(protocol of Mikolajczyk et al., "A comparison of affine region
detectors", IJCV 2005, as used with the Oxford graf/wall/... sequences).

Regions are the ellipses of a .hesaff.sift file:  (x-u, y-v) M (x-u, y-v)^T = 1,
M = [[a, b], [b, c]]  (README:27-44 of the reference).

  python tools/repeatability.py A.hesaff.sift B.hesaff.sift --H H.txt --size1 W H --size2 W H
  python tools/repeatability.py --synthetic            # needs the GPU: detects on a synthetic
                                                       # image and on homography-warped copies
  python tools/repeatability.py --graf DIR [--devices 0-7]   # a directory in the layout of the Oxford sequences (img1..6.ppm,
                                                       # H1to2p..H1to6p): hesaff --batch over it, then the table (BASELINE config 5)

The Oxford sequences are not available offline; --synthetic substitutes a band-noise image
warped by a viewpoint-like family of homographies (stated in the output).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_sift(path):
    """.hesaff.sift (text) or .hesaff.bin (binary sidecar) -> (regions [n,5] float64: u v a b c, descriptors [n,128] uint8)."""
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == b"HESAFFB1":
        import hesaff_amd
        rows = hesaff_amd.read_bin(path)
        reg = np.stack([rows[k].astype(np.float64) for k in ("x", "y", "a", "b", "c")], 1) if len(rows) else np.zeros((0, 5))
        return reg, np.ascontiguousarray(rows["desc"])
    with open(path, "rb") as f:
        dim = int(f.readline()); n = int(f.readline())
        data = np.loadtxt(f, dtype=np.float64, ndmin=2) if n > 0 else np.zeros((0, 5 + dim))
    assert data.shape == (n, 5 + dim), (data.shape, n, dim)
    return data[:, :5].copy(), data[:, 5:].astype(np.uint8)


def project(H, pts):
    q = np.c_[pts, np.ones(len(pts))] @ H.T
    return q[:, :2] / q[:, 2:3]


def map_regions(H, reg):
    """Ellipses of image 1 -> image 2 through the local affine approximation of H at the centre."""
    u, v = reg[:, 0], reg[:, 1]
    den = H[2, 0] * u + H[2, 1] * v + H[2, 2]
    X = (H[0, 0] * u + H[0, 1] * v + H[0, 2]) / den
    Y = (H[1, 0] * u + H[1, 1] * v + H[1, 2]) / den
    # Jacobian of (X, Y) wrt (u, v)
    J = np.empty((len(reg), 2, 2))
    J[:, 0, 0] = (H[0, 0] - H[2, 0] * X) / den; J[:, 0, 1] = (H[0, 1] - H[2, 1] * X) / den
    J[:, 1, 0] = (H[1, 0] - H[2, 0] * Y) / den; J[:, 1, 1] = (H[1, 1] - H[2, 1] * Y) / den
    M = np.empty((len(reg), 2, 2))
    M[:, 0, 0] = reg[:, 2]; M[:, 0, 1] = M[:, 1, 0] = reg[:, 3]; M[:, 1, 1] = reg[:, 4]
    Ji = np.linalg.inv(J)
    M2 = np.transpose(Ji, (0, 2, 1)) @ M @ Ji
    return np.c_[X, Y, M2[:, 0, 0], M2[:, 0, 1], M2[:, 1, 1]]


def ellipse_area(reg):
    return np.pi / np.sqrt(np.maximum(reg[:, 2] * reg[:, 4] - reg[:, 3] ** 2, 1e-300))


def overlap_error(r1, r2, grid=48):
    """1 - |A n B| / |A u B| for paired ellipses r1[i], r2[i] (same frame), by rasterisation on a
    grid x grid lattice over the joint bounding box.  Both regions are first rescaled so that r1 has
    the area of a circle of radius 30 (the protocol's size normalisation)."""
    s = np.sqrt(ellipse_area(r1) / (np.pi * 30.0 ** 2))          # linear scale factor to remove
    d = (r2[:, :2] - r1[:, :2]) / s[:, None]
    a1, b1, c1 = (r1[:, 2:5] * (s ** 2)[:, None]).T
    a2, b2, c2 = (r2[:, 2:5] * (s ** 2)[:, None]).T

    def half_extent(a, b, c):
        det = a * c - b * b
        return np.sqrt(c / det), np.sqrt(a / det)                 # bounding box half sizes of x^T M x <= 1

    hx1, hy1 = half_extent(a1, b1, c1); hx2, hy2 = half_extent(a2, b2, c2)
    x0 = np.minimum(-hx1, d[:, 0] - hx2); x1 = np.maximum(hx1, d[:, 0] + hx2)
    y0 = np.minimum(-hy1, d[:, 1] - hy2); y1 = np.maximum(hy1, d[:, 1] + hy2)
    t = (np.arange(grid) + 0.5) / grid
    gx = x0[:, None] + (x1 - x0)[:, None] * t[None, :]           # [n, grid]
    gy = y0[:, None] + (y1 - y0)[:, None] * t[None, :]
    X = gx[:, None, :]; Y = gy[:, :, None]
    in1 = a1[:, None, None] * X * X + 2 * b1[:, None, None] * X * Y + c1[:, None, None] * Y * Y <= 1.0
    Xd = X - d[:, 0, None, None]; Yd = Y - d[:, 1, None, None]
    in2 = a2[:, None, None] * Xd * Xd + 2 * b2[:, None, None] * Xd * Yd + c2[:, None, None] * Yd * Yd <= 1.0
    inter = (in1 & in2).sum(axis=(1, 2)).astype(np.float64)
    union = (in1 | in2).sum(axis=(1, 2)).astype(np.float64)
    return 1.0 - inter / np.maximum(union, 1.0)


def evaluate(reg1, desc1, reg2, desc2, H, size1, size2, max_error=0.4, chunk=20000):
    """-> dict with repeatability and matching score (one-to-one correspondences, overlap error < max_error)."""
    w1, h1 = size1; w2, h2 = size2
    Hi = np.linalg.inv(H)
    # regions whose centre lies in the part of the scene visible in both images
    p12 = project(H, reg1[:, :2]); p21 = project(Hi, reg2[:, :2])
    k1 = (p12[:, 0] >= 0) & (p12[:, 0] < w2) & (p12[:, 1] >= 0) & (p12[:, 1] < h2)
    k2 = (p21[:, 0] >= 0) & (p21[:, 0] < w1) & (p21[:, 1] >= 0) & (p21[:, 1] < h1)
    r1 = map_regions(H, reg1[k1]); d1 = desc1[k1]
    r2 = reg2[k2]; d2 = desc2[k2]
    n1, n2 = len(r1), len(r2)
    if min(n1, n2) == 0:
        return {"n1": int(n1), "n2": int(n2), "correspondences": 0, "repeatability": 0.0, "matches": 0, "matching_score": 0.0}
    # candidate pairs: centres closer than the sum of the mean radii (blocks of rows of the distance matrix; pairs come out
    # ordered by (i, j))
    rad1 = np.sqrt(ellipse_area(r1) / np.pi); rad2 = np.sqrt(ellipse_area(r2) / np.pi)
    pi_, pj_ = [], []
    for s0 in range(0, n1, 512):
        x = r1[s0:s0 + 512, 0, None] - r2[None, :, 0]; y = r1[s0:s0 + 512, 1, None] - r2[None, :, 1]
        ii, jj = np.nonzero(x * x + y * y < (rad1[s0:s0 + 512, None] + rad2[None, :]) ** 2)
        pi_.append(ii + s0); pj_.append(jj)
    pi_ = np.concatenate(pi_).astype(np.int64); pj_ = np.concatenate(pj_).astype(np.int64)
    # overlap error >= 1 - min(area) / max(area): pairs that this bound already rejects are not rasterised
    ar1 = ellipse_area(r1)[pi_]; ar2 = ellipse_area(r2)[pj_]
    maybe = np.minimum(ar1, ar2) > (1.0 - max_error) * np.maximum(ar1, ar2)
    pi_, pj_ = pi_[maybe], pj_[maybe]
    err = np.empty(len(pi_))
    for s in range(0, len(pi_), chunk):
        err[s:s + chunk] = overlap_error(r1[pi_[s:s + chunk]], r2[pj_[s:s + chunk]])
    good = err < max_error
    # one-to-one: greedy by increasing overlap error
    order = np.argsort(err[good], kind="stable")
    gi, gj, ge = pi_[good][order], pj_[good][order], err[good][order]
    used1 = np.zeros(n1, bool); used2 = np.zeros(n2, bool)
    pairs = []
    for a, b in zip(gi, gj):
        if not used1[a] and not used2[b]:
            used1[a] = used2[b] = True
            pairs.append((a, b))
    ncorr = len(pairs)
    # matching score: a correspondence counts when the region of image 2 is also the nearest
    # neighbour of its partner in descriptor space
    matches = 0
    if ncorr:
        # squared distances are integers below 2^24 (128 x 255^2): exact in float64 products
        D2 = d2.astype(np.float64); n2sq = (D2 * D2).sum(axis=1)
        pa = np.asarray([a for a, _ in pairs]); pb = np.asarray([b for _, b in pairs])
        for s0 in range(0, ncorr, 1024):
            D1 = d1[pa[s0:s0 + 1024]].astype(np.float64)
            dist = n2sq[None, :] - 2.0 * (D1 @ D2.T)            # + |d1|^2, constant per row
            matches += int((np.argmin(dist, axis=1) == pb[s0:s0 + 1024]).sum())
    return {"n1": int(n1), "n2": int(n2), "correspondences": int(ncorr), "repeatability": ncorr / min(n1, n2),
            "matches": int(matches), "matching_score": matches / min(n1, n2), "max_overlap_error": max_error}


def viewpoint_homography(w, h, angle_deg, zoom=1.0):
    """Homography of a plane seen after rotating the camera about the vertical axis by angle_deg
    (focal length = image width), centred on the image: the graf-like viewpoint change."""
    f = float(w)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    t = np.deg2rad(angle_deg)
    R = np.array([[np.cos(t), 0, np.sin(t)], [0, 1, 0], [-np.sin(t), 0, np.cos(t)]])
    n = np.array([0, 0, 1.0]); d = 1.0
    tr = np.array([-np.sin(t) * zoom, 0, (1 - np.cos(t)) * zoom + (zoom - 1.0)])
    Hn = R + np.outer(tr, n) / d
    H = K @ Hn @ np.linalg.inv(K)
    return H / H[2, 2]


def warp_image(img, H, out_size):
    """img seen through H (image1 -> image2 coordinates), bilinear, outside = mid grey."""
    from scipy.ndimage import map_coordinates
    w2, h2 = out_size
    yy, xx = np.mgrid[0:h2, 0:w2].astype(np.float64)
    src = project(np.linalg.inv(H), np.c_[xx.ravel(), yy.ravel()])
    out = map_coordinates(img.astype(np.float32), [src[:, 1], src[:, 0]], order=1, mode="constant", cval=127.0)
    return np.clip(np.rint(out.reshape(h2, w2)), 0, 255).astype(np.uint8)


def synthetic_sequence(width=800, height=640, angles=(10, 20, 30, 40, 50), seed=1234):
    import hesaff_amd
    from hesaff_amd.synth import band_noise_image
    base = band_noise_image(height, width, seed)
    imgs = [base]; Hs = [np.eye(3)]
    for a in angles:
        H = viewpoint_homography(width, height, a)
        imgs.append(warp_image(base, H, (width, height))); Hs.append(H)
    with hesaff_amd.HesaffContext(device=0) as ctx:
        res = ctx.detect_batch(imgs)
        mr = ctx.params.mrSize
    regs = []
    for _, keys in res:
        e = hesaff_amd.ellipse(keys, mr).astype(np.float64)
        regs.append((np.c_[keys["x"].astype(np.float64), keys["y"].astype(np.float64), e], np.ascontiguousarray(keys["desc"])))
    out = []
    for a, H, (r, d) in zip(angles, Hs[1:], regs[1:]):
        ev = evaluate(regs[0][0], regs[0][1], r, d, H, (width, height), (width, height))
        ev["viewpoint_deg"] = a
        out.append(ev)
    return {"data": "synthetic: band-noise %dx%d image and copies warped by a camera rotation about the vertical axis "
                    "(the Oxford sequences are not available offline)" % (width, height), "pairs": out}


def write_sequence_files(out_dir, width=800, height=640, angles=(10, 20, 30, 40, 50), seed=1234, quality=92, photo=None):
    """A graf-like sequence on disk, laid out like the Oxford sets: img1.jpg .. imgN.jpg (colour JPEG, 4:2:0, so that the
    library's own JPEG decoder is on the path exactly as with the real data) and H1to2p .. H1toNp (3x3 text).
    photo: an HxWx3 uint8 photograph to warp instead of the band-noise image (width / height are then its size).
    -> (image paths, homographies)."""
    from PIL import Image
    from hesaff_amd.synth import band_noise_image
    if photo is not None:
        base = np.ascontiguousarray(photo, np.uint8)
        height, width = base.shape[:2]
    else:
        g = band_noise_image(height, width, seed).astype(np.float32)
        t1 = band_noise_image(height, width, seed + 1).astype(np.float32)
        t2 = band_noise_image(height, width, seed + 2).astype(np.float32)
        base = np.stack([g, 0.75 * g + 0.25 * t1, 0.75 * g + 0.25 * t2], axis=2)
        base = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    os.makedirs(out_dir, exist_ok=True)
    paths, Hs = [], []
    for k, a in enumerate((0,) + tuple(angles)):
        H = np.eye(3) if k == 0 else viewpoint_homography(width, height, a)
        img = base if k == 0 else np.stack([warp_image(base[:, :, c], H, (width, height)) for c in range(3)], axis=2)
        q = os.path.join(out_dir, "img%d.jpg" % (k + 1))
        Image.fromarray(img, "RGB").save(q, "JPEG", quality=quality, subsampling=2)
        paths.append(q); Hs.append(H)
        if k > 0:
            np.savetxt(os.path.join(out_dir, "H1to%dp" % (k + 1)), H)
    return paths, Hs


def evaluate_sequence_files(paths, Hs, size, angles=None):
    """Repeatability / matching score of image 1 against every other image from their .hesaff.sift files."""
    r1, d1 = read_sift(paths[0] + ".hesaff.sift")
    out = []
    for k in range(1, len(paths)):
        r2, d2 = read_sift(paths[k] + ".hesaff.sift")
        ev = evaluate(r1, d1, r2, d2, Hs[k], size, size)
        ev["pair"] = "img1 -> img%d" % (k + 1)
        if angles is not None:
            ev["viewpoint_deg"] = angles[k - 1]
        out.append(ev)
    return out


def sequence_through_cli(out_dir, width=800, height=640, angles=(10, 20, 30, 40, 50), fast=0, photo=None):
    """BASELINE.json config 5 on the synthetic stand-in: the sequence as JPEG files -> `hesaff --batch` (decode threads,
    device, writer threads) -> the evaluation above.  photo: index of one of scikit-learn's sample photographs (0: china.jpg,
    1: flower.jpg) to warp instead of the band-noise image: a planar scene under a viewpoint change, like graf."""
    import subprocess
    pic = None
    if photo is not None:
        from hesaff_amd.synth import load_sample_photos
        pics = load_sample_photos()
        if not pics:
            raise RuntimeError("scikit-learn's sample photographs are not installed")
        pic = pics[photo]
        height, width = pic.shape[:2]
    paths, Hs = write_sequence_files(out_dir, width, height, angles, photo=pic)
    lst = os.path.join(out_dir, "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(paths) + "\n")
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    r = subprocess.run([exe, "--batch", lst] + (["--fast", str(fast)] if fast else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hesaff --batch failed: " + r.stderr[-2000:])
    return {"data": "graf-like sequence: %s (%dx%d) and %d copies warped by a camera rotation about the "
                    "vertical axis, stored as JPEG (quality 92, 4:2:0) and read back by the library's JPEG decoder; the Oxford "
                    "sequences are not available offline" % ("a colour band-noise image" if photo is None else
                                                              "the photograph %s of scikit-learn's sample images" % ("china.jpg", "flower.jpg")[photo],
                                                              width, height, len(angles)),
            "command": "python tools/repeatability.py --synthetic-files DIR%s%s   (hesaff --batch list.txt, then the evaluation)"
                       % ((" --fast %d" % fast) if fast else "", (" --photo %d" % photo) if photo is not None else ""),
            "fast": fast,
            "cli_stdout_tail": r.stdout.strip().splitlines()[-1],
            "pairs": evaluate_sequence_files(paths, Hs, (width, height), list(angles))}


def image_size(path):
    """(width, height) of an image file through the library's own reader (the cv::imread of hesaff.cpp:137)."""
    import hesaff_amd
    img = hesaff_amd.read_image(path)
    return int(img.shape[1]), int(img.shape[0])


def graf_directory(d, devices=None, fast=0, table=True):
    """BASELINE.json config 5 in ONE command for a directory in the layout of the Oxford affine-covariant-regions sequences
    (graf, wall, boat, ...; the format README:27-44 of the reference is written for): img1..imgN.{ppm,pgm,png,jpg} and the
    homographies H1to2p..H1toNp (3 x 3, text).  Every image goes through `hesaff --batch` (one list, sharded over `devices`
    like any other list), then image 1 is evaluated against every other image under its homography."""
    import re
    import subprocess
    imgs = {}
    for f in sorted(os.listdir(d)):
        m = re.fullmatch(r"img(\d+)\.(ppm|pgm|pbm|pnm|png|jpg|jpeg)", f, re.IGNORECASE)
        if m and int(m.group(1)) not in imgs:
            imgs[int(m.group(1))] = os.path.join(d, f)
    if 1 not in imgs or len(imgs) < 2:
        raise RuntimeError("%s: no img1.* plus at least one more imgK.* (ppm, pgm, png or jpg)" % d)
    ks = [k for k in sorted(imgs) if k != 1]
    Hs = {}
    for k in ks:
        for name in ("H1to%dp" % k, "H1to%d" % k, "H1to%dp.txt" % k):
            q = os.path.join(d, name)
            if os.path.exists(q):
                Hs[k] = np.loadtxt(q).reshape(3, 3)
                break
        else:
            raise RuntimeError("%s: no homography H1to%dp for %s" % (d, k, os.path.basename(imgs[k])))
    paths = [imgs[1]] + [imgs[k] for k in ks]
    lst = os.path.join(d, "hesaff_list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(paths) + "\n")
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    cmd = [exe, "--batch", lst] + (["--devices", devices] if devices else []) + (["--fast", str(fast)] if fast else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hesaff --batch failed: " + (r.stderr or r.stdout)[-2000:])
    size1 = image_size(imgs[1])
    r1, d1 = read_sift(imgs[1] + ".hesaff.sift")
    pairs = []
    for k in ks:
        r2, d2 = read_sift(imgs[k] + ".hesaff.sift")
        ev = evaluate(r1, d1, r2, d2, Hs[k], size1, image_size(imgs[k]))
        ev["pair"] = "img1 -> img%d" % k
        pairs.append(ev)
    out = {"data": "%s: %d images in the layout of the Oxford affine-covariant-regions sequences (img1..N, H1toKp)" % (os.path.abspath(d), len(paths)),
           "command": " ".join(cmd), "fast": fast, "cli_stdout_tail": r.stdout.strip().splitlines()[-1], "pairs": pairs}
    if table:
        keys = [k for k in ("n1", "n2", "correspondences", "repeatability", "matches", "matching_score") if pairs and k in pairs[0]]
        print("%-14s" % "pair" + "".join("%16s" % k for k in keys), file=sys.stderr)
        for ev in pairs:
            print("%-14s" % ev["pair"] + "".join(("%16.4f" % ev[k]) if isinstance(ev[k], float) else ("%16d" % ev[k]) for k in keys), file=sys.stderr)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("files", nargs="*")
    ap.add_argument("--H", help="text file with the 3x3 homography image1 -> image2")
    ap.add_argument("--size1", nargs=2, type=int, metavar=("W", "H"))
    ap.add_argument("--size2", nargs=2, type=int, metavar=("W", "H"))
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--synthetic-files", metavar="DIR", help="write the synthetic sequence as JPEG files into DIR, run `hesaff --batch` on it, evaluate")
    ap.add_argument("--fast", type=int, default=0, help="with --synthetic-files: hesaff_params.fast (0 = parity mode through the CLI)")
    ap.add_argument("--photo", type=int, default=None, help="with --synthetic-files: warp this sample photograph (0: china.jpg, 1: flower.jpg) instead of band noise")
    ap.add_argument("--graf", metavar="DIR", help="a directory in the layout of the Oxford sequences (img1..N.ppm|pgm|png|jpg, H1to2p..H1toNp): "
                                                  "`hesaff --batch` over its images, then image 1 against every other image; table on stderr, JSON on stdout")
    ap.add_argument("--devices", help="with --graf: passed to `hesaff --batch --devices` (0-7, 0,2, all)")
    args = ap.parse_args()
    if args.graf:
        print(json.dumps(graf_directory(args.graf, devices=args.devices, fast=args.fast), indent=1))
        return
    if args.synthetic_files:
        print(json.dumps(sequence_through_cli(args.synthetic_files, fast=args.fast, photo=args.photo), indent=1))
        return
    if args.synthetic:
        print(json.dumps(synthetic_sequence()))
        return
    if len(args.files) != 2 or not args.H or not args.size1 or not args.size2:
        ap.error("two .hesaff.sift files, --H, --size1 and --size2 are required (or --synthetic)")
    r1, d1 = read_sift(args.files[0]); r2, d2 = read_sift(args.files[1])
    H = np.loadtxt(args.H).reshape(3, 3)
    print(json.dumps(evaluate(r1, d1, r2, d2, H, tuple(args.size1), tuple(args.size2))))


if __name__ == "__main__":
    main()
