"""GPU parity tests proper: libhesaff_amd (HIP, through the C ABI) against the CPU oracle on
the same seeded inputs.  Bar: bit-exact for every float plane, keypoint field and
descriptor byte (the product path and the oracle evaluate the same IEEE expression
trees); (a,b,c) of the exported ellipse within 1e-4 relative (north_star tolerance).
"""
import numpy as np
import pytest

from hesaff_amd.synth import band_noise_image

pytestmark = pytest.mark.gpu

SMALL_BANDS = ((1.5, 40.0), (3.0, 40.0), (6.0, 50.0))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ne = _bits(a) != _bits(b)
    # +0 / -0 are the same value for every later operation
    ne &= ~((a == 0) & (b == 0))
    if ne.any():
        idx = np.argwhere(ne)[:5]
        raise AssertionError("%s: %d of %d floats differ, first at %s: %s vs %s" % (
            what, int(ne.sum()), a.size, idx.tolist(), a[tuple(idx[0])], b[tuple(idx[0])]))


def test_device_math_matches_libm(ctx, oracle):
    rng = np.random.default_rng(5)
    n = 1 << 20
    y = (rng.standard_normal(n) * 50).astype(np.float32)
    x = (rng.standard_normal(n) * 50).astype(np.float32)
    y[:1000] = 0; x[1000:2000] = 0; x[2000:3000] = 1.0
    at, _ = ctx.math(y, x)
    L = oracle.lib()
    ref = np.array([L.ho_atan2f(float(a), float(b)) for a, b in zip(y[:200000], x[:200000])], np.float32)
    assert_bit_equal(at[:200000], ref, "atan2f")
    e = rng.uniform(-0.5, 0.5, 200000).astype(np.float32)
    _, pw = ctx.math(e, e)
    refp = np.array([L.ho_pow2f(float(a)) for a in e], np.float32)
    assert_bit_equal(pw, refp, "powf(2,.)")


def test_range_free_gradient_forms_equal_the_general_ones(ctx):
    """k_sift_grad evaluates sqrt and atan2f without the range handling of the compiler's division / square root when
    the patch was photometrically normalised: pixels are then multiples of 2^-17 in [0, 255] and their differences zero
    or normal.  On that domain the short forms must return the general forms' bits (the general atan2f is pinned to libm
    in test_device_math_matches_libm)."""
    rng = np.random.default_rng(11)
    n = 1 << 24
    q = np.float32(2.0 ** -17)

    def pix(m):   # pixel values as the kernel produces them: multiples of 2^-17 in [0, 255], many clamped or nearly equal
        kind = rng.integers(0, 4, m)
        v = rng.integers(0, 255 * 2 ** 17 + 1, m).astype(np.float64) * 2.0 ** -17
        v = np.where(kind == 0, np.round(v), v)                       # integers
        v = np.where(kind == 1, np.minimum(v, 2.0 ** -17 * rng.integers(0, 64, m)), v)   # near the lower clamp
        return v.astype(np.float32)
    a, b, c2, d = pix(n), pix(n), pix(n), pix(n)
    near = rng.random(n) < 0.3
    b = np.where(near, a + q * rng.integers(-3, 4, n).astype(np.float32), b)     # gradients of a few quanta
    b = np.clip(b, 0, 255).astype(np.float32)
    gx = (b - a).astype(np.float32); gy = (d - c2).astype(np.float32)
    gx[:4096] = 0; gy[2048:6144] = 0                                              # zero operands, both zero
    gx[6144:8192] = q; gy[8192:10240] = np.float32(255.0)                         # extreme quotients
    nz = np.concatenate([gx[gx != 0], gy[gy != 0]])
    assert np.all(np.abs(nz) >= q)                                                # the domain claimed in the kernel
    og, on, gg, gn = ctx.math_sift(gy, gx)
    assert_bit_equal(on, og, "atan2f without range handling")
    assert_bit_equal(gn, gg, "sqrt without range handling")


@pytest.mark.parametrize("shape", [(61, 83), (128, 200), (7, 9), (33, 600)])
@pytest.mark.parametrize("sigma", [0.62, 0.7, 0.9, 1.2262737, 1.5198685, 2.4525473, 4.3])
def test_gaussian_blur(ctx, oracle, shape, sigma):
    rng = np.random.default_rng(11)
    img = (rng.random(shape) * 255).astype(np.float32)
    ref = np.empty_like(img)
    oracle.lib().ho_gaussian_blur(img, shape[0], shape[1], sigma, ref)
    assert_bit_equal(ctx.gaussian_blur(img, sigma), ref, "blur sigma=%g" % sigma)


def test_hessian_and_half(ctx, oracle):
    rng = np.random.default_rng(12)
    for shape in ((97, 131), (40, 700), (3, 3)):
        im = (rng.random(shape) * 255).astype(np.float32)
        r = np.empty_like(im)
        oracle.lib().ho_hessian_response(im, shape[0], shape[1], 2.56, r)
        # the frame of the reference's response plane is uninitialised memory (pyramid.cpp:70), the library writes 0
        assert_bit_equal(ctx.hessian_response(im, 2.56)[1:-1, 1:-1], r[1:-1, 1:-1], "hessian %s" % (shape,))
    img = (rng.random((97, 131)) * 255).astype(np.float32)
    ref = np.empty_like(img)
    oracle.lib().ho_hessian_response(img, 97, 131, 2.56, ref)
    assert_bit_equal(ctx.hessian_response(img, 2.56), ref, "hessian")
    refh = np.empty((48, 65), np.float32)
    oracle.lib().ho_half_image(img, 97, 131, refh)
    assert_bit_equal(ctx.half_image(img), refh, "half")


@pytest.mark.parametrize("hw,seed", [((131, 77), 7), ((96, 96), 8), ((160, 250), 9), ((480, 640), 1234)])
def test_pyramid_planes(ctx, oracle, hw, seed):
    img = band_noise_image(hw[0], hw[1], seed, SMALL_BANDS if hw[0] < 400 else None or SMALL_BANDS)
    o = oracle.OracleRun(oracle.gray_from_u8(img), keep_planes=True, detect_only=True)
    pyr = ctx.pyramid(img)
    assert len(pyr) == o.n_octaves()
    for oi, (Ls, Rs) in enumerate(pyr):
        for l in range(5):
            assert_bit_equal(Ls[l], o.plane(oi, 0, l), "octave %d L%d" % (oi, l))
            ref = o.plane(oi, 1, l)
            assert_bit_equal(Rs[l][1:-1, 1:-1], ref[1:-1, 1:-1], "octave %d R%d" % (oi, l))


@pytest.mark.parametrize("hw,seed", [((131, 77), 7), ((96, 96), 8), ((240, 320), 10), ((480, 640), 1234)])
def test_hessian_keypoints(ctx, oracle, hw, seed):
    img = band_noise_image(hw[0], hw[1], seed, SMALL_BANDS)
    o = oracle.OracleRun(oracle.gray_from_u8(img), detect_only=True)
    f, i, n = ctx.hessian_keypoints(img)
    of, oi = o.hessian()
    assert n == o.n_hessian
    assert np.array_equal(i, oi), "type/octave/level/r0/c0 (detection order)"
    assert_bit_equal(f[:, :5], of[:, :5], "x,y,s,pd,response")


def test_affine_shape_stage(ctx, oracle):
    img = band_noise_image(240, 320, 21, SMALL_BANDS)
    o = oracle.OracleRun(oracle.gray_from_u8(img), keep_planes=True)
    f, i = o.hessian()
    U, ci = o.affine()
    # keypoints of octave 0, level 1 on their prevBlur plane
    sel = np.where((i[:, 1] == 0) & (i[:, 2] == 1))[0]
    assert len(sel) > 50
    blur = o.plane(0, 0, 1)
    conv, Ug, it = ctx.find_affine_shape(blur, f[sel][:, :4])
    assert np.array_equal(conv, ci[sel, 0])
    ok = conv == 1
    assert np.array_equal(it[ok], ci[sel, 1][ok])
    assert_bit_equal(Ug[ok], U[sel][ok], "U")


def test_normalize_affine_and_sift_stage(ctx, oracle):
    img = band_noise_image(300, 400, 22)
    gray = oracle.gray_from_u8(img)
    o = oracle.OracleRun(gray)
    g, t, d = o.keys()
    assert o.n_keys > 100
    L = oracle.lib()
    kp = g[:, :3].copy(); A = g[:, 3:7].copy()
    rej, patches = ctx.normalize_affine(gray, kp, A)
    assert not rej.any()
    ref = np.zeros((len(kp), 41, 41), np.float32)
    for k in range(len(kp)):
        r = L.ho_normalize_affine(gray, gray.shape[0], gray.shape[1], float(kp[k, 0]), float(kp[k, 1]), float(kp[k, 2]), A[k], ref[k].reshape(-1))
        assert r == 0
    assert_bit_equal(patches, ref, "patches")
    desc = ctx.sift(ref)
    assert np.array_equal(desc, d), "descriptors from oracle patches"


@pytest.mark.parametrize("hw,seed,bands", [((131, 77), 7, SMALL_BANDS), ((96, 96), 8, SMALL_BANDS), ((480, 640), 1234, None),
                                           ((20, 15), 1, SMALL_BANDS), ((12, 40), 2, SMALL_BANDS), ((1080, 1920), 1235, None)])
def test_end_to_end(ctx, oracle, hw, seed, bands):
    img = band_noise_image(hw[0], hw[1], seed, bands) if bands else band_noise_image(hw[0], hw[1], seed)
    o = oracle.OracleRun(oracle.gray_from_u8(img))
    (n_hess, keys), = ctx.detect_batch([img])
    g, t, d = o.keys()
    assert n_hess == o.n_hessian
    assert len(keys) == o.n_keys
    if len(keys) == 0:
        return
    assert np.array_equal(keys["desc"], d), "128-D integer descriptors"
    assert np.array_equal(keys["type"], t)
    for j, name in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
        assert_bit_equal(keys[name], g[:, j], name)
    import hesaff_amd
    txt = hesaff_amd.format_sift(keys, ctx.params.mrSize)
    assert txt == o.export_text()


# ------------------------------------------------------------------------------------------
# golden files, batching, colour input, device-resident entry point, CLI, full-size properties
# ------------------------------------------------------------------------------------------
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name", ["band_131x77", "band_96x96", "band_160x120", "tiny_20x15", "thin_12x40"])
def test_golden_files(ctx, name):
    import hesaff_amd
    img = hesaff_amd.read_pnm(os.path.join(GOLD, name + ".pgm"))
    (n_hess, keys), = ctx.detect_batch([img])
    txt = hesaff_amd.format_sift(keys, ctx.params.mrSize)
    assert txt == open(os.path.join(GOLD, name + ".hesaff.sift"), "rb").read()


def test_cli_drop_in(tmp_path):
    """`hesaff <image>` writes <image>.hesaff.sift and prints the reference's stdout line (hesaff.cpp:168-175)."""
    src = os.path.join(GOLD, "band_160x120.pgm")
    dst = tmp_path / "img.pgm"
    shutil.copy(src, dst)
    r = subprocess.run([os.path.join(ROOT, "hesaff_amd", "bin", "hesaff"), str(dst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = re.fullmatch(r"Detected (\d+) keypoints and (\d+) affine shapes in [0-9.e+-]+ sec\.\n", r.stdout)
    assert m, r.stdout
    out = (tmp_path / "img.pgm.hesaff.sift").read_bytes()
    assert out == open(os.path.join(GOLD, "band_160x120.hesaff.sift"), "rb").read()
    assert int(m.group(2)) == int(out.split(b"\n")[1])


def test_cli_batch_mode(tmp_path):
    """`hesaff --batch list` (extension): every image gets the file the single-image form writes."""
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    names = []
    for i, (h, w, seed) in enumerate(((120, 160, 5), (131, 77, 7), (120, 160, 6))):
        img = band_noise_image(h, w, seed, SMALL_BANDS)
        p = tmp_path / ("b%d.pgm" % i)
        p.write_bytes(b"P5\n%d %d\n255\n" % (w, h) + img.tobytes())
        names.append(str(p))
    lst = tmp_path / "list.txt"
    lst.write_text("# three images, two sizes\n" + "\n".join(names) + "\n")
    r = subprocess.run([exe, "--batch", str(lst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert len(lines) == 4 and re.fullmatch(r"Detected \d+ keypoints and \d+ affine shapes in 3 images in [0-9.e+-]+ sec\.", lines[3]), r.stdout
    batch_out = [open(n + ".hesaff.sift", "rb").read() for n in names]
    for n, want in zip(names, batch_out):
        os.remove(n + ".hesaff.sift")
        r1 = subprocess.run([exe, n], capture_output=True, text=True)
        assert r1.returncode == 0, r1.stderr
        assert open(n + ".hesaff.sift", "rb").read() == want and len(want) > 1000
    # options in any order, before or after the list (they were only accepted after it)
    for n in names:
        os.remove(n + ".hesaff.sift")
    r = subprocess.run([exe, "--output", "both", "--devices", "0", "--batch", str(lst), "--schedule", "dynamic"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert [open(n + ".hesaff.sift", "rb").read() for n in names] == batch_out and all(os.path.exists(n + ".hesaff.bin") for n in names)
    assert subprocess.run([exe, "--devices", "0", "--batch"], capture_output=True, text=True).returncode == 1


def test_cli_multi_device_shards_and_isolates_bad_files(tmp_path):
    """`hesaff --batch list --devices 0,0`: two device contexts on two host threads of one process, the list split into
    contiguous shards (hesaff_shard_range) -- every .hesaff.sift byte equals the single-context run; an unreadable file
    is reported and skipped without losing the rest of the batch."""
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    names = []
    for i, (h, w, seed) in enumerate(((120, 160, 5), (131, 77, 7), (120, 160, 6), (200, 140, 8), (96, 96, 9), (120, 160, 10), (77, 131, 11))):
        img = band_noise_image(h, w, seed, SMALL_BANDS)
        q = tmp_path / ("m%d.pgm" % i)
        q.write_bytes(b"P5\n%d %d\n255\n" % (w, h) + img.tobytes())
        names.append(str(q))
    bad = tmp_path / "broken.pgm"
    bad.write_bytes(b"P5\n10 10\n255\nshort")
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(names[:3] + [str(bad)] + names[3:]) + "\n")

    def run(devs, extra=()):
        for n in names:
            if os.path.exists(n + ".hesaff.sift"):
                os.remove(n + ".hesaff.sift")
        r = subprocess.run([exe, "--batch", str(lst), "--devices", devs] + list(extra), capture_output=True, text=True)
        assert r.returncode == 1, (r.stdout, r.stderr)               # the broken file
        assert "broken.pgm" in r.stderr and "skipped" in r.stderr
        m = re.search(r"Detected (\d+) keypoints and (\d+) affine shapes in 7 images", r.stdout)
        assert m, r.stdout
        return [open(n + ".hesaff.sift", "rb").read() for n in names], (int(m.group(1)), int(m.group(2)))
    one, tot1 = run("0")
    two, tot2 = run("0,0")
    three, tot3 = run("0,0,0")
    assert tot1 == tot2 == tot3 and tot1[1] > 300
    assert one == two == three and all(len(b) > 100 for b in one)
    # dynamic schedule (device contexts pull blocks of the list) and the binary sidecar next to the text: the same text files
    dyn, totd = run("0,0", ("--schedule", "dynamic", "--output", "both"))
    assert dyn == one and totd == tot1 and all(os.path.getsize(n + ".hesaff.bin") == 16 + 148 * int(b.split(b"\n")[1]) for n, b in zip(names, one))
    r = subprocess.run([exe, "--batch", str(lst), "--devices", "0-99"], capture_output=True, text=True)
    assert r.returncode == 1 and "--devices" in r.stderr
    r = subprocess.run([exe, "--batch", str(lst), "--devices", "0-99999999999"], capture_output=True, text=True)   # no expansion before the check
    assert r.returncode == 1 and "--devices" in r.stderr
    r = subprocess.run([exe, "--batch", str(lst), "--output", "xml"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    # --fast 2: same keypoints (row counts), other descriptors for the larger windows
    fast, totf = run("0", ("--fast", "2"))
    assert totf == tot1 and [b.split(b"\n")[1] for b in fast] == [b.split(b"\n")[1] for b in one] and fast != one


def test_same_bytes_under_one_and_two_ranks(tmp_path):
    """the same 8 images under world 1 and under world 2 (one process per rank, torch.distributed with gloo,
    both ranks on this box's GPU): every .hesaff.sift byte is equal, and equal to the C++ CLI's.  The driver's real runs
    put one GPU under each rank and gather the counts with RCCL."""
    import json
    import sys
    names = []
    for i, (h, w, seed) in enumerate(((120, 160, 21), (120, 160, 22), (131, 77, 23), (200, 140, 24), (96, 96, 25), (120, 160, 26), (77, 131, 27), (160, 120, 28))):
        img = band_noise_image(h, w, seed, SMALL_BANDS)
        q = tmp_path / ("r%d.pgm" % i)
        q.write_bytes(b"P5\n%d %d\n255\n" % (w, h) + img.tobytes())
        names.append(str(q))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(names) + "\n")
    tool = os.path.join(ROOT, "tools", "batch_ranks.py")

    def collect():
        out = [open(n + ".hesaff.sift", "rb").read() for n in names]
        for n in names:
            os.remove(n + ".hesaff.sift")
        return out
    r = subprocess.run([sys.executable, tool, str(lst)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    one = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    b1 = collect()
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", tool, str(lst), "--one-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    two = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    b2 = collect()
    assert one["world"] == 1 and two["world"] == 2 and two["per_rank_images"] == [4, 4]
    assert (one["images"], one["hessian_keypoints"], one["descriptors"]) == (two["images"], two["hessian_keypoints"], two["descriptors"])
    assert one["images"] == 8 and one["descriptors"] > 300
    assert b1 == b2 and all(len(b) > 100 for b in b1)
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    r = subprocess.run([exe, "--batch", str(lst)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert collect() == b1


def test_two_contexts_on_two_threads_shard_a_batch(ctx):
    """Image-level sharding inside one process: two contexts driven by two host threads, contiguous blocks
    from shard_range, results identical to one context over the whole list."""
    import threading
    import hesaff_amd
    from hesaff_amd.shard import shard_range
    imgs = [band_noise_image(150 + 10 * (i % 3), 200 + 8 * (i % 2), 60 + i, SMALL_BANDS) for i in range(9)]
    want = ctx.detect_batch(imgs)
    got = [None] * len(imgs)
    errs = []

    def worker(rank):
        try:
            lo, hi = shard_range(len(imgs), rank, 2)
            with hesaff_amd.HesaffContext(device=0) as c2:
                for k, r in enumerate(c2.detect_batch(imgs[lo:hi])):
                    got[lo + k] = r
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for (nw, kw), (ng, kg) in zip(want, got):
        assert nw == ng and kw.tobytes() == kg.tobytes()


def test_batch_and_mixed_sizes(ctx):
    a = band_noise_image(200, 300, 31, SMALL_BANDS)
    b = band_noise_image(200, 300, 32, SMALL_BANDS)
    c = band_noise_image(131, 77, 7, SMALL_BANDS)
    single = {k: ctx.detect_batch([im])[0] for k, im in (("a", a), ("b", b), ("c", c))}
    res = ctx.detect_batch([a, c, b, a, c])
    for got, want in zip(res, ["a", "c", "b", "a", "c"]):
        assert got[0] == single[want][0]
        assert got[1].tobytes() == single[want][1].tobytes(), "result depends on the batch position"


def test_colour_input(ctx, oracle):
    rng = np.random.default_rng(4)
    base = band_noise_image(150, 210, 33, SMALL_BANDS).astype(np.int32)
    bgr = np.stack([np.clip(base + rng.integers(-20, 20, base.shape), 0, 255) for _ in range(3)], axis=-1).astype(np.uint8)
    o = oracle.OracleRun(oracle.gray_from_u8(bgr))
    (n_hess, keys), = ctx.detect_batch([bgr])
    g, t, d = o.keys()
    assert n_hess == o.n_hessian and len(keys) == o.n_keys and o.n_keys > 50
    assert np.array_equal(keys["desc"], d)
    assert_bit_equal(keys["x"], g[:, 0], "x")


def test_device_resident_entry_point(ctx):
    import torch
    import hesaff_amd
    imgs = np.stack([band_noise_image(240, 320, 40 + i, SMALL_BANDS) for i in range(3)])
    host = ctx.detect_batch(list(imgs))
    p = hesaff_amd.default_params(); p.max_batch = 4
    with hesaff_amd.HesaffContext(p, device=0) as c2:
        t = torch.from_numpy(imgs).cuda()
        ch, cd, dkeys, total = c2.detect_batch_device(t.data_ptr(), 3, 320, 240)
        assert [int(v) for v in ch] == [h[0] for h in host]
        assert [int(v) for v in cd] == [len(h[1]) for h in host] and total == sum(len(h[1]) for h in host)
        # copy the device records back through torch and compare bytes
        import ctypes
        buf = torch.empty(total * 164, dtype=torch.uint8, device="cuda")
        # the HIP runtime torch loaded (a second runtime in the process would not see the GPU)
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(dkeys), ctypes.c_size_t(total * 164), 3) == 0
        got = buf.cpu().numpy().tobytes()
        assert got == b"".join(h[1].tobytes() for h in host)


def test_full_size_4k(ctx, oracle):
    """BASELINE config size (3840x2160): full parity against the oracle plus size-independent properties."""
    import hesaff_amd
    img = band_noise_image(2160, 3840, 1234)
    (n1, k1), = ctx.detect_batch([img])
    (n2, k2), = ctx.detect_batch([img])
    assert n1 == n2 and k1.tobytes() == k2.tobytes(), "run-to-run determinism"
    assert len(k1) <= n1 and len(k1) > 50000
    # rectified shapes: a12 == 0, det == 1 (helpers.cpp:95-96, affine.cpp:105)
    assert not k1["a12"].any()
    assert np.abs(k1["a11"].astype(np.float64) * k1["a22"] - 1).max() < 1e-5
    assert (np.abs(k1["response"]) >= np.float32(16.0 / 3.0) ** 2).all() and set(np.unique(k1["type"])) <= {0, 1, 2}
    # descriptor: clipped at 0.2 then renormalised and x512 -> no element above 255, norm ~ 512
    nrm = np.sqrt((k1["desc"].astype(np.float64) ** 2).sum(axis=1))
    assert nrm.min() > 400 and nrm.max() < 520
    o = oracle.OracleRun(oracle.gray_from_u8(img))
    g, t, d = o.keys()
    assert n1 == o.n_hessian and len(k1) == o.n_keys
    assert np.array_equal(k1["desc"], d)
    for j, name in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
        assert_bit_equal(k1[name], g[:, j], name)
    # exported ellipse within the north_star tolerance (1e-4 relative) of the oracle's closed form
    e = hesaff_amd.ellipse(k1[:2000], ctx.params.mrSize)
    ref = np.zeros((2000, 3), np.float32)
    for i in range(2000):
        oracle.lib().ho_ellipse(g[i], ctx.params.mrSize, ref[i])
    assert np.abs(e - ref).max() <= 1e-4 * np.abs(ref).max()


def _params(**kw):
    import hesaff_amd
    p = hesaff_amd.default_params()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _assert_keys_equal_oracle(keys, n_hess, o, what=""):
    g, t, d = o.keys()
    assert n_hess == o.n_hessian and len(keys) == o.n_keys, (what, n_hess, o.n_hessian, len(keys), o.n_keys)
    if len(keys):
        assert np.array_equal(keys["desc"], d), what
        assert np.array_equal(keys["type"], t), what
        for j, name in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
            assert_bit_equal(keys[name], g[:, j], "%s %s" % (what, name))


# Non-default parameters (every reference-side field of hesaff_params): pyramid.h:18-41, affine.h:17-46,
# siftdesc.h:19-32, hesaff.cpp:154-163.  mrSize = 1 sends every small keypoint through normalizeAffine's
# direct branch (imageToPatchScale <= 0.4, affine.cpp:137-141); initialSigma 1.0 / 2.0 / 0.45 give blur tap
# counts outside 9..15 (LDS-tile pyramid kernel, SymmRowSmallFilter order for K <= 5; two-pass kernels above
# 15 taps) and, at 0.45, no initial blur at all (pyramid.cpp:276).
NONDEFAULT = [
    dict(threshold=9.0),
    dict(threshold=2.5, edgeEigenValueRatio=4.0),
    dict(mrSize=1.0),
    dict(mrSize=2.0, maxBinValue=0.1),
    dict(mrSize=9.0),
    dict(maxIterations=3),
    dict(maxIterations=40, convergenceThreshold=0.01),
    dict(convergenceThreshold=0.2),
    dict(initialSigma=1.0),
    dict(initialSigma=2.0),
    dict(initialSigma=0.45, threshold=3.0),
    dict(maxBinValue=0.05),
    dict(maxBinValue=1.0),
    dict(upscaleInputImage=1),                       # pyramid.cpp:267-271: first octave on the 2x up-sampled image, pixelDistance 0.5
    dict(upscaleInputImage=1, initialSigma=0.9),     # ... and no initial blur (initialSigma <= 2 * 0.5)
]


@pytest.mark.parametrize("kw", NONDEFAULT, ids=lambda kw: ",".join("%s=%g" % kv for kv in kw.items()))
def test_non_default_parameters(oracle, kw):
    import hesaff_amd
    p = _params(**kw)
    imgs = [band_noise_image(300, 420, 91), band_noise_image(200, 260, 92, SMALL_BANDS)]
    with hesaff_amd.HesaffContext(p, device=0) as c2:
        res = c2.detect_batch(imgs)
        total = 0
        for img, (n_hess, keys) in zip(imgs, res):
            o = oracle.OracleRun(oracle.gray_from_u8(img), params=p)
            _assert_keys_equal_oracle(keys, n_hess, o, str(kw))
            assert hesaff_amd.format_sift(keys, p.mrSize) == o.export_text()
            total += len(keys)
        assert total > 30, "parameter set leaves too few keypoints for a meaningful comparison"


def test_direct_branch_is_taken_with_small_mr_size(oracle):
    """With mrSize = 1 the windows of the finest keypoints have imageToPatchScale = P0/41 <= 0.4 (P0 <= 15)."""
    p = _params(mrSize=1.0)
    o = oracle.OracleRun(oracle.gray_from_u8(band_noise_image(300, 420, 91)), params=p)
    g, _, _ = o.keys()
    P0 = 2 * np.ceil(g[:, 2] * np.float32(1.0)).astype(int) + 1
    assert (P0 / 41.0 <= 0.4).sum() > 100 and (P0 / 41.0 > 0.4).sum() > 20


def test_normalize_affine_direct_branch_stage(ctx, oracle):
    """hesaff_stage_normalize_affine with scales small enough for affine.cpp:137-141 (no smoothing), anisotropic
    shapes, next to keypoints that take the smoothing branch, against the oracle's normalizeAffine."""
    rng = np.random.default_rng(77)
    gray = oracle.gray_from_u8(band_noise_image(260, 340, 23))
    n = 300
    kp = np.zeros((n, 3), np.float32)
    kp[:, 0] = rng.uniform(40, 300, n); kp[:, 1] = rng.uniform(40, 220, n)
    kp[:, 2] = np.where(np.arange(n) % 3 == 0, rng.uniform(1.6, 4.0, n), rng.uniform(0.2, 1.5, n))   # s * mrSize <= 7 -> P0 <= 15
    a11 = rng.uniform(0.6, 1.7, n).astype(np.float32)
    A = np.stack([a11, np.zeros(n, np.float32), rng.uniform(-0.5, 0.5, n).astype(np.float32), (np.float32(1) / a11)], axis=1).astype(np.float32)
    rej, patches = ctx.normalize_affine(gray, kp, A)
    oh = oracle.OracleHandle()
    n_direct = 0
    for k in range(n):
        r, ref = oh.normalize_affine(gray, kp[k, 0], kp[k, 1], kp[k, 2], A[k])
        assert bool(rej[k]) == bool(r), k
        if not r:
            assert_bit_equal(patches[k], ref, "patch %d (s=%g)" % (k, kp[k, 2]))
            P0 = 2 * int(np.ceil(np.float32(kp[k, 2]) * ctx.params.mrSize)) + 1
            n_direct += P0 / 41.0 <= 0.4
    assert n_direct > 100 and (rej == 0).sum() - n_direct > 50


def test_stage_entry_points_with_non_default_sift_clip(oracle):
    """computeSiftDescriptor with maxBinValue 0.08 / 0.5 through hesaff_stage_sift (production descriptor kernels)."""
    import hesaff_amd
    img = band_noise_image(200, 280, 24)
    gray = oracle.gray_from_u8(img)
    o = oracle.OracleRun(gray)
    g, _, _ = o.keys()
    oh0 = oracle.OracleHandle()
    patches = np.stack([oh0.normalize_affine(gray, g[k, 0], g[k, 1], g[k, 2], g[k, 3:7])[1] for k in range(min(len(g), 200))])
    for mbv in (0.08, 0.5):
        p = _params(maxBinValue=mbv)
        oh = oracle.OracleHandle(p)
        with hesaff_amd.HesaffContext(p, device=0) as c2:
            got = c2.sift(patches)
        want = np.stack([oh.sift(pp) for pp in patches])
        assert np.array_equal(got, want), mbv


def _device_keys(dkeys, total):
    """Copy `total` hesaff_keypoint records from the library's device buffer through torch's HIP runtime."""
    import ctypes
    import torch
    import hesaff_amd
    buf = torch.empty(max(total, 1) * 164, dtype=torch.uint8, device="cuda")
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    if total:
        assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(dkeys), ctypes.c_size_t(total * 164), 3) == 0
    return np.frombuffer(buf.cpu().numpy().tobytes()[: total * 164], dtype=hesaff_amd.KEYPOINT_DTYPE)


@pytest.mark.parametrize("width,height,name", [(1920, 1080, "BASELINE config 3: batch of 256 x 1920x1080"),
                                               (3840, 2160, "per-GPU share of BASELINE config 4: 256 x 3840x2160")])
def test_full_batch_configs(oracle, width, height, name):
    """BASELINE.json's batch configurations through hesaff_detect_batch_device at their full size: size-independent
    properties on the whole batch, and every field / byte of two sampled images against the oracle."""
    import torch
    import hesaff_amd
    from hesaff_amd.synth import band_noise_batch_torch
    B = 256
    imgs = band_noise_batch_torch(B, height, width, seed=4321, device="cuda")
    p = _params(max_batch=B)
    with hesaff_amd.HesaffContext(p, device=0) as c2:
        ch, cd, dkeys, total = c2.detect_batch_device(imgs.data_ptr(), B, width, height)
        keys = _device_keys(dkeys, total).copy()
        ch2, cd2, dkeys2, total2 = c2.detect_batch_device(imgs.data_ptr(), B, width, height)
        assert total2 == total and np.array_equal(ch, ch2) and np.array_equal(cd, cd2)
        assert _device_keys(dkeys2, total2).tobytes() == keys.tobytes(), "run-to-run determinism of the whole batch"
    assert total == int(cd.sum()) and (cd <= ch).all() and (cd > 0.5 * ch).all()
    mpx = width * height / 1e6
    assert (cd > 5000 * mpx).all() and (cd < 40000 * mpx).all()
    # rectified shapes: a12 == 0, det == 1 (helpers.cpp:95-96, affine.cpp:105); thresholded responses; clipped descriptors
    assert not keys["a12"].any()
    assert np.abs(keys["a11"].astype(np.float64) * keys["a22"] - 1).max() < 1e-5
    assert (np.abs(keys["response"]) >= np.float32(16.0 / 3.0) ** 2).all() and set(np.unique(keys["type"])) <= {0, 1, 2}
    assert (keys["x"] > 0).all() and (keys["x"] < width).all() and (keys["y"] > 0).all() and (keys["y"] < height).all()
    nrm = np.sqrt((keys["desc"][::97].astype(np.float64) ** 2).sum(axis=1))
    assert nrm.min() > 400 and nrm.max() < 520
    starts = np.concatenate([[0], np.cumsum(cd)])
    for b in (0, 171):
        o = oracle.OracleRun(oracle.gray_from_u8(imgs[b].cpu().numpy()))
        _assert_keys_equal_oracle(keys[starts[b]:starts[b + 1]], int(ch[b]), o, "%s, image %d" % (name, b))
    del imgs
    torch.cuda.empty_cache()


def test_out_of_memory_is_recoverable(oracle):
    """HESAFF_ERR_NOMEM in the middle of a buffer plan (a capacity request no device can hold) leaves the context
    usable: the earlier geometry is planned afresh instead of running on half-replaced buffers."""
    import hesaff_amd
    img = band_noise_image(240, 320, 55, SMALL_BANDS)
    p = _params(max_kpts_per_mpx=100_000_000)   # 16 Mpx x 1e8 = 1.6e9 candidates: ~700 GB of lists
    with hesaff_amd.HesaffContext(device=0) as ref_ctx:
        (n0, k0), = ref_ctx.detect_batch([img])
    with hesaff_amd.HesaffContext(p, device=0) as c2:
        small = np.full((64, 64), 7, np.uint8)
        assert c2.detect_batch([small])[0][0] == 0
        with pytest.raises(hesaff_amd.HesaffError) as e:
            c2.detect_batch([np.zeros((4000, 4000), np.uint8)])
        assert e.value.code == -5, e.value
        assert c2.detect_batch([small])[0][0] == 0            # the geometry used before the failure
        (n1, k1), = c2.detect_batch([img])                     # and a new one
        assert n1 == n0 and k1.tobytes() == k0.tobytes()


def test_bad_arguments_are_rejected():
    import hesaff_amd
    for kw in (dict(initialSigma=0.0), dict(initialSigma=float("nan")), dict(mrSize=-1.0), dict(maxIterations=0),
               dict(threshold=float("inf")), dict(initialSigma=100.0)):
        with pytest.raises(hesaff_amd.HesaffError) as e:
            hesaff_amd.HesaffContext(_params(**kw), device=0)
        assert e.value.code == -2, kw


def test_fast_level_2_keeps_geometry_and_small_windows(ctx):
    """hesaff_params.fast = 2 replaces normalizeAffine's warp + blur by samples of the matching scale-space level for every window
    larger than the 41 x 41 patch - another algorithm for those keypoints, not an approximation of the arithmetic.  What must
    hold: detection, affine shapes and the set of described keypoints are those of parity mode, keypoints with small windows keep
    their parity descriptor bit for bit, the others stay correlated with it - also on the 2x up-sampled pyramid (the
    level is chosen in pixels of the ORIGINAL image).  fast = 1 (withdrawn) is refused."""
    import hesaff_amd
    imgs = [band_noise_image(480, 640, 77)]
    for up in (0, 1):
        with hesaff_amd.HesaffContext(_params(upscaleInputImage=up), device=0) as c1, \
                hesaff_amd.HesaffContext(_params(fast=2, upscaleInputImage=up), device=0) as c2:
            (n1, k1), = c1.detect_batch(imgs)
            (n2, k2), = c2.detect_batch(imgs)
            (_, k3), = c2.detect_batch(imgs)
            mr = c1.params.mrSize
        assert n1 == n2 and len(k1) == len(k2) > 3000 and k2.tobytes() == k3.tobytes()
        for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "type"):
            assert np.array_equal(k1[f], k2[f]), f
        P0 = 2 * np.ceil(k1["s"] * np.float32(mr)).astype(np.int64) + 1
        small = P0 + 2 <= 41
        assert small.sum() > 1000 and (~small).sum() > 500
        assert np.array_equal(k1["desc"][small], k2["desc"][small])
        a = k1["desc"][~small].astype(np.float64); b = k2["desc"][~small].astype(np.float64)
        cos = (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-9)
        assert np.median(cos) > 0.97 and np.percentile(cos, 10) > 0.9, (up, float(np.median(cos)), float(np.percentile(cos, 10)))
    with pytest.raises(hesaff_amd.HesaffError):
        hesaff_amd.HesaffContext(_params(fast=1), device=0)


def test_survey_probe_output_md5_on_gpu(ctx):
    """The product's .hesaff.sift of SURVEY App. C's 640x480 probe image has the md5 the survey recorded from the
    COMPILED reference's output file (e004ba88...): 4183 rows, every coordinate, ellipse term and descriptor byte."""
    import hashlib
    import json
    import hesaff_amd
    man = json.load(open(os.path.join(GOLD, "manifest.json")))["probe_vga"]
    img = hesaff_amd.read_pnm(os.path.join(GOLD, "probe_vga.pgm"))
    (n_hess, keys), = ctx.detect_batch([img])
    rec = man["survey_recorded"]
    assert (n_hess, len(keys)) == (rec["hessian"], rec["descriptors"])
    md5 = hashlib.md5(hesaff_amd.format_sift(keys, ctx.params.mrSize)).hexdigest()
    assert md5.startswith(rec["sift_md5_prefix"]) and md5 == man["sift_md5"]


def test_empty_and_featureless_inputs(ctx, oracle):
    """No images, an image without a single extremum, and a batch mixing such images with normal ones."""
    import hesaff_amd
    assert ctx.detect_batch([]) == []
    flat = np.full((200, 300), 127, np.uint8)
    ramp = np.tile(np.arange(300, dtype=np.uint8), (200, 1))
    for im in (flat, ramp):
        (nh, keys), = ctx.detect_batch([im])
        o = oracle.OracleRun(oracle.gray_from_u8(im))
        assert nh == o.n_hessian == 0 and len(keys) == o.n_keys == 0
        assert hesaff_amd.format_sift(keys, ctx.params.mrSize) == b"128\n0\n"
    normal = band_noise_image(200, 300, 31, SMALL_BANDS)
    res = ctx.detect_batch([flat, normal, ramp, normal])
    assert res[0][0] == 0 and res[2][0] == 0 and len(res[0][1]) == 0 and len(res[2][1]) == 0
    assert res[1][0] > 100 and res[1][1].tobytes() == res[3][1].tobytes() == ctx.detect_batch([normal])[0][1].tobytes()


def test_capacity_error_is_reported():
    """More keypoints than max_kpts_per_mpx allows: HESAFF_ERR_CAPACITY, nothing truncated silently."""
    import hesaff_amd
    p = hesaff_amd.default_params()
    p.max_kpts_per_mpx = 1000   # the library's floor: capacity = max(4096, 1000 per megapixel)
    with hesaff_amd.HesaffContext(p, device=0) as small:
        with pytest.raises(hesaff_amd.HesaffError) as e:
            small.detect_batch([band_noise_image(720, 1280, 1234)])   # ~12 k keypoints against 4096
        assert "capacity" in str(e.value).lower()
        # the context stays usable
        (nh, keys), = small.detect_batch([np.full((64, 64), 10, np.uint8)])
        assert nh == 0 and len(keys) == 0


def test_cli_reads_png(tmp_path):
    """The same image as PNG (all five filter types) gives the golden file of its PGM form."""
    import hesaff_amd
    from tests.test_host_side import _png_bytes
    img = hesaff_amd.read_pnm(os.path.join(GOLD, "band_160x120.pgm"))
    dst = tmp_path / "img.png"
    dst.write_bytes(_png_bytes(img[:, :, None], 0))
    r = subprocess.run([os.path.join(ROOT, "hesaff_amd", "bin", "hesaff"), str(dst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "img.png.hesaff.sift").read_bytes() == open(os.path.join(GOLD, "band_160x120.hesaff.sift"), "rb").read()


def test_cli_reads_jpeg(tmp_path):
    """`hesaff img.jpg`: the baseline JPEG decoder feeds the detector the pixels libjpeg decodes (committed fixture + the
    PPM of its libjpeg-turbo decode): both files give the same .hesaff.sift."""
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    outs = []
    for name in ("jpeg_420_q85.jpg", "jpeg_420_q85.ppm"):
        dst = tmp_path / name
        shutil.copy(os.path.join(GOLD, name), dst)
        r = subprocess.run([exe, str(dst)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append((tmp_path / (name + ".hesaff.sift")).read_bytes())
    assert outs[0] == outs[1] and len(outs[0]) > 5000


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def test_bench_two_ranks_on_one_gpu():
    """bench.py's multi-rank path (barriers, max-over-ranks time, count gather) with two ranks sharing
    this box's GPU: BENCH_DIST_BACKEND=gloo moves the three small collectives to CPU tensors; the
    driver's real runs use RCCL, one GPU per rank."""
    import json
    import sys
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "3",
           "--width", "640", "--height", "480", "--cpu-images", "1", "--cpu-workers", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout     # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["images_per_gpu_per_step"] == 3 and abs(d["images_per_s"] * d["ms_per_step"] / 1e3 - 6) < 1e-6
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0   # rank 0 keeps the CPU leg when world > 1


def test_bench_one_rank_through_rccl():
    """bench.py under torchrun with ONE rank and the real backend: process group on RCCL, barrier, max all-reduce of the time and
    the count all-gather all run on the GPU (two ranks cannot share a device under RCCL, hence the gloo knob in the other tests)."""
    import json
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("BENCH_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "2",
           "--width", "640", "--height", "480", "--no-cpu-baseline", "--no-host-path"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["collective_backend"] == "nccl (RCCL)" and d["value"] > 0 and d["config"]["per_rank_images_timed"] == [2]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses for N = 1): bench.py starts the two
    ranks itself, before it touches the GPU, and rank 0's line says n_gpus == 2.  Strong scaling: 5 images per step in
    total, split 3 + 2 by hesaff_shard_range."""
    import json
    import sys
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2",
           "--width", "640", "--height", "480", "--no-cpu-baseline", "--no-host-path", "--scaling", "strong", "--global-images", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert sorted(d["config"]["per_rank_images_timed"]) == [2, 3] and d["config"]["images_per_step_all_ranks"] == 5
    assert abs(d["images_per_s"] * d["ms_per_step"] / 1e3 - 5) < 1e-6
    # a launcher that started a different number of ranks than --gpus asks for is an error, not a mislabelled line
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env2, cwd=ROOT, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_detect_batch_cb_streams_the_same_records(oracle):
    """hesaff_detect_batch_cb (bounded pinned memory: a ring of three result blocks) hands over, chunk by chunk, exactly
    the records hesaff_detect_batch returns; more chunks than ring blocks, mixed sizes, an image without keypoints."""
    import hesaff_amd
    sizes = [(120, 160), (96, 96), (120, 160), (131, 77), (96, 96), (120, 160), (120, 160), (77, 131), (120, 160), (96, 96), (120, 160)]
    imgs = [band_noise_image(h, w, 40 + i, SMALL_BANDS) for i, (h, w) in enumerate(sizes)]
    imgs[3] = np.full((131, 77), 90, np.uint8)     # featureless
    p = hesaff_amd.default_params(); p.max_batch = 2
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        want = ctx.detect_batch(imgs)
        got = {}
        calls = []

        def sink(idx, res):
            calls.append(list(idx))
            for i, r in zip(idx, res):
                assert i not in got
                got[i] = r
            return 0
        ctx.detect_batch_cb(imgs, sink)
        assert sorted(got) == list(range(len(imgs))) and len(calls) >= 7 and max(len(c) for c in calls) <= 2
        for i, (nh, keys) in enumerate(want):
            assert got[i][0] == nh and got[i][1].tobytes() == keys.tobytes(), i
        assert len(want[3][1]) == 0 and sum(len(k) for _, k in want) > 500
        # a sink that reports failure stops the run with an error
        with pytest.raises(hesaff_amd.HesaffError):
            ctx.detect_batch_cb(imgs, lambda idx, res: 1)
        # ... and the context is still usable
        again = ctx.detect_batch(imgs[:3])
        assert all(a[1].tobytes() == b[1].tobytes() for a, b in zip(again, want[:3]))


def test_process_files_pipeline(tmp_path, oracle):
    """hesaff_process_files (decode threads -> device -> writer threads; what `hesaff --batch` runs): every readable file gets
    the bytes the oracle writes for its pixels; an unreadable input and an unwritable output are reported per file and do
    not stop the others; explicit output names are honoured."""
    import hesaff_amd
    from tests import _oracle
    sizes = [(120, 160), (120, 160), (96, 96), (131, 77), (120, 160), (120, 160), (120, 160), (96, 96), (200, 140)]
    paths, outs, texts = [], [], []
    for i, (h, w) in enumerate(sizes):
        img = band_noise_image(h, w, 60 + i, SMALL_BANDS)
        q = tmp_path / ("f%02d.pgm" % i)
        q.write_bytes(b"P5\n%d %d\n255\n" % (w, h) + img.tobytes())
        paths.append(str(q)); outs.append(None)
        texts.append(_oracle.OracleRun(_oracle.gray_from_u8(img)).export_text())
    bad_in = tmp_path / "broken.pgm"; bad_in.write_bytes(b"P5\n10 10\n255\nshort")
    paths.insert(2, str(bad_in)); outs.insert(2, None); texts.insert(2, None)
    outs[4] = str(tmp_path / "elsewhere.txt")                          # explicit output name
    outs[6] = str(tmp_path / "no_such_dir" / "x.hesaff.sift")         # cannot be written
    p = hesaff_amd.default_params(); p.max_batch = 2
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        st = ctx.process_files(paths, outs, decode_threads=3, write_threads=3)
    OK, IO = 0, -4
    for i, (rc, stage, nh, nd) in enumerate(st):
        if i == 2:
            assert (rc, stage, nh, nd) == (IO, 1, 0, 0)                # HESAFF_FILE_UNREADABLE
        elif i == 6:
            assert rc == IO and stage == 2 and nd > 0                  # HESAFF_FILE_DETECTED, not written
        else:
            assert rc == OK and stage == 3, (i, rc, stage)
            name = outs[i] or paths[i] + ".hesaff.sift"
            assert open(name, "rb").read() == texts[i] and nd == int(texts[i].split(b"\n")[1]), i
    assert not os.path.exists(paths[4] + ".hesaff.sift")
    # defaults (threads = 0: auto, names = the reference's) on the readable files only; text + binary sidecar
    good = [q for i, q in enumerate(paths) if i != 2]
    with hesaff_amd.HesaffContext(device=0) as ctx:
        ctx.set_output_format(3)
        st = ctx.process_files(good)
        mr = ctx.params.mrSize
    assert all(rc == OK and stage == 3 for rc, stage, _, _ in st)
    for q, t in zip(good, [t for t in texts if t is not None]):
        assert open(q + ".hesaff.sift", "rb").read() == t
        rows = hesaff_amd.read_bin(q + ".hesaff.bin")
        lines = t.split(b"\n")
        assert len(rows) == int(lines[1])
        for i in range(0, len(rows), 37):          # the text is the %g print of the sidecar's floats
            tok = lines[2 + i].split()
            assert tok[:5] == [b"%g" % float(rows[k][i]) for k in ("x", "y", "a", "b", "c")] and [int(v) for v in tok[5:]] == rows["desc"][i].tolist()


def test_process_files_mixed_formats_against_the_oracle(tmp_path, oracle):
    """One list with PGM, colour PPM, PNG, sequential and progressive JPEG, BMP and TIFF files of several sizes through hesaff_process_files:
    every output equals the oracle's text for the pixels the in-tree readers deliver (pixels == Pillow's, i.e. libjpeg / zlib)."""
    import hesaff_amd
    from PIL import Image
    from tests import _oracle
    rng = np.random.default_rng(17)
    paths = []
    for i, (h, w) in enumerate(((150, 200), (150, 200), (131, 177), (200, 260), (150, 200), (96, 128), (150, 200), (131, 177), (150, 200), (96, 128))):
        g = band_noise_image(h, w, 300 + i, SMALL_BANDS)
        rgb = np.stack([g, np.roll(g, 3, 1), np.roll(g, 5, 0)], 2)
        kind = ("pgm", "ppm", "png", "jpg", "pjpg", "gpng", "bmp", "gbmp", "tif", "gtif")[i]
        q = str(tmp_path / ("m%d.%s" % (i, {"pjpg": "jpg", "gpng": "png", "gbmp": "bmp", "gtif": "tif"}.get(kind, kind))))
        if kind == "pgm":
            open(q, "wb").write(b"P5\n%d %d\n255\n" % (w, h) + g.tobytes())
        elif kind == "ppm":
            open(q, "wb").write(b"P6\n%d %d\n255\n" % (w, h) + rgb.tobytes())
        elif kind == "png":
            Image.fromarray(rgb, "RGB").save(q, "PNG")
        elif kind == "gpng":
            Image.fromarray(g, "L").save(q, "PNG")
        elif kind == "bmp":
            Image.fromarray(rgb, "RGB").save(q, "BMP")
        elif kind == "gbmp":
            Image.fromarray(g, "L").save(q, "BMP")
        elif kind == "tif":
            Image.fromarray(rgb, "RGB").save(q, "TIFF", compression="tiff_lzw", tiffinfo={317: 2})
        elif kind == "gtif":
            Image.fromarray(g, "L").save(q, "TIFF", compression="tiff_adobe_deflate")
        elif kind == "jpg":
            Image.fromarray(rgb, "RGB").save(q, "JPEG", quality=88, subsampling=1)
        else:
            Image.fromarray(rgb, "RGB").save(q, "JPEG", quality=80, subsampling=2, progressive=True)
        paths.append(q)
    p = hesaff_amd.default_params(); p.max_batch = 3
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        st = ctx.process_files(paths, decode_threads=2, write_threads=2)
    for q, (rc, stage, nh, nd) in zip(paths, st):
        assert rc == 0 and stage == 3, (q, rc, stage)
        pix = hesaff_amd.read_image(q)
        ref = np.asarray(Image.open(q).convert("RGB" if pix.ndim == 3 else "L"))
        assert np.array_equal(pix, ref), q
        o = _oracle.OracleRun(_oracle.gray_from_u8(pix))
        assert open(q + ".hesaff.sift", "rb").read() == o.export_text() and nd == o.n_keys and nh == o.n_hessian and nd > 50, q


def test_sequence_with_homographies_through_cli_and_repeatability_tool(tmp_path, oracle):
    """BASELINE.json config 5 on its offline stand-in (the Oxford data are not available): a graf-like sequence - one 800x640
    colour image and copies under known homographies, stored as 4:2:0 JPEG files next to H1toNp files like the Oxford sets -
    goes through `hesaff --batch` (in-tree JPEG decoder -> device -> writer threads) and tools/repeatability.py.  Every output
    file equals what the oracle writes for the decoded pixels, so repeatability and matching score are the restated
    reference's by construction; the table itself is checked for sanity."""
    import sys
    import hesaff_amd
    from tests import _oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import repeatability as rp
    angles = (20, 40)
    res = rp.sequence_through_cli(str(tmp_path / "seq"), 800, 640, angles)
    paths = [str(tmp_path / "seq" / ("img%d.jpg" % (k + 1))) for k in range(1 + len(angles))]
    from PIL import Image
    for q in paths:
        pix = hesaff_amd.read_image(q)
        assert pix.shape == (640, 800, 3) and np.array_equal(pix, np.asarray(Image.open(q).convert("RGB")))   # == libjpeg's pixels
        o = _oracle.OracleRun(_oracle.gray_from_u8(pix))
        assert open(q + ".hesaff.sift", "rb").read() == o.export_text() and o.n_keys > 2000
    assert os.path.exists(str(tmp_path / "seq" / "H1to2p"))
    pairs = res["pairs"]
    assert [e["viewpoint_deg"] for e in pairs] == list(angles)
    assert pairs[0]["repeatability"] > 0.6 and pairs[0]["matching_score"] > 0.5, pairs
    assert pairs[1]["repeatability"] > 0.4 and pairs[0]["repeatability"] > pairs[1]["repeatability"], pairs
    for e in pairs:
        assert e["correspondences"] >= e["matches"] > 500 and e["correspondences"] <= min(e["n1"], e["n2"])


def _structured_image(kind, h, w, rng):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "checker":
        s = int(rng.integers(3, 17))
        img = (((yy // s) + (xx // s)) % 2) * 255
    elif kind == "blobs":
        img = np.zeros((h, w))
        for _ in range(int(rng.integers(5, 60))):
            cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(1.5, 25)
            a = rng.uniform(0.3, 1.0); th = rng.uniform(0, np.pi)
            u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
            img += rng.choice([-1.0, 1.0]) * rng.uniform(60, 400) * np.exp(-(u * u + (v / a) ** 2) / (2 * r * r))
        img = 128 + img
    elif kind == "lines":
        img = np.full((h, w), 30.0)
        for _ in range(int(rng.integers(3, 25))):
            a, b, c = rng.normal(), rng.normal(), rng.uniform(-1, 1) * max(h, w)
            img[np.abs(a * xx + b * yy + c) / np.hypot(a, b) < rng.uniform(0.6, 3.0)] = rng.uniform(100, 255)
    elif kind == "saturated":
        img = band_noise_image(h, w, int(rng.integers(1 << 30)), SMALL_BANDS).astype(np.float64) * 3.0 - 256
    else:   # noise at pixel scale on top of a ramp
        img = xx * (255.0 / max(w - 1, 1)) + rng.normal(0, 25, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_fuzz_ragged_sizes_and_structured_content(ctx, oracle):
    """Forty images of odd sizes and unnatural content (checkerboards, saturated blobs, thin lines,
    clipped noise, ramps) as one ragged batch: every field and byte equals the oracle's."""
    rng = np.random.default_rng(20261001)
    kinds = ["checker", "blobs", "lines", "saturated", "ramp"]
    imgs = []
    for i in range(40):
        h, w = int(rng.integers(13, 260)), int(rng.integers(13, 330))
        imgs.append(_structured_image(kinds[i % len(kinds)], h, w, rng))
    res = ctx.detect_batch(imgs)
    total = 0
    for i, (img, (n_hess, keys)) in enumerate(zip(imgs, res)):
        o = oracle.OracleRun(oracle.gray_from_u8(img))
        g, t, d = o.keys()
        assert n_hess == o.n_hessian and len(keys) == o.n_keys, (i, img.shape, kinds[i % 5])
        if len(keys):
            assert np.array_equal(keys["desc"], d), (i, img.shape, kinds[i % 5])
            assert np.array_equal(keys["type"], t)
            for j, name in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
                assert_bit_equal(keys[name], g[:, j], "%s of image %d (%s %s)" % (name, i, kinds[i % 5], img.shape))
        total += len(keys)
    assert total > 3000


# ---------------------------------------------------------------------------------------------------------------
# exportKeypoints on the device (kernels_export.h; hesaff.cpp:107-130): the writer threads of hesaff_process_files only write()
# ---------------------------------------------------------------------------------------------------------------
def _fmt_g_inputs():
    rng = np.random.default_rng(77)
    every_exponent = (np.arange(0, 256, dtype=np.uint32)[:, None] << 23 | rng.integers(0, 1 << 23, (256, 64), dtype=np.uint32)).reshape(-1)
    parts = [
        rng.integers(0, 2**32, 1_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32),            # any bit pattern
        every_exponent.view(np.float32), (every_exponent | 0x80000000).view(np.float32),                    # every binade, both signs
        (np.arange(0, 256, dtype=np.uint32) << 23).view(np.float32),                                        # every power of two, 0, inf
        ((np.arange(1, 256, dtype=np.uint32) << 23) - 1).view(np.float32),                                  # ... and the value below it
        rng.integers(0, 1 << 23, 50_000, dtype=np.uint32).view(np.float32),                                 # denormals
        (10.0 ** np.arange(-45, 39, dtype=np.float64)).astype(np.float32),                                  # decimal powers
        np.nextafter((10.0 ** np.arange(-44, 39, dtype=np.float64)).astype(np.float32), np.float32(0)),     # ... and the value below
        rng.uniform(0, 4096, 500_000).astype(np.float32),                                                   # coordinates
        (10.0 ** rng.uniform(-8, 2, 500_000) * rng.choice([-1.0, 1.0], 500_000)).astype(np.float32),        # ellipse terms
        np.array([0.0, -0.0, 999999.5, 999999.4, 999999.96, 9.9999995e-5, 123456.5, 1234565.0, 8388608.0, 8388607.5, 16777216.0, 1e22,
                  1e-22, 9.5e-23, 1e23, np.inf, -np.inf, np.nan, -np.nan, 3.4028235e38, 1.17549435e-38, 1e-45], np.float32),
        (np.arange(0, 500_000, dtype=np.float32) + 0.5) / np.float32(8.0),                                  # exact binary ties
        (np.arange(1, 300_001, dtype=np.float64) * 1e-6 + 0.5e-6).astype(np.float32),                      # decimal near-ties
    ]
    return np.ascontiguousarray(np.concatenate(parts), np.float32)


def test_device_jpeg_pixels_equal_the_host_reader(ctx, tmp_path):
    """The device half of the JPEG reader (kernels_jpeg.h: inverse DCT, fancy / replicating up-sampling, colour conversion, run by
    hesaff_process_files on the coefficients of every JPEG of its list) delivers the bytes of hesaff_read_jpeg - which
    tests/test_host_side.py pins to libjpeg-turbo's - for 4:4:4, 4:2:2, 4:2:0, 4:4:0 and grey files, odd sizes down to 1 x 1, restart
    intervals, sequential and progressive, several images of one layout per call, the committed fixtures and the photographs."""
    Image = pytest.importorskip("PIL.Image")
    import hesaff_amd
    rng = np.random.default_rng(3)

    def synth(h, w, color):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / 7.0 + yy / 11.0), 127 + 100 * np.cos(xx / 5.0 - yy / 13.0),
                         127 + 90 * np.sin(xx / 3.0) * np.cos(yy / 4.0)], -1) + rng.normal(0, 25, (h, w, 3))
        a = np.clip(base, 0, 255).astype(np.uint8)
        return a if color else a[..., 0]
    n = 0
    p = str(tmp_path / "t.jpg")
    for (h, w) in [(64, 64), (61, 83), (7, 9), (1, 1), (17, 3), (120, 211), (33, 2), (1, 40)]:
        for sub in [0, 1, 2, "4:4:0", "gray"]:
            for q, extra in [(30, {}), (75, {"restart_marker_blocks": 3}), (100, {}), (85, {"progressive": True, "restart_marker_blocks": 2})]:
                kw = dict(quality=q, **extra)
                if sub != "gray":
                    kw["subsampling"] = sub
                try:
                    Image.fromarray(synth(h, w, sub != "gray")).save(p, "JPEG", **kw)
                except Exception:   # noqa: BLE001  (an encoder option this Pillow does not know)
                    continue
                want = hesaff_amd.read_image(p)
                lay, blob = hesaff_amd.read_jpeg_coefficients(p)
                got = ctx.jpeg_pixels(lay, blob)[0]
                assert got.shape == want.shape and np.array_equal(got, want), (h, w, sub, q, extra)
                n += 1
    assert n >= 120
    # luma sampled 4x1 / 1x4 (a 4:2:0 file with the sampling byte patched, tests/test_host_side.py): the replicating 4:1 up-sampling
    import io
    buf = io.BytesIO()
    Image.fromarray(synth(64, 64, True)).save(buf, "JPEG", quality=85, subsampling=2)
    raw = bytearray(buf.getvalue())
    sof = raw.find(b"\xff\xc0")
    for hv in (0x41, 0x14):
        raw[sof + 11] = hv
        open(p, "wb").write(raw)
        lay, blob = hesaff_amd.read_jpeg_coefficients(p)
        assert max(lay.hx[1], lay.vx[1]) == 4
        assert np.array_equal(ctx.jpeg_pixels(lay, blob)[0], hesaff_amd.read_image(p)), hex(hv)
    # several images of one layout in one call (what a chunk is), a size that is no multiple of the MCU
    blobs, wants = [], []
    for k in range(5):
        Image.fromarray(synth(203, 317, True)).save(p, "JPEG", quality=70 + 5 * k, subsampling=2)
        lay, blob = hesaff_amd.read_jpeg_coefficients(p)
        blobs.append(blob); wants.append(hesaff_amd.read_image(p))
    got = ctx.jpeg_pixels(lay, np.stack(blobs))
    assert np.array_equal(got, np.stack(wants))
    files = [os.path.join(GOLD, f) for f in sorted(os.listdir(GOLD)) if f.endswith(".jpg")]
    from hesaff_amd.synth import sample_photo_paths
    files += list(sample_photo_paths())
    assert len(files) >= 5
    for f in files:
        lay, blob = hesaff_amd.read_jpeg_coefficients(f)
        assert np.array_equal(ctx.jpeg_pixels(lay, blob)[0], hesaff_amd.read_image(f)), f
    # damaged files: whatever the host reader makes of them, the device path makes the same (hesaff <file> and hesaff --batch agree)
    raw = open(os.path.join(GOLD, "jpeg_420_q85.jpg"), "rb").read()
    praw = open(os.path.join(GOLD, "jpeg_prog_420_q80.jpg"), "rb").read()
    n_damaged = 0
    for data in (raw, praw):
        for cut in (len(data) // 3, len(data) // 2, len(data) - 3):
            open(p, "wb").write(data[:cut])
            try:
                want = hesaff_amd.read_image(p)
            except hesaff_amd.HesaffError:
                with pytest.raises(hesaff_amd.HesaffError):
                    hesaff_amd.read_jpeg_coefficients(p)
                continue
            lay, blob = hesaff_amd.read_jpeg_coefficients(p)
            assert np.array_equal(ctx.jpeg_pixels(lay, blob)[0], want), cut
            n_damaged += 1
    assert n_damaged >= 3
    # a layout that does not describe the blob is refused, not read
    lay.bw[0] += 1
    with pytest.raises(hesaff_amd.HesaffError):
        ctx.jpeg_pixels(lay, blob)


def test_device_float_print_equals_printf_g(ctx):
    """The device's "%g" (export_fmt.h compiled for gfx950: 128-bit fast path, 256-bit division everywhere else) prints every
    float like snprintf("%g") on the host does - the format of operator<<(ostream&, float), hesaff.cpp:125."""
    import hesaff_amd
    v = _fmt_g_inputs()
    assert hesaff_amd.load_library().hesaff_test_fmt_g(v, len(v)) == 0          # host build of the same header == libc
    text, lens = ctx.fmt_g(v)
    want = np.char.mod("%g", v.astype(np.float64))
    neg_nan = np.isnan(v) & (v.view(np.uint32) >> 31 == 1)                        # numpy prints "nan" for both signs, glibc "-nan"
    want[neg_nan] = "-nan"
    got = text.view("S16").reshape(-1)
    assert lens.min() >= 1 and lens.max() <= 12
    bad = np.flatnonzero(got != want.astype("S16"))
    assert len(bad) == 0, [(float(v[i]), got[i], want[i]) for i in bad[:10]]
    assert np.array_equal(lens, np.char.str_len(want))


def test_device_export_equals_host_writer_and_oracle(ctx, oracle):
    """hesaff_stage_export (the kernels hesaff_process_files formats every chunk with) gives, byte for byte, the file the host
    writer gives and the oracle's exportKeypoints text: golden images, SURVEY App. C's probe (its md5), a 3840x2160 image; the
    sidecar rows equal hesaff_write_bin's."""
    import hashlib
    import json
    import hesaff_amd
    mr = ctx.params.mrSize
    for name in ["band_131x77", "band_96x96", "band_160x120", "tiny_20x15"]:
        img = hesaff_amd.read_pnm(os.path.join(GOLD, name + ".pgm"))
        (_, keys), = ctx.detect_batch([img])
        want = open(os.path.join(GOLD, name + ".hesaff.sift"), "rb").read()
        assert ctx.export(keys) == want == oracle.OracleRun(oracle.gray_from_u8(img)).export_text(), name
    man = json.load(open(os.path.join(GOLD, "manifest.json")))["probe_vga"]
    (_, keys), = ctx.detect_batch([hesaff_amd.read_pnm(os.path.join(GOLD, "probe_vga.pgm"))])
    assert hashlib.md5(ctx.export(keys)).hexdigest() == man["sift_md5"]
    img = band_noise_image(2160, 3840, 1234)
    (_, keys), = ctx.detect_batch([img])
    text = ctx.export(keys)
    assert text == hesaff_amd.format_sift(keys, mr) and len(keys) > 100000 and len(text) > 30_000_000
    assert text == oracle.OracleRun(oracle.gray_from_u8(img)).export_text()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        q = os.path.join(d, "x.bin")
        hesaff_amd.write_bin(q, keys, mr)
        assert ctx.export(keys, fmt=2) == open(q, "rb").read()
        # the writer half: header + rows as they are
        L = hesaff_amd.load_library()
        body = text[text.index(b"\n", 4) + 1:]
        assert L.hesaff_write_sift_rows(os.fsencode(q), body, len(body), len(keys)) == 0 and open(q, "rb").read() == text


def test_device_export_of_arbitrary_records(ctx):
    """Records no image produces - non-finite and denormal coordinates, degenerate shapes, every descriptor byte value, row
    counts around the 64-row blocks of the write kernel - still print like the host writer prints them."""
    import hesaff_amd
    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 65, 127, 4097, 70001):
        keys = np.zeros(n, hesaff_amd.KEYPOINT_DTYPE)
        for f in ("x", "y", "s", "a11", "a12", "a21", "a22"):
            keys[f] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
        half = n // 2
        keys["x"][:half] = rng.uniform(0, 4000, half); keys["y"][:half] = rng.uniform(0, 2200, half); keys["s"][:half] = rng.uniform(0.5, 90, half)
        keys["a11"][:half] = rng.uniform(0.3, 3, half); keys["a12"][:half] = 0; keys["a21"][:half] = rng.uniform(-2, 2, half)
        keys["a22"][:half] = 1.0 / keys["a11"][:half]
        keys["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
        if n > 300:
            keys["desc"][:256] = np.arange(256, dtype=np.uint8)[:, None]
            keys["desc"][256] = 255; keys["desc"][257] = 0
        for mr in (5.196152, 1.0):
            assert ctx.export(keys, mr) == hesaff_amd.format_sift(keys, mr), (n, mr)
        bin_rows = ctx.export(keys, 5.196152, fmt=2)
        assert len(bin_rows) == 16 + 148 * n and bin_rows[:8] == b"HESAFFB1"
        rows = np.frombuffer(bin_rows[16:], dtype=hesaff_amd.BIN_ROW_DTYPE)
        assert np.array_equal(rows["desc"], keys["desc"]) and np.array_equal(rows["x"].view(np.uint32), keys["x"].view(np.uint32))
        if n:
            e = hesaff_amd.ellipse(keys[: min(n, 500)], 5.196152)
            assert np.array_equal(np.stack([rows[k][: len(e)] for k in "abc"], 1).view(np.uint32), e.view(np.uint32))


def test_process_files_isolates_images_the_device_refuses(tmp_path, oracle):
    """one image the device cannot take - a side above 65535 pixels, or more keypoints than max_kpts_per_mpx plans
    for - is reported per file (HESAFF_FILE_REJECTED) and the rest of the list is written."""
    import hesaff_amd
    from tests import _oracle
    paths = []
    imgs = [band_noise_image(120, 160, 500 + i, SMALL_BANDS) for i in range(5)]
    for i, img in enumerate(imgs):
        q = tmp_path / ("g%d.pgm" % i)
        q.write_bytes(b"P5\n160 120\n255\n" + img.tobytes())
        paths.append(str(q))
    wide = tmp_path / "wide.pgm"
    wide.write_bytes(b"P5\n70000 8\n255\n" + bytes(70000 * 8))
    paths.insert(2, str(wide))
    p = hesaff_amd.default_params(); p.max_batch = 2
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        st = ctx.process_files(paths, decode_threads=2, write_threads=2)
    for i, (rc, stage, nh, nd) in enumerate(st):
        if i == 2:
            assert (rc, stage) == (-2, 4)                     # HESAFF_ERR_ARG, HESAFF_FILE_REJECTED
        else:
            assert (rc, stage) == (0, 3), (i, rc, stage)
            img = imgs[i if i < 2 else i - 1]
            assert open(paths[i] + ".hesaff.sift", "rb").read() == _oracle.OracleRun(_oracle.gray_from_u8(img)).export_text()
    # capacity: 1000 keypoints per Mpx planned (4096 at least), a 640x480 dense image has more than 4096 candidates
    dense = tmp_path / "dense.pgm"
    dense.write_bytes(b"P5\n640 480\n255\n" + band_noise_image(480, 640, 9).tobytes())
    paths2 = [paths[0], str(dense), paths[1], paths[3]]
    p = hesaff_amd.default_params(); p.max_batch = 1; p.max_kpts_per_mpx = 1000
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        st = ctx.process_files(paths2, decode_threads=1, write_threads=1)
    assert [(rc, stage) for rc, stage, _, _ in st] == [(0, 3), (-3, 4), (0, 3), (0, 3)]


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 4's workload through the RCCL path and real photographs
# ---------------------------------------------------------------------------------------------------------------
def test_config4_workload_through_rccl_on_one_rank():
    """BASELINE.json config 4 - 2048 images of 3840x2160, image-sharded, RCCL gather of counts - with the one rank this box has:
    `torchrun --nproc-per-node 1 bench.py --gpus 1 --scaling strong --global-images 2048` on the real backend.  The rank owns
    all 2048 images (256 distinct ones, cycled, in library calls of 256): the gathered counts are 8 x the counts of the
    256-image job.  Asking for more ranks than the node has GPUs fails in one line, before anything is started."""
    import json
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("BENCH_DIST_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = {}
    for total in (2048, 256):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong", "--global-images", str(total),
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-path"]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        out[total] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    d = out[2048]
    assert d["n_gpus"] == 1 and d["collective_backend"] == "nccl (RCCL)" and d["scaling"] == "strong"
    assert d["config"]["images_per_step_all_ranks"] == 2048 and d["config"]["per_rank_images_timed"] == [2048]
    assert d["config"]["images_per_library_call"] == 256 and d["config"]["width"] == 3840 and d["config"]["height"] == 2160
    assert d["config"]["descriptors_timed_all_ranks"] == 8 * out[256]["config"]["descriptors_timed_all_ranks"] > 8 * 256 * 100000
    assert d["config"]["hessian_keypoints_timed_all_ranks"] == 8 * out[256]["config"]["hessian_keypoints_timed_all_ranks"]
    # more ranks than GPUs: refused by the parent, nothing is spawned
    import hesaff_amd
    n = hesaff_amd.load_library().hesaff_device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    msg = (r.stderr + r.stdout).strip()
    assert r.returncode != 0 and "visible GPU" in msg and len(msg.splitlines()) == 1, msg[-500:]


def _photo_paths():
    from hesaff_amd.synth import sample_photo_paths
    q = sample_photo_paths()
    if not q:
        pytest.skip("scikit-learn's sample photographs are not installed")
    return q


def test_photographs_native_size_through_cli(tmp_path, oracle):
    """Two real photographs (scikit-learn's china.jpg and flower.jpg, 640x427 colour JPEG) through `hesaff <file>` - in-tree JPEG
    reader, grey conversion of hesaff.cpp:138-148, the whole path - against the oracle on the same decoded pixels (which are
    libjpeg's): every byte of the output file.  README:1-14: photographs are what the reference is for."""
    import hesaff_amd
    from PIL import Image
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    for src in _photo_paths():
        dst = tmp_path / os.path.basename(src)
        shutil.copy(src, dst)
        r = subprocess.run([exe, str(dst)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        pix = hesaff_amd.read_image(str(dst))
        assert pix.shape == (427, 640, 3) and np.array_equal(pix, np.asarray(Image.open(str(dst)).convert("RGB")))
        o = oracle.OracleRun(oracle.gray_from_u8(pix))
        assert o.n_keys > 500
        assert (tmp_path / (os.path.basename(src) + ".hesaff.sift")).read_bytes() == o.export_text()
        m = re.search(r"Detected (\d+) keypoints and (\d+) affine shapes", r.stdout)
        assert m and (int(m.group(1)), int(m.group(2))) == (o.n_hessian, o.n_keys)


def test_photograph_mosaic_4k_against_the_oracle(ctx, oracle):
    """A 3840x2160 colour mosaic of the two photographs (flipped tiles, hesaff_amd/synth.py) through hesaff_detect_batch beside a
    second mosaic and a dense band-noise image of the same size: every field and descriptor byte of the photograph mosaic
    equals the oracle's, the window sizes have the distribution of photographs (median P0 about 31)."""
    from hesaff_amd.synth import load_sample_photos, photo_mosaic
    _photo_paths()
    photos = load_sample_photos()
    m0 = photo_mosaic(2160, 3840, 0, photos)
    m1 = photo_mosaic(2160, 3840, 1, photos)
    res = ctx.detect_batch([m0, m1])
    (n0, k0), (n1, k1) = res
    o = oracle.OracleRun(oracle.gray_from_u8(m0))
    _assert_keys_equal_oracle(k0, n0, o, "photograph mosaic")
    assert 15000 < len(k0) < 80000 and 15000 < len(k1) < 80000 and k0.tobytes() != k1.tobytes()
    assert ctx.export(k0) == o.export_text()
    P0 = 2 * np.ceil(k0["s"] * np.float32(ctx.params.mrSize)).astype(np.int64) + 1
    assert 25 <= np.median(P0) <= 45 and P0.max() > 150, (float(np.median(P0)), int(P0.max()))


def test_process_files_resume_skips_complete_outputs(tmp_path, oracle):
    """hesaff_set_resume (checkpoint / resume): a second run over the same list reads no image whose complete
    output exists, re-does the ones whose output is missing or torn, and leaves identical files; outputs are written under a
    temporary name and renamed, so nothing but complete files ever carries the final name."""
    import hesaff_amd
    from tests import _oracle
    paths, texts = [], []
    for i in range(6):
        img = band_noise_image(120, 160, 700 + i, SMALL_BANDS)
        if i == 4:
            img = np.full((120, 160), 77, np.uint8)      # featureless: "128\n0\n" (a chunk whose text is empty)
        q = tmp_path / ("r%d.pgm" % i)
        q.write_bytes(b"P5\n160 120\n255\n" + img.tobytes())
        paths.append(str(q)); texts.append(_oracle.OracleRun(_oracle.gray_from_u8(img)).export_text())
    p = hesaff_amd.default_params(); p.max_batch = 2
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        ctx.set_output_format(3)
        ctx.set_resume(True)
        st = ctx.process_files(paths)
        assert all((rc, stage) == (0, 3) for rc, stage, _, _ in st)
        assert not [f for f in os.listdir(tmp_path) if f.endswith(".part")]
        os.remove(paths[1] + ".hesaff.sift")                                              # missing
        open(paths[3] + ".hesaff.sift", "wb").write(texts[3][: len(texts[3]) // 2])       # torn: no final newline
        open(paths[4] + ".hesaff.bin", "ab").write(b"\0")                                 # size does not match the row count
        for q in (paths[0], paths[2], paths[5]):
            os.remove(q)                                                                  # complete outputs: the image is not even opened
        st = ctx.process_files(paths)
        for i, (rc, stage, nh, nd) in enumerate(st):
            if i in (0, 2, 5):
                assert (rc, stage, nh) == (0, 5, -1) and nd == int(texts[i].split(b"\n")[1]), (i, rc, stage, nh, nd)   # HESAFF_FILE_SKIPPED
            else:
                assert (rc, stage) == (0, 3) and (nh > 0 or i == 4), (i, rc, stage)
        assert texts[4] == b"128\n0\n"
        for q, t in zip(paths, texts):
            assert open(q + ".hesaff.sift", "rb").read() == t
            assert hesaff_amd.load_library().hesaff_output_is_complete(os.fsencode(q + ".hesaff.bin"), 2) == int(t.split(b"\n")[1])
        ctx.set_resume(False)
        st = ctx.process_files(paths[1:2])
        assert st[0][1] == 3


@pytest.mark.gpu
def test_budgeted_file_path_writes_the_same_bytes(tmp_path, oracle):
    """the file path confined to one device's share of the host - bench.py's `--budgeted-child` (CPU mask set before
    anything is loaded, thread counts from hesaff_host_plan_for) and `hesaff --batch ... --host-share 8` under the same mask - writes
    the bytes the unconfined run writes (which are the oracle's): neither the thread counts nor the pinned read buffers change a file."""
    import hashlib
    import json
    import sys
    import hesaff_amd
    from tests import _oracle
    n, W, H = 20, 320, 240
    d = tmp_path / "files"; d.mkdir()
    paths = []
    for i in range(n):
        img = band_noise_image(H, W, 900 + i, SMALL_BANDS)
        q = d / ("img%04d.pgm" % i)
        q.write_bytes(b"P5\n%d %d\n255\n" % (W, H) + img.tobytes())
        paths.append(str(q))
    want0 = _oracle.OracleRun(_oracle.gray_from_u8(band_noise_image(H, W, 900, SMALL_BANDS))).export_text()
    p = hesaff_amd.default_params(); p.max_batch = 4
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        ctx.set_output_format(3)
        st = ctx.process_files(paths)          # the library's "auto" plan, twice: the second run reads into the recycled pinned buffers
        st = ctx.process_files(paths)
    assert all((rc, stage) == (0, 3) for rc, stage, _, _ in st)
    text = [open(q + ".hesaff.sift", "rb").read() for q in paths]
    side = [open(q + ".hesaff.bin", "rb").read() for q in paths]
    assert text[0] == want0
    for q in paths:
        os.remove(q + ".hesaff.sift"); os.remove(q + ".hesaff.bin")
    cpus = sorted(os.sched_getaffinity(0))[:2]
    cfg = {"dir": str(d), "n": n, "chunk": 4, "device": 0, "cpus": cpus, "rank": 0, "world": 1, "md5": True}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--budgeted-child", json.dumps(cfg)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["cpus"] == cpus and out["plan"]["cpus"] <= 2 and out["plan"]["decode_threads"] + out["plan"]["write_threads"] == 2
    assert out["text"]["md5"] == [hashlib.md5(t).hexdigest() for t in text] and out["text"]["failed_files"] == 0
    assert out["sidecar"]["md5"] == [hashlib.md5(t).hexdigest() for t in side]
    assert out["text"]["cpu_seconds_per_image"] > 0
    # the CLI with the same share
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    lst = tmp_path / "list.txt"; lst.write_text("\n".join(paths) + "\n")
    r = subprocess.run([exe, "--batch", str(lst), "--output", "both", "--host-share", "8"], capture_output=True, text=True,
                       preexec_fn=lambda: os.sched_setaffinity(0, set(cpus)))
    assert r.returncode == 0, r.stderr
    assert [open(q + ".hesaff.sift", "rb").read() for q in paths] == text and [open(q + ".hesaff.bin", "rb").read() for q in paths] == side
    assert subprocess.run([exe, "--batch", str(lst), "--host-share", "0"], capture_output=True, text=True).returncode == 1


@pytest.mark.gpu
def test_graf_layout_directory_in_one_command(tmp_path, oracle):
    """BASELINE.json config 5: `tools/repeatability.py --graf DIR` on a directory in the layout of the Oxford sequences -
    img1..img4.ppm (the format graf ships in) and H1to2p..H1to4p - runs `hesaff --batch` over it and prints the table.  Every output equals
    the oracle's for the file's pixels; the table equals what evaluate() gives on those files; --devices is passed through."""
    import json
    import sys
    import hesaff_amd
    from tests import _oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import repeatability as rp
    seq = tmp_path / "jpg"
    angles = (15, 30, 45)
    paths, Hs = rp.write_sequence_files(str(seq), 480, 400, angles)
    d = tmp_path / "graf"; d.mkdir()
    for k, q in enumerate(paths):      # the same pixels as binary PPM, named like the Oxford sets
        pix = hesaff_amd.read_image(q)
        (d / ("img%d.ppm" % (k + 1))).write_bytes(b"P6\n%d %d\n255\n" % (pix.shape[1], pix.shape[0]) + pix.tobytes())
        if k > 0:
            np.savetxt(str(d / ("H1to%dp" % (k + 1))), Hs[k])
    (d / "readme.txt").write_text("not an image")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "repeatability.py"), "--graf", str(d), "--devices", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout)
    assert [e["pair"] for e in res["pairs"]] == ["img1 -> img2", "img1 -> img3", "img1 -> img4"] and "--devices 0" in res["command"]
    assert "repeatability" in r.stderr and "img1 -> img4" in r.stderr        # the table
    for k in range(4):
        q = str(d / ("img%d.ppm" % (k + 1)))
        o = _oracle.OracleRun(_oracle.gray_from_u8(hesaff_amd.read_image(q)))
        assert open(q + ".hesaff.sift", "rb").read() == o.export_text() and o.n_keys > 500
    r1, d1 = rp.read_sift(str(d / "img1.ppm.hesaff.sift"))
    for k, e in enumerate(res["pairs"]):
        r2, d2 = rp.read_sift(str(d / ("img%d.ppm.hesaff.sift" % (k + 2))))
        want = rp.evaluate(r1, d1, r2, d2, Hs[k + 1], (480, 400), (480, 400))
        assert all(e[key] == want[key] for key in want), (e, want)
    assert res["pairs"][0]["repeatability"] > res["pairs"][2]["repeatability"] > 0.2
    # a directory without homographies / without img1 is refused with a message, not a traceback of the CLI
    os.remove(str(d / "H1to3p"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "repeatability.py"), "--graf", str(d)], capture_output=True, text=True)
    assert r.returncode != 0 and "H1to3p" in r.stderr


@pytest.mark.gpu
def test_the_caller_sleeps_while_the_device_works():
    """Round 5: hipEventSynchronize spins in this runtime even on hipEventBlockingSync events - one busy core per context while a batch's kernels
    run.  The library waits with hipEventQuery + nanosleep (hs_wait_event, pipeline.hip): the calling thread's own CPU time
    (CLOCK_THREAD_CPUTIME_ID) inside a batch call must be a small part of the call's wall time."""
    import time
    import hesaff_amd
    imgs = [band_noise_image(1080, 1920, 4000 + i) for i in range(12)]
    p = hesaff_amd.default_params(); p.max_batch = len(imgs)
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        ctx.detect_batch_raw(imgs)                      # plan, buffers, pinned blocks
        c0, t0 = time.thread_time(), time.perf_counter()
        for _ in range(10):
            res = ctx.detect_batch_raw(imgs)
        cpu, wall = time.thread_time() - c0, time.perf_counter() - t0
    assert sum(r.count_desc for r in res) > 100000
    assert wall > 0.02 and cpu < 0.5 * wall, (cpu, wall)    # a spinning wait measures cpu ~= wall (0.96-1.0 before the change; 0.06 after)


@pytest.mark.gpu
def test_octave_map_epochs_wrap_around(oracle):
    """The order-key map (octaveMap, pyramid.cpp:189-193,226) is never reset between octaves or batches: every pass bids with keys of a fresh,
    smaller epoch, and the map is refilled when the epochs run out - after 127 passes for a 3840 x 2160 image (25 key bits), i.e. in the 16th batch
    of 8 octaves.  Forty batches through one context cross that point twice: every batch must return the first one's bytes (which are the oracle's)."""
    import hesaff_amd
    from tests import _oracle
    img = band_noise_image(2160, 3840, 77)
    p = hesaff_amd.default_params(); p.max_batch = 1
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        nh0, first = ctx.detect_batch([img])[0]
        for it in range(40):
            nh, again = ctx.detect_batch([img])[0]
            assert nh == nh0 and len(again) == len(first) and again.tobytes() == first.tobytes(), it
        small = band_noise_image(480, 640, 78, SMALL_BANDS)      # another geometry: another key width, the map starts over
        nh_s, got = ctx.detect_batch([small])[0]
        nh_b, back = ctx.detect_batch([img])[0]
        assert nh_b == nh0 and back.tobytes() == first.tobytes()
    o = _oracle.OracleRun(_oracle.gray_from_u8(small))
    assert nh_s == o.n_hessian and len(got) == o.n_keys and hesaff_amd.format_sift(got, p.mrSize) == o.export_text()
    assert len(first) > 100000
    # ... and the 3840 x 2160 bytes every batch returned ARE the oracle's (one oracle run, about 15 s)
    ob = _oracle.OracleRun(_oracle.gray_from_u8(img))
    g, t, d = ob.keys()
    assert nh0 == ob.n_hessian and len(first) == ob.n_keys
    assert np.array_equal(first["desc"], d) and np.array_equal(first["type"], t)
    for j, name in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
        assert_bit_equal(first[name], g[:, j], name)


@pytest.mark.gpu
def test_image_and_window_beyond_the_old_size_ceiling(oracle):
    """the reference has no bound on the window normalizeAffine warps (affine.cpp:120-124 resizes its workspace per call);
    the library refused every image with sqrt(W H) above ~7000 at plan time (LDS of the large-window row kernel sized for the largest window
    the image could hold).  A 12000 x 8000 image (sqrt = 9798) with one blob whose window is 7359 pixels a side - above the 6912 that four
    wavefronts' rows fit in a CU's LDS, so k_patch_large_rows runs with two per block - plus windows in the other size bins: every
    field and descriptor equal to the oracle's."""
    import hesaff_amd
    H, W = 8000, 12000
    yy = np.arange(H, dtype=np.float32)[:, None]; xx = np.arange(W, dtype=np.float32)[None, :]
    acc = np.full((H, W), 40.0, np.float32)
    for cy, cx, sy, sx, amp in ((H * 0.5, W * 0.5, 700.0, 700.0, 180.0), (H * 0.2, W * 0.15, 150.0, 150.0, 120.0), (H * 0.8, W * 0.85, 60.0, 90.0, 100.0),
                                (H * 0.25, W * 0.8, 20.0, 20.0, 150.0)):
        acc += np.float32(amp) * np.exp(-0.5 * ((yy - np.float32(cy)) / np.float32(sy)) ** 2) * np.exp(-0.5 * ((xx - np.float32(cx)) / np.float32(sx)) ** 2)
    img = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    del acc
    p = hesaff_amd.default_params(); p.max_batch = 1
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        (nh, keys), = ctx.detect_batch([img])
    o = oracle.OracleRun(oracle.gray_from_u8(img))
    g, t, d = o.keys()
    assert nh == o.n_hessian and len(keys) == o.n_keys and o.n_keys >= 4
    P0 = 2 * np.ceil(g[:, 2] * np.float32(p.mrSize)).astype(int) + 1
    assert P0.max() + 2 > 6912 and (P0 > 512).sum() >= 2, P0
    assert np.array_equal(keys["desc"], d) and np.array_equal(keys["type"], t)
    for j, name in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
        assert_bit_equal(keys[name], g[:, j], name)


@pytest.mark.gpu
def test_large_window_forms_in_one_batch(oracle):
    """k_patch_large_rows runs the windows of a batch in two forms (round 6): up to 1280 pixels a side three window rows per wavefront
    step, larger ones one row per step in a launch of their own, both sized from the batch's largest window (affine.cpp:114-135 for P > 512).
    Two 3840 x 2160 images of elongated Gaussian blobs: windows of 579 .. 1205 pixels in the first, 1519 and 1861 in the second - as one
    batch (split launches), then the first image alone (one launch): every field and descriptor equals the oracle's."""
    import hesaff_amd
    H, W = 2160, 3840
    yy = np.arange(H, dtype=np.float32)[:, None]; xx = np.arange(W, dtype=np.float32)[None, :]

    def blobs(spec):
        acc = np.full((H, W), 30.0, np.float32)
        for sg, cy, cx in spec:
            acc += np.float32(150) * np.exp(-0.5 * ((yy - cy) / np.float32(sg)) ** 2) * np.exp(-0.5 * ((xx - cx) / np.float32(sg * 1.15)) ** 2)
        return np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    a = blobs(zip((52, 60, 70, 85, 100, 112, 122, 135, 150, 175),
                  *zip((400, 450), (400, 1300), (400, 2200), (450, 3200), (1150, 700), (1150, 1900), (1200, 3100), (1750, 600), (1700, 1800), (1650, 3000))))
    b = blobs([(165, 1080, 1000), (135, 1080, 2900), (60, 300, 400), (75, 300, 1900), (90, 1850, 1900), (50, 1900, 3500)])
    p = hesaff_amd.default_params(); p.max_batch = 2
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        both = ctx.detect_batch([a, b])
        alone = ctx.detect_batch([a])
    sizes = []
    for img, (nh, keys) in ((a, both[0]), (b, both[1]), (a, alone[0])):
        o = oracle.OracleRun(oracle.gray_from_u8(img))
        g, t, d = o.keys()
        assert nh == o.n_hessian and len(keys) == o.n_keys and o.n_keys >= 2
        assert np.array_equal(keys["desc"], d) and np.array_equal(keys["type"], t)
        for j, name in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
            assert_bit_equal(keys[name], g[:, j], name)
        sizes.append(2 * np.ceil(g[:, 2] * np.float32(p.mrSize)).astype(int) + 3)
    assert (sizes[0] > 512).sum() >= 10 and sizes[0].max() <= 1280 and sizes[1].min() > 1280, sizes
