"""CPU suite: host-side pieces of the product (tables, PNM reader, .hesaff.sift writer,
pinned libm restatements) against the oracle / libm / golden files.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import hesaff_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_tables_bit_identical_to_oracle(oracle):
    L = oracle.lib()
    m = np.zeros((19, 19), np.float32); L.ho_gauss_mask(19, m.reshape(-1))
    assert np.array_equal(hesaff_amd.table_gauss_mask(19).view(np.uint32), m.view(np.uint32))
    m = np.zeros((41, 41), np.float32); L.ho_circ_gauss_mask(41, m.reshape(-1))
    assert np.array_equal(hesaff_amd.table_circ_gauss_mask(41).view(np.uint32), m.view(np.uint32))
    b0 = np.zeros(41, np.int32); b1 = np.zeros(41, np.int32); w0 = np.zeros(41, np.float32); w1 = np.zeros(41, np.float32)
    L.ho_sift_tables(b0, b1, w0, w1)
    pb0, pb1, pw0, pw1 = hesaff_amd.table_sift_bins()
    assert np.array_equal(pb0, b0) and np.array_equal(pb1, b1) and np.array_equal(pw0, w0) and np.array_equal(pw1, w1)
    for sigma in [0.62, 0.7, 0.8, 1.2262737, 1.5198685, 1.545008, 1.946588, 2.4525473, 5.0, 33.3]:
        k = L.ho_gauss_ksize(sigma)
        ref = np.zeros(k, np.float32); L.ho_gauss_kernel(k, sigma, ref)
        got = hesaff_amd.table_gauss_kernel(sigma)
        assert len(got) == k and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), sigma


def test_pyramid_kernel_sizes_match_survey():
    # K = 11 (initial), 9, 11, 13, 15 for the per-octave blurs
    ks = [len(hesaff_amd.table_gauss_kernel(s)) for s in (1.5198685, 1.2262737, 1.545008, 1.946588, 2.4525473)]
    assert ks == [11, 9, 11, 13, 15]


@pytest.fixture(scope="module")
def hmath_host(tmp_path_factory):
    d = tmp_path_factory.mktemp("hm")
    src = d / "hm.cpp"
    src.write_text('#include "%s/hesaff_amd/csrc/hmath.h"\n'
                   'extern "C" void hm_atan2f_v(int n,const float*y,const float*x,float*o){for(int i=0;i<n;i++)o[i]=hm_atan2f(y[i],x[i]);}\n'
                   'extern "C" void hm_pow2f_v(int n,const float*y,float*o){for(int i=0;i<n;i++)o[i]=hm_pow2f(y[i]);}\n'
                   'extern "C" void hm_orient_v(int n,const float*y,float*o){for(int i=0;i<n;i++)o[i]=hm_sift_orient_coord(y[i]);}\n' % ROOT)
    so = d / "hm.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)])
    return C.CDLL(str(so))


def test_hmath_restatements_equal_glibc(hmath_host):
    """hm_atan2f == atan2f and hm_pow2f == powf(2,.) of this image's glibc, bit for bit."""
    libm = C.CDLL("libm.so.6")
    libm.atan2f.restype = C.c_float; libm.atan2f.argtypes = [C.c_float, C.c_float]
    libm.powf.restype = C.c_float; libm.powf.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(1)
    n = 400000
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    hmath_host.hm_atan2f_v.argtypes = [C.c_int, f32p, f32p, f32p]
    hmath_host.hm_pow2f_v.argtypes = [C.c_int, f32p, f32p]
    # gradient-like operands (differences of values in 0..255) and random bit patterns
    y = np.concatenate([(rng.standard_normal(n // 2) * 40).astype(np.float32), rng.integers(0, 2**32, n // 2, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    x = np.concatenate([(rng.standard_normal(n // 2) * 40).astype(np.float32), rng.integers(0, 2**32, n // 2, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    y[:2000] = 0; x[2000:4000] = 0; x[4000:6000] = 1.0; y[6000:8000] = -0.0
    got = np.zeros(n, np.float32)
    hmath_host.hm_atan2f_v(n, y, x, got)
    sub = 60000
    ref = np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y[:sub], x[:sub])] +
                   [libm.atan2f(float(a), float(b)) for a, b in zip(y[-sub:], x[-sub:])], np.float32)
    g2 = np.concatenate([got[:sub], got[-sub:]])
    same = (g2.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(g2) & np.isnan(ref))
    assert same.all(), int((~same).sum())
    e = rng.uniform(-1.0, 1.0, 100000).astype(np.float32)
    e[:3] = [1.0 / 3.0, 0.0, -0.5]
    gp = np.zeros_like(e)
    hmath_host.hm_pow2f_v(len(e), e, gp)
    rp = np.array([libm.powf(2.0, float(a)) for a in e], np.float32)
    assert np.array_equal(gp.view(np.uint32), rp.view(np.uint32))
    # siftdesc.cpp:65 in double, as numpy evaluates it (IEEE add / mul / div)
    hmath_host.hm_orient_v.argtypes = [C.c_int, f32p, f32p]
    ori = np.concatenate([rng.uniform(-np.pi, np.pi, 500000), [0.0, -0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2]]).astype(np.float32)
    go = np.zeros_like(ori)
    hmath_host.hm_orient_v(len(ori), ori, go)
    ro = (np.float64(8.0) * (ori.astype(np.float64) + 2 * np.pi) / (2 * np.pi)).astype(np.float32)
    assert np.array_equal(go.view(np.uint32), ro.view(np.uint32))


def test_read_pnm_and_grey_conversion(tmp_path, oracle):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (13, 17), dtype=np.uint8)
    p = tmp_path / "a.pgm"
    p.write_bytes(b"P5\n# comment\n17 13\n255\n" + g.tobytes())
    assert np.array_equal(hesaff_amd.read_pnm(str(p)), g)
    c = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    p = tmp_path / "b.ppm"
    p.write_bytes(b"P6 7 5 255\n" + c.tobytes())
    assert np.array_equal(hesaff_amd.read_pnm(str(p)), c)
    with pytest.raises(hesaff_amd.HesaffError):
        hesaff_amd.read_pnm(str(tmp_path / "missing.pgm"))
    # grey of a grey image is the byte value exactly (hesaff.cpp:145 with B=G=R)
    assert np.array_equal(oracle.gray_from_u8(g), g.astype(np.float32))


def _keys_from_text(txt):
    lines = txt.decode().split("\n")
    assert lines[0] == "128"
    n = int(lines[1])
    rows = [l.split(" ") for l in lines[2:2 + n]]
    return n, rows


@pytest.mark.parametrize("name", ["band_131x77", "band_96x96", "band_160x120", "tiny_20x15", "thin_12x40"])
def test_writer_reproduces_golden_files(oracle, name):
    """hesaff_format_sift (product) on the oracle's keypoints == golden .hesaff.sift bytes."""
    img = hesaff_amd.read_pnm(os.path.join(GOLD, name + ".pgm"))
    o = oracle.OracleRun(oracle.gray_from_u8(img))
    g, t, d = o.keys()
    keys = np.zeros(len(g), hesaff_amd.KEYPOINT_DTYPE)
    for j, f in enumerate(["x", "y", "s", "a11", "a12", "a21", "a22", "response"]):
        keys[f] = g[:, j]
    keys["type"] = t; keys["desc"] = d
    txt = hesaff_amd.format_sift(keys, hesaff_amd.default_params().mrSize)
    gold = open(os.path.join(GOLD, name + ".hesaff.sift"), "rb").read()
    assert txt == gold
    n, rows = _keys_from_text(txt)
    assert n == len(keys) and all(len(r) == 5 + 128 for r in rows)


def test_write_sift_file(tmp_path):
    keys = np.zeros(2, hesaff_amd.KEYPOINT_DTYPE)
    keys["x"] = [10.5, 123456.7]; keys["y"] = [3.25, 0.000123]; keys["s"] = [2.0, 3.0]
    keys["a11"] = [1.0, 2.0]; keys["a21"] = [0.0, 0.3]; keys["a22"] = [1.0, 0.5]
    keys["desc"][0, :3] = [0, 9, 255]; keys["desc"][1, 127] = 100
    p = tmp_path / "o.sift"
    hesaff_amd.write_sift(str(p), keys, 5.196152)
    lines = p.read_text().split("\n")
    assert lines[0] == "128" and lines[1] == "2" and lines[4] == ""
    r0 = lines[2].split(" ")
    assert r0[0] == "10.5" and r0[1] == "3.25" and r0[5:8] == ["0", "9", "255"]
    r1 = lines[3].split(" ")
    assert r1[0] == "123457" and r1[1] == "0.000123"      # 6 significant digits like operator<<(float)
    # ellipse of the identity shape: a = c = 1/(mrSize*s)^2, b = 0
    e = hesaff_amd.ellipse(keys[:1], 5.196152)[0]
    assert abs(e[0] - 1.0 / (5.196152 * 2.0) ** 2) < 1e-9 and e[1] == 0 and abs(e[2] - e[0]) < 1e-12


def test_fast_float_formatter_equals_printf_g():
    """The writer's own "%g" (six significant digits from one extended-precision scaling, libc on
    near-ties) produces snprintf's bytes: every kind of float the file can hold, by the million."""
    import ctypes as C
    L = hesaff_amd.load_library()
    rng = np.random.default_rng(7)
    parts = [
        rng.integers(0, 2**32, 3_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32),          # any bit pattern
        (rng.uniform(0, 4096, 2_000_000)).astype(np.float32),                                             # coordinates
        (10.0 ** rng.uniform(-8, 2, 2_000_000) * rng.choice([-1.0, 1.0], 2_000_000)).astype(np.float32),  # ellipse terms
        np.array([0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 999999.5, 999999.4, 999999.96, 1e-5, 9.9999995e-5, 1e-4, 100000.0,
                  123456.5, 1234565.0, 0.000123456789, 3.4028235e38, 1.17549435e-38, 1e-45, 5e-324, np.inf, -np.inf, np.nan,
                  2.5, 0.125, 1048576.0, 8388608.0, 16777216.0, 0.3, 1e10, 1e-10, 65504.0, 1e22, 1e-22, 1e23, 9.5e-23], np.float32),
        (np.arange(0, 2_000_000, dtype=np.float32) + 0.5) / np.float32(8.0),                               # exact binary ties .x5
        (np.arange(1, 1_000_001, dtype=np.float64) * 1e-6 + 0.5e-6).astype(np.float32),                   # decimal near-ties
    ]
    v = np.ascontiguousarray(np.concatenate(parts), np.float32)
    assert L.hesaff_test_fmt_g(v, len(v)) == 0


def test_multithreaded_writer_same_bytes(tmp_path):
    rng = np.random.default_rng(11)
    n = 30000
    keys = np.zeros(n, hesaff_amd.KEYPOINT_DTYPE)
    keys["x"] = rng.uniform(0, 3840, n); keys["y"] = rng.uniform(0, 2160, n); keys["s"] = rng.uniform(1, 30, n)
    keys["a11"] = rng.uniform(0.5, 2, n); keys["a21"] = rng.uniform(-1, 1, n); keys["a22"] = 1.0 / keys["a11"]
    keys["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    mr = hesaff_amd.default_params().mrSize
    ref = hesaff_amd.format_sift(keys, mr)
    for t in (0, 2, 5, 16):
        assert hesaff_amd.format_sift_mt(keys, mr, t) == ref, t
    assert hesaff_amd.format_sift_mt(keys[:0], mr, 4) == b"128\n0\n"
    # batch writer: one file per image, any thread count, same bytes as the single writer
    chunks = [keys[:7000], keys[7000:7000], keys[7000:25000], keys[25000:]]
    paths = [str(tmp_path / ("img%d.ppm.hesaff.sift" % i)) for i in range(len(chunks))]
    hesaff_amd.write_sift_batch(paths, chunks, mr, threads=3)
    for p, c in zip(paths, chunks):
        assert open(p, "rb").read() == hesaff_amd.format_sift(c, mr)


def _png_bytes(pix, ctype, depth=8, filters=(0, 1, 2, 3, 4), palette=None, idat_split=3):
    """Minimal PNG writer for the decoder tests: pix = uint8/uint16 array [H, W, samples]; every row
    is encoded with the next filter type of `filters`."""
    import struct, zlib
    H, W, nch = pix.shape
    if depth == 16:
        rows = [pix[y].astype(">u2").tobytes() for y in range(H)]
    elif depth == 8:
        rows = [pix[y].astype(np.uint8).tobytes() for y in range(H)]
    else:   # packed samples, most significant bits first
        rows = []
        for y in range(H):
            bits = "".join(format(int(v), "0%db" % depth) for v in pix[y].ravel())
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    bpp = max(1, nch * depth // 8)
    raw = bytearray(); prev = bytes(len(rows[0]))
    for y, cur in enumerate(rows):
        ft = filters[y % len(filters)]
        out = bytearray(len(cur))
        for i, v in enumerate(cur):
            a = cur[i - bpp] if i >= bpp else 0; b = prev[i]; c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) >> 1
            else:
                q = a + b - c; pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (v - p) & 255
        raw.append(ft); raw += out; prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    z = zlib.compress(bytes(raw), 6)
    cut = [len(z) * i // idat_split for i in range(idat_split + 1)]
    b = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 0))
    b += chunk(b"gAMA", struct.pack(">I", 45455))
    if palette is not None:
        b += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    for i in range(idat_split):
        b += chunk(b"IDAT", z[cut[i]:cut[i + 1]])
    return b + chunk(b"IEND", b"")


def test_read_png_matches_imread_semantics(tmp_path):
    """hesaff_read_png: every filter type, grey / RGB / alpha / palette / 16-bit / packed grey, split IDAT."""
    rng = np.random.default_rng(17)
    H, W = 23, 37

    def roundtrip(name, data):
        p = tmp_path / name
        p.write_bytes(data)
        return hesaff_amd.read_image(str(p))

    g = rng.integers(0, 256, (H, W, 1), dtype=np.uint8)
    assert np.array_equal(roundtrip("g8.png", _png_bytes(g, 0)), g[:, :, 0])
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(roundtrip("rgb.png", _png_bytes(rgb, 2)), rgb)
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    assert np.array_equal(roundtrip("rgba.png", _png_bytes(rgba, 6, filters=(4, 3, 1))), rgba[:, :, :3])   # alpha dropped
    ga = rng.integers(0, 256, (H, W, 2), dtype=np.uint8)
    assert np.array_equal(roundtrip("ga.png", _png_bytes(ga, 4)), ga[:, :, 0])
    g16 = rng.integers(0, 65536, (H, W, 1), dtype=np.uint16)
    assert np.array_equal(roundtrip("g16.png", _png_bytes(g16, 0, depth=16)), (g16[:, :, 0] >> 8).astype(np.uint8))   # high byte
    rgb16 = rng.integers(0, 65536, (H, W, 3), dtype=np.uint16)
    assert np.array_equal(roundtrip("rgb16.png", _png_bytes(rgb16, 2, depth=16, filters=(3, 4))), (rgb16 >> 8).astype(np.uint8))
    for depth, scale in ((1, 255), (2, 85), (4, 17)):
        gp = rng.integers(0, 1 << depth, (H, W, 1), dtype=np.uint8)
        assert np.array_equal(roundtrip("g%d.png" % depth, _png_bytes(gp, 0, depth=depth, filters=(0, 2))), gp[:, :, 0] * scale)
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (H, W, 1), dtype=np.uint8)
    assert np.array_equal(roundtrip("pal8.png", _png_bytes(idx, 3, palette=pal)), pal[idx[:, :, 0]])
    assert np.array_equal(roundtrip("pal4.png", _png_bytes(idx, 3, depth=4, palette=pal, filters=(0,))), pal[idx[:, :, 0]])
    # PNM still goes through the same entry point; damaged files are errors, not garbage
    (tmp_path / "a.pgm").write_bytes(b"P5\n%d %d\n255\n" % (W, H) + g.tobytes())
    assert np.array_equal(hesaff_amd.read_image(str(tmp_path / "a.pgm")), g[:, :, 0])
    bad = bytearray(_png_bytes(g, 0)); bad[60] ^= 0x40
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    with pytest.raises(hesaff_amd.HesaffError):
        hesaff_amd.read_image(str(tmp_path / "bad.png"))
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0" + bytes(100))
    with pytest.raises(hesaff_amd.HesaffError):
        hesaff_amd.read_image(str(tmp_path / "x.jpg"))


def _adam7_png_bytes(pix, ctype, depth=8, palette=None):
    """An Adam7-interlaced PNG of pix [H, W, samples] (Pillow cannot write one): each of the seven passes is the sub-image
    of the PNG specification's section 8.2, filtered and packed like an image of its own."""
    import struct, zlib
    H, W, nch = pix.shape
    raw = bytearray()
    for xs, ys, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = pix[ys::dy, xs::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        # the filtered rows of this pass = the raw stream of a stand-alone (non-interlaced) PNG of the sub-image
        one = _png_bytes(sub, ctype, depth=depth, palette=palette, idat_split=1)
        i = one.index(b"IDAT")
        n = struct.unpack(">I", one[i - 4:i])[0]
        raw += zlib.decompress(one[i + 4:i + 4 + n])

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    b = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1))
    if palette is not None:
        b += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    z = zlib.compress(bytes(raw), 6)
    return b + chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")


def test_read_png_adam7_interlaced(tmp_path):
    """cv::imread reads interlaced PNG files (hesaff.cpp:137); the pixels are lossless, so the in-tree reader must return what
    Pillow (libpng) returns and what was encoded: sizes around the 8 x 8 Adam7 cell (empty passes), every colour type and depth."""
    from PIL import Image
    rng = np.random.default_rng(23)
    for (H, W) in ((1, 1), (2, 3), (5, 4), (8, 8), (9, 17), (23, 37), (64, 50)):
        cases = [("g8", rng.integers(0, 256, (H, W, 1), dtype=np.uint8), 0, 8, None),
                 ("rgb", rng.integers(0, 256, (H, W, 3), dtype=np.uint8), 2, 8, None),
                 ("rgba", rng.integers(0, 256, (H, W, 4), dtype=np.uint8), 6, 8, None),
                 ("g16", rng.integers(0, 65536, (H, W, 1), dtype=np.uint16), 0, 16, None),
                 ("g2", rng.integers(0, 4, (H, W, 1), dtype=np.uint8), 0, 2, None),
                 ("g1", rng.integers(0, 2, (H, W, 1), dtype=np.uint8), 0, 1, None),
                 ("pal4", rng.integers(0, 16, (H, W, 1), dtype=np.uint8), 3, 4, rng.integers(0, 256, (16, 3), dtype=np.uint8))]
        for name, pix, ctype, depth, pal in cases:
            q = tmp_path / ("%s_%dx%d.png" % (name, W, H))
            q.write_bytes(_adam7_png_bytes(pix, ctype, depth, pal))
            got = hesaff_amd.read_image(str(q))
            if ctype == 3:
                want = pal[pix[:, :, 0]]
            elif depth == 16:
                want = (pix[:, :, 0] >> 8).astype(np.uint8)
            elif depth < 8:
                want = pix[:, :, 0] * (255 // ((1 << depth) - 1))
            elif ctype == 0:
                want = pix[:, :, 0]
            else:
                want = pix[:, :, :3]
            assert np.array_equal(got, want), (name, H, W)
            if depth == 8 and ctype in (0, 2):     # ... and libpng agrees that this is what the file holds
                assert np.array_equal(np.asarray(Image.open(str(q))), want), (name, H, W)
    # a truncated pass stream is an error
    bad = _adam7_png_bytes(rng.integers(0, 256, (9, 9, 1), dtype=np.uint8), 0)
    (tmp_path / "cut.png").write_bytes(bad[:-40] + bad[-12:])
    with pytest.raises(hesaff_amd.HesaffError):
        hesaff_amd.read_image(str(tmp_path / "cut.png"))


def test_read_pnm_every_form_imread_reads(tmp_path):
    """hesaff_read_pnm follows OpenCV's PxM decoder (cv::imread, hesaff.cpp:137): plain and binary PBM / PGM / PPM, any maxval."""
    from PIL import Image
    rng = np.random.default_rng(29)
    H, W = 7, 11

    def rd(name, data):
        q = tmp_path / name
        q.write_bytes(data)
        return hesaff_amd.read_image(str(q)), str(q)

    g = rng.integers(0, 256, (H, W), dtype=np.uint8)
    c = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    # plain forms at maxval 255 = the binary forms, = Pillow
    got, q = rd("p2.pgm", b"P2\n# made by hand\n%d %d\n255\n" % (W, H) + b"\n".join(b" ".join(b"%d" % v for v in row) for row in g) + b"\n")
    assert np.array_equal(got, g) and np.array_equal(np.asarray(Image.open(q)), g)
    got, q = rd("p3.ppm", b"P3 %d %d 255 " % (W, H) + b" ".join(b"%d" % v for v in c.ravel()))
    assert np.array_equal(got, c) and np.array_equal(np.asarray(Image.open(q)), c)
    # plain samples with another maxval: clamped to it, then i * 255 / maxval (integer division, grfmt_pxm.cpp)
    v = rng.integers(0, 40, (H, W))
    got, _ = rd("p2_31.pgm", b"P2 %d %d 31\n" % (W, H) + b" ".join(b"%d" % x for x in v.ravel()) + b"\n")
    assert np.array_equal(got, (np.minimum(v, 31) * 255 // 31).astype(np.uint8))
    # binary 8-bit samples are taken as they are, whatever maxval says
    got, _ = rd("p5_100.pgm", b"P5 %d %d 100\n" % (W, H) + g.tobytes())
    assert np.array_equal(got, g)
    # 16-bit samples: the high byte (binary: big-endian)
    g16 = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    got, _ = rd("p5_16.pgm", b"P5 %d %d 65535\n" % (W, H) + g16.astype(">u2").tobytes())
    assert np.array_equal(got, (g16 >> 8).astype(np.uint8))
    c16 = rng.integers(0, 1024, (H, W, 3), dtype=np.uint16)
    got, _ = rd("p6_16.ppm", b"P6 %d %d 1023\n" % (W, H) + c16.astype(">u2").tobytes())
    assert np.array_equal(got, (c16 >> 8).astype(np.uint8))
    got, _ = rd("p2_16.pgm", b"P2 %d %d 65535\n" % (W, H) + b" ".join(b"%d" % x for x in g16.ravel()))
    assert np.array_equal(got, (g16 >> 8).astype(np.uint8))
    # bitmaps: 1 = black
    bits = rng.integers(0, 2, (H, W), dtype=np.uint8)
    packed = np.packbits(bits, axis=1).tobytes()
    got, q = rd("p4.pbm", b"P4 %d %d\n" % (W, H) + packed)
    assert np.array_equal(got, (1 - bits) * 255) and np.array_equal(np.asarray(Image.open(q).convert("L")), (1 - bits) * 255)
    got, _ = rd("p1.pbm", b"P1\n%d %d\n" % (W, H) + b"\n".join(b"".join(b"%d" % x for x in row) for row in bits))
    assert np.array_equal(got, (1 - bits) * 255)
    # damaged: too few samples, a header that promises more than the file holds, maxval out of range
    for name, data in (("short.pgm", b"P2 4 4 255 1 2 3"), ("huge.pgm", b"P5 60000 60000 255\n" + bytes(100)), ("mv.pgm", b"P5 2 2 70000\n" + bytes(8)),
                       ("p7.pam", b"P7\nWIDTH 2\n")):
        (tmp_path / name).write_bytes(data)
        with pytest.raises(hesaff_amd.HesaffError):
            hesaff_amd.read_image(str(tmp_path / name))


# ------------------------------------------------------------------------------------------
# JPEG: hesaff_read_jpeg restates libjpeg's integer algorithms at cv::imread's settings
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["jpeg_gray_q90", "jpeg_420_q85", "jpeg_422_q70_rst", "jpeg_prog_420_q80", "jpeg_prog_gray_q60"])
def test_read_jpeg_golden_pixels(name):
    """Committed JPEG files and the pixels libjpeg-turbo (Pillow 12.2) decoded them to when the fixtures were made."""
    import hesaff_amd
    got = hesaff_amd.read_image(os.path.join(GOLD, name + ".jpg"))
    ref_path = os.path.join(GOLD, name + (".pgm" if got.ndim == 2 else ".ppm"))
    assert np.array_equal(got, hesaff_amd.read_pnm(ref_path))


def test_read_jpeg_equals_libjpeg_on_many_encodings(tmp_path):
    """Every pixel equals libjpeg-turbo's (through Pillow, when installed): JDCT_ISLOW, fancy up-sampling, JFIF colour
    conversion -- 4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1, grey; odd sizes down to 1x1; restart intervals; optimised tables;
    sequential and progressive (spectral selection + successive approximation, libjpeg's default scan script)."""
    Image = pytest.importorskip("PIL.Image")
    import io
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
    for (h, w) in [(64, 64), (61, 83), (7, 9), (1, 1), (17, 3), (120, 211), (33, 2)]:
        for sub in [0, 1, 2, "4:4:0", "4:1:1", "gray"]:
            for q, extra in [(30, {}), (75, {"restart_marker_blocks": 3}), (95, {"optimize": True}), (100, {}),
                             (30, {"progressive": True}), (85, {"progressive": True, "restart_marker_blocks": 2}), (100, {"progressive": True})]:
                im = Image.fromarray(synth(h, w, sub != "gray"))
                kw = dict(quality=q, **extra)
                if sub != "gray":
                    kw["subsampling"] = sub
                buf = io.BytesIO()
                try:
                    im.save(buf, "JPEG", **kw)
                except Exception:   # noqa: BLE001  (an encoder option this Pillow does not know)
                    continue
                open(p, "wb").write(buf.getvalue())
                ref = np.asarray(Image.open(p))
                got = hesaff_amd.read_image(p)
                assert got.shape == ref.shape and np.array_equal(got, ref), (h, w, sub, q, extra)
                n += 1
    assert n > 200
    # 4:1:1 (luma 4x1) and its transpose (1x4): Pillow does not write them, but a 4:2:0 file whose luma sampling byte is patched has
    # the same six blocks per MCU - the stream decodes, into other positions - and exercises the replicating 4:1 up-sampling
    buf = io.BytesIO()
    Image.fromarray(synth(64, 64, True)).save(buf, "JPEG", quality=85, subsampling=2)
    raw = bytearray(buf.getvalue())
    sof = raw.find(b"\xff\xc0")
    assert sof > 0 and raw[sof + 11] == 0x22
    for hv in (0x41, 0x14):
        raw[sof + 11] = hv
        open(p, "wb").write(raw)
        ref = np.asarray(Image.open(p).convert("RGB"))
        assert np.array_equal(hesaff_amd.read_image(p), ref), hex(hv)
    # truncated / corrupt files: an error or an image, never a crash
    raw = open(os.path.join(GOLD, "jpeg_420_q85.jpg"), "rb").read()
    for cut in (2, 20, 200, len(raw) // 2, len(raw) - 3):
        open(p, "wb").write(raw[:cut])
        try:
            hesaff_amd.read_image(p)
        except hesaff_amd.HesaffError:
            pass


def test_jpeg_coefficients_for_the_device_pixel_stage(tmp_path):
    """hesaff_read_jpeg_coefficients (the host half of the JPEG reader when the device makes the pixels): layout of every sampling
    scheme, blob size, and the SAME quantised coefficients and tables from the sequential and the progressive encoding of an
    image (libjpeg codes identical coefficient arrays either way) - i.e. the progressive scans (spectral selection, successive
    approximation) assemble exactly what the sequential decoder reads.  The pixel half is checked on the GPU against
    hesaff_read_jpeg (tests/test_gpu_parity.py)."""
    Image = pytest.importorskip("PIL.Image")
    import hesaff_amd
    rng = np.random.default_rng(5)
    p = str(tmp_path / "t.jpg")
    n = 0
    for (h, w) in [(64, 64), (61, 83), (7, 9), (1, 1), (120, 211)]:
        a = np.clip(rng.normal(128, 60, (h, w, 3)), 0, 255).astype(np.uint8)
        for sub, hv in [(0, (1, 1)), (1, (2, 1)), (2, (2, 2)), ("4:4:0", (1, 2)), ("gray", (1, 1))]:
            blobs = []
            for prog in (False, True):
                im = Image.fromarray(a if sub != "gray" else a[..., 0])
                kw = dict(quality=80, progressive=prog)
                if sub != "gray":
                    kw["subsampling"] = sub
                try:
                    im.save(p, "JPEG", **kw)
                except Exception:   # noqa: BLE001
                    break
                lay, blob = hesaff_amd.read_jpeg_coefficients(p)
                nc = 1 if sub == "gray" else 3
                assert (lay.width, lay.height, lay.channels) == (w, h, nc)
                if nc == 3:
                    assert (lay.hx[1], lay.vx[1]) == hv and (lay.hx[0], lay.vx[0]) == (1, 1)
                    assert lay.cw[1] == -(-w // hv[0]) and lay.chgt[1] == -(-h // hv[1]) and lay.cw[0] == w
                blocks = sum(lay.bw[i] * lay.bh[i] for i in range(nc))
                assert blob.size == 1024 + 128 * blocks
                assert all(lay.bw[i] * 8 >= lay.cw[i] and lay.bh[i] * 8 >= lay.chgt[i] for i in range(nc))
                assert int(blob[384:388].view(np.int32)[0]) == (1 if nc == 3 else 0)
                q = blob[:128 * nc].view(np.uint16)
                assert q.min() >= 1
                blobs.append(blob)
            if len(blobs) == 2:
                assert np.array_equal(blobs[0], blobs[1]), (h, w, sub)
                n += 1
    assert n >= 20
    # not a JPEG, truncated JPEG: an error or a blob, never a crash
    raw = open(os.path.join(GOLD, "jpeg_420_q85.jpg"), "rb").read()
    for cut in (0, 2, 20, 200, len(raw) // 2, len(raw) - 3):
        open(p, "wb").write(raw[:cut])
        try:
            hesaff_amd.read_jpeg_coefficients(p)
        except hesaff_amd.HesaffError:
            pass


def _idct_islow_numpy(coef, quant):
    """jidctint.c (JDCT_ISLOW) on an array of blocks [n, 8, 8] int16 with one quantisation table [8, 8]: int64 arithmetic (a legal
    stream never leaves 32 bits), the same constants, shifts and range limit as idct_islow / k_jpeg_idct -> uint8 [n, 8, 8]."""
    F = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069, f2053=16819,
             f2562=20995, f3072=25172)

    def pass1d(d, shift):   # d: [..., 8] along the transformed axis (last)
        z2, z3 = d[..., 2], d[..., 6]
        z1 = (z2 + z3) * F["f0541"]
        tmp2 = z1 + z3 * -F["f1847"]
        tmp3 = z1 + z2 * F["f0765"]
        tmp0 = (d[..., 0] + d[..., 4]) << 13
        tmp1 = (d[..., 0] - d[..., 4]) << 13
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        t0, t1, t2, t3 = d[..., 7], d[..., 5], d[..., 3], d[..., 1]
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F["f1175"]
        t0, t1, t2, t3 = t0 * F["f0298"], t1 * F["f2053"], t2 * F["f3072"], t3 * F["f1501"]
        z1, z2, z3, z4 = z1 * -F["f0899"], z2 * -F["f2562"], z3 * -F["f1961"] + z5, z4 * -F["f0390"] + z5
        t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
        out = np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3], -1)
        return (out + (1 << (shift - 1))) >> shift
    d = coef.astype(np.int64) * quant.astype(np.int64)[None]
    ws = pass1d(np.swapaxes(d, 1, 2), 13 - 2)          # columns: transform along the row index
    ws = np.swapaxes(ws, 1, 2)
    o = pass1d(ws, 13 + 2 + 3)                          # rows
    v = o & 1023
    v = np.where(v >= 512, v - 1024, v) + 128
    return np.clip(v, 0, 255).astype(np.uint8)


def test_jpeg_coefficient_blob_transforms_to_the_readers_pixels(tmp_path):
    """CPU-side pin of the host half of the device JPEG path: the coefficient blob of a grey JPEG, put through a numpy restatement of
    libjpeg's islow inverse DCT (the transform k_jpeg_idct runs), gives exactly the pixels hesaff_read_jpeg decodes - sequential and
    progressive, restart intervals, a size that is no multiple of the block, the committed grey fixtures."""
    Image = pytest.importorskip("PIL.Image")
    import hesaff_amd
    rng = np.random.default_rng(9)
    files = [os.path.join(GOLD, "jpeg_gray_q90.jpg"), os.path.join(GOLD, "jpeg_prog_gray_q60.jpg")]
    yy, xx = np.mgrid[0:67, 0:93]
    a = np.clip(127 + 90 * np.sin(xx / 6.0) * np.cos(yy / 9.0) + rng.normal(0, 30, (67, 93)), 0, 255).astype(np.uint8)
    for k, kw in enumerate([dict(quality=92), dict(quality=40, progressive=True), dict(quality=75, restart_marker_blocks=2)]):
        q = str(tmp_path / ("g%d.jpg" % k))
        Image.fromarray(a).save(q, "JPEG", **kw)
        files.append(q)
    for f in files:
        lay, blob = hesaff_amd.read_jpeg_coefficients(f)
        assert lay.channels == 1
        bw, bh = lay.bw[0], lay.bh[0]
        quant = blob[:128].view(np.uint16).reshape(8, 8)
        coef = blob[1024:1024 + bw * bh * 128].view(np.int16).reshape(bw * bh, 8, 8)
        blocks = _idct_islow_numpy(coef, quant).reshape(bh, bw, 8, 8)
        plane = blocks.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)
        want = hesaff_amd.read_image(f)
        assert np.array_equal(plane[:lay.height, :lay.width], want), f
    # colour without sub-sampling: three planes + jdcolor.c's fixed-point conversion (what k_jpeg_pixels does for mode "none")
    rgb = np.clip(np.stack([a, a[::-1], a[:, ::-1]], -1).astype(np.int32) + rng.integers(-20, 20, (67, 93, 3)), 0, 255).astype(np.uint8)
    for k, kw in enumerate([dict(quality=88), dict(quality=60, progressive=True)]):
        q = str(tmp_path / ("c%d.jpg" % k))
        Image.fromarray(rgb).save(q, "JPEG", subsampling=0, **kw)
        lay, blob = hesaff_amd.read_jpeg_coefficients(q)
        assert lay.channels == 3 and list(lay.hx) == [1, 1, 1] and int(blob[384:388].view(np.int32)[0]) == 1
        planes, at = [], 1024
        for c in range(3):
            bw, bh = lay.bw[c], lay.bh[c]
            coef = blob[at:at + bw * bh * 128].view(np.int16).reshape(bw * bh, 8, 8)
            at += bw * bh * 128
            quant = blob[128 * c:128 * c + 128].view(np.uint16).reshape(8, 8)
            blocks = _idct_islow_numpy(coef, quant).reshape(bh, bw, 8, 8)
            planes.append(blocks.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)[:lay.height, :lay.width].astype(np.int64))
        Y, cb, cr = planes[0], planes[1] - 128, planes[2] - 128
        got = np.stack([np.clip(Y + ((91881 * cr + 32768) >> 16), 0, 255), np.clip(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16), 0, 255),
                        np.clip(Y + ((116130 * cb + 32768) >> 16), 0, 255)], -1).astype(np.uint8)
        assert np.array_equal(got, hesaff_amd.read_image(q)), kw


def test_binary_sidecar_holds_the_rows_of_the_text_file(tmp_path):
    """hesaff_write_bin (optional sidecar): the same rows as the text export - x, y, the ellipse (a, b, c)
    and the 128 bytes - unprinted; the text file is the 6-significant-digit print of exactly these floats."""
    import hesaff_amd
    rng = np.random.default_rng(11)
    for n in (0, 1, 9000):
        keys = np.zeros(n, hesaff_amd.KEYPOINT_DTYPE)
        keys["x"] = rng.uniform(0, 3840, n); keys["y"] = rng.uniform(0, 2160, n); keys["s"] = rng.uniform(1, 40, n)
        keys["a11"] = rng.uniform(0.5, 2, n); keys["a21"] = rng.uniform(-1, 1, n); keys["a22"] = 1.0 / np.maximum(keys["a11"], 1e-3)
        keys["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
        mr = hesaff_amd.default_params().mrSize
        q = str(tmp_path / ("k%d.hesaff.bin" % n))
        hesaff_amd.write_bin(q, keys, mr)
        assert os.path.getsize(q) == 16 + 148 * n
        rows = hesaff_amd.read_bin(q)
        assert len(rows) == n and np.array_equal(rows["desc"], keys["desc"])
        assert np.array_equal(rows["x"], keys["x"]) and np.array_equal(rows["y"], keys["y"])
        e = hesaff_amd.ellipse(keys, mr)
        assert np.array_equal(np.stack([rows["a"], rows["b"], rows["c"]], 1), e.astype(np.float32).reshape(n, 3))
        # the text file prints these floats with %g
        text = hesaff_amd.format_sift(keys, mr).split(b"\n")
        assert int(text[1]) == n
        for i in (0, n // 2, n - 1) if n else ():
            tok = text[2 + i].split()
            want = [b"%g" % float(v) for v in (rows["x"][i], rows["y"][i], rows["a"][i], rows["b"][i], rows["c"][i])]
            assert tok[:5] == want and [int(t) for t in tok[5:]] == rows["desc"][i].tolist()
    with pytest.raises(hesaff_amd.HesaffError):
        (tmp_path / "junk.bin").write_bytes(b"not a sidecar")
        hesaff_amd.read_bin(str(tmp_path / "junk.bin"))


STRICT = 0x100   # HESAFF_OUT_STRICT


def test_output_is_complete_counts_rows_and_writers_leave_no_torn_file(tmp_path):
    """hesaff_set_resume's test of an existing output: a text file cut at a ROW BOUNDARY - what a killed writer that
    does not go through <name>.part + rename leaves, e.g. the reference binary - is not complete; every writer of the library
    (single- and multi-threaded text, sidecar, batch, the *_rows forms) goes through a temporary name + rename and leaves nothing behind.
    ADVICE r05: the row count reads the whole file, so it is the STRICT form (HESAFF_OUT_STRICT, hesaff_set_resume(ctx, 2)); the default
    test is three small reads and exact for files the library's renaming writers made."""
    import hesaff_amd
    L = hesaff_amd.load_library()
    rng = np.random.default_rng(5)
    n = 9000
    keys = np.zeros(n, hesaff_amd.KEYPOINT_DTYPE)
    keys["x"] = rng.uniform(0, 3840, n); keys["y"] = rng.uniform(0, 2160, n); keys["s"] = rng.uniform(1, 30, n)
    keys["a11"] = rng.uniform(0.5, 2, n); keys["a21"] = rng.uniform(-1, 1, n); keys["a22"] = 1.0 / keys["a11"]
    keys["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    mr = hesaff_amd.default_params().mrSize
    text = hesaff_amd.format_sift(keys, mr)
    q = str(tmp_path / "a.pgm.hesaff.sift")
    for threads in (1, 4):     # hesaff_write_sift_mt: the block-wise single-thread form and the one-buffer form
        assert L.hesaff_write_sift_mt(os.fsencode(q), keys.ctypes.data_as(C.c_void_p), n, mr, threads) == 0
        assert open(q, "rb").read() == text and os.listdir(tmp_path) == [os.path.basename(q)]
        assert L.hesaff_output_is_complete(os.fsencode(q), 1) == n and L.hesaff_output_is_complete(os.fsencode(q), 1 | STRICT) == n
    b = str(tmp_path / "a.pgm.hesaff.bin")
    hesaff_amd.write_bin(b, keys, mr)
    assert L.hesaff_output_is_complete(os.fsencode(b), 2) == n and not [f for f in os.listdir(tmp_path) if ".part" in f]
    # the *_rows writers (rows formatted elsewhere - on the device): header + rows in one writev, the same file
    body = text.split(b"\n", 2)[2]
    r = str(tmp_path / "rows.hesaff.sift")
    assert L.hesaff_write_sift_rows(os.fsencode(r), body, len(body), n) == 0
    assert open(r, "rb").read() == text and not [f for f in os.listdir(tmp_path) if ".part" in f]
    binrows = open(b, "rb").read()[16:]
    r2 = str(tmp_path / "rows.hesaff.bin")
    assert L.hesaff_write_bin_rows(os.fsencode(r2), binrows, n) == 0 and open(r2, "rb").read() == open(b, "rb").read()
    # truncated at a row boundary after 75 % of the rows: header intact, last byte a newline, longer than 266 bytes per row on average
    lines = text.split(b"\n")
    cut = b"\n".join(lines[: 2 + (3 * n) // 4]) + b"\n"
    t = str(tmp_path / "torn.hesaff.sift")
    open(t, "wb").write(cut)
    assert L.hesaff_output_is_complete(os.fsencode(t), 1 | STRICT) == -1
    assert L.hesaff_output_is_complete(os.fsencode(t), 1) == n    # (the O(1) form cannot see it: the stated limit of the default)
    open(t, "wb").write(text + lines[5] + b"\n")       # one row too many
    assert L.hesaff_output_is_complete(os.fsencode(t), 1 | STRICT) == -1
    open(t, "wb").write(text[:-1])                     # no final newline
    assert L.hesaff_output_is_complete(os.fsencode(t), 1) == -1 and L.hesaff_output_is_complete(os.fsencode(t), 1 | STRICT) == -1
    open(t, "wb").write(text[: len(text) // 3])        # cut far too short for its row count: the O(1) form sees that
    assert L.hesaff_output_is_complete(os.fsencode(t), 1) == -1
    open(t, "wb").write(b"128\n0\n")
    assert L.hesaff_output_is_complete(os.fsencode(t), 1) == 0
    open(t, "wb").write(open(b, "rb").read()[:-148])   # sidecar with a row missing
    assert L.hesaff_output_is_complete(os.fsencode(t), 2) == -1
    # an unwritable target reports an error and leaves nothing behind
    bad = str(tmp_path / "no_such_dir" / "x.hesaff.sift")
    assert L.hesaff_write_sift_rows(os.fsencode(bad), body, len(body), n) != 0
    # a target that is not a regular file is written in place (/dev/stdout, a FIFO): here a FIFO with a reader thread
    import threading
    fifo = str(tmp_path / "pipe.hesaff.sift")
    os.mkfifo(fifo)
    got = []
    th = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    th.start()
    assert L.hesaff_write_sift_rows(os.fsencode(fifo), body, len(body), n) == 0
    th.join(30)
    assert got and got[0] == text and not [f for f in os.listdir(tmp_path) if ".part" in f]
    # an existing writable file in a directory that takes no new file: written in place as well
    if os.geteuid() != 0:    # (root creates files anywhere)
        ro = tmp_path / "ro"; ro.mkdir()
        target = str(ro / "y.hesaff.sift"); open(target, "wb").write(b"old")
        os.chmod(ro, 0o555)
        try:
            assert L.hesaff_write_sift_rows(os.fsencode(target), body, len(body), n) == 0 and open(target, "rb").read() == text
        finally:
            os.chmod(ro, 0o755)


def test_host_plan_is_one_rule():
    """hesaff_host_plan_for: threads per device = f(CPUs this process may use, devices that share them) - the rule the
    CLI, hesaff_process_files' "auto", bench.py and tools/batch_ranks.py all use."""
    import hesaff_amd
    host = int(hesaff_amd.load_library().hesaff_host_threads())
    assert 1 <= host <= 64
    for dev in (1, 2, 8, 64):
        hp = hesaff_amd.host_plan(dev)
        cpus = max(1, host // dev)
        stage = max(1, min(4, cpus // 4))
        pool = max(2, cpus - stage)
        assert hp == {"cpus": cpus, "decode_threads": max(1, pool // 4), "write_threads": pool - max(1, pool // 4), "stage_threads": stage}
    with pytest.raises(hesaff_amd.HesaffError):
        hesaff_amd.host_plan(0)
    # hesaff_process_files gets the plan's two pool counts, not the plan: the staging threads it derives from them are the plan's
    # (8 CPUs -> pool 6, stage 2; 16 CPUs -> pool 12, stage 4)
    L = hesaff_amd.load_library()
    by_pool = {}
    for cpus in range(1, 257):
        stage = max(1, min(4, cpus // 4))
        by_pool.setdefault(max(2, cpus - stage), set()).add(stage)
    for pool, stages in by_pool.items():      # the rule is not one-to-one (7 and 8 CPUs share a pool of 6): the larger count is taken
        assert L.hesaff_stage_threads_for_pool(pool) == max(stages), (pool, stages)
    assert L.hesaff_stage_threads_for_pool(6) == 2 and L.hesaff_stage_threads_for_pool(12) == 4 and L.hesaff_stage_threads_for_pool(2) == 1


def test_read_bmp_matches_imread_semantics(tmp_path):
    """hesaff_read_bmp (cv::imread hesaff.cpp:137 for Windows bitmaps, OpenCV's grfmt_bmp conventions): files written by Pillow (1-, 8-bit
    palette, 24-, 32-bit) against Pillow's own decoder, and hand-made files for what Pillow does not write: 4-bit palette, top-down rows,
    16-bit 5-5-5 / 5-6-5, RLE8 with every escape, an OS/2 core header; damaged files are refused, never read past."""
    import struct
    import hesaff_amd
    from PIL import Image
    rng = np.random.default_rng(11)
    for H, W in ((37, 53), (8, 8), (1, 1), (5, 1024)):
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        q = str(tmp_path / "c.bmp"); Image.fromarray(rgb).save(q)
        assert np.array_equal(hesaff_amd.read_image(q), rgb)
        rgba = np.dstack([rgb, rng.integers(0, 256, (H, W, 1), dtype=np.uint8)])
        q = str(tmp_path / "a.bmp"); Image.fromarray(rgba, "RGBA").save(q)
        assert np.array_equal(hesaff_amd.read_image(q), rgb)              # fourth byte dropped
        g = rng.integers(0, 256, (H, W), dtype=np.uint8)
        q = str(tmp_path / "g.bmp"); Image.fromarray(g).save(q)           # 8-bit, grey palette -> one channel
        assert np.array_equal(hesaff_amd.read_image(q), g)
        p = Image.fromarray(g).convert("P"); p.putpalette(list(rng.integers(0, 256, 768)))
        q = str(tmp_path / "p.bmp"); p.save(q)
        assert np.array_equal(hesaff_amd.read_image(q), np.asarray(Image.open(q).convert("RGB")))
        bits = rng.integers(0, 2, (H, W), dtype=np.uint8)
        q = str(tmp_path / "b.bmp"); Image.fromarray(bits * 255).convert("1").save(q)
        assert np.array_equal(hesaff_amd.read_image(q), bits * 255)

    def bmp(w, h, bpp, body, comp=0, palette=b"", masks=b"", top_down=False):
        off = 14 + 40 + len(masks) + len(palette)
        hdr = struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, bpp, comp, len(body), 2835, 2835, len(palette) // 4, 0)
        return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + hdr + masks + palette + body

    H, W = 9, 13
    # 4-bit palette, bottom-up and top-down
    pal = rng.integers(0, 256, (16, 4), dtype=np.uint8); pal[:, 3] = 0
    idx = rng.integers(0, 16, (H, W), dtype=np.uint8)
    stride = ((W * 4 + 31) // 32) * 4
    rows = []
    for y in range(H):
        r = bytearray(stride)
        for x in range(W):
            r[x >> 1] |= int(idx[y, x]) << (0 if x & 1 else 4)
        rows.append(bytes(r))
    want = pal[idx][:, :, [2, 1, 0]]
    for td in (False, True):
        q = str(tmp_path / "n4.bmp"); open(q, "wb").write(bmp(W, H, 4, b"".join(rows if td else rows[::-1]), palette=pal.tobytes(), top_down=td))
        assert np.array_equal(hesaff_amd.read_image(q), want), td
        assert np.array_equal(np.asarray(Image.open(q).convert("RGB")), want), td
    # 16 bits: 5-5-5 (BI_RGB) and 5-6-5 (BI_BITFIELDS); low bits stay zero (icvCvt_BGR5552BGR / 5652BGR)
    v = rng.integers(0, 65536, (H, W)).astype(np.uint16)
    stride = ((W * 16 + 31) // 32) * 4
    body = b"".join(v[y].astype("<u2").tobytes().ljust(stride, b"\0") for y in range(H - 1, -1, -1))
    q = str(tmp_path / "n555.bmp"); open(q, "wb").write(bmp(W, H, 16, body))
    want = np.dstack([(v >> 7) & 0xF8, (v >> 2) & 0xF8, (v << 3) & 0xFF]).astype(np.uint8)
    assert np.array_equal(hesaff_amd.read_image(q), want)
    q = str(tmp_path / "n565.bmp"); open(q, "wb").write(bmp(W, H, 16, body, comp=3, masks=struct.pack("<III", 0xF800, 0x7E0, 0x1F)))
    want = np.dstack([(v >> 8) & 0xF8, (v >> 3) & 0xFC, (v << 3) & 0xFF]).astype(np.uint8)
    assert np.array_equal(hesaff_amd.read_image(q), want)
    # RLE8: runs, literal runs (odd length: padded), end of line, a delta, end of bitmap; skipped pixels keep palette entry 0
    pal = rng.integers(0, 256, (256, 4), dtype=np.uint8); pal[:, 3] = 0
    exp = np.zeros((4, 8), np.uint8)          # indices, FILE row order (bottom row first)
    s = bytes([5, 7]) + bytes([0, 3, 1, 2, 3, 0]) + bytes([0, 0])                 # row 0: five 7s, literal 1 2 3 (+ pad), end of line
    exp[0, :5] = 7; exp[0, 5:8] = (1, 2, 3)
    s += bytes([2, 9]) + bytes([0, 2, 3, 1]) + bytes([2, 4]) + bytes([0, 0])      # row 1: two 9s, move +3,+1 -> row 2 column 5: two 4s, end of line
    exp[1, :2] = 9; exp[2, 5:7] = 4
    s += bytes([8, 200]) + bytes([0, 1])                                          # row 3: eight 200s, end of bitmap
    exp[3, :] = 200
    q = str(tmp_path / "rle8.bmp"); open(q, "wb").write(bmp(8, 4, 8, s, comp=1, palette=pal.tobytes()))
    want = pal[exp[::-1]][:, :, [2, 1, 0]]
    assert np.array_equal(hesaff_amd.read_image(q), want)
    assert np.array_equal(np.asarray(Image.open(q).convert("RGB")), want)
    # OS/2 core header (12 bytes, 3-byte palette entries), 8 bits
    idx = rng.integers(0, 256, (H, W), dtype=np.uint8)
    pal3 = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    stride = (W + 3) & ~3
    body = b"".join(idx[y].tobytes().ljust(stride, b"\0") for y in range(H - 1, -1, -1))
    off = 14 + 12 + 768
    core = b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + struct.pack("<IHHHH", 12, W, H, 1, 8) + pal3.tobytes() + body
    q = str(tmp_path / "core.bmp"); open(q, "wb").write(core)
    assert np.array_equal(hesaff_amd.read_image(q), pal3[idx][:, :, [2, 1, 0]])
    # damaged: truncated pixel data, absurd sizes, an unknown compression, a palette that runs past the file
    good = open(str(tmp_path / "c.bmp"), "rb").read()
    for bad in (good[: len(good) - 7], good[:30], bmp(W, H, 24, b"\0" * 10), bmp(1 << 20, 1 << 20, 24, b""), bmp(W, H, 8, b"\0" * 400, comp=7),
                bmp(W, H, 8, b"", palette=b"")[:54], b"BM" + b"\0" * 10):
        q = str(tmp_path / "bad.bmp"); open(q, "wb").write(bad)
        with pytest.raises(hesaff_amd.HesaffError):
            hesaff_amd.read_image(q)


def _tiff_bytes(w, h, bits, photo, spp, chunks, be=False, tile=None, rps=None, planar=1, cmap=None, extra=None, comp=1):
    """A minimal TIFF writer for the reader's tests: `chunks` = the strips' / tiles' bytes in file order."""
    import struct
    E = ">" if be else "<"
    ents = []

    def ent(tag, typ, vals):
        ents.append((tag, typ, list(vals)))
    ent(256, 4, [w]); ent(257, 4, [h]); ent(258, 3, [bits] * spp); ent(259, 3, [comp]); ent(262, 3, [photo]); ent(277, 3, [spp])
    if planar != 1:
        ent(284, 3, [planar])
    if extra is not None:
        ent(338, 3, [extra])
    if cmap is not None:
        ent(320, 3, cmap)
    if tile:
        ent(322, 3, [tile[0]]); ent(323, 3, [tile[1]])
    else:
        ent(278, 4, [rps or h])
    off_tag, cnt_tag = (324, 325) if tile else (273, 279)
    ent(off_tag, 4, [0] * len(chunks)); ent(cnt_tag, 4, [len(c) for c in chunks])
    ents.sort()
    n = len(ents)
    ifd_at = 8
    extra_at = ifd_at + 2 + 12 * n + 4
    blob = b""
    recs = []
    for tag, typ, vals in ents:
        sz = {1: 1, 3: 2, 4: 4}[typ]
        raw = b"".join(struct.pack(E + {1: "B", 3: "H", 4: "I"}[typ], v) for v in vals)
        if len(raw) <= 4:
            recs.append([tag, typ, len(vals), raw.ljust(4, b"\0"), None])
        else:
            recs.append([tag, typ, len(vals), None, len(blob)]); blob += raw + (b"\0" if len(raw) & 1 else b"")
    data_at = extra_at + len(blob)
    offs, at = [], data_at
    for c in chunks:
        offs.append(at); at += len(c)
    for r in recs:
        if r[0] == off_tag:
            raw = b"".join(struct.pack(E + "I", o) for o in offs)
            if len(raw) <= 4:
                r[3] = raw.ljust(4, b"\0")
            else:
                blob = blob[: r[4]] + raw + blob[r[4] + len(raw):]
    out = (b"MM" if be else b"II") + struct.pack(E + "HI", 42, ifd_at) + struct.pack(E + "H", n)
    for tag, typ, cnt, inline, at_blob in recs:
        out += struct.pack(E + "HHI", tag, typ, cnt) + (inline if inline is not None else struct.pack(E + "I", extra_at + at_blob))
    return out + struct.pack(E + "I", 0) + blob + b"".join(chunks)


def test_read_tiff_matches_imread_semantics(tmp_path):
    """hesaff_read_tiff (cv::imread hesaff.cpp:137 for TIFF: libtiff's RGBA interface, alpha dropped): files written by Pillow's libtiff
    (uncompressed, LZW with and without the horizontal predictor, PackBits, Deflate; grey, RGB, palette, bilevel, several strip heights)
    against Pillow's decoder, and hand-made files for the rest: tiles, planar samples, big-endian, MinIsWhite, 4-bit grey, an 8-bit
    colour map, unassociated alpha.  Unsupported and damaged files are refused."""
    import hesaff_amd
    from PIL import Image
    rng = np.random.default_rng(23)
    for H, W in ((37, 53), (64, 64), (1, 1), (300, 211)):
        # smooth + noisy content: long LZW strings, table resets (the 300 x 211 file exceeds 4096 codes per strip), literal runs
        ramp = (np.add.outer(np.arange(H), np.arange(W)) // 3 % 256).astype(np.uint8)
        g = np.where(rng.random((H, W)) < 0.3, rng.integers(0, 256, (H, W)), ramp).astype(np.uint8)
        rgb = np.stack([g, np.roll(g, 2, 1), 255 - g], 2)
        for comp, kw in (("raw", {}), ("tiff_lzw", {}), ("tiff_lzw", {"tiffinfo": {317: 2}}), ("packbits", {}), ("tiff_adobe_deflate", {}),
                         ("tiff_adobe_deflate", {"tiffinfo": {317: 2}}), ("tiff_lzw", {"tiffinfo": {278: 7}})):
            for name, arr, mode in (("g", g, "L"), ("c", rgb, "RGB")):
                q = str(tmp_path / ("%s_%s.tif" % (name, comp)))
                Image.fromarray(arr, mode).save(q, "TIFF", compression=comp, **kw)
                assert np.array_equal(hesaff_amd.read_image(q), arr), (H, W, comp, kw, name)
        p = Image.fromarray(g).convert("P"); p.putpalette(list(rng.integers(0, 256, 768)))
        q = str(tmp_path / "p.tif"); p.save(q, "TIFF", compression="tiff_lzw")
        assert np.array_equal(hesaff_amd.read_image(q), np.asarray(Image.open(q).convert("RGB")))
        bits = rng.integers(0, 2, (H, W), dtype=np.uint8)
        q = str(tmp_path / "b.tif"); Image.fromarray(bits * 255).convert("1").save(q, "TIFF", compression="packbits")
        assert np.array_equal(hesaff_amd.read_image(q), np.asarray(Image.open(q).convert("L")))
    H, W = 19, 27
    g = rng.integers(0, 256, (H, W), dtype=np.uint8)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)

    def put(name, blob):
        q = str(tmp_path / name); open(q, "wb").write(blob); return q
    # tiles of 16 x 16 (padded at the right and bottom edges), little- and big-endian
    for be in (False, True):
        tiles = []
        for ty in range(0, H, 16):
            for tx in range(0, W, 16):
                t = np.zeros((16, 16, 3), np.uint8); blk = rgb[ty:ty + 16, tx:tx + 16]; t[: blk.shape[0], : blk.shape[1]] = blk
                tiles.append(t.tobytes())
        assert np.array_equal(hesaff_amd.read_image(put("tiles.tif", _tiff_bytes(W, H, 8, 2, 3, tiles, be=be, tile=(16, 16)))), rgb), be
        # strips of 5 rows, planar: all strips of R, then G, then B
        strips = [rgb[y:y + 5, :, k].tobytes() for k in range(3) for y in range(0, H, 5)]
        assert np.array_equal(hesaff_amd.read_image(put("planar.tif", _tiff_bytes(W, H, 8, 2, 3, strips, be=be, rps=5, planar=2))), rgb), be
    # MinIsWhite 8 bit, 4-bit MinIsBlack (v * 255 / 15), bilevel MinIsWhite
    assert np.array_equal(hesaff_amd.read_image(put("w.tif", _tiff_bytes(W, H, 8, 0, 1, [g.tobytes()]))), 255 - g)
    n4 = rng.integers(0, 16, (H, W), dtype=np.uint8)
    rows = b"".join(bytes(((int(r[i]) << 4) | (int(r[i + 1]) if i + 1 < W else 0)) for i in range(0, W, 2)) for r in n4)
    assert np.array_equal(hesaff_amd.read_image(put("n4.tif", _tiff_bytes(W, H, 4, 1, 1, [rows]))), (n4.astype(np.int32) * 255 // 15).astype(np.uint8))
    b1 = rng.integers(0, 2, (H, W), dtype=np.uint8)
    rows = b"".join(np.packbits(r).tobytes() for r in b1)
    assert np.array_equal(hesaff_amd.read_image(put("b1w.tif", _tiff_bytes(W, H, 1, 0, 1, [rows]))), (1 - b1) * 255)
    # palette: a 16-bit colour map is reduced with >> 8, a map whose entries all stay below 256 is an 8-bit map (libtiff's checkcmap)
    cm16 = rng.integers(0, 65536, 768); cm8 = rng.integers(0, 256, 768)
    for cm, f in ((cm16, lambda v: v >> 8), (cm8, lambda v: v)):
        got = hesaff_amd.read_image(put("pal.tif", _tiff_bytes(W, H, 8, 3, 1, [g.tobytes()], cmap=[int(v) for v in cm])))
        want = np.stack([f(cm[k * 256 + g.astype(np.int64)]) for k in range(3)], 2).astype(np.uint8)
        assert np.array_equal(got, want)
    # RGB + alpha: unassociated alpha (ExtraSamples = 2) is multiplied in as libtiff does, associated alpha (1) is dropped
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    a = rgba[:, :, 3:4].astype(np.int32)
    assert np.array_equal(hesaff_amd.read_image(put("ua.tif", _tiff_bytes(W, H, 8, 2, 4, [rgba.tobytes()], extra=2))), ((rgba[:, :, :3].astype(np.int32) * a + 127) // 255).astype(np.uint8))
    assert np.array_equal(hesaff_amd.read_image(put("aa.tif", _tiff_bytes(W, H, 8, 2, 4, [rgba.tobytes()], extra=1))), rgba[:, :, :3])
    # refused: 16-bit samples, YCbCr, JPEG compression, BigTIFF, truncated data, offsets past the file, an absurd size
    good = _tiff_bytes(W, H, 8, 1, 1, [g.tobytes()])
    g16 = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    bad = [_tiff_bytes(W, H, 16, 1, 1, [g16.tobytes()]), _tiff_bytes(W, H, 8, 6, 3, [rgb.tobytes()]), _tiff_bytes(W, H, 8, 1, 1, [g.tobytes()], comp=7),
           b"II+\0" + good[4:], good[: len(good) - 11], good[:40], _tiff_bytes(1 << 17, 1 << 17, 8, 1, 1, [b"\0" * 16]), b"II*\0\xff\xff\xff\x7f"]
    for blob in bad:
        with pytest.raises(hesaff_amd.HesaffError):
            hesaff_amd.read_image(put("bad.tif", blob))
