"""AddressSanitizer + UBSan run of the host side of the C ABI (image readers, text writer) on damaged inputs.

CPU only.  The readers replace cv::imread (hesaff.cpp:137) and the writer exportKeypoints (hesaff.cpp:107-130); the
reference trusts its input, a drop-in that serves batches must refuse a damaged file without reading or writing memory
it does not own.  GPU AddressSanitizer is not available on the pool, so this is the sanitizer coverage the product gets."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SRC = [os.path.join(ROOT, "hesaff_amd", "csrc", "hostio.cpp"), os.path.join(ROOT, "hesaff_amd", "csrc", "jpeg_decode.cpp"),
       os.path.join(ROOT, "tests", "native", "hostio_sanitize.cpp")]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("san") / "hostio_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-o", out] + SRC + ["-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


def _run(driver, args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver] + args, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, "sanitizer or driver failure (rc %d)\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    return r.stdout


def _mutants(data, rng, n):
    out = []
    for _ in range(n):
        b = bytearray(data)
        kind = rng.randrange(4)
        if kind == 0 and len(b) > 8:                       # truncated
            b = b[: rng.randrange(1, len(b))]
        elif kind == 1:                                    # a few flipped bytes, header region favoured
            for _ in range(rng.randrange(1, 6)):
                pos = rng.randrange(min(len(b), 600)) if rng.random() < 0.7 else rng.randrange(len(b))
                b[pos] = rng.randrange(256)
        elif kind == 2:                                    # a 16/32-bit field blown up (sizes, lengths, counts)
            pos = rng.randrange(max(1, min(len(b), 400) - 4))
            b[pos:pos + 4] = bytes([0xFF, 0xFF, 0xFF, rng.randrange(256)])
        else:                                              # a slice removed or duplicated
            i = rng.randrange(len(b)); j = min(len(b), i + rng.randrange(1, 64))
            b = b[:i] + (b[i:j] * 2 if rng.random() < 0.5 else b"") + b[j:]
        out.append(bytes(b))
    return out


def _dht_mutants(data, rng, n):
    """Structure-aware damage of a JPEG's Huffman tables: the 16 code-length counts of a DHT table are redistributed while
    their sum (the number of symbols) stays what the segment length says, so the table passes every length check and only
    its prefix-code property breaks (over-subscribed lengths, all codes one bit long, ...).  A plain byte flip changes the
    sum and is refused before the table is ever built."""
    out = []
    segs = []
    pos = 2
    while pos + 4 <= len(data) and data[pos] == 0xFF:
        m = data[pos + 1]
        if m == 0xD8 or m == 0x01 or 0xD0 <= m <= 0xD7:
            pos += 2
            continue
        ln = (data[pos + 2] << 8) | data[pos + 3]
        if m == 0xC4:
            o = pos + 4
            while o + 17 <= pos + 2 + ln:
                nv = sum(data[o + 1:o + 17])
                segs.append((o + 1, nv))
                o += 17 + nv
        if m == 0xDA:
            break
        pos += 2 + ln
    for _ in range(n):
        if not segs:
            break
        b = bytearray(data)
        for at, nv in rng.sample(segs, rng.randrange(1, len(segs) + 1)):
            counts = [0] * 16
            kind = rng.randrange(4)
            if kind == 0:                                   # everything at one short length
                counts[rng.randrange(0, 4)] = nv
            elif kind == 1:                                 # random split, short lengths favoured
                left = nv
                for l in range(16):
                    c = min(left, 255, rng.randrange(0, max(1, left // 2) + 1)) if l < 15 else min(left, 255)
                    counts[l] = c; left -= c
                counts[15] = min(255, counts[15] + left)
            elif kind == 2:                                 # move a few codes to a shorter length
                counts = list(b[at:at + 16])
                for _ in range(rng.randrange(1, 5)):
                    src = rng.randrange(16)
                    if counts[src] == 0:
                        continue
                    dst = rng.randrange(0, src + 1)
                    k = rng.randrange(1, counts[src] + 1)
                    if counts[dst] + k <= 255:
                        counts[src] -= k; counts[dst] += k
            else:                                           # exactly one code too many at some length
                counts = list(b[at:at + 16])
                l = rng.randrange(0, 9)
                room = (1 << (l + 1))
                take = [i for i in range(16) if i != l and counts[i] > 0]
                while counts[l] <= room and take and sum(counts) == nv:
                    i = take[-1]
                    counts[i] -= 1; counts[l] += 1
                    if counts[i] == 0:
                        take.pop()
                    if counts[l] > 255:
                        break
                counts[l] = min(counts[l], 255)
            if sum(counts) == nv and all(0 <= c <= 255 for c in counts):
                b[at:at + 16] = bytes(counts)
        out.append(bytes(b))
    return out


# SOI + one DHT whose counts say "200 codes of length 1" + EOI.  The lookahead fill of Huff::build wrote
# past look_len[512] / look_val[512] (UBSan: index 512 out of bounds); the reader must refuse the table instead.
OVERSUBSCRIBED_DHT = b"\xff\xd8" + b"\xff\xc4" + bytes([0, 2 + 17 + 200]) + bytes([0x10, 200] + [0] * 15) + bytes(range(200)) + b"\xff\xd9"


def test_oversubscribed_huffman_table_is_refused(driver, tmp_path):
    assert len(OVERSUBSCRIBED_DHT) == 225
    paths = []
    for k, blob in enumerate([OVERSUBSCRIBED_DHT,
                              # three one-bit codes; 129 two-bit codes; a full tree plus one
                              b"\xff\xd8\xff\xc4" + bytes([0, 2 + 17 + 3]) + bytes([0x10, 3] + [0] * 15) + bytes(3) + b"\xff\xd9",
                              b"\xff\xd8\xff\xc4" + bytes([0, 2 + 17 + 129]) + bytes([0x11, 0, 129] + [0] * 14) + bytes(129) + b"\xff\xd9",
                              b"\xff\xd8\xff\xc4" + bytes([0, 2 + 17 + 5]) + bytes([0x01, 1, 1, 3] + [0] * 13) + bytes(5) + b"\xff\xd9"]):
        p = str(tmp_path / ("dht%d.jpg" % k)); open(p, "wb").write(blob); paths.append(p)
    out = _run(driver, ["read"] + paths)
    assert "read ok=0" in out, out


def test_readers_refuse_damaged_files_without_memory_errors(driver, tmp_path):
    import numpy as np
    from tests.test_host_side import _adam7_png_bytes, _png_bytes
    rng = random.Random(20260202)
    nrng = np.random.default_rng(5)
    seeds = [f for f in sorted(os.listdir(GOLDEN)) if f.rsplit(".", 1)[-1] in ("pgm", "ppm", "jpg")]
    assert any(f.endswith(".jpg") for f in seeds) and any(f.endswith(".pgm") for f in seeds)
    blobs = [(f, open(os.path.join(GOLDEN, f), "rb").read()) for f in seeds]
    # PNG seeds of every colour type the reader accepts (grey, RGB, palette, alpha, 16 bit, packed grey)
    pal = nrng.integers(0, 256, (16, 3), dtype=np.uint8)
    blobs += [("g8.png", _png_bytes(nrng.integers(0, 256, (23, 31, 1), dtype=np.uint8), 0)),
              ("rgb.png", _png_bytes(nrng.integers(0, 256, (19, 17, 3), dtype=np.uint8), 2)),
              ("rgba.png", _png_bytes(nrng.integers(0, 256, (9, 13, 4), dtype=np.uint8), 6)),
              ("g16.png", _png_bytes(nrng.integers(0, 65536, (11, 7, 1), dtype=np.uint16), 0, depth=16)),
              ("g2.png", _png_bytes(nrng.integers(0, 4, (10, 21, 1), dtype=np.uint8), 0, depth=2, filters=(0, 2))),
              ("pal4.png", _png_bytes(nrng.integers(0, 16, (12, 15, 1), dtype=np.uint8), 3, depth=4, palette=pal, filters=(0,))),
              # round 4: Adam7-interlaced PNG and the other PNM forms imread reads (plain, 16-bit, bitmaps, maxval != 255)
              ("a7_rgb.png", _adam7_png_bytes(nrng.integers(0, 256, (19, 23, 3), dtype=np.uint8), 2)),
              ("a7_g1.png", _adam7_png_bytes(nrng.integers(0, 2, (17, 29, 1), dtype=np.uint8), 0, depth=1)),
              ("a7_pal4.png", _adam7_png_bytes(nrng.integers(0, 16, (9, 9, 1), dtype=np.uint8), 3, depth=4, palette=pal)),
              ("p2.pgm", b"P2\n# c\n13 9\n200\n" + b" ".join(b"%d" % v for v in nrng.integers(0, 256, 13 * 9)) + b"\n"),
              ("p3.ppm", b"P3 5 4 1023 " + b" ".join(b"%d" % v for v in nrng.integers(0, 1024, 60))),
              ("p5_16.pgm", b"P5 11 7 65535\n" + nrng.integers(0, 65536, 77, dtype=np.uint16).astype(">u2").tobytes()),
              ("p4.pbm", b"P4 13 6\n" + nrng.integers(0, 256, 12, dtype=np.uint8).tobytes()),
              ("p1.pbm", b"P1 7 5\n" + b"".join(b"%d" % v for v in nrng.integers(0, 2, 35)))]
    # round 6: Windows bitmaps (24-bit, 4-bit palette, 16-bit with bit fields, RLE8 with escapes)
    import struct

    def bmp(w, h, bpp, body, comp=0, palette=b"", masks=b""):
        off = 14 + 40 + len(masks) + len(palette)
        hdr = struct.pack("<IiiHHIIiiII", 40, w, h, 1, bpp, comp, len(body), 2835, 2835, len(palette) // 4, 0)
        return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + hdr + masks + palette + body
    blobs += [("c24.bmp", bmp(13, 9, 24, nrng.integers(0, 256, 40 * 9, dtype=np.uint8).tobytes())),
              ("p4.bmp", bmp(13, 9, 4, nrng.integers(0, 256, 8 * 9, dtype=np.uint8).tobytes(), palette=nrng.integers(0, 256, 64, dtype=np.uint8).tobytes())),
              ("b565.bmp", bmp(13, 9, 16, nrng.integers(0, 256, 28 * 9, dtype=np.uint8).tobytes(), comp=3, masks=struct.pack("<III", 0xF800, 0x7E0, 0x1F))),
              ("rle8.bmp", bmp(8, 4, 8, bytes([5, 7, 0, 3, 1, 2, 3, 0, 0, 0, 2, 9, 0, 2, 3, 1, 2, 4, 0, 0, 8, 200, 0, 1]), comp=1,
                               palette=nrng.integers(0, 256, 1024, dtype=np.uint8).tobytes()))]
    # ... and baseline TIFF: LZW / PackBits / Deflate strips written by Pillow's libtiff, hand-made tiles, planar samples and a colour map
    from io import BytesIO
    from PIL import Image
    from tests.test_host_side import _tiff_bytes
    ramp = (np.add.outer(np.arange(40), np.arange(61)) // 3 % 256).astype(np.uint8)
    tg = np.where(nrng.random((40, 61)) < 0.3, nrng.integers(0, 256, (40, 61)), ramp).astype(np.uint8)
    trgb = np.stack([tg, np.roll(tg, 2, 1), 255 - tg], 2)
    for name, arr, mode, comp, kw in (("lzw.tif", trgb, "RGB", "tiff_lzw", {"tiffinfo": {317: 2}}), ("pb.tif", tg, "L", "packbits", {}),
                                      ("zip.tif", trgb, "RGB", "tiff_adobe_deflate", {}), ("lzw1.tif", tg, "L", "tiff_lzw", {"tiffinfo": {278: 40}})):
        bio = BytesIO(); Image.fromarray(arr, mode).save(bio, "TIFF", compression=comp, **kw); blobs.append((name, bio.getvalue()))
    tiles = []
    for ty in range(0, 40, 16):
        for tx in range(0, 61, 16):
            t = np.zeros((16, 16, 3), np.uint8); blk = trgb[ty:ty + 16, tx:tx + 16]; t[: blk.shape[0], : blk.shape[1]] = blk; tiles.append(t.tobytes())
    blobs += [("tiles.tif", _tiff_bytes(61, 40, 8, 2, 3, tiles, tile=(16, 16))),
              ("planar_be.tif", _tiff_bytes(61, 40, 8, 2, 3, [trgb[y:y + 8, :, k].tobytes() for k in range(3) for y in range(0, 40, 8)], be=True, rps=8, planar=2)),
              ("pal.tif", _tiff_bytes(61, 40, 8, 3, 1, [tg.tobytes()], cmap=[int(v) for v in nrng.integers(0, 65536, 768)]))]
    paths = []
    for f, data in blobs:
        p0 = str(tmp_path / ("intact_" + f)); open(p0, "wb").write(data); paths.append(p0)   # the intact file too
        for k, m in enumerate(_mutants(data, rng, 150)):
            p = str(tmp_path / ("%s.%03d" % (f, k)))
            open(p, "wb").write(m)
            paths.append(p)
        if f.endswith(".jpg"):
            for k, m in enumerate(_dht_mutants(data, rng, 120)):
                p = str(tmp_path / ("%s.dht%03d" % (f, k)))
                open(p, "wb").write(m)
                paths.append(p)
    # plus files that are not images at all
    for k, blob in enumerate([b"", b"P5", b"P5\n99999999 99999999\n255\n", b"\x89PNG\r\n\x1a\n", b"\xff\xd8\xff", b"\xff\xd8" + b"\xff\xc0" * 40]):
        p = str(tmp_path / ("junk%d" % k)); open(p, "wb").write(blob); paths.append(p)
    out = ""
    for i in range(0, len(paths), 400):
        out = _run(driver, ["read"] + paths[i:i + 400])
    assert "read ok=" in out


def test_text_writer_on_arbitrary_bit_patterns(driver):
    for seed, n in [(1, 1), (2, 333), (3, 20000)]:
        out = _run(driver, ["format", str(seed), str(n)])
        assert "same=1" in out


ENGINE_SRC = [os.path.join(ROOT, "hesaff_amd", "csrc", "hostio.cpp"), os.path.join(ROOT, "hesaff_amd", "csrc", "jpeg_decode.cpp"),
              os.path.join(ROOT, "tests", "native", "engine_sanitize.cpp")]


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_file_pipeline_threads_under_sanitizers(san, tmp_path):
    """The host-only half of hesaff_process_files (hesaff_amd/csrc/chunk_engine.h: decoder threads, the bounded look-ahead window,
    chunk formation, the ring of three result blocks, writer threads, per-file status) under ThreadSanitizer and under
    AddressSanitizer + UBSan, with the device replaced by a mock loop of the same threading shape (tests/native/
    engine_sanitize.cpp).  The GPU suite runs the same code for its results; this run is for its races and lifetimes."""
    import numpy as np
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "engine_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-o", exe] + ENGINE_SRC + ["-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    rng = np.random.default_rng(3)
    names = []
    for i in range(45):
        h, w = [(30, 40), (30, 40), (25, 31), (30, 40), (30, 40)][i % 5]      # runs of equal and unequal sizes
        q = tmp_path / ("i%02d.pgm" % i)
        q.write_bytes(b"P5\n%d %d\n255\n" % (w, h) + rng.integers(0, 256, (h, w), dtype=np.uint8).tobytes())
        names.append(str(q))
    bad = tmp_path / "bad.pgm"; bad.write_bytes(b"P5\n9 9\n255\nxx")
    names.insert(7, str(bad)); names.append(str(bad))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    # fmt + 4: the mock device hands the writers finished rows (ChunkDone::text / bin) like the GPU formatter does
    for max_batch, dt, wt, fmt in ((1, 1, 1, 1), (4, 3, 3, 3), (64, 8, 2, 2), (3, 2, 8, 1), (4, 2, 3, 7), (2, 1, 2, 5), (8, 2, 2, 6), (4, 3, 2, 16 + 7), (2, 2, 2, 16 + 2)):
        for n_ in names:
            for ext in (".hesaff.sift", ".hesaff.bin"):
                if os.path.exists(n_ + ext):
                    os.remove(n_ + ext)
        r = subprocess.run([exe, str(max_batch), str(dt), str(wt), str(fmt)] + names, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (max_batch, dt, wt, fmt, r.stdout[-500:], r.stderr[-4000:])
        assert "files=47 written=45 unreadable=2 other=0" in r.stdout, r.stdout
        rows = int(r.stdout.strip().rsplit("rows=", 1)[1])
        if fmt & 16:   # readers fill the (mock) context's page-locked buffers: some chunks travel without a staging copy, every buffer comes back
            assert "pinned_chunks=" in r.stdout and "out=0" in r.stdout and "pinned_chunks=0 " not in r.stdout, r.stdout
        fmt &= 3
        got = 0
        for n_ in names:
            if n_ == str(bad):
                continue
            if fmt & 1:
                got += int(open(n_ + ".hesaff.sift", "rb").read().split(b"\n", 2)[1])
            if fmt & 2:
                assert (os.path.getsize(n_ + ".hesaff.bin") - 16) % 148 == 0
        assert (fmt & 1) == 0 or got == rows
    # a long list of one geometry starts and ends with smaller chunks (pipeline fill and drain), FileIO::chunk_limit
    same = []
    for i in range(64):
        q = tmp_path / ("s%02d.pgm" % i)
        q.write_bytes(b"P5\n16 12\n255\n" + rng.integers(0, 256, (12, 16), dtype=np.uint8).tobytes())
        same.append(str(q))
    r = subprocess.run([exe, "8", "2", "2", "2"] + same, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    assert "chunks=2,4,8,8,8,8,8,8,4,4,2\n" in r.stdout and "files=64 written=64" in r.stdout, r.stdout
    # JPEG files as coefficient blobs (hesaff_read_jpeg_coefficients_alloc + FileIO's pool of recycled blobs), mixed with PGM files
    gold = os.path.join(ROOT, "tests", "golden")
    jpgs = [os.path.join(gold, f) for f in sorted(os.listdir(gold)) if f.endswith(".jpg")]
    assert len(jpgs) >= 5
    mixed = []
    for k in range(6):
        for j in jpgs:
            q = tmp_path / ("j%d_%s" % (k, os.path.basename(j)))
            q.write_bytes(open(j, "rb").read())
            mixed.append(str(q))
        mixed.append(same[k])
    r = subprocess.run([exe, "4", "3", "2", str(2 + 4 + 8)] + mixed, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    assert "files=%d written=%d unreadable=0" % (len(mixed), len(mixed)) in r.stdout, r.stdout
    # ArrayIO (hesaff_detect_batch_cb): chunk source + sink hand-over, with a sink that fails mid-run (sink_rc is read by the
    # staging thread while the caller's thread writes it)
    for max_batch, n_img in ((1, 7), (4, 45), (8, 64)):
        r = subprocess.run([exe, "array", str(max_batch), str(n_img)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (max_batch, n_img, r.stdout[-500:], r.stderr[-4000:])
