"""Long-list soak of hesaff_process_files: N files (hard links to a few distinct images on a RAM disk) -> binary sidecars;
reports the rate, the per-file status histogram and the peak resident set of this process (host memory must stay bounded
however long the list is: about 2 max_batch decoded images + three pinned result blocks).

    python scripts/soak_process_files.py [--files 2048] [--width 1920 --height 1080] [--chunk 64]"""
import argparse
import os
import resource
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=2048)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--format", type=int, default=2, help="1 text, 2 binary sidecar, 3 both")
    ap.add_argument("--jpeg", action="store_true", help="colour JPEG files (mosaics of the sample photographs, 4:2:0, quality 90) instead of PGM: coefficient blobs, pixels on the device")
    a = ap.parse_args()
    import torch
    import hesaff_amd
    from hesaff_amd.synth import band_noise_batch_torch
    imgs = band_noise_batch_torch(a.distinct, a.height, a.width, seed=4321, device="cuda").cpu().numpy()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="hesaff_soak_", dir=base)
    try:
        hdr = b"P5\n%d %d\n255\n" % (a.width, a.height)
        src = []
        ext = "jpg" if a.jpeg else "pgm"
        if a.jpeg:
            from PIL import Image
            from hesaff_amd import synth
            photos = synth.load_sample_photos()
        for i in range(a.distinct):
            q = os.path.join(tmp, "src%03d.%s" % (i, ext))
            if a.jpeg:
                Image.fromarray(synth.photo_mosaic(a.height, a.width, i, photos=photos)).save(q, quality=90, subsampling=2)
            else:
                with open(q, "wb") as f:
                    f.write(hdr); f.write(imgs[i].tobytes())
            src.append(q)
        paths = []
        for i in range(a.files):
            q = os.path.join(tmp, "img%05d.%s" % (i, ext))
            os.link(src[i % a.distinct], q)
            paths.append(q)
        del imgs
        p = hesaff_amd.default_params()
        p.max_batch = a.chunk
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        with hesaff_amd.HesaffContext(p, device=0) as ctx:
            ctx.set_output_format(a.format)
            ctx.process_files(paths[: 2 * a.chunk])
            rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
            t0 = time.perf_counter()
            st = ctx.process_files(paths)
            dt = time.perf_counter() - t0
        rss2 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        ok = sum(1 for s in st if s[0] == 0 and s[1] == 3)
        rows = sum(s[3] for s in st)
        out_bytes = sum(os.path.getsize(q + e) for q in paths for e in ((".hesaff.sift",) if a.format == 1 else (".hesaff.bin",) if a.format == 2 else (".hesaff.sift", ".hesaff.bin")))
        print({"files": a.files, "input": "colour JPEG" if a.jpeg else "PGM", "size": "%dx%d" % (a.width, a.height), "chunk": a.chunk, "written": ok, "images_per_s": a.files / dt,
               "descriptors_per_s": rows / dt, "seconds": dt, "output_GB": out_bytes / 1e9,
               "peak_rss_GB_before": rss0 / 1e6, "peak_rss_GB_after_warmup_of_%d_files" % (2 * a.chunk): rss1 / 1e6,
               "peak_rss_GB_after_%d_files" % a.files: rss2 / 1e6})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
