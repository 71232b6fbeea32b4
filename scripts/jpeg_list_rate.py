# usage (through gpurun): python scripts/jpeg_list_rate.py [files=512] [width=3840] [height=2160]
# hesaff_process_files on a list of colour JPEG photographs (mosaics of the two sample photographs, 4:2:0, quality 90, on a RAM disk)
# (JPEG_SUBSAMPLING=0|1|2, JPEG_PROGRESSIVE=1 for other encodings) with 2+2, 4+4 and 8+8 host threads: images/s with the pixels made on the device (the product) and - tuning build,
# HESAFF_DEVICE_JPEG=0 - with the whole decode on the host threads.  One JSON line per case.
import json, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def one_case(paths, dt, wt, fmt):
    import hesaff_amd
    p = hesaff_amd.default_params()
    p.max_batch = int(os.environ.get("JPEG_CHUNK", "32"))   # images per device chunk
    ctx = hesaff_amd.HesaffContext(p, device=0)
    ctx.set_output_format(fmt)
    ctx.process_files(paths[:64], decode_threads=dt, write_threads=wt)   # warm-up: plan, pinned blocks
    for q in paths:
        for ext in (".hesaff.sift", ".hesaff.bin"):
            if os.path.exists(q + ext):
                os.remove(q + ext)
    t = time.perf_counter()
    st = ctx.process_files(paths, decode_threads=dt, write_threads=wt)
    dt_s = time.perf_counter() - t
    ctx.close()
    assert all(s[0] == 0 for s in st), [s for s in st if s[0] != 0][:3]
    return {"images": len(paths), "chunk_images": p.max_batch, "decode_threads": dt, "write_threads": wt, "format": "text" if fmt == 1 else "sidecar",
            "device_jpeg": os.environ.get("HESAFF_DEVICE_JPEG", "1") != "0", "images_per_s": len(paths) / dt_s,
            "descriptors": int(sum(s[3] for s in st))}

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--case":
        paths = [ln.strip() for ln in open(sys.argv[2])]
        print(json.dumps(one_case(paths, int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
    from PIL import Image
    from hesaff_amd import synth
    tmp = tempfile.mkdtemp(prefix="hesaff_jpg_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        paths = []
        photos = synth.load_sample_photos()
        for i in range(n):   # 32 distinct mosaics, cycled
            q = os.path.join(tmp, "p%04d.jpg" % i)
            if i < 32:
                Image.fromarray(synth.photo_mosaic(H, W, i, photos=photos)).save(q, quality=90, subsampling=int(os.environ.get("JPEG_SUBSAMPLING", "2")),
                                                                               progressive=os.environ.get("JPEG_PROGRESSIVE", "0") == "1")
            else:
                shutil.copyfile(paths[i % 32], q)
            paths.append(q)
        lst = os.path.join(tmp, "list.txt")
        open(lst, "w").write("\n".join(paths) + "\n")
        print(json.dumps({"files": n, "width": W, "height": H, "bytes_per_file": os.path.getsize(paths[0]),
                          "subsampling": {"0": "4:4:4", "1": "4:2:2", "2": "4:2:0"}[os.environ.get("JPEG_SUBSAMPLING", "2")],
                          "progressive": os.environ.get("JPEG_PROGRESSIVE", "0") == "1"}))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        tuning = os.path.join(root, "hesaff_amd", "libhesaff_amd_tuning.so")
        for dev in ("1", "0"):
            for dt, wt in ((2, 2), (4, 4), (8, 8)):
                env = dict(os.environ, HESAFF_AMD_LIB=tuning, HESAFF_DEVICE_JPEG=dev)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--case", lst, str(dt), str(wt), "2"], env=env, capture_output=True, text=True)
                print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED %s" % r.stderr[-400:])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
