import sys, os, tempfile, shutil, time
sys.path.insert(0, ".")
import numpy as np, torch, hesaff_amd
from hesaff_amd.synth import band_noise_batch_torch
imgs = band_noise_batch_torch(8, 2160, 3840, seed=1234, device="cuda").cpu().numpy()
tmp = tempfile.mkdtemp(prefix="big_", dir="/dev/shm")
try:
    paths = []
    for i in range(128):
        q = os.path.join(tmp, "i%03d.pgm" % i)
        open(q, "wb").write(b"P5\n3840 2160\n255\n" + imgs[i % 8].tobytes())
        paths.append(q)
    p = hesaff_amd.default_params(); p.max_batch = 128
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        t0 = time.time(); st = ctx.process_files(paths, decode_threads=4, write_threads=4); dt = time.time() - t0
        assert all(s[0] == 0 and s[1] == 3 for s in st), [s for s in st if s[0] != 0][:3]
        total = sum(os.path.getsize(q + ".hesaff.sift") for q in paths)
        print("one chunk of 128 UHD images: %.2f GB of text in one device buffer, %.1f s" % (total / 1e9, dt))
        ref = ctx.detect_batch([imgs[5]])[0][1]
        want = hesaff_amd.format_sift(ref, ctx.params.mrSize)
    for k in (5, 13, 125):
        assert open(paths[k] + ".hesaff.sift", "rb").read() == want, k
    assert open(paths[127] + ".hesaff.sift", "rb").read() == open(paths[7] + ".hesaff.sift", "rb").read()
    print("ok")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
