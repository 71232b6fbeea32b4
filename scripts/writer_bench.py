"""Host-side throughput of the .hesaff.sift text export: rows/s and MB/s
of hesaff_format_sift_mt for one UHD image's worth of rows, and of hesaff_write_sift_batch."""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hesaff_amd

L = hesaff_amd.load_library()
rng = np.random.default_rng(1)
n = 117000
keys = np.zeros(n, hesaff_amd.KEYPOINT_DTYPE)
keys["x"] = rng.uniform(0, 3840, n); keys["y"] = rng.uniform(0, 2160, n); keys["s"] = rng.uniform(1, 30, n)
keys["a11"] = rng.uniform(0.5, 2, n); keys["a21"] = rng.uniform(-1, 1, n); keys["a22"] = 1.0 / keys["a11"]
keys["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
mr = C.c_float(hesaff_amd.default_params().mrSize)
print("host cpus:", os.cpu_count())
for t in (1, 2, 4, 8, 16, 32, 64):
    best = 1e9
    for rep in range(4):
        buf = C.c_void_p(); ln = C.c_size_t()
        t0 = time.perf_counter()
        L.hesaff_format_sift_mt(keys.ctypes.data, n, mr, t, C.byref(buf), C.byref(ln))
        dt = time.perf_counter() - t0
        L.hesaff_free(buf)
        best = min(best, dt)
    print("format_sift_mt threads %2d: %6.1f ms  %5.2f M rows/s  %6.0f MB/s" % (t, best * 1e3, n / best / 1e6, ln.value / best / 1e6))
nimg = 64
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
paths = [os.path.join(d, "img%03d.ppm.hesaff.sift" % i) for i in range(nimg)]
for t in (1, 8, 32, 64):
    t0 = time.perf_counter()
    hesaff_amd.write_sift_batch(paths, [keys] * nimg, mr.value, threads=t)
    dt = time.perf_counter() - t0
    print("write_sift_batch %d images, threads %2d: %6.1f ms  %5.1f images/s  %5.2f M rows/s" % (nimg, t, dt * 1e3, nimg / dt, nimg * n / dt / 1e6))
for p in paths:
    os.remove(p)
os.rmdir(d)
