"""Host entry point (hesaff_detect_batch: H2D + run + D2H) against the device-resident one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hesaff_amd
from hesaff_amd.synth import band_noise_batch_torch
N, MB = int(os.environ.get("N", "128")), int(os.environ.get("MB", "32"))
dev = torch.device("cuda", 0)
imgs = band_noise_batch_torch(N, 2160, 3840, seed=1234, device=dev)
host = [np.ascontiguousarray(imgs[i].cpu().numpy()) for i in range(N)]
p = hesaff_amd.default_params(); p.max_batch = MB
ctx = hesaff_amd.HesaffContext(p, device=0)
ctx.detect_batch(host[:MB])
import ctypes as C
from hesaff_amd import _binding
for rep in range(2):
    t0 = time.perf_counter(); res = ctx.detect_batch(host); dt = time.perf_counter() - t0
    nk = sum(len(k) for _, k in res)
    print("host path (python binding, results copied to numpy): %d images in %.0f ms -> %.1f images/s, %.2f M kp/s" % (N, dt * 1e3, N / dt, nk / dt / 1e6), flush=True)
# the C entry point alone
ptrs = (C.c_void_p * N)(*[im.ctypes.data for im in host]); ws = (C.c_int * N)(*[3840] * N); hs = (C.c_int * N)(*[2160] * N)
st = (C.c_int * N)(*[3840] * N); chs = (C.c_int * N)(*[1] * N); resa = (_binding._Result * N)()
for rep in range(2):
    t0 = time.perf_counter(); rc = ctx.L.hesaff_detect_batch(ctx.h, N, ptrs, ws, hs, st, chs, resa); dt = time.perf_counter() - t0
    nk = sum(r.count_desc for r in resa)
    print("host path (C entry point): rc %d, %d images in %.0f ms -> %.1f images/s, %.2f M kp/s" % (rc, N, dt * 1e3, N / dt, nk / dt / 1e6), flush=True)
t0 = time.perf_counter(); n = 0
for c in range(0, N, MB):
    ch, cd, _, tot = ctx.detect_batch_device(imgs[c:c + MB].data_ptr(), MB, 3840, 2160); n += int(cd.sum())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("device path: %d images in %.0f ms -> %.1f images/s, %.2f M kp/s" % (N, dt * 1e3, N / dt, n / dt / 1e6))
ctx.close()
