"""Interleaved A/B of a few creation orders of the compute streams (tuning build, HESAFF_ORDER; see scripts/sweep_stream_pairs.py).
usage on the GPU box: HESAFF_AMD_LIB=hesaff_amd/libhesaff_amd_tuning.so python scripts/ab_stream_orders.py [batch] [rounds] order..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hesaff_amd
from hesaff_amd.synth import band_noise_batch_torch

B = int(sys.argv[1]); rounds = int(sys.argv[2]); orders = sys.argv[3:]
H, W = 2160, 3840
imgs = band_noise_batch_torch(B, H, W, seed=1234, device="cuda")
res = {o: [] for o in orders}
for r in range(rounds):
    for o in (orders if r % 2 == 0 else orders[::-1]):
        os.environ[os.environ.get("AB_VAR", "HESAFF_ORDER")] = o
        p = hesaff_amd.default_params(); p.max_batch = B
        with hesaff_amd.HesaffContext(p, device=0) as ctx:
            ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2):
                ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
            torch.cuda.synchronize()
            res[o].append((time.perf_counter() - t0) / 2 * 1e3)
for o in orders:
    v = res[o]
    print("%s  mean %.1f  min %.1f  max %.1f  %s" % (o, sum(v) / len(v), min(v), max(v), " ".join("%.1f" % x for x in v)), flush=True)
