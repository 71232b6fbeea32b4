"""Does a device-to-host copy on its own stream slow the kernels of a batch down?  
usage on the GPU box: python scripts/micro/copy_vs_compute.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hesaff_amd
from hesaff_amd.synth import band_noise_batch_torch

B, H, W = 32, 2160, 3840
imgs = band_noise_batch_torch(B, H, W, seed=1234, device="cuda")
p = hesaff_amd.default_params(); p.max_batch = B
ctx = hesaff_amd.HesaffContext(p, device=0)
ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
n = int(1.34e9)
dsrc = torch.empty(n, dtype=torch.uint8, device="cuda"); hdst = torch.empty(n, dtype=torch.uint8).pin_memory()
hsrc = torch.empty(n // 5, dtype=torch.uint8).pin_memory(); ddst = torch.empty(n // 5, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()


def steps(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


print("kernels alone: %.1f ms per batch of %d" % (steps(6), B))
for what in ("D2H 1.34 GB per batch", "H2D 0.27 GB per batch"):
    stop = False

    def copier():
        with torch.cuda.stream(side):
            while not stop:
                if what.startswith("D2H"):
                    hdst.copy_(dsrc, non_blocking=True)
                else:
                    ddst.copy_(hsrc, non_blocking=True)
                side.synchronize()
                time.sleep(0.08)   # about one copy per batch, like the pipeline
    th = threading.Thread(target=copier); th.start()
    print("with %-24s %.1f ms per batch" % (what + ":", steps(8)))
    stop = True; th.join()
ctx.close()
