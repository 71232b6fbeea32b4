import sys, os
sys.path.insert(0, ".")
import torch, hesaff_amd, bench
from hesaff_amd.synth import band_noise_batch_torch
imgs = band_noise_batch_torch(32, 2160, 3840, seed=1234, device="cuda")
host = list(imgs.cpu().numpy()); del imgs
fmt = int(sys.argv[1])
r = bench.file_path_leg(hesaff_amd, host, 3840, 2160, 320, 32, 0, fmt=fmt, decode_threads=2, write_threads=int(sys.argv[2]))
print(fmt, round(r["images_per_s"], 1))
