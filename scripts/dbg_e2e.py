# usage (through gpurun): HESAFF_DEBUG=1 python scripts/dbg_e2e.py <format 1|2> <write threads> [files] [chunk]
# the file path of bench.py's end_to_end leg with the chunk engine's per-chunk host / device timings on stderr
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hesaff_amd, bench
from hesaff_amd.synth import band_noise_batch_torch
imgs = band_noise_batch_torch(32, 2160, 3840, seed=1234, device="cuda")
host = list(imgs.cpu().numpy()); del imgs
fmt = int(sys.argv[1])
files = int(sys.argv[3]) if len(sys.argv) > 3 else 512
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 32
r = bench.file_path_leg(hesaff_amd, host, 3840, 2160, files, chunk, 0, fmt=fmt, decode_threads=2, write_threads=int(sys.argv[2]))
print(fmt, files, chunk, round(r["images_per_s"], 1), round(r["seconds"], 3))
