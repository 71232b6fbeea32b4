"""End-to-end file path (hesaff_process_files: PGM files on a RAM disk -> .hesaff.sift / .hesaff.bin files) against the size of its
host-thread pool: usage (on the GPU box): python scripts/e2e_thread_sweep.py [n_files]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
import hesaff_amd  # noqa: E402
from hesaff_amd.synth import band_noise_batch_torch  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 384
imgs = band_noise_batch_torch(64, 2160, 3840, seed=1234, device="cuda")
host = list(imgs.cpu().numpy())
del imgs
rows = []
for dt, wt in ((1, 1), (2, 2), (2, 4), (4, 4), (4, 12)):
    for fmt in (1, 2):
        r = bench.file_path_leg(hesaff_amd, host, 3840, 2160, n_files, 32, 0, fmt=fmt, decode_threads=dt, write_threads=wt)
        rows.append({"decode_threads": dt, "write_threads": wt, "pool": dt + wt, "output": "text" if fmt == 1 else "sidecar",
                     "images_per_s": r.get("images_per_s"), "output_GB_per_s": r.get("output_GB_per_s")})
        print(rows[-1], flush=True)
print(json.dumps({"what": "hesaff_process_files on %d UHD PGM files, chunks of 32, one pool of decode + write threads (host quota: %d CPUs)"
                          % (n_files, hesaff_amd.load_library().hesaff_host_threads()), "rows": rows}))
