"""One process with N device contexts against N processes with one context each, on the SAME device and list: what the runtime's per-process
event thread costs.  usage (on the GPU box): python scripts/ranks_vs_contexts.py [n_files] [N ...]
Both forms run `hesaff --batch` to binary sidecars with the same host share per context (--host-share); wall seconds and the children's CPU seconds
(getrusage) per image."""
import os
import resource
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
from hesaff_amd.synth import band_noise_batch_torch  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 512
counts = [int(a) for a in sys.argv[2:]] or [2, 4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "hesaff_amd", "bin", "hesaff")
imgs = band_noise_batch_torch(64, 2160, 3840, seed=1234, device="cuda")
host = list(imgs.cpu().numpy())
del imgs
torch.cuda.empty_cache()
tmp, paths, _ = bench._e2e_files(host, 3840, 2160, n_files, 32, 1)


def cpu_children():
    r = resource.getrusage(resource.RUSAGE_CHILDREN)
    return r.ru_utime + r.ru_stime


def clean():
    for p in paths:
        for ext in (".hesaff.bin", ".hesaff.sift"):
            try:
                os.remove(p + ext)
            except OSError:
                pass


def lists(n):
    out = []
    for r in range(n):
        lo, hi = len(paths) * r // n, len(paths) * (r + 1) // n
        q = os.path.join(tmp, "list_%d_%d.txt" % (n, r))
        open(q, "w").write("\n".join(paths[lo:hi]) + "\n")
        out.append(q)
    return out


try:
    whole = os.path.join(tmp, "list_all.txt")
    open(whole, "w").write("\n".join(paths) + "\n")
    subprocess.run([exe, "--batch", lists(8)[0], "--devices", "0", "--output", "bin"], capture_output=True)   # warm the page cache and the box
    for n in counts:
        clean()
        c0, t0 = cpu_children(), time.perf_counter()
        r = subprocess.run([exe, "--batch", whole, "--devices", ",".join(["0"] * n), "--output", "bin"], capture_output=True, text=True)
        dt, cpu = time.perf_counter() - t0, cpu_children() - c0
        assert r.returncode == 0, r.stderr[-500:]
        print("one process, %d contexts on device 0:      %6.1f images/s  %.2f s wall  CPU %.2f s = %.2f ms per image" % (n, n_files / dt, dt, cpu, 1e3 * cpu / n_files))
        clean()
        c0, t0 = cpu_children(), time.perf_counter()
        ps = [subprocess.Popen([exe, "--batch", q, "--devices", "0", "--output", "bin", "--host-share", str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for q in lists(n)]
        rcs = [p.wait() for p in ps]
        dt, cpu = time.perf_counter() - t0, cpu_children() - c0
        assert not any(rcs), rcs
        print("%d processes, one context each on device 0: %6.1f images/s  %.2f s wall  CPU %.2f s = %.2f ms per image" % (n, n_files / dt, dt, cpu, 1e3 * cpu / n_files))
finally:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
