"""`hesaff --batch` inside a two-CPU share (taskset), with and without the runtime's helper threads at nice 19 (--runtime-nice 1 / 0) and with
HSA_OVERRIDE_CPU_AFFINITY_DEBUG=0 (the runtime's threads inside the share): images/s and the process's CPU seconds per image.
usage (on the GPU box): python scripts/cli_share_probe.py [n_files] [cpus]"""
import os
import resource
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
from hesaff_amd.synth import band_noise_batch_torch  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 768
ncpu = int(sys.argv[2]) if len(sys.argv) > 2 else 2
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "hesaff_amd", "bin", "hesaff")
imgs = band_noise_batch_torch(64, 2160, 3840, seed=1234, device="cuda")
host = list(imgs.cpu().numpy())
del imgs
torch.cuda.empty_cache()
tmp, paths, _ = bench._e2e_files(host, 3840, 2160, n_files, 32, 1)
cpus = ",".join(str(c) for c in sorted(os.sched_getaffinity(0))[:ncpu])
share = max(1, len(os.sched_getaffinity(0)) // ncpu)
try:
    lst = os.path.join(tmp, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    subprocess.run([exe, "--batch", lst, "--output", "bin"], capture_output=True)   # warm-up: page cache, the box
    for out in ("text", "bin"):
        for nice in (0, 1):
            for p in paths:
                for ext in (".hesaff.sift", ".hesaff.bin"):
                    if os.path.exists(p + ext):
                        os.remove(p + ext)
            r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
            t0 = time.perf_counter()
            r = subprocess.run(["taskset", "-c", cpus, exe, "--batch", lst, "--output", out, "--host-share", str(share), "--runtime-nice", str(nice)],
                               capture_output=True, text=True, env=dict(os.environ, HSA_OVERRIDE_CPU_AFFINITY_DEBUG="0"))
            dt = time.perf_counter() - t0
            r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            cpu = r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime
            if nice and r.stderr.strip():
                print("   ", r.stderr.strip().splitlines()[-1])
            tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            own = float(tail.split(" images in ")[1].split(" sec")[0]) if " images in " in tail else float("nan")   # the CLI's own clock: contexts + list, no process start
            print("%d CPUs (--host-share %d), output %-4s runtime threads %s: rc %d  %6.1f images/s by the CLI's clock (%.1f with process start)  CPU %.2f ms per image  busy %.2f" %
                  (ncpu, share, out, "nice 19" if nice else "as they are", r.returncode, n_files / own, n_files / dt, 1e3 * cpu / n_files, cpu / dt))
finally:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
