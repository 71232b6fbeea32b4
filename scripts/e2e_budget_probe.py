"""The file path inside a CPU share, with the CPU seconds per thread: usage (on the GPU box): python scripts/e2e_budget_probe.py [n_files] [cpus ...]
For every CPU count given (default 2 4) a child process (bench.py --budgeted-child: CPU mask first, then the library, thread counts from
hesaff_host_plan_for) runs the list to text and to sidecars; prints rate, CPU seconds per image and the per-thread table."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
from hesaff_amd.synth import band_noise_batch_torch  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 384
shares = [int(a) for a in sys.argv[2:]] or [2, 4]
imgs = band_noise_batch_torch(64, 2160, 3840, seed=1234, device="cuda")
host = list(imgs.cpu().numpy())
del imgs
torch.cuda.empty_cache()
tmp, paths, _ = bench._e2e_files(host, 3840, 2160, n_files, 32, 1)
try:
    allowed = sorted(os.sched_getaffinity(0))
    for k in shares:
        cfg = {"dir": tmp, "n": len(paths), "chunk": 32, "device": 0, "cpus": allowed[:k], "rank": 0, "world": 1}
        if os.environ.get("PROBE_POOL"):   # e.g. PROBE_POOL=1,2: decode and write threads instead of the rule's
            cfg["pool"] = [int(v) for v in os.environ["PROBE_POOL"].split(",")]
        if os.environ.get("PROBE_RENICE"):   # experiment: the runtime's helper threads at this nice value (bench.py: budgeted_child)
            cfg["renice_runtime"] = int(os.environ["PROBE_RENICE"])
        if os.environ.get("PROBE_UNCONFINED_RUNTIME"):
            cfg["confine_runtime"] = False
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "bench.py"), "--budgeted-child", json.dumps(cfg)],
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print("child failed", r.stderr[-1000:])
            continue
        if os.environ.get("HESAFF_DEBUG"):   # tuning build: the chunk engine's per-chunk log of the sidecar run (the last 14 chunks)
            print("\n".join([ln for ln in r.stderr.splitlines() if "] chunk" in ln and "caller" not in ln][-14:]))
        d = json.loads(r.stdout.strip().splitlines()[-1])
        for name in ("text", "sidecar"):
            q = d[name]
            print("cpus %d plan %s %-7s %6.1f images/s  cpu %.2f ms/image  busy %.2f  exited threads (pool + staging) %.2f s of %.2f s" %
                  (k, d["plan"], name, q["images_per_s"], 1e3 * q["cpu_seconds_per_image"], q["cpus_busy"], q["threads_exited_cpu_seconds"], q["cpu_seconds"]))
            for t in q["threads_cpu_seconds"][:8]:
                print("      %-16s %6.2f s (%.2f in the kernel)  allowed %s" % (t["name"], t["cpu_seconds"], t.get("in_kernel_seconds", 0.0), t["allowed"]))
finally:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
