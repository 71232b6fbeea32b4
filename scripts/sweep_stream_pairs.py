"""Which logical streams of a context should share a HIP stream (= a hardware queue: the runtime has four per priority, and kernels on one
queue do not overlap)?  This sweeps all 105 ways to pair the eight logical streams (0 main, 1-4 patch bins 0-3, 5 descriptor,
6 descriptor 2, 7 affine) through the tuning build's HESAFF_GROUPS (eight digits: the group of each logical stream) and times the step.
usage on the GPU box: HESAFF_AMD_LIB=hesaff_amd/libhesaff_amd_tuning.so python scripts/sweep_stream_pairs.py [batch] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hesaff_amd
from hesaff_amd.synth import band_noise_batch_torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 2160, 3840
imgs = band_noise_batch_torch(B, H, W, seed=1234, device="cuda")


def pairings(items):
    if not items:
        yield []
        return
    a = items[0]
    for i in range(1, len(items)):
        rest = items[1:i] + items[i + 1:]
        for p in pairings(rest):
            yield [(a, items[i])] + p


def run(order):
    os.environ["HESAFF_GROUPS"] = order
    p = hesaff_amd.default_params(); p.max_batch = B
    with hesaff_amd.HesaffContext(p, device=0) as ctx:
        ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3


names = ["main", "bin0", "bin1", "bin2", "bin3", "desc", "desc2", "affine"]
res = []
base = run("01120332")
for pr in pairings(list(range(8))):
    g = [0] * 8
    for gi, (a, b) in enumerate(pr):
        g[a] = gi; g[b] = gi
    order = "".join(str(x) for x in g)
    ms = run(order)
    res.append((ms, order, pr))
    print("%.1f ms  %s  %s" % (ms, order, " | ".join("%s+%s" % (names[a], names[b]) for a, b in pr)), flush=True)
res.sort()
print("grouping in use 01120332: %.1f ms (again: %.1f)" % (base, run("01120332")))
print("best five:")
for ms, order, pr in res[:5]:
    print("  %.1f ms  %s  %s" % (ms, order, " | ".join("%s+%s" % (names[a], names[b]) for a, b in pr)))
print("worst: %.1f ms %s" % (res[-1][0], res[-1][1]))
json.dump({"batch": B, "steps": steps, "default_ms": base, "results": [{"ms": ms, "order": o} for ms, o, _ in res]}, open("gpurun_out/stream_pairs.json", "w"))
