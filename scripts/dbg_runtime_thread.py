"""Which work makes the HIP runtime's own thread (CPU mask 0-255) burn CPU?  usage (on the GPU box): python scripts/dbg_runtime_thread.py
Loops of 16 x 32 UHD images, per-thread CPU seconds (user + kernel) around each: (a) kernels only (images resident in HBM),
(b) pinned host -> device copies only (512 x 8.3 MB through torch), (c) hesaff_detect_batch from host arrays (copies + kernels + results);
during a second pass of (a) a sampler reads /proc/self/task/<tid>/syscall and the state of the busiest thread every millisecond."""
import collections
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import hesaff_amd  # noqa: E402
from hesaff_amd.synth import band_noise_batch_torch  # noqa: E402


def table(tag, th0, th1, dt):
    rows = sorted(((th1[t][1] - th0.get(t, (0, 0.0, 0, 0.0))[1], th1[t][3] - th0.get(t, (0, 0.0, 0, 0.0))[3], th1[t][0], th1[t][2], t) for t in th1), reverse=True)
    print("%s: %.2f s wall" % (tag, dt))
    for s, k, nm, al, t in rows:
        if s >= 0.02:
            print("      %-16s %6.2f s (%.2f in the kernel)  allowed %s  tid %s" % (nm, s, k, al, t))
    return rows[0][4]


imgs = band_noise_batch_torch(32, 2160, 3840, seed=1234, device="cuda")
host = imgs.cpu().pin_memory()
p = hesaff_amd.default_params()
p.max_batch = 32
with hesaff_amd.HesaffContext(p, device=0) as ctx:
    ctx.detect_batch_device(imgs.data_ptr(), 32, 3840, 2160)
    torch.cuda.synchronize()
    th0, t0 = bench._thread_cpu(), time.perf_counter()
    for _ in range(16):
        ctx.detect_batch_device(imgs.data_ptr(), 32, 3840, 2160)
    torch.cuda.synchronize()
    busy = table("(a) kernels only, 512 images", th0, bench._thread_cpu(), time.perf_counter() - t0)
    seen, stop = collections.Counter(), []

    def sampler():
        while not stop:
            try:
                sc = open("/proc/self/task/%s/syscall" % busy).read().split()
                st = open("/proc/self/task/%s/stat" % busy).read()
                state = st[st.rindex(")") + 2]
                seen[(state, " ".join(sc[:3]))] += 1
            except OSError as e:
                seen[("err", str(e))] += 1
            time.sleep(0.001)

    th = threading.Thread(target=sampler)
    th.start()
    for _ in range(16):
        ctx.detect_batch_device(imgs.data_ptr(), 32, 3840, 2160)
    torch.cuda.synchronize()
    stop.append(1)
    th.join()
    print("samples of thread %s (state, syscall number, first two arguments):" % busy)
    for k, v in seen.most_common(12):
        print("      %6d  %s" % (v, k))
    try:
        print("      its open files:", {fd: os.readlink("/proc/self/fd/%s" % fd) for fd in os.listdir("/proc/self/fd") if "kfd" in os.readlink("/proc/self/fd/%s" % fd) or "dri" in os.readlink("/proc/self/fd/%s" % fd)})
    except OSError:
        pass
    dst = torch.empty_like(imgs)
    th0, t0 = bench._thread_cpu(), time.perf_counter()
    for _ in range(16):
        for i in range(32):
            dst[i].copy_(host[i], non_blocking=True)
        torch.cuda.synchronize()
    table("(b) 512 pinned -> device copies of 8.3 MB", th0, bench._thread_cpu(), time.perf_counter() - t0)
    arr = list(host.numpy())
    ctx.detect_batch_raw(arr)
    th0, t0 = bench._thread_cpu(), time.perf_counter()
    for _ in range(16):
        ctx.detect_batch_raw(arr)
    table("(c) hesaff_detect_batch from host arrays, 512 images", th0, bench._thread_cpu(), time.perf_counter() - t0)
    # (d) / (e): is it the dispatch rate or the cross-stream events?  plain torch launches, no hesaff code
    x = torch.zeros(1 << 24, device="cuda")
    y = torch.zeros(1 << 24, device="cuda")
    torch.cuda.synchronize()
    th0, t0 = bench._thread_cpu(), time.perf_counter()
    for _ in range(4000):
        x.add_(1.0)
    torch.cuda.synchronize()
    table("(d) 4000 launches of a 64 MB add on one stream", th0, bench._thread_cpu(), time.perf_counter() - t0)
    s2 = torch.cuda.Stream()
    th0, t0 = bench._thread_cpu(), time.perf_counter()
    for _ in range(2000):
        x.add_(1.0)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            y.add_(1.0)
    torch.cuda.synchronize()
    table("(e) 2000 + 2000 launches on two streams, an event record + wait per pair", th0, bench._thread_cpu(), time.perf_counter() - t0)
    # (f): the same 4000 launches replayed from a captured graph (40 replays of 100 kernel nodes)
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        x.add_(1.0)
        cs.synchronize()
        with torch.cuda.graph(g, stream=cs):
            for _ in range(100):
                x.add_(1.0)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    th0, t0 = bench._thread_cpu(), time.perf_counter()
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    table("(f) 40 replays of a graph of 100 such launches", th0, bench._thread_cpu(), time.perf_counter() - t0)
    th0, t0 = bench._thread_cpu(), time.perf_counter()
    for _ in range(400):
        g.replay()
    torch.cuda.synchronize()
    table("(g) 400 replays", th0, bench._thread_cpu(), time.perf_counter() - t0)
