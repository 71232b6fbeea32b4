"""Practical HBM bandwidth of this box for simple streaming patterns (torch kernels)."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 30   # 4 GB of float32
a = torch.empty(n, device=dev, dtype=torch.float32).normal_()
b = torch.empty_like(a)
c = torch.empty_like(a)
def t(f, bytes_, name, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-28s %7.2f ms  %6.2f TB/s" % (name, dt * 1e3, bytes_ / dt / 1e12), flush=True)
t(lambda: b.copy_(a), 8.0 * n, "copy (1R + 1W)")
t(lambda: b.fill_(1.0), 4.0 * n, "fill (1W)")
t(lambda: a.sum(), 4.0 * n, "sum (1R)")
t(lambda: torch.add(a, b, out=c), 12.0 * n, "add (2R + 1W)")
t(lambda: torch.sincos if False else (b.copy_(a), c.copy_(a)), 16.0 * n, "2 copies (2R + 2W)")
