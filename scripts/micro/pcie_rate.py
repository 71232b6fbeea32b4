import torch, time
for n in (256<<20, 1<<30, 2<<30):
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, f in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        print(name, n >> 20, "MiB: %.1f GB/s" % (3 * n / (time.perf_counter() - t0) / 1e9), flush=True)
    # both directions at once on two streams
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
        with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
    torch.cuda.synchronize()
    print("both", n >> 20, "MiB: %.1f GB/s each way" % (3 * n / (time.perf_counter() - t0) / 1e9), flush=True)
    del h, d, h2, d2
