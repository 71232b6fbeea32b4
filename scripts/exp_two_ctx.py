"""Experiment: N contexts on one GPU, each fed 1/N of the batch from its own host thread."""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hesaff_amd
from hesaff_amd.synth import band_noise_batch_torch
B, H, W = int(os.environ.get("B", "256")), 2160, 3840
dev = torch.device("cuda", 0)
imgs = band_noise_batch_torch(B, H, W, seed=1234, device=dev)
torch.cuda.synchronize()
for nctx in (1, 2, 1, 2):
    per = B // nctx
    ctxs = []
    for i in range(nctx):
        p = hesaff_amd.default_params(); p.max_batch = per
        # tuning build: the second context's streams at another priority = hardware queues of their own (CTX_PRIOS="0,1")
        os.environ["HESAFF_CTX_PRIO"] = os.environ.get("CTX_PRIOS", "0,0").split(",")[i]
        ctxs.append(hesaff_amd.HesaffContext(p, device=0))
    res = [0] * nctx
    def work(i, steps):
        n = 0
        for _ in range(steps):
            ch, cd, _, total = ctxs[i].detect_batch_device(imgs[i * per:(i + 1) * per].data_ptr(), per, W, H)
            n += int(cd.sum())
        res[i] = n
    def run(steps):
        th = [threading.Thread(target=work, args=(i, steps)) for i in range(nctx)]
        for t in th: t.start()
        for t in th: t.join()
    run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 3
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("contexts %d: %.1f ms per %d images, %.2fM kp/s" % (nctx, dt / steps * 1e3, B, sum(res) / dt / 1e6), flush=True)
    for c in ctxs: c.close()
