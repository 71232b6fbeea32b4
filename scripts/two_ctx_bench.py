"""Host path with one and with two contexts on the same device (two host threads, alternate halves of the batch):
do the pyramid / detection of one context and the per-keypoint stages of the other overlap on the GPU?

    python scripts/two_ctx_bench.py [--batch 128]"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import hesaff_amd
from hesaff_amd.synth import band_noise_batch_torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--chunk", type=int, default=64)
    args = ap.parse_args()
    B = args.batch
    imgs = band_noise_batch_torch(B, 2160, 3840, seed=1234, device="cuda")
    host = list(imgs.cpu().numpy())
    del imgs
    torch.cuda.empty_cache()

    def make():
        p = hesaff_amd.default_params(); p.max_batch = args.chunk
        return hesaff_amd.HesaffContext(params=p)

    def run(ctx, part, out, k):
        res = ctx.detect_batch_raw(part)
        out[k] = sum(r.count_desc for r in res)

    for nctx in (1, 2, 3):
        ctxs = [make() for _ in range(nctx)]
        parts = [host[i::nctx] for i in range(nctx)]   # interleaved halves: equal work
        for rep in range(2):   # first repetition: warm-up (buffers)
            out = [0] * nctx
            t0 = time.perf_counter()
            th = [threading.Thread(target=run, args=(ctxs[k], parts[k], out, k)) for k in range(nctx)]
            for t in th: t.start()
            for t in th: t.join()
            dt = time.perf_counter() - t0
        print("contexts %d: %.1f images/s (%.0f ms, %d descriptors)" % (nctx, B / dt, dt * 1e3, sum(out)), flush=True)
        for c in ctxs: c.close()


if __name__ == "__main__":
    main()
