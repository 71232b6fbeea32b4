#!/usr/bin/env python3
"""bench.py -- keypoints/s + images/s of the Hessian-Affine + SIFT hot path on 4K grey batches.

A "step" is one pass of the whole hot path (grey -> pyramid + det-Hessian -> extrema ->
affine -> patch -> SIFT -> ordered records) over one batch of synthetic 3840x2160 8-bit
images that are already resident in HBM.  One process per GPU; images shard across ranks
with no data-path collective, one all-gather of counts at the end (RCCL).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant pyramid kernel
(k_blur_hess: Gaussian + det-of-Hessian), timed live with HIP events on the library's
stream; `cpu_baseline` is the CPU oracle (a port of the reference, 1 thread) on a bounded
sample of the same images.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(batch, width, height, launches_per_step):
    """HBM bytes per k_blur_hess_march launch (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc
    passes of this same command, see profiles/README.md); None when no matching profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    if (d.get("batch"), d.get("width"), d.get("height")) != (batch, width, height):
        return None
    return d.get("bytes_per_launch_avg")


def _cpu_oracle_worker(img):
    """One image through the CPU oracle in a fresh process (cpu_baseline_multicore); no GPU, no torch."""
    from tests import _oracle
    t0 = time.perf_counter()
    o = _oracle.OracleRun(_oracle.gray_from_u8(img))
    return o.n_keys, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step (BASELINE.json: 2048 UHD images over 8 GPUs)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=1, help="images of the batch timed on the CPU oracle")
    ap.add_argument("--cpu-workers", type=int, default=-1,
                    help="worker processes of the multi-core CPU baseline, one image each (-1: min(32, host cpus / 4); 0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_DIST_BACKEND=gloo (testing only): the multi-rank code path on a box with fewer GPUs than
    # ranks - ranks share the visible devices and the three small collectives run on CPU tensors.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libhesaff_amd has no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    import hesaff_amd
    from hesaff_amd.synth import band_noise_batch_torch

    B, H, W = args.batch, args.height, args.width
    # weak scaling: every rank owns B distinct images (global image index = rank*B + i)
    imgs = band_noise_batch_torch(B, H, W, seed=1234 + rank * B, device=dev)
    torch.cuda.synchronize()

    p = hesaff_amd.default_params()
    p.max_batch = B
    ctx = hesaff_amd.HesaffContext(p, device=local_rank)

    def step():
        return ctx.detect_batch_device(imgs.data_ptr(), B, W, H)

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    bh_ms = 0.0; bh_bytes = 0.0; bh_launches = 0
    stage = {"pyramid_ms": 0.0, "detect_ms": 0.0, "affine_ms": 0.0, "patch_ms": 0.0, "sift_ms": 0.0, "pack_ms": 0.0}
    n_desc = 0; n_hess = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ch, cd, _, total = step()
        n_desc += int(cd.sum()); n_hess += int(ch.sum())
        tm = ctx.timings()
        bh_ms += tm.blur_hess_ms; bh_bytes += tm.blur_hess_bytes; bh_launches += tm.blur_hess_launches
        for k in stage:
            stage[k] += getattr(tm, k)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    from hesaff_amd.shard import gather_counts
    counts = gather_counts([n_hess, n_desc, B * args.steps], device=coll_dev if world > 1 else None)
    tot_hess, tot_desc, tot_imgs = [int(v) for v in counts.sum(axis=0)]

    if rank == 0:
        achieved = (bh_bytes / 1e9) / (bh_ms / 1e3) if bh_ms > 0 else 0.0
        out = {
            "metric": "keypoints/sec (descriptors written), 4K grayscale batch",
            "value": tot_desc / dt,
            "unit": "keypoints/s",
            "images_per_s": tot_imgs / dt,
            "hessian_keypoints_per_s": tot_hess / dt,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "batch of %d x %dx%d 8-bit grey images per GPU per step, band-noise synthetic, default params" % (B, W, H),
                       "images_per_gpu_per_step": B, "width": W, "height": H, "sharding": "image-level, %d rank(s)" % world,
                       "descriptors_per_image": tot_desc / max(tot_imgs, 1)},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage.items()},
            "roofline": {"bound": "hbm",
                         "kernel": "k_blur_hess_march (Gaussian + det-of-Hessian; 4 launches per octave = 58 B/px algorithmic: "
                                   "12 B/px each + 8 B/px for the fused R0 + 2 B/px for the fused decimation)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(B, W, H, bh_launches // max(args.steps, 1)), "launches": bh_launches,
                         "avg_launch_ms": bh_ms / max(bh_launches, 1), "bytes_per_launch_avg": bh_bytes / max(bh_launches, 1)},
        }
        if not args.no_cpu_baseline and world == 1:
            from tests import _oracle
            host = imgs[: args.cpu_images].cpu().numpy()
            t1 = time.perf_counter()
            nk = 0
            for i in range(len(host)):
                o = _oracle.OracleRun(_oracle.gray_from_u8(host[i]))
                nk += o.n_keys
            cdt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": nk / cdt, "unit": "keypoints/s", "cores": 1, "kind": "port",
                                   "images_per_s": len(host) / cdt,
                                   "sample": "%d of the %d batch images (%dx%d), oracle/libhesaff_oracle.so, 1 thread, %.1f s"
                                             % (len(host), B, W, H, cdt),
                                   "host_cpus": os.cpu_count()}
            # the same oracle, one worker process per image over W cores
            nw = args.cpu_workers if args.cpu_workers >= 0 else min(32, max(1, (os.cpu_count() or 4) // 4))
            nw = min(nw, B)
            if nw > 1:
                import multiprocessing as mp
                sample = [imgs[i].cpu().numpy() for i in range(nw)]
                t1 = time.perf_counter()
                with mp.get_context("spawn").Pool(nw) as pool:
                    res = pool.map(_cpu_oracle_worker, sample)
                mdt = time.perf_counter() - t1
                out["cpu_baseline_multicore"] = {"value": sum(r[0] for r in res) / mdt, "unit": "keypoints/s", "cores": nw, "kind": "port",
                                                 "images_per_s": nw / mdt,
                                                 "sample": "%d of the %d batch images, one oracle process each, %.1f s wall (slowest worker %.1f s)"
                                                           % (nw, B, mdt, max(r[1] for r in res))}
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
