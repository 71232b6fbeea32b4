"""One process per GPU over a list of images: the `torchrun` form of `hesaff --batch <list> --devices ...`.

    python tools/batch_ranks.py list.txt                                   # one process, device 0
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/batch_ranks.py list.txt

Rank r takes the contiguous block shard_range(n, r, world) of the list (the rule of hesaff_shard_range and of the
multi-device CLI), runs it through hesaff_process_files on device LOCAL_RANK (--one-device: every rank on device 0, for a box with
one GPU), which writes <image>.hesaff.sift next to every image (hesaff.cpp:170-176).  No feature data crosses ranks; the only collective
is the all-gather of [Hessian keypoints, descriptors, images] (RCCL; BENCH_DIST_BACKEND=gloo moves it to CPU tensors).
Rank 0 prints one JSON line with the totals."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hesaff_amd  # noqa: E402
from hesaff_amd.shard import gather_counts, shard_range  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("list")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--chunk", type=int, default=32, help="images per device chunk (hesaff_params.max_batch)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(0 if args.one_device else local)
        dist.init_process_group(backend=backend)
    names = [ln.strip() for ln in open(args.list) if ln.strip() and not ln.startswith("#")]
    lo, hi = shard_range(len(names), rank, world)
    p = hesaff_amd.default_params()
    p.max_batch = max(1, args.chunk)
    ctx = hesaff_amd.HesaffContext(p, device=0 if args.one_device else local)
    # the rank's shard through hesaff_process_files (what `hesaff --batch` runs per device): decode ahead, device, rows formatted
    # on the device, write behind - with this rank's share of the host threads
    hp = hesaff_amd.host_plan(max(world, 1))   # the library's one rule (hesaff_host_plan_for): this rank's share of the host
    st = ctx.process_files(names[lo:hi], decode_threads=hp["decode_threads"], write_threads=hp["write_threads"])
    bad = [names[lo + i] for i, s_ in enumerate(st) if s_[0] != 0]
    if bad:
        print("rank %d: %d file(s) failed, first: %s" % (rank, len(bad), bad[0]), file=sys.stderr)
    # totals over the files that were written in THIS run (stage 3); a skipped file (resume) carries count_hessian = -1
    nh = sum(max(0, s_[2]) for s_ in st if s_[0] == 0 and s_[1] == 3); nd = sum(max(0, s_[3]) for s_ in st if s_[0] == 0 and s_[1] == 3)
    device = None
    if world > 1 and backend == "nccl":
        import torch
        device = torch.device("cuda", 0 if args.one_device else local)
    tot = gather_counts([nh, nd, hi - lo, len(bad)], device=device)
    failed = int(tot[:, 3].sum())
    if rank == 0:
        print(json.dumps({"world": world, "images": int(tot[:, 2].sum()), "hessian_keypoints": int(tot[:, 0].sum()),
                          "descriptors": int(tot[:, 1].sum()), "per_rank_images": [int(x) for x in tot[:, 2]], "failed_files": failed,
                          "host_plan_per_rank": hp}))
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    # every rank leaves with the same status, after the collective: a job with unreadable or rejected images is not a success
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
