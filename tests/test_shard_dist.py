"""CPU suite: the N>1 path -- image sharding + the one collective (count all-gather) over
gloo with world_size 2, and shard invariance of the partition."""
import os
import socket
import sys

import numpy as np
import pytest

from hesaff_amd.shard import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in [0, 1, 7, 8, 255, 256, 2048]:
        for world in [1, 2, 3, 4, 8]:
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(2048, 3, 8) == (768, 1024)     # config 4: 256 images per GPU
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_c_abi_shard_range_equals_python():
    """hesaff_shard_range (C ABI, used by `hesaff --batch --devices`) is the same partition as shard.py's."""
    import ctypes as C
    import hesaff_amd
    L = hesaff_amd.load_library()
    lo = C.c_int(); hi = C.c_int()
    for n in [0, 1, 7, 8, 255, 256, 2048, 100003]:
        for world in [1, 2, 3, 4, 8]:
            for r in range(world):
                assert L.hesaff_shard_range(n, r, world, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_range(n, r, world)
    assert L.hesaff_shard_range(4, 4, 4, C.byref(lo), C.byref(hi)) == -2
    assert L.hesaff_device_count() >= 0


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from hesaff_amd.shard import gather_counts, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_images = 11
        lo, hi = shard_range(n_images, rank, world)
        # stand-in per-image counts that depend only on the global image index
        per_image = [(1000 + 7 * i, 900 + 5 * i) for i in range(lo, hi)]
        local = [sum(p[0] for p in per_image), sum(p[1] for p in per_image), hi - lo]
        allc = gather_counts(local)
        q.put((rank, allc.tolist()))
    finally:
        dist.destroy_process_group()


def test_gather_counts_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                      # every rank sees the same table
    tab = np.array(res[0])
    assert tab.shape == (2, 3) and tab[:, 2].sum() == 11
    assert tab[:, 0].sum() == sum(1000 + 7 * i for i in range(11))
    assert tab[:, 1].sum() == sum(900 + 5 * i for i in range(11))


def test_gather_counts_single_process_identity():
    from hesaff_amd.shard import gather_counts
    out = gather_counts([3, 2, 1])
    assert out.shape == (1, 3) and out[0].tolist() == [3, 2, 1]


def test_bench_refuses_a_world_size_other_than_gpus():
    """bench.py --gpus N under a launcher that started another number of ranks must fail loudly (before it needs a GPU)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "--gpus 3" in r.stderr and "2 rank" in r.stderr
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "--gpus 8" in r.stderr
