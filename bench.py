#!/usr/bin/env python3
"""bench.py -- keypoints/s + images/s of the Hessian-Affine + SIFT hot path on 4K grey batches.

A "step" is one pass of the whole hot path (grey -> pyramid + det-Hessian -> extrema ->
affine -> patch -> SIFT -> ordered records) over one batch of synthetic 3840x2160 8-bit
images.  One process per GPU; images shard across ranks with no data-path collective, one
all-gather of counts at the end (RCCL).

Prints ONE JSON line (rank 0):
  value         descriptors/s with the batch already resident in HBM when the timed region starts and the
                records left in HBM (hesaff_detect_batch_device) -- the harness contract for `value`;
  host_path     the same batch through hesaff_detect_batch: host images in -> H2D -> kernels -> D2H -> host
                records out, chunks pipelined ("H2D -> D2H inclusive"), in the same run;
  end_to_end    hesaff_process_files on EVERY rank at once: PGM files on a RAM disk -> pool threads read them into pinned buffers ->
                device (rows of the .hesaff.sift files formatted by the GPU) -> the pool write()s; 2 + 2 pool threads per rank, not
                confined to a CPU share; aggregate over the ranks, CPU seconds per image beside the rate;
  end_to_end_budgeted  the same, every rank's run as a child process pinned - before it loads anything - to one device's share of
                the host (CPU quota / 8) with the thread counts of the library's rule (hesaff_host_plan_for): the stand-in for the
                per-device end-to-end rate of an 8-GPU node;
  text_export   the host formatter alone (hesaff_write_sift_batch, the stage API's writer) on a bounded subset;
  photo_density the same device-resident step on mosaics of real photographs (scikit-learn's sample images);
  roofline      the dominant pyramid kernel (k_blur_hess_march: Gaussian + det-of-Hessian), timed live with HIP
                events on the library's stream;
  cpu_baseline  the CPU oracle (a port of the reference, 1 thread) on a bounded sample of the same images;
                cpu_baseline_multicore: one oracle process per physical core.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
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


def host_cpu_info():
    """CPU model, logical / physical core counts and SMT state of this host (for the CPU baseline legs)."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        model = None
        cores = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        info["model"] = model
        if cores:
            info["physical_cores"] = len(cores)
    except OSError:
        pass
    try:
        info["smt"] = open("/sys/devices/system/cpu/smt/active").read().strip() == "1"
    except OSError:
        pass
    try:
        info["usable_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    # a container's CPU bandwidth limit (cgroup v2 cpu.max = "<quota> <period>" or "max <period>"): the number of
    # CPUs' worth of time this process tree may use, whatever the affinity mask says
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            info["cgroup_cpu_quota"] = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return info


def hbm_probe(torch, dev):
    """What THIS box's HBM delivers to plain streaming kernels (torch elementwise kernels on 2 GB operands), measured in the
    same run as the roofline lines: boxes of the pool differ by several per cent, and the guide's 8 TB/s is a pin rate."""
    n = 1 << 29
    a = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    out = {}

    def t(f, nbytes, name, reps=5):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        out[name] = nbytes / ((time.perf_counter() - t0) / reps) / 1e9
    t(lambda: b.copy_(a), 8.0 * n, "copy_1r1w_GBs")
    t(lambda: b.fill_(1.0), 4.0 * n, "fill_1w_GBs")
    t(lambda: torch.add(a, b, out=c), 12.0 * n, "add_2r1w_GBs")
    # read-only legs: the yardstick of a kernel that only reads (k_extrema_march).  torch.sum is a two-stage reduction that does not
    # stream at the copy rate on this device; the dot product of two arrays (rocBLAS, 2 reads, no write) does better - the larger is taken
    t(lambda: torch.sum(a), 4.0 * n, "sum_1r_GBs")
    t(lambda: torch.dot(a, b), 8.0 * n, "dot_2r_GBs")
    out["read_only_GBs"] = max(out["sum_1r_GBs"], out["dot_2r_GBs"])
    del a, b, c
    torch.cuda.empty_cache()
    return out


def fast_leg(hesaff_amd, torch, imgs, device, B, H, W, level, steps):
    """The same device-resident step with hesaff_params.fast = level (NOT bit-exact)."""
    p = hesaff_amd.default_params()
    p.max_batch = B
    p.fast = level
    with hesaff_amd.HesaffContext(p, device=device) as ctx:
        ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nd = 0
        for _ in range(steps):
            _, cd, _, _ = ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
            nd += int(cd.sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"value": nd / dt, "unit": "keypoints/s", "images_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps}


def density_leg(hesaff_amd, torch, dev, device, B, H, W, seed, steps):
    """The same device-resident step on photographs: B distinct mosaics of scikit-learn's two sample photographs at their native
    resolution (flips and shifts per image, hesaff_amd/synth.py); the band-noise family one octave coarser stands in when the
    photographs are not installed.  Per-stage times of the step are reported: on this content the pyramid and detection stages
    (fixed cost per pixel) weigh three times what they weigh on the dense headline images."""
    from hesaff_amd.synth import BANDS_NATURAL, band_noise_batch_torch, load_sample_photos, photo_mosaic_batch_torch
    photos = load_sample_photos()
    if photos:
        imgs = photo_mosaic_batch_torch(B, H, W, first_index=seed, device=dev, photos=photos)
        what = ("%d distinct %dx%d mosaics of scikit-learn's sample photographs (china.jpg, flower.jpg at native resolution, random "
                "flips per tile, random shift per image; decoded by the in-tree JPEG reader)" % (B, W, H))
    else:
        imgs = band_noise_batch_torch(B, H, W, seed=seed, device=dev, bands=BANDS_NATURAL)
        what = "%d x %dx%d band-noise images one octave coarser than the headline family (sample photographs not installed)" % (B, W, H)
    torch.cuda.synchronize()
    p = hesaff_amd.default_params()
    p.max_batch = B
    stage = {"pyramid_ms": 0.0, "detect_ms": 0.0, "affine_ms": 0.0, "patch_ms": 0.0, "sift_ms": 0.0, "pack_ms": 0.0, "total_ms": 0.0}
    with hesaff_amd.HesaffContext(p, device=device) as ctx:
        ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
        ctx.set_profiling(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nd = nh = 0
        for _ in range(steps):
            ch, cd, _, _ = ctx.detect_batch_device(imgs.data_ptr(), B, W, H)
            nd += int(cd.sum()); nh += int(ch.sum())
            tm = ctx.timings()
            for k in stage:
                stage[k] += getattr(tm, k) / steps
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    del imgs
    torch.cuda.empty_cache()
    return {"value": nd / dt, "unit": "keypoints/s", "images_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
            "descriptors_per_image": nd / (B * steps), "hessian_keypoints_per_image": nh / (B * steps),
            "descriptors_per_mpx": nd / (B * steps) / (W * H / 1e6), "photographs": bool(photos),
            "stage_ms_per_step": stage, "serial_head_fraction": (stage["pyramid_ms"] + stage["detect_ms"]) / max(stage["total_ms"], 1e-9),
            "what": "the same device-resident step on " + what + "; not the headline workload"}


def jpeg_path_leg(hesaff_amd, W, H, n_files, chunk, device, decode_threads, write_threads):
    """hesaff_process_files on a list of COLOUR JPEG photographs (mosaics of the two sample photographs, 4:2:0, quality 90) on a RAM
    disk, binary sidecars written there: the host threads do the entropy decoding only (hesaff_read_jpeg_coefficients), the inverse
    DCT, chroma up-sampling and colour conversion of a chunk run on the device (kernels_jpeg.h).  cv::imread + the detector,
    hesaff.cpp:137-180, for the format of the Oxford / graf images."""
    try:
        from PIL import Image
    except ImportError:
        return {"skipped": "Pillow is not installed (it only writes the test files)"}
    from hesaff_amd import synth
    photos = synth.load_sample_photos()
    if not photos:
        return {"skipped": "scikit-learn's sample photographs are not installed"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="hesaff_jpg_", dir=base)
    try:
        paths = []
        for i in range(n_files):   # 32 distinct mosaics, cycled (encoding them is the slow part of this leg)
            q = os.path.join(tmp, "p%04d.jpg" % i)
            if i < 32:
                Image.fromarray(synth.photo_mosaic(H, W, i, photos=photos)).save(q, quality=90, subsampling=2)
            else:
                shutil.copyfile(paths[i % 32], q)
            paths.append(q)
        p = hesaff_amd.default_params()
        p.max_batch = chunk
        with hesaff_amd.HesaffContext(p, device=device) as ctx:
            ctx.set_output_format(2)
            ctx.process_files(paths[:2 * chunk], decode_threads=decode_threads, write_threads=write_threads)   # warm-up: plan, pinned blocks
            for q in paths:
                if os.path.exists(q + ".hesaff.bin"):
                    os.remove(q + ".hesaff.bin")
            c0 = _cpu_seconds()
            t = time.perf_counter()
            st = ctx.process_files(paths, decode_threads=decode_threads, write_threads=write_threads)
            dt = time.perf_counter() - t
            cpu_s = _cpu_seconds() - c0
        bad = [s for s in st if s[0] != 0]
        return {"images": n_files, "images_per_s": n_files / dt, "value": sum(s[3] for s in st) / dt, "unit": "keypoints/s", "seconds": dt,
                "cpu_seconds_per_image": cpu_s / n_files, "cpus_busy": cpu_s / dt,
                "failed_files": len(bad), "input_bytes_per_file": os.path.getsize(paths[0]), "chunk_images": chunk,
                "decode_threads": decode_threads, "write_threads": write_threads, "output": "binary sidecar",
                "what": "hesaff_process_files: %d colour JPEG files (%dx%d mosaics of two photographs, 32 distinct, 4:2:0, quality 90) on a RAM disk -> %d host "
                        "threads (entropy decoding only) -> coefficient blobs to the device -> inverse DCT, up-sampling, colour conversion, "
                        "grey conversion and the whole hot path there -> sidecar files; one timed run, fill and drain included"
                        % (n_files, W, H, decode_threads + write_threads)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _e2e_files(host_imgs, W, H, n_files, chunk, world):
    """n PGM files of the bench images on a RAM disk -> (tmp dir, paths, header length), or (None, reason, 0) when it is too small."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="hesaff_e2e_", dir=base)
    # inputs 1 byte per pixel, outputs about 5.3 bytes per pixel at the dense images' 14 k descriptors per Mpx
    per_image = W * H * 7
    free = shutil.disk_usage(tmp).free
    try:   # a RAM disk's pages are host memory: never plan for more than a quarter of what is available, over all ranks
        avail = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable:")][0]
        free = min(free, avail)
    except (OSError, IndexError, ValueError):
        pass
    n = int(max(0, min(n_files, (free * (0.6 if world == 1 else 0.25) / max(world, 1)) // per_image)))
    if n < 2 * chunk:
        shutil.rmtree(tmp, ignore_errors=True)
        return None, "RAM disk too small: %d bytes free for %d images on %d rank(s)" % (free, n_files, world), 0
    paths = []
    hdr = b"P5\n%d %d\n255\n" % (W, H)
    for i in range(n):
        q = os.path.join(tmp, "img%04d.pgm" % i)
        with open(q, "wb") as f:
            f.write(hdr); f.write(host_imgs[i % len(host_imgs)].tobytes())
        paths.append(q)
    return tmp, paths, len(hdr)


def _cpu_seconds():
    """user + system CPU time of this process, all threads (getrusage): what a file leg costs the host."""
    import resource
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def _thread_cpu():
    """{tid: (name, CPU seconds, allowed CPUs)} of this process's threads, from /proc (threads that have exited are not listed)."""
    out = {}
    try:
        hz = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                st = open("/proc/self/task/%s/stat" % tid).read()
                name = st[st.index("(") + 1: st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                allowed = [ln.split(":", 1)[1].strip() for ln in open("/proc/self/task/%s/status" % tid) if ln.startswith("Cpus_allowed_list")]
                out[tid] = (name, (int(f[11]) + int(f[12])) / hz, allowed[0] if allowed else "?", int(f[12]) / hz)
            except (OSError, ValueError, IndexError):
                pass
    except OSError:
        pass
    return out


def _timed_file_run(hesaff_amd, paths, chunk, device, fmt, decode_threads, write_threads, sync=None, threads_table=False, profiling=1):
    """One warm-up (3 chunks) and one timed hesaff_process_files over `paths`; -> dict with wall seconds, CPU seconds, rows, bytes."""
    p = hesaff_amd.default_params()
    p.max_batch = chunk
    ext = ".hesaff.sift" if fmt == 1 else ".hesaff.bin"
    with hesaff_amd.HesaffContext(p, device=device) as ctx:
        ctx.set_output_format(fmt)
        ctx.set_profiling(profiling)   # 1: per-stage HIP events (device_export_ms); 0: what the CLI runs
        warm = ctx.process_files(paths[: 3 * chunk], decode_threads=decode_threads, write_threads=write_threads)   # buffers (all three pinned blocks, the readers' pinned buffers), page cache, thread start-up
        for q in paths[: 3 * chunk]:
            os.remove(q + ext)
        if sync:
            sync()
        th0 = _thread_cpu() if threads_table else None
        c0 = _cpu_seconds()
        t0 = time.perf_counter()
        st = ctx.process_files(paths, decode_threads=decode_threads, write_threads=write_threads)
        dt = time.perf_counter() - t0
        cpu = _cpu_seconds() - c0
        th1 = _thread_cpu() if threads_table else None
        threads = int(ctx.L.hesaff_host_threads())
        tmx = ctx.timings()
    bad = [i for i, s_ in enumerate(st) if s_[0] != 0 or s_[1] != 3] + [i for i, s_ in enumerate(warm) if s_[0] != 0]
    nbytes = sum(os.path.getsize(q + ext) for q in paths)
    rows = sum(s_[3] for s_ in st)
    n = len(paths)
    extra = {}
    if threads_table:   # CPU seconds per thread that still exists after the run (pool and staging threads have exited: their time is the remainder)
        rows_t = sorted(((th1[t][1] - th0.get(t, (0, 0.0, 0, 0.0))[1], th1[t][0], th1[t][2], th1[t][3] - th0.get(t, (0, 0.0, 0, 0.0))[3]) for t in th1), reverse=True)
        extra["threads_cpu_seconds"] = [{"name": nm, "cpu_seconds": round(s_, 3), "in_kernel_seconds": round(k_, 3), "allowed": al} for s_, nm, al, k_ in rows_t if s_ >= 0.02]
        extra["threads_exited_cpu_seconds"] = round(cpu - sum(r_[0] for r_ in rows_t), 3)
    return {**extra, "images": n, "images_per_s": n / dt, "value": rows / dt, "unit": "keypoints/s", "seconds": dt, "chunk_images": chunk,
            "output": "text (.hesaff.sift, the reference's format)" if fmt == 1 else "binary sidecar (.hesaff.bin, 148 bytes per row)",
            "failed_files": len(bad), "output_GB_per_s": nbytes / dt / 1e9, "output_GB": nbytes / 1e9,
            "rows": rows, "output_bytes": nbytes, "cpu_seconds": cpu, "cpu_seconds_per_image": cpu / n, "cpus_busy": cpu / dt,
            "device_export_ms_last_chunk": tmx.export_ms, "device_export_rows_last_chunk": tmx.export_rows,
            "decode_threads": decode_threads, "write_threads": write_threads, "host_threads_available": threads}


def file_path_leg(hesaff_amd, host_imgs, W, H, n_files, chunk, device, fmt=1, decode_threads=2, write_threads=2, world=1, sync=None):
    """hesaff_process_files (what `hesaff --batch` runs: decode threads -> chunks through the device, rows formatted there ->
    writer threads that only write) on n_files binary PGM files of the bench images on a RAM disk, every <name>.hesaff.sift
    written there too.  One timed run over the whole list, pipeline fill and drain included.  Every rank runs this at the same
    time (`sync` = barrier before the timed run) with the same host-thread budget."""
    tmp, paths, hl = _e2e_files(host_imgs, W, H, n_files, chunk, world)
    if tmp is None:
        if sync:
            sync()
        return {"skipped": paths}
    try:
        r = _timed_file_run(hesaff_amd, paths, chunk, device, fmt, decode_threads, write_threads, sync)
        n = len(paths)
        ext = ".hesaff.sift" if fmt == 1 else ".hesaff.bin"
        r.update({"input_GB": n * (W * H + hl) / 1e9, "target": tmp.rsplit("/", 1)[0],
                  "what": "hesaff_process_files: %d binary PGM files (%dx%d, the bench images) on a RAM disk -> read straight into pinned buffers by the pool's threads "
                          "-> chunks of %d images through the device (copy in, kernels, rows of the output files formatted on the device, copy out, all overlapped) "
                          "-> the pool (%d + %d threads) write()s -> %d %s files on the RAM disk; one timed run, pipeline fill and drain included "
                          "(hesaff.cpp:133-180 for a list of files); threads NOT confined to a CPU share: see end_to_end_budgeted for one device's share of the host"
                          % (n, W, H, chunk, decode_threads, write_threads, n, ext)})
        return r
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


BUDGET_SHARE = 8   # devices that share the host in the budgeted leg: one rank of an 8-GPU node (BASELINE.json config 4)


def budgeted_child(cfg):
    """`bench.py --budgeted-child <json>`: the file path of ONE device inside its share of the host.  The CPU mask is set FIRST - before
    numpy, torch or the library exist in this process - so every thread the run creates (the pool, the staging thread, the HIP
    runtime's own) is confined to it, and the library's own rule (hesaff_host_plan_for) sees the share as "the host"."""
    os.sched_setaffinity(0, set(cfg["cpus"]))
    if cfg.get("confine_runtime", True):
        # ROCr gives its helper threads (the asynchronous-event thread: one busy core per process while the device works) the whole machine's
        # CPU mask unless told not to; with this they inherit the share's mask like every other thread of the child
        os.environ["HSA_OVERRIDE_CPU_AFFINITY_DEBUG"] = "0"
    import hesaff_amd
    hp = hesaff_amd.host_plan(1)          # == host_plan(BUDGET_SHARE) on the whole quota
    if cfg.get("pool"):                   # experiments: another split of the pool inside the same CPU mask
        hp = dict(hp, decode_threads=int(cfg["pool"][0]), write_threads=int(cfg["pool"][1]))
    paths = [os.path.join(cfg["dir"], "img%04d.pgm" % i) for i in range(cfg["n"])]

    def sync():   # all ranks' children start their timed run together (files in a directory every rank knows)
        d = cfg.get("sync_dir")
        if not d or cfg["world"] <= 1:
            return
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "ready.%d.%d" % (cfg["phase"][0], cfg["rank"])), "w").close()
        t_end = time.time() + 180
        while time.time() < t_end and len([q for q in os.listdir(d) if q.startswith("ready.%d." % cfg["phase"][0])]) < cfg["world"]:
            time.sleep(0.002)
        cfg["phase"][0] += 1

    cfg["phase"] = [0]
    out = {"cpus": sorted(os.sched_getaffinity(0)), "plan": hp}
    if cfg.get("renice_runtime") is not None:
        # The HIP runtime's helper threads at a lower priority than the pool (what `hesaff --batch` does on a CPU-starved plan): the threads that
        # appear while the first context of the process is made are the runtime's - its event thread spins while the device works, and inside
        # a two-CPU share that spinning otherwise takes a third of a CPU from the threads that write
        before = set(os.listdir("/proc/self/task"))
        with hesaff_amd.HesaffContext(hesaff_amd.default_params(), device=cfg["device"]):
            pass
        moved = []
        for tid in set(os.listdir("/proc/self/task")) - before:
            try:
                os.setpriority(os.PRIO_PROCESS, int(tid), int(cfg["renice_runtime"])); moved.append(int(tid))
            except OSError:
                pass
        out["reniced_threads"] = len(moved)
    for name, fmt in (("text", 1), ("sidecar", 2)):
        r = _timed_file_run(hesaff_amd, paths, cfg["chunk"], cfg["device"], fmt, hp["decode_threads"], hp["write_threads"], sync, threads_table=True, profiling=0)
        ext = ".hesaff.sift" if fmt == 1 else ".hesaff.bin"
        if cfg.get("md5"):   # tests: what was written
            import hashlib
            r["md5"] = [hashlib.md5(open(q + ext, "rb").read()).hexdigest() for q in paths]
        for q in paths:
            if os.path.exists(q + ext):
                os.remove(q + ext)
        out[name] = r
    print(json.dumps(out))
    return 0


def budgeted_leg(hesaff_amd, host_imgs, W, H, n_files, chunk, device, rank, world, sync, confine_runtime=True, renice_runtime=None):
    """The file path inside ONE device's share of the host: the CPUs this job may use (affinity mask, capped by the
    cgroup quota: hesaff_host_threads) divided by BUDGET_SHARE = 8 devices, whatever `world` is; rank r's child process is pinned
    to the r-th such slice before it loads anything, takes its thread counts from the library's rule and reports CPU seconds
    (getrusage) beside the rate.  Returns the child's dict or {"skipped": ...}."""
    import subprocess
    quota = int(hesaff_amd.load_library().hesaff_host_threads())
    k = max(1, quota // BUDGET_SHARE)
    allowed = sorted(os.sched_getaffinity(0))
    cpus = [allowed[(rank * k + j) % len(allowed)] for j in range(k)]
    tmp, paths, hl = _e2e_files(host_imgs, W, H, n_files, chunk, world)
    if sync:
        sync()
    if tmp is None:
        return {"skipped": paths}
    if rank == 0:   # what an interrupted earlier run may have left behind (the other ranks' children need seconds before they write here)
        shutil.rmtree(os.path.join(os.path.dirname(tmp), "hesaff_sync_%s" % os.environ.get("MASTER_PORT", "0")), ignore_errors=True)
    try:
        cfg = {"dir": tmp, "n": len(paths), "chunk": chunk, "device": device, "cpus": cpus, "rank": rank, "world": world, "confine_runtime": confine_runtime, "renice_runtime": renice_runtime,
               "sync_dir": os.path.join(os.path.dirname(tmp), "hesaff_sync_%s" % os.environ.get("MASTER_PORT", "0"))}
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--budgeted-child", json.dumps(cfg)], capture_output=True, text=True, timeout=900)
        if r.returncode != 0 or not r.stdout.strip():
            return {"skipped": "the child failed (%d): %s" % (r.returncode, r.stderr[-300:])}
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d["quota_cpus"] = quota
        d["share"] = "1/%d of the host: %d CPU(s) of %d" % (BUDGET_SHARE, k, quota)
        return d
    except subprocess.TimeoutExpired:
        return {"skipped": "the child did not finish in 900 s"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        if rank == 0:
            shutil.rmtree(os.path.join(os.path.dirname(tmp), "hesaff_sync_%s" % os.environ.get("MASTER_PORT", "0")), ignore_errors=True)


def contexts_leg(host_imgs, W, H, n_files, chunk, device, k=4):
    """What INTEGRATION.md recommends for several devices on one host - ONE process with a context per device (`hesaff --batch --devices a,b,..`) -
    against one process per device, measured on this one device: k contexts on it in one process, then k processes with one context
    each (`--host-share k`), the same list to binary sidecars; images/s and the children's CPU seconds per image (getrusage).
    The runtime's event thread is one per PROCESS: k processes carry k of them."""
    import resource
    import subprocess
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    tmp, paths, _ = _e2e_files(host_imgs, W, H, n_files, chunk, 1)
    if tmp is None:
        return {"skipped": paths}

    def cpu_children():
        r = resource.getrusage(resource.RUSAGE_CHILDREN)
        return r.ru_utime + r.ru_stime

    def clean():
        for q in paths:
            if os.path.exists(q + ".hesaff.bin"):
                os.remove(q + ".hesaff.bin")
    try:
        lists = []
        for r in range(k + 1):   # lists[0..k-1]: the k shards; lists[k]: the whole list
            lo, hi = (len(paths) * r // k, len(paths) * (r + 1) // k) if r < k else (0, len(paths))
            q = os.path.join(tmp, "list_%d.txt" % r)
            open(q, "w").write("\n".join(paths[lo:hi]) + "\n")
            lists.append(q)
        dev = str(device)
        subprocess.run([exe, "--batch", lists[0], "--devices", dev, "--output", "bin"], capture_output=True, timeout=600)   # warm-up: page cache, the box
        out = {"what": "%d UHD PGM files -> sidecars on device %s through the CLI: ONE process with %d contexts (--devices %s) against %d processes "
                       "with one context each (--host-share %d); CPU = the children's user + system seconds (getrusage).  The start-up of the "
                       "processes (runtime, contexts, pinned blocks: about a second each, serialised inside one process) is inside both figures and "
                       "dominates the wall time of so short a list: cpu_seconds_per_image is the figure to read" % (len(paths), dev, k, ",".join([dev] * k), k, k),
               "images": len(paths), "contexts": k}
        clean()
        c0, t0 = cpu_children(), time.perf_counter()
        r = subprocess.run([exe, "--batch", lists[k], "--devices", ",".join([dev] * k), "--output", "bin"], capture_output=True, text=True, timeout=900)
        dt, cpu = time.perf_counter() - t0, cpu_children() - c0
        out["one_process"] = {"images_per_s": len(paths) / dt, "seconds": dt, "cpu_seconds_per_image": cpu / len(paths), "rc": r.returncode}
        clean()
        c0, t0 = cpu_children(), time.perf_counter()
        ps = [subprocess.Popen([exe, "--batch", q, "--devices", dev, "--output", "bin", "--host-share", str(k)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for q in lists[:k]]
        rcs = [p_.wait(timeout=900) for p_ in ps]
        dt, cpu = time.perf_counter() - t0, cpu_children() - c0
        out["one_process_per_context"] = {"images_per_s": len(paths) / dt, "seconds": dt, "cpu_seconds_per_image": cpu / len(paths), "rc": max(abs(x) for x in rcs)}
        return out
    except (OSError, subprocess.TimeoutExpired) as e:
        return {"skipped": "the CLI runs failed: %s" % e}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def launch_ranks(n):
    """Start n ranks of this script under torch.distributed.run on this node and wait for them.  Nothing in this process
    has initialised the GPU (torch is not even imported yet); the ranks are ordinary child processes."""
    import socket
    import subprocess
    if os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl":
        import torch   # device_count() does not initialise the GPU (the ranks are started as children afterwards)
        have = torch.cuda.device_count()
        if n > have:
            raise SystemExit("bench.py: --gpus %d but this node has %d visible GPU(s): one rank per GPU over RCCL needs %d" % (n, have, n))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--budgeted-child":
        raise SystemExit(budgeted_child(json.loads(sys.argv[2])))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step (BASELINE.json: 2048 UHD images over 8 GPUs)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--density", choices=("dense", "natural", "photo"), default="dense",
                    help="image family: dense = the headline workload (band noise, about 14 k descriptors per Mpx); natural = the same family one octave "
                         "coarser (2.5 k per Mpx); photo = mosaics of scikit-learn's sample photographs (about 3.5 k per Mpx)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-inclusive and text-export legs")
    ap.add_argument("--host-chunk", type=int, default=64, help="images per pipelined chunk of the host path (hesaff_params.max_batch)")
    ap.add_argument("--export-images", type=int, default=32, help="images of the batch written as .hesaff.sift text (RAM disk)")
    ap.add_argument("--cpu-images", type=int, default=8, help="images of the batch timed on the CPU oracle, 1 thread (about 14 s each; SURVEY.md 8d: at least 8 for 4K)")
    ap.add_argument("--cpu-workers", type=int, default=-1,
                    help="worker processes of the multi-core CPU baseline, one image each (-1: one per physical core, at most the batch; 0: skip)")
    ap.add_argument("--fast-steps", type=int, default=2, help="steps of the fast-mode leg (hesaff_params.fast = 2) of the default run (0: skip)")
    ap.add_argument("--photo-steps", type=int, default=2, help="steps of the extra leg on photographs of the default run (0: skip)")
    ap.add_argument("--e2e-decode-threads", type=int, default=2, help="decoder threads of the end-to-end leg, per rank")
    ap.add_argument("--e2e-write-threads", type=int, default=2, help="writer threads of the end-to-end leg, per rank (the leg is not confined to a CPU share: end_to_end_budgeted is)")
    ap.add_argument("--e2e-images", type=int, default=512, help="image files of the measured end-to-end file path (0: skip; fewer when the RAM disk is small)")
    ap.add_argument("--jpeg-images", type=int, default=384, help="colour JPEG photographs of the JPEG file-path leg (0: skip)")
    ap.add_argument("--no-budgeted", dest="budgeted", action="store_false", help="skip the end_to_end_budgeted leg (the file path confined to one device's share of the host's CPUs)")
    ap.add_argument("--e2e-chunk", type=int, default=32, help="images per device chunk of the end-to-end leg (hesaff_params.max_batch)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank owns --batch images per step; strong: --global-images images per step in total, "
                         "split evenly over the ranks (BASELINE.json config 4: 2048 UHD images over 8 GPUs)")
    ap.add_argument("--global-images", type=int, default=2048, help="images per step of the whole job under --scaling strong")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself (one process per GPU under
    # torch.distributed.run) BEFORE anything here touches torch or the GPU, relays rank 0's JSON line and exits with the
    # launcher's status.  Under a launcher (WORLD_SIZE set) the world size must be the one asked for.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a line "
                         "for a different world size" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_DIST_BACKEND=gloo (testing only): the multi-rank code path on a box with fewer GPUs than
    # ranks - ranks share the visible devices and the three small collectives run on CPU tensors.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    # under a launcher the process group is always initialised - also for one rank, so that `torchrun --nproc-per-node 1
    # bench.py` exercises the RCCL barrier / all-reduce / all-gather on a one-GPU box; plain `python bench.py` has no group
    grouped = world > 1 or "WORLD_SIZE" in os.environ
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        import datetime
        tmo = datetime.timedelta(minutes=30)   # rank 0 alone runs the CPU legs at the end; no collective waits for them
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend=backend, timeout=tmo)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libhesaff_amd has no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    import hesaff_amd
    from hesaff_amd.synth import band_noise_batch_torch

    H, W = args.height, args.width
    from hesaff_amd.shard import shard_range
    if args.scaling == "strong":
        # strong scaling: the job's images per step are fixed; rank r owns the contiguous block shard_range gives it
        # (hesaff_shard_range, the product's own rule) and walks it in library calls of at most --batch images
        g_lo, g_hi = shard_range(args.global_images, rank, world)
        per_rank = g_hi - g_lo
        B = max(1, min(args.batch, per_rank))
    else:
        # weak scaling: every rank owns --batch distinct images (global image index = rank * batch + i)
        per_rank = B = args.batch
        g_lo = rank * B
    from hesaff_amd.synth import BANDS, BANDS_NATURAL, photo_mosaic_batch_torch
    bands = BANDS_NATURAL if args.density == "natural" else BANDS
    # B distinct images per rank (seeded by global image index); a rank whose share exceeds B cycles through them
    if args.density == "photo":
        imgs = photo_mosaic_batch_torch(B, H, W, first_index=g_lo, device=dev)
    else:
        imgs = band_noise_batch_torch(B, H, W, seed=1234 + g_lo, device=dev, bands=bands)
    torch.cuda.synchronize()

    p = hesaff_amd.default_params()
    p.max_batch = B
    ctx = hesaff_amd.HesaffContext(p, device=local_rank)

    bh = {"ms": 0.0, "bytes": 0.0, "launches": 0, "ex_ms": 0.0, "ex_bytes": 0.0, "ex_launches": 0}
    stage = {"pyramid_ms": 0.0, "detect_ms": 0.0, "affine_ms": 0.0, "patch_ms": 0.0, "sift_ms": 0.0, "pack_ms": 0.0, "total_ms": 0.0}
    tot = {"desc": 0, "hess": 0, "imgs": 0, "pyr_bytes": 0.0}

    def step(timed):
        # one pass of the hot path over this rank's images of a step
        done = 0
        while done < per_rank:
            nb = min(B, per_rank - done)
            ch, cd, _, total = ctx.detect_batch_device(imgs.data_ptr(), nb, W, H)
            done += nb
            if timed:
                tot["desc"] += int(cd.sum()); tot["hess"] += int(ch.sum()); tot["imgs"] += nb
                tm = ctx.timings()
                bh["ms"] += tm.blur_hess_ms; bh["bytes"] += tm.blur_hess_bytes; bh["launches"] += tm.blur_hess_launches
                bh["ex_ms"] += tm.extrema_ms; bh["ex_bytes"] += tm.extrema_bytes; bh["ex_launches"] += tm.extrema_launches
                tot["pyr_bytes"] += tm.pyramid_bytes
                for k in stage:
                    stage[k] += getattr(tm, k)

    for _ in range(args.warmup):
        step(False)
    ctx.set_profiling(2)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    from hesaff_amd.shard import gather_counts
    n_hess, n_desc = tot["hess"], tot["desc"]
    bh_ms, bh_bytes, bh_launches = bh["ms"], bh["bytes"], bh["launches"]
    counts = gather_counts([n_hess, n_desc, tot["imgs"]], device=coll_dev if grouped else None)
    tot_hess, tot_desc, tot_imgs = [int(v) for v in counts.sum(axis=0)]
    per_rank_images = [int(v) for v in counts[:, 2]]
    ctx.close()

    # ---- host-inclusive legs, EVERY rank at the same time: hesaff_detect_batch, then the file path ----
    host_path = None
    end_to_end = None
    end_to_end_budgeted = None
    host_imgs = None
    if not args.no_host_path:
        hp = hesaff_amd.default_params()
        hp.max_batch = max(1, min(args.host_chunk, B))
        hctx = hesaff_amd.HesaffContext(hp, device=local_rank)
        host_imgs = list(imgs.cpu().numpy())           # B pageable host images; the library stages them through pinned memory
        hctx.detect_batch_raw(host_imgs)                   # warm-up: device buffers of both pipeline slots and one pinned result block per chunk
        hsteps = max(1, min(args.steps, 2))
        barrier()
        t1 = time.perf_counter()
        hdesc = 0
        for _ in range(hsteps):
            res = hctx.detect_batch_raw(host_imgs)
            hdesc += sum(r.count_desc for r in res)
        barrier()
        hdt = time.perf_counter() - t1
        if grouped:
            t = torch.tensor([hdt], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hdt = float(t.item())
        hc = gather_counts([hdesc, B * hsteps], device=coll_dev if grouped else None).sum(axis=0)
        host_path = {"value": float(hc[0]) / hdt, "unit": "keypoints/s", "images_per_s": float(hc[1]) / hdt, "steps": hsteps,
                     "ms_per_step": hdt / hsteps * 1e3, "chunk_images": int(hp.max_batch),
                     "what": "hesaff_detect_batch: pageable host images -> pinned staging -> H2D -> kernels -> D2H -> pinned host records; "
                             "chunks of %d images, H2D of chunk i+1 and D2H of chunk i-1 beside the kernels of chunk i" % hp.max_batch,
                     "h2d_bytes_per_step": B * H * W, "d2h_bytes_per_step": int(hdesc // hsteps) * 164}
        hctx.close()
        # the whole file path, measured on every rank at once: image files -> decode -> device -> .hesaff.sift files (hesaff.cpp:133-180)
        if args.e2e_images > 0:
            quota_cpus = int(hesaff_amd.load_library().hesaff_host_threads())

            def combine(r):
                """One rank's file-leg dict -> the aggregate over all ranks (rank 0), None elsewhere."""
                barrier()
                ok = "images_per_s" in r
                mine = [r["images"], r["rows"], r["failed_files"], r["output_bytes"], 1, int(r["cpu_seconds"] * 1e6)] if ok else [0, 0, 0, 0, 0, 0]
                sec = r["seconds"] if ok else 0.0
                if grouped:
                    t = torch.tensor([sec], device=coll_dev, dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    sec = float(t.item())
                g = gather_counts(mine, device=coll_dev if grouped else None)
                if rank != 0:
                    return None
                if int(g[:, 4].sum()) != world or sec <= 0:
                    return r if not ok else {"skipped": "the leg did not run on every rank"}
                tot = g.sum(axis=0)
                cpu_s = float(tot[5]) / 1e6
                rate = float(tot[0]) / sec
                r.update({"images": int(tot[0]), "images_per_s": rate, "value": float(tot[1]) / sec, "rows": int(tot[1]),
                          "failed_files": int(tot[2]), "output_bytes": int(tot[3]), "output_GB": float(tot[3]) / 1e9,
                          "output_GB_per_s": float(tot[3]) / sec / 1e9, "seconds": sec, "ranks": world,
                          "per_rank_images": [int(v) for v in g[:, 0]],
                          "cpu_seconds": cpu_s, "cpu_seconds_per_image": cpu_s / max(float(tot[0]), 1.0), "cpus_busy": cpu_s / sec,
                          # devices that could run at this leg's per-device rate on the CPU seconds the quota provides
                          "max_devices_at_this_quota": int(quota_cpus / max(cpu_s / max(float(tot[0]), 1.0) * (rate / world), 1e-9)),
                          "quota_cpus": quota_cpus,
                          "fraction_of_host_path": rate / host_path["images_per_s"]})
                return r

            def e2e(fmt):
                return combine(file_path_leg(hesaff_amd, host_imgs, W, H, args.e2e_images, args.e2e_chunk, local_rank, fmt=fmt,
                                             decode_threads=args.e2e_decode_threads, write_threads=args.e2e_write_threads, world=world, sync=barrier))
            end_to_end = e2e(1)
            eb = e2e(2)
            if rank == 0 and end_to_end is not None:
                end_to_end["binary_sidecar"] = eb
            # the same list inside ONE device's share of the host: a child process per rank, pinned before it loads anything
            if args.budgeted:
                def budgeted(confine_runtime, renice_runtime=None):
                    bd = budgeted_leg(hesaff_amd, host_imgs, W, H, args.e2e_images, args.e2e_chunk, local_rank, rank, world, barrier, confine_runtime, renice_runtime)
                    legs = {}
                    for name in ("text", "sidecar"):
                        legs[name] = combine(dict(bd[name]) if name in bd else {"skipped": bd.get("skipped", "no result")})
                    return bd, legs
                bd, legs = budgeted(True)
                bd_u, legs_u = budgeted(False)
                bd_n, legs_n = budgeted(True, 19)
                if rank == 0:
                    end_to_end_budgeted = {
                        "what": "hesaff_process_files as in end_to_end, but every rank's run is a child process confined (sched_setaffinity before it "
                                "loads numpy, torch or the library; HSA_OVERRIDE_CPU_AFFINITY_DEBUG=0 so that the HIP runtime's helper threads stay inside "
                                "the mask too) to ONE device's share of the host - the CPUs this job may use divided by %d "
                                "devices, whatever --gpus is - with the thread counts the library's rule gives for that share "
                                "(hesaff_host_plan_for: include/hesaff_amd.h); cpu_seconds = getrusage of the child around the timed run; "
                                "max_devices_at_this_quota = quota_cpus / (cpu_seconds_per_image x images_per_s per device). "
                                "A thread write()s about 6 GB/s of new RAM-disk pages: the text leg (46 MB per dense UHD image) is bound by that, "
                                "the sidecar (17 MB) is not.  runtime_threads_unconfined: the same without the environment variable (the runtime's "
                                "event thread then runs on CPUs outside the share: the confinement leaks by that thread)" % BUDGET_SHARE,
                        "share": bd.get("share"), "cpus_rank0": bd.get("cpus"), "plan": bd.get("plan"), "text": legs["text"], "binary_sidecar": legs["sidecar"],
                        "runtime_threads_unconfined": {"text": legs_u["text"], "binary_sidecar": legs_u["sidecar"]},
                        "runtime_threads_niced": {"what": "confined as the first form, and the threads the HIP runtime created while the child's first context was made moved to nice 19 "
                                                          "(below the pool's nice 10) - what `hesaff --batch` does on a CPU-starved plan: the event thread's spinning yields to the threads that write",
                                                  "reniced_threads": bd_n.get("reniced_threads"), "text": legs_n["text"], "binary_sidecar": legs_n["sidecar"]}}
                if world == 1:
                    end_to_end_budgeted["contexts_in_one_process"] = contexts_leg(host_imgs, W, H, args.e2e_images, args.e2e_chunk, local_rank)

    # ---- from here on rank 0 alone (no collective follows: the other ranks are done) ----
    if grouped and rank != 0:
        dist.destroy_process_group()
        return
    fast_modes = None
    if args.fast_steps > 0 and not args.no_host_path:
        fast_modes = {"what": "the same device-resident step with hesaff_params.fast = 2 (windows larger than the 41 x 41 patch sampled from the "
                              "scale space instead of warped and blurred, affine.cpp:114-135); NOT bit-exact, not the headline value; "
                              "measured effect on descriptors and matching: DESIGN.md, profiles/",
                      "fast_2": fast_leg(hesaff_amd, torch, imgs, local_rank, B, H, W, 2, args.fast_steps)}
    del imgs
    torch.cuda.empty_cache()
    probe = hbm_probe(torch, dev)
    photo = None
    if args.photo_steps > 0 and args.density == "dense" and not args.no_host_path:
        photo = density_leg(hesaff_amd, torch, dev, local_rank, B, H, W, g_lo, args.photo_steps)
    # the host formatter alone (the writer of the stage API; the file path above formats on the device)
    text_export = None
    if not args.no_host_path and args.export_images > 0:
        hp = hesaff_amd.default_params()
        hp.max_batch = max(1, min(args.host_chunk, B))
        ne = min(args.export_images, B)
        with hesaff_amd.HesaffContext(hp, device=local_rank) as hctx:
            res = hctx.detect_batch_raw(host_imgs[:ne])
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            tmp = tempfile.mkdtemp(prefix="hesaff_bench_", dir=base)
            try:
                paths = [os.path.join(tmp, "img%04d.hesaff.sift" % i) for i in range(ne)]
                hctx.write_sift_batch_raw(paths, res, hp.mrSize, 0)            # warm-up (page cache, thread pool)
                t2 = time.perf_counter()
                hctx.write_sift_batch_raw(paths, res, hp.mrSize, 0)
                edt = time.perf_counter() - t2
                rows = sum(r.count_desc for r in res)
                nbytes = sum(os.path.getsize(q) for q in paths)
                text_export = {"images": ne, "rows_per_s": rows / edt, "images_per_s": ne / edt, "text_GB_per_s": nbytes / edt / 1e9,
                               "bytes_per_image": nbytes / ne, "threads": int(hctx.L.hesaff_host_threads()), "target": tmp.rsplit("/", 1)[0],
                               "what": "hesaff_write_sift_batch (host formatter, the reference's text format, hesaff.cpp:107-130) of %d images "
                                       "of the batch on every host thread the CPU quota allows, nothing else running" % ne}
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    jpeg_path = None
    if args.jpeg_images > 0 and not args.no_host_path and args.density == "dense":
        jpeg_path = jpeg_path_leg(hesaff_amd, W, H, args.jpeg_images, args.e2e_chunk, local_rank, args.e2e_decode_threads, args.e2e_write_threads)
    cpu_sample = host_imgs

    if rank == 0:
        achieved = (bh_bytes / 1e9) / (bh_ms / 1e3) if bh_ms > 0 else 0.0
        st = {k: v / args.steps for k, v in stage.items()}
        pyr_bytes_step = tot["pyr_bytes"] / args.steps
        ex_achieved = (bh["ex_bytes"] / 1e9) / (bh["ex_ms"] / 1e3) if bh["ex_ms"] > 0 else 0.0
        if fast_modes:
            fast_modes["fast_2"]["speed_up_over_parity_rank0"] = (dt / args.steps * 1e3) / fast_modes["fast_2"]["ms_per_step"] if per_rank == B else None
        out = {
            "metric": "keypoints/sec (descriptors written), 4K grayscale batch",
            "value": tot_desc / dt,
            "unit": "keypoints/s",
            "images_per_s": tot_imgs / dt,
            "hessian_keypoints_per_s": tot_hess / dt,
            "n_gpus": world,
            "collective_backend": (backend if backend != "nccl" else "nccl (RCCL)") if grouped else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "batch of %d x %dx%d 8-bit grey images per GPU per step%s, %s, default params"
                                   % (per_rank, W, H, (" (%d distinct, cycled; library calls of %d)" % (B, B)) if per_rank > B else "",
                                      {"dense": "band-noise synthetic", "natural": "band-noise synthetic (natural density)",
                                       "photo": "mosaics of two photographs"}[args.density]),
                       "images_per_gpu_per_step": per_rank, "images_per_library_call": B, "images_per_step_all_ranks": tot_imgs // max(args.steps, 1),
                       "per_rank_images_timed": per_rank_images, "width": W, "height": H, "sharding": "image-level, contiguous blocks, %d rank(s), no data-path collective; one all-gather of counts" % world,
                       "descriptors_per_image": tot_desc / max(tot_imgs, 1),
                       "descriptors_timed_all_ranks": tot_desc, "hessian_keypoints_timed_all_ranks": tot_hess,
                       "value_is": "device-resident: inputs in HBM before the timed region, records left in HBM (hesaff_detect_batch_device)"},
            "host_path": host_path,
            "text_export": text_export,
            "end_to_end": end_to_end,
            "end_to_end_budgeted": end_to_end_budgeted,
            "photo_density": photo,
            "jpeg_path": jpeg_path,
            "fast_modes": fast_modes,
            "hbm_probe": probe,
            "roofline_detect": {"bound": "hbm", "kernel": "k_extrema_march (the three 3x3x3 extrema scans of an octave in one launch; SURVEY.md 8d: "
                                                          "B_ext = 20 bytes per pixel and octave, five response planes read once)",
                                "achieved": ex_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ex_achieved / HBM_PEAK_GBS,
                                "frac_of_read_probe": ex_achieved / probe["read_only_GBs"] if probe.get("read_only_GBs") else None,
                                "launches": bh["ex_launches"], "avg_launch_ms": bh["ex_ms"] / max(bh["ex_launches"], 1),
                                "stage": {"what": "whole detection stage (map clear, extrema, localisation, dedupe, ordering scans) against B_ext",
                                          "achieved": (bh["ex_bytes"] / args.steps / 1e9) / (st["detect_ms"] / 1e3) if st["detect_ms"] > 0 else 0.0}},
            "stage_ms_per_step": {"serial_on_main_stream": {"pyramid_ms": st["pyramid_ms"], "detect_ms": st["detect_ms"], "pack_ms": st["pack_ms"]},
                                  "concurrent_stream_busy_time": {"affine_ms": st["affine_ms"], "patch_ms": st["patch_ms"], "sift_ms": st["sift_ms"]},
                                  "device_total_ms": st["total_ms"],
                                  "note": "affine / patch / sift run concurrently on three streams over groups of images (three-deep pipeline): "
                                          "each figure is the sum of that stage's own event brackets on its own stream, they overlap in "
                                          "wall-clock time and add up to more than device_total_ms minus the serial stages"},
            "roofline": {"bound": "hbm",
                         "kernel": "k_blur_hess_march (Gaussian + det-of-Hessian; 4 launches per octave = 58 B/px algorithmic: "
                                   "12 B/px each + 8 B/px for the fused R0 + 2 B/px for the fused decimation)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(B, W, H, bh_launches // max(args.steps, 1)), "launches": bh_launches,
                         "avg_launch_ms": bh_ms / max(bh_launches, 1), "bytes_per_launch_avg": bh_bytes / max(bh_launches, 1),
                         "stage": {"what": "whole pyramid stage incl. grey conversion and initial blur: B_pyr = 5 N0 + 58 sum N_k per image (SURVEY.md 8d)",
                                   "achieved": (pyr_bytes_step / 1e9) / (st["pyramid_ms"] / 1e3) if st["pyramid_ms"] > 0 else 0.0,
                                   "frac": ((pyr_bytes_step / 1e9) / (st["pyramid_ms"] / 1e3)) / HBM_PEAK_GBS if st["pyramid_ms"] > 0 else 0.0}},
        }
        if not args.no_cpu_baseline:   # rank 0 only, also when world > 1 (the other ranks are done)
            from tests import _oracle
            if cpu_sample is None:
                cpu_sample = list(band_noise_batch_torch(min(B, 16), H, W, seed=1234 + g_lo, device=dev, bands=bands).cpu().numpy()) \
                    if args.density != "photo" else list(photo_mosaic_batch_torch(min(B, 16), H, W, first_index=g_lo, device=dev).cpu().numpy())
            cpu = host_cpu_info()
            host = cpu_sample[: max(1, args.cpu_images)]
            t1 = time.perf_counter()
            nk = 0
            for im in host:
                o = _oracle.OracleRun(_oracle.gray_from_u8(im))
                nk += o.n_keys
            cdt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": nk / cdt, "unit": "keypoints/s", "cores": 1, "kind": "port",
                                   "images_per_s": len(host) / cdt,
                                   "sample": "%d of the %d batch images (%dx%d), oracle/libhesaff_oracle.so, 1 thread, %.1f s (a port: the reference's loops, "
                                             "with the Gaussian blur - cv::GaussianBlur, SIMD code in OpenCV too - vectorised across pixels in the reference's "
                                             "order of operations, AVX2 / AVX-512 where the CPU has them)" % (len(host), B, W, H, cdt),
                                   "cpu": cpu}
            # the same oracle, one worker process per physical core, one image each (>= 8 images)
            phys = cpu.get("physical_cores") or max(1, (cpu.get("logical_cpus") or 4) // 2)
            usable = cpu.get("usable_cpus") or cpu.get("logical_cpus") or phys
            quota = cpu.get("cgroup_cpu_quota")
            auto = min(phys, usable)
            if quota:       # more workers than the CPU-time limit only time-slice (measured: 128 workers on a 16-CPU quota
                auto = max(8, min(auto, int(quota)))    # are slower in aggregate than 32); never fewer than 8 images
            nw = args.cpu_workers if args.cpu_workers >= 0 else auto
            nw = min(nw, len(cpu_sample))
            if nw > 1:
                import multiprocessing as mp
                sample = cpu_sample[:nw]
                t1 = time.perf_counter()
                with mp.get_context("spawn").Pool(nw) as pool:
                    res = pool.map(_cpu_oracle_worker, sample)
                mdt = time.perf_counter() - t1
                out["cpu_baseline_multicore"] = {"value": sum(r[0] for r in res) / mdt, "unit": "keypoints/s", "cores": nw, "kind": "port",
                                                 "images_per_s": nw / mdt,
                                                 "sample": "%d of the %d batch images, %d oracle worker processes (one per physical core, capped by the "
                                                           "container's CPU quota of %s CPUs), one image each, %.1f s wall (slowest worker %.1f s)"
                                                           % (nw, B, nw, ("%g" % quota) if quota else "unlimited", mdt, max(r[1] for r in res)),
                                                 "cpu": cpu}
        print(json.dumps(out))
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
