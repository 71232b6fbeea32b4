"""ctypes binding of libhesaff_amd.so (C ABI in include/hesaff_amd.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class HesaffError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hesaff_amd error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    """hesaff_params (include/hesaff_amd.h): reference defaults + capacity knobs."""
    _fields_ = [
        ("threshold", C.c_float),
        ("edgeEigenValueRatio", C.c_float),
        ("initialSigma", C.c_float),
        ("maxIterations", C.c_int),
        ("convergenceThreshold", C.c_float),
        ("mrSize", C.c_float),
        ("maxBinValue", C.c_float),
        ("upscaleInputImage", C.c_int),
        ("max_batch", C.c_int),
        ("max_kpts_per_mpx", C.c_int),
        ("fast", C.c_int),
    ]


class _Result(C.Structure):
    _fields_ = [("count_hessian", C.c_int32), ("count_desc", C.c_int32), ("keys", C.c_void_p)]


class Timings(C.Structure):
    _fields_ = [
        ("pyramid_ms", C.c_float), ("detect_ms", C.c_float), ("affine_ms", C.c_float), ("patch_ms", C.c_float),
        ("sift_ms", C.c_float), ("pack_ms", C.c_float), ("total_ms", C.c_float), ("blur_hess_ms", C.c_float), ("blur_hess_launches", C.c_int32),
        ("blur_hess_bytes", C.c_double), ("pyramid_bytes", C.c_double),
        ("extrema_ms", C.c_float), ("extrema_launches", C.c_int32), ("extrema_bytes", C.c_double),
        ("export_ms", C.c_float), ("export_rows", C.c_int32),
    ]


# struct Keypoint of hesaff.cpp:41-48 == hesaff_keypoint, 164 bytes
KEYPOINT_DTYPE = np.dtype([
    ("x", "<f4"), ("y", "<f4"), ("s", "<f4"), ("a11", "<f4"), ("a12", "<f4"), ("a21", "<f4"), ("a22", "<f4"),
    ("response", "<f4"), ("type", "<i4"), ("desc", "u1", (128,)),
])
assert KEYPOINT_DTYPE.itemsize == 164

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

ABI_VERSION = 7   # HESAFF_ABI_VERSION of the include/hesaff_amd.h these ctypes structs mirror


class JpegLayout(C.Structure):
    """hesaff_jpeg_layout: what the JPEG images of one device chunk share."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32)] + \
               [(n, C.c_int32 * 3) for n in ("h", "v", "hx", "vx", "bw", "bh", "cw", "chgt")]


class HostPlan(C.Structure):
    """hesaff_host_plan: one device's share of the host (hesaff_host_plan_for)"""
    _fields_ = [("cpus", C.c_int), ("decode_threads", C.c_int), ("write_threads", C.c_int), ("stage_threads", C.c_int)]


def host_plan(devices_sharing_host=1):
    """The library's one rule for host threads per device: dict(cpus, decode_threads, write_threads, stage_threads)."""
    hp = HostPlan()
    rc = load_library().hesaff_host_plan_for(int(devices_sharing_host), C.byref(hp))
    if rc != 0:
        raise HesaffError(rc, "hesaff_host_plan_for(%r)" % (devices_sharing_host,))
    return {k: int(getattr(hp, k)) for k, _ in HostPlan._fields_}


class FileStatus(C.Structure):
    """hesaff_file_status"""
    _fields_ = [("rc", C.c_int32), ("stage", C.c_int32), ("count_hessian", C.c_int32), ("count_desc", C.c_int32)]


CHUNK_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_Result))

_lib = None


def lib_path():
    # HESAFF_AMD_LIB: alternative build of the same library (kernel tuning A/B runs)
    return os.environ.get("HESAFF_AMD_LIB") or os.path.join(_HERE, "libhesaff_amd.so")


def load_library():
    """Load libhesaff_amd.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so).  Two HIP runtimes in one
    # process cannot both see the GPU, so when torch is installed it is imported FIRST and
    # libhesaff_amd.so binds to the runtime torch already loaded (same SONAME).  Without torch
    # (e.g. the hesaff CLI) the library uses the system ROCm runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    p = lib_path()
    if not os.path.exists(p):
        raise HesaffError(-1, "%s not built: run `make -C hesaff_amd/csrc` (or __graft_entry__.build())" % p)
    L = C.CDLL(p)
    vp = C.c_void_p
    L.hesaff_version.restype = C.c_char_p
    L.hesaff_abi_version.argtypes = []
    L.hesaff_sizeof_params.argtypes = []; L.hesaff_sizeof_params.restype = C.c_size_t
    L.hesaff_sizeof_timings.argtypes = []; L.hesaff_sizeof_timings.restype = C.c_size_t
    if (L.hesaff_abi_version() != ABI_VERSION or L.hesaff_sizeof_params() != C.sizeof(Params)
            or L.hesaff_sizeof_timings() != C.sizeof(Timings)):
        raise HesaffError(-2, "%s has ABI version %d (params %d bytes, timings %d bytes); this binding mirrors version %d (%d, %d)"
                          % (p, L.hesaff_abi_version(), L.hesaff_sizeof_params(), L.hesaff_sizeof_timings(), ABI_VERSION,
                             C.sizeof(Params), C.sizeof(Timings)))
    L.hesaff_default_params.argtypes = [C.POINTER(Params)]
    L.hesaff_create.argtypes = [C.POINTER(vp), C.POINTER(Params), C.c_int]
    L.hesaff_destroy.argtypes = [vp]; L.hesaff_destroy.restype = None
    L.hesaff_last_error.argtypes = [vp]; L.hesaff_last_error.restype = C.c_char_p
    L.hesaff_detect_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_Result)]
    L.hesaff_detect_batch_cb.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), CHUNK_SINK, vp]
    L.hesaff_process_files.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(FileStatus)]
    L.hesaff_write_sift_mt.argtypes = [C.c_char_p, vp, C.c_int, C.c_float, C.c_int]
    L.hesaff_write_bin.argtypes = [C.c_char_p, vp, C.c_int, C.c_float]
    L.hesaff_set_output_format.argtypes = [vp, C.c_int]
    L.hesaff_set_resume.argtypes = [vp, C.c_int]
    L.hesaff_output_is_complete.argtypes = [C.c_char_p, C.c_int]
    L.hesaff_set_pinned_read_budget.argtypes = [vp, C.c_size_t, C.c_size_t]
    L.hesaff_set_pool_priority.argtypes = [vp, C.c_int]
    L.hesaff_stage_threads_for_pool.argtypes = [C.c_int]
    L.hesaff_detect_batch_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, _i32p, _i32p, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.hesaff_set_profiling.argtypes = [vp, C.c_int]
    L.hesaff_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.hesaff_ellipse.argtypes = [vp, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hesaff_ellipse.restype = None
    L.hesaff_write_sift.argtypes = [C.c_char_p, vp, C.c_int, C.c_float]
    L.hesaff_format_sift.argtypes = [vp, C.c_int, C.c_float, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.hesaff_format_sift_mt.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.hesaff_host_threads.argtypes = []
    L.hesaff_host_threads.restype = C.c_int
    L.hesaff_host_plan_for.argtypes = [C.c_int, C.POINTER(HostPlan)]
    L.hesaff_host_plan_for.restype = C.c_int
    L.hesaff_write_sift_batch.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(_Result), C.c_float, C.c_int]
    L.hesaff_test_fmt_g.argtypes = [_f32p, C.c_int]
    L.hesaff_free.argtypes = [vp]; L.hesaff_free.restype = None
    L.hesaff_read_pnm.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hesaff_read_png.argtypes = L.hesaff_read_pnm.argtypes
    L.hesaff_read_image.argtypes = L.hesaff_read_pnm.argtypes
    L.hesaff_read_bmp.argtypes = L.hesaff_read_pnm.argtypes
    L.hesaff_read_tiff.argtypes = L.hesaff_read_pnm.argtypes
    L.hesaff_read_jpeg.argtypes = L.hesaff_read_pnm.argtypes
    L.hesaff_stage_gaussian_blur.argtypes = [vp, _f32p, C.c_int, C.c_int, C.c_float, _f32p]
    L.hesaff_stage_hessian_response.argtypes = [vp, _f32p, C.c_int, C.c_int, C.c_float, _f32p]
    L.hesaff_stage_half_image.argtypes = [vp, _f32p, C.c_int, C.c_int, _f32p]
    L.hesaff_stage_pyramid.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.hesaff_stage_hessian_keypoints.argtypes = [vp, _u8p, C.c_int, C.c_int, C.c_int, _f32p, _i32p, C.POINTER(C.c_int)]
    L.hesaff_stage_find_affine_shape.argtypes = [vp, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p, _f32p, _i32p]
    L.hesaff_stage_rectify.argtypes = [vp, C.c_int, _f32p]
    L.hesaff_stage_normalize_affine.argtypes = [vp, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _i32p, _f32p]
    L.hesaff_stage_sift.argtypes = [vp, C.c_int, _f32p, _u8p]
    L.hesaff_stage_math.argtypes = [vp, C.c_int, _f32p, _f32p, _f32p, _f32p]
    L.hesaff_stage_math_sift.argtypes = [vp, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]
    L.hesaff_stage_export.argtypes = [vp, vp, C.c_int, C.c_float, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.hesaff_stage_fmt_g.argtypes = [vp, C.c_int, _f32p, vp, _i32p]
    L.hesaff_read_jpeg_coefficients.argtypes = [C.c_char_p, C.POINTER(JpegLayout), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.hesaff_stage_jpeg_pixels.argtypes = [vp, C.POINTER(JpegLayout), C.c_int, vp, C.c_size_t, vp]
    L.hesaff_write_sift_rows.argtypes = [C.c_char_p, vp, C.c_size_t, C.c_int]
    L.hesaff_write_bin_rows.argtypes = [C.c_char_p, vp, C.c_int]
    L.hesaff_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hesaff_table_gauss_mask.argtypes = [C.c_int, _f32p]
    L.hesaff_table_circ_gauss_mask.argtypes = [C.c_int, _f32p]
    L.hesaff_table_sift_bins.argtypes = [_i32p, _i32p, _f32p, _f32p]
    L.hesaff_table_gauss_kernel.argtypes = [C.c_float, C.c_int, vp, C.POINTER(C.c_int)]
    _lib = L
    return L


# every symbol include/hesaff_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "hesaff_version", "hesaff_default_params", "hesaff_create", "hesaff_destroy", "hesaff_last_error",
    "hesaff_detect_batch", "hesaff_detect_batch_device", "hesaff_set_profiling", "hesaff_get_timings", "hesaff_ellipse",
    "hesaff_write_sift", "hesaff_format_sift", "hesaff_free", "hesaff_read_pnm", "hesaff_stage_gaussian_blur",
    "hesaff_stage_hessian_response", "hesaff_stage_half_image", "hesaff_stage_pyramid", "hesaff_stage_hessian_keypoints",
    "hesaff_stage_find_affine_shape", "hesaff_stage_rectify", "hesaff_stage_normalize_affine", "hesaff_stage_sift",
    "hesaff_stage_math", "hesaff_stage_math_sift", "hesaff_table_gauss_mask", "hesaff_table_circ_gauss_mask", "hesaff_table_sift_bins",
    "hesaff_table_gauss_kernel", "hesaff_format_sift_mt", "hesaff_write_sift_batch", "hesaff_test_fmt_g",
    "hesaff_read_png", "hesaff_read_image", "hesaff_device_count", "hesaff_shard_range", "hesaff_read_jpeg",
    "hesaff_host_threads", "hesaff_host_plan_for", "hesaff_abi_version", "hesaff_sizeof_params", "hesaff_sizeof_timings", "hesaff_detect_batch_cb",
    "hesaff_process_files", "hesaff_write_sift_mt", "hesaff_write_bin", "hesaff_set_output_format",
    "hesaff_write_sift_rows", "hesaff_write_bin_rows", "hesaff_stage_export", "hesaff_stage_fmt_g", "hesaff_set_resume",
    "hesaff_output_is_complete", "hesaff_read_jpeg_coefficients", "hesaff_read_jpeg_coefficients_alloc", "hesaff_stage_jpeg_pixels",
    "hesaff_read_pnm_alloc", "hesaff_read_image_alloc", "hesaff_set_pinned_read_budget", "hesaff_set_pool_priority",
    "hesaff_stage_threads_for_pool", "hesaff_read_bmp", "hesaff_read_tiff",
]


def default_params():
    p = Params()
    load_library().hesaff_default_params(C.byref(p))
    return p


def table_gauss_mask(size):
    m = np.zeros((size, size), np.float32)
    load_library().hesaff_table_gauss_mask(size, m)
    return m


def table_circ_gauss_mask(size):
    m = np.zeros((size, size), np.float32)
    load_library().hesaff_table_circ_gauss_mask(size, m)
    return m


def table_sift_bins():
    b0 = np.zeros(41, np.int32); b1 = np.zeros(41, np.int32); w0 = np.zeros(41, np.float32); w1 = np.zeros(41, np.float32)
    load_library().hesaff_table_sift_bins(b0, b1, w0, w1)
    return b0, b1, w0, w1


def table_gauss_kernel(sigma):
    L = load_library()
    k = C.c_int()
    L.hesaff_table_gauss_kernel(sigma, 0, None, C.byref(k))
    taps = np.zeros(k.value, np.float32)
    L.hesaff_table_gauss_kernel(sigma, k.value, taps.ctypes.data, C.byref(k))
    return taps


def ellipse(keys, mr_size):
    """(a,b,c) of each record, hesaff.cpp:115-123 in closed form."""
    L = load_library()
    keys = np.ascontiguousarray(keys, dtype=KEYPOINT_DTYPE)
    out = np.zeros((len(keys), 3), np.float32)
    a = C.c_float(); b = C.c_float(); c = C.c_float()
    base = keys.ctypes.data
    for i in range(len(keys)):
        L.hesaff_ellipse(base + i * 164, mr_size, C.byref(a), C.byref(b), C.byref(c))
        out[i] = (a.value, b.value, c.value)
    return out


def format_sift(keys, mr_size):
    """exportKeypoints hesaff.cpp:107-130 -> bytes of the .hesaff.sift file."""
    L = load_library()
    keys = np.ascontiguousarray(keys, dtype=KEYPOINT_DTYPE)
    buf = C.c_void_p(); n = C.c_size_t()
    rc = L.hesaff_format_sift(keys.ctypes.data, len(keys), mr_size, C.byref(buf), C.byref(n))
    if rc != 0:
        raise HesaffError(rc, "hesaff_format_sift")
    try:
        return C.string_at(buf.value, n.value)
    finally:
        L.hesaff_free(buf)


def format_sift_mt(keys, mr_size, threads=0):
    """Same bytes as format_sift, rows formatted by `threads` host threads (0 = auto)."""
    L = load_library()
    keys = np.ascontiguousarray(keys, dtype=KEYPOINT_DTYPE)
    buf = C.c_void_p(); n = C.c_size_t()
    rc = L.hesaff_format_sift_mt(keys.ctypes.data, len(keys), C.c_float(mr_size), threads, C.byref(buf), C.byref(n))
    if rc != 0:
        raise HesaffError(rc, "hesaff_format_sift_mt")
    try:
        return C.string_at(buf.value, n.value)
    finally:
        L.hesaff_free(buf)


def write_sift_batch(paths, key_arrays, mr_size, threads=0):
    """One .hesaff.sift per image: key_arrays[i] (KEYPOINT_DTYPE) -> paths[i], images spread over host threads."""
    L = load_library()
    n = len(paths)
    arrs = [np.ascontiguousarray(k, dtype=KEYPOINT_DTYPE) for k in key_arrays]
    res = (_Result * n)()
    for i, a in enumerate(arrs):
        res[i].count_hessian = len(a); res[i].count_desc = len(a); res[i].keys = a.ctypes.data
    cp = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    rc = L.hesaff_write_sift_batch(n, cp, res, C.c_float(mr_size), threads)
    if rc != 0:
        raise HesaffError(rc, "hesaff_write_sift_batch")


def write_sift(path, keys, mr_size):
    keys = np.ascontiguousarray(keys, dtype=KEYPOINT_DTYPE)
    rc = load_library().hesaff_write_sift(os.fsencode(path), keys.ctypes.data, len(keys), mr_size)
    if rc != 0:
        raise HesaffError(rc, "hesaff_write_sift(%s)" % path)


BIN_ROW_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("a", "<f4"), ("b", "<f4"), ("c", "<f4"), ("desc", "u1", (128,))])


def write_bin(path, keys, mr_size):
    keys = np.ascontiguousarray(keys, dtype=KEYPOINT_DTYPE)
    rc = load_library().hesaff_write_bin(os.fsencode(path), keys.ctypes.data, len(keys), mr_size)
    if rc != 0:
        raise HesaffError(rc, "hesaff_write_bin(%s)" % path)


def read_bin(path):
    """.hesaff.bin (hesaff_write_bin) -> structured array of BIN_ROW_DTYPE."""
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) != 16 or head[:8] != b"HESAFFB1":
            raise HesaffError(-4, "%s is not a .hesaff.bin file" % path)
        dim, n = np.frombuffer(head[8:], "<u4")
        if dim != 128:
            raise HesaffError(-4, "descriptor dimension %d" % dim)
        rows = np.frombuffer(f.read(), dtype=BIN_ROW_DTYPE)
    if len(rows) != n:
        raise HesaffError(-4, "%s is truncated" % path)
    return rows


def read_image(path):
    """PGM/PPM or PNG by magic number -> uint8 array HxW (grey) or HxWx3."""
    return read_pnm(path, _fn="hesaff_read_image")


def read_jpeg_coefficients(path):
    """The host half of the JPEG reader (entropy decoding only) -> (JpegLayout, blob as a uint8 array)."""
    L = load_library()
    lay = JpegLayout(); blob = C.c_void_p(); nb = C.c_size_t()
    rc = L.hesaff_read_jpeg_coefficients(os.fsencode(path), C.byref(lay), C.byref(blob), C.byref(nb))
    if rc != 0:
        raise HesaffError(rc, "hesaff_read_jpeg_coefficients(%s)" % path)
    try:
        arr = np.frombuffer(C.string_at(blob.value, nb.value), dtype=np.uint8).copy()
    finally:
        L.hesaff_free(blob)
    return lay, arr


def read_pnm(path, _fn="hesaff_read_pnm"):
    L = load_library()
    data = C.c_void_p(); w = C.c_int(); h = C.c_int(); ch = C.c_int()
    rc = getattr(L, _fn)(os.fsencode(path), C.byref(data), C.byref(w), C.byref(h), C.byref(ch))
    if rc != 0:
        raise HesaffError(rc, "%s(%s)" % (_fn, path))
    try:
        n = w.value * h.value * ch.value
        arr = np.frombuffer(C.string_at(data.value, n), dtype=np.uint8).copy()
    finally:
        L.hesaff_free(data)
    return arr.reshape((h.value, w.value) if ch.value == 1 else (h.value, w.value, 3))


class HesaffContext:
    """One device context (hesaff_create / hesaff_destroy)."""

    def __init__(self, params=None, device=0):
        self.L = load_library()
        self.params = params if params is not None else default_params()
        self.h = C.c_void_p()
        rc = self.L.hesaff_create(C.byref(self.h), C.byref(self.params), device)
        if rc != 0:
            raise HesaffError(rc, self.L.hesaff_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.hesaff_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise HesaffError(rc, self.L.hesaff_last_error(self.h).decode())

    # ---- whole path ----
    def detect_batch(self, images):
        """images: list of uint8 arrays HxW (grey) or HxWx3.  -> list of (count_hessian, keys[KEYPOINT_DTYPE])."""
        n = len(images)
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
        chs = (C.c_int * n)(*[1 if im.ndim == 2 else 3 for im in imgs])
        st = (C.c_int * n)(*[im.shape[1] * (1 if im.ndim == 2 else 3) for im in imgs])
        res = (_Result * n)()
        self._check(self.L.hesaff_detect_batch(self.h, n, ptrs, ws, hs, st, chs, res))
        out = []
        for r in res:
            if r.count_desc > 0:
                # one copy out of the library-owned (pinned) result buffer, valid until the next call
                keys = np.frombuffer((C.c_char * (r.count_desc * 164)).from_address(r.keys), dtype=KEYPOINT_DTYPE).copy()
            else:
                keys = np.zeros(0, KEYPOINT_DTYPE)
            out.append((r.count_hessian, keys))
        return out

    def detect_batch_raw(self, images):
        """hesaff_detect_batch without copying the records out: -> ctypes array of hesaff_result whose `keys` point into
        library-owned pinned memory (valid until the next call on this context)."""
        n = len(images)
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
        chs = (C.c_int * n)(*[1 if im.ndim == 2 else 3 for im in imgs])
        st = (C.c_int * n)(*[im.shape[1] * (1 if im.ndim == 2 else 3) for im in imgs])
        res = (_Result * n)()
        self._check(self.L.hesaff_detect_batch(self.h, n, ptrs, ws, hs, st, chs, res))
        return res

    def detect_batch_cb(self, images, sink):
        """hesaff_detect_batch_cb: sink(image_indices, [(count_hessian, keys copy), ...]) is called once per chunk with
        records that are valid only during the call (bounded pinned memory); a truthy return value stops the run."""
        n = len(images)
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
        chs = (C.c_int * n)(*[1 if im.ndim == 2 else 3 for im in imgs])
        st = (C.c_int * n)(*[im.shape[1] * (1 if im.ndim == 2 else 3) for im in imgs])

        def _sink(_user, m, idx, res):
            out = []
            for i in range(m):
                r = res[i]
                if r.count_desc > 0:
                    buf = (C.c_char * (r.count_desc * KEYPOINT_DTYPE.itemsize)).from_address(r.keys)
                    keys = np.frombuffer(buf, dtype=KEYPOINT_DTYPE).copy()
                else:
                    keys = np.zeros(0, KEYPOINT_DTYPE)
                out.append((r.count_hessian, keys))
            return 1 if sink([idx[i] for i in range(m)], out) else 0
        cb = CHUNK_SINK(_sink)
        self._check(self.L.hesaff_detect_batch_cb(self.h, n, ptrs, ws, hs, st, chs, cb, None))

    def set_output_format(self, fmt):
        """1 = text (.hesaff.sift, default), 2 = binary sidecar (.hesaff.bin), 3 = both."""
        self._check(self.L.hesaff_set_output_format(self.h, fmt))

    def set_resume(self, on=True, strict=False):
        """hesaff_process_files skips images whose complete output exists (strict: the rows of an existing text file are counted too)."""
        self._check(self.L.hesaff_set_resume(self.h, (2 if strict else 1) if on else 0))

    def set_pinned_read_budget(self, max_bytes, keep_bytes):
        """page-locked read buffers of process_files: at most max_bytes at any time, keep_bytes kept from call to call"""
        self._check(self.L.hesaff_set_pinned_read_budget(self.h, max_bytes, keep_bytes))

    def set_pool_priority(self, mode):
        """-1: the pool of process_files steps down (nice 10) when the host plan is CPU-starved (default); 0 never; 1 always"""
        self._check(self.L.hesaff_set_pool_priority(self.h, mode))

    def process_files(self, paths, out_paths=None, decode_threads=0, write_threads=0):
        """hesaff_process_files: image files -> <name>.hesaff.sift through the decode / device / write pipeline.
        -> list of (rc, stage, count_hessian, count_desc) per file."""
        n = len(paths)
        cp = (C.c_char_p * n)(*[os.fsencode(q) for q in paths])
        op = None
        if out_paths is not None:
            op = (C.c_char_p * n)(*[None if q is None else os.fsencode(q) for q in out_paths])
        st = (FileStatus * n)()
        self._check(self.L.hesaff_process_files(self.h, n, cp, op, decode_threads, write_threads, st))
        return [(s.rc, s.stage, s.count_hessian, s.count_desc) for s in st]

    def write_sift_batch_raw(self, paths, results, mr_size, threads=0):
        """hesaff_write_sift_batch on hesaff_result records (e.g. a slice of detect_batch_raw's return value)."""
        n = len(paths)
        arr = (_Result * n)(*[results[i] for i in range(n)])
        cp = (C.c_char_p * n)(*[os.fsencode(q) for q in paths])
        rc = self.L.hesaff_write_sift_batch(n, cp, arr, C.c_float(mr_size), threads)
        if rc != 0:
            raise HesaffError(rc, "hesaff_write_sift_batch")

    def detect_batch_device(self, d_ptr, n, width, height):
        """Inputs resident in HBM (uint8 [n,H,W], raw device pointer).  -> (count_hessian[n], count_desc[n], d_keys, total)."""
        ch = np.zeros(n, np.int32); cd = np.zeros(n, np.int32)
        dk = C.c_void_p(); tot = C.c_int64()
        self._check(self.L.hesaff_detect_batch_device(self.h, n, C.c_void_p(d_ptr), width, height, ch, cd, C.byref(dk), C.byref(tot)))
        return ch, cd, dk.value, tot.value

    def set_profiling(self, level):
        self._check(self.L.hesaff_set_profiling(self.h, level))

    def timings(self):
        t = Timings()
        self._check(self.L.hesaff_get_timings(self.h, C.byref(t)))
        return t

    # ---- stage entry points (one reference operator each) ----
    def gaussian_blur(self, img, sigma):
        img = np.ascontiguousarray(img, np.float32); out = np.empty_like(img)
        self._check(self.L.hesaff_stage_gaussian_blur(self.h, img, img.shape[0], img.shape[1], sigma, out))
        return out

    def hessian_response(self, img, norm):
        img = np.ascontiguousarray(img, np.float32); out = np.empty_like(img)
        self._check(self.L.hesaff_stage_hessian_response(self.h, img, img.shape[0], img.shape[1], norm, out))
        return out

    def half_image(self, img):
        img = np.ascontiguousarray(img, np.float32)
        out = np.empty((img.shape[0] // 2, img.shape[1] // 2), np.float32)
        self._check(self.L.hesaff_stage_half_image(self.h, img, img.shape[0], img.shape[1], out))
        return out

    def pyramid(self, gray_u8):
        """-> list over octaves of (L[5,rows,cols], R[5,rows,cols])."""
        g = np.ascontiguousarray(gray_u8, np.uint8)
        no = C.c_int(); nf = C.c_size_t()
        self._check(self.L.hesaff_stage_pyramid(self.h, None, g.shape[0], g.shape[1], None, C.byref(no), C.byref(nf)))
        buf = np.empty(max(nf.value, 1), np.float32)
        self._check(self.L.hesaff_stage_pyramid(self.h, g.ctypes.data, g.shape[0], g.shape[1], buf.ctypes.data, C.byref(no), C.byref(nf)))
        out = []; off = 0; r, c = g.shape
        for _ in range(no.value):
            n = r * c
            Ls = buf[off:off + 5 * n].reshape(5, r, c); off += 5 * n
            Rs = buf[off:off + 5 * n].reshape(5, r, c); off += 5 * n
            out.append((Ls, Rs))
            r //= 2; c //= 2
        return out

    def hessian_keypoints(self, gray_u8, cap=None):
        """-> f[n,5] = x,y,s,pd,response ; i[n,5] = type,octave,level,r0,c0 (reference order)."""
        g = np.ascontiguousarray(gray_u8, np.uint8)
        if cap is None:
            cap = max(4096, int(g.size * 0.05))
        f = np.zeros((cap, 5), np.float32); i = np.zeros((cap, 5), np.int32); cnt = C.c_int()
        self._check(self.L.hesaff_stage_hessian_keypoints(self.h, g, g.shape[0], g.shape[1], cap, f, i, C.byref(cnt)))
        n = min(cnt.value, cap)
        return f[:n].copy(), i[:n].copy(), cnt.value

    def find_affine_shape(self, blur, kp):
        blur = np.ascontiguousarray(blur, np.float32); kp = np.ascontiguousarray(kp, np.float32).reshape(-1, 4)
        n = len(kp)
        conv = np.zeros(n, np.int32); U = np.zeros((n, 4), np.float32); it = np.zeros(n, np.int32)
        self._check(self.L.hesaff_stage_find_affine_shape(self.h, blur, blur.shape[0], blur.shape[1], n, kp, conv, U, it))
        return conv, U, it

    def rectify(self, A):
        A = np.ascontiguousarray(A, np.float32).reshape(-1, 4).copy()
        self._check(self.L.hesaff_stage_rectify(self.h, len(A), A))
        return A

    def normalize_affine(self, img, kp, A):
        img = np.ascontiguousarray(img, np.float32); kp = np.ascontiguousarray(kp, np.float32).reshape(-1, 3)
        A = np.ascontiguousarray(A, np.float32).reshape(-1, 4)
        n = len(kp)
        rej = np.zeros(n, np.int32); patches = np.zeros((n, 41 * 41), np.float32)
        self._check(self.L.hesaff_stage_normalize_affine(self.h, img, img.shape[0], img.shape[1], n, kp, A, rej, patches))
        return rej, patches.reshape(n, 41, 41)

    def sift(self, patches):
        p = np.ascontiguousarray(patches, np.float32).reshape(-1, 41 * 41)
        d = np.zeros((len(p), 128), np.uint8)
        self._check(self.L.hesaff_stage_sift(self.h, len(p), p, d))
        return d

    def export(self, keys, mr_size=None, fmt=1):
        """exportKeypoints on the device (hesaff_stage_export): bytes of the .hesaff.sift file (fmt 1) or of the sidecar (fmt 2)."""
        keys = np.ascontiguousarray(keys, dtype=KEYPOINT_DTYPE)
        buf = C.c_void_p(); n = C.c_size_t()
        mr = self.params.mrSize if mr_size is None else mr_size
        self._check(self.L.hesaff_stage_export(self.h, keys.ctypes.data, len(keys), C.c_float(mr), fmt, C.byref(buf), C.byref(n)))
        try:
            return C.string_at(buf.value, n.value)
        finally:
            self.L.hesaff_free(buf)

    def fmt_g(self, v):
        """The device's "%g" print of float32 values -> list of bytes."""
        v = np.ascontiguousarray(v, np.float32).reshape(-1)
        text = np.zeros((len(v), 16), np.uint8); lens = np.zeros(len(v), np.int32)
        self._check(self.L.hesaff_stage_fmt_g(self.h, len(v), v, text.ctypes.data, lens))
        return text, lens

    def jpeg_pixels(self, layout, blobs):
        """The device half of the JPEG reader: blobs [n, blob_bytes] uint8 of one layout -> pixels [n, H, W(, 3)] uint8."""
        blobs = np.ascontiguousarray(blobs, np.uint8)
        if blobs.ndim == 1:
            blobs = blobs[None]
        n = blobs.shape[0]
        shape = (n, layout.height, layout.width) + ((3,) if layout.channels == 3 else ())
        out = np.zeros(shape, np.uint8)
        self._check(self.L.hesaff_stage_jpeg_pixels(self.h, C.byref(layout), n, blobs.ctypes.data, blobs.shape[1], out.ctypes.data))
        return out

    def math(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        at = np.zeros_like(a); pw = np.zeros_like(a)
        self._check(self.L.hesaff_stage_math(self.h, a.size, a.reshape(-1), b.reshape(-1), at.reshape(-1), pw.reshape(-1)))
        return at, pw

    def math_sift(self, gy, gx):
        """-> ori_general, ori_nd, grad_general, grad_nd (hesaff_stage_math_sift)."""
        gy = np.ascontiguousarray(gy, np.float32).reshape(-1); gx = np.ascontiguousarray(gx, np.float32).reshape(-1)
        outs = [np.zeros_like(gy) for _ in range(4)]
        self._check(self.L.hesaff_stage_math_sift(self.h, gy.size, gy, gx, *outs))
        return outs
