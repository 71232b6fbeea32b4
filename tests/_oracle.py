"""ctypes binding of the CPU oracle (oracle/libhesaff_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under hesaff_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libhesaff_oracle.so")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build():
    src = os.path.join(_ROOT, "oracle", "hesaff_oracle.cpp")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.ho_gauss_ksize.argtypes = [C.c_float]; L.ho_gauss_ksize.restype = C.c_int
    L.ho_gauss_kernel.argtypes = [C.c_int, C.c_float, f32p]
    L.ho_gaussian_blur.argtypes = [f32p, C.c_int, C.c_int, C.c_float, f32p]
    L.ho_hessian_response.argtypes = [f32p, C.c_int, C.c_int, C.c_float, f32p]
    L.ho_half_image.argtypes = [f32p, C.c_int, C.c_int, f32p]
    L.ho_double_image.argtypes = [f32p, C.c_int, C.c_int, f32p]
    L.ho_gray_from_u8.argtypes = [u8p, C.c_int, C.c_int, f32p]
    L.ho_interpolate.argtypes = [f32p, C.c_int, C.c_int] + [C.c_float] * 6 + [f32p, C.c_int, C.c_int]
    L.ho_interpolate.restype = C.c_int
    L.ho_solve_linear3x3.argtypes = [f32p, f32p]
    L.ho_inv_sqrt.argtypes = [f32p, f32p]
    L.ho_get_eigenvalues.argtypes = [C.c_float] * 4 + [f32p]; L.ho_get_eigenvalues.restype = C.c_int
    L.ho_rectify.argtypes = [f32p]
    L.ho_gauss_mask.argtypes = [C.c_int, f32p]
    L.ho_circ_gauss_mask.argtypes = [C.c_int, f32p]
    L.ho_sift_tables.argtypes = [i32p, i32p, f32p, f32p]
    L.ho_atan2f.argtypes = [C.c_float, C.c_float]; L.ho_atan2f.restype = C.c_float
    L.ho_pow2f.argtypes = [C.c_float]; L.ho_pow2f.restype = C.c_float
    L.ho_find_affine_shape.argtypes = [f32p, C.c_int, C.c_int] + [C.c_float] * 4 + [f32p, i32p]
    L.ho_find_affine_shape.restype = C.c_int
    L.ho_normalize_affine.argtypes = [f32p, C.c_int, C.c_int] + [C.c_float] * 3 + [f32p, f32p]
    L.ho_normalize_affine.restype = C.c_int
    L.ho_sift.argtypes = [f32p, f32p]
    L.ho_create.restype = C.c_void_p
    L.ho_destroy.argtypes = [C.c_void_p]
    L.ho_set_keep_planes.argtypes = [C.c_void_p, C.c_int]
    L.ho_set_detect_only.argtypes = [C.c_void_p, C.c_int]
    L.ho_detect.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int]
    L.ho_set_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float]
    L.ho_set_upscale.argtypes = [C.c_void_p, C.c_int]
    L.ho_h_find_affine_shape.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int] + [C.c_float] * 4 + [f32p, i32p]
    L.ho_h_find_affine_shape.restype = C.c_int
    L.ho_h_normalize_affine.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int] + [C.c_float] * 3 + [f32p, f32p]
    L.ho_h_normalize_affine.restype = C.c_int
    L.ho_h_sift.argtypes = [C.c_void_p, f32p, f32p]
    for n in ("ho_num_hessian", "ho_num_keys", "ho_num_octaves"):
        getattr(L, n).argtypes = [C.c_void_p]; getattr(L, n).restype = C.c_int
    L.ho_num_candidates.argtypes = [C.c_void_p]; L.ho_num_candidates.restype = C.c_long
    L.ho_get_hessian.argtypes = [C.c_void_p, C.c_int, f32p, i32p]
    L.ho_get_affine.argtypes = [C.c_void_p, C.c_int, f32p, i32p]
    L.ho_get_key.argtypes = [C.c_void_p, C.c_int, f32p, i32p, u8p]
    L.ho_get_keys.argtypes = [C.c_void_p, f32p, i32p, u8p]
    L.ho_plane_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ho_get_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, f32p]
    L.ho_export.argtypes = [C.c_void_p, C.c_char_p, C.c_long]; L.ho_export.restype = C.c_long
    L.ho_ellipse.argtypes = [f32p, C.c_float, f32p]
    _lib = L
    return L


def gray_from_u8(img):
    """hesaff.cpp:138-148 grey conversion; img is HxW (grey) or HxWx3 (BGR) uint8."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ch = 1 if img.ndim == 2 else 3
    out = np.empty(img.shape[:2], dtype=np.float32)
    lib().ho_gray_from_u8(img.reshape(-1), out.size, ch, out.reshape(-1))
    return out


def set_params(h, params):
    """Copy the reference-side fields of a hesaff_params-like object (hesaff_amd.Params) into an oracle handle."""
    lib().ho_set_params(h, params.threshold, params.edgeEigenValueRatio, params.initialSigma, int(params.maxIterations),
                        params.convergenceThreshold, params.mrSize, params.maxBinValue)
    lib().ho_set_upscale(h, int(getattr(params, "upscaleInputImage", 0)))


class OracleHandle:
    """An oracle object with (optionally non-default) parameters, for the stage-level functions."""

    def __init__(self, params=None):
        self.h = lib().ho_create()
        if params is not None:
            set_params(self.h, params)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ho_destroy(self.h)
            self.h = None

    def normalize_affine(self, gray, x, y, s, A):
        patch = np.zeros(41 * 41, np.float32)
        rej = lib().ho_h_normalize_affine(self.h, gray, gray.shape[0], gray.shape[1], float(x), float(y), float(s),
                                          np.ascontiguousarray(A, np.float32), patch)
        return rej, patch.reshape(41, 41)

    def find_affine_shape(self, blur, x, y, s, pd):
        A = np.zeros(4, np.float32); it = np.zeros(1, np.int32)
        conv = lib().ho_h_find_affine_shape(self.h, blur, blur.shape[0], blur.shape[1], float(x), float(y), float(s), float(pd), A, it)
        return conv, A, int(it[0])

    def sift(self, patch):
        p = np.ascontiguousarray(patch, np.float32).reshape(-1).copy()
        vec = np.zeros(128, np.float32)
        lib().ho_h_sift(self.h, p, vec)
        return vec.astype(np.uint8)


class OracleRun:
    """Full reference-order pipeline on one float32 grey image."""

    def __init__(self, gray, keep_planes=False, detect_only=False, params=None):
        L = lib()
        gray = np.ascontiguousarray(gray, dtype=np.float32)
        self.h = L.ho_create()
        if params is not None:
            set_params(self.h, params)
        L.ho_set_keep_planes(self.h, int(keep_planes))
        L.ho_set_detect_only(self.h, int(detect_only))
        L.ho_detect(self.h, gray, gray.shape[0], gray.shape[1])
        self.n_hessian = L.ho_num_hessian(self.h)
        self.n_keys = L.ho_num_keys(self.h)
        self.n_candidates = L.ho_num_candidates(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ho_destroy(self.h)
            self.h = None

    def hessian(self):
        """-> (f[n,6] = x,y,s,pd,response,0 ; i[n,5] = type,octave,level,r0,c0)"""
        f = np.zeros((self.n_hessian, 6), np.float32)
        i = np.zeros((self.n_hessian, 5), np.int32)
        for k in range(self.n_hessian):
            lib().ho_get_hessian(self.h, k, f[k], i[k])
        return f, i

    def affine(self):
        """-> (U[n,4] un-rectified, i[n,2] = converged, iters)"""
        f = np.zeros((self.n_hessian, 4), np.float32)
        i = np.zeros((self.n_hessian, 2), np.int32)
        for k in range(self.n_hessian):
            lib().ho_get_affine(self.h, k, f[k], i[k])
        return f, i

    def keys(self):
        """-> geom[n,8] = x,y,s,a11,a12,a21,a22,response ; type[n] ; desc[n,128] u8"""
        n = self.n_keys
        g = np.zeros((n, 8), np.float32)
        t = np.zeros((n,), np.int32)
        d = np.zeros((n, 128), np.uint8)
        if n:
            lib().ho_get_keys(self.h, g, t, d)
        return g, t, d

    def key_sources(self):
        n = self.n_keys
        src = np.zeros(n, np.int32)
        f = np.zeros(8, np.float32); i = np.zeros(2, np.int32); d = np.zeros(128, np.uint8)
        for k in range(n):
            lib().ho_get_key(self.h, k, f, i, d)
            src[k] = i[1]
        return src

    def n_octaves(self):
        return lib().ho_num_octaves(self.h)

    def plane(self, octave, which, level):
        r = C.c_int(); c = C.c_int()
        lib().ho_plane_dims(self.h, octave, C.byref(r), C.byref(c))
        out = np.empty((r.value, c.value), np.float32)
        lib().ho_get_plane(self.h, octave, which, level, out)
        return out

    def export_text(self):
        n = lib().ho_export(self.h, None, 0)
        buf = C.create_string_buffer(n)
        lib().ho_export(self.h, buf, n)
        return buf.raw[:n]
