// device_common.h -- scalar device helpers of the hesaff hot path (gfx950).
//
// Each function states the reference expression it evaluates (file:line under the
// reference tree) and keeps its float operation ORDER: this translation unit is compiled
// with -ffp-contract=off, IEEE division/sqrt and denormals on, so a given expression
// tree rounds exactly like the reference's scalar x86-64 build.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hmath.h"

#define HS_PATCH 41                 // patchSize, affine.h:42 / siftdesc.h:30
#define HS_PATCH_PIX (41 * 41)
#define HS_SMM 19                   // smmWindowSize, affine.h:43
#define HS_SMM_PIX (19 * 19)
#define HS_BORDER 5                 // PyramidParams::border, pyramid.h:39
#define HS_NSCALES 3                // numberOfScales, pyramid.h:35
#define HS_MAX_OCTAVES 16

// A batch of equally sized float planes: [img][rows][pitch]
struct DPlane {
   float *p;
   int rows, cols, pitch;
   long long img_stride;   // floats between consecutive images
   __host__ __device__ float *img(int b) const { return p + (long long)b * img_stride; }
};

// constants of one context, uploaded once (tables) / per call (thresholds)
struct DConsts {
   float edgeScoreThreshold, finalThreshold, positiveThreshold, negativeThreshold;  // pyramid.h:60-64
   float convergenceThreshold;  // affine.h:41
   float affInitialSigma;       // affine.h:40
   float mrSize;                // affine.h:44
   float maxBinValue;           // siftdesc.h:29
   int maxIterations;           // affine.h:39
   float pd0;                   // pixelDistance of octave 0: 1, or 0.5 with upscaleInputImage (pyramid.cpp:264,270)
};

// Pins a value: everything it depends on (in particular its global loads) is issued before this
// point instead of being sunk by the compiler into a later conditional block, where each load
// would be followed by its own full memory round trip.
#define HS_KEEP(x) asm volatile("" : "+v"(x))

// Ordering of LDS traffic between the lanes of ONE wavefront: the LDS executes a wave's
// instructions in order, so a compiler-level barrier (no reordering of memory operations) plus
// draining the LDS counter is enough.  A wavefront-scope C++ fence would also wait for the
// wave's outstanding GLOBAL stores (vmcnt(0)) - a full memory round trip per window row.
#define HS_WAVE_LDS_SYNC()                                  \
   do {                                                     \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
      __builtin_amdgcn_wave_barrier();                      \
   } while (0)

// Streaming accesses: data written once and consumed by a later kernel (or read exactly once) is stored / loaded
// non-temporally so that it does not displace what this kernel and its neighbours re-read from L2 / the memory-side cache
// (measured: k_sift_grad 21.6 -> 19.9 ms from its 12.8 KB of gradient pairs per keypoint alone).  Each site has a switch
// (HS_NT_*) for A/B runs of the tuning build.
typedef float hs_nt2 __attribute__((ext_vector_type(2)));
typedef float hs_nt4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void hs_store_nt(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void hs_store_nt2(float *p, float a, float b) { hs_nt2 v; v.x = a; v.y = b; __builtin_nontemporal_store(v, reinterpret_cast<hs_nt2 *>(p)); }
__device__ __forceinline__ void hs_store_nt4(float *p, float a, float b, float c, float d)
{
   hs_nt4 v; v.x = a; v.y = b; v.z = c; v.w = d;
   __builtin_nontemporal_store(v, reinterpret_cast<hs_nt4 *>(p));
}
__device__ __forceinline__ float hs_load_nt(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float4 hs_load_nt4(const float4 *p)
{
   const hs_nt4 v = __builtin_nontemporal_load(reinterpret_cast<const hs_nt4 *>(p));
   return make_float4(v.x, v.y, v.z, v.w);
}
#ifndef HS_NT_VO
#define HS_NT_VO 1        // k_sift_grad's gradient pairs: stores (21.3 -> 19.9 ms; k_sift_hist 14.1 -> 13.3)
#endif
#ifndef HS_NT_VO_LD
#define HS_NT_VO_LD 0     // ... and k_sift_hist's loads of them (measured: 13.7 -> 17.1 ms, not used)
#endif
#ifndef HS_NT_PATCH
#define HS_NT_PATCH 1     // patch kernels' 41 x 41 outputs (the kernels themselves unchanged, k_sift_grad 20.1 -> 18.7 ms)
#endif
#ifndef HS_NT_PYR
#define HS_NT_PYR 1       // pyramid planes (k_blur_hess_march<9 / 13 / 15>: -16 / -9 / -2 %)
#endif
#ifndef HS_NT_PYR_R
#define HS_NT_PYR_R HS_NT_PYR   // ... the response planes separately (read back by k_extrema_march right after the octave's blurs)
#endif
#ifndef HS_NT_GRAY
#define HS_NT_GRAY 1      // the float grey plane written by the initial blur (pyramid stage 18.4-19.6 -> 17.8-18.2 ms at B = 128)
#endif
#ifndef HS_NT_EXT
#define HS_NT_EXT 0       // k_extrema_march's reads of the five response planes
#endif
#ifndef HS_NT_SGRAD_LD
#define HS_NT_SGRAD_LD 0  // k_sift_grad's reads of the patch (its last reader)
#endif
#ifndef HS_NT_MEANVAR
#define HS_NT_MEANVAR 0   // k_sift_meanvar's patch reads (1: second pass: no change; 2: both passes: 9.8 -> 11.8 ms)
#endif

// ---- helpers.cpp:227-240 : one bilinear tap; `outside` is OR-ed like `ret` ----
// (int)floor(w) of the reference is cvttss2si (INT_MIN on NaN/overflow -> "outside");
// comparing the floored float gives the same classification without the cast.
// Branch-free: an outside tap reads pixel (0,0) and discards the value, so that a loop of
// taps can issue all its loads before the first use.  Needs an image of at least 2x2.
__device__ __forceinline__ float hs_bilinear(const float *__restrict__ im, int pitch, int width, int height,
                                             float wx, float wy, bool &outside)
{
   const float fx = floorf(wx), fy = floorf(wy);
   const bool in = (fx >= 0.0f && fy >= 0.0f && fx < (float)width && fy < (float)height);
   const int x = in ? (int)fx : 0, y = in ? (int)fy : 0;
   wx -= fx;
   wy -= fy;
   const float *p = im + (long long)y * pitch + x;
   const float p00 = p[0], p01 = p[1], p10 = p[pitch], p11 = p[pitch + 1];
   const float v = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
   outside = outside || !in;
   return in ? v : 0.0f;
}

#ifndef HS_TAP_FRACT
#define HS_TAP_FRACT 1
#endif
// The tap of a window that lies inside the plane (all four corners tested: hs_window_outside): no bounds test, no selects,
// a 32-bit element offset.  Same arithmetic as the inside case of hs_bilinear.
__device__ __forceinline__ float hs_tap_inside_ptr(const float *__restrict__ im, int pitch, float wx, float wy)
{
#if HS_TAP_FRACT
   // inside the plane w >= 0: (int)floorf(w) == (int)w (conversion truncates) and w - floorf(w) is v_fract_f32's exact value;
   // two instructions per axis instead of three (helpers.cpp:227-232)
   const uint32_t off = (uint32_t)(int)wy * (uint32_t)pitch + (uint32_t)(int)wx;
   wx = __builtin_amdgcn_fractf(wx);
   wy = __builtin_amdgcn_fractf(wy);
#else
   const float fx = floorf(wx), fy = floorf(wy);
   wx -= fx;
   wy -= fy;
   const uint32_t off = (uint32_t)(int)fy * (uint32_t)pitch + (uint32_t)(int)fx;
#endif
   const float *p = im + off;
   const float p00 = p[0], p01 = p[1], p10 = p[pitch], p11 = p[pitch + 1];
   return (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
}

// The same tap through a buffer resource: one image plane (< 4 GB) described by four scalar
// registers, per-lane 32-bit byte offsets, the second row reached through the scalar offset
// operand - no 64-bit per-lane pointer arithmetic.  The plane base must be wave-uniform.
typedef unsigned int hs_v2u __attribute__((ext_vector_type(2)));
struct HsPlaneBuf {
   __amdgpu_buffer_rsrc_t rsrc;
   uint32_t pitch, pitch_bytes;
};
__device__ __forceinline__ HsPlaneBuf hs_plane_buf(const float *base, int rows, int pitch)
{
   const unsigned long long a = reinterpret_cast<unsigned long long>(base);
   const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)a), hi = __builtin_amdgcn_readfirstlane((unsigned int)(a >> 32));
   HsPlaneBuf b;
   b.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                              (int)((unsigned int)rows * (unsigned int)pitch * 4u), 0x00020000);
   b.pitch = (uint32_t)pitch;
   b.pitch_bytes = (uint32_t)pitch * 4u;
   return b;
}
__device__ __forceinline__ float hs_bilinear_buf(const HsPlaneBuf &im, int width, int height, float wx, float wy, bool &outside)
{
   const float fx = floorf(wx), fy = floorf(wy);
   const bool in = (fx >= 0.0f && fy >= 0.0f && fx < (float)width && fy < (float)height);
   wx -= fx;
   wy -= fy;
   const uint32_t off = in ? ((uint32_t)(int)fy * im.pitch + (uint32_t)(int)fx) * 4u : 0u;
   const hs_v2u r0 = __builtin_amdgcn_raw_buffer_load_b64(im.rsrc, (int)off, 0, 0);
   const hs_v2u r1 = __builtin_amdgcn_raw_buffer_load_b64(im.rsrc, (int)off, (int)im.pitch_bytes, 0);
   const float p00 = __uint_as_float(r0.x), p01 = __uint_as_float(r0.y), p10 = __uint_as_float(r1.x), p11 = __uint_as_float(r1.y);
   const float v = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
   outside = outside || !in;
   return in ? v : 0.0f;
}

// ---- helpers.cpp:46-88 solveLinear3x3 (value swaps, partial pivoting) ----
__device__ __forceinline__ void hs_swap(float &a, float &b) { const float t = a; a = b; b = t; }
__device__ __forceinline__ void hs_solve3x3(float *A, float *b)
{
   // Scalars and selects, not the arrays: the compiler turns "if (i == 1) swap(A[3], A[0]) ... if (i == 2) swap(A[6], A[0])" into ONE swap
   // at the dynamic index 3 * i, which sends A and b to scratch memory (48 bytes per lane in k_localize) - and a kernel that uses scratch
   // costs the host a runtime thread that is busy for as long as the device works (profiles/r05_notes.md).  The row exchanges move values
   // only, so the arithmetic and its order are the reference's.
   float a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5], a6 = A[6], a7 = A[7], a8 = A[8], b0 = b[0], b1 = b[1], b2 = b[2];
   float vp = fabsf(a0);
   const float tmp = fabsf(a3);
   bool p1 = false, p2 = false;   // pivot row: 1, 2 (else 0)
   if (tmp > vp) { p1 = true; vp = tmp; }
   if (fabsf(a6) > vp) { p1 = false; p2 = true; }
   {  // row 0 <-> pivot row
      const float r0 = p1 ? a3 : (p2 ? a6 : a0), r1 = p1 ? a4 : (p2 ? a7 : a1), r2 = p1 ? a5 : (p2 ? a8 : a2), rb = p1 ? b1 : (p2 ? b2 : b0);
      a3 = p1 ? a0 : a3; a4 = p1 ? a1 : a4; a5 = p1 ? a2 : a5; b1 = p1 ? b0 : b1;
      a6 = p2 ? a0 : a6; a7 = p2 ? a1 : a7; a8 = p2 ? a2 : a8; b2 = p2 ? b0 : b2;
      a0 = r0; a1 = r1; a2 = r2; b0 = rb;
   }
   vp = a3 / a0; a4 -= vp * a1; a5 -= vp * a2; b1 -= vp * b0;
   vp = a6 / a0; a7 -= vp * a1; a8 -= vp * a2; b2 -= vp * b0;
   if (fabsf(a4) < fabsf(a7)) { hs_swap(a7, a4); hs_swap(a8, a5); hs_swap(b2, b1); }
   vp = a7 / a4;
   a8 -= vp * a5;
   b2 -= vp * b1;
   b2 = (b2) / a8;
   b1 = (b1 - a5 * b2) / a4;
   b0 = (b0 - a2 * b2 - a1 * b1) / a0;
   b[0] = b0; b[1] = b1; b[2] = b2;
}

// ---- helpers.cpp:149-175 invSqrt (double inside) ----
__device__ inline void hs_inv_sqrt(float &a, float &b, float &c, float &l1, float &l2)
{
   double t, r;
   if (b != 0) {
      r = double(c - a) / (2 * b);
      if (r >= 0) t = 1.0 / (r + sqrt(1 + r * r));
      else t = -1.0 / (-r + sqrt(1 + r * r));
      r = 1.0 / sqrt(1 + t * t);
      t = t * r;
   } else {
      r = 1;
      t = 0;
   }
   double x = 1.0 / sqrt(r * r * a - 2 * r * t * b + t * t * c);
   double z = 1.0 / sqrt(t * t * a + 2 * r * t * b + r * r * c);
   const double d = sqrt(x * z);
   x /= d;
   z /= d;
   if (x < z) { l1 = float(z); l2 = float(x); } else { l1 = float(x); l2 = float(z); }
   a = float(r * r * x + t * t * z);
   b = float(-r * t * x + t * r * z);
   c = float(t * t * x + r * r * z);
}

// ---- helpers.cpp:177-188 getEigenvalues ----
__device__ __forceinline__ bool hs_eigenvalues(float a, float b, float c, float d, float &l1, float &l2)
{
   const float trace = a + d;
   const float delta1 = (trace * trace - 4 * (a * d - b * c));
   if (delta1 < 0) return false;
   const float delta = sqrtf(delta1);
   l1 = (trace + delta) / 2.0f;
   l2 = (trace - delta) / 2.0f;
   return true;
}

// ---- helpers.cpp:90-97 rectifyAffineTransformationUpIsUp (double inside) ----
__device__ inline void hs_rectify(float &a11, float &a12, float &a21, float &a22)
{
   const double a = a11, b = a12, c = a21, d = a22;
   const double det = sqrt(fabs(a * d - b * c));
   const double b2a2 = sqrt(b * b + a * a);
   a11 = (float)(b2a2 / det);
   a12 = 0;
   a21 = (float)((d * b + c * a) / (b2a2 * det));
   a22 = (float)(det / b2a2);
}

// ---- helpers.cpp:191-207 interpolateCheckBorders for a 41x41 result ----
__device__ inline bool hs_check_borders(int imRows, int imCols, float ofsx, float ofsy, float a11, float a12,
                                        float a21, float a22)
{
   const int width = imCols - 2, height = imRows - 2;
   const float h = (float)(HS_PATCH >> 1);
   const float xs[4] = {-h, -h, h, h};
   const float ys[4] = {-h, h, -h, h};
   for (int i = 0; i < 4; i++) {
      const float imx = ofsx + xs[i] * a11 + ys[i] * a12;
      const float imy = ofsy + xs[i] * a21 + ys[i] * a22;
      if (floorf(imx) <= 0 || floorf(imy) <= 0 || ceilf(imx) >= width || ceilf(imy) >= height) return true;
   }
   return false;
}

// ---- affine.cpp:14-33 / siftdesc.cpp:123-134 gradient stencil on a size x size tile ----
__device__ __forceinline__ void hs_grad(const float *img, int size, int r, int c, float &gx, float &gy)
{
   if (c == 0) gx = img[r * size + c + 1] - img[r * size + c];
   else if (c == size - 1) gx = img[r * size + c] - img[r * size + c - 1];
   else gx = img[r * size + c + 1] - img[r * size + c - 1];
   if (r == 0) gy = img[(r + 1) * size + c] - img[r * size + c];
   else if (r == size - 1) gy = img[r * size + c] - img[(r - 1) * size + c];
   else gy = img[(r + 1) * size + c] - img[(r - 1) * size + c];
}

// ---- pyramid.cpp:95-100 det-of-Hessian at the centre of a 3x3 neighbourhood ----
__device__ __forceinline__ float hs_hessian(float v11, float v12, float v13, float v21, float v22, float v23,
                                            float v31, float v32, float v33, float norm2)
{
   const float Lxx = (v21 - 2 * v22 + v23);
   const float Lyy = (v12 - 2 * v22 + v32);
   const float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
   return (Lxx * Lyy - Lxy * Lxy) * norm2;
}
