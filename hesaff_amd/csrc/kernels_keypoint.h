// kernels_keypoint.h -- per-keypoint kernels: Baumberg affine iteration, up-is-up
// rectification + border test, affine patch normalisation, SIFT descriptor, result packing.
// Reference: affine.cpp, helpers.cpp, siftdesc.cpp, hesaff.cpp:72-105.
//
// Float sums that the reference accumulates sequentially (SMM sums affine.cpp:57-68,
// photometric mean/variance helpers.cpp:253-266, histogram cells siftdesc.cpp:75-78, norms
// siftdesc.cpp:86-90) are accumulated in the SAME order here: one lane per sum, or one
// thread per histogram cell walking its pixels in raster order.  No shuffle trees, no
// float atomics (both would change the rounding and flip descriptor bytes).
#pragma once
#include <type_traits>
#include "device_common.h"
#include "kernels_pyramid.h"

struct PlaneTab {   // prevBlur planes of the batch: L[octave][level], level 0..2
   DPlane L[HS_MAX_OCTAVES][HS_NSCALES];
};

struct KpTables {   // device tables built on the host once per context
   const float *smm_mask;    // 19x19 computeGaussMask, helpers.cpp:104
   const float *sift_mask;   // 41x41 computeCircularGaussMask, helpers.cpp:131
   const int32_t *mask_idx;  // raster-ordered indices of the pixels with sift_mask > 0
   int n_masked;
   // per masked pixel (slot order = mask_idx order, padded to 1280): the constants k_sift_grad needs for it, so that a
   // thread fetches them in one round of loads: {left, right, up, down} stencil neighbours as LDS byte offsets into the
   // patch, and {output slot r * 40 + c (or -1: row / column 40, no weight), mask value bits}
   const int4 *sgrad_nb;
   const int2 *sgrad_om;
   // layout of a keypoint's gradient pairs in HBM (kernels_sift.h: HS_VO_COMPACT): per patch row r < 40 {first item of the row in the
   // keypoint's block minus f_lo, f_lo, f_hi, 0} in 16-byte items (an empty row: f_lo > f_hi), and per item of the block the item of
   // k_sift_grad's 40 x 20 LDS tile it is a copy of (padding items: tile item 0, which holds no masked pixel)
   const int4 *vo_rows;
   const uint16_t *vo_src;
   const int32_t *bin0, *bin1;   // precomputeBinsAndWeights siftdesc.cpp:18 (already x8)
   const float *w0, *w1;
   const float *patch_taps;      // Gaussian taps of every odd P0, concatenated
   const int32_t *patch_tap_off; // offset of P0's taps at index (P0-1)/2
   const int32_t *patch_tap_k;   // K of P0
   int max_p0;
};

// ---------------------------------------------------------------------------------------
// k_affine: AffineShape::findAffineShape affine.cpp:35-100.
//  per iteration: 361 bilinear taps -> LDS img; gradients and the three products per pixel ->
//  LDS; three lanes add the 361 terms of a,b,c in index order; one lane runs the
//  double-precision invSqrt and the U update; result broadcast through LDS.
// ---------------------------------------------------------------------------------------
struct AffineOut {
   int32_t *converged;   // 1 = onAffineShapeFound was called
   float *U;             // [n][4] a11,a12,a21,a22
   int32_t *iters;
};

// ---------------------------------------------------------------------------------------
// hs_affine_groups: the same iteration with FOUR keypoints per wavefront (16 lanes each).
// Per iteration a keypoint needs 361 parallel taps + products, then three 361-term sequential
// sums and one double-precision invSqrt.  With one keypoint per wavefront the sums keep 3 lanes
// busy and the invSqrt one lane; with four keypoints side by side those serial stretches are
// shared by four (12 and 4 lanes busy) and the wavefront executes ~2.3x fewer instructions per
// keypoint.  The 16-lane groups are persistent: a group whose keypoint converged (or was
// rejected) takes the next keypoint at the end of the round, so keypoints with different
// iteration counts do not wait for each other.
// block = 64 threads (one wavefront), LDS 4 x 3 x 364 floats + mask (19 KB).
// ---------------------------------------------------------------------------------------
#define HS_AFF_G 4
// interpolate()'s return flag ("some tap fell outside the image", helpers.cpp:209-244) for the P x P window of
// normalizeAffine's smoothing branch (affine.cpp:126), without visiting the P^2 taps.  The tap coordinate
//    w(j, i) = fl(fl(ofs + fl(j * a_row)) + fl(i * a_col))        j, i in [-half, half]
// is monotone in i for fixed j and in j for fixed i (a float product or sum with one operand fixed is monotone under
// round-to-nearest), so over the grid it takes its extremes at the four corners; a tap is inside iff
// 0 <= floor(w) < limit, i.e. 0 <= w < limit for the integer limits cols - 1 / rows - 1.  Hence "some tap outside" <=>
// "some corner outside".  Non-finite values: an infinite product at an inner index is also infinite at the corner of the
// same sign, and a NaN operand makes every coordinate NaN, so a corner fails whenever any tap would.
__device__ inline bool hs_window_outside(int imRows, int imCols, float ofsx, float ofsy, float a11, float a12, float a21, float a22, int half)
{
   const float width = (float)(imCols - 1), height = (float)(imRows - 1);
   bool outside = false;
#pragma unroll
   for (int q = 0; q < 4; q++) {
      const int j = (q & 1) ? half : -half, i = (q & 2) ? half : -half;
      const float rx = ofsx + (float)j * a12, ry = ofsy + (float)j * a22;
      const float fx = floorf(rx + (float)i * a11), fy = floorf(ry + (float)i * a21);
      outside = outside || !(fx >= 0.0f && fy >= 0.0f && fx < width && fy < height);
   }
   return outside;
}

#define HS_AFF_NT 23    // ceil(361 / 16)
// the parity kernel's own grouping (tuning: -DHS_AFFP_G=2 gives each keypoint 32 lanes: half the LDS per wavefront, twice the
// wavefronts per CU, but the serial sums and the double-precision tail then serve two keypoints instead of four)
#ifndef HS_AFFP_G
#define HS_AFFP_G 4
#endif
#define HS_AFFP_L (64 / HS_AFFP_G)
#define HS_AFFP_SH (HS_AFFP_G == 4 ? 4 : (HS_AFFP_G == 2 ? 5 : 6))
#define HS_AFFP_NT ((HS_SMM_PIX + HS_AFFP_L - 1) / HS_AFFP_L)
#ifndef HS_AFF_XCD
#define HS_AFF_XCD 1
#endif
#ifndef HS_AFF_BATCHES
#define HS_AFF_BATCHES 2
#endif
#define HS_AFF_ARR 364  // 361 rounded up to a multiple of 4 floats

struct AffKp { const float *blur; int rows, cols, pitch; float x, y, s, pd; };

template <class Fetch>
__device__ __forceinline__ void hs_affine_groups(uint32_t first, uint32_t n, const float *__restrict__ mask_g, const DConsts &k, AffineOut out,
                                                 Fetch fetch)
{
   __shared__ __attribute__((aligned(16))) float s_arr[HS_AFFP_G][3][HS_AFF_ARR];   // img, then a terms | b terms | c terms
   __shared__ float s_mask[HS_AFF_ARR];
   __shared__ float s_bc[HS_AFFP_G][8];
   const int lane = threadIdx.x, grp = lane >> HS_AFFP_SH, li = lane & (HS_AFFP_L - 1);
   for (int i = lane; i < HS_SMM_PIX; i += 64) s_mask[i] = mask_g[i];
   float *s_img = s_arr[grp][0], *s_pa = s_arr[grp][0], *s_pb = s_arr[grp][1], *s_pc = s_arr[grp][2];   // the a terms replace the image
   // XCD-aware order: block b runs on XCD b % 8 (observed dispatch order, a speed assumption only).  The keypoints are
   // ordered by image, octave, level and raster position, so neighbours in the list sample the same cache lines of the
   // same plane: each XCD takes one contiguous eighth of the list, its blocks stride inside that eighth, and those lines
   // are fetched into ONE private L2 instead of eight.  (Grids that are not a multiple of 8 blocks keep the plain stride.)
   uint32_t hstep = gridDim.x * HS_AFFP_G, h_end = n;
   uint32_t h = first + blockIdx.x * HS_AFFP_G + grp;
   if (HS_AFF_XCD && (gridDim.x & 7u) == 0u && n > first) {
      const uint32_t n_items = (n - first + HS_AFFP_G - 1) / HS_AFFP_G;   // groups of HS_AFFP_G keypoints
      const uint32_t xcd = blockIdx.x & 7u, rank = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
      const uint32_t it_lo = (uint32_t)(((unsigned long long)n_items * xcd) >> 3), it_hi = (uint32_t)(((unsigned long long)n_items * (xcd + 1)) >> 3);
      hstep = per_xcd * HS_AFFP_G;
      h_end = min(first + it_hi * HS_AFFP_G, n);
      h = first + (it_lo + rank) * HS_AFFP_G + grp;
   }
#define HS_AFF_END h_end
   if (k.maxIterations <= 0) {   // no iteration: U = identity, not converged
      for (; h < HS_AFF_END; h += hstep)
         if (li == 0) {
            out.converged[h] = 0; out.iters[h] = 0;
            out.U[4 * h + 0] = 1.0f; out.U[4 * h + 1] = 0.0f; out.U[4 * h + 2] = 0.0f; out.U[4 * h + 3] = 1.0f;
         }
      return;
   }
   // per-group state, replicated in the group's 16 lanes
   const float *blur = nullptr;
   int pitch = 0, width = 0, height = 0, l = 0;
   float lx = 0, ly = 0, ratio = 0, u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f;
   float eigen_ratio_act = 0.0f, eigen_ratio_bef = 0.0f;   // used by lane li == 0
   bool active = h < HS_AFF_END;
   auto load_kp = [&]() {
      if (active) {
         const AffKp q = fetch(h);
         blur = q.blur; pitch = q.pitch; width = q.cols - 1; height = q.rows - 1;
         lx = q.x / q.pd; ly = q.y / q.pd;
         ratio = q.s / (k.affInitialSigma * q.pd);
         u11 = 1.0f; u12 = 0.0f; u21 = 0.0f; u22 = 1.0f;
         eigen_ratio_act = 0.0f; eigen_ratio_bef = 0.0f;
         l = 0;
      }
   };
   load_kp();
   HS_WAVE_LDS_SYNC();
   while (__ballot(active) != 0ull) {
      bool win_in = true;
      if (active) {
         const float a11 = u11 * ratio, a12 = u12 * ratio, a21 = u21 * ratio, a22 = u22 * ratio;
         // interpolate(), helpers.cpp:209-244 (return value ignored at affine.cpp:47): 23 taps per lane in two batches.
         // The tap coordinates are monotone in i and in j (hs_window_outside): when the four corners of the 19 x 19 grid
         // are inside the level, every tap is, and the taps need no bounds test, no selects and only 32-bit offsets;
         // a window that touches the border (a few per cent) keeps the tested tap, which zeroes what lies outside.
         win_in = !hs_window_outside(height + 1, width + 1, lx, ly, a11, a12, a21, a22, HS_SMM >> 1);
      }
      // one decision per wavefront (a branch per group would split the batches of gathers): the untested taps when the
      // windows of all its active keypoints are inside, the tested ones for everybody otherwise
      const bool all_in = __ballot(active && !win_in) == 0ull;
      auto sample = [&](auto inside_c) {
         constexpr bool INSIDE = decltype(inside_c)::value;
         const float a11 = u11 * ratio, a12 = u12 * ratio, a21 = u21 * ratio, a22 = u22 * ratio;
#pragma unroll
         for (int half = 0; half < HS_AFF_BATCHES; half++) {
            constexpr int NB = (HS_AFFP_NT + HS_AFF_BATCHES - 1) / HS_AFF_BATCHES;
            float sv[NB];
#pragma unroll
            for (int t = 0; t < NB; t++) {
               const int idx = min(li + HS_AFFP_L * (half * NB + t), HS_SMM_PIX - 1);
               const int jj = idx / HS_SMM, ii = idx - jj * HS_SMM;
               const int j = jj - (HS_SMM >> 1), i = ii - (HS_SMM >> 1);
               const float rx = lx + (float)j * a12;
               const float ry = ly + (float)j * a22;
               const float wx = rx + (float)i * a11;
               const float wy = ry + (float)i * a21;
               if (INSIDE) {
                  sv[t] = hs_tap_inside_ptr(blur, pitch, wx, wy);
               } else {
                  bool outside = false;
                  sv[t] = hs_bilinear(blur, pitch, width, height, wx, wy, outside);
               }
            }
#pragma unroll
            for (int t = 0; t < NB; t++) HS_KEEP(sv[t]);
#pragma unroll
            for (int t = 0; t < NB; t++) {
               const int idx = li + HS_AFFP_L * (half * NB + t);
               if (idx < HS_SMM_PIX) s_img[idx] = sv[t];
            }
         }
      };
      if (active) {
         if (all_in) sample(std::true_type{});
         else sample(std::false_type{});
      }
      HS_WAVE_LDS_SYNC();
      if (active) {
         // computeGradient affine.cpp:14-33 + products affine.cpp:62-68.  The a terms take the place of
         // the sampled image in LDS, so they wait in registers until every lane has read its neighbours.
         float pa[HS_AFFP_NT];
#pragma unroll
         for (int t = 0; t < HS_AFFP_NT; t++) {
            const int idx = min(li + HS_AFFP_L * t, HS_SMM_PIX - 1);
            const int r = idx / HS_SMM, c = idx - r * HS_SMM;
            // hs_grad with clamped neighbour indices (one-sided differences at the tile border)
            const float gxx = s_img[idx + (c < HS_SMM - 1 ? 1 : 0)] - s_img[idx - (c > 0 ? 1 : 0)];
            const float gyy = s_img[idx + (r < HS_SMM - 1 ? HS_SMM : 0)] - s_img[idx - (r > 0 ? HS_SMM : 0)];
            const float v = s_mask[idx];
            const float gxy = gxx * gyy;
            pa[t] = gxx * gxx * v;
            if (li + HS_AFFP_L * t < HS_SMM_PIX) {
               s_pb[idx] = gxy * v;
               s_pc[idx] = gyy * gyy * v;
            }
         }
         HS_WAVE_LDS_SYNC();
#pragma unroll
         for (int t = 0; t < HS_AFFP_NT; t++) {
            const int idx = li + HS_AFFP_L * t;
            if (idx < HS_SMM_PIX) s_pa[idx] = pa[t];
         }
      }
      HS_WAVE_LDS_SYNC();
      if (active && li < 3) {
         // 361 terms in index order (affine.cpp:57-68); float4 LDS reads, the adds stay sequential
         const float *pp = s_arr[grp][li];
         const float4 *p4 = reinterpret_cast<const float4 *>(pp);
         float acc = 0.0f;
         for (int i0 = 0; i0 < HS_SMM_PIX / 4; i0 += 10) {   // 90 = 9 x 10 float4, ten reads in flight
            float4 q[10];
#pragma unroll
            for (int u = 0; u < 10; u++) q[u] = p4[i0 + u];
#pragma unroll
            for (int u = 0; u < 10; u++) { acc += q[u].x; acc += q[u].y; acc += q[u].z; acc += q[u].w; }
         }
         acc += pp[HS_SMM_PIX - 1];   // 361 = 4 * 90 + 1
         s_bc[grp][li] = acc / (float)HS_SMM_PIX;
      }
      HS_WAVE_LDS_SYNC();
      if (active && li == 0) {
         float a = s_bc[grp][0], b = s_bc[grp][1], c = s_bc[grp][2];
         float l1, l2;
         hs_inv_sqrt(a, b, c, l1, l2);
         eigen_ratio_bef = eigen_ratio_act;
         eigen_ratio_act = 1 - l2 / l1;
         const float u11t = u11, u12t = u12;
         const float n11 = a * u11t + b * u21, n12 = a * u12t + b * u22;
         const float n21 = b * u11t + c * u21, n22 = b * u12t + c * u22;
         int state = 0;   // 0 continue, 1 break (rejected), 2 converged
         if (!hs_eigenvalues(n11, n12, n21, n22, l1, l2)) state = 1;
         else if ((l1 / l2 > 6) || (l2 / l1 > 6)) state = 1;
         else if (eigen_ratio_act < k.convergenceThreshold && eigen_ratio_bef < k.convergenceThreshold) state = 2;
         s_bc[grp][3] = n11; s_bc[grp][4] = n12; s_bc[grp][5] = n21; s_bc[grp][6] = n22;
         s_bc[grp][7] = __int_as_float(state);
      }
      HS_WAVE_LDS_SYNC();
      if (active) {
         u11 = s_bc[grp][3]; u12 = s_bc[grp][4]; u21 = s_bc[grp][5]; u22 = s_bc[grp][6];
         const int state = __float_as_int(s_bc[grp][7]);
         if (state != 0 || l + 1 >= k.maxIterations) {
            // affine.cpp:88-99: converged -> onAffineShapeFound(..., l); otherwise the keypoint is dropped
            if (li == 0) {
               out.converged[h] = (state == 2) ? 1 : 0;
               out.iters[h] = (state == 2) ? l : 0;
               out.U[4 * h + 0] = u11; out.U[4 * h + 1] = u12; out.U[4 * h + 2] = u21; out.U[4 * h + 3] = u22;
            }
            h += hstep;
            active = h < HS_AFF_END;
            load_kp();
         } else {
            l++;
         }
      }
      HS_WAVE_LDS_SYNC();
   }
}


#ifndef HS_AFF_WAVES
#define HS_AFF_WAVES 0   // tuning: wavefronts per SIMD to hold the register allocation to (0: the compiler's choice)
#endif
__global__ __launch_bounds__(64, HS_AFF_WAVES) void k_affine(PlaneTab pt, HessList hl, uint32_t h_lo, uint32_t h_hi, const uint32_t *__restrict__ n_ptr,
                                               KpTables tb, DConsts k, AffineOut out)
{
   const uint32_t n = min(min(*n_ptr, hl.cap), h_hi);   // keypoints [h_lo, h_hi) of the list
   hs_affine_groups(
      h_lo, n, tb.smm_mask, k, out, [&](uint32_t h) {
      const int meta = hl.meta[h];
      const int b = meta >> 8, octave = (meta >> 4) & 15, level = (meta >> 2) & 3;
      const DPlane &P = pt.L[octave][level];
      AffKp q;
      q.blur = P.img(b); q.rows = P.rows; q.cols = P.cols; q.pitch = P.pitch;
      q.x = hl.x[h]; q.y = hl.y[h]; q.s = hl.s[h]; q.pd = k.pd0 * (float)(1 << octave);
      return q;
   });
}

// stage API flavour: n keypoints on a single plane, pixelDistance given explicitly
__global__ __launch_bounds__(64) void k_affine_stage(DPlane P, const float *__restrict__ kp /*n x 4*/, int n, KpTables tb, DConsts k,
                                                     AffineOut out)
{
   hs_affine_groups(0u, (uint32_t)n, tb.smm_mask, k, out, [&](uint32_t h) {
      AffKp q;
      q.blur = P.img(0); q.rows = P.rows; q.cols = P.cols; q.pitch = P.pitch;
      q.x = kp[4 * h]; q.y = kp[4 * h + 1]; q.s = kp[4 * h + 2]; q.pd = kp[4 * h + 3];
      return q;
   });
}

// ---------------------------------------------------------------------------------------
// k_prepare_patch: onAffineShapeFound hesaff.cpp:72-82 up to the border test of
// normalizeAffine affine.cpp:108-113: rectify (helpers.cpp:90-97), mrScale / P0 / scale,
// interpolateCheckBorders.  Thread per keypoint.  Survivors are binned by window size P so
// that each patch kernel launch has a uniform LDS footprint.
// ---------------------------------------------------------------------------------------
#define HS_NBINS 5   // window size P: 0: <=41, 1: <=64 (full blur in LDS); 2: <=128 (row-streamed, LDS); 3: <=512, 4: larger (row-streamed, HBM)
#define HS_BIN3_PMAX 512
__host__ __device__ inline int hs_patch_bin(int P) { return P <= 41 ? 0 : (P <= 64 ? 1 : (P <= 128 ? 2 : (P <= HS_BIN3_PMAX ? 3 : 4))); }

struct PatchWork {
   float *A;            // [n][4] rectified a11,a12,a21,a22
   int32_t *P0;         // 2*int(mrScale)+1 ; 0 = dead before the patch stage
   int32_t *alive;      // 1 while the keypoint is still a candidate for output
   uint32_t *bin_count; // [HS_NBINS]
   uint32_t *bin_work;  // [HS_NBINS] next unclaimed item of each bin's list (dynamic scheduling of the patch kernels)
   uint32_t *bin_items; // [HS_NBINS][cap]
   uint32_t cap;
};

// window side of normalizeAffine's smoothing branch from the scale alone (affine.cpp:106-109,120): 0 when P0 is out of range
__device__ __forceinline__ int hs_window_p0(float s, float mrSize)
{
   const float mrScale = ceilf(s * mrSize);
   // int(mrScale): saturate; the tap table bound rejects such a window anyway
   const int m = (mrScale < 1.0e6f) ? (int)mrScale : 1000000;
   return 2 * m + 1;
}

// Upper bound, per image, of the T' rows the huge windows (last bin) of its keypoints need: depends on the scales
// only, so it is known right after detection and the host can size / group the patch stage without waiting for the
// affine iteration.  rows[b] += P for every Hessian keypoint whose window falls into the last bin.
// rows[nimg + 1] (one past the per-image sums) receives the largest such P of the batch: the row kernel's LDS is sized for the
// windows that exist, not for the largest the image could hold.
__global__ __launch_bounds__(256) void k_image_large_rows(HessList hl, const uint32_t *__restrict__ n_ptr, float mrSize, uint32_t *__restrict__ rows, int nimg);

template <bool RECTIFY>
__device__ __forceinline__ void hs_prepare_patch_body(const HessList &hl, uint32_t h_lo, uint32_t n, const AffineOut &aff, int imRows, int imCols,
                                                      const DConsts &k, const KpTables &tb, const PatchWork &pw)
{
   for (uint32_t h0 = h_lo + blockIdx.x * blockDim.x; h0 < n; h0 += gridDim.x * blockDim.x) {
      const uint32_t h = h0 + threadIdx.x;
      const bool valid = h < n;
      int alive = 0, P0 = 0;
      if (valid && (!RECTIFY || aff.converged[h])) {
         float a11, a12, a21, a22;
         if (RECTIFY) {
            a11 = aff.U[4 * h]; a12 = aff.U[4 * h + 1]; a21 = aff.U[4 * h + 2]; a22 = aff.U[4 * h + 3];
            hs_rectify(a11, a12, a21, a22);
            pw.A[4 * h] = a11; pw.A[4 * h + 1] = a12; pw.A[4 * h + 2] = a21; pw.A[4 * h + 3] = a22;
         } else {
            a11 = pw.A[4 * h]; a12 = pw.A[4 * h + 1]; a21 = pw.A[4 * h + 2]; a22 = pw.A[4 * h + 3];
         }
         P0 = hs_window_p0(hl.s[h], k.mrSize);
         const float scale = (float)P0 / (float)HS_PATCH;
         bool rej = hs_check_borders(imRows, imCols, hl.x[h], hl.y[h], a11 * scale, a12 * scale, a21 * scale, a22 * scale);
         // smoothing branch (affine.cpp:114-135): a window that leaves the image rejects the keypoint
         if (!rej && (double)scale > 0.4) rej = hs_window_outside(imRows, imCols, hl.x[h], hl.y[h], a11, a12, a21, a22, (P0 + 2) >> 1);
         // P0 > max_p0: the P x P window cannot fit into the image (a11*a22 = 1), the window test has
         // rejected it; no taps are tabulated for it.
         alive = (!rej && P0 <= tb.max_p0) ? 1 : 0;
      }
      if (valid) {
         pw.alive[h] = alive;
         pw.P0[h] = alive ? P0 : 0;
      }
      int bin = -1;
      if (alive) {
         const float scale = (float)P0 / (float)HS_PATCH;
         const int P = ((double)scale > 0.4) ? P0 + 2 : 0;
         bin = hs_patch_bin(P);
      }
      // one atomic per (wave, bin) instead of one per keypoint
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int bq = 0; bq < HS_NBINS; bq++) {
         const unsigned long long m = __ballot(bin == bq);
         if (m == 0ull) continue;
         const int leader = __ffsll((long long)m) - 1;
         uint32_t base = 0;
         if (lane == leader) base = atomicAdd(pw.bin_count + bq, (uint32_t)__popcll(m));
         base = __shfl(base, leader, 64);
         if (bin == bq) pw.bin_items[(size_t)bq * pw.cap + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = h;
      }
   }
}

__global__ __launch_bounds__(256) void k_image_large_rows(HessList hl, const uint32_t *__restrict__ n_ptr, float mrSize, uint32_t *__restrict__ rows, int nimg)
{
   const uint32_t n = min(*n_ptr, hl.cap);
   for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) {
      const int P0 = hs_window_p0(hl.s[h], mrSize);
      const float scale = (float)P0 / (float)HS_PATCH;
      const int P = ((double)scale > 0.4) ? P0 + 2 : 0;
      if (hs_patch_bin(P) == HS_NBINS - 1 && P0 < (1 << 20)) {
         atomicAdd(rows + (hl.meta[h] >> 8), (uint32_t)P);
         atomicMax(rows + nimg + 1, (uint32_t)P);
      }
   }
}

__global__ __launch_bounds__(256) void k_prepare_patch(HessList hl, uint32_t h_lo, uint32_t h_hi, const uint32_t *__restrict__ n_ptr, AffineOut aff,
                                                       int imRows, int imCols, DConsts k, KpTables tb, PatchWork pw)
{
   hs_prepare_patch_body<true>(hl, h_lo, min(min(*n_ptr, hl.cap), h_hi), aff, imRows, imCols, k, tb, pw);   // keypoints [h_lo, h_hi) of the list
}

// stage API: the rectified matrices are already in pw.A (normalizeAffine's own arguments)
__global__ __launch_bounds__(256) void k_prepare_patch_given_A(HessList hl, const uint32_t *__restrict__ n_ptr, int imRows, int imCols,
                                                               DConsts k, KpTables tb, PatchWork pw)
{
   AffineOut none;
   none.converged = nullptr; none.U = nullptr; none.iters = nullptr;
   hs_prepare_patch_body<false>(hl, 0u, min(*n_ptr, hl.cap), none, imRows, imCols, k, tb, pw);
}

// ---------------------------------------------------------------------------------------
// k_pack: stable compaction of the described keypoints into hesaff_keypoint records
// (hesaff.cpp:87-91), rank from the exclusive scan of `alive`.
// ---------------------------------------------------------------------------------------
struct KeyRec {   // == hesaff_keypoint (include/hesaff_amd.h), 164 bytes
   float x, y, s, a11, a12, a21, a22, response;
   int32_t type;
   uint8_t desc[128];
};

// A block takes 64 consecutive Hessian keypoints: their head fields come in as coalesced runs of the nine arrays, their descriptors as one
// 8 KB run, the records are assembled in LDS in rank order - the ranks of a run's survivors are consecutive (stable compaction) - and leave as
// ONE contiguous run of dwords.  (Round 4's form - 32 lanes per keypoint, lane 0 fetching nine fields from nine lines and storing nine
// dwords - ran at 2.2 TB/s: 4.6 ms per 256 UHD images; profiles/r05_notes.md.)
#define HS_PACK_KP 64
#define HS_PACK_DW 41   // dwords per record
__global__ __launch_bounds__(256) void k_pack(HessList hl, const uint32_t *__restrict__ n_ptr, PatchWork pw,
                                              const uint32_t *__restrict__ rank, const uint8_t *__restrict__ desc, KeyRec *__restrict__ out)
{
   static_assert(sizeof(KeyRec) == 4 * HS_PACK_DW, "record layout");
   __shared__ uint32_t s_rec[HS_PACK_KP * HS_PACK_DW];
   __shared__ int s_lr[HS_PACK_KP];   // rank inside the block's run of records, -1: not described
   const uint32_t n = min(*n_ptr, hl.cap);
   const int tid = threadIdx.x;
   const uint32_t *desc32 = reinterpret_cast<const uint32_t *>(desc);
   uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
   for (uint32_t h0 = blockIdx.x * HS_PACK_KP; h0 < n; h0 += gridDim.x * HS_PACK_KP) {
      const uint32_t base = rank[h0];   // exclusive scan of `alive`: records of earlier keypoints
      if (tid < HS_PACK_KP) {
         const uint32_t h = h0 + tid;
         const bool live = h < n && pw.alive[h];
         const int lr = live ? (int)(rank[h] - base) : -1;
         s_lr[tid] = lr;
         if (live) {
            uint32_t *r = s_rec + lr * HS_PACK_DW;
            const float4 A = *reinterpret_cast<const float4 *>(pw.A + 4 * (size_t)h);
            r[0] = __float_as_uint(hl.x[h]); r[1] = __float_as_uint(hl.y[h]); r[2] = __float_as_uint(hl.s[h]);
            r[3] = __float_as_uint(A.x); r[4] = __float_as_uint(A.y); r[5] = __float_as_uint(A.z); r[6] = __float_as_uint(A.w);
            r[7] = __float_as_uint(hl.response[h]);
            r[8] = (uint32_t)(hl.meta[h] & 3);
         }
      }
      __syncthreads();
      int n_live = 0;
#pragma unroll
      for (int q = 0; q < HS_PACK_KP * 32 / 256; q++) {
         const int idx = tid + 256 * q, j = idx >> 5, d = idx & 31;
         const int lr = s_lr[j];
         if (lr >= 0) s_rec[lr * HS_PACK_DW + 9 + d] = desc32[(size_t)(h0 + j) * 32 + d];
      }
      // the run's length: the last described keypoint's rank + 1 (s_lr is non-decreasing over the described ones)
      for (int j = HS_PACK_KP - 1; j >= 0; j--)
         if (s_lr[j] >= 0) { n_live = s_lr[j] + 1; break; }
      __syncthreads();
      const uint32_t total = (uint32_t)n_live * HS_PACK_DW;
      uint32_t *o = out32 + (size_t)base * HS_PACK_DW;
      for (uint32_t i = tid; i < total; i += 256) o[i] = s_rec[i];
      __syncthreads();
   }
}

// device check of hmath.h against the host (stage API)
__global__ void k_math(int n, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ at, float *__restrict__ pw)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   at[i] = hm_atan2f(a[i], b[i]);
   pw[i] = hm_pow2f(a[i]);
}

__global__ void k_rectify_stage(int n, float *A)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   hs_rectify(A[4 * i], A[4 * i + 1], A[4 * i + 2], A[4 * i + 3]);
}
