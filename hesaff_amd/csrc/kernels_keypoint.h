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
   const int32_t *bin0, *bin1;   // precomputeBinsAndWeights siftdesc.cpp:18 (already x8)
   const float *w0, *w1;
   const float *patch_taps;      // Gaussian taps of every odd P0, concatenated
   const int32_t *patch_tap_off; // offset of P0's taps at index (P0-1)/2
   const int32_t *patch_tap_k;   // K of P0
   int max_p0;
};

// ---------------------------------------------------------------------------------------
// k_affine: AffineShape::findAffineShape affine.cpp:35-100, one wavefront (one 64-thread
// block) per Hessian keypoint.
//  per iteration: 361 bilinear taps spread over the 64 lanes -> LDS img; gradients and the
//  three products per pixel -> LDS; lanes 0,1,2 add the 361 terms of a,b,c in index order;
//  lane 0 runs the double-precision invSqrt and the U update; result broadcast through LDS.
// ---------------------------------------------------------------------------------------
struct AffineOut {
   int32_t *converged;   // 1 = onAffineShapeFound was called
   float *U;             // [n][4] a11,a12,a21,a22
   int32_t *iters;
};

__device__ __forceinline__ void hs_affine_one(const float *__restrict__ blur, int rows, int cols, int pitch, float x, float y,
                                              float s, float pd, const float *__restrict__ mask, const DConsts &k,
                                              float *s_img, float *s_pa, float *s_pb, float *s_pc, float *s_bc,
                                              int &conv_out, float *U_out, int &iters_out)
{
   const int lane = threadIdx.x;
   float eigen_ratio_act = 0.0f, eigen_ratio_bef = 0.0f;
   float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f;
   const float lx = x / pd, ly = y / pd;
   const float ratio = s / (k.affInitialSigma * pd);
   const int width = cols - 1, height = rows - 1;
   int converged = 0, iters = 0;
   for (int l = 0; l < k.maxIterations; l++) {
      const float a11 = u11 * ratio, a12 = u12 * ratio, a21 = u21 * ratio, a22 = u22 * ratio;
      // interpolate(), helpers.cpp:209-244 (return value ignored at affine.cpp:47)
      // the 6 taps of a lane are gathered together (clamped index, branch-free tap), then stored
      {
         constexpr int NT = (HS_SMM_PIX + 63) / 64;
         float sv[NT];
#pragma unroll
         for (int it = 0; it < NT; it++) {
            const int idx = min(lane + 64 * it, HS_SMM_PIX - 1);
            const int jj = idx / HS_SMM, ii = idx - jj * HS_SMM;
            const int j = jj - (HS_SMM >> 1), i = ii - (HS_SMM >> 1);
            const float rx = lx + (float)j * a12;
            const float ry = ly + (float)j * a22;
            const float wx = rx + (float)i * a11;
            const float wy = ry + (float)i * a21;
            bool outside = false;
            sv[it] = hs_bilinear(blur, pitch, width, height, wx, wy, outside);
         }
#pragma unroll
         for (int it = 0; it < NT; it++) HS_KEEP(sv[it]);
#pragma unroll
         for (int it = 0; it < NT; it++) {
            const int idx = lane + 64 * it;
            if (idx < HS_SMM_PIX) s_img[idx] = sv[it];
         }
      }
      __syncthreads();
      // computeGradient affine.cpp:14-33 + products affine.cpp:62-68
      for (int idx = lane; idx < HS_SMM_PIX; idx += 64) {
         const int r = idx / HS_SMM, c = idx - r * HS_SMM;
         float gxx, gyy;
         hs_grad(s_img, HS_SMM, r, c, gxx, gyy);
         const float v = mask[idx];
         const float gxy = gxx * gyy;
         s_pa[idx] = gxx * gxx * v;
         s_pb[idx] = gxy * v;
         s_pc[idx] = gyy * gyy * v;
      }
      __syncthreads();
      if (lane < 3) {
         // 361 terms in index order (affine.cpp:57-68); float4 LDS reads, the adds stay sequential
         const float4 *p4 = reinterpret_cast<const float4 *>(lane == 0 ? s_pa : (lane == 1 ? s_pb : s_pc));
         float acc = 0.0f;
#pragma unroll 2
         for (int i = 0; i < HS_SMM_PIX / 4; i++) {
            const float4 q = p4[i];
            acc += q.x; acc += q.y; acc += q.z; acc += q.w;
         }
         acc += (lane == 0 ? s_pa : (lane == 1 ? s_pb : s_pc))[HS_SMM_PIX - 1];   // 361 = 4 * 90 + 1
         s_bc[lane] = acc / (float)HS_SMM_PIX;
      }
      __syncthreads();
      if (lane == 0) {
         float a = s_bc[0], b = s_bc[1], c = s_bc[2];
         hs_inv_sqrt(a, b, c, l1, l2);
         eigen_ratio_bef = eigen_ratio_act;
         eigen_ratio_act = 1 - l2 / l1;
         const float u11t = u11, u12t = u12;
         u11 = a * u11t + b * u21; u12 = a * u12t + b * u22;
         u21 = b * u11t + c * u21; u22 = b * u12t + c * u22;
         int state = 0;   // 0 continue, 1 break (rejected), 2 converged
         if (!hs_eigenvalues(u11, u12, u21, u22, l1, l2)) state = 1;
         else if ((l1 / l2 > 6) || (l2 / l1 > 6)) state = 1;
         else if (eigen_ratio_act < k.convergenceThreshold && eigen_ratio_bef < k.convergenceThreshold) state = 2;
         s_bc[3] = u11; s_bc[4] = u12; s_bc[5] = u21; s_bc[6] = u22;
         s_bc[7] = __int_as_float(state);
      }
      __syncthreads();
      u11 = s_bc[3]; u12 = s_bc[4]; u21 = s_bc[5]; u22 = s_bc[6];
      const int state = __float_as_int(s_bc[7]);
      __syncthreads();
      if (state == 2) { converged = 1; iters = l; break; }
      if (state == 1) break;
   }
   conv_out = converged;
   iters_out = iters;
   U_out[0] = u11; U_out[1] = u12; U_out[2] = u21; U_out[3] = u22;
}

__global__ __launch_bounds__(64) void k_affine(PlaneTab pt, HessList hl, uint32_t h_lo, uint32_t h_hi, const uint32_t *__restrict__ n_ptr,
                                               KpTables tb, DConsts k, AffineOut out)
{
   __shared__ __attribute__((aligned(16))) float s_img[HS_SMM_PIX + 3], s_pa[HS_SMM_PIX + 3], s_pb[HS_SMM_PIX + 3], s_pc[HS_SMM_PIX + 3];
   __shared__ float s_bc[8];
   const uint32_t n = min(min(*n_ptr, hl.cap), h_hi);   // keypoints [h_lo, h_hi) of the list
   for (uint32_t h = h_lo + blockIdx.x; h < n; h += gridDim.x) {
      const int meta = hl.meta[h];
      const int b = meta >> 8, octave = (meta >> 4) & 15, level = (meta >> 2) & 3;
      const DPlane &P = pt.L[octave][level];
      const float pd = (float)(1 << octave);
      int conv, iters;
      float U[4];
      hs_affine_one(P.img(b), P.rows, P.cols, P.pitch, hl.x[h], hl.y[h], hl.s[h], pd, tb.smm_mask, k, s_img, s_pa, s_pb, s_pc,
                    s_bc, conv, U, iters);
      if (threadIdx.x == 0) {
         out.converged[h] = conv;
         out.iters[h] = iters;
         out.U[4 * h + 0] = U[0]; out.U[4 * h + 1] = U[1]; out.U[4 * h + 2] = U[2]; out.U[4 * h + 3] = U[3];
      }
   }
}

// stage API flavour: n keypoints on a single plane, pixelDistance given explicitly
__global__ __launch_bounds__(64) void k_affine_stage(DPlane P, const float *__restrict__ kp /*n x 4*/, int n, KpTables tb, DConsts k,
                                                     AffineOut out)
{
   __shared__ __attribute__((aligned(16))) float s_img[HS_SMM_PIX + 3], s_pa[HS_SMM_PIX + 3], s_pb[HS_SMM_PIX + 3], s_pc[HS_SMM_PIX + 3];
   __shared__ float s_bc[8];
   for (int h = blockIdx.x; h < n; h += gridDim.x) {
      int conv, iters;
      float U[4];
      hs_affine_one(P.img(0), P.rows, P.cols, P.pitch, kp[4 * h], kp[4 * h + 1], kp[4 * h + 2], kp[4 * h + 3], tb.smm_mask, k,
                    s_img, s_pa, s_pb, s_pc, s_bc, conv, U, iters);
      if (threadIdx.x == 0) {
         out.converged[h] = conv;
         out.iters[h] = iters;
         out.U[4 * h + 0] = U[0]; out.U[4 * h + 1] = U[1]; out.U[4 * h + 2] = U[2]; out.U[4 * h + 3] = U[3];
      }
   }
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
   uint32_t *bin_items; // [HS_NBINS][cap]
   uint32_t cap;
};

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
         const float s = hl.s[h];
         const float mrScale = ceilf(s * k.mrSize);
         // int(mrScale): saturate like the table bound below would reject anyway
         const int m = (mrScale < 1.0e6f) ? (int)mrScale : 1000000;
         P0 = 2 * m + 1;
         const float scale = (float)P0 / (float)HS_PATCH;
         const bool rej = hs_check_borders(imRows, imCols, hl.x[h], hl.y[h], a11 * scale, a12 * scale, a21 * scale, a22 * scale);
         // P0 > max_p0: the P x P window cannot fit into the image (a11*a22 = 1), the
         // reference rejects it in interpolate(); no taps are tabulated for it.
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

__global__ __launch_bounds__(256) void k_prepare_patch(HessList hl, uint32_t h_lo, const uint32_t *__restrict__ n_ptr, AffineOut aff, int imRows,
                                                       int imCols, DConsts k, KpTables tb, PatchWork pw)
{
   hs_prepare_patch_body<true>(hl, h_lo, min(*n_ptr, hl.cap), aff, imRows, imCols, k, tb, pw);
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

__global__ __launch_bounds__(256) void k_pack(HessList hl, const uint32_t *__restrict__ n_ptr, PatchWork pw,
                                              const uint32_t *__restrict__ rank, const uint8_t *__restrict__ desc, KeyRec *__restrict__ out)
{
   const uint32_t n = min(*n_ptr, hl.cap);
   // 32 threads per keypoint: lane 0 writes the 36-byte head, all write 4 descriptor bytes
   const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, lane = threadIdx.x & 31;
   const uint32_t stride = (gridDim.x * blockDim.x) >> 5;
   for (uint32_t h = g; h < n; h += stride) {
      if (!pw.alive[h]) continue;
      KeyRec *o = out + rank[h];
      if (lane == 0) {
         o->x = hl.x[h]; o->y = hl.y[h]; o->s = hl.s[h];
         o->a11 = pw.A[4 * h]; o->a12 = pw.A[4 * h + 1]; o->a21 = pw.A[4 * h + 2]; o->a22 = pw.A[4 * h + 3];
         o->response = hl.response[h];
         o->type = hl.meta[h] & 3;
      }
      const uint32_t v = *(const uint32_t *)(desc + (size_t)h * 128 + 4 * lane);
      *(uint32_t *)(o->desc + 4 * lane) = v;
   }
}

// zero flags[i] for i in [*n_ptr, cap): slots the current batch did not touch
__global__ __launch_bounds__(256) void k_clear_tail(int32_t *__restrict__ flags, const uint32_t *__restrict__ n_ptr, uint32_t cap)
{
   const uint32_t n = min(*n_ptr, cap);
   for (uint32_t i = n + blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) flags[i] = 0;
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
