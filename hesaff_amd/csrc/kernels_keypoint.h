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
      for (int idx = lane; idx < HS_SMM_PIX; idx += 64) {
         const int jj = idx / HS_SMM, ii = idx - jj * HS_SMM;
         const int j = jj - (HS_SMM >> 1), i = ii - (HS_SMM >> 1);
         const float rx = lx + (float)j * a12;
         const float ry = ly + (float)j * a22;
         const float wx = rx + (float)i * a11;
         const float wy = ry + (float)i * a21;
         bool outside = false;
         s_img[idx] = hs_bilinear(blur, pitch, width, height, wx, wy, outside);
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
         const float *p = lane == 0 ? s_pa : (lane == 1 ? s_pb : s_pc);
         float acc = 0.0f;
         for (int i = 0; i < HS_SMM_PIX; i++) acc += p[i];
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

__global__ __launch_bounds__(64) void k_affine(PlaneTab pt, HessList hl, const uint32_t *__restrict__ n_ptr, KpTables tb, DConsts k,
                                               AffineOut out)
{
   __shared__ float s_img[HS_SMM_PIX + 3], s_pa[HS_SMM_PIX + 3], s_pb[HS_SMM_PIX + 3], s_pc[HS_SMM_PIX + 3];
   __shared__ float s_bc[8];
   const uint32_t n = min(*n_ptr, hl.cap);
   for (uint32_t h = blockIdx.x; h < n; h += gridDim.x) {
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
   __shared__ float s_img[HS_SMM_PIX + 3], s_pa[HS_SMM_PIX + 3], s_pb[HS_SMM_PIX + 3], s_pc[HS_SMM_PIX + 3];
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
#define HS_NBINS 4   // 0: P<=41 (LDS), 1: P<=64 (LDS), 2: P<=128 (LDS), 3: larger (global scratch)
__host__ __device__ inline int hs_patch_bin(int P) { return P <= 41 ? 0 : (P <= 64 ? 1 : (P <= 128 ? 2 : 3)); }

struct PatchWork {
   float *A;            // [n][4] rectified a11,a12,a21,a22
   int32_t *P0;         // 2*int(mrScale)+1 ; 0 = dead before the patch stage
   int32_t *alive;      // 1 while the keypoint is still a candidate for output
   uint32_t *bin_count; // [HS_NBINS]
   uint32_t *bin_items; // [HS_NBINS][cap]
   uint32_t cap;
};

template <bool RECTIFY>
__device__ __forceinline__ void hs_prepare_patch_body(const HessList &hl, uint32_t n, const AffineOut &aff, int imRows, int imCols,
                                                      const DConsts &k, const KpTables &tb, const PatchWork &pw)
{
   for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) {
      int alive = 0, P0 = 0;
      if (!RECTIFY || aff.converged[h]) {
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
      pw.alive[h] = alive;
      pw.P0[h] = alive ? P0 : 0;
      if (alive) {
         const float scale = (float)P0 / (float)HS_PATCH;
         const int P = ((double)scale > 0.4) ? P0 + 2 : 0;
         const int bin = hs_patch_bin(P);
         const uint32_t slot = atomicAdd(pw.bin_count + bin, 1u);
         pw.bin_items[(size_t)bin * pw.cap + slot] = h;
      }
   }
}

__global__ __launch_bounds__(256) void k_prepare_patch(HessList hl, const uint32_t *__restrict__ n_ptr, AffineOut aff, int imRows,
                                                       int imCols, DConsts k, KpTables tb, PatchWork pw)
{
   hs_prepare_patch_body<true>(hl, min(*n_ptr, hl.cap), aff, imRows, imCols, k, tb, pw);
}

// stage API: the rectified matrices are already in pw.A (normalizeAffine's own arguments)
__global__ __launch_bounds__(256) void k_prepare_patch_given_A(HessList hl, const uint32_t *__restrict__ n_ptr, int imRows, int imCols,
                                                               DConsts k, KpTables tb, PatchWork pw)
{
   AffineOut none;
   none.converged = nullptr; none.U = nullptr; none.iters = nullptr;
   hs_prepare_patch_body<false>(hl, min(*n_ptr, hl.cap), none, imRows, imCols, k, tb, pw);
}

// ---------------------------------------------------------------------------------------
// SIFT on a 41x41 patch held in LDS: computeSiftDescriptor siftdesc.cpp:115-140.
//   s_patch[1681] in/out (photometrically normalised in place, helpers.cpp:246-281)
//   s_val[1681]   mask*grad, s_o[1681] orientation coordinate `o` (siftdesc.cpp:59,65)
//   s_vec[128], s_misc[8]
// Block of 256 threads; all threads must call it (contains __syncthreads()).
// ---------------------------------------------------------------------------------------
__device__ inline void hs_sift_block(float *s_patch, float *s_val, float *s_o, float *s_vec, float *s_misc,
                                     const KpTables &tb, const DConsts &k, uint8_t *__restrict__ desc_out)
{
   const int tid = threadIdx.x;
   // photometricallyNormalize: sequential mean / variance over the masked pixels
   if (tid == 0) {
      float sum = 0.0f, gsum = 0.0f;
      for (int i = 0; i < HS_PATCH_PIX; i++)
         if (tb.sift_mask[i] > 0) { sum += s_patch[i]; gsum++; }
      sum = sum / gsum;
      float var = 0.0f;
      for (int i = 0; i < HS_PATCH_PIX; i++)
         if (tb.sift_mask[i] > 0) var += (sum - s_patch[i]) * (sum - s_patch[i]);
      var = sqrtf(var / gsum);
      s_misc[0] = sum;
      s_misc[1] = var;
   }
   __syncthreads();
   {
      const float sum = s_misc[0], var = s_misc[1];
      if (!((double)var < 0.0001)) {
         const float fac = 50.0f / var;
         for (int i = tid; i < HS_PATCH_PIX; i += blockDim.x) {
            float v = 128 + fac * (s_patch[i] - sum);
            if (v > 255) v = 255;
            if (v < 0) v = 0;
            s_patch[i] = v;
         }
      }
   }
   __syncthreads();
   // gradient magnitude / orientation, siftdesc.cpp:123-137, and the per-pixel part of samplePatch
   for (int i = tid; i < HS_PATCH_PIX; i += blockDim.x) {
      const int r = i / HS_PATCH, c = i - r * HS_PATCH;
      float gx, gy;
      hs_grad(s_patch, HS_PATCH, r, c, gx, gy);
      const float grad = sqrtf(gx * gx + gy * gy);
      const float ori = hm_atan2f(gy, gx);
      s_val[i] = tb.sift_mask[i] * grad;
      // float(orientationBins) * (ori + 2*M_PI) / (2*M_PI), evaluated in double (M_PI)
      s_o[i] = (float)((double)8.0f * ((double)ori + 2 * 3.14159265358979323846) / (2 * 3.14159265358979323846));
   }
   if (tid < 128) s_vec[tid] = 0.0f;
   __syncthreads();
   // samplePatch siftdesc.cpp:51-81: thread t owns vec[t] = cell (br, bc, bo) and walks the
   // patch in raster order, adding exactly the terms the reference adds to that cell.
   if (tid < 128) {
      const int my_br = (tid >> 5) * 8, my_bc = ((tid >> 3) & 3) * 8, my_bo = tid & 7;
      float acc = 0.0f;
      // rows whose bins touch this cell: 8*br .. 8*br+15 (step = 1/8), clipped to the patch
      const int rlo = (tid >> 5) * 8, rhi = min(rlo + 16, HS_PATCH);
      const int clo = ((tid >> 3) & 3) * 8, chi = min(clo + 16, HS_PATCH);
      for (int r = rlo; r < rhi; ++r) {
         const int br0 = tb.bin0[r], br1 = tb.bin1[r];
         const float wr0 = tb.w0[r], wr1 = tb.w1[r];
         const bool m0 = (br0 == my_br), m1 = (br1 == my_br);
         if (!(m0 || m1)) continue;
         for (int c = clo; c < chi; ++c) {
            const int bc0 = tb.bin0[c], bc1 = tb.bin1[c];
            const bool n0 = (bc0 == my_bc), n1 = (bc1 == my_bc);
            if (!(n0 || n1)) continue;
            const float val = s_val[r * HS_PATCH + c];
            const float o = s_o[r * HS_PATCH + c];
            int bo0 = (int)o;
            const float wo1 = o - (float)bo0;
            bo0 &= 7;
            const int bo1 = (bo0 + 1) & 7;
            const float wo0 = 1.0f - wo1;
            float wo;
            if (bo0 == my_bo) wo = wo0;
            else if (bo1 == my_bo) wo = wo1;
            else continue;
            const float wc0 = tb.w0[c] * val, wc1 = tb.w1[c] * val;
            float v;
            if (m0 && n0) { v = wr0 * wc0; if (v > 0) acc += v * wo; }
            if (m0 && n1) { v = wr0 * wc1; if (v > 0) acc += v * wo; }
            if (m1 && n0) { v = wr1 * wc0; if (v > 0) acc += v * wo; }
            if (m1 && n1) { v = wr1 * wc1; if (v > 0) acc += v * wo; }
         }
      }
      s_vec[tid] = acc;
   }
   __syncthreads();
   // sample() siftdesc.cpp:98-113: normalize, clip, renormalize, quantise
   for (int pass = 0; pass < 2; pass++) {
      if (tid == 0) {
         float vectlen = 0.0f;
         for (int i = 0; i < 128; i++) { const float v = s_vec[i]; vectlen += v * v; }
         vectlen = sqrtf(vectlen);
         s_misc[2] = 1.0f / vectlen;
         s_misc[3] = 0.0f;
      }
      __syncthreads();
      if (tid < 128) {
         float v = s_vec[tid] * s_misc[2];
         if (pass == 0 && v > k.maxBinValue) { v = k.maxBinValue; s_misc[3] = 1.0f; }
         s_vec[tid] = v;
      }
      __syncthreads();
      const bool changed = s_misc[3] != 0.0f;
      __syncthreads();
      if (!changed) break;
   }
   if (tid < 128) {
      const float q = 512.0f * s_vec[tid];
      int bq = (q == q) ? (int)q : 0;   // NaN -> 0 (x86: INT_MIN, then the uchar cast gives 0)
      bq = min(bq, 255);
      desc_out[tid] = (uint8_t)bq;
   }
   __syncthreads();
}

// stand-alone SIFT over patches in global memory (stage API)
__global__ __launch_bounds__(256) void k_sift_stage(const float *__restrict__ patches, int n, KpTables tb, DConsts k,
                                                    uint8_t *__restrict__ desc)
{
   __shared__ float s_patch[HS_PATCH_PIX], s_val[HS_PATCH_PIX], s_o[HS_PATCH_PIX], s_vec[128], s_misc[8];
   for (int h = blockIdx.x; h < n; h += gridDim.x) {
      for (int i = threadIdx.x; i < HS_PATCH_PIX; i += blockDim.x) s_patch[i] = patches[(size_t)h * HS_PATCH_PIX + i];
      __syncthreads();
      hs_sift_block(s_patch, s_val, s_o, s_vec, s_misc, tb, k, desc + (size_t)h * 128);
   }
}

// ---------------------------------------------------------------------------------------
// k_patch_sift<BIN>: normalizeAffine affine.cpp:114-144 + computeSiftDescriptor, one
// 256-thread block per keypoint, grid-stride over the bin's work list.
//   smoothing branch (imageToPatchScale > 0.4):
//     1. warp the ORIGINAL image with the det-1 matrix A into a P x P window   (affine.cpp:126)
//     2. gaussianBlurInplace(window, 1.5*scale)   (affine.cpp:129, pinned order, K from P0)
//     3. resample to 41x41 with step `scale` around (P>>1, P>>1)             (affine.cpp:131)
//   direct branch: one warp with A*scale (affine.cpp:137-141).
// BIN 0..2 keep the window (S) and the row-pass plane (T) in LDS; BIN 3 uses a per-block
// global scratch slot and only evaluates the blur where step 3 reads it (same values).
// ---------------------------------------------------------------------------------------
struct PatchIO {
   DPlane image;        // original float image batch
   float *patches;      // optional [n][1681] output (stage API / debugging), may be null
   uint8_t *desc;       // [n][128]
   float *scratch;      // BIN 3: gridDim.x slots of scratch_stride floats
   long long scratch_stride;
};

template <int BIN>
__global__ __launch_bounds__(256) void k_patch_sift(HessList hl, PatchWork pw, PatchIO io, KpTables tb, DConsts k, int do_sift)
{
   extern __shared__ __attribute__((aligned(16))) float smem[];
   constexpr int PMAX = BIN == 0 ? 41 : (BIN == 1 ? 64 : (BIN == 2 ? 128 : 0));
   constexpr int WIN = PMAX * PMAX;                     // floats for S and for T
   constexpr int REGION = (2 * WIN > 2 * HS_PATCH_PIX) ? 2 * WIN : 2 * HS_PATCH_PIX;
   float *S = smem;                 // P x P warped window   | later s_val
   float *T = smem + WIN;           // P x P row pass        | later s_o (REGION >= 2*1681)
   float *s_patch = smem + REGION;  // 1681
   float *s_vec = s_patch + HS_PATCH_PIX;   // 128
   float *s_misc = s_vec + 128;             // 8
   float *s_taps = s_misc + 8;              // up to 31 taps for BIN<3 (K <= 0.22*128+1)
   __shared__ int s_flag;

   const int tid = threadIdx.x;
   const uint32_t cnt = min(pw.bin_count[BIN], pw.cap);
   const int imRows = io.image.rows, imCols = io.image.cols, imPitch = io.image.pitch;
   const int width = imCols - 1, height = imRows - 1;

   for (uint32_t wi = blockIdx.x; wi < cnt; wi += gridDim.x) {
      const uint32_t h = pw.bin_items[(size_t)BIN * pw.cap + wi];
      const int b = hl.meta[h] >> 8;
      const float *img = io.image.img(b);
      const float x = hl.x[h], y = hl.y[h];
      const float a11 = pw.A[4 * h], a12 = pw.A[4 * h + 1], a21 = pw.A[4 * h + 2], a22 = pw.A[4 * h + 3];
      const int P0 = pw.P0[h];
      const float scale = (float)P0 / (float)HS_PATCH;
      if (tid == 0) s_flag = 0;
      __syncthreads();
      bool rejected = false;
      if (!((double)scale > 0.4)) {
         // direct branch
         const float b11 = a11 * scale, b12 = a12 * scale, b21 = a21 * scale, b22 = a22 * scale;
         for (int idx = tid; idx < HS_PATCH_PIX; idx += 256) {
            const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
            const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
            const float rx = x + (float)j * b12, ry = y + (float)j * b22;
            const float wx = rx + (float)i * b11, wy = ry + (float)i * b21;
            bool outside = false;
            s_patch[idx] = hs_bilinear(img, imPitch, width, height, wx, wy, outside);
         }
         __syncthreads();
      } else {
         const int P = P0 + 2, half = P >> 1;
         const int K = tb.patch_tap_k[(P0 - 1) >> 1];
         const float *taps_g = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
         const int r = K >> 1;
         float *Sg = S, *Tg = T;
         if (BIN == 3) {
            Sg = io.scratch + (long long)blockIdx.x * io.scratch_stride;
            Tg = Sg + (long long)P * P;
         } else {
            for (int i = tid; i < K; i += 256) s_taps[i] = taps_g[i];
         }
         const float *taps = (BIN == 3) ? taps_g : s_taps;
         // 1. warp, affine.cpp:126 ; touching the image boundary rejects the keypoint
         bool outside = false;
         for (int idx = tid; idx < P * P; idx += 256) {
            const int jj = idx / P, ii = idx - jj * P;
            const int j = jj - half, i = ii - half;
            const float rx = x + (float)j * a12, ry = y + (float)j * a22;
            const float wx = rx + (float)i * a11, wy = ry + (float)i * a21;
            Sg[idx] = hs_bilinear(img, imPitch, width, height, wx, wy, outside);
         }
         if (outside) s_flag = 1;
         __syncthreads();
         rejected = s_flag != 0;
         if (!rejected) {
            const int pm = P - 1;
            if (BIN != 3) {
               // 2a. row pass (RowFilter order; SymmRowSmallFilter order for K <= 5)
               for (int idx = tid; idx < P * P; idx += 256) {
                  const int yy = idx / P, xx = idx - yy * P;
                  const float *Srow = Sg + yy * P;
                  float t;
                  if (K <= 5) {
                     t = Srow[xx] * taps[r] + (Srow[max(xx - 1, 0)] + Srow[min(xx + 1, pm)]) * taps[r + 1];
                     if (K == 5) t = t + (Srow[max(xx - 2, 0)] + Srow[min(xx + 2, pm)]) * taps[r + 2];
                  } else {
                     t = taps[0] * Srow[min(max(xx - r, 0), pm)];
                     for (int j = 1; j < K; j++) t += taps[j] * Srow[min(max(xx - r + j, 0), pm)];
                  }
                  Tg[idx] = t;
               }
               __syncthreads();
               // 2b. column pass (SymmColumnFilter order), result back into S
               for (int idx = tid; idx < P * P; idx += 256) {
                  const int yy = idx / P, xx = idx - yy * P;
                  float d = taps[r] * Tg[idx];
                  for (int j = 1; j <= r; j++) d += taps[r + j] * (Tg[min(yy + j, pm) * P + xx] + Tg[max(yy - j, 0) * P + xx]);
                  Sg[idx] = d;
               }
               __syncthreads();
               // 3. resample, affine.cpp:131 : interpolate(smoothed, P>>1, P>>1, scale,0,0,scale)
               for (int idx = tid; idx < HS_PATCH_PIX; idx += 256) {
                  const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
                  const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
                  const float c0 = (float)half;
                  const float rx = c0 + (float)j * 0.0f, ry = c0 + (float)j * scale;
                  const float wx = rx + (float)i * scale, wy = ry + (float)i * 0.0f;
                  bool o2 = false;
                  s_patch[idx] = hs_bilinear(Sg, P, P - 1, P - 1, wx, wy, o2);
               }
               __syncthreads();
            } else {
               // BIN 3 (large windows, global scratch): the resample reads the blurred window
               // only at <= 82 columns and <= 82 rows; evaluate the blur there and nowhere
               // else.  Column list / row list = floor(c0 + i*scale) and +1, i = -20..20.
               // Tg is reused as: rowpass[P][82] then blurred[82][82].
               const float c0 = (float)half;
               // 2a. row pass for every row, needed columns only
               for (int idx = tid; idx < P * 82; idx += 256) {
                  const int yy = idx / 82, q = idx - yy * 82;
                  const float wq = c0 + (float)((q >> 1) - 20) * scale;
                  const int xx = (int)floorf(wq) + (q & 1);
                  const float *Srow = Sg + (long long)yy * P;
                  float t = taps[0] * Srow[min(max(xx - r, 0), pm)];
                  for (int j = 1; j < K; j++) t += taps[j] * Srow[min(max(xx - r + j, 0), pm)];
                  Tg[idx] = t;
               }
               __syncthreads();
               // 2b. column pass at the 82 x 82 needed positions -> D (stored after rowpass)
               float *Dg = Tg + (long long)P * 82;
               for (int idx = tid; idx < 82 * 82; idx += 256) {
                  const int p = idx / 82, q = idx - p * 82;
                  const float wp = c0 + (float)((p >> 1) - 20) * scale;
                  const int yy = (int)floorf(wp) + (p & 1);
                  float d = taps[r] * Tg[(long long)min(max(yy, 0), pm) * 82 + q];
                  for (int j = 1; j <= r; j++)
                     d += taps[r + j] * (Tg[(long long)min(max(yy + j, 0), pm) * 82 + q] + Tg[(long long)min(max(yy - j, 0), pm) * 82 + q]);
                  Dg[idx] = d;
               }
               __syncthreads();
               // 3. resample from the sparse blurred samples (same four taps, same weights)
               for (int idx = tid; idx < HS_PATCH_PIX; idx += 256) {
                  const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
                  const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
                  const float rx = c0 + (float)j * 0.0f, ry = c0 + (float)j * scale;
                  float wx = rx + (float)i * scale, wy = ry + (float)i * 0.0f;
                  const float fx = floorf(wx), fy = floorf(wy);
                  wx -= fx; wy -= fy;
                  const int q = 2 * ii, p = 2 * jj;
                  const float p00 = Dg[p * 82 + q], p01 = Dg[p * 82 + q + 1], p10 = Dg[(p + 1) * 82 + q], p11 = Dg[(p + 1) * 82 + q + 1];
                  s_patch[idx] = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
               }
               __syncthreads();
            }
         }
      }
      if (rejected) {
         if (tid == 0) pw.alive[h] = 0;
         __syncthreads();
         continue;
      }
      if (io.patches)
         for (int i = tid; i < HS_PATCH_PIX; i += 256) io.patches[(size_t)h * HS_PATCH_PIX + i] = s_patch[i];
      if (do_sift) hs_sift_block(s_patch, smem, smem + HS_PATCH_PIX, s_vec, s_misc, tb, k, io.desc + (size_t)h * 128);
      __syncthreads();
   }
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
