// kernels_patch.h -- affine patch normalisation (AffineShape::normalizeAffine,
// affine.cpp:102-144), one 256-thread block per keypoint.  Keypoints are binned by the side P of
// the warped window (hs_patch_bin); the default path only EXTRACTS the 41x41 patch (to HBM, the
// descriptor runs in kernels_sift.h):
//   k_patch_extract_small<0|1>  P <= 41 | 64 : window S and row-pass plane T in LDS, stored with
//                                      replicated borders; tap count as template parameter
//   k_patch_mid<128|512>        P <= 128 | 512 : row-streamed; only the 82 blurred columns / rows
//                                      the 41x41 resample reads are evaluated, T' (P x 82, padded)
//                                      in a per-block HBM slot
//   k_patch_large_rows + k_patch_large_finish  P > 512 : one wavefront per chunk of window ROWS
//                                      writes T' rows to HBM (all rows of all huge keypoints run
//                                      in parallel), then one block per keypoint finishes.
// The earlier fused form (descriptor inside the patch kernel: k_patch_small<BIN, true>,
// k_patch_mid<.., true>, hs_sift_block) is kept behind HESAFF_SIFT=fused / HESAFF_SMALL=old and
// cross-checked against the default path by the GPU tests.
// Skipping blur outputs nobody reads does not change any value that is read: every
// evaluated tap sum uses the pinned cv::GaussianBlur order:
//   row   : t = k[0]*S[x-r]; t += k[j]*S[x-r+j]  (j ascending)       K > 5
//           S0*k0 + (S-1+S1)*k1 + (S-2+S2)*k2                         K <= 5
//   column: d = k[r]*T[y];  d += k[r+j]*(T[y+j] + T[y-j])
#pragma once
#include "kernels_keypoint.h"

#define HS_SIFT_ARR 1684   // 1681 rounded up to a multiple of 4 floats
#define HS_SIFT_TAB 232    // 4 x 41 bin/weight entries + 4 x 16 cell weights
#define HS_NEED 82         // blurred columns (and rows) the 41x41 resample reads: 2 per output

struct PatchIO {
   DPlane image;         // original float image batch (normalizeAffine samples the ORIGINAL image, hesaff.cpp:82)
   float *patches;       // optional [n][1681] output, row index h - h_base; may be null
   uint32_t h_base;
   uint8_t *desc;        // [n][128]
   float *trows;         // large bin: T' rows, [rows][82]
   const uint32_t *row_prefix;   // large bin: exclusive prefix of P over the bin's items (+ total)
   uint32_t item0, item1;        // large bin: item range of this round
   int bin;                      // large bin index (3 or 4)
};

__device__ __forceinline__ float hs_serial_sum(const float *__restrict__ v, int n)
{
   float acc = 0.0f;
   const float4 *v4 = reinterpret_cast<const float4 *>(v);
   const int n4 = n >> 2;
   for (int i = 0; i < n4; i++) {
      const float4 q = v4[i];
      acc += q.x; acc += q.y; acc += q.z; acc += q.w;
   }
   for (int i = n4 << 2; i < n; i++) acc += v[i];
   return acc;
}

// ---------------------------------------------------------------------------------------
// SIFT on a 41x41 patch held in LDS: computeSiftDescriptor siftdesc.cpp:115-140.
//   s_patch[1681] in/out (photometrically normalised in place, helpers.cpp:246-281)
//   s_va[2*1684]  16-byte aligned scratch: serial-sum operands, then (mask*grad, o) pairs
//   s_vec[128]    16-byte aligned, s_misc[8], s_tab[HS_SIFT_TAB]
// Block of 256 threads; all threads must call it (contains __syncthreads()).
// Sequential float sums stay sequential (one thread adds, in the reference's order); all
// the work around them is parallel and branch-free:
//  * masked pixels are gathered into a contiguous array, the serial thread only runs the
//    dependent chain of additions over float4 LDS reads;
//  * histogram: one wavefront; lane = (spatial cell, orientation pair {q, q+4}) walks the
//    cell's 16x16 pixel support in raster order.  Each pixel adds at most one term to a
//    histogram bin (siftdesc.cpp:75-78); where the reference adds nothing this adds 0.0f.
// Everything that does not depend on the keypoint (bin tables, cell weights, this thread's
// slice of the circular mask and of the masked-pixel index list) is loaded ONCE per block
// by hs_sift_setup and kept in LDS / registers across the block's keypoint loop.
// flags: profiling ablations only (HESAFF_ABLATE), 0 on every product path.
// ---------------------------------------------------------------------------------------
#define HS_SIFT_PIX_IT 7   // ceil(1681 / 256)
#define HS_SIFT_MSK_IT 5   // ceil(1245 / 256)

struct SiftRegs {
   float mask[HS_SIFT_PIX_IT];   // sift_mask[tid + 256 k]
   int midx[HS_SIFT_MSK_IT];     // mask_idx[tid + 256 k] (0 beyond n_masked)
};

__device__ __forceinline__ void hs_sift_setup(const KpTables &tb, float *s_tab, SiftRegs &rg)
{
   const int tid = threadIdx.x;
   int *s_bin0 = reinterpret_cast<int *>(s_tab), *s_bin1 = s_bin0 + HS_PATCH;
   float *s_w0 = s_tab + 2 * HS_PATCH, *s_w1 = s_tab + 3 * HS_PATCH, *s_cw = s_tab + 4 * HS_PATCH + 4;
   if (tid < HS_PATCH) {
      s_bin0[tid] = tb.bin0[tid]; s_bin1[tid] = tb.bin1[tid];
      s_w0[tid] = tb.w0[tid]; s_w1[tid] = tb.w1[tid];
   }
#pragma unroll
   for (int k = 0; k < HS_SIFT_PIX_IT; k++) { const int i = tid + 256 * k; rg.mask[k] = (i < HS_PATCH_PIX) ? tb.sift_mask[i] : 0.0f; }
#pragma unroll
   for (int k = 0; k < HS_SIFT_MSK_IT; k++) { const int i = tid + 256 * k; rg.midx[k] = (i < tb.n_masked) ? tb.mask_idx[i] : 0; }
   __syncthreads();
   if (tid < 64) {
      // cell weights: spatial bin b gets weight w1[r] from rows with bin1 == b and w0[r] from
      // rows with bin0 == b (siftdesc.cpp:55-56,61-62); clamped bins carry weight 0.
      const int b = tid >> 4, i = tid & 15, r = 8 * b + i;
      float w = 0.0f;
      if (r < HS_PATCH) {
         if (s_bin0[r] == 8 * b && s_w0[r] != 0.0f) w = s_w0[r];
         else if (s_bin1[r] == 8 * b) w = s_w1[r];
      }
      s_cw[tid] = w;
   }
   __syncthreads();
}

__device__ inline void hs_sift_block(float *s_patch, float *s_va, float *s_vec, float *s_misc, const float *s_tab, const SiftRegs &rg,
                                     const KpTables &tb, const DConsts &k, uint8_t *__restrict__ desc_out, int flags = 0)
{
   const int tid = threadIdx.x;
   const int nm = (flags & 4) ? 8 : tb.n_masked;
   const float *s_cw = s_tab + 4 * HS_PATCH + 4;
   // photometricallyNormalize helpers.cpp:253-260: mean over the pixels with mask > 0,
   // raster order.  gsum counts them in float: exact, == (float)nm.
#pragma unroll
   for (int q = 0; q < HS_SIFT_MSK_IT; q++) { const int i = tid + 256 * q; if (i < nm) s_va[i] = s_patch[rg.midx[q]]; }
   __syncthreads();
   if (tid == 0) s_misc[0] = hs_serial_sum(s_va, nm) / (float)nm;
   __syncthreads();
   {
      const float sum = s_misc[0];
#pragma unroll
      for (int q = 0; q < HS_SIFT_MSK_IT; q++) { const int i = tid + 256 * q; if (i < nm) { const float d = sum - s_va[i]; s_va[i] = d * d; } }   // helpers.cpp:266
   }
   __syncthreads();
   if (tid == 0) s_misc[1] = sqrtf(hs_serial_sum(s_va, nm) / (float)nm);   // helpers.cpp:268
   __syncthreads();
   {
      const float sum = s_misc[0], var = s_misc[1];
      if (!((double)var < 0.0001)) {
         const float fac = 50.0f / var;
#pragma unroll
         for (int q = 0; q < HS_SIFT_PIX_IT; q++) {
            const int i = tid + 256 * q;
            if (i < HS_PATCH_PIX) {
               float v = 128 + fac * (s_patch[i] - sum);
               if (v > 255) v = 255;
               if (v < 0) v = 0;
               s_patch[i] = v;
            }
         }
      }
   }
   __syncthreads();
   // gradient magnitude / orientation, siftdesc.cpp:123-137, and the per-pixel part of samplePatch
   float2 *s_vo = reinterpret_cast<float2 *>(s_va);
#pragma unroll 1
   for (int q = 0; q < HS_SIFT_PIX_IT; q++) {
      const int i = tid + 256 * q;
      if (i < HS_PATCH_PIX) {
         const int r = i / HS_PATCH, c = i - r * HS_PATCH;
         float gx, gy;
         hs_grad(s_patch, HS_PATCH, r, c, gx, gy);
         const float grad = sqrtf(gx * gx + gy * gy);
         const float ori = hm_atan2f_sel(gy, gx);
         // float(orientationBins) * (ori + 2*M_PI) / (2*M_PI), evaluated in double (M_PI)
         const float o = hm_sift_orient_coord(ori);
         s_vo[i] = make_float2(tb.sift_mask[i] * grad, o);
      }
      // the loop is unrolled only so that rg.mask[q] is a register; do not let the scheduler
      // interleave the iterations (7 atan2 bodies in flight cost ~80 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
   }
   __syncthreads();
   // samplePatch siftdesc.cpp:51-81
   if (tid < 64 && !(flags & 2)) {
      const int cell = tid >> 2, cb_r = cell >> 2, cb_c = cell & 3;
      const int bA = tid & 3, bB = bA + 4;
      const int pA = (bA + 7) & 7, pB = (bB + 7) & 7;   // a pixel whose bo0 is pA feeds bin bA through bo1
      float cwc[16];
#pragma unroll
      for (int j = 0; j < 16; j++) cwc[j] = s_cw[cb_c * 16 + j];
      float accA = 0.0f, accB = 0.0f;
      for (int i = 0; i < 16; i++) {
         const int r = 8 * cb_r + i;   // <= 39
         const float wr = s_cw[cb_r * 16 + i];
         const float2 *row = s_vo + r * HS_PATCH + 8 * cb_c;
#pragma unroll 2
         for (int j = 0; j < 16; j++) {
            const float2 q = row[j];
            const float wc = cwc[j] * q.x;   // w[c] * (mask*grad)
            const float v = wr * wc;
            const int bo0 = ((int)q.y) & 7;
            const float wo1 = q.y - (float)(int)q.y;
            const float wo0 = 1.0f - wo1;
            const bool pos = v > 0.0f;
            const float t0 = pos ? v * wo0 : 0.0f;   // goes to bin bo0
            const float t1 = pos ? v * wo1 : 0.0f;   // goes to bin bo0+1
            accA += (bo0 == bA) ? t0 : ((bo0 == pA) ? t1 : 0.0f);
            accB += (bo0 == bB) ? t0 : ((bo0 == pB) ? t1 : 0.0f);
         }
      }
      s_vec[cell * 8 + bA] = accA;
      s_vec[cell * 8 + bB] = accB;
   }
   __syncthreads();
   // sample() siftdesc.cpp:98-113: normalize, clip, renormalize, quantise (s_va is free again)
   for (int pass = 0; pass < 2; pass++) {
      if (tid < 128) { const float v = s_vec[tid]; s_va[tid] = v * v; }
      __syncthreads();
      if (tid == 0) {
         const float vectlen = sqrtf(hs_serial_sum(s_va, 128));
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
   __shared__ __attribute__((aligned(16))) float s_va[2 * HS_SIFT_ARR], s_vec[128];
   __shared__ float s_patch[HS_PATCH_PIX], s_misc[8], s_tab[HS_SIFT_TAB];
   SiftRegs rg;
   hs_sift_setup(tb, s_tab, rg);
   for (int h = blockIdx.x; h < n; h += gridDim.x) {
      for (int i = threadIdx.x; i < HS_PATCH_PIX; i += blockDim.x) s_patch[i] = patches[(size_t)h * HS_PATCH_PIX + i];
      __syncthreads();
      hs_sift_block(s_patch, s_va, s_vec, s_misc, s_tab, rg, tb, k, desc + (size_t)h * 128);
   }
}

// shared tail of every patch kernel: optional patch dump + descriptor
__device__ __forceinline__ void hs_patch_finish(uint32_t h, float *s_patch, float *s_va, float *s_vec, float *s_misc, const float *s_tab,
                                                const SiftRegs &rg, const PatchIO &io, const KpTables &tb, const DConsts &k, int flags)
{
   if (io.patches)
      for (int i = threadIdx.x; i < HS_PATCH_PIX; i += 256) io.patches[(size_t)(h - io.h_base) * HS_PATCH_PIX + i] = s_patch[i];
   if (flags & 1) hs_sift_block(s_patch, s_va, s_vec, s_misc, s_tab, rg, tb, k, io.desc + (size_t)h * 128, flags);
   __syncthreads();
}

// resample of affine.cpp:131 from a fully blurred P x P window in LDS
__device__ __forceinline__ void hs_resample_full(const float *S, int P, float scale, float *s_patch)
{
   const float c0 = (float)(P >> 1);
   for (int idx = threadIdx.x; idx < HS_PATCH_PIX; idx += 256) {
      const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
      const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
      const float rx = c0 + (float)j * 0.0f, ry = c0 + (float)j * scale;
      const float wx = rx + (float)i * scale, wy = ry + (float)i * 0.0f;
      bool o2 = false;
      s_patch[idx] = hs_bilinear(S, P, P - 1, P - 1, wx, wy, o2);
   }
}

// ---------------------------------------------------------------------------------------
// k_patch_small<BIN>: P <= 41 (BIN 0) or <= 64 (BIN 1), plus the direct branch
// (imageToPatchScale <= 0.4, affine.cpp:137-141).  grid-stride over the bin's work list.
// LDS: S | T (WIN floats each; later the SIFT scratch), s_vec, s_patch, s_misc, s_tab, taps.
// ---------------------------------------------------------------------------------------
template <int BIN, bool FUSED>
__global__ __launch_bounds__(256) void k_patch_small(HessList hl, PatchWork pw, PatchIO io, KpTables tb, DConsts k, int flags)
{
   extern __shared__ __attribute__((aligned(16))) float smem[];
   constexpr int PMAX = BIN == 0 ? 41 : 64;
   constexpr int WIN = (PMAX * PMAX + 3) & ~3;
   constexpr int REGION = (2 * WIN > 2 * HS_SIFT_ARR) ? 2 * WIN : 2 * HS_SIFT_ARR;
   float *S = smem, *T = smem + WIN;
   // extraction only (FUSED = false): just S | T | taps; the resampled patch goes straight to HBM
   float *s_vec = smem + (FUSED ? REGION : 2 * WIN);
   float *s_patch = FUSED ? s_vec + 128 : nullptr;
   float *s_misc = FUSED ? s_patch + HS_SIFT_ARR : nullptr;
   float *s_tab = FUSED ? s_misc + 8 : nullptr;
   float *s_taps = FUSED ? s_tab + HS_SIFT_TAB : s_vec;   // K <= 15
   __shared__ int s_flag;

   const int tid = threadIdx.x;
   SiftRegs rg;
   if (FUSED) hs_sift_setup(tb, s_tab, rg);
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
         // direct branch, affine.cpp:137-141
         const float b11 = a11 * scale, b12 = a12 * scale, b21 = a21 * scale, b22 = a22 * scale;
         for (int idx = tid; idx < HS_PATCH_PIX; idx += 256) {
            const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
            const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
            const float rx = x + (float)j * b12, ry = y + (float)j * b22;
            const float wx = rx + (float)i * b11, wy = ry + (float)i * b21;
            bool outside = false;
            (FUSED ? s_patch : io.patches + (size_t)(h - io.h_base) * HS_PATCH_PIX)[idx] = hs_bilinear(img, imPitch, width, height, wx, wy, outside);
         }
         __syncthreads();
      } else {
         const int P = P0 + 2, half = P >> 1, pm = P - 1;
         const int K = tb.patch_tap_k[(P0 - 1) >> 1];
         const float *taps_g = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
         const int r = K >> 1;
         if (tid < K) s_taps[tid] = taps_g[tid];
         // 1. warp, affine.cpp:126 ; touching the image boundary rejects the keypoint
         // all gathers of a batch are issued before the first use (clamped index, branch-free tap)
         bool outside = false;
         constexpr int WNIT = BIN == 0 ? 7 : 8;
         const int PP = P * P;
         for (int ib = 0; ib < PP; ib += 256 * WNIT) {
            float wv[WNIT];
#pragma unroll
            for (int it = 0; it < WNIT; it++) {
               const int idx = min(ib + tid + 256 * it, PP - 1);
               const int jj = idx / P, ii = idx - jj * P;
               const int j = jj - half, i = ii - half;
               const float rx = x + (float)j * a12, ry = y + (float)j * a22;
               const float wx = rx + (float)i * a11, wy = ry + (float)i * a21;
               wv[it] = (flags & 16) ? 1.0f : hs_bilinear(img, imPitch, width, height, wx, wy, outside);
            }
#pragma unroll
            for (int it = 0; it < WNIT; it++) {
               const int idx = ib + tid + 256 * it;
               if (idx < PP) S[idx] = wv[it];
            }
         }
         if (outside) s_flag = 1;
         __syncthreads();
         rejected = s_flag != 0;
         if (!rejected) {
            // 2a. row pass
            for (int idx = tid; idx < ((flags & 8) ? 0 : P * P); idx += 256) {
               const int yy = idx / P, xx = idx - yy * P;
               const float *Srow = S + yy * P;
               float t;
               if (K <= 5) {
                  t = Srow[xx] * s_taps[r] + (Srow[max(xx - 1, 0)] + Srow[min(xx + 1, pm)]) * s_taps[r + 1];
                  if (K == 5) t = t + (Srow[max(xx - 2, 0)] + Srow[min(xx + 2, pm)]) * s_taps[r + 2];
               } else {
                  t = s_taps[0] * Srow[min(max(xx - r, 0), pm)];
                  for (int j = 1; j < K; j++) t += s_taps[j] * Srow[min(max(xx - r + j, 0), pm)];
               }
               T[idx] = t;
            }
            __syncthreads();
            // 2b. column pass, result back into S
            for (int idx = tid; idx < ((flags & 8) ? 0 : P * P); idx += 256) {
               const int yy = idx / P, xx = idx - yy * P;
               float d = s_taps[r] * T[idx];
               for (int j = 1; j <= r; j++) d += s_taps[r + j] * (T[min(yy + j, pm) * P + xx] + T[max(yy - j, 0) * P + xx]);
               S[idx] = d;
            }
            __syncthreads();
            // 3. resample, affine.cpp:131
            hs_resample_full(S, P, scale, FUSED ? s_patch : io.patches + (size_t)(h - io.h_base) * HS_PATCH_PIX);
            __syncthreads();
         }
      }
      if (rejected) {
         if (tid == 0) pw.alive[h] = 0;
         __syncthreads();
         continue;
      }
      if (FUSED) hs_patch_finish(h, s_patch, smem, s_vec, s_misc, s_tab, rg, io, tb, k, flags);
   }
}

// resample of affine.cpp:131 from the blurred window, separable bookkeeping: the sample
// coordinate of output (jj, ii) is (c0 + (ii - 20) * scale, c0 + (jj - 20) * scale) (the cross
// terms of the interpolate() call are multiplied by 0.0f and vanish exactly), so the integer
// part and the fraction are tabulated once per keypoint for the 41 positions of an axis:
//   tab_i[m] = floor(w_m) (or -1 when a tap would leave the window), tab_f[m] = w_m - floor(w_m).
__device__ __forceinline__ void hs_resample_table(int P, float scale, int *tab_i, float *tab_f)
{
   const int m = threadIdx.x;
   if (m < HS_PATCH) {
      const float c0 = (float)(P >> 1);
      const float w = c0 + (float)(m - (HS_PATCH >> 1)) * scale;
      const float f = floorf(w);
      const bool in = f >= 0.0f && f < (float)(P - 1);   // helpers.cpp:227-240 with width = height = P - 1
      tab_i[m] = in ? (int)f : -1;
      tab_f[m] = w - f;
   }
}

__device__ __forceinline__ void hs_resample_full_tab(const float *S, int P, const int *tab_i, const float *tab_f, float *out)
{
   for (int idx = threadIdx.x; idx < HS_PATCH_PIX; idx += 256) {
      const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
      const int xi = tab_i[ii], yi = tab_i[jj];
      const float wx = tab_f[ii], wy = tab_f[jj];
      const bool in = (xi | yi) >= 0;
      const float *p = S + (in ? yi * P + xi : 0);
      const float p00 = p[0], p01 = p[1], p10 = p[P], p11 = p[P + 1];
      const float v = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
      out[idx] = in ? v : 0.0f;
   }
}

// ---------------------------------------------------------------------------------------
// k_patch_extract_small<BIN>: the extraction-only form of k_patch_small (the descriptor runs in
// kernels_sift.h): warp -> blur -> resample for windows P <= 43 (BIN 0) / 66 (BIN 1), result
// straight to io.patches.  Same arithmetic, cheaper addressing:
//   * the window S is stored with r replicated columns on either side and the row-pass plane
//     T with r replicated rows above and below (BORDER_REPLICATE materialised), so no tap
//     needs an index clamp;
//   * the tap count K is a template parameter (K = 5..15 here, affine.cpp:129): the loops
//     are unrolled, taps sit in registers and all LDS reads of an output are in flight together
//     (the rolled loop waited one LDS round trip per tap);
//   * idx -> (row, column) uses a float reciprocal (exact for these sizes) instead of the
//     ~20-instruction integer division.
// LDS: S[PMAX][PMAX + 14] | T[PMAX + 14][PMAX] | taps  (18 KB / 40 KB: 8 / 4 blocks per CU).
// ---------------------------------------------------------------------------------------
#define HS_SMALL_RMAX 7
#ifndef HS_WNIT1
#define HS_WNIT1 6
#endif
#ifndef HS_WNIT0
#define HS_WNIT0 4
#endif

__device__ __forceinline__ int hs_div_small(int idx, float inv)   // floor(idx / P) for idx < 2^16, P < 2^8, inv = 1.0f / P
{
   return (int)(((float)idx + 0.5f) * inv);
}

template <int KT, int SPITCH, int TPITCH>
__device__ __forceinline__ void hs_small_blur(float *S, float *T, int P, const float *s_taps, int Krt)
{
   const int K = KT ? KT : Krt, r = K >> 1;
   const int tid = threadIdx.x;
   const float invP = 1.0f / (float)P;
   float kk[KT ? KT : 2 * HS_SMALL_RMAX + 1];
#pragma unroll
   for (int j = 0; j < (KT ? KT : 2 * HS_SMALL_RMAX + 1); j++) kk[j] = (j < K) ? s_taps[j] : 0.0f;
   // replicated border columns of S
   {
      const float inv2r = 1.0f / (float)(2 * r);
      for (int i = tid; i < 2 * r * P; i += 256) {
         const int yy = hs_div_small(i, inv2r), m = i - yy * 2 * r;
         float *row = S + yy * SPITCH;
         if (m < r) row[m] = row[r];
         else row[P + m] = row[r + P - 1];   // column r + P + (m - r)
      }
   }
   __syncthreads();
   // row pass: T[r + y][x]; rows y = 0 and y = P-1 are also written into the r border rows
   for (int idx = tid; idx < P * P; idx += 256) {
      const int yy = hs_div_small(idx, invP), xx = idx - yy * P;
      const float *sp = S + yy * SPITCH + xx;   // sp[j] = S[clamp(xx - r + j)]
      float t;
      if (K <= 5) {
         t = sp[r] * kk[r] + (sp[r - 1] + sp[r + 1]) * kk[r + 1];
         if (K == 5) t = t + (sp[r - 2] + sp[r + 2]) * kk[r + 2];
      } else {
         t = kk[0] * sp[0];
#pragma unroll
         for (int j = 1; j < K; j++) t += kk[j] * sp[j];
      }
      T[(r + yy) * TPITCH + xx] = t;
      if (yy == 0)
         for (int j = 0; j < r; j++) T[j * TPITCH + xx] = t;
      if (yy == P - 1)
         for (int j = 0; j < r; j++) T[(r + P + j) * TPITCH + xx] = t;
   }
   __syncthreads();
   // column pass, blurred window back into S with pitch P
   for (int idx = tid; idx < P * P; idx += 256) {
      const int yy = hs_div_small(idx, invP), xx = idx - yy * P;
      const float *tp = T + (r + yy) * TPITCH + xx;
      float d = kk[r] * tp[0];
#pragma unroll
      for (int j = 1; j <= r; j++) d += kk[r + j] * (tp[j * TPITCH] + tp[-j * TPITCH]);
      S[idx] = d;
   }
   __syncthreads();
}

template <int BIN>
__global__ __launch_bounds__(256) void k_patch_extract_small(HessList hl, PatchWork pw, PatchIO io, KpTables tb, int flags)
{
   extern __shared__ __attribute__((aligned(16))) float smem[];
   constexpr int PMAX = BIN == 0 ? 41 : 64;   // the bins are cut on P = P0 + 2 (hs_patch_bin)
   constexpr int SPITCH = PMAX + 2 * HS_SMALL_RMAX, TPITCH = PMAX;
   constexpr int SSZ = (PMAX * SPITCH + 3) & ~3;
   float *S = smem, *T = smem + SSZ, *s_taps = T + (PMAX + 2 * HS_SMALL_RMAX) * TPITCH;
   __shared__ int s_flag;
   __shared__ int s_tab_i[HS_PATCH];
   __shared__ float s_tab_f[HS_PATCH];

   const int tid = threadIdx.x;
   const uint32_t cnt = min(pw.bin_count[BIN], pw.cap);
   const int imCols = io.image.cols, imRows = io.image.rows, imPitch = io.image.pitch;
   const int width = imCols - 1, height = imRows - 1;

   for (uint32_t wi = blockIdx.x; wi < cnt; wi += gridDim.x) {
      const uint32_t h = pw.bin_items[(size_t)BIN * pw.cap + wi];
      const int b = hl.meta[h] >> 8;
      const float *img = io.image.img(b);
      const float x = hl.x[h], y = hl.y[h];
      const float a11 = pw.A[4 * h], a12 = pw.A[4 * h + 1], a21 = pw.A[4 * h + 2], a22 = pw.A[4 * h + 3];
      const int P0 = pw.P0[h];
      const float scale = (float)P0 / (float)HS_PATCH;
      float *out = io.patches + (size_t)(h - io.h_base) * HS_PATCH_PIX;
      const HsPlaneBuf pbuf = hs_plane_buf(img, imRows, imPitch);
      if (!((double)scale > 0.4)) {
         // direct branch, affine.cpp:137-141
         const float b11 = a11 * scale, b12 = a12 * scale, b21 = a21 * scale, b22 = a22 * scale;
         // 7 taps per thread in two batches (4 + 3): one batch of 7 costs 23 more VGPRs and with them
         // three of the eight wavefronts a SIMD can hold
#pragma unroll
         for (int h0 = 0; h0 < HS_SIFT_PIX_IT; h0 += 4) {
            float dv[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
               if (h0 + t < HS_SIFT_PIX_IT) {
                  const int idx = min(tid + 256 * (h0 + t), HS_PATCH_PIX - 1);
                  const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
                  const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
                  const float rx = x + (float)j * b12, ry = y + (float)j * b22;
                  const float wx = rx + (float)i * b11, wy = ry + (float)i * b21;
                  bool outside = false;
                  dv[t] = hs_bilinear_buf(pbuf, width, height, wx, wy, outside);
               }
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
               if (h0 + t < HS_SIFT_PIX_IT) HS_KEEP(dv[t]);
#pragma unroll
            for (int t = 0; t < 4; t++) {
               const int idx = tid + 256 * (h0 + t);
               if (h0 + t < HS_SIFT_PIX_IT && idx < HS_PATCH_PIX) out[idx] = dv[t];
            }
         }
         continue;
      }
      const int P = P0 + 2, half = P >> 1;
      const int K = tb.patch_tap_k[(P0 - 1) >> 1], r = K >> 1;
      const float *taps_g = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
      if (tid == 0) s_flag = 0;
      if (tid < K) s_taps[tid] = taps_g[tid];
      hs_resample_table(P, scale, s_tab_i, s_tab_f);
      __syncthreads();
      // 1. warp, affine.cpp:126 ; touching the image boundary rejects the keypoint.  All gathers of
      // a batch are issued before the first use (clamped index, branch-free tap).
      bool outside = false;
      constexpr int WNIT = BIN == 0 ? HS_WNIT0 : HS_WNIT1;
      const int PP = P * P;
      const float invP = 1.0f / (float)P;
      for (int ib = 0; ib < PP; ib += 256 * WNIT) {
         float wv[WNIT];
#pragma unroll
         for (int it = 0; it < WNIT; it++) {
            const int idx = min(ib + tid + 256 * it, PP - 1);
            const int jj = hs_div_small(idx, invP), ii = idx - jj * P;
            const int j = jj - half, i = ii - half;
            const float rx = x + (float)j * a12, ry = y + (float)j * a22;
            const float wx = rx + (float)i * a11, wy = ry + (float)i * a21;
            wv[it] = hs_bilinear_buf(pbuf, width, height, wx, wy, outside);
         }
#pragma unroll
         for (int it = 0; it < WNIT; it++) HS_KEEP(wv[it]);
#pragma unroll
         for (int it = 0; it < WNIT; it++) {
            const int idx = ib + tid + 256 * it;
            if (idx < PP) {
               const int jj = hs_div_small(idx, invP), ii = idx - jj * P;
               S[jj * SPITCH + r + ii] = wv[it];
            }
         }
      }
      if (outside) s_flag = 1;
      __syncthreads();
      if (s_flag != 0) {
         if (tid == 0) pw.alive[h] = 0;
         __syncthreads();
         continue;
      }
      // 2. blur, affine.cpp:129 (pinned cv::GaussianBlur order, see the file header)
      switch (K) {
         case 3: hs_small_blur<3, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
         case 5: hs_small_blur<5, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
         case 7: hs_small_blur<7, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
         case 9: hs_small_blur<9, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
         case 11: hs_small_blur<11, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
         case 13: hs_small_blur<13, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
         case 15: hs_small_blur<15, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
         default: hs_small_blur<0, SPITCH, TPITCH>(S, T, P, s_taps, K); break;
      }
      // 3. resample, affine.cpp:131
      hs_resample_full_tab(S, P, s_tab_i, s_tab_f, out);
      __syncthreads();
   }
}

// Four column-pass sums at once: rows (y0, y0+1) x needed columns (q, q+1) of the row-pass plane
// Tp[rows][82].  Each sum keeps the SymmColumnFilter order d = k[r]*T[y]; d += k[r+j]*(T[y+j]+T[y-j]);
// the four chains are interleaved so that their loads overlap.
__device__ __forceinline__ void hs_colpass4(const float *__restrict__ Tp, int y0, int q, int pm, const float *__restrict__ taps, int r,
                                            float &p00, float &p01, float &p10, float &p11)
{
   const int ya = min(max(y0, 0), pm), yb = min(max(y0 + 1, 0), pm);
   const float kc = taps[r];
   float d00 = kc * Tp[(long long)ya * HS_NEED + q], d01 = kc * Tp[(long long)ya * HS_NEED + q + 1];
   float d10 = kc * Tp[(long long)yb * HS_NEED + q], d11 = kc * Tp[(long long)yb * HS_NEED + q + 1];
#pragma unroll 2
   for (int j = 1; j <= r; j++) {
      const float kj = taps[r + j];
      const float *ap = Tp + (long long)min(y0 + j, pm) * HS_NEED + q, *am = Tp + (long long)max(y0 - j, 0) * HS_NEED + q;
      const float *bp = Tp + (long long)min(y0 + 1 + j, pm) * HS_NEED + q, *bm = Tp + (long long)max(y0 + 1 - j, 0) * HS_NEED + q;
      const float s00 = ap[0] + am[0], s01 = ap[1] + am[1], s10 = bp[0] + bm[0], s11 = bp[1] + bm[1];
      d00 += kj * s00; d01 += kj * s01; d10 += kj * s10; d11 += kj * s11;
   }
   p00 = d00; p01 = d01; p10 = d10; p11 = d11;
}

// resample of affine.cpp:131 when only the row-pass plane at the 82 needed columns exists:
// the four blurred neighbours of each output are column-pass sums evaluated on the spot.
// xq(q) = floor(c0 + ((q>>1)-20)*scale) + (q&1) is the needed column (and row) list.
__device__ __forceinline__ void hs_resample_reduced(const float *__restrict__ Tp, int P, float scale, const float *__restrict__ taps, int r,
                                                    float *s_patch)
{
   const float c0 = (float)(P >> 1);
   const int pm = P - 1;
   for (int idx = threadIdx.x; idx < HS_PATCH_PIX; idx += 256) {
      const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
      const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
      const float rx = c0 + (float)j * 0.0f, ry = c0 + (float)j * scale;
      float wx = rx + (float)i * scale, wy = ry + (float)i * 0.0f;
      const float fx = floorf(wx), fy = floorf(wy);
      wx -= fx; wy -= fy;
      float p00, p01, p10, p11;
      hs_colpass4(Tp, (int)fy, 2 * ii, pm, taps, r, p00, p01, p10, p11);
      s_patch[idx] = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
   }
}

// The same four column-pass sums with the loads batched: the two chains share their rows (row
// y0 + j of chain (y0) is row (y0 + 1) + (j - 1) of chain (y0 + 1)), and the loads of JC tap steps
// are issued together - the plane is read through L2 / HBM, where a load per tap step followed
// by its use is a full round trip per step.  CLAMP = false: T points at window row 0 of a plane
// stored with r replicated rows above and below (k_patch_mid's HBM slot), no index clamps.
//   chain a = window row y0, chain b = window row y0 + 1, columns q and q + 1.
template <int JC, bool CLAMP>
__device__ __forceinline__ void hs_colpass4_rows(const float *__restrict__ T, int y0, int q, int pm, const float *__restrict__ taps, int r,
                                                 float &p00, float &p01, float &p10, float &p11)
{
   // row y of the window -> float2 at T[row][q]; CLAMP: BORDER_REPLICATE by index clamp (unpadded plane)
   auto ld = [&](int y) {
      const int row = CLAMP ? min(max(y, 0), pm) : y;
      return *reinterpret_cast<const float2 *>(T + row * HS_NEED + q);
   };
   const float2 c0 = ld(y0), c1 = ld(y0 + 1);
   const float kc = taps[r];
   float d00 = kc * c0.x, d01 = kc * c0.y, d10 = kc * c1.x, d11 = kc * c1.y;
   float2 pj = c1;   // row y0 + j      (j = 1)
   float2 mj = c0;   // row y0 + 1 - j  (j = 1)
   for (int j0 = 1; j0 <= r; j0 += JC) {
      float2 pn[JC], mn[JC];   // rows y0 + j + 1 and y0 - j
#pragma unroll
      for (int u = 0; u < JC; u++) {
         const int j = min(j0 + u, r);   // steps past r re-read step r's rows, unused
         pn[u] = ld(y0 + j + 1);
         mn[u] = ld(y0 - j);
      }
#pragma unroll
      for (int u = 0; u < JC; u++) {
         const int j = j0 + u;
         if (j <= r) {   // block-uniform
            const float kj = taps[r + j];
            const float s00 = pj.x + mn[u].x, s01 = pj.y + mn[u].y, s10 = pn[u].x + mj.x, s11 = pn[u].y + mj.y;
            d00 += kj * s00; d01 += kj * s01; d10 += kj * s10; d11 += kj * s11;
            pj = pn[u];
            mj = mn[u];
         }
      }
   }
   p00 = d00; p01 = d01; p10 = d10; p11 = d11;
}

// resample of affine.cpp:131 from the row-pass plane at the 82 needed columns (PADDED: T points at
// window row 0 of a plane with r replicated rows above and below; otherwise rows are clamped)
template <bool PADDED>
__device__ __forceinline__ void hs_resample_reduced_batched(const float *__restrict__ T, int P, float scale, const float *__restrict__ taps, int r,
                                                            float *s_patch)
{
   const float c0 = (float)(P >> 1);
   for (int idx = threadIdx.x; idx < HS_PATCH_PIX; idx += 256) {
      const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
      const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
      const float rx = c0 + (float)j * 0.0f, ry = c0 + (float)j * scale;
      float wx = rx + (float)i * scale, wy = ry + (float)i * 0.0f;
      const float fx = floorf(wx), fy = floorf(wy);
      wx -= fx; wy -= fy;
      const int y0 = min(max((int)fy, 0), P - 2);   // always inside: |j * scale| < P0 / 2
      float p00, p01, p10, p11;
      hs_colpass4_rows<8, !PADDED>(T, y0, 2 * ii, P - 1, taps, r, p00, p01, p10, p11);
      s_patch[idx] = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
   }
}

// one window row: warp (affine.cpp:126) into the wave's LDS row, then the row pass at the 82
// needed columns.  Called by all 64 lanes of a wave.  The LDS row is stored with r replicated
// border samples on either side (BORDER_REPLICATE), so the tap loop has no index clamps:
//   srow[r + x] = S[x],  srow[0..r) = S[0],  srow[r+P .. r+P+r) = S[P-1]     (needs P + 2r floats)
// The image gathers of NIT x 64 window pixels are issued together (branch-free taps, clamped
// index) before any of them is used; taps are read from LDS (`taps`, broadcast reads).
// Lane i < 41 owns the output pair (2i, 2i + 1); the two accumulation chains share their reads.
template <int NIT>
__device__ __forceinline__ void hs_row_stream(const HsPlaneBuf &img, int width, int height, float x, float y,
                                              float a11, float a12, float a21, float a22, int P, int yy, float scale,
                                              const float *__restrict__ taps, int K, float *__restrict__ srow, float *__restrict__ out82,
                                              bool &outside, int pad_r = 0)
{
   const int lane = threadIdx.x & 63, half = P >> 1, pm = P - 1, r = K >> 1;
   const int j = yy - half;
   const float rx = x + (float)j * a12, ry = y + (float)j * a22;
   for (int xb = 0; xb < P; xb += 64 * NIT) {
      float v[NIT];
#pragma unroll
      for (int it = 0; it < NIT; it++) {
         const int xx = min(xb + lane + 64 * it, pm);   // lanes past the row re-sample its last pixel (not stored)
         const int i = xx - half;
         const float wx = rx + (float)i * a11, wy = ry + (float)i * a21;
         v[it] = hs_bilinear_buf(img, width, height, wx, wy, outside);
      }
#pragma unroll
      for (int it = 0; it < NIT; it++) HS_KEEP(v[it]);
#pragma unroll
      for (int it = 0; it < NIT; it++) {
         const int xx = xb + lane + 64 * it;
         if (xx < P) srow[r + xx] = v[it];
      }
   }
   HS_WAVE_LDS_SYNC();
   {
      const float first = srow[r], last = srow[r + pm];
      for (int i = lane; i < r; i += 64) { srow[i] = first; srow[r + P + i] = last; }
   }
   HS_WAVE_LDS_SYNC();
   // Lane i < 41 owns the output pair q = 2i, 2i + 1: the two blurred columns floor(w) and floor(w) + 1
   // that output pixel i of the 41x41 resample reads.  Their tap windows overlap in all but one
   // sample, so the pair costs K + 1 LDS reads instead of 2K.  (0 <= floor(w) <= P - 2 always:
   // |(i - 20) * scale| < P0 / 2.)
   if (lane < HS_PATCH) {
      const float c0 = (float)half;
      const float w = c0 + (float)(lane - 20) * scale;
      const int x0 = min(max((int)floorf(w), 0), pm - 1);
      const float *s = srow + x0;   // s[jt] = S[clamp(x0 - r + jt)],  s[jt + 1] = S[clamp(x0 + 1 - r + jt)]
      float prev = s[1];
      float t0 = taps[0] * s[0], t1 = taps[0] * prev;
#pragma unroll 8
      for (int jt = 1; jt < K; jt++) {
         const float k = taps[jt];
         const float nxt = s[jt + 1];
         const float p0 = k * prev, p1 = k * nxt;
         t0 += p0;
         t1 += p1;
         prev = nxt;
      }
      if (out82) {
         float2 *o = reinterpret_cast<float2 *>(out82) + lane;
         *o = make_float2(t0, t1);
         // padded T' plane: the first / last window row is replicated pad_r times above / below (wave-uniform)
         if (pad_r > 0 && (yy == 0 || yy == pm)) {
            const int step = (yy == 0) ? -(HS_NEED / 2) : (HS_NEED / 2);
            for (int jr = 1; jr <= pad_r; jr++) o[jr * step] = make_float2(t0, t1);
         }
      } else {
         srow[0] = t0 + t1;   // ablation only: keep the sums alive without the global store
      }
   }
   HS_WAVE_LDS_SYNC();
}

// ---------------------------------------------------------------------------------------
// k_patch_mid: 64 < P <= 128.  Each of the 4 waves streams window rows (warp -> row pass at
// the 82 needed columns) into Tp[P][82] in LDS; then the resample evaluates the column
// pass where it reads.  LDS ~52 KB -> 3 blocks per CU.
// ---------------------------------------------------------------------------------------
#define HS_MID_PMAX 128
#define HS_MID_SROW 160   // 128 + 2 x 14 border samples, padded
#define HS_BIG_SROW 704   // 512 + 2 x 57 border samples, padded
#define HS_BIG_TAPS 128   // K <= 113 for P <= 512
#define HS_MID_RPAD 14    // K / 2 for P <= 128
#define HS_BIG_RPAD 57    // K / 2 for P <= 512

// PMAX = 128: bin 2, T' (P x 82) in LDS.  PMAX = 512: bin 3, same structure with T' in a
// per-block slot of HBM scratch (io.trows), written and re-read by the same block (L2-hot).
// FUSED = false: extraction only (the descriptor runs in kernels_sift.h), no SIFT scratch in LDS.
// TPG: T' rows in a per-block HBM slot instead of LDS (always for PMAX = 512).
template <int PMAX, bool FUSED, bool TPG = (PMAX > 128)>
__global__ __launch_bounds__(256) void k_patch_mid(HessList hl, PatchWork pw, PatchIO io, KpTables tb, DConsts k, int flags)
{
   constexpr bool BIG = PMAX > HS_MID_PMAX;
   static_assert(TPG || !BIG, "P > 128 does not fit T' in LDS");
   constexpr int BIN = BIG ? 3 : 2;
   constexpr int SROW = BIG ? HS_BIG_SROW : HS_MID_SROW;
   constexpr int NTAP = BIG ? HS_BIG_TAPS : 32;
   // T' in LDS (bin 2) doubles as the SIFT scratch; bin 3 keeps T' in HBM and needs LDS scratch only when fused
   constexpr int REGION = TPG ? (FUSED ? 2 * HS_SIFT_ARR : 0) : HS_MID_PMAX * HS_NEED;
   constexpr int NIT = BIG ? 4 : 2;
   extern __shared__ __attribute__((aligned(16))) float smem[];
   float *s_vec = smem + REGION;
   float *s_patch = s_vec + (FUSED ? 128 : 0);
   float *s_misc = s_patch + HS_SIFT_ARR;
   float *s_tab = s_misc + 8;
   float *s_taps = s_tab + (FUSED ? HS_SIFT_TAB : 0);
   float *s_srow = s_taps + NTAP;                      // 4 waves x SROW
   __shared__ int s_flag;
   // TPG: T' with r replicated rows above and below, row index r + y (hs_colpass4_padded)
   constexpr int RPAD = BIG ? HS_BIG_RPAD : HS_MID_RPAD;
   float *Tp = TPG ? io.trows + (size_t)blockIdx.x * ((size_t)(PMAX + 2 * RPAD) * HS_NEED) : smem;

   const int tid = threadIdx.x, wave = tid >> 6;
   SiftRegs rg;
   if (FUSED) hs_sift_setup(tb, s_tab, rg);
   const uint32_t cnt = min(pw.bin_count[BIN], pw.cap);
   const int imPitch = io.image.pitch, width = io.image.cols - 1, height = io.image.rows - 1;

   for (uint32_t wi = blockIdx.x; wi < cnt; wi += gridDim.x) {
      const uint32_t h = pw.bin_items[(size_t)BIN * pw.cap + wi];
      const int b = hl.meta[h] >> 8;
      const HsPlaneBuf ib = hs_plane_buf(io.image.img(b), io.image.rows, imPitch);
      const float x = hl.x[h], y = hl.y[h];
      const float a11 = pw.A[4 * h], a12 = pw.A[4 * h + 1], a21 = pw.A[4 * h + 2], a22 = pw.A[4 * h + 3];
      const int P0 = pw.P0[h], P = P0 + 2;
      const float scale = (float)P0 / (float)HS_PATCH;
      const int K = tb.patch_tap_k[(P0 - 1) >> 1];
      const float *taps_g = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
      if (tid == 0) s_flag = 0;
      if (tid < K) s_taps[tid] = taps_g[tid];
      __syncthreads();
      bool outside = false;
#pragma unroll 1
      for (int yy = wave; yy < ((flags & 64) ? 4 : P); yy += 4)
         hs_row_stream<NIT>(ib, width, height, x, y, a11, a12, a21, a22, P, yy,
                            scale, s_taps, (flags & 8) ? 3 : K, s_srow + wave * SROW,
                            (flags & 128) ? nullptr : Tp + (size_t)(yy + (TPG ? (K >> 1) : 0)) * HS_NEED, outside, TPG ? (K >> 1) : 0);
      if (outside) s_flag = 1;
      __syncthreads();   // workgroup-scope release/acquire: the T' rows of all four waves are visible
      if (s_flag != 0) {
         if (tid == 0) pw.alive[h] = 0;
         __syncthreads();
         continue;
      }
      if (TPG) hs_resample_reduced_batched<true>(Tp + (K >> 1) * HS_NEED, P, scale, s_taps, K >> 1, s_patch);
      else hs_resample_reduced(Tp, P, scale, s_taps, (flags & 32) ? 1 : (K >> 1), s_patch);
      __syncthreads();
      if (FUSED) hs_patch_finish(h, s_patch, smem, s_vec, s_misc, s_tab, rg, io, tb, k, flags);
      else {
         for (int i = tid; i < HS_PATCH_PIX; i += 256) io.patches[(size_t)(h - io.h_base) * HS_PATCH_PIX + i] = s_patch[i];
         __syncthreads();
      }
   }
}

// ---------------------------------------------------------------------------------------
// Large windows (P > 128, ~2 % of the keypoints, most of the blur work).
// k_patch_large_rows: grid-stride over ALL window rows of the round's keypoints, one
//   wavefront per row (binary search of the row id in the prefix of P); writes T' rows to HBM.
// k_patch_large_finish: one block per keypoint: column pass at the resample taps + SIFT.
// ---------------------------------------------------------------------------------------
#define HS_LARGE_CHUNK 16   // consecutive window rows per wavefront task

// dynamic LDS: per wave  srow_stride floats (window row + borders)  +  tap_stride floats (taps)
__global__ __launch_bounds__(256) void k_patch_large_rows(HessList hl, PatchWork pw, PatchIO io, KpTables tb, int srow_stride, int tap_stride, int flags)
{
   extern __shared__ __attribute__((aligned(16))) float smem[];
   const int wave = threadIdx.x >> 6;
   float *srow = smem + (size_t)wave * (srow_stride + tap_stride);
   float *stap = srow + srow_stride;
   const uint32_t *pre = io.row_prefix;
   const uint32_t row_lo = pre[io.item0], row_hi = pre[io.item1];
   const int imPitch = io.image.pitch, width = io.image.cols - 1, height = io.image.rows - 1;
   const uint32_t ntasks = (row_hi - row_lo + HS_LARGE_CHUNK - 1) / HS_LARGE_CHUNK;
   for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
      uint32_t row = row_lo + task * HS_LARGE_CHUNK;
      const uint32_t row_end = min(row + HS_LARGE_CHUNK, row_hi);
      // item of the first row: largest kk in [item0, item1) with pre[kk] <= row (one search per task)
      uint32_t lo = io.item0, hi = io.item1;
      while (hi - lo > 1) {
         const uint32_t mid = (lo + hi) >> 1;
         if (pre[mid] <= row) lo = mid; else hi = mid;
      }
      uint32_t it = lo;
      while (row < row_end) {
         const uint32_t it_rows_end = min(pre[it + 1], row_end);
         const uint32_t h = pw.bin_items[(size_t)io.bin * pw.cap + it];
         const int b = hl.meta[h] >> 8;
         const int P0 = pw.P0[h], P = P0 + 2;
         const float scale = (float)P0 / (float)HS_PATCH;
         const int K = tb.patch_tap_k[(P0 - 1) >> 1];
         const float *taps = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
         const HsPlaneBuf ib = hs_plane_buf(io.image.img(b), io.image.rows, imPitch);
         const float kx = hl.x[h], ky = hl.y[h];
         const float a11 = pw.A[4 * h], a12 = pw.A[4 * h + 1], a21 = pw.A[4 * h + 2], a22 = pw.A[4 * h + 3];
         const uint32_t first = pre[it];
         // this item's taps -> the wave's LDS tap buffer (broadcast reads in the tap loop)
         for (int i = threadIdx.x & 63; i < K; i += 64) stap[i] = taps[i];
         HS_WAVE_LDS_SYNC();
         bool outside = false;
         for (; row < it_rows_end; row++)
            hs_row_stream<8>(ib, width, height, kx, ky, a11, a12, a21, a22, P, (int)(row - first), scale, stap, K, srow,
                             io.trows + (size_t)(row - row_lo) * HS_NEED, outside);
         if (outside) pw.alive[h] = 0;   // every writer stores the same value
         it++;
      }
   }
}

__global__ __launch_bounds__(256, 5) void k_patch_large_finish(HessList hl, PatchWork pw, PatchIO io, KpTables tb, DConsts k, int flags)
{
   __shared__ __attribute__((aligned(16))) float s_va[2 * HS_SIFT_ARR], s_vec[128];
   __shared__ float s_patch[HS_SIFT_ARR], s_misc[8], s_tab[HS_SIFT_TAB];
   const uint32_t *pre = io.row_prefix;
   const uint32_t row_lo = pre[io.item0];
   SiftRegs rg;
   hs_sift_setup(tb, s_tab, rg);
   for (uint32_t it = io.item0 + blockIdx.x; it < io.item1; it += gridDim.x) {
      const uint32_t h = pw.bin_items[(size_t)io.bin * pw.cap + it];
      if (!pw.alive[h]) continue;   // uniform for the block
      const int P0 = pw.P0[h], P = P0 + 2;
      const float scale = (float)P0 / (float)HS_PATCH;
      const int K = tb.patch_tap_k[(P0 - 1) >> 1];
      const float *taps = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
      hs_resample_reduced_batched<false>(io.trows + (size_t)(pre[it] - row_lo) * HS_NEED, P, scale, taps, K >> 1, s_patch);
      __syncthreads();
      hs_patch_finish(h, s_patch, s_va, s_vec, s_misc, s_tab, rg, io, tb, k, flags);
   }
}

// P of the i-th item of the large bin (scan operand for the row prefix)
struct LoadLargeP {
   const uint32_t *items;
   const int32_t *P0;
   __device__ uint32_t operator()(long long i) const { return (uint32_t)(P0[items[i]] + 2); }
};
