// kernels_sift.h -- SIFT descriptor (siftdesc.cpp:115-140, helpers.cpp:246-281) over patches
// held in HBM, as three kernels whose parallel axis matches the structure of the reference's
// arithmetic instead of fighting it:
//
//  k_sift_meanvar   the photometric mean / variance are long SEQUENTIAL float sums
//                   (helpers.cpp:253-266, 1245 terms each).  One THREAD per keypoint runs the
//                   chain, 64 keypoints per wavefront at full lane efficiency; the patch columns
//                   are transposed through LDS so that global loads stay coalesced.
//  k_sift_hist      one WAVEFRONT per keypoint: normalise, gradients + hm_atan2f, then the
//                   4x4x8 histogram with lane = (spatial cell, orientation pair) walking its
//                   16x16 support in raster order (siftdesc.cpp:51-81).  No inter-wave barriers.
//  k_sift_quantize  normalize / clip / renormalize / quantise (siftdesc.cpp:83-113): the two
//                   128-term sequential sums again run one thread per keypoint.
//
// All sums are accumulated in the reference's order; nothing is re-associated.
#pragma once
#include "kernels_patch.h"

struct SiftIO {
   const float *patches;     // [n][1681] (index = h - h_lo)
   const int32_t *alive;     // [n] flags, indexed by h
   float *meanvar;           // [n][2]
   float *vec;               // [n][128] un-normalised histogram
   uint8_t *desc;            // [n][128] (index = h, absolute)
   uint32_t h_lo, h_hi;
};

#define SM_TILE 64
#define SM_STRIDE 65   // LDS row stride: lane k walking row k is conflict-free

// grid: ceil(n / 64) blocks of 64 threads
__global__ __launch_bounds__(64) void k_sift_meanvar(SiftIO io, KpTables tb)
{
   __shared__ float s_tile[SM_TILE * SM_STRIDE];
   const int lane = threadIdx.x;
   const uint32_t n = io.h_hi - io.h_lo;
   const uint32_t k0 = blockIdx.x * SM_TILE;           // first keypoint (relative) of this block
   const int nm = tb.n_masked;
   const float gsum = (float)nm;
   const uint32_t kmine = min(k0 + lane, n - 1);
   // pass 0: sum ; pass 1: sum of squared deviations
   float sum = 0.0f, mean = 0.0f;
   for (int pass = 0; pass < 2; pass++) {
      float acc = 0.0f;
      for (int c0 = 0; c0 < nm; c0 += SM_TILE) {
         const int cnt = min(SM_TILE, nm - c0);
         // stage: row k = keypoint k0+k, column l = masked pixel c0+l ; coalesced along l
         const int pix = (lane < cnt) ? tb.mask_idx[c0 + lane] : 0;
#pragma unroll 8
         for (int k = 0; k < SM_TILE; k++) {
            const uint32_t kp = min(k0 + k, n - 1);
            s_tile[k * SM_STRIDE + lane] = io.patches[(size_t)kp * HS_PATCH_PIX + pix];
         }
         __syncthreads();
         const float *row = s_tile + lane * SM_STRIDE;
         if (pass == 0) {
            for (int l = 0; l < cnt; l++) acc += row[l];                                   // helpers.cpp:257
         } else {
            for (int l = 0; l < cnt; l++) { const float d = mean - row[l]; acc += d * d; }   // helpers.cpp:266
         }
         __syncthreads();
      }
      if (pass == 0) { sum = acc; mean = sum / gsum; }
      else {
         const float var = sqrtf(acc / gsum);   // helpers.cpp:268
         if (k0 + lane < n) { io.meanvar[2 * (size_t)kmine] = mean; io.meanvar[2 * (size_t)kmine + 1] = var; }
      }
   }
}

// one wavefront (64-thread block) per keypoint, grid-stride over [h_lo, h_hi)
__global__ __launch_bounds__(64) void k_sift_hist(SiftIO io, KpTables tb, int flags)
{
   __shared__ __attribute__((aligned(16))) float s_vo[2 * HS_SIFT_ARR];
   __shared__ float s_patch[HS_SIFT_ARR], s_tab[HS_SIFT_TAB];
   const int tid = threadIdx.x;
   {
      int *s_bin0 = reinterpret_cast<int *>(s_tab), *s_bin1 = s_bin0 + HS_PATCH;
      float *s_w0 = s_tab + 2 * HS_PATCH, *s_w1 = s_tab + 3 * HS_PATCH, *s_cw = s_tab + 4 * HS_PATCH + 4;
      if (tid < HS_PATCH) { s_bin0[tid] = tb.bin0[tid]; s_bin1[tid] = tb.bin1[tid]; s_w0[tid] = tb.w0[tid]; s_w1[tid] = tb.w1[tid]; }
      __syncthreads();
      // cell weights, see hs_sift_setup
      const int b = tid >> 4, i = tid & 15, r = 8 * b + i;
      float w = 0.0f;
      if (r < HS_PATCH) {
         if (s_bin0[r] == 8 * b && s_w0[r] != 0.0f) w = s_w0[r];
         else if (s_bin1[r] == 8 * b) w = s_w1[r];
      }
      s_cw[tid] = w;
      __syncthreads();
   }
   const float *s_cw = s_tab + 4 * HS_PATCH + 4;
   const int cell = tid >> 2, cb_r = cell >> 2, cb_c = cell & 3;
   const int bA = tid & 3, bB = bA + 4;
   const int pA = (bA + 7) & 7, pB = (bB + 7) & 7;
   float cwc[16];
#pragma unroll
   for (int j = 0; j < 16; j++) cwc[j] = s_cw[cb_c * 16 + j];
   float2 *vo = reinterpret_cast<float2 *>(s_vo);

   for (uint32_t h = io.h_lo + blockIdx.x; h < io.h_hi; h += gridDim.x) {
      if (!io.alive[h]) continue;   // block-uniform
      const uint32_t k = h - io.h_lo;
      const float mean = io.meanvar[2 * (size_t)k], var = io.meanvar[2 * (size_t)k + 1];
      const float *gp = io.patches + (size_t)k * HS_PATCH_PIX;
      // photometric normalisation helpers.cpp:269-280 while loading
      if (!((double)var < 0.0001)) {
         const float fac = 50.0f / var;
         for (int i = tid; i < HS_PATCH_PIX; i += 64) {
            float v = 128 + fac * (gp[i] - mean);
            if (v > 255) v = 255;
            if (v < 0) v = 0;
            s_patch[i] = v;
         }
      } else {
         for (int i = tid; i < HS_PATCH_PIX; i += 64) s_patch[i] = gp[i];
      }
      __syncthreads();
      // gradient magnitude / orientation siftdesc.cpp:123-137
#pragma unroll 1
      for (int i = tid; i < HS_PATCH_PIX; i += 64) {
         const int r = i / HS_PATCH, c = i - r * HS_PATCH;
         float gx, gy;
         hs_grad(s_patch, HS_PATCH, r, c, gx, gy);
         const float grad = sqrtf(gx * gx + gy * gy);
         const float ori = hm_atan2f(gy, gx);
         const float o = (float)((double)8.0f * ((double)ori + 2 * 3.14159265358979323846) / (2 * 3.14159265358979323846));
         vo[i] = make_float2(tb.sift_mask[i] * grad, o);
      }
      __syncthreads();
      // samplePatch siftdesc.cpp:51-81, see hs_sift_block
      float accA = 0.0f, accB = 0.0f;
      if (!(flags & 2)) {
         for (int i = 0; i < 16; i++) {
            const int r = 8 * cb_r + i;
            const float wr = s_cw[cb_r * 16 + i];
            const float2 *row = vo + r * HS_PATCH + 8 * cb_c;
#pragma unroll 8
            for (int j = 0; j < 16; j++) {
               const float2 q = row[j];
               const float wc = cwc[j] * q.x;
               const float v = wr * wc;
               const int bo0 = ((int)q.y) & 7;
               const float wo1 = q.y - (float)(int)q.y;
               const float wo0 = 1.0f - wo1;
               const bool pos = v > 0.0f;
               const float t0 = pos ? v * wo0 : 0.0f;
               const float t1 = pos ? v * wo1 : 0.0f;
               accA += (bo0 == bA) ? t0 : ((bo0 == pA) ? t1 : 0.0f);
               accB += (bo0 == bB) ? t0 : ((bo0 == pB) ? t1 : 0.0f);
            }
         }
      }
      io.vec[(size_t)k * 128 + cell * 8 + bA] = accA;
      io.vec[(size_t)k * 128 + cell * 8 + bB] = accB;
      __syncthreads();
   }
}

// sample() siftdesc.cpp:98-113 after samplePatch: one thread per keypoint for the serial norms.
// grid: ceil(n / 64) blocks of 64 threads; LDS tile 64 keypoints x 128 bins.
__global__ __launch_bounds__(64) void k_sift_quantize(SiftIO io, DConsts kc)
{
   __shared__ float s_t[SM_TILE * 129];
   const int lane = threadIdx.x;
   const uint32_t n = io.h_hi - io.h_lo;
   const uint32_t k0 = blockIdx.x * SM_TILE;
   // coalesced load: row k = keypoint, 128 consecutive floats
   for (int k = 0; k < SM_TILE; k++) {
      const uint32_t kp = min(k0 + k, n - 1);
      s_t[k * 129 + lane] = io.vec[(size_t)kp * 128 + lane];
      s_t[k * 129 + 64 + lane] = io.vec[(size_t)kp * 128 + 64 + lane];
   }
   __syncthreads();
   float *v = s_t + lane * 129;
   {
      float vectlen = 0.0f;
      for (int i = 0; i < 128; i++) { const float x = v[i]; vectlen += x * x; }   // siftdesc.cpp:86-90
      vectlen = sqrtf(vectlen);
      const float fac = 1.0f / vectlen;
      bool changed = false;
      for (int i = 0; i < 128; i++) {
         float x = v[i] * fac;
         if (x > kc.maxBinValue) { x = kc.maxBinValue; changed = true; }
         v[i] = x;
      }
      if (changed) {
         float l2 = 0.0f;
         for (int i = 0; i < 128; i++) { const float x = v[i]; l2 += x * x; }
         l2 = sqrtf(l2);
         const float f2 = 1.0f / l2;
         for (int i = 0; i < 128; i++) v[i] *= f2;
      }
      for (int i = 0; i < 128; i++) {
         const float q = 512.0f * v[i];
         int bq = (q == q) ? (int)q : 0;
         v[i] = (float)min(bq, 255);
      }
   }
   __syncthreads();
   // coalesced byte store: 4 bytes per lane, two keypoints per iteration
   for (int k = 0; k < SM_TILE; k++) {
      const uint32_t kp = k0 + k;
      if (kp >= n) break;
      const uint32_t h = io.h_lo + kp;
      if (!io.alive[h]) continue;
      if (lane < 32) {
         const float *r = s_t + k * 129 + 4 * lane;
         const uint32_t w = (uint32_t)r[0] | ((uint32_t)r[1] << 8) | ((uint32_t)r[2] << 16) | ((uint32_t)r[3] << 24);
         *reinterpret_cast<uint32_t *>(io.desc + (size_t)h * 128 + 4 * lane) = w;
      }
   }
}
