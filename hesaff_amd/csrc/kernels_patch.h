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
// Skipping blur outputs nobody reads does not change any value that is read: every
// evaluated tap sum uses the pinned cv::GaussianBlur order:
//   row   : t = k[0]*S[x-r]; t += k[j]*S[x-r+j]  (j ascending)       K > 5
//           S0*k0 + (S-1+S1)*k1 + (S-2+S2)*k2                         K <= 5
//   column: d = k[r]*T[y];  d += k[r+j]*(T[y+j] + T[y-j])
#pragma once
#include "kernels_keypoint.h"

#define HS_PATCH_ARR 1684  // 1681 rounded up to a multiple of 4 floats
#define HS_PATCH_PIX_IT 7  // ceil(1681 / 256)
#define HS_NEED 82         // blurred columns (and rows) the 41x41 resample reads: 2 per output

struct PatchIO {
   DPlane image;         // original float image batch (normalizeAffine samples the ORIGINAL image, hesaff.cpp:82)
   float *patches;       // [n][1681] output, row index h - h_base
   uint32_t h_base;
   float *trows;         // T' rows: per-block slots (bins 2, 3) or [rows][82] of the large bin
   const uint32_t *row_prefix;   // large bin: exclusive prefix of P over the bin's items (+ total), k_large_prefix
   uint32_t trows_cap;           // large bin: rows the T' buffer holds
   uint32_t *overflow;           // set when the large bin's rows exceed trows_cap (reported as an error by the host)
};

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
__global__ __launch_bounds__(256) void k_patch_extract_small(HessList hl, PatchWork pw, PatchIO io, KpTables tb)
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
         for (int h0 = 0; h0 < HS_PATCH_PIX_IT; h0 += 4) {
            float dv[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
               if (h0 + t < HS_PATCH_PIX_IT) {
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
               if (h0 + t < HS_PATCH_PIX_IT) HS_KEEP(dv[t]);
#pragma unroll
            for (int t = 0; t < 4; t++) {
               const int idx = tid + 256 * (h0 + t);
               if (h0 + t < HS_PATCH_PIX_IT && idx < HS_PATCH_PIX) out[idx] = dv[t];
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
#define HS_MID_BLOCKS (256 * 7)   // persistent grids of the row-streamed bins: one T' slot per block
#define HS_BIG_BLOCKS (256 * 8)

// PMAX = 128: bin 2; PMAX = 512: bin 3.  T' (P x 82, padded with K/2 replicated rows above and below) lives in
// a per-block slot of HBM scratch (io.trows), written and re-read by the same block (L2-hot).
template <int PMAX>
__global__ __launch_bounds__(256) void k_patch_mid(HessList hl, PatchWork pw, PatchIO io, KpTables tb)
{
   constexpr bool BIG = PMAX > HS_MID_PMAX;
   constexpr int BIN = BIG ? 3 : 2;
   constexpr int SROW = BIG ? HS_BIG_SROW : HS_MID_SROW;
   constexpr int NTAP = BIG ? HS_BIG_TAPS : 32;
   constexpr int NIT = BIG ? 4 : 2;
   extern __shared__ __attribute__((aligned(16))) float smem[];
   float *s_patch = smem;
   float *s_taps = s_patch + HS_PATCH_ARR;
   float *s_srow = s_taps + NTAP;                      // 4 waves x SROW
   __shared__ int s_flag;
   constexpr int RPAD = BIG ? HS_BIG_RPAD : HS_MID_RPAD;
   float *Tp = io.trows + (size_t)blockIdx.x * ((size_t)(PMAX + 2 * RPAD) * HS_NEED);

   const int tid = threadIdx.x, wave = tid >> 6;
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
      for (int yy = wave; yy < P; yy += 4)
         hs_row_stream<NIT>(ib, width, height, x, y, a11, a12, a21, a22, P, yy, scale, s_taps, K, s_srow + wave * SROW,
                            Tp + (size_t)(yy + (K >> 1)) * HS_NEED, outside, K >> 1);
      if (outside) s_flag = 1;
      __syncthreads();   // workgroup-scope release/acquire: the T' rows of all four waves are visible
      if (s_flag != 0) {
         if (tid == 0) pw.alive[h] = 0;
         __syncthreads();
         continue;
      }
      hs_resample_reduced_batched<true>(Tp + (K >> 1) * HS_NEED, P, scale, s_taps, K >> 1, s_patch);
      __syncthreads();
      for (int i = tid; i < HS_PATCH_PIX; i += 256) io.patches[(size_t)(h - io.h_base) * HS_PATCH_PIX + i] = s_patch[i];
      __syncthreads();
   }
}

// ---------------------------------------------------------------------------------------
// Large windows (P > 128, ~2 % of the keypoints, most of the blur work).
// k_patch_large_rows: grid-stride over ALL window rows of the round's keypoints, one
//   wavefront per row (binary search of the row id in the prefix of P); writes T' rows to HBM.
// k_patch_large_finish: one block per keypoint: column pass at the resample taps.
// ---------------------------------------------------------------------------------------
#define HS_LARGE_CHUNK 16   // consecutive window rows per wavefront task

// dynamic LDS: per wave  srow_stride floats (window row + borders)  +  tap_stride floats (taps)
__global__ __launch_bounds__(256) void k_patch_large_rows(HessList hl, PatchWork pw, PatchIO io, KpTables tb, int srow_stride, int tap_stride)
{
   extern __shared__ __attribute__((aligned(16))) float smem[];
   const int wave = threadIdx.x >> 6;
   float *srow = smem + (size_t)wave * (srow_stride + tap_stride);
   float *stap = srow + srow_stride;
   const uint32_t *pre = io.row_prefix;
   const uint32_t n_items = min(pw.bin_count[HS_NBINS - 1], pw.cap);
   if (n_items == 0) return;
   const uint32_t row_hi = pre[n_items];
   if (row_hi > io.trows_cap) {   // the host sized the buffer from an upper bound: cannot happen, but never write past it
      if (threadIdx.x == 0 && blockIdx.x == 0) *io.overflow = 1u;
      return;
   }
   const int imPitch = io.image.pitch, width = io.image.cols - 1, height = io.image.rows - 1;
   const uint32_t ntasks = (row_hi + HS_LARGE_CHUNK - 1) / HS_LARGE_CHUNK;
   for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
      uint32_t row = task * HS_LARGE_CHUNK;
      const uint32_t row_end = min(row + HS_LARGE_CHUNK, row_hi);
      // item of the first row: largest kk in [0, n_items) with pre[kk] <= row (one search per task)
      uint32_t lo = 0, hi = n_items;
      while (hi - lo > 1) {
         const uint32_t mid = (lo + hi) >> 1;
         if (pre[mid] <= row) lo = mid; else hi = mid;
      }
      uint32_t it = lo;
      while (row < row_end) {
         const uint32_t it_rows_end = min(pre[it + 1], row_end);
         const uint32_t h = pw.bin_items[(size_t)(HS_NBINS - 1) * pw.cap + it];
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
                             io.trows + (size_t)row * HS_NEED, outside);
         if (outside) pw.alive[h] = 0;   // every writer stores the same value
         it++;
      }
   }
}

__global__ __launch_bounds__(256) void k_patch_large_finish(PatchWork pw, PatchIO io, KpTables tb)
{
   __shared__ float s_patch[HS_PATCH_ARR];
   const uint32_t *pre = io.row_prefix;
   const uint32_t n_items = min(pw.bin_count[HS_NBINS - 1], pw.cap);
   for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
      const uint32_t h = pw.bin_items[(size_t)(HS_NBINS - 1) * pw.cap + it];
      if (!pw.alive[h]) continue;   // uniform for the block
      const int P0 = pw.P0[h], P = P0 + 2;
      const float scale = (float)P0 / (float)HS_PATCH;
      const int K = tb.patch_tap_k[(P0 - 1) >> 1];
      const float *taps = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
      hs_resample_reduced_batched<false>(io.trows + (size_t)pre[it] * HS_NEED, P, scale, taps, K >> 1, s_patch);
      __syncthreads();
      for (int i = threadIdx.x; i < HS_PATCH_PIX; i += 256) io.patches[(size_t)(h - io.h_base) * HS_PATCH_PIX + i] = s_patch[i];
      __syncthreads();
   }
}

// Exclusive prefix of the window sides P over the items of the large bin (row ids of their T' rows); one block.
// pre[0..n] (pre[n] = total rows).  n is read from the device-side bin counter: no host round trip.
__global__ __launch_bounds__(256) void k_large_prefix(PatchWork pw, uint32_t *__restrict__ pre)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t n = min(pw.bin_count[HS_NBINS - 1], pw.cap);
   const uint32_t *items = pw.bin_items + (size_t)(HS_NBINS - 1) * pw.cap;
   uint32_t carry = 0;
   for (uint32_t base = 0; base < n; base += 256) {
      const uint32_t i = base + threadIdx.x;
      const uint32_t v = (i < n) ? (uint32_t)(pw.P0[items[i]] + 2) : 0u;
      uint32_t tot;
      const uint32_t ex = hs_block_exclusive_scan(v, s_wave, tot);
      if (i < n) pre[i] = carry + ex;
      carry += tot;
   }
   if (threadIdx.x == 0) pre[n] = carry;
}
