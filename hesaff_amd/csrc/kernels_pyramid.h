// kernels_pyramid.h -- scale-space kernels: grey conversion, separable Gaussian with the
// fused det-of-Hessian response and 2x decimation, extrema scan, localisation, ordering.
// Reference: pyramid.cpp, helpers.cpp:283-295,331-339, hesaff.cpp:138-148.
#pragma once
#include "device_common.h"

// ---------------------------------------------------------------------------------------
// k_gray: 8-bit (1 or 3 interleaved channels) -> float32 grey, hesaff.cpp:145
//   out = ((float(c0) + c1) + c2) / 3.0f   (channels == 1: c0 = c1 = c2, like cv::imread)
// grid (ceil(cols/256), rows, B)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gray(const uint8_t *__restrict__ src, int channels, long long src_img_stride,
                                              int src_row_stride, DPlane dst)
{
   const int c = blockIdx.x * 256 + threadIdx.x;
   const int r = blockIdx.y, b = blockIdx.z;
   if (c >= dst.cols) return;
   const uint8_t *p = src + (long long)b * src_img_stride + (long long)r * src_row_stride + (long long)c * channels;
   const float c0 = (float)p[0];
   const float c1 = (float)(channels == 3 ? p[1] : p[0]);
   const float c2 = (float)(channels == 3 ? p[2] : p[0]);
   dst.img(b)[(long long)r * dst.pitch + c] = (c0 + c1 + c2) / 3.0f;
}

// ---------------------------------------------------------------------------------------
// k_blur_hess_tile (LDS tile; the fallback for non-default initialSigma, i.e. tap counts K other
// than 9, 11, 13, 15, which k_blur_hess_march is instantiated for): separable Gaussian (pinned cv::GaussianBlur order, see
// DESIGN.md) of one 64x16 tile + 1-pixel halo, then the det-of-Hessian response of the
// blurred tile (pyramid.cpp:63-114) and optionally the 2x decimated copy
// (helpers.cpp:331-339).
//   row pass   : t = k[0]*S[x-r]; t += k[j]*S[x-r+j], j = 1..K-1            (RowFilter, K > 5)
//                S0*k0 + (S-1+S1)*k1 + (S-2+S2)*k2                            (SymmRowSmallFilter, K <= 5)
//   column pass: d = k[r]*T[y];  d += k[r+j]*(T[y+j] + T[y-j]), j = 1..r     (SymmColumnFilter)
// K <= 15 (pyramid sigmas give 9..15).  grid (ceil(cols/64), ceil(rows/16), B), block 256.
// ---------------------------------------------------------------------------------------
#define BH_TW 64
#define BH_TH 16
#define BH_RMAX 7
#define BH_INW (BH_TW + 2 + 2 * BH_RMAX)   // 80
#define BH_INH (BH_TH + 2 + 2 * BH_RMAX)   // 32

template <bool WRITE_L, bool WRITE_R, bool WRITE_HALF>
__global__ __launch_bounds__(256) void k_blur_hess_tile(DPlane in, DPlane outL, DPlane outR, DPlane outHalf,
                                                         const float *__restrict__ taps, int K, float norm2)
{
   __shared__ float s_in[BH_INH][BH_INW + 1];
   __shared__ float s_row[BH_INH][BH_TW + 2 + 1];
   __shared__ float s_blur[BH_TH + 2][BH_TW + 2 + 1];
   __shared__ float s_k[2 * BH_RMAX + 1];

   const int tid = threadIdx.x;
   const int b = blockIdx.z;
   const int x0 = blockIdx.x * BH_TW, y0 = blockIdx.y * BH_TH;
   const int r = K >> 1;
   const int rows = in.rows, cols = in.cols;
   const float *src = in.img(b);

   if (tid < K) s_k[tid] = taps[tid];
   const int inW = BH_TW + 2 + 2 * r, inH = BH_TH + 2 + 2 * r;
   for (int idx = tid; idx < inW * inH; idx += 256) {
      const int ly = idx / inW, lx = idx - ly * inW;
      int gy = y0 - 1 - r + ly, gx = x0 - 1 - r + lx;
      gy = min(max(gy, 0), rows - 1);
      gx = min(max(gx, 0), cols - 1);
      s_in[ly][lx] = src[(long long)gy * in.pitch + gx];
   }
   __syncthreads();
   // row pass over inH rows x (TW+2) columns
   for (int idx = tid; idx < inH * (BH_TW + 2); idx += 256) {
      const int ly = idx / (BH_TW + 2), ox = idx - ly * (BH_TW + 2);
      float t;
      if (K == 1) t = s_in[ly][ox];   // ksize 1: cv::GaussianBlur copies
      else if (K <= 5) {
         // SymmRowSmallFilter (ksize 3 / 5): S0*k0 + (S-1 + S1)*k1 [+ (S-2 + S2)*k2]
         const float *sp = &s_in[ly][ox + r];
         t = sp[0] * s_k[r] + (sp[-1] + sp[1]) * s_k[r + 1];
         if (K == 5) t = t + (sp[-2] + sp[2]) * s_k[r + 2];
      } else {
         t = s_k[0] * s_in[ly][ox];
         for (int j = 1; j < K; j++) t += s_k[j] * s_in[ly][ox + j];
      }
      s_row[ly][ox] = t;
   }
   __syncthreads();
   // column pass over (TH+2) x (TW+2)
   for (int idx = tid; idx < (BH_TH + 2) * (BH_TW + 2); idx += 256) {
      const int oy = idx / (BH_TW + 2), ox = idx - oy * (BH_TW + 2);
      float d = s_k[r] * s_row[oy + r][ox];
      for (int j = 1; j <= r; j++) d += s_k[r + j] * (s_row[oy + r + j][ox] + s_row[oy + r - j][ox]);
      s_blur[oy][ox] = d;
   }
   __syncthreads();
   // outputs: 64x16 interior
   for (int idx = tid; idx < BH_TH * BH_TW; idx += 256) {
      const int ty = idx / BH_TW, tx = idx - ty * BH_TW;
      const int y = y0 + ty, x = x0 + tx;
      if (y >= rows || x >= cols) continue;
      const int oy = ty + 1, ox = tx + 1;
      const float v22 = s_blur[oy][ox];
      if (WRITE_L) outL.img(b)[(long long)y * outL.pitch + x] = v22;
      if (WRITE_HALF) {
         if (((y | x) & 1) == 0 && (y >> 1) < outHalf.rows && (x >> 1) < outHalf.cols)
            outHalf.img(b)[(long long)(y >> 1) * outHalf.pitch + (x >> 1)] = v22;
      }
      if (WRITE_R) {
         float resp = 0.0f;
         if (y > 0 && y < rows - 1 && x > 0 && x < cols - 1)
            resp = hs_hessian(s_blur[oy - 1][ox - 1], s_blur[oy - 1][ox], s_blur[oy - 1][ox + 1], s_blur[oy][ox - 1], v22,
                              s_blur[oy][ox + 1], s_blur[oy + 1][ox - 1], s_blur[oy + 1][ox], s_blur[oy + 1][ox + 1], norm2);
         outR.img(b)[(long long)y * outR.pitch + x] = resp;
      }
   }
}

// ---------------------------------------------------------------------------------------
// k_hess: det-of-Hessian of a plane (R0 of every octave, pyramid.cpp:230), frame = 0.
// grid (ceil(cols/256), rows, B)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hess(DPlane in, DPlane out, float norm2)
{
   const int x = blockIdx.x * 256 + threadIdx.x;
   const int y = blockIdx.y, b = blockIdx.z;
   if (x >= in.cols) return;
   float resp = 0.0f;
   if (y > 0 && y < in.rows - 1 && x > 0 && x < in.cols - 1) {
      const float *p = in.img(b) + (long long)y * in.pitch + x;
      const int s = in.pitch;
      resp = hs_hessian(p[-s - 1], p[-s], p[-s + 1], p[-1], p[0], p[1], p[s - 1], p[s], p[s + 1], norm2);
   }
   out.img(b)[(long long)y * out.pitch + x] = resp;
}

// k_double: doubleImage helpers.cpp:297-329, the 2x bilinear up-sampling of the upscaleInputImage path
// (pyramid.cpp:267-271).  The reference's loops index the source with its BYTE stride (in[input.step]) and leave the
// last row / column of the result unwritten; this is the evident intent with the same float expressions:
//    n(2r,   2c)   = in(r, c)
//    n(2r+1, 2c)   = 0.5f  * (in(r, c) + in(r+1, c))
//    n(2r,   2c+1) = 0.5f  * (in(r, c) + in(r, c+1))
//    n(2r+1, 2c+1) = 0.25f * (((in(r, c) + in(r, c+1)) + in(r+1, c)) + in(r+1, c+1))
// with r+1 / c+1 clamped to the last row / column (replicated edge).  grid (ceil(outcols/256), outrows, B)
__global__ __launch_bounds__(256) void k_double(DPlane in, DPlane out)
{
   const int x = blockIdx.x * 256 + threadIdx.x;
   const int y = blockIdx.y, b = blockIdx.z;
   if (x >= out.cols) return;
   const int r = y >> 1, c = x >> 1, r1 = min(r + 1, in.rows - 1), c1 = min(c + 1, in.cols - 1);
   const float *p = in.img(b);
   const float v00 = p[(long long)r * in.pitch + c], v01 = p[(long long)r * in.pitch + c1];
   const float v10 = p[(long long)r1 * in.pitch + c], v11 = p[(long long)r1 * in.pitch + c1];
   float v;
   if ((y & 1) == 0) v = (x & 1) == 0 ? v00 : 0.5f * (v00 + v01);
   else v = (x & 1) == 0 ? 0.5f * (v00 + v10) : 0.25f * (v00 + v01 + v10 + v11);
   out.img(b)[(long long)y * out.pitch + x] = v;
}

// k_half: halfImage helpers.cpp:331-339 ; grid (ceil(outcols/256), outrows, B)
__global__ __launch_bounds__(256) void k_half(DPlane in, DPlane out)
{
   const int x = blockIdx.x * 256 + threadIdx.x;
   const int y = blockIdx.y, b = blockIdx.z;
   if (x >= out.cols) return;
   out.img(b)[(long long)y * out.pitch + x] = in.img(b)[(long long)(2 * y) * in.pitch + 2 * x];
}

// ---------------------------------------------------------------------------------------
// Generic two-pass Gaussian for arbitrary K (stage API gaussianBlur; not on the batch path).
// K <= 5 uses the SymmRowSmallFilter order (see kernels_patch.h for the same rule).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur_rows_generic(DPlane in, DPlane tmp, const float *__restrict__ taps, int K)
{
   const int x = blockIdx.x * 256 + threadIdx.x;
   const int y = blockIdx.y, b = blockIdx.z;
   if (x >= in.cols) return;
   const float *S = in.img(b) + (long long)y * in.pitch;
   const int r = K >> 1, cm = in.cols - 1;
   float t;
   if (K <= 5) {
      t = S[x] * taps[r] + (S[max(x - 1, 0)] + S[min(x + 1, cm)]) * taps[r + 1];
      if (K == 5) t = t + (S[max(x - 2, 0)] + S[min(x + 2, cm)]) * taps[r + 2];
   } else {
      t = taps[0] * S[min(max(x - r, 0), cm)];
      for (int j = 1; j < K; j++) t += taps[j] * S[min(max(x - r + j, 0), cm)];
   }
   tmp.img(b)[(long long)y * tmp.pitch + x] = t;
}
__global__ __launch_bounds__(256) void k_blur_cols_generic(DPlane tmp, DPlane out, const float *__restrict__ taps, int K)
{
   const int x = blockIdx.x * 256 + threadIdx.x;
   const int y = blockIdx.y, b = blockIdx.z;
   if (x >= tmp.cols) return;
   const float *T = tmp.img(b);
   const int r = K >> 1, rm = tmp.rows - 1;
   float d = taps[r] * T[(long long)y * tmp.pitch + x];
   for (int j = 1; j <= r; j++)
      d += taps[r + j] * (T[(long long)min(y + j, rm) * tmp.pitch + x] + T[(long long)max(y - j, 0) * tmp.pitch + x]);
   out.img(b)[(long long)y * out.pitch + x] = d;
}

// value of the lane below / above (wavefront shift by one lane: one VALU move with a DPP control, no trip through the LDS crossbar);
// lane 0 / lane 63 keep their own value
__device__ __forceinline__ float hs_from_lane_below(float v)
{
   return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float hs_from_lane_above(float v)
{
   return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// Candidates of one octave: findLevelKeypoints pyramid.cpp:206-222 appends (img,level,r,c) for every pixel that
// passes isMax / isMin (:39-61); unordered here, the order is restored by the bitmask ranks in k_scatter_ordered.
// A candidate carries the 19 response values localizeKeypoint's first iteration reads (pyramid.cpp:132-150): k_extrema_march has
// them in registers when it finds the extremum, and 89 % of the candidates converge in that iteration - so k_localize reads one
// 96-byte record per candidate instead of nine scattered cache lines of three response planes (33 M candidates per 256 UHD
// images: 3 GB written + 3 GB read instead of 19-38 GB of line fetches).
struct CandRec {
   uint32_t id0, id1;   // img<<2 | level, r<<16 | c;  id0 == HS_CAND_HOLE: an unused slot of a wavefront's block of 64
   float v[19];         // cur 3x3 row-major (rows r-1..r+1, columns c-1..c+1) | low: centre, left, right, up, down | high: the same five
   uint32_t pad[3];
};
#define HS_CAND_HOLE 0xffffffffu
#ifndef HS_CAND_PAYLOAD
#define HS_CAND_PAYLOAD 1   // 0 (A/B only): candidates carry their id alone and k_localize reads every neighbourhood from the planes
#endif
#ifndef HS_CAND_BLOCK
#define HS_CAND_BLOCK 64u   // slots a wavefront reserves at a time (one global atomic per 64 candidates)
#endif
struct CandList {
   uint32_t *count;   // device counter (slots handed out, holes included)
   CandRec *items;
   uint32_t cap;
   uint32_t *overflow;   // set to 1 when a list ran out of capacity
};

// ---------------------------------------------------------------------------------------
// k_localize: localizeKeypoint pyramid.cpp:122-204 (without the order-dependent octaveMap
// test, which k_dedupe applies afterwards) + getHessianPointType pyramid.cpp:24-37.
// Grid-stride over the octave's candidates.  A surviving candidate becomes a record and
// bids for its final pixel with atomicMin(map[r,c], order key): the reference's
// first-come-first-kept octaveMap rule (pyramid.cpp:189-193) == lowest (level,r0,c0) wins.
// ---------------------------------------------------------------------------------------
struct OctaveCtx {
   DPlane R[HS_NSCALES + 2];   // responses R0..R4
   DPlane L[HS_NSCALES + 2];   // blurs L0..L3 (L[4] unused)
   float sigma[HS_NSCALES + 2];   // curSigma of each level (sigma[1..3] used as curScale)
   float pixelDistance;
   int octave;
   uint32_t *map;                 // [B][rows][cols] order keys of the CURRENT epoch, anything larger = free
   // Every pass over an octave bids with keys of its own epoch in the bits above the key, and epochs count DOWN: whatever earlier passes (other
   // octaves, earlier batches) left in a cell is larger than any bid of this pass and loses the atomicMin - nothing has to be reset between
   // passes (rounds 1-3 filled the whole map per octave, round 4 had every record give its cell back: 1.1 ms per 256 UHD images).  The host
   // refills the map with 0xFFFFFFFF when the epochs run out (127 passes for UHD: key < 3 x 8.3 M needs 25 bits).
   uint32_t map_epoch;            // epoch << key bits
   long long word_base;           // first bitmask word of this octave inside one image
   long long words_per_image;     // bitmask words of a whole image (all octaves)
   int words_per_row;
};

struct RecList {   // surviving localisations of the whole batch (unordered)
   uint32_t *count;
   uint32_t cap;
   float *x, *y, *s, *response;
   int32_t *meta;       // img<<8 | octave<<4 | level<<2 | type
   uint32_t *cell;      // final pixel index rf*cols+cf inside the octave plane (for the map)
   uint32_t *key;       // level*N + r0*cols + c0
   long long *word;     // bitmask word index of (img,octave,level,r0,c0>>6)
   uint32_t *bit;       // c0 & 63
};

__global__ __launch_bounds__(256) void k_localize(OctaveCtx oc, CandList cl, RecList rl, DConsts k)
{
   const uint32_t n = min(*cl.count, cl.cap);
   if (blockIdx.x == 0 && threadIdx.x == 0 && *cl.count > cl.cap) *cl.overflow = 1u;
   const int rows = oc.R[0].rows, cols = oc.R[0].cols, pitch = oc.R[0].pitch;
   __shared__ uint32_t s_cnt, s_base;
   // uniform trip count: the survivors of a round take their record slots with ONE global atomic per block (a returning atomic
   // per wavefront on one address - 100 k of them per 32 UHD images - was most of this kernel's time)
   for (uint32_t ci0 = blockIdx.x * blockDim.x; ci0 < n; ci0 += gridDim.x * blockDim.x) {
      const uint32_t ci = ci0 + threadIdx.x;
      if (threadIdx.x == 0) s_cnt = 0;
      __syncthreads();
      bool keep = false;
      float o_x = 0, o_y = 0, o_s = 0, o_val = 0;
      int o_meta = 0;
      uint32_t o_cell = 0, o_key = 0, o_bit = 0, o_map = 0, rank = 0;
      long long o_word = 0;
      do {
      if (ci >= n) break;
      const float4 *rec = reinterpret_cast<const float4 *>(cl.items + ci);
      const float4 q0 = rec[0];
      const uint32_t id0 = __float_as_uint(q0.x), id1 = __float_as_uint(q0.y);
      if (id0 == HS_CAND_HOLE) break;
#if HS_CAND_PAYLOAD
      const float4 q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5];
#else
      const float4 q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0;
#endif
      const int b = (int)(id0 >> 2), level = (int)(id0 & 3u);
      const int r0 = (int)(id1 >> 16), c0 = (int)(id1 & 0xffffu);
      // level = i-2 : low = R[level], cur = R[level+1], high = R[level+2]
      const float *low = oc.R[level].img(b), *cur = oc.R[level + 1].img(b), *high = oc.R[level + 2].img(b);
      float bb[3] = {0.0f, 0.0f, 0.0f};
      float val = 0.0f;
      int r = r0, c = c0, nr = r0, nc = c0;
      bool dead = false;
      // the neighbourhood of the current centre: cur 3x3 (m = row - 1, centre, row + 1), low / high crosses
      float m00 = q0.z, m01 = q0.w, m02 = q1.x, c10 = q1.y, c00 = q1.z, c12 = q1.w, p20 = q2.x, p21 = q2.y, p22 = q2.z;
      float lc = q2.w, ll = q3.x, lr = q3.y, lu = q3.z, ld = q3.w;
      float hc = q4.x, hl = q4.y, hr = q4.z, hu = q4.w, hd = q5.x;
      for (int iter = 0; iter < 5; iter++) {
         r = nr; c = nc;
         if (iter > 0 || !HS_CAND_PAYLOAD) {   // the centre moved (one candidate in nine): its neighbourhood comes from the planes
            const float *pc = cur + (long long)r * pitch + c;
            const float *pl = low + (long long)r * pitch + c;
            const float *ph = high + (long long)r * pitch + c;
            m00 = pc[-pitch - 1]; m01 = pc[-pitch]; m02 = pc[-pitch + 1];
            c10 = pc[-1]; c00 = pc[0]; c12 = pc[1];
            p20 = pc[pitch - 1]; p21 = pc[pitch]; p22 = pc[pitch + 1];
            lc = pl[0]; ll = pl[-1]; lr = pl[1]; lu = pl[-pitch]; ld = pl[pitch];
            hc = ph[0]; hl = ph[-1]; hr = ph[1]; hu = ph[-pitch]; hd = ph[pitch];
         }
         const float dxx = c10 - 2.0f * c00 + c12;
         const float dyy = m01 - 2.0f * c00 + p21;
         const float dss = lc - 2.0f * c00 + hc;
         const float dxy = 0.25f * (p22 - p20 - m02 + m00);
         if (iter == 0) {
            const float edgeScore = (dxx + dyy) * (dxx + dyy) / (dxx * dyy - dxy * dxy);
            if (edgeScore >= k.edgeScoreThreshold || edgeScore < 0) { dead = true; break; }
         }
         const float dxs = 0.25f * (hr - hl - lr + ll);
         const float dys = 0.25f * (hd - hu - ld + lu);
         float A[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss};
         const float dx = 0.5f * (c12 - c10);
         const float dy = 0.5f * (p21 - m01);
         const float ds = 0.5f * (hc - lc);
         bb[0] = -dx; bb[1] = -dy; bb[2] = -ds;
         hs_solve3x3(A, bb);
         if (bb[0] != bb[0] || bb[1] != bb[1] || bb[2] != bb[2]) { dead = true; break; }
         val = c00 + 0.5f * (dx * bb[0] + dy * bb[1] + ds * bb[2]);
         // MAX_SUBPIXEL_SHIFT is the double 0.6 (pyramid.cpp:117): compare in double
         if ((double)bb[0] > 0.6) { if (c < cols - 3) nc++; else { dead = true; break; } }
         if ((double)bb[1] > 0.6) { if (r < rows - 3) nr++; else { dead = true; break; } }
         if ((double)bb[0] < -0.6) { if (c > 3) nc--; else { dead = true; break; } }
         if ((double)bb[1] < -0.6) { if (r > 3) nr--; else { dead = true; break; } }
         if (nr == r && nc == c) break;
      }
      if (dead) break;
      if (fabsf(bb[0]) > 1.5f || fabsf(bb[1]) > 1.5f || fabsf(bb[2]) > 1.5f || fabsf(val) < k.finalThreshold) break;
      const float curScale = oc.sigma[level + 1];
      const float scale = curScale * hm_pow2f(bb[2] / (float)HS_NSCALES);
      int type;
      if (val < 0) type = 2;
      else {
         const float *p = oc.L[level + 1].img(b) + (long long)r * oc.L[level + 1].pitch + c;
         const float Lxx = (p[-1] - 2 * p[0] + p[1]);
         type = (Lxx < 0) ? 0 : 1;
      }
      const float pd = oc.pixelDistance;
      o_x = pd * ((float)c + bb[0]);
      o_y = pd * ((float)r + bb[1]);
      o_s = pd * scale;
      o_val = val;
      o_meta = (b << 8) | (oc.octave << 4) | (level << 2) | type;
      o_cell = (uint32_t)(r * cols + c);
      o_key = (uint32_t)level * (uint32_t)(rows * cols) + (uint32_t)(r0 * cols + c0);
      o_word = (long long)b * oc.words_per_image + oc.word_base + ((long long)level * rows + r0) * oc.words_per_row + (c0 >> 6);
      o_bit = (uint32_t)(c0 & 63);
      o_map = (uint32_t)b;
      keep = true;
      } while (false);
      if (keep) rank = atomicAdd(&s_cnt, 1u);
      __syncthreads();
      if (threadIdx.x == 0 && s_cnt > 0) s_base = atomicAdd(rl.count, s_cnt);
      __syncthreads();
      if (keep) {
         const uint32_t slot = s_base + rank;
         if (slot >= rl.cap) { *cl.overflow = 1u; }
         else {
            rl.x[slot] = o_x;
            rl.y[slot] = o_y;
            rl.s[slot] = o_s;
            rl.response[slot] = o_val;
            rl.meta[slot] = o_meta;
            rl.cell[slot] = o_cell;
            rl.key[slot] = o_key;
            rl.word[slot] = o_word;
            rl.bit[slot] = o_bit;
            atomicMin(oc.map + (long long)o_map * rows * cols + o_cell, o_key | oc.map_epoch);   // (the epoch: OctaveCtx)
         }
      }
   }
}

// k_dedupe: records [start, count) of this octave; the map winner sets its bitmask bit.
__global__ __launch_bounds__(256) void k_dedupe(OctaveCtx oc, RecList rl, const uint32_t *__restrict__ start_ptr,
                                                unsigned long long *__restrict__ bitmask)
{
   const uint32_t start = *start_ptr;
   const uint32_t n = min(*rl.count, rl.cap);
   const long long N = (long long)oc.R[0].rows * oc.R[0].cols;
   for (uint32_t i = start + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int b = rl.meta[i] >> 8;
      if (oc.map[(long long)b * N + rl.cell[i]] == (rl.key[i] | oc.map_epoch)) atomicOr(bitmask + rl.word[i], 1ull << rl.bit[i]);
      else rl.word[i] = -1;   // lost the octaveMap race
   }
}


// ---------------------------------------------------------------------------------------
// Exclusive scan of 32-bit counts (three phases, 4096 items per block).  LOAD turns the
// i-th source element into its count (popcount of a bitmask word, or a flag).
// ---------------------------------------------------------------------------------------
#define SCAN_ITEMS 16
#define SCAN_BLOCK (256 * SCAN_ITEMS)

// load16: a thread's SCAN_ITEMS consecutive items (base a multiple of 16, all inside the array) as 16-byte loads
struct LoadPopc {
   const unsigned long long *p;
   __device__ uint32_t operator()(long long i) const { return (uint32_t)__popcll(p[i]); }
   __device__ void load16(long long base, uint32_t *v) const
   {
      const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p + base);
#pragma unroll
      for (int i = 0; i < 8; i++) { const ulonglong2 w = q[i]; v[2 * i] = (uint32_t)__popcll(w.x); v[2 * i + 1] = (uint32_t)__popcll(w.y); }
   }
};
struct LoadU32 {
   const uint32_t *p;
   __device__ uint32_t operator()(long long i) const { return p[i]; }
   __device__ void load16(long long base, uint32_t *v) const
   {
      const uint4 *q = reinterpret_cast<const uint4 *>(p + base);
#pragma unroll
      for (int i = 0; i < 4; i++) { const uint4 w = q[i]; v[4 * i] = w.x; v[4 * i + 1] = w.y; v[4 * i + 2] = w.z; v[4 * i + 3] = w.w; }
   }
};
struct LoadFlagI32 {
   const int32_t *p;
   __device__ uint32_t operator()(long long i) const { return p[i] != 0 ? 1u : 0u; }
   __device__ void load16(long long base, uint32_t *v) const
   {
      const int4 *q = reinterpret_cast<const int4 *>(p + base);
#pragma unroll
      for (int i = 0; i < 4; i++) { const int4 w = q[i]; v[4 * i] = w.x != 0; v[4 * i + 1] = w.y != 0; v[4 * i + 2] = w.z != 0; v[4 * i + 3] = w.w != 0; }
   }
};

__device__ __forceinline__ uint32_t hs_block_exclusive_scan(uint32_t v, uint32_t *s_wave /*4*/, uint32_t &block_total)
{
   // inclusive scan inside the wave (integers: order-free)
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   uint32_t inc = v;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
   }
   if (lane == 63) s_wave[w] = inc;
   __syncthreads();
   uint32_t base = 0, tot = 0;
#pragma unroll
   for (int i = 0; i < 4; i++) {
      const uint32_t t = s_wave[i];
      if (i < w) base += t;
      tot += t;
   }
   block_total = tot;
   __syncthreads();
   return base + inc - v;
}

template <class LOAD>
__global__ __launch_bounds__(256) void k_scan_reduce(LOAD load, long long n, uint32_t *__restrict__ block_sums)
{
   __shared__ uint32_t s_wave[4];
   const long long base = (long long)blockIdx.x * SCAN_BLOCK;
   uint32_t v = 0;
#pragma unroll
   for (int i = 0; i < SCAN_ITEMS; i++) {
      const long long idx = base + (long long)i * 256 + threadIdx.x;
      if (idx < n) v += load(idx);
   }
   uint32_t tot;
   hs_block_exclusive_scan(v, s_wave, tot);
   if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums in place; total -> *total_out
__global__ __launch_bounds__(256) void k_scan_sums(uint32_t *__restrict__ block_sums, int nblocks, uint32_t *__restrict__ total_out)
{
   __shared__ uint32_t s_wave[4];
   uint32_t carry = 0;
   for (int base = 0; base < nblocks; base += 256) {
      const int i = base + threadIdx.x;
      const uint32_t v = (i < nblocks) ? block_sums[i] : 0u;
      uint32_t tot;
      const uint32_t ex = hs_block_exclusive_scan(v, s_wave, tot);
      if (i < nblocks) block_sums[i] = carry + ex;
      carry += tot;
   }
   if (threadIdx.x == 0) *total_out = carry;
}

template <class LOAD>
__global__ __launch_bounds__(256) void k_scan_down(LOAD load, long long n, const uint32_t *__restrict__ block_sums,
                                                   uint32_t *__restrict__ out)
{
   __shared__ uint32_t s_wave[4];
   // thread owns SCAN_ITEMS consecutive items so that the scan is in index order
   const long long base = (long long)blockIdx.x * SCAN_BLOCK + (long long)threadIdx.x * SCAN_ITEMS;
   static_assert(SCAN_ITEMS == 16, "load16 / the 16-byte stores below");
   uint32_t vals[SCAN_ITEMS];
   uint32_t sum = 0;
   // (base is a multiple of 16 items; the source may start anywhere - the alive flags sit behind another array - so its alignment is tested)
   const bool whole = base + SCAN_ITEMS <= n && (reinterpret_cast<uintptr_t>(load.p + base) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out + base) & 15u) == 0;
   if (whole) load.load16(base, vals);          // 128 (64) consecutive bytes per thread as 16-byte loads; item by item it was 16 loads of 8 (4) bytes, every one a line of its own
#pragma unroll
   for (int i = 0; i < SCAN_ITEMS; i++) {
      const long long idx = base + i;
      if (!whole) vals[i] = (idx < n) ? load(idx) : 0u;
      sum += vals[i];
   }
   uint32_t tot;
   uint32_t ex = hs_block_exclusive_scan(sum, s_wave, tot) + block_sums[blockIdx.x];
   if (whole) {
      uint4 *o = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
      for (int i = 0; i < SCAN_ITEMS; i += 4) {
         uint4 w;
         w.x = ex; ex += vals[i]; w.y = ex; ex += vals[i + 1]; w.z = ex; ex += vals[i + 2]; w.w = ex; ex += vals[i + 3];
         o[i >> 2] = w;
      }
   } else {
#pragma unroll
      for (int i = 0; i < SCAN_ITEMS; i++) {
         const long long idx = base + i;
         if (idx < n) out[idx] = ex;
         ex += vals[i];
      }
   }
}

// ---------------------------------------------------------------------------------------
// k_scatter_ordered: rank = prefix[word] + popcount(bits below) puts every surviving record
// at its position in the reference's detection order (image, octave, level, raster of the
// initial extremum).
// ---------------------------------------------------------------------------------------
struct HessList {   // ordered Hessian keypoints of the batch = onHessianKeypointDetected calls
   float *x, *y, *s, *response;
   int32_t *meta;    // img<<8 | octave<<4 | level<<2 | type
   int32_t *r0c0;    // r0<<16 | c0 (provenance, for the stage API)
   uint32_t cap;
};

// Two launches: the record goes to its rank as ONE 32-byte item (the six fields side by side) and a second, streaming kernel deals the items out to
// the six arrays of the Hessian list.  Scattering the six fields themselves (rounds 1-4) was six partial-line stores per record, 33 M records per
// 256 UHD images: 2.7 ms; a 32-byte store is one whole sector, and the second pass is coalesced on both sides.
struct __attribute__((aligned(32))) HessItem { float x, y, s, response; int32_t meta, r0c0; uint32_t pad0, pad1; };
__global__ __launch_bounds__(256) void k_scatter_ordered(RecList rl, const unsigned long long *__restrict__ bitmask,
                                                         const uint32_t *__restrict__ prefix, HessItem *__restrict__ items, uint32_t cap)
{
   const uint32_t n = min(*rl.count, rl.cap);
   for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const long long w = rl.word[i];
      if (w < 0) continue;
      const uint32_t bit = rl.bit[i];
      const unsigned long long below = bitmask[w] & ((1ull << bit) - 1ull);
      const uint32_t rank = prefix[w] + (uint32_t)__popcll(below);
      if (rank >= cap) continue;
      float4 *o = reinterpret_cast<float4 *>(items + rank);
      o[0] = make_float4(rl.x[i], rl.y[i], rl.s[i], rl.response[i]);
      // (r0, c0 is recovered from the key by the stage API; the batch path does not read it)
      o[1] = make_float4(__int_as_float(rl.meta[i]), __int_as_float((int32_t)rl.key[i]), 0.0f, 0.0f);
   }
}
__global__ __launch_bounds__(256) void k_hess_deal(const HessItem *__restrict__ items, const uint32_t *__restrict__ n_ptr, HessList hl)
{
   const uint32_t n = min(*n_ptr, hl.cap);
   for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
      const float4 a = reinterpret_cast<const float4 *>(items + r)[0], b = reinterpret_cast<const float4 *>(items + r)[1];
      hl.x[r] = a.x; hl.y[r] = a.y; hl.s[r] = a.z; hl.response[r] = a.w;
      hl.meta[r] = __float_as_int(b.x); hl.r0c0[r] = __float_as_int(b.y);
   }
}

// per-image counts from an exclusive prefix sampled at image boundaries:
// counts[b] = prefix_at(b+1) - prefix_at(b), the last boundary being *total.
__global__ void k_image_counts(const uint32_t *__restrict__ prefix, long long stride, int nimg, const uint32_t *__restrict__ total,
                               int32_t *__restrict__ starts /*nimg+1*/)
{
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b > nimg) return;
   starts[b] = (b == nimg) ? (int32_t)*total : (int32_t)prefix[(long long)b * stride];
}

// ---------------------------------------------------------------------------------------
// k_blur_hess_march<K,...> (v2): the roofline kernel.  Same contract and the same pinned
// operation order as k_blur_hess_tile, different schedule:
//  * one WAVEFRONT owns a strip of 256 blurred columns (4 per lane; 248 of them are stored,
//    the outer 4+4 are the halo the Hessian needs) and marches down a band of rows;
//  * each input row is read once from HBM/L2 into a per-wave LDS row (coalesced dword loads,
//    replicate border = clamped per-lane column index), the row pass reads its 4+2r inputs
//    back as aligned float4;
//  * the K most recent row-pass results live in a REGISTER ring (statically indexed: the row
//    loop is unrolled K times), so the column pass never touches memory;
//  * the three most recent blurred rows stay in registers for the 3x3 Hessian; horizontal
//    neighbours come from the adjacent lanes (shuffle);
//  * outputs are float4 stores (blur, response) and float2 (decimated next-octave level).
// Reads per output pixel: (256+16)/248 x (HB+2r+2)/HB ~ 1.2x, out of L2; HBM sees each input
// byte about once.  No block-level synchronisation: the 4 waves of a block are independent.
// grid (ceil(strips/4), bands, B), block 256.
// ---------------------------------------------------------------------------------------
#define BM_STRIP 248
#define BM_ROWBUF 272   // floats: 8 + 256 + 8

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f hs_hessian2(v2f ul, v2f uc, v2f ur, v2f ml, v2f mc, v2f mr, v2f dl, v2f dc, v2f dr, float norm2)
{
   // pyramid.cpp:95-100 on two adjacent columns at once (component-wise IEEE mul/add/sub/div)
   const v2f Lxx = (ml - 2.0f * mc) + mr;
   const v2f Lyy = (uc - 2.0f * mc) + dc;
   const v2f Lxy = (((ur - ul) + dl) - dr) / 4.0f;
   return ((Lxx * Lyy) - (Lxy * Lxy)) * norm2;
}

// WRITE_R0: additionally emit the response of the INPUT plane (R0 = hessianResponse(L0),
// pyramid.cpp:230) from the input rows that pass through LDS anyway; used by the first blur
// of every octave so that L0 is read from HBM once instead of twice.
// SRC8: the input rows are taken from the 8-bit source images and converted on the fly (hesaff.cpp:145, k_gray's
// expression); the float grey plane normalizeAffine samples later is written from the same registers.  Used by the
// initial blur (pyramid.cpp:276-280): the image is read once as bytes instead of once as bytes and once as floats.
struct GraySrc {
   const uint8_t *p;
   int channels;             // 1 or 3 interleaved
   long long img_stride;     // bytes between images
   int row_stride;           // bytes between rows
};

#ifndef HS_MARCH15_WAVES
#define HS_MARCH15_WAVES 0   // tuning: wavefronts per SIMD the K = 15 instantiation (144 VGPRs) is held to (0: the compiler's choice, 3)
#endif
template <int K, bool WRITE_L, bool WRITE_R, bool WRITE_HALF, bool WRITE_R0 = false, bool SRC8 = false>
__global__ __launch_bounds__(256, (K == 15 ? HS_MARCH15_WAVES : 0)) void k_blur_hess_march(DPlane in, DPlane outL, DPlane outR, DPlane outHalf,
                                                          const float *__restrict__ taps, float norm2, int band_rows,
                                                          DPlane outR0 = DPlane(), float norm2_in = 0.0f, GraySrc gs = GraySrc(), DPlane outGray = DPlane())
{
   constexpr int R = K >> 1;
   constexpr int U = K + 1;                      // ring size and unroll factor (even: static prefetch parity)
   constexpr int W0 = 8 - R;                     // row-buffer float of column x-R, relative to 4*lane
#ifndef HS_R0_LDS3
#define HS_R0_LDS3 0   // tuning: 1 = the R0 epilogue re-reads its three input rows from LDS instead of carrying two of them in registers
#endif
   constexpr bool R0L3 = WRITE_R0 && HS_R0_LDS3;
   constexpr int NB = R0L3 ? 3 : 2;              // input rows kept in LDS (the row being filtered, the row being staged [, one more])
   __shared__ __attribute__((aligned(16))) float s_rows[4][NB][BM_ROWBUF];

   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int strip = blockIdx.x * 4 + wave;
   const int rows = in.rows, cols = in.cols, pitch = in.pitch;
   const int xs = strip * BM_STRIP;
   if (xs >= cols) return;
   const int b = blockIdx.z;
   const int yh0 = blockIdx.y * band_rows, yh1 = min(yh0 + band_rows, rows);
   const int steps = (yh1 - yh0) + 2 * R + 2;
   const float *src = in.img(b);

   float kk[K];
#pragma unroll
   for (int j = 0; j < K; j++) kk[j] = taps[j];

   // row buffer float f <-> column xs - 12 + f ; this lane loads f = lane + 64 m
   int cx[5];
#pragma unroll
   for (int m = 0; m < 5; m++) cx[m] = min(max(xs - 12 + lane + 64 * m, 0), cols - 1);
   const int xl = xs - 4 + 4 * lane;   // first of this lane's 4 columns
   const bool store_lane = lane >= 1 && lane <= 62 && xl < cols;
   const bool full4 = xl + 3 < cols;
   // response is 0 on the image frame (pyramid.cpp:70: never written there)
   const bool cin0 = xl > 0 && xl < cols - 1, cin1 = xl + 1 > 0 && xl + 1 < cols - 1;
   const bool cin2 = xl + 2 > 0 && xl + 2 < cols - 1, cin3 = xl + 3 > 0 && xl + 3 < cols - 1;

   v2f ringA[U], ringB[U];                       // row-pass results of columns (0,1) and (2,3)
#pragma unroll
   for (int u = 0; u < U; u++) { ringA[u] = (v2f)(0.0f); ringB[u] = (v2f)(0.0f); }
   // two previous blurred rows as the five overlapping column pairs (x-1,x) (x,x+1) ... (x+3,x+4)
   v2f P2[5], P1[5];
#pragma unroll
   for (int i = 0; i < 5; i++) { P2[i] = (v2f)(0.0f); P1[i] = (v2f)(0.0f); }
   // WRITE_R0: the two previous INPUT rows in the same form (registers, like P2 / P1)
   v2f Q0[5], Q1[5];
#pragma unroll
   for (int i = 0; i < 5; i++) { Q0[i] = (v2f)(0.0f); Q1[i] = (v2f)(0.0f); }

   float pre[2][5];
   auto load_row = [&](int t, float *dst5) {
      const int yu = yh0 - 1 - R + t;
      const int y = min(max(yu, 0), rows - 1);
      if (SRC8) {
         const uint8_t *rp = gs.p + (long long)b * gs.img_stride + (long long)y * gs.row_stride;
         if (gs.channels == 1) {
            // grey input: cv::imread replicates the byte into B, G, R and (float(c) + c + c) / 3.0f == float(c) exactly
            // (3c <= 765 is exact, and so is its quotient by 3): no arithmetic needed
#pragma unroll
            for (int m = 0; m < 5; m++) dst5[m] = (float)rp[cx[m]];
         } else {
#pragma unroll
            for (int m = 0; m < 5; m++) {
               const uint8_t *q = rp + (long long)cx[m] * 3;
               dst5[m] = ((float)q[0] + (float)q[1] + (float)q[2]) / 3.0f;   // hesaff.cpp:145
            }
         }
         // the band's own rows x the strip's own columns (row-buffer floats 12 .. 259): each grey pixel is written exactly once
         if (yu >= yh0 && yu < yh1) {
            float *go = outGray.img(b) + (long long)y * outGray.pitch + (xs - 12);
#pragma unroll
            for (int m = 0; m < 5; m++) {
               const int f = lane + 64 * m;
               if (f >= 12 && f < 12 + BM_STRIP && xs - 12 + f < cols) { if (HS_NT_GRAY) hs_store_nt(go + f, dst5[m]); else go[f] = dst5[m]; }
            }
         }
         return;
      }
      const float *rp = src + (long long)y * pitch;
#pragma unroll
      for (int m = 0; m < 4; m++) dst5[m] = rp[cx[m]];
      dst5[4] = rp[cx[4]];   // lanes >= 16 load a clamped in-image column they never store
   };
   auto stage_row = [&](int buf, const float *src5) {
      float *wb = s_rows[wave][buf];
#pragma unroll
      for (int m = 0; m < 4; m++) wb[lane + 64 * m] = src5[m];
      if (lane < 16) wb[lane + 256] = src5[4];
   };
   load_row(0, pre[0]);
   stage_row(0, pre[0]);
   load_row(1, pre[1]);

   for (int t0 = 0; t0 < steps; t0 += U) {
#pragma unroll
      for (int u = 0; u < U; u++) {
         // No control flow in the unrolled body (the compiler otherwise splits the register ring
         // into scalars at every branch and copies it back into pairs): the step count is padded
         // to a multiple of U; padded steps read clamped rows and store nothing.
         const int t = t0 + u;           // t & 1 == u & 1
         load_row(t + 2, pre[u & 1]);    // row t's registers are free (staged during step t-1)
         // ---- row pass of input row t: acc = k[0]*S[x-R]; acc += k[j]*S[x-R+j] (RowFilter order), this lane's four columns ----
         {
            // The K + 3 samples the four chains read come in as the aligned 16-byte quads that cover them (3 or 5 conflict-free
            // ds_read_b128 at a lane stride of 16 bytes; rounds 1-3 read them as K + 2 overlapping pairs at a 2-way bank conflict each:
            // 49 % conflict cycles, the LDS 72 % busy under the 15-tap launch), and the chains run as scalar operations on them.
            // Measured neutral on the launch time (profiles/r04_notes.md: these launches are held by the power limit, the clock
            // falls to 1.6-1.9 GHz under them), kept for the registers (K = 15: 124 instead of 133) and the idle LDS.
            constexpr int QF = (W0 / 4) * 4;                            // first quad, in floats relative to 4 * lane
            constexpr int NQ = (W0 + K + 2) / 4 - W0 / 4 + 1;
            constexpr int O = W0 - QF;                                  // S[O + c + j] = tap j of column c
            const float4 *qb = reinterpret_cast<const float4 *>(__builtin_assume_aligned(s_rows[wave][R0L3 ? (t % 3) : (u & 1)] + 4 * lane + QF, 16));
            float S[4 * NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
               float4 v = qb[q];
               // all four components count as used: the compiler otherwise narrows the first and last quad to the floats the chains
               // read and splits every quad into ds_read2_b32 pairs again
               HS_KEEP(v.x); HS_KEEP(v.y); HS_KEEP(v.z); HS_KEEP(v.w);
               S[4 * q] = v.x; S[4 * q + 1] = v.y; S[4 * q + 2] = v.z; S[4 * q + 3] = v.w;
            }
            float a0 = kk[0] * S[O], a1 = kk[0] * S[O + 1], a2 = kk[0] * S[O + 2], a3 = kk[0] * S[O + 3];
#pragma unroll
            for (int j = 1; j < K; j++) {
               const float p0 = kk[j] * S[O + j], p1 = kk[j] * S[O + 1 + j], p2 = kk[j] * S[O + 2 + j], p3 = kk[j] * S[O + 3 + j];
               a0 += p0; a1 += p1; a2 += p2; a3 += p3;
            }
            v2f a, bq;
            a.x = a0; a.y = a1; bq.x = a2; bq.y = a3;
            ringA[u] = a;
            ringB[u] = bq;
         }
         // ---- column pass: blurred row yl = yh0 - 1 + (t - 2R): d = k[R]*T[y]; d += k[R+j]*(T[y+j]+T[y-j]) ----
         // (for t < 2R the ring is not full yet: yl < yh0, nothing is stored)
         {
            const int yl = yh0 - 1 + (t - 2 * R);
            v2f la = kk[R] * ringA[(u - R + U) % U], lb = kk[R] * ringB[(u - R + U) % U];
#pragma unroll
            for (int j = 1; j <= R; j++) {
               const v2f sa = ringA[(u - R + j + U) % U] + ringA[(u - R - j + 2 * U) % U];
               const v2f sb = ringB[(u - R + j + U) % U] + ringB[(u - R - j + 2 * U) % U];
               const v2f qa = kk[R + j] * sa, qb = kk[R + j] * sb;
               la = la + qa;
               lb = lb + qb;
            }
            const float left = hs_from_lane_below(lb.y);    // column x-1
            const float right = hs_from_lane_above(la.x);   // column x+4
            v2f P0[5];
            P0[0].x = left; P0[0].y = la.x;
            P0[1] = la;
            P0[2].x = la.y; P0[2].y = lb.x;
            P0[3] = lb;
            P0[4].x = lb.y; P0[4].y = right;
            const bool row_in = yl >= yh0 && yl < yh1;
            if (row_in && store_lane) {
               if (WRITE_L) {
                  float *o = outL.img(b) + (long long)yl * outL.pitch + xl;
                  if (full4) { if (HS_NT_PYR) hs_store_nt4(o, la.x, la.y, lb.x, lb.y); else *reinterpret_cast<float4 *>(o) = make_float4(la.x, la.y, lb.x, lb.y); }
                  else {
                     const float v[4] = {la.x, la.y, lb.x, lb.y};
                     for (int c = 0; c < 4; c++)
                        if (xl + c < cols) o[c] = v[c];
                  }
               }
               if (WRITE_HALF) {
                  if ((yl & 1) == 0 && (yl >> 1) < outHalf.rows) {
                     float *o = outHalf.img(b) + (long long)(yl >> 1) * outHalf.pitch + (xl >> 1);
                     if ((xl >> 1) + 1 < outHalf.cols) { if (HS_NT_PYR) hs_store_nt2(o, la.x, lb.x); else *reinterpret_cast<float2 *>(o) = make_float2(la.x, lb.x); }
                     else if ((xl >> 1) < outHalf.cols) o[0] = la.x;
                  }
               }
            }
            // ---- response of row yh = yl - 1 from rows (P2, P1, P0) ----
            if (WRITE_R) {
               const int yh = yl - 1;
               if (yh >= yh0 && yh < yh1 && store_lane) {
                  const bool yin = yh > 0 && yh < rows - 1;
                  const v2f ra = hs_hessian2(P2[0], P2[1], P2[2], P1[0], P1[1], P1[2], P0[0], P0[1], P0[2], norm2);
                  const v2f rbv = hs_hessian2(P2[2], P2[3], P2[4], P1[2], P1[3], P1[4], P0[2], P0[3], P0[4], norm2);
                  const float r0 = (yin && cin0) ? ra.x : 0.0f, r1 = (yin && cin1) ? ra.y : 0.0f;
                  const float r2 = (yin && cin2) ? rbv.x : 0.0f, r3 = (yin && cin3) ? rbv.y : 0.0f;
                  float *o = outR.img(b) + (long long)yh * outR.pitch + xl;
                  if (full4) { if (HS_NT_PYR_R) hs_store_nt4(o, r0, r1, r2, r3); else *reinterpret_cast<float4 *>(o) = make_float4(r0, r1, r2, r3); }
                  else {
                     const float v[4] = {r0, r1, r2, r3};
                     for (int c = 0; c < 4; c++)
                        if (xl + c < cols) o[c] = v[c];
                  }
               }
            }
#pragma unroll
            for (int i = 0; i < 5; i++) { P2[i] = P1[i]; P1[i] = P0[i]; }
         }
         // ---- response of the input plane: row y(t-1) from the input rows t-2, t-1 (registers) and t (LDS) ----
         // Every lane reads its four columns of row t as one aligned float4 (conflict-free); the columns x-1 and x+4 come
         // from the neighbouring lanes by shuffle instead of two more LDS reads at a 16-byte lane stride (4-way conflicts).
         if (WRITE_R0) {
            auto row_pairs = [&](int buf, v2f *Q) {
               const float4 m = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(s_rows[wave][buf] + 4 * lane + 8, 16));
               const float e0 = hs_from_lane_below(m.w), e5 = hs_from_lane_above(m.x);
               Q[0].x = e0; Q[0].y = m.x;
               Q[1].x = m.x; Q[1].y = m.y;
               Q[2].x = m.y; Q[2].y = m.z;
               Q[3].x = m.z; Q[3].y = m.w;
               Q[4].x = m.w; Q[4].y = e5;
            };
            v2f Q2[5];
            if (R0L3) { row_pairs((t + 1) % 3, Q0); row_pairs((t + 2) % 3, Q1); row_pairs(t % 3, Q2); }
            else row_pairs(u & 1, Q2);
            const int yr = yh0 - 1 - R + (t - 1);
            if (yr >= yh0 && yr < yh1 && store_lane) {
               const bool yin = yr > 0 && yr < rows - 1;
               const v2f ra = hs_hessian2(Q0[0], Q0[1], Q0[2], Q1[0], Q1[1], Q1[2], Q2[0], Q2[1], Q2[2], norm2_in);
               const v2f rbv = hs_hessian2(Q0[2], Q0[3], Q0[4], Q1[2], Q1[3], Q1[4], Q2[2], Q2[3], Q2[4], norm2_in);
               const float r0 = (yin && cin0) ? ra.x : 0.0f, r1 = (yin && cin1) ? ra.y : 0.0f;
               const float r2 = (yin && cin2) ? rbv.x : 0.0f, r3 = (yin && cin3) ? rbv.y : 0.0f;
               float *o = outR0.img(b) + (long long)yr * outR0.pitch + xl;
               if (full4) { if (HS_NT_PYR_R) hs_store_nt4(o, r0, r1, r2, r3); else *reinterpret_cast<float4 *>(o) = make_float4(r0, r1, r2, r3); }
               else {
                  const float v[4] = {r0, r1, r2, r3};
                  for (int c = 0; c < 4; c++)
                     if (xl + c < cols) o[c] = v[c];
               }
            }
            if (!R0L3) {
#pragma unroll
               for (int i = 0; i < 5; i++) { Q0[i] = Q1[i]; Q1[i] = Q2[i]; }
            }
         }
         // ---- stage row t+1 (loaded during step t-1) for the next step ----
         stage_row(R0L3 ? ((t + 1) % 3) : ((u + 1) & 1), pre[(u + 1) & 1]);
      }
   }
}

struct FivePlanes { DPlane R[5]; };

// ---------------------------------------------------------------------------------------
// k_extrema_march: the same three extrema scans, HBM-streaming form.  One wavefront marches
// down a strip of 248 columns (lane = 4 adjacent columns, one halo lane per side) over a band
// of rows, each response value is read from HBM exactly once (float4, two rows ahead) and
// lives in a register ring of 5 rows x 5 planes.  The 27-neighbour test is evaluated as a
// separable max / min:
//    "no neighbour strictly greater than val"  <=>  !(max27 > val)
//    max27 = max over x-1..x+1 of ( max over the 3 planes of ( max over rows y-1..y+1 ) )
// (v_max3_f32 / v_min3_f32 skip NaN operands exactly like the comparisons of
// pyramid.cpp:39-61 let a NaN neighbour pass), ~45 VALU operations per pixel for all three
// levels instead of ~400.  Candidates are collected in LDS and flushed with one global atomic.
// grid (ceil(cols/248), ceil(rows/band), B), block 64.
// ---------------------------------------------------------------------------------------
#define EXM_STRIP 248
#ifndef EXM_RS
#define EXM_RS 5      // ring rows: 3 under test + EXM_RS - 3 in flight (tuning; profiles/r05_notes.md)
#endif
#define EXM_AHEAD (EXM_RS - 3)
#ifndef EXM_WAVES
#define EXM_WAVES 0   // tuning: wavefronts per SIMD the register allocation is held to (0: the compiler's choice, 3)
#endif

__device__ __forceinline__ float hs_max3(float a, float b, float c)
{
   float r;
   asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
   return r;
}
__device__ __forceinline__ float hs_min3(float a, float b, float c)
{
   float r;
   asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
   return r;
}

__global__ __launch_bounds__(64, EXM_WAVES) void k_extrema_march(FivePlanes fp, float posThr, float negThr, CandList cl, int band)
{
   const int lane = threadIdx.x, b = blockIdx.z;
   const int rows = fp.R[0].rows, cols = fp.R[0].cols, pitch = fp.R[0].pitch;
   const int ya = max((int)blockIdx.y * band, HS_BORDER), yb = min(((int)blockIdx.y + 1) * band, rows - HS_BORDER);   // scanned rows
   if (ya >= yb) return;
   const int x = (int)blockIdx.x * EXM_STRIP - 4 + 4 * lane;   // my 4 columns; lanes 0 and 63 only feed their neighbours
   // Always a valid aligned 16-byte load: lanes outside the image read some in-range columns whose
   // values can only reach columns that are not scanned (HS_BORDER >= 1 away from the frame).
   const int xc = min(max(x, 0), pitch - 4);
   const bool own = lane >= 1 && lane <= 62;
   uint32_t colmask = 0;   // bit c: column x + c is scanned
#pragma unroll
   for (int c = 0; c < 4; c++)
      if (own && x + c >= HS_BORDER && x + c < cols - HS_BORDER) colmask |= 1u << c;
   const float *base[5];
#pragma unroll
   for (int p = 0; p < 5; p++) base[p] = fp.R[p].img(b) + xc;
   // candidate slots: the wavefront reserves blocks of HS_CAND_BLOCK slots of the octave's list (one global atomic per block) and
   // fills them in the order it finds its candidates; what is left of a block is marked as holes
   uint32_t wbase = 0, wused = HS_CAND_BLOCK;   // wave-uniform: no block yet

   float4 ring[5][EXM_RS];
   const int nsteps = yb - ya + 2;   // step k brings row ya - 1 + k; rows ya .. yb-1 are tested at steps 2 .. nsteps-1
   // prologue: rows of steps 0 .. EXM_AHEAD - 1
#pragma unroll
   for (int k = 0; k < EXM_AHEAD; k++) {
      const long long off = (long long)min(ya - 1 + k, rows - 1) * pitch;
#pragma unroll
      for (int p = 0; p < 5; p++) ring[p][k] = HS_NT_EXT ? hs_load_nt4(reinterpret_cast<const float4 *>(base[p] + off)) : *reinterpret_cast<const float4 *>(base[p] + off);
   }
   for (int k0 = 0; k0 < nsteps; k0 += EXM_RS) {
#pragma unroll
      for (int u = 0; u < EXM_RS; u++) {
         const int k = k0 + u;
         {
            // EXM_AHEAD rows ahead, into the slot whose row (k - 3) is no longer needed
            const long long off = (long long)min(ya - 1 + k + EXM_AHEAD, rows - 1) * pitch;
#pragma unroll
            for (int p = 0; p < 5; p++) ring[p][(u + EXM_AHEAD) % EXM_RS] = HS_NT_EXT ? hs_load_nt4(reinterpret_cast<const float4 *>(base[p] + off)) : *reinterpret_cast<const float4 *>(base[p] + off);
         }
         const int y = ya - 2 + k;        // row under test: slots (u-2, u-1, u) = rows y-1, y, y+1
         const int s0 = (u + EXM_RS - 2) % EXM_RS, s1 = (u + EXM_RS - 1) % EXM_RS, s2 = u;
         float A[3][4], I[3][4];   // per level: max / min over 3 rows x 3 planes, per column
         {
            float vmx[5][4], vmn[5][4];
#pragma unroll
            for (int p = 0; p < 5; p++) {
               const float a[4] = {ring[p][s0].x, ring[p][s0].y, ring[p][s0].z, ring[p][s0].w};
               const float m[4] = {ring[p][s1].x, ring[p][s1].y, ring[p][s1].z, ring[p][s1].w};
               const float z[4] = {ring[p][s2].x, ring[p][s2].y, ring[p][s2].z, ring[p][s2].w};
#pragma unroll
               for (int c = 0; c < 4; c++) { vmx[p][c] = hs_max3(a[c], m[c], z[c]); vmn[p][c] = hs_min3(a[c], m[c], z[c]); }
            }
#pragma unroll
            for (int l = 0; l < 3; l++)
#pragma unroll
               for (int c = 0; c < 4; c++) {
                  A[l][c] = hs_max3(vmx[l][c], vmx[l + 1][c], vmx[l + 2][c]);
                  I[l][c] = hs_min3(vmn[l][c], vmn[l + 1][c], vmn[l + 2][c]);
               }
         }
         uint32_t hits = 0;   // bit 4 * l + c
#pragma unroll
         for (int l = 0; l < 3; l++) {
            const float Al = hs_from_lane_below(A[l][3]), Ar = hs_from_lane_above(A[l][0]);
            const float Il = hs_from_lane_below(I[l][3]), Ir = hs_from_lane_above(I[l][0]);
            const float M[4] = {hs_max3(Al, A[l][0], A[l][1]), hs_max3(A[l][0], A[l][1], A[l][2]), hs_max3(A[l][1], A[l][2], A[l][3]), hs_max3(A[l][2], A[l][3], Ar)};
            const float N[4] = {hs_min3(Il, I[l][0], I[l][1]), hs_min3(I[l][0], I[l][1], I[l][2]), hs_min3(I[l][1], I[l][2], I[l][3]), hs_min3(I[l][2], I[l][3], Ir)};
            const float v[4] = {ring[l + 1][s1].x, ring[l + 1][s1].y, ring[l + 1][s1].z, ring[l + 1][s1].w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
               const bool pos = v[c] > posThr && !(M[c] > v[c]);
               const bool neg = v[c] < negThr && !(N[c] < v[c]);
               if (pos || neg) hits |= 1u << (4 * l + c);
            }
         }
         hits &= colmask * 0x111u;
         if (k < 2 || k >= nsteps) hits = 0;
         if (__ballot(hits != 0u) != 0ull) {
            // The candidates of this row leave with the 19 values localizeKeypoint's first iteration reads.
#pragma unroll
            for (int l = 0; l < 3; l++) {
               if (__ballot(((hits >> (4 * l)) & 15u) != 0u) == 0ull) continue;   // wave-uniform: nothing at this level
               // columns x - 1 and x + 4 belong to the neighbouring lanes: the level's own plane (l + 1) in the three rows, the planes
               // below and above it in the middle row
#if HS_CAND_PAYLOAD
               const float lowL = hs_from_lane_below(ring[l][s1].w), lowR = hs_from_lane_above(ring[l][s1].x);
               const float highL = hs_from_lane_below(ring[l + 2][s1].w), highR = hs_from_lane_above(ring[l + 2][s1].x);
               const float cL0 = hs_from_lane_below(ring[l + 1][s0].w), cR0 = hs_from_lane_above(ring[l + 1][s0].x);
               const float cL1 = hs_from_lane_below(ring[l + 1][s1].w), cR1 = hs_from_lane_above(ring[l + 1][s1].x);
               const float cL2 = hs_from_lane_below(ring[l + 1][s2].w), cR2 = hs_from_lane_above(ring[l + 1][s2].x);
#else
               const float lowL = 0, lowR = 0, highL = 0, highR = 0, cL0 = 0, cR0 = 0, cL1 = 0, cR1 = 0, cL2 = 0, cR2 = 0;
#endif
               const float cu[3][6] = {
                  {cL0, ring[l + 1][s0].x, ring[l + 1][s0].y, ring[l + 1][s0].z, ring[l + 1][s0].w, cR0},
                  {cL1, ring[l + 1][s1].x, ring[l + 1][s1].y, ring[l + 1][s1].z, ring[l + 1][s1].w, cR1},
                  {cL2, ring[l + 1][s2].x, ring[l + 1][s2].y, ring[l + 1][s2].z, ring[l + 1][s2].w, cR2}};
               const float lo1[6] = {lowL, ring[l][s1].x, ring[l][s1].y, ring[l][s1].z, ring[l][s1].w, lowR};
               const float hi1[6] = {highL, ring[l + 2][s1].x, ring[l + 2][s1].y, ring[l + 2][s1].z, ring[l + 2][s1].w, highR};
               const float lo0[4] = {ring[l][s0].x, ring[l][s0].y, ring[l][s0].z, ring[l][s0].w}, lo2[4] = {ring[l][s2].x, ring[l][s2].y, ring[l][s2].z, ring[l][s2].w};
               const float hi0[4] = {ring[l + 2][s0].x, ring[l + 2][s0].y, ring[l + 2][s0].z, ring[l + 2][s0].w}, hi2[4] = {ring[l + 2][s2].x, ring[l + 2][s2].y, ring[l + 2][s2].z, ring[l + 2][s2].w};
#pragma unroll
               for (int c = 0; c < 4; c++) {
                  const bool h = (hits >> (4 * l + c)) & 1u;
                  const unsigned long long m = __ballot(h);
                  if (m == 0ull) continue;   // wave-uniform
                  const uint32_t total = (uint32_t)__popcll(m);
                  if (wused + total > HS_CAND_BLOCK) {
                     if (wused < HS_CAND_BLOCK && (uint32_t)lane < HS_CAND_BLOCK - wused && wbase + wused + lane < cl.cap) cl.items[wbase + wused + lane].id0 = HS_CAND_HOLE;
                     uint32_t nb = 0;
                     if (lane == 0) nb = atomicAdd(cl.count, HS_CAND_BLOCK);
                     wbase = __builtin_amdgcn_readfirstlane(nb);
                     wused = 0;
                  }
                  if (h) {
                     const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                     const uint32_t slot = wbase + wused + rank;
                     if (slot < cl.cap) {
                        float4 *dst = reinterpret_cast<float4 *>(cl.items + slot);
                        dst[0] = make_float4(__uint_as_float(((uint32_t)b << 2) | (uint32_t)l), __uint_as_float(((uint32_t)y << 16) | (uint32_t)(x + c)), cu[0][c], cu[0][c + 1]);
#if HS_CAND_PAYLOAD
                        dst[1] = make_float4(cu[0][c + 2], cu[1][c], cu[1][c + 1], cu[1][c + 2]);
                        dst[2] = make_float4(cu[2][c], cu[2][c + 1], cu[2][c + 2], lo1[c + 1]);
                        dst[3] = make_float4(lo1[c], lo1[c + 2], lo0[c], lo2[c]);
                        dst[4] = make_float4(hi1[c + 1], hi1[c], hi1[c + 2], hi0[c]);
                        dst[5] = make_float4(hi2[c], 0.0f, 0.0f, 0.0f);
#endif
                     }
                  }
                  wused += total;
               }
            }
         }
      }
   }
   // the unused rest of the last block
   if (wused < HS_CAND_BLOCK && (uint32_t)lane < HS_CAND_BLOCK - wused && wbase + wused + lane < cl.cap) cl.items[wbase + wused + lane].id0 = HS_CAND_HOLE;
}
