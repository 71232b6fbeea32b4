// kernels_sift.h -- SIFT descriptor (siftdesc.cpp:115-140, helpers.cpp:246-281) over patches
// held in HBM, as four kernels whose parallel axis matches the structure of the reference's
// arithmetic instead of fighting it:
//
//  k_sift_meanvar   the photometric mean / variance are long SEQUENTIAL float sums
//                   (helpers.cpp:253-266, 1245 terms each).  One THREAD per keypoint runs the
//                   chain, 16 keypoints per wavefront (a quarter of the lanes add: the kernel is bound by its
//                   two reads of every patch, and the smaller working set lets the memory-side cache serve the second);
//                   the patch columns are transposed through LDS so that global loads stay coalesced.
//  k_sift_grad      one block per keypoint, one THREAD per pixel: normalise (once, through LDS),
//                   gradient, hm_atan2f_tab -> (mask*grad, o) pairs of the 40x40 weighted pixels, collected in
//                   an LDS tile and written as whole cache lines (the kernel is bound by this stream as much as
//                   by its arithmetic).
//  k_sift_hist      FOUR keypoints per wavefront, lane = (keypoint, spatial cell): the cell's 8
//                   orientation bins live in LDS ([bin][lane], conflict-free) and the lane walks
//                   its 16x16 support in raster order (siftdesc.cpp:51-81); the wave fetches each step's
//                   rows coalesced and hands them out through LDS.
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

#define HS_VO_DIM 40                          // rows/columns of the patch that carry weight in samplePatch
#define HS_VO_TILE (HS_VO_DIM * HS_VO_DIM)    // float2 of k_sift_grad's LDS tile (row-major 40 x 40)
// The gradient pairs of a keypoint in HBM (written once by k_sift_grad, read by k_sift_hist: the largest stream of the descriptor stage, and
// what both kernels are bound by - profiles/r05_notes.md).  HS_VO_COMPACT: only the 16-byte items (2 pixels) of every row's span inside the
// circular mask are stored, row after row (row r: items f_lo(r) .. f_hi(r) of its 20) - 642 items + 6 zero items instead of 800: 10.4 KB
// instead of 12.8 KB per keypoint.  KpTables::vo_rows / vo_src (host-built from the mask itself) describe the layout to both kernels.
#ifndef HS_VO_COMPACT
#define HS_VO_COMPACT 1
#endif
#define HS_VO_ITEMS (HS_VO_COMPACT ? 648 : 800)   // 16-byte items per keypoint; the items from HS_VO_ZERO on are zero
#define HS_VO_ZERO (HS_VO_COMPACT ? 642 : 0)      // an item that is (0, 0, 0, 0) in every keypoint's block (plain layout: row 0 holds no masked pixel)
#define HS_VO_PITCH (2 * HS_VO_ITEMS)             // float2 per keypoint in the gradient-pair buffer
#define HS_SIFT_MSK_IT 5   // ceil(1245 / 256): pixels inside the circular mask per thread of a 256-thread block
#define SM_TILE 64

// grid: ceil(n / SM_KP) blocks of 64 threads.
// SM_KP = keypoints per wavefront.  64 (every lane owns a keypoint) is the form with the fewest instructions, but a wavefront
// then re-reads its 64 patches (430 KB) in the second pass long after the first, and the kernel is bound by HBM (VALU 9 % busy).
// With 16 keypoints per wavefront (lanes 0..15 add; tiles of 16 keypoints x 256 masked pixels, the same 64 loads in flight) the
// patches that a CU's resident wavefronts hold between their two passes shrink to about 0.75 MB - 190 MB over the device, inside
// its 256 MB memory-side cache, which then serves most of the second pass.  Measured per 256 UHD images, kernel alone:
// SM_KP = 64 / 32 / 16 / 8: 111.5 / 100.8 / 89.0 / 254.9 ms (8: the additions become the bottleneck); step 868 / 862 / 854 / 954 ms.
#ifndef SM_KP
#define SM_KP 16
#endif
#ifndef SM_PX
#define SM_PX (64 * 64 / SM_KP)      // masked pixels per tile step (a multiple of 64)
#endif
#ifndef SM_UNROLL
#define SM_UNROLL 8
#endif
#define SM_ROW (SM_PX + 1)           // LDS row stride: lane k walking row k is conflict-free
__global__ __launch_bounds__(64) void k_sift_meanvar(SiftIO io, KpTables tb)
{
   __shared__ float s_tile[SM_KP * SM_ROW];
   const int lane = threadIdx.x;
   const uint32_t n = io.h_hi - io.h_lo;
   const uint32_t k0 = blockIdx.x * SM_KP;           // first keypoint (relative) of this block
   const int nm = tb.n_masked;
   const float gsum = (float)nm;
   const uint32_t kmine = min(k0 + (uint32_t)min(lane, SM_KP - 1), n - 1);
   // pass 0: sum ; pass 1: sum of squared deviations
   float sum = 0.0f, mean = 0.0f;
   for (int pass = 0; pass < 2; pass++) {
      float acc = 0.0f;
      for (int c0 = 0; c0 < nm; c0 += SM_PX) {
         const int cnt = min(SM_PX, nm - c0);
         // stage: row k = keypoint k0+k, column l = masked pixel c0+l ; coalesced along l (64 lanes x SM_PX / 64 columns each)
         int pix[SM_PX / 64];
#pragma unroll
         for (int u = 0; u < SM_PX / 64; u++) pix[u] = (lane + 64 * u < cnt) ? tb.mask_idx[c0 + lane + 64 * u] : 0;
#pragma unroll SM_UNROLL
         for (int k = 0; k < SM_KP; k++) {
            const uint32_t kp = min(k0 + k, n - 1);
            const float *pp = io.patches + (size_t)kp * HS_PATCH_PIX;
#pragma unroll
            for (int u = 0; u < SM_PX / 64; u++) {
               const float *q = pp + pix[u];
               s_tile[k * SM_ROW + lane + 64 * u] = (HS_NT_MEANVAR == 2 || (HS_NT_MEANVAR == 1 && pass == 1)) ? hs_load_nt(q) : *q;
            }
         }
         __syncthreads();
         if (lane < SM_KP) {
            const float *row = s_tile + lane * SM_ROW;
            if (pass == 0) {
               for (int l = 0; l < cnt; l++) acc += row[l];                                   // helpers.cpp:257
            } else {
               for (int l = 0; l < cnt; l++) { const float d = mean - row[l]; acc += d * d; }   // helpers.cpp:266
            }
         }
         __syncthreads();
      }
      if (pass == 0) { sum = acc; mean = sum / gsum; }
      else {
         const float var = sqrtf(acc / gsum);   // helpers.cpp:268
         if (lane < SM_KP && k0 + lane < n) { io.meanvar[2 * (size_t)kmine] = mean; io.meanvar[2 * (size_t)kmine + 1] = var; }
      }
   }
}

// k_sift_grad: photometric normalisation (helpers.cpp:269-280) + gradient magnitude and
// orientation (siftdesc.cpp:123-137) + the per-pixel factors of samplePatch.  Every pixel of a patch is
// normalised once into LDS, then one thread per pixel takes the gradient stencil from LDS.
// Output: vo[k][r][c] = (mask*grad, o), r, c < 40, with o = float(8 * (atan2f + 2 pi) / (2 pi)) evaluated in
// double like the reference.  Row and column 40 are not produced: their spatial weights are zero (bin0 and bin1
// both clamped, siftdesc.cpp:33-44) so samplePatch adds nothing for them; neither are the pixels outside the
// circular mask (mask*grad = 0 adds nothing either).
// This kernel is bound by VALU issue (correctly rounded sqrt and two divisions, fdlibm atan2f, a double-precision
// quotient per pixel), so everything that does not depend on the keypoint is computed ONCE per thread: blocks are
// persistent (grid-stride over keypoints), a thread owns the same five pixels of every patch and keeps their
// neighbour addresses, output offset and mask value in registers; the interval constants of atanf come from an
// LDS table (hm_atan2f_tab: no select trees, no divergence).  The next keypoint's pixels are requested before
// the current one is evaluated.
// grid: min(n, 256 * 8) blocks of 256 threads.
#ifndef HS_SGRAD_TILE
#define HS_SGRAD_TILE 1
#endif
#ifndef HS_SGRAD_WAVES
#define HS_SGRAD_WAVES 0   // tuning: wavefronts per SIMD to hold the register allocation to (0: the compiler's choice)
#endif
__global__ __launch_bounds__(256, HS_SGRAD_WAVES) void k_sift_grad(SiftIO io, KpTables tb, float2 *__restrict__ vo)
{
   __shared__ float s_p[HS_PATCH_PIX];
   __shared__ float s_at[HM_ATAN_TAB_FLOATS];
#if HS_SGRAD_TILE
   // The keypoint's 40 x 40 pairs are collected here and leave as whole 16-byte items of consecutive addresses (the items of the rows'
   // masked spans; zeros for the pixels of an item outside the mask, which no thread ever writes): 81 full cache lines per keypoint
   // instead of 8-byte pieces that start and end in the middle of lines.  The kernel is bound by this write stream (without the
   // stores it runs in half the time).
   __shared__ __attribute__((aligned(16))) float2 s_vo[HS_VO_TILE];
#endif
   const int tid = threadIdx.x;
   const uint32_t n = io.h_hi - io.h_lo;
#if HS_SGRAD_TILE
   for (int i = tid; i < HS_VO_TILE; i += 256) s_vo[i] = make_float2(0.0f, 0.0f);
   // this thread's items of the keypoint's block in HBM: which tile item each of them is (KpTables::vo_src)
   constexpr int VO_NI = (HS_VO_ITEMS + 255) / 256;
   int vsrc[VO_NI];
#pragma unroll
   for (int i = 0; i < VO_NI; i++) vsrc[i] = tb.vo_src[min(tid + 256 * i, HS_VO_ITEMS - 1)];
#endif
   {
      const float at_init[HM_ATAN_TAB_FLOATS] = HM_ATAN_TAB_INIT;
      if (tid < HM_ATAN_TAB_FLOATS) s_at[tid] = at_init[tid];
   }
   // this thread's pixels inside the circular mask (1245 of 1681, helpers.cpp:131): stencil neighbours (LDS byte offsets),
   // output slot and mask value come from one table row per pixel (KpTables::sgrad_*), all requested in one round
   int4 nbq[HS_SIFT_MSK_IT];
   int2 omq[HS_SIFT_MSK_IT];
#pragma unroll
   for (int q = 0; q < HS_SIFT_MSK_IT; q++) { nbq[q] = tb.sgrad_nb[tid + 256 * q]; omq[q] = tb.sgrad_om[tid + 256 * q]; }
   uint32_t k = blockIdx.x;
   if (k >= n) return;
   // first keypoint's operands
   float pv[HS_PATCH_PIX_IT];
   int alive = io.alive[io.h_lo + k];
   float mean = io.meanvar[2 * (size_t)k], var = io.meanvar[2 * (size_t)k + 1];
   {
      const float *gp = io.patches + (size_t)k * HS_PATCH_PIX;
#pragma unroll
      for (int q = 0; q < HS_PATCH_PIX_IT; q++) pv[q] = HS_NT_SGRAD_LD ? hs_load_nt(gp + min(tid + 256 * q, HS_PATCH_PIX - 1)) : gp[min(tid + 256 * q, HS_PATCH_PIX - 1)];
   }
   for (; k < n; k += gridDim.x) {
      const bool cur_alive = alive != 0;
      // normalise this keypoint's pixels into LDS (helpers.cpp:269-280: ALL pixels, not only the masked ones)
      const bool cur_norm = !((double)var < 0.0001);   // helpers.cpp:270
      if (cur_alive) {
         const bool norm = cur_norm;
         const float fac = 50.0f / var;
#pragma unroll
         for (int q = 0; q < HS_PATCH_PIX_IT; q++) {
            const int i = tid + 256 * q;
            if (i < HS_PATCH_PIX) {
               float v = pv[q];
               if (norm) { v = 128 + fac * (v - mean); v = v > 255 ? 255.0f : v; v = v < 0 ? 0.0f : v; }
               s_p[i] = v;
            }
         }
      }
      // request the next keypoint's operands before evaluating this one
      const uint32_t kn = k + gridDim.x;
      if (kn < n) {
         alive = io.alive[io.h_lo + kn];
         mean = io.meanvar[2 * (size_t)kn]; var = io.meanvar[2 * (size_t)kn + 1];
         const float *gp = io.patches + (size_t)kn * HS_PATCH_PIX;
#pragma unroll
         for (int q = 0; q < HS_PATCH_PIX_IT; q++) pv[q] = HS_NT_SGRAD_LD ? hs_load_nt(gp + min(tid + 256 * q, HS_PATCH_PIX - 1)) : gp[min(tid + 256 * q, HS_PATCH_PIX - 1)];
      }
      __syncthreads();
      if (cur_alive) {
         // ND (block-uniform): the patch was photometrically normalised.  Its pixels are then 128 + fac * (v - mean) clamped
         // to [0, 255]: multiples of 2^-17, so gx and gy are zero or at least 2^-17 in magnitude (never denormal) and
         // gx^2 + gy^2 is zero or at least 2^-34: the square root and the two divisions of atan2f need no range handling
         // (hmath.h: hm_sqrt_normal, hm_atan2f_tab_nd).  A flat patch (variance below 1e-4, helpers.cpp:270) keeps its
         // raw pixels and the general forms.
         auto pixels = [&](auto nd_c) {
            constexpr bool ND = decltype(nd_c)::value;
#pragma unroll
            for (int q = 0; q < HS_SIFT_MSK_IT; q++) {
               if (omq[q].x >= 0) {
                  const char *sp = reinterpret_cast<const char *>(s_p);
                  const float gx = *reinterpret_cast<const float *>(sp + nbq[q].y) - *reinterpret_cast<const float *>(sp + nbq[q].x);
                  const float gy = *reinterpret_cast<const float *>(sp + nbq[q].w) - *reinterpret_cast<const float *>(sp + nbq[q].z);
                  const float grad = ND ? hm_sqrt_normal(gx * gx + gy * gy) : sqrtf(gx * gx + gy * gy);
                  const float ori = ND ? hm_atan2f_tab_nd(gy, gx, s_at) : hm_atan2f_tab(gy, gx, s_at);
                  const float o = hm_sift_orient_coord(ori);
#if HS_SGRAD_TILE
                  s_vo[omq[q].x] = make_float2(__int_as_float(omq[q].y) * grad, o);
#elif HS_NT_VO
                  hs_store_nt2(reinterpret_cast<float *>(out + omq[q].x), __int_as_float(omq[q].y) * grad, o);
#else
                  out[omq[q].x] = make_float2(__int_as_float(omq[q].y) * grad, o);
#endif
               }
               // the loop is unrolled only so that the per-pixel constants are registers; do not let the scheduler
               // interleave the iterations (five atan2 bodies in flight cost ~60 VGPRs)
               __builtin_amdgcn_sched_barrier(0);
            }
         };
         if (cur_norm) pixels(std::true_type{});
         else pixels(std::false_type{});
      }
#if HS_SGRAD_TILE
      __syncthreads();
      if (cur_alive) {
         float4 *o4 = reinterpret_cast<float4 *>(vo + (size_t)k * HS_VO_PITCH);
         const float4 *t4 = reinterpret_cast<const float4 *>(s_vo);
#pragma unroll
         for (int i = 0; i < VO_NI; i++) {
            const int e = tid + 256 * i;
            if (e < HS_VO_ITEMS) {
               const float4 v = t4[vsrc[i]];
               hs_store_nt4(reinterpret_cast<float *>(o4 + e), v.x, v.y, v.z, v.w);
            }
         }
      }
#endif
      __syncthreads();
   }
}


// device check (stage API): the per-pixel forms of k_sift_grad, general and range-free, on caller-supplied operands
__global__ void k_math_sift(int n, const float *__restrict__ gy, const float *__restrict__ gx, float *__restrict__ ori_g,
                            float *__restrict__ ori_nd, float *__restrict__ grad_g, float *__restrict__ grad_nd)
{
   __shared__ float s_at[HM_ATAN_TAB_FLOATS];
   {
      const float at_init[HM_ATAN_TAB_FLOATS] = HM_ATAN_TAB_INIT;
      if (threadIdx.x < HM_ATAN_TAB_FLOATS) s_at[threadIdx.x] = at_init[threadIdx.x];
   }
   __syncthreads();
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const float y = gy[i], x = gx[i];
      ori_g[i] = hm_atan2f(y, x);
      ori_nd[i] = hm_atan2f_tab_nd(y, x, s_at);
      grad_g[i] = sqrtf(x * x + y * y);
      grad_nd[i] = hm_sqrt_normal(x * x + y * y);
   }
}

// k_sift_hist: samplePatch (siftdesc.cpp:51-81).  A wavefront takes FOUR keypoints; lane =
// (keypoint, spatial cell) owns the cell's 8 orientation bins as 8 LDS words laid out
// [bin][lane] (a lane only ever touches its own bank) and walks the cell's 16x16 pixel support
// in raster order, which is the order the reference adds a bin's terms in.  Per pixel: one
// product chain wr*(wc*val), two read-modify-writes at bins bo0 and bo0+1 (dynamic index, hence
// LDS and not registers).  Where the reference adds nothing (val <= 0) this adds +0.0f.
// The (mask*grad, o) rows of a step are fetched by the whole wave as consecutive 16-byte items and handed
// out through LDS (see "Row staging" below), the next step's rows in flight while the current ones are consumed.
// grid-stride over groups of 4 keypoints, block 64.
#ifndef HS_HIST_TRIM
#define HS_HIST_TRIM 1
#endif
#define HS_HIST_AHEAD 1   // steps the row items are requested ahead (2, with a second set of five registers, measured the same: profiles/r05_notes.md)
#ifndef HS_HIST_WAVES
#define HS_HIST_WAVES 0   // tuning: wavefronts per SIMD to hold the register allocation to (0: the compiler's choice)
#endif
__global__ __launch_bounds__(64, HS_HIST_WAVES) void k_sift_hist(SiftIO io, KpTables tb, const float2 *__restrict__ vo)
{
   __shared__ __attribute__((aligned(2048))) float s_acc[8 * 64];
   __shared__ float s_cw[64];   // [spatial bin][offset 0..15]
   const int tid = threadIdx.x;
   const int kq = tid >> 4, cell = tid & 15, cb_r = cell >> 2, cb_c = cell & 3;
   {
      // cell weights: spatial bin b gets w1[r] from rows with bin1 == b, w0[r] from rows with bin0 == b
      // (siftdesc.cpp:55-56,61-62); clamped bins carry weight 0
      const int b = tid >> 4, i = tid & 15, r = 8 * b + i;
      float w = 0.0f;
      if (r < HS_PATCH) {
         if (tb.bin0[r] == 8 * b && tb.w0[r] != 0.0f) w = tb.w0[r];
         else if (tb.bin1[r] == 8 * b) w = tb.w1[r];
      }
      s_cw[tid] = w;
      __syncthreads();
   }
   float cwc[16];
#pragma unroll
   for (int j = 0; j < 16; j++) cwc[j] = s_cw[cb_c * 16 + j];
   const uint32_t n = io.h_hi - io.h_lo;
   float *acc = s_acc + tid;
   // LDS pointers are 32-bit offsets (address space 3 behind the generic pointer): the lane's base with the bin bits clear
   typedef __attribute__((address_space(3))) float lds_float;
   const uint32_t acc_base = (uint32_t)(uintptr_t)(lds_float *)acc;
   // Row staging.  At step i the 64 lanes need, of each of the wave's four keypoints, the rows 8 cb_r + i (cb_r = 0..3):
   // 16 rows x 40 pixels x 8 bytes = 5 KB.  Read lane by lane (8 x 16 bytes each, rows up to 320 bytes and keypoints 10.4 KB
   // apart) every load instruction touches ~32 cache lines (texture addresser 78 % busy).  Instead the wave fetches the
   // 16 rows as 320 consecutive 16-byte items (5 per lane, 3 rows per instruction), parks them in LDS and every lane
   // takes its 8 items from there; the next step's items are in flight meanwhile (20 registers instead of 32).
   __shared__ __attribute__((aligned(16))) float4 s_rows[16 * (HS_VO_DIM / 2)];
   // staged item e = tid + 64 u is item f of row 8 cb + i of keypoint kq_e (e = (4 kq_e + cb) * 20 + f).  s_vtab[row * 20 + f] = its place
   // in the keypoint's block in HBM: vo_rows[row].x + f inside the row's span, HS_VO_ZERO (a zero item) outside; the lane walks its five
   // table entries down the rows (40 bytes per step) - one 16-bit LDS read and one add per item and step
   __shared__ uint16_t s_vtab[HS_VO_DIM * (HS_VO_DIM / 2)];
   for (int q = tid; q < HS_VO_DIM * (HS_VO_DIM / 2); q += 64) {
      const int r = q / (HS_VO_DIM / 2), f = q - r * (HS_VO_DIM / 2);
      const int4 rw = tb.vo_rows[r];
      s_vtab[q] = (uint16_t)((f >= rw.y && f <= rw.z) ? rw.x + f : HS_VO_ZERO);
   }
   __syncthreads();
   int st_kq[5];
   const uint16_t *st_tab[5];   // &s_vtab[(8 cb) * 20 + f] of item u
#pragma unroll
   for (int u = 0; u < 5; u++) {
      const int e = tid + 64 * u, row16 = e / (HS_VO_DIM / 2);   // row16 = 4 * keypoint + row of cells
      st_kq[u] = row16 >> 2;
      st_tab[u] = s_vtab + 8 * (row16 & 3) * (HS_VO_DIM / 2) + (e - row16 * (HS_VO_DIM / 2));
   }
   const float4 *my_rows = s_rows + (4 * kq + cb_r) * (HS_VO_DIM / 2) + 4 * cb_c;
   for (uint32_t g = blockIdx.x; 4 * g < n; g += gridDim.x) {
      const uint32_t k = 4 * g + kq;
      const bool valid = k < n && io.alive[io.h_lo + min(k, n - 1)];
#pragma unroll
      for (int b = 0; b < 8; b++) acc[64 * b] = 0.0f;
      // the group's pairs; a group that runs past the list re-reads its last keypoint (values unused)
      const float4 *g4 = reinterpret_cast<const float4 *>(vo + (size_t)(4 * g) * HS_VO_PITCH);
      const int kmax = (int)min(3u, n - 1 - 4 * g);   // last keypoint of the group that exists
      int st_src[5];   // the item's keypoint block inside the group (a keypoint past the list: the last one that exists)
#pragma unroll
      for (int u = 0; u < 5; u++) st_src[u] = min(st_kq[u], kmax) * HS_VO_ITEMS;
      // five named registers, not an array: carried around the loop an array ends up in scratch memory
#if HS_NT_VO_LD
#define HS_VO_LD(p) hs_load_nt4(p)
#else
#define HS_VO_LD(p) (*(p))
#endif
      auto item_at = [&](int u, int i) { return st_src[u] + (int)st_tab[u][i * (HS_VO_DIM / 2)]; };   // float4 offset of staged item u at step i
      float4 st0 = HS_VO_LD(g4 + item_at(0, 0)), st1 = HS_VO_LD(g4 + item_at(1, 0)), st2 = HS_VO_LD(g4 + item_at(2, 0)), st3 = HS_VO_LD(g4 + item_at(3, 0)),
             st4 = HS_VO_LD(g4 + item_at(4, 0));
      // One step: park the step's items, request the items of step i + HS_HIST_AHEAD into the registers they came from, consume.
      auto step = [&](int i, float4 &a0, float4 &a1, float4 &a2, float4 &a3, float4 &a4) {
         HS_WAVE_LDS_SYNC();   // every lane has taken the previous step's items
         s_rows[tid] = a0; s_rows[tid + 64] = a1; s_rows[tid + 128] = a2; s_rows[tid + 192] = a3; s_rows[tid + 256] = a4;
         HS_WAVE_LDS_SYNC();
         {
            const int in = min(i + HS_HIST_AHEAD, 15);   // (unconditionally: the last rounds re-read row 15's items)
            a0 = HS_VO_LD(g4 + item_at(0, in)); a1 = HS_VO_LD(g4 + item_at(1, in)); a2 = HS_VO_LD(g4 + item_at(2, in)); a3 = HS_VO_LD(g4 + item_at(3, in));
            a4 = HS_VO_LD(g4 + item_at(4, in));
         }
         if (valid) {
            float4 cur[8];
#pragma unroll
            for (int m = 0; m < 8; m++) cur[m] = my_rows[m];
            const float wr = s_cw[cb_r * 16 + i];
#pragma unroll
            for (int j = 0; j < 16; j++) {
               const float qx = (j & 1) ? cur[j >> 1].z : cur[j >> 1].x;
               const float qy = (j & 1) ? cur[j >> 1].w : cur[j >> 1].y;
               const float wc = cwc[j] * qx;   // w[c] * (mask*grad)
               const float v = wr * wc;
#if HS_HIST_TRIM
               // siftdesc.cpp:63-77 in 14 instead of 19 vector instructions.  qy = o lies in [4, 12] (atan2f + 2 pi over 2 pi / 8), so
               // o - (int)o is v_fract_f32 (exact); v is a product of non-negative finite factors (weights, mask, a gradient of
               // finite pixels), i.e. +0 or positive: where the reference skips a term (`v > 0` false) v * wo is +0 and adding it
               // leaves the bin as it is, so the compare and the two selects go; the two bin addresses are bit-field inserts of
               // (int)o << 8 and ((int)o << 8) + 256 into the lane's base (s_acc is 2 KB-aligned, bits 8..10 select the bin).
               const float wo1 = __builtin_amdgcn_fractf(qy);
               const float wo0 = 1.0f - wo1;
               const uint32_t x8 = (uint32_t)(int)qy << 8;
               lds_float *p0 = (lds_float *)(uintptr_t)((x8 & 0x700u) | acc_base);
               lds_float *p1 = (lds_float *)(uintptr_t)(((x8 + 0x100u) & 0x700u) | acc_base);
               const float t0 = v * wo0, t1 = v * wo1;
               const float a0 = *p0, a1 = *p1;   // bo0 != bo1: both reads in flight together
               *p0 = a0 + t0;
               *p1 = a1 + t1;
#else
               const int io0 = (int)qy;
               const int bo0 = io0 & 7, bo1 = (io0 + 1) & 7;
               const float wo1 = qy - (float)io0;
               const float wo0 = 1.0f - wo1;
               const bool pos = v > 0.0f;
               const float t0 = pos ? v * wo0 : 0.0f;   // goes to bin bo0
               const float t1 = pos ? v * wo1 : 0.0f;   // goes to bin bo0 + 1
               const float a0 = acc[64 * bo0], a1 = acc[64 * bo1];   // bo0 != bo1: both reads in flight together
               acc[64 * bo0] = a0 + t0;
               acc[64 * bo1] = a1 + t1;
#endif
            }
         }
      };
#pragma unroll 1
      for (int i = 0; i < 16; i++) step(i, st0, st1, st2, st3, st4);
      if (k < n) {
         float4 *dst = reinterpret_cast<float4 *>(io.vec + (size_t)k * 128 + cell * 8);
         dst[0] = make_float4(acc[0], acc[64], acc[128], acc[192]);
         dst[1] = make_float4(acc[256], acc[320], acc[384], acc[448]);
      }
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
