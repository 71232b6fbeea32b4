// kernels_patch.h -- affine patch normalisation (AffineShape::normalizeAffine,
// affine.cpp:102-144), one 256-thread block per keypoint.  Keypoints are binned by the side P of
// the warped window (hs_patch_bin); the default path only EXTRACTS the 41x41 patch (to HBM, the
// descriptor runs in kernels_sift.h):
//   k_patch_extract_small<0|1>  P <= 41 | 64 : window S and row-pass plane T in LDS, stored with
//                                      replicated borders; tap count as template parameter
//   k_patch_mid<128|512>        P <= 128 | 512 : row-streamed, three window rows per wavefront at a time; only the
//                                      82 blurred columns / rows the 41x41 resample reads are evaluated, T'
//                                      (P x 82, padded) in a per-block HBM slot, its rows staged through LDS
//                                      for the column pass
//   k_patch_large_rows + k_patch_large_finish  P > 512 : one wavefront per chunk of window ROWS
//                                      writes T' rows to HBM (all rows of all huge keypoints run
//                                      in parallel), then one block per keypoint finishes.
// Skipping blur outputs nobody reads does not change any value that is read: every
// evaluated tap sum uses the pinned cv::GaussianBlur order:
//   row   : t = k[0]*S[x-r]; t += k[j]*S[x-r+j]  (j ascending)       K > 5
//           S0*k0 + (S-1+S1)*k1 + (S-2+S2)*k2                         K <= 5
//   column: d = k[r]*T[y];  d += k[r+j]*(T[y+j] + T[y-j])
//
// What keeps the instruction count down:
//   * k_prepare_patch already decided whether the P x P window leaves the image (hs_window_outside:
//     the corner test is exactly interpolate()'s return flag), so no tap of these kernels tests bounds
//     and no kernel has a reject path;
//   * tap coordinates come from two small tables per window, R[row] = (ofsx + j*a12, ofsy + j*a22) and
//     C[col] = (i*a11, i*a21): (wx, wy) = R + C is the reference's rx + i*a11, ry + i*a21 in one packed add;
//   * both blur passes work on PAIRS of adjacent outputs with packed FP32 (v_pk_mul_f32 / v_pk_add_f32:
//     two IEEE products / sums per instruction, same rounding as the scalar forms, no FMA); bin 1 register-blocks
//     4 columns / 4 rows per lane instead (fewer LDS bytes per output);
//   * the gathers of a window are issued in batches whose size is fixed per window (no tap evaluated twice, no
//     branch inside a batch); finished patches are stored non-temporally (their readers are later kernels).
#pragma once
#include <type_traits>
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

// ---- helpers.cpp:227-240 : one bilinear tap that is known to lie inside the image ----
// (0 <= floor(w) < cols-1 / rows-1 was established for the whole window by hs_window_outside; the buffer
// resource still bounds every load to the plane.)
__device__ __forceinline__ float hs_tap_inside(const HsPlaneBuf &im, float wx, float wy)
{
#if HS_TAP_FRACT
   const uint32_t off = ((uint32_t)(int)wy * im.pitch + (uint32_t)(int)wx) * 4u;   // w >= 0: truncation is floor (see hs_tap_inside_ptr)
   wx = __builtin_amdgcn_fractf(wx);
   wy = __builtin_amdgcn_fractf(wy);
#else
   const float fx = floorf(wx), fy = floorf(wy);
   wx -= fx;
   wy -= fy;
   const uint32_t off = ((uint32_t)(int)fy * im.pitch + (uint32_t)(int)fx) * 4u;
#endif
   const hs_v2u r0 = __builtin_amdgcn_raw_buffer_load_b64(im.rsrc, (int)off, 0, 0);
   const hs_v2u r1 = __builtin_amdgcn_raw_buffer_load_b64(im.rsrc, (int)off, (int)im.pitch_bytes, 0);
   const float p00 = __uint_as_float(r0.x), p01 = __uint_as_float(r0.y), p10 = __uint_as_float(r1.x), p11 = __uint_as_float(r1.y);
   return (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
}

// Coordinate tables of one window: interpolate() (helpers.cpp:209-244) evaluates, for output (j, i),
//    rx = ofsx + j*a12;  ry = ofsy + j*a22;  wx = rx + i*a11;  wy = ry + i*a21        (j, i ints promoted to float)
// R[jj] = (rx, ry) for j = jj - half and C[ii] = (i*a11, i*a21) for i = ii - half hold the same roundings.
__device__ __forceinline__ v2f hs_row_coord(float ofsx, float ofsy, float a12, float a22, int j)
{
   v2f r;
   r.x = ofsx + (float)j * a12;
   r.y = ofsy + (float)j * a22;
   return r;
}
__device__ __forceinline__ v2f hs_col_coord(float a11, float a21, int i)
{
   v2f c;
   c.x = (float)i * a11;
   c.y = (float)i * a21;
   return c;
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

__device__ __forceinline__ int hs_div_small(int idx, float inv)   // floor(idx / P) for idx < 2^16, P < 2^8, inv = 1.0f / P
{
   return (int)(((float)idx + 0.5f) * inv);
}

// S: blurred window with row pitch `pitch`
#ifndef HS_RESAMPLE_COLS
#define HS_RESAMPLE_COLS 1
#endif
template <int PITCH>
__device__ __forceinline__ void hs_resample_full_tab(const float *S, const int *tab_i, const float *tab_f, float *out)
{
#if HS_RESAMPLE_COLS
   // Thread t < 246 owns column ii = t % 41 of the rows jr, jr + 6, ... (jr = t / 41): the column's table entries and
   // 1 - wx are registers, a row's entries are one broadcast read, and output index jj * 41 + ii = t + 246 k needs no
   // division (the flat form below spends a third of its instructions on idx -> (jj, ii) and the four table reads).
   constexpr int RPP = 256 / HS_PATCH;   // 6 rows per pass
   const int t = threadIdx.x;
   const int jr = hs_div_small(t, 1.0f / (float)HS_PATCH), ii = t - jr * HS_PATCH;
   if (jr < RPP) {
      const int xi = tab_i[ii];
      const float wx = tab_f[ii], wx1 = 1.0f - wx;
#pragma unroll
      for (int k = 0; k < (HS_PATCH + RPP - 1) / RPP; k++) {
         const int jj = jr + RPP * k;
         if (jj < HS_PATCH) {
            const int yi = tab_i[jj];
            const float wy = tab_f[jj];
            const bool in = (xi | yi) >= 0;
            const float *p = S + (in ? yi * PITCH + xi : 0);
            const float p00 = p[0], p01 = p[1], p10 = p[PITCH], p11 = p[PITCH + 1];
            const float v = (1.0f - wy) * (wx1 * p00 + wx * p01) + (wy) * (wx1 * p10 + wx * p11);
            float *o = out + t + RPP * HS_PATCH * k;
            if (HS_NT_PATCH) hs_store_nt(o, in ? v : 0.0f); else *o = in ? v : 0.0f;
         }
      }
   }
#else
   constexpr int pitch = PITCH;
   for (int idx = threadIdx.x; idx < HS_PATCH_PIX; idx += 256) {
      const int jj = hs_div_small(idx, 1.0f / (float)HS_PATCH), ii = idx - jj * HS_PATCH;
      const int xi = tab_i[ii], yi = tab_i[jj];
      const float wx = tab_f[ii], wy = tab_f[jj];
      const bool in = (xi | yi) >= 0;
      const float *p = S + (in ? yi * pitch + xi : 0);
      const float p00 = p[0], p01 = p[1], p10 = p[pitch], p11 = p[pitch + 1];
      const float v = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + (wy) * ((1.0f - wx) * p10 + wx * p11);
      if (HS_NT_PATCH) hs_store_nt(out + idx, in ? v : 0.0f); else out[idx] = in ? v : 0.0f;
   }
#endif
}

// ---------------------------------------------------------------------------------------
// k_patch_extract_small<BIN>: warp -> blur -> resample for windows P <= 41 (BIN 0) / 63 (BIN 1),
// result straight to io.patches.
//   * the window S is stored with r replicated columns on either side and the row-pass plane
//     T with r replicated rows above and below (BORDER_REPLICATE materialised), so no tap
//     needs an index clamp;
//   * the tap count K is a template parameter (K = 3..15 here, affine.cpp:129): the loops
//     are unrolled, taps sit in registers and all LDS reads of an output are in flight together;
//   * a thread computes the outputs of two adjacent columns at once (packed FP32); all row pitches
//     are even so that the pairs of the column pass are aligned 8-byte LDS accesses;
//   * idx -> (row, column) uses a float reciprocal (exact for these sizes) instead of the
//     ~20-instruction integer division.
// LDS: S[PMAX][SPITCH] | T[PMAX + 14][TPITCH] | taps | R | C   (19 KB / 40 KB: 8 / 4 blocks per CU).
// ---------------------------------------------------------------------------------------
#define HS_SMALL_RMAX 7
#define HS_WNIT0 4
#define HS_WNIT1 6
#ifndef HS_SMALL_WAVES
#define HS_SMALL_WAVES 6   // wavefronts per SIMD the register allocation of bin 0 is held to (0: the compiler's choice, 4);
#endif                     // measured: 17.0 / 15.7 / 14.3 ms per 32 UHD images at 4 / 5 / 6; bin 1 spills at 6 and stays at the compiler's choice
#ifndef HS_SMALL_BLK_BIN
#define HS_SMALL_BLK_BIN 1   // first bin whose blur runs register-blocked (measured: bin 1 -7 %, bin 0 +3 %: its windows waste more of a quad)
#endif
#ifndef HS_MID_WAVES
#define HS_MID_WAVES 0
#endif
// HS_TAPS_SCALAR: the row pass of the row-streamed windows takes its taps as SCALAR operands.  Everything that selects a window is the same
// for all lanes of a wavefront (the wavefront's task, its keypoint, P, K, the tap table's offset), but the compiler cannot know that of a value
// derived from threadIdx.x or read from LDS; with the wave index / the claimed item passed through v_readfirstlane the whole chain of per-item
// parameters lives in scalar registers, and taps[jt] is an s_load from the tap table in global memory (2.4 KB at most, scalar-cache resident)
// instead of a broadcast ds_read from a staged copy: a third of the row pass's LDS instructions gone, on kernels whose LDS pipe is 45-60 % busy.
// (The table is read through a pointer to the constant address space: what tells the compiler that the memory does not change during the
// kernel, without which a uniform load still goes through the vector memory path.)
#ifndef HS_TAPS_SCALAR
#define HS_TAPS_SCALAR 1
#endif
typedef __attribute__((address_space(4))) const float hs_cfloat;
#if HS_TAPS_SCALAR
typedef hs_cfloat *hs_row_taps;
#define HS_ROW_TAPS(global_table, lds_copy) (reinterpret_cast<hs_cfloat *>(reinterpret_cast<uintptr_t>(global_table)))
#else
typedef const float *hs_row_taps;
#define HS_ROW_TAPS(global_table, lds_copy) (lds_copy)
#endif
#ifndef HS_MID_SCALAR
#define HS_MID_SCALAR 1     // the row pass of the row-streamed windows as scalar sliding-window chains: every sample is read from LDS once (the pair form reads
                            // it twice) and no register pairs are assembled; k_patch_mid<512> 195.5 -> 178.1, <128> 149.5 -> 144.0 ms per 256 images
#endif
#ifndef HS_ABL_MIDCONF
#define HS_ABL_MIDCONF 0
#endif
#ifndef HS_MID_SCALAR_TAIL
#define HS_MID_SCALAR_TAIL 0   // the same form in the one- and two-row variants (tail rows of a window, k_patch_large_rows): measured slower (182.8 vs 179.4, 33.9 vs 32.2 ms)
#endif
#ifndef HS_SMALL_WCOLS
#define HS_SMALL_WCOLS 0   // tuning: the warp of the LDS-window bins with a fixed window column per thread (below); measured slower (profiles/r05_notes.md)
#endif
#ifndef HS_SMALL_SCALAR
#define HS_SMALL_SCALAR 1   // the plain (pair) form of the LDS-window blur with scalar instead of packed operations: -186 register moves, no scratch; bin 0 156.6 -> 153.7 ms per 256 images
#endif

template <int BIN> struct SmallGeom {
   static constexpr int PMAX = BIN == 0 ? 41 : 63;   // largest window of the bin: the bins are cut on P = P0 + 2 <= 41 | 64, P is odd
   // even pitches; one spare column each for the second output of an odd window's last column pair:
   // the row pass of column pair (P - 1, P) reads S up to index P + 2 r, and writes T / the blurred S up to column P
   static constexpr int SPITCH = (PMAX + 2 * HS_SMALL_RMAX + 2) & ~1;   // 56 | 78
   static constexpr int TPITCH = (PMAX + 2) & ~1;                       // 42 | 64
   static constexpr int SSZ = PMAX * SPITCH;
   static constexpr int TSZ = (PMAX + 2 * HS_SMALL_RMAX) * TPITCH;
   static constexpr int FLOATS = SSZ + TSZ + 16 + 4 * (PMAX + 1);       // + taps + R and C tables (float2 each)
   static_assert(SPITCH % 2 == 0 && TPITCH % 2 == 0 && SSZ % 2 == 0 && TSZ % 2 == 0, "pairs must stay 8-byte aligned");
   static_assert(SPITCH >= PMAX + 2 * HS_SMALL_RMAX + 1 && TPITCH >= PMAX + 1, "spare column");
   // the register-blocked passes read up to 3 floats past the last row of S (into T) and up to 3 rows past T (into the
   // tap / table area): values that only feed outputs which are not stored, but the addresses must stay inside the block's LDS
   static_assert(((PMAX + 3) & ~3) + 2 * HS_SMALL_RMAX - (PMAX + 2 * HS_SMALL_RMAX) <= 3, "rows read past T");
   static_assert(FLOATS - SSZ - TSZ >= 3 * TPITCH || BIN == 1, "overshoot stays inside the allocation");
};

// both blur passes of one window; KT = 0: run-time tap count (any odd K <= 15)
template <int KT, int SPITCH, int TPITCH, bool BLK>
__device__ __forceinline__ void hs_small_blur(float *S, float *T, int P, const float *s_taps, int Krt)
{
   const int K = KT ? KT : Krt, r = K >> 1;
   const int tid = threadIdx.x;
   float kk[KT ? KT : 2 * HS_SMALL_RMAX + 1];
#pragma unroll
   for (int j = 0; j < (KT ? KT : 2 * HS_SMALL_RMAX + 1); j++) kk[j] = (j < K) ? s_taps[j] : 0.0f;
   // replicated border columns of S
   if (r > 0) {
      const float inv2r = 1.0f / (float)(2 * r);
      for (int i = tid; i < 2 * r * P; i += 256) {
         const int yy = hs_div_small(i, inv2r), m = i - yy * 2 * r;
         float *row = S + yy * SPITCH;
         if (m < r) row[m] = row[r];
         else row[P + m] = row[r + P - 1];   // column r + P + (m - r)
      }
   }
   __syncthreads();
   if (BLK && KT >= 7) {
      // Register-blocked form (the tap counts these bins really see).  The plain form below reads K + 1 dwords of S
      // per two row-pass outputs and K pairs of T per two column-pass outputs; the LDS pipe (not the VALU) then sets
      // the pace.  Here a lane produces 4 adjacent columns of a row from K + 3 dwords, and 4 consecutive rows of a
      // column pair from K + 3 pairs: 2.5 x fewer LDS bytes per output; every output is still its own sequential chain.
      constexpr int KK = KT ? KT : 1, RR = KK >> 1;
      const int PQ = (P + 3) >> 2;                 // column quads per row / row quads per column pair
      const float invPQ = 1.0f / (float)PQ;
      for (int idx = tid; idx < P * PQ; idx += 256) {
         const int yy = hs_div_small(idx, invPQ), xx = 4 * (idx - yy * PQ);
         const float *sp = S + yy * SPITCH + xx;   // sp[j] = S[clamp(xx - r + j)]; outputs past column P - 1 read what follows the row, not stored
         float g[KK + 3];
#pragma unroll
         for (int j = 0; j < KK + 3; j += 2) {
            const v2f q = *reinterpret_cast<const v2f *>(sp + j);
            g[j] = q.x;
            g[j + 1] = q.y;
         }
         float t[4];
#pragma unroll
         for (int m = 0; m < 4; m++) {
            float a = kk[0] * g[m];
#pragma unroll
            for (int j = 1; j < KK; j++) a += kk[j] * g[m + j];
            t[m] = a;
         }
         v2f t01, t23;
         t01.x = t[0]; t01.y = t[1]; t23.x = t[2]; t23.y = t[3];
         const bool second = xx + 2 < P;           // pair (xx + 2, xx + 3) holds a window column (column P is the spare one)
         float *tq = T + (RR + yy) * TPITCH + xx;
         *reinterpret_cast<v2f *>(tq) = t01;
         if (second) *reinterpret_cast<v2f *>(tq + 2) = t23;
         if (yy == 0)
            for (int j = 0; j < RR; j++) {
               *reinterpret_cast<v2f *>(T + j * TPITCH + xx) = t01;
               if (second) *reinterpret_cast<v2f *>(T + j * TPITCH + xx + 2) = t23;
            }
         if (yy == P - 1)
            for (int j = 0; j < RR; j++) {
               *reinterpret_cast<v2f *>(T + (RR + P + j) * TPITCH + xx) = t01;
               if (second) *reinterpret_cast<v2f *>(T + (RR + P + j) * TPITCH + xx + 2) = t23;
            }
      }
      __syncthreads();
      const int PC = (P + 1) >> 1;
      const float invPC = 1.0f / (float)PC;
      for (int idx = tid; idx < PQ * PC; idx += 256) {
         const int yb = hs_div_small(idx, invPC), xx = 2 * (idx - yb * PC), y0 = 4 * yb;
         const float *tp = T + y0 * TPITCH + xx;   // T row y0 + j = window row y0 + j - r; rows past the plane (last block only) feed unstored outputs
         v2f c[KK + 3];
#pragma unroll
         for (int j = 0; j < KK + 3; j++) c[j] = *reinterpret_cast<const v2f *>(tp + j * TPITCH);
#pragma unroll
         for (int m = 0; m < 4; m++) {
            float dx = kk[RR] * c[m + RR].x, dy = kk[RR] * c[m + RR].y;
#pragma unroll
            for (int j = 1; j <= RR; j++) {
               dx += kk[RR + j] * (c[m + RR + j].x + c[m + RR - j].x);
               dy += kk[RR + j] * (c[m + RR + j].y + c[m + RR - j].y);
            }
            v2f d;
            d.x = dx; d.y = dy;
            if (y0 + m < P) *reinterpret_cast<v2f *>(S + (y0 + m) * SPITCH + xx) = d;
         }
      }
      __syncthreads();
      return;
   }
   const int PC = (P + 1) >> 1;                 // column pairs per row (an odd window's last pair computes one spare output)
   const float invPC = 1.0f / (float)PC;
   // row pass: T[r + y][x]; rows y = 0 and y = P-1 are also written into the r border rows
   for (int idx = tid; idx < P * PC; idx += 256) {
      const int yy = hs_div_small(idx, invPC), xx = 2 * (idx - yy * PC);
      const float *sp = S + yy * SPITCH + xx;   // sp[j] = S[clamp(xx - r + j)], sp[j + 1] the same for column xx + 1
      v2f t;
      auto G = [&](int j) { v2f g; g.x = sp[j]; g.y = sp[j + 1]; return g; };
      if (K == 1) t = G(0);
      else if (K <= 5) {
         t = G(r) * kk[r] + (G(r - 1) + G(r + 1)) * kk[r + 1];
         if (K == 5) t = t + (G(r - 2) + G(r + 2)) * kk[r + 2];
      } else if (HS_SMALL_SCALAR) {
         // the two chains as scalar operations: the window values are read once each (sp[j + 1] of tap j is sp[j] of tap j + 1),
         // no register pairs have to be assembled for packed operations that issue no faster than two scalar ones
         float t0 = kk[0] * sp[0], t1 = kk[0] * sp[1];
#pragma unroll
         for (int j = 1; j < K; j++) { t0 += kk[j] * sp[j]; t1 += kk[j] * sp[j + 1]; }
         t.x = t0; t.y = t1;
      } else {
         t = kk[0] * G(0);
#pragma unroll
         for (int j = 1; j < K; j++) t += kk[j] * G(j);
      }
      *reinterpret_cast<v2f *>(T + (r + yy) * TPITCH + xx) = t;
      if (yy == 0)
         for (int j = 0; j < r; j++) *reinterpret_cast<v2f *>(T + j * TPITCH + xx) = t;
      if (yy == P - 1)
         for (int j = 0; j < r; j++) *reinterpret_cast<v2f *>(T + (r + P + j) * TPITCH + xx) = t;
   }
   __syncthreads();
   // column pass, blurred window back into S with pitch SPITCH
   for (int idx = tid; idx < P * PC; idx += 256) {
      const int yy = hs_div_small(idx, invPC), xx = 2 * (idx - yy * PC);
      const float *tp = T + (r + yy) * TPITCH + xx;
      auto TT = [&](int j) { return *reinterpret_cast<const v2f *>(tp + j * TPITCH); };
      v2f d;
      if (HS_SMALL_SCALAR) {
         const v2f c = TT(0);
         float d0 = kk[r] * c.x, d1 = kk[r] * c.y;
#pragma unroll
         for (int j = 1; j <= r; j++) {
            const v2f a = TT(j), b = TT(-j);
            d0 += kk[r + j] * (a.x + b.x);
            d1 += kk[r + j] * (a.y + b.y);
         }
         d.x = d0; d.y = d1;
      } else {
         d = kk[r] * TT(0);
#pragma unroll
         for (int j = 1; j <= r; j++) d += kk[r + j] * (TT(j) + TT(-j));
      }
      *reinterpret_cast<v2f *>(S + yy * SPITCH + xx) = d;
   }
   __syncthreads();
}

template <int BIN>
__global__ __launch_bounds__(256, (BIN == 0 ? HS_SMALL_WAVES : 0)) void k_patch_extract_small(HessList hl, PatchWork pw, PatchIO io, KpTables tb)
{
   extern __shared__ __attribute__((aligned(16))) float smem[];
   typedef SmallGeom<BIN> GM;
   constexpr int PMAX = GM::PMAX, SPITCH = GM::SPITCH, TPITCH = GM::TPITCH;
   float *S = smem, *T = smem + GM::SSZ, *s_taps = T + GM::TSZ;
   v2f *s_R = reinterpret_cast<v2f *>(s_taps + 16), *s_C = s_R + (PMAX + 1);
   __shared__ int s_tab_i[HS_PATCH];
   __shared__ float s_tab_f[HS_PATCH];

   const int tid = threadIdx.x;
   const uint32_t cnt = min(pw.bin_count[BIN], pw.cap);
   const int imCols = io.image.cols, imRows = io.image.rows, imPitch = io.image.pitch;
   const int width = imCols - 1, height = imRows - 1;

   // static striding: the windows of these two bins cost about the same, and a claim per item on one counter
   // (2 M items per batch) would serialise in L2
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
         // direct branch, affine.cpp:137-141: the taps keep their bounds test (interpolateCheckBorders only
         // looked at the patch corners; the reference asserts that nothing is outside)
         const float b11 = a11 * scale, b12 = a12 * scale, b21 = a21 * scale, b22 = a22 * scale;
         // 7 taps per thread in two batches (4 + 3): one batch of 7 costs 23 more VGPRs
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
      if (tid < K) s_taps[tid] = taps_g[tid];
      if (tid < P) s_R[tid] = hs_row_coord(x, y, a12, a22, tid - half);
      else if (tid >= 128 && tid - 128 < P) s_C[tid - 128] = hs_col_coord(a11, a21, tid - 128 - half);
      hs_resample_table(P, scale, s_tab_i, s_tab_f);
      __syncthreads();
      // 1. warp, affine.cpp:126 (the window lies inside the image: k_prepare_patch).  All gathers of a
      // batch are issued before the first use.
      constexpr int WNIT = BIN == 0 ? HS_WNIT0 : HS_WNIT1;
      constexpr bool BLK = BIN >= HS_SMALL_BLK_BIN;
      const int PP = P * P;
      const float invP = 1.0f / (float)P;
      // The P x P taps are taken in slots of 256 (one per thread), up to WNIT slots per batch with all gathers of a
      // batch issued before the first use.  The batch size is a compile-time constant picked per window (a switch on a
      // block-uniform value): a fixed WNIT would evaluate 1024 taps for a window of 23 x 23 = 529, and a test per slot
      // inside the batch would serialise the gathers (measured: slower than the waste).
#if HS_SMALL_WCOLS
      // Thread t owns window column ii = t % P of the rows jr, jr + rpp, ... (jr = t / P, rpp = 256 / P rows per pass; threads
      // past rpp * P sample the last row again and store nothing): C[ii] and the LDS store address are per-thread constants, R[jj]
      // is a broadcast read, and no tap needs idx -> (jj, ii) (twice per tap in the flat form: before the gather and at the store).
      const int rpp = hs_div_small(256, invP);
      const int jr = hs_div_small(tid, invP), ii0 = tid - jr * P;
      const v2f cC = s_C[ii0];
      float *sdst = S + jr * SPITCH + r + ii0;
      const bool wact = jr < rpp;
      auto warp_batch = [&](auto nbc, int kb) {
         constexpr int NB = decltype(nbc)::value;
         float wv[NB];
#pragma unroll
         for (int it = 0; it < NB; it++) {
            const v2f w = s_R[min(jr + rpp * (kb + it), P - 1)] + cC;
            wv[it] = hs_tap_inside(pbuf, w.x, w.y);
         }
#pragma unroll
         for (int it = 0; it < NB; it++) HS_KEEP(wv[it]);
#pragma unroll
         for (int it = 0; it < NB; it++)
            if (wact && jr + rpp * (kb + it) < P) sdst[rpp * (kb + it) * SPITCH] = wv[it];
      };
      {
         int kb = 0;
         for (int rem = hs_div_small(P + rpp - 1, 1.0f / (float)rpp); rem > 0;) {
            const int nbatch = (rem + WNIT - 1) / WNIT;   // batches left; this one takes an even share of the passes
            const int nb = (rem + nbatch - 1) / nbatch;
            switch (nb) {
               case 1: warp_batch(std::integral_constant<int, 1>{}, kb); break;
               case 2: warp_batch(std::integral_constant<int, 2>{}, kb); break;
               case 3: warp_batch(std::integral_constant<int, 3>{}, kb); break;
               case 4: warp_batch(std::integral_constant<int, 4>{}, kb); break;
               case 5: if (WNIT >= 5) warp_batch(std::integral_constant<int, (WNIT >= 5 ? 5 : 1)>{}, kb); break;
               default: if (WNIT >= 6) warp_batch(std::integral_constant<int, (WNIT >= 6 ? 6 : 1)>{}, kb); break;
            }
            kb += nb;
            rem -= nb;
         }
      }
#else
      auto warp_batch = [&](auto nbc, int ib) {
         constexpr int NB = decltype(nbc)::value;
         float wv[NB];
#pragma unroll
         for (int it = 0; it < NB; it++) {
            const int idx = min(ib + tid + 256 * it, PP - 1);
            const int jj = hs_div_small(idx, invP), ii = idx - jj * P;
            const v2f w = s_R[jj] + s_C[ii];
            wv[it] = hs_tap_inside(pbuf, w.x, w.y);
         }
#pragma unroll
         for (int it = 0; it < NB; it++) HS_KEEP(wv[it]);
#pragma unroll
         for (int it = 0; it < NB; it++) {
            const int idx = ib + tid + 256 * it;
            if (idx < PP) {
               const int jj = hs_div_small(idx, invP), ii = idx - jj * P;
               S[jj * SPITCH + r + ii] = wv[it];
            }
         }
      };
      {
         int ib = 0;
         for (int rem = (PP + 255) >> 8; rem > 0;) {
            const int nbatch = (rem + WNIT - 1) / WNIT;   // batches left; this one takes an even share of the slots
            const int nb = (rem + nbatch - 1) / nbatch;
            switch (nb) {
               case 1: warp_batch(std::integral_constant<int, 1>{}, ib); break;
               case 2: warp_batch(std::integral_constant<int, 2>{}, ib); break;
               case 3: warp_batch(std::integral_constant<int, 3>{}, ib); break;
               case 4: warp_batch(std::integral_constant<int, 4>{}, ib); break;
               case 5: if (WNIT >= 5) warp_batch(std::integral_constant<int, (WNIT >= 5 ? 5 : 1)>{}, ib); break;
               default: if (WNIT >= 6) warp_batch(std::integral_constant<int, (WNIT >= 6 ? 6 : 1)>{}, ib); break;
            }
            ib += 256 * nb;
            rem -= nb;
         }
      }
#endif
      __syncthreads();
      // 2. blur, affine.cpp:129 (pinned cv::GaussianBlur order, see the file header)
      switch (K) {
         case 3: hs_small_blur<3, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
         case 5: hs_small_blur<5, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
         case 7: hs_small_blur<7, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
         case 9: hs_small_blur<9, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
         case 11: hs_small_blur<11, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
         case 13: hs_small_blur<13, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
         case 15: hs_small_blur<15, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
         default: hs_small_blur<0, SPITCH, TPITCH, BLK>(S, T, P, s_taps, K); break;
      }
      // 3. resample, affine.cpp:131
      hs_resample_full_tab<SPITCH>(S, s_tab_i, s_tab_f, out);
      __syncthreads();
   }
}

// The four column-pass sums an output of the 41x41 resample needs, from the row-pass plane T[rows][82] that holds
// only the needed columns: window rows y0 (chain a) and y0 + 1 (chain b), needed columns q and q + 1 (one aligned pair).
// Each sum keeps the SymmColumnFilter order d = k[r]*T[y]; d += k[r+j]*(T[y+j]+T[y-j]); the two chains share their
// rows (row y0 + j of chain a is row (y0 + 1) + (j - 1) of chain b), the loads of JC tap steps are issued together -
// the plane is read through L2 / HBM, where a load per tap step followed by its use is a full round trip per step -
// and the two columns of a row ride in one packed operation.  CLAMP = false: T points at window row 0 of a plane
// stored with r replicated rows above and below (k_patch_mid's HBM slot), no index clamps.
template <int JC, bool CLAMP, class TAPS = const float *>
__device__ __forceinline__ void hs_colpass4_rows(const float *__restrict__ T, int y0, int q, int pm, TAPS taps, int r,
                                                 v2f &pa, v2f &pb)
{
   // row y of the window -> pair at T[row][q]; CLAMP: BORDER_REPLICATE by index clamp (unpadded plane)
   auto ld = [&](int y) {
      const int row = CLAMP ? min(max(y, 0), pm) : y;
      return *reinterpret_cast<const v2f *>(T + row * HS_NEED + q);
   };
   const v2f c0 = ld(y0), c1 = ld(y0 + 1);
   const float kc = taps[r];
   v2f da = kc * c0, db = kc * c1;
   v2f pj = c1;   // row y0 + j      (j = 1)
   v2f mj = c0;   // row y0 + 1 - j  (j = 1)
   for (int j0 = 1; j0 <= r; j0 += JC) {
      v2f pn[JC], mn[JC];   // rows y0 + j + 1 and y0 - j
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
            const v2f sa = pj + mn[u], sb = pn[u] + mj;
            da += kj * sa;
            db += kj * sb;
            pj = pn[u];
            mj = mn[u];
         }
      }
   }
   pa = da;
   pb = db;
}

// resample of affine.cpp:131 from the row-pass plane at the 82 needed columns (PADDED: T points at
// window row 0 of a plane with r replicated rows above and below; otherwise rows are clamped)
template <bool PADDED, class TAPS = const float *>
__device__ __forceinline__ void hs_resample_reduced_batched(const float *__restrict__ T, int P, float scale, TAPS taps, int r,
                                                            float *s_patch)
{
   const float c0 = (float)(P >> 1);
   for (int idx = threadIdx.x; idx < HS_PATCH_PIX; idx += 256) {
      const int jj = hs_div_small(idx, 1.0f / (float)HS_PATCH), ii = idx - jj * HS_PATCH;
      const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
      const float rx = c0 + (float)j * 0.0f, ry = c0 + (float)j * scale;
      float wx = rx + (float)i * scale, wy = ry + (float)i * 0.0f;
      const float fx = floorf(wx), fy = floorf(wy);
      wx -= fx; wy -= fy;
      const int y0 = min(max((int)fy, 0), P - 2);   // always inside: |j * scale| < P0 / 2
      v2f pa, pb;   // (p00, p01), (p10, p11)
      hs_colpass4_rows<8, !PADDED, TAPS>(T, y0, 2 * ii, P - 1, taps, r, pa, pb);
      s_patch[idx] = (1.0f - wy) * ((1.0f - wx) * pa.x + wx * pa.y) + (wy) * ((1.0f - wx) * pb.x + wx * pb.y);
   }
}

// The same resample with the rows of T' staged through LDS.  Output row jj reads the plane rows y0(jj) - r .. y0(jj) + 1 + r
// and y0 grows by `scale` (1.5 .. 12.5) per output row while a window of rows is K + 1 = 9 * scale long: every row of T'
// is wanted by about ten output rows.  Read straight from the HBM slot (L2) that is (K + 1) dependent 8-byte loads per
// output, 21 round trips per wavefront and keypoint - 46 % of k_patch_mid<128>, 29 % of <512> in the ablation.  Here as
// many consecutive output rows as fit are taken per round: their rows of T' come in with one coalesced sweep, and the
// column pass reads LDS.  Same sums, same order.  T: window row 0 of the padded plane; chunk_rows >= K + 2.
template <class TAPS = const float *>
__device__ __forceinline__ void hs_resample_chunked(const float *__restrict__ T, int P, float scale, TAPS taps, int r,
                                                    float *__restrict__ s_chunk, int chunk_rows, float *s_patch)
{
   const int tid = threadIdx.x;
   const float c0 = (float)(P >> 1);
   auto y0_of = [&](int jj) {   // first of the two window rows output row jj interpolates between (as in hs_resample_reduced_batched)
      const float wy = c0 + (float)(jj - (HS_PATCH >> 1)) * scale;
      return min(max((int)floorf(wy), 0), P - 2);
   };
   for (int jj0 = 0; jj0 < HS_PATCH;) {
      const int ylo = y0_of(jj0) - r;
      int jj1 = jj0;
      while (jj1 + 1 < HS_PATCH && y0_of(jj1 + 1) + r + 2 - ylo <= chunk_rows) jj1++;
      const int nrows = y0_of(jj1) + r + 2 - ylo;
      {
         const v2f *src = reinterpret_cast<const v2f *>(T + ylo * HS_NEED);   // ylo >= -r: the plane has r rows above window row 0
         v2f *dst = reinterpret_cast<v2f *>(s_chunk);
         const int n2 = nrows * (HS_NEED / 2);
#pragma unroll 4
         for (int i = tid; i < n2; i += 256) dst[i] = src[i];
      }
      __syncthreads();
      for (int idx = jj0 * HS_PATCH + tid; idx < (jj1 + 1) * HS_PATCH; idx += 256) {
         const int jj = hs_div_small(idx, 1.0f / (float)HS_PATCH), ii = idx - jj * HS_PATCH;
         const int j = jj - (HS_PATCH >> 1), i = ii - (HS_PATCH >> 1);
         const float rx = c0 + (float)j * 0.0f, ry = c0 + (float)j * scale;
         float wx = rx + (float)i * scale, wy = ry + (float)i * 0.0f;
         const float fx = floorf(wx), fy = floorf(wy);
         wx -= fx; wy -= fy;
         const int y0 = min(max((int)fy, 0), P - 2);
         const float *tc = s_chunk + (y0 - ylo) * HS_NEED + 2 * ii;
         auto ld = [&](int dy) { return *reinterpret_cast<const v2f *>(tc + dy * HS_NEED); };
         const v2f q0 = ld(0), q1 = ld(1);
         const float kc = taps[r];
         v2f da = kc * q0, db = kc * q1;   // chains of window rows y0 and y0 + 1
         v2f pj = q1, mj = q0;             // rows y0 + j and y0 + 1 - j at j = 1
#pragma unroll 4
         for (int jt = 1; jt <= r; jt++) {
            const v2f pn = ld(jt + 1), mn = ld(-jt);
            const float kj = taps[r + jt];
            const v2f sa = pj + mn, sb = pn + mj;
            da += kj * sa;
            db += kj * sb;
            pj = pn;
            mj = mn;
         }
         s_patch[idx] = (1.0f - wy) * ((1.0f - wx) * da.x + wx * da.y) + (wy) * ((1.0f - wx) * db.x + wx * db.y);
      }
      __syncthreads();
      jj0 = jj1 + 1;
   }
}

// Gathers of NB x 64 consecutive window pixels (columns xb + lane + 64 it) of NR rows of one window, all issued before
// the first use, then stored into the rows' LDS lines (row i at srow + i * sstride, r border samples to the left).
// rc[i]: the row term of the tap coordinate; ctab: the window's column table in LDS, or nullptr (computed per tap).
template <int NR, int NB>
__device__ __forceinline__ void hs_gather_slots(const HsPlaneBuf &img, const v2f *rc, const v2f *__restrict__ ctab, float a11, float a21, int half,
                                                int xb, int P, int r, float *__restrict__ srow, int sstride)
{
   const int lane = threadIdx.x & 63, pm = P - 1;
   float v[NR][NB];
#pragma unroll
   for (int it = 0; it < NB; it++) {
      const int xx = min(xb + lane + 64 * it, pm);   // lanes past the row re-sample its last pixel (not stored)
      const v2f c = ctab ? ctab[xx] : hs_col_coord(a11, a21, xx - half);
#pragma unroll
      for (int i = 0; i < NR; i++) {
         const v2f w = rc[i] + c;
         v[i][it] = hs_tap_inside(img, w.x, w.y);
      }
   }
#pragma unroll
   for (int it = 0; it < NB; it++)
#pragma unroll
      for (int i = 0; i < NR; i++) HS_KEEP(v[i][it]);
#pragma unroll
   for (int it = 0; it < NB; it++) {
      const int xx = xb + lane + 64 * it;
      if (xx < P) {
#pragma unroll
         for (int i = 0; i < NR; i++) srow[i * sstride + r + xx] = v[i][it];
      }
   }
}

// all P pixels of NR rows: batches of NIT slots, and a last batch of exactly the slots that are left (a batch size is a
// compile-time constant: the remainder goes by its binary digits, at most three narrower batches; a test per slot inside
// one batch would serialise its gathers)
template <int NR, int NIT>
__device__ __forceinline__ void hs_gather_rows(const HsPlaneBuf &img, const v2f *rc, const v2f *__restrict__ ctab, float a11, float a21, int half,
                                               int P, int r, float *__restrict__ srow, int sstride)
{
   static_assert(NIT == 1 || NIT == 2 || NIT == 4 || NIT == 8, "power of two");
   int xb = 0;
   for (; ((P - xb + 63) >> 6) >= NIT; xb += 64 * NIT)
      hs_gather_slots<NR, NIT>(img, rc, ctab, a11, a21, half, xb, P, r, srow, sstride);
   const int rem = (P - xb + 63) >> 6;   // slots left, < NIT (wave-uniform)
   if (NIT > 4 && (rem & 4)) { hs_gather_slots<NR, 4>(img, rc, ctab, a11, a21, half, xb, P, r, srow, sstride); xb += 256; }
   if (NIT > 2 && (rem & 2)) { hs_gather_slots<NR, 2>(img, rc, ctab, a11, a21, half, xb, P, r, srow, sstride); xb += 128; }
   if (NIT > 1 && (rem & 1)) { hs_gather_slots<NR, 1>(img, rc, ctab, a11, a21, half, xb, P, r, srow, sstride); }
}

// one window row: warp (affine.cpp:126) into the wave's LDS row, then the row pass at the 82
// needed columns.  Called by all 64 lanes of a wave.  The LDS row is stored with r replicated
// border samples on either side (BORDER_REPLICATE), so the tap loop has no index clamps:
//   srow[r + x] = S[x],  srow[0..r) = S[0],  srow[r+P .. r+P+r) = S[P-1]     (needs P + 2r floats)
// The image gathers of NIT x 64 window pixels are issued together before any of them is used; taps are read from
// LDS (`taps`, broadcast reads).  ctab: the window's column table C[ii] = (i*a11, i*a21) in LDS, or nullptr
// (huge windows: computed per tap).
// Lane i < 41 owns the output pair (2i, 2i + 1); the two accumulation chains run as one packed chain.
template <int NIT, class TAPS = const float *>
__device__ __forceinline__ void hs_row_stream(const HsPlaneBuf &img, float x, float y, float a11, float a12, float a21, float a22, int P, int yy,
                                              float scale, const v2f *__restrict__ ctab, TAPS taps, int K,
                                              float *__restrict__ srow, float *__restrict__ out82, int pad_r = 0)
{
   const int lane = threadIdx.x & 63, half = P >> 1, pm = P - 1, r = K >> 1;
   const v2f rc = hs_row_coord(x, y, a12, a22, yy - half);
   hs_gather_rows<1, NIT>(img, &rc, ctab, a11, a21, half, P, r, srow, 0);
   HS_WAVE_LDS_SYNC();
   {
      const float first = srow[r], last = srow[r + pm];
      for (int i = lane; i < r; i += 64) { srow[i] = first; srow[r + P + i] = last; }
   }
   HS_WAVE_LDS_SYNC();
   // Lane i < 41 owns the output pair q = 2i, 2i + 1: the two blurred columns floor(w) and floor(w) + 1
   // that output pixel i of the 41x41 resample reads.  (0 <= floor(w) <= P - 2 always: |(i - 20) * scale| < P0 / 2.)
   if (lane < HS_PATCH) {
      const float c0 = (float)half;
      const float w = c0 + (float)(lane - 20) * scale;
      const int x0 = min(max((int)floorf(w), 0), pm - 1);
      const float *s = srow + x0;   // s[jt] = S[clamp(x0 - r + jt)],  s[jt + 1] = S[clamp(x0 + 1 - r + jt)]
      auto G = [&](int jt) { v2f g; g.x = s[jt]; g.y = s[jt + 1]; return g; };
      // RowFilter order; a window of this path has P0 >= 63, i.e. K = odd(int(9 * P0 / 41 + 1)) >= 15 (never the K <= 5 form)
      v2f t;
      if (HS_MID_SCALAR_TAIL) {   // scalar sliding-window chains (see hs_row_stream3)
         float a0 = s[0], a1 = s[1];
         const float k0 = taps[0];
         float t0 = k0 * a0, t1 = k0 * a1;
#pragma unroll 8
         for (int jt = 1; jt < K; jt++) {
            const float k = taps[jt];
            a0 = a1;
            a1 = s[jt + 1];
            t0 += k * a0; t1 += k * a1;
         }
         t.x = t0; t.y = t1;
      } else {
         t = taps[0] * G(0);
#pragma unroll 8
         for (int jt = 1; jt < K; jt++) t += taps[jt] * G(jt);
      }
      v2f *o = reinterpret_cast<v2f *>(out82) + lane;
      *o = t;
      // padded T' plane: the first / last window row is replicated pad_r times above / below (wave-uniform)
      if (pad_r > 0 && (yy == 0 || yy == pm)) {
         const int step = (yy == 0) ? -(HS_NEED / 2) : (HS_NEED / 2);
         for (int jr = 1; jr <= pad_r; jr++) o[jr * step] = t;
      }
   }
   HS_WAVE_LDS_SYNC();
}

// Two window rows at once (rows yyA and yyB of the same window, one LDS row each): the gathers of both rows are in
// flight together and the two row-pass chains interleave, so that neither the memory round trip nor the dependent
// accumulation of one row leaves the wavefront without work.  Same operations per row as hs_row_stream.
template <int NIT, class TAPS = const float *>
__device__ __forceinline__ void hs_row_stream2(const HsPlaneBuf &img, float x, float y, float a12, float a22, int P, int yyA, int yyB,
                                               float scale, const v2f *__restrict__ ctab, TAPS taps, int K,
                                               float *__restrict__ srowA, float *__restrict__ srowB, float *__restrict__ outA, float *__restrict__ outB, int pad_r,
                                               float a11 = 0.0f, float a21 = 0.0f)   // a11, a21: only read when ctab == nullptr
{
   const int lane = threadIdx.x & 63, half = P >> 1, pm = P - 1, r = K >> 1;
   const v2f rc[2] = {hs_row_coord(x, y, a12, a22, yyA - half), hs_row_coord(x, y, a12, a22, yyB - half)};
   hs_gather_rows<2, NIT>(img, rc, ctab, a11, a21, half, P, r, srowA, (int)(srowB - srowA));
   HS_WAVE_LDS_SYNC();
   {
      const float fA = srowA[r], lA = srowA[r + pm], fB = srowB[r], lB = srowB[r + pm];
      for (int i = lane; i < r; i += 64) { srowA[i] = fA; srowA[r + P + i] = lA; srowB[i] = fB; srowB[r + P + i] = lB; }
   }
   HS_WAVE_LDS_SYNC();
   if (lane < HS_PATCH) {
      const float c0 = (float)half;
      const float w = c0 + (float)(lane - 20) * scale;
      const int x0 = min(max((int)floorf(w), 0), pm - 1);
      const float *sA = srowA + x0, *sB = srowB + x0;
      auto GA = [&](int jt) { v2f g; g.x = sA[jt]; g.y = sA[jt + 1]; return g; };
      auto GB = [&](int jt) { v2f g; g.x = sB[jt]; g.y = sB[jt + 1]; return g; };
      v2f tA, tB;
      if (HS_MID_SCALAR_TAIL) {   // scalar sliding-window chains (see hs_row_stream3)
         float a0 = sA[0], a1 = sA[1], b0 = sB[0], b1 = sB[1];
         const float k0 = taps[0];
         float tA0 = k0 * a0, tA1 = k0 * a1, tB0 = k0 * b0, tB1 = k0 * b1;
#pragma unroll 4
         for (int jt = 1; jt < K; jt++) {
            const float k = taps[jt];
            a0 = a1; b0 = b1;
            a1 = sA[jt + 1]; b1 = sB[jt + 1];
            tA0 += k * a0; tA1 += k * a1;
            tB0 += k * b0; tB1 += k * b1;
         }
         tA.x = tA0; tA.y = tA1; tB.x = tB0; tB.y = tB1;
      } else {
         tA = taps[0] * GA(0); tB = taps[0] * GB(0);
#pragma unroll 4
         for (int jt = 1; jt < K; jt++) {
            const float k = taps[jt];
            tA += k * GA(jt);
            tB += k * GB(jt);
         }
      }
      v2f *oA = reinterpret_cast<v2f *>(outA) + lane, *oB = reinterpret_cast<v2f *>(outB) + lane;
      *oA = tA;
      *oB = tB;
      // padded T' plane: the first / last window row is replicated pad_r times above / below (wave-uniform)
      if (pad_r > 0 && yyA == 0)
         for (int jr = 1; jr <= pad_r; jr++) oA[-jr * (HS_NEED / 2)] = tA;
      if (pad_r > 0 && yyB == pm)
         for (int jr = 1; jr <= pad_r; jr++) oB[jr * (HS_NEED / 2)] = tB;
   }
   HS_WAVE_LDS_SYNC();
}

// Three window rows at once: the row pass produces 41 output pairs per row, i.e. 41 of a wavefront's 64 lanes in
// hs_row_stream / hs_row_stream2.  Three rows are 123 pair tasks = two per lane at 96 % lane use: task t = lane + 64 s
// (s = 0, 1) is pair t % 41 of row t / 41.  Every task is the same RowFilter chain as before (one lane, ascending taps);
// only the assignment of chains to lanes changes.  srow: three LDS rows `sstride` floats apart; out[i]: T' row of window
// row yy[i].
template <int NIT, class TAPS = const float *>
__device__ __forceinline__ void hs_row_stream3(const HsPlaneBuf &img, float x, float y, float a12, float a22, int P, int yy0, int yy1, int yy2,
                                               float scale, const v2f *__restrict__ ctab, TAPS taps, int K,
                                               float *__restrict__ srow, int sstride, float *__restrict__ out0, float *__restrict__ out1,
                                               float *__restrict__ out2, int pad_r, float a11 = 0.0f, float a21 = 0.0f)   // a11, a21: only read when ctab == nullptr
{
   const int lane = threadIdx.x & 63, half = P >> 1, pm = P - 1, r = K >> 1;
   const v2f rc[3] = {hs_row_coord(x, y, a12, a22, yy0 - half), hs_row_coord(x, y, a12, a22, yy1 - half),
                      hs_row_coord(x, y, a12, a22, yy2 - half)};
   float *srow1 = srow + sstride, *srow2 = srow + 2 * sstride;
   hs_gather_rows<3, NIT>(img, rc, ctab, a11, a21, half, P, r, srow, sstride);
   HS_WAVE_LDS_SYNC();
   {
      const float f0 = srow[r], l0 = srow[r + pm], f1 = srow1[r], l1 = srow1[r + pm], f2 = srow2[r], l2 = srow2[r + pm];
      for (int i = lane; i < r; i += 64) {
         srow[i] = f0; srow[r + P + i] = l0;
         srow1[i] = f1; srow1[r + P + i] = l1;
         srow2[i] = f2; srow2[r + P + i] = l2;
      }
   }
   HS_WAVE_LDS_SYNC();
   {
      // slot A: task lane (rows 0 and 1); slot B: task 64 + lane (rows 1 and 2; lanes 59..63 repeat task 122, not stored)
      const int tA = lane, tB = min(64 + lane, 3 * HS_PATCH - 1);
      const int rowA = tA >= HS_PATCH ? 1 : 0, rowB = tB >= 2 * HS_PATCH ? 2 : 1;
      const int colA = tA - HS_PATCH * rowA, colB = tB - HS_PATCH * rowB;
      const float c0 = (float)half;
      const float wA = c0 + (float)(colA - 20) * scale, wB = c0 + (float)(colB - 20) * scale;
#if HS_ABL_MIDCONF   // ablation only (results invalid): every lane starts at its own column, i.e. consecutive LDS words - what the row pass costs WITHOUT its bank conflicts
      const int xA = min(colA, pm - 1), xB = min(colB, pm - 1);
#else
      const int xA = min(max((int)floorf(wA), 0), pm - 1), xB = min(max((int)floorf(wB), 0), pm - 1);
#endif
      const float *sA = srow + rowA * sstride + xA, *sB = srow + rowB * sstride + xB;
      auto GA = [&](int jt) { v2f g; g.x = sA[jt]; g.y = sA[jt + 1]; return g; };
      auto GB = [&](int jt) { v2f g; g.x = sB[jt]; g.y = sB[jt + 1]; return g; };
      v2f tA2, tB2;
      if (HS_MID_SCALAR) {
         // four scalar chains on a sliding window: every sample of the two rows is read once (the pair form reads it twice, as the
         // second element of tap j and the first of tap j + 1) and no register pairs are assembled
         float a0 = sA[0], a1 = sA[1], b0 = sB[0], b1 = sB[1];
         const float k0 = taps[0];
         float tA0 = k0 * a0, tA1 = k0 * a1, tB0 = k0 * b0, tB1 = k0 * b1;
#pragma unroll 4
         for (int jt = 1; jt < K; jt++) {
            const float k = taps[jt];
            a0 = a1; b0 = b1;
            a1 = sA[jt + 1]; b1 = sB[jt + 1];
            tA0 += k * a0; tA1 += k * a1;
            tB0 += k * b0; tB1 += k * b1;
         }
         tA2.x = tA0; tA2.y = tA1; tB2.x = tB0; tB2.y = tB1;
      } else {
         tA2 = taps[0] * GA(0); tB2 = taps[0] * GB(0);
#pragma unroll 4
         for (int jt = 1; jt < K; jt++) {
            const float k = taps[jt];
            tA2 += k * GA(jt);
            tB2 += k * GB(jt);
         }
      }
      const int yA = rowA ? yy1 : yy0, yB = rowB == 2 ? yy2 : yy1;
      v2f *oA = reinterpret_cast<v2f *>(rowA ? out1 : out0) + colA;
      v2f *oB = reinterpret_cast<v2f *>(rowB == 2 ? out2 : out1) + colB;
      *oA = tA2;
      if (64 + lane < 3 * HS_PATCH) *oB = tB2;
      // padded T' plane: the first / last window row is replicated pad_r times above / below
      if (pad_r > 0) {
         if (yA == 0) for (int jr = 1; jr <= pad_r; jr++) oA[-jr * (HS_NEED / 2)] = tA2;
         if (yA == pm) for (int jr = 1; jr <= pad_r; jr++) oA[jr * (HS_NEED / 2)] = tA2;
         if (64 + lane < 3 * HS_PATCH) {
            if (yB == 0) for (int jr = 1; jr <= pad_r; jr++) oB[-jr * (HS_NEED / 2)] = tB2;
            if (yB == pm) for (int jr = 1; jr <= pad_r; jr++) oB[jr * (HS_NEED / 2)] = tB2;
         }
      }
   }
   HS_WAVE_LDS_SYNC();
}

// ---------------------------------------------------------------------------------------
// k_patch_mid: 64 < P <= 128 (bin 2) and 128 < P <= 512 (bin 3).  Each of the 4 waves streams window rows
// (warp -> row pass at the 82 needed columns) into T' (P x 82, padded with K/2 replicated rows above and below) in a
// per-block slot of HBM scratch (io.trows), written and re-read by the same block (L2-hot); then the resample
// evaluates the column pass where it reads.
// ---------------------------------------------------------------------------------------
#define HS_MID_PMAX 128
#define HS_MID_SROW 160   // 128 + 2 x 14 border samples, padded
#define HS_BIG_SROW 640   // 514 + 2 x 57 border samples, padded
#define HS_BIG_TAPS 128   // K <= 113 for P <= 512
#define HS_MID_RPAD 14    // K / 2 for P <= 128
#define HS_BIG_RPAD 57    // K / 2 for P <= 512
#ifndef HS_CHUNK_MIN_ROWS
#define HS_CHUNK_MIN_ROWS 4   // measured 1 / 4 / 7 / 13: 15.4 / 15.0 / 15.5 / 16.4 ms (bin 3, 32 UHD images)
#endif
#ifndef HS_MID_NIT3_BIG
#define HS_MID_NIT3_BIG 2
#endif
#ifndef HS_MID_NIT3_SMALL
#define HS_MID_NIT3_SMALL 1
#endif
#define HS_MID_BLOCKS (256 * 7)   // persistent grids of the row-streamed bins: one T' slot per block
#define HS_BIG_BLOCKS (256 * 8)

template <int PMAX> struct MidGeom {
   static constexpr bool BIG = PMAX > HS_MID_PMAX;
   static constexpr int SROW = BIG ? HS_BIG_SROW : HS_MID_SROW;
   static constexpr int NTAP = BIG ? HS_BIG_TAPS : 32;
   static constexpr int RPAD = BIG ? HS_BIG_RPAD : HS_MID_RPAD;
   // rows of T' the column pass stages per round (the LDS lines of the window rows are free by then and are reused)
   static constexpr int CHUNK_MIN = BIG ? 0 : 64;
   static constexpr int SROWS = 12 * SROW > CHUNK_MIN * HS_NEED ? 12 * SROW : CHUNK_MIN * HS_NEED;
   static constexpr int CHUNK_ROWS = SROWS / HS_NEED;
   // s_patch | taps | C table (float2 x (PMAX + 2)) | 4 waves x 3 rows, later the T' chunk
   static constexpr int FLOATS = HS_PATCH_ARR + NTAP + 2 * (PMAX + 2) + SROWS;
};

template <int PMAX>
__global__ __launch_bounds__(256, HS_MID_WAVES) void k_patch_mid(HessList hl, PatchWork pw, PatchIO io, KpTables tb)
{
   typedef MidGeom<PMAX> GM;
   constexpr int BIN = GM::BIG ? 3 : 2;
   constexpr int NIT = GM::BIG ? 4 : 2;
   constexpr int NIT3 = GM::BIG ? HS_MID_NIT3_BIG : HS_MID_NIT3_SMALL;   // gathers in flight per row of the three-row form
   extern __shared__ __attribute__((aligned(16))) float smem[];
   float *s_patch = smem;
   float *s_taps = s_patch + HS_PATCH_ARR;
   v2f *s_C = reinterpret_cast<v2f *>(s_taps + GM::NTAP);
   float *s_srow = reinterpret_cast<float *>(s_C + (PMAX + 2));   // 4 waves x 3 rows x SROW
   float *Tp = io.trows + (size_t)blockIdx.x * ((size_t)(PMAX + 2 * GM::RPAD) * HS_NEED);

   const int tid = threadIdx.x, wave = tid >> 6;
   const uint32_t cnt = min(pw.bin_count[BIN], pw.cap);
   const int imPitch = io.image.pitch;

   // Dynamic scheduling: a block claims the next item of the bin's list when it is free, so blocks that draw cheap
   // windows (the cost of an item grows with P^2: 16x across a bin) take more of them instead of idling at the end of
   // the launch.  (The two LDS-window bins stride statically: uniform cost, and 2 M claims on one counter would serialise.)
   __shared__ uint32_t s_item;
   for (;;) {
      if (tid == 0) s_item = atomicAdd(pw.bin_work + BIN, 1u);
      __syncthreads();
      const uint32_t wi = HS_TAPS_SCALAR ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item) : s_item;
      if (wi >= cnt) break;
      const uint32_t h = pw.bin_items[(size_t)BIN * pw.cap + wi];
      const int b = hl.meta[h] >> 8;
      const HsPlaneBuf ib = hs_plane_buf(io.image.img(b), io.image.rows, imPitch);
      const float x = hl.x[h], y = hl.y[h];
      const float a11 = pw.A[4 * h], a12 = pw.A[4 * h + 1], a21 = pw.A[4 * h + 2], a22 = pw.A[4 * h + 3];
      const int P0 = pw.P0[h], P = P0 + 2, half = P >> 1;
      const float scale = (float)P0 / (float)HS_PATCH;
      const int K = tb.patch_tap_k[(P0 - 1) >> 1];
      const float *taps_g = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
      const hs_row_taps ktap = HS_ROW_TAPS(taps_g, s_taps);   // the row pass's taps: scalar loads from the table, or the LDS copy
      if (!HS_TAPS_SCALAR && tid < K) s_taps[tid] = taps_g[tid];
      for (int m = tid; m < P; m += 256) s_C[m] = hs_col_coord(a11, a21, m - half);
      __syncthreads();
      float *srowA = s_srow + wave * 3 * GM::SROW, *srowB = srowA + GM::SROW;
#pragma unroll 1
      for (int yy = wave; yy < P; yy += 12) {
         // rows yy, yy + 4 and yy + 8 of this wavefront together; the last one or two rows in the narrower forms
         const int rr = K >> 1;
         if (yy + 8 < P)
            hs_row_stream3<NIT3, hs_row_taps>(ib, x, y, a12, a22, P, yy, yy + 4, yy + 8, scale, s_C, ktap, K, srowA, GM::SROW,
                                 Tp + (size_t)(yy + rr) * HS_NEED, Tp + (size_t)(yy + 4 + rr) * HS_NEED, Tp + (size_t)(yy + 8 + rr) * HS_NEED, rr);
         else if (yy + 4 < P)
            hs_row_stream2<NIT, hs_row_taps>(ib, x, y, a12, a22, P, yy, yy + 4, scale, s_C, ktap, K, srowA, srowB,
                                Tp + (size_t)(yy + (K >> 1)) * HS_NEED, Tp + (size_t)(yy + 4 + (K >> 1)) * HS_NEED, K >> 1);
         else
            hs_row_stream<NIT, hs_row_taps>(ib, x, y, a11, a12, a21, a22, P, yy, scale, s_C, ktap, K, srowA,
                               Tp + (size_t)(yy + (K >> 1)) * HS_NEED, K >> 1);
      }
      __syncthreads();   // workgroup-scope release/acquire: the T' rows of all four waves are visible
      // staged column pass when a round holds at least HS_CHUNK_MIN_ROWS output rows (a round of one or two rows leaves
      // most of the block's threads idle: 41 outputs per row); otherwise straight from the slot
      if ((float)(K + 1) + (float)(HS_CHUNK_MIN_ROWS - 1) * scale <= (float)GM::CHUNK_ROWS)
         hs_resample_chunked<hs_row_taps>(Tp + (K >> 1) * HS_NEED, P, scale, ktap, K >> 1, s_srow, GM::CHUNK_ROWS, s_patch);
      else
         hs_resample_reduced_batched<true, hs_row_taps>(Tp + (K >> 1) * HS_NEED, P, scale, ktap, K >> 1, s_patch);
      __syncthreads();
      for (int i = tid; i < HS_PATCH_PIX; i += 256) {
         float *po = io.patches + (size_t)(h - io.h_base) * HS_PATCH_PIX + i;
         if (HS_NT_PATCH) hs_store_nt(po, s_patch[i]); else *po = s_patch[i];
      }
      __syncthreads();
   }
}

// ---------------------------------------------------------------------------------------
// Large windows (P > 512: well under 1 % of the keypoints, ~15 % of all window pixels).
// k_large_prefix: exclusive prefix of the window sides over the bin's items = first T' row of every item.
// k_patch_large_rows: grid-stride over ALL window rows of the group's huge keypoints, one wavefront per chunk
//   of rows (binary search of the row id in the prefix); writes T' rows to HBM.
// k_patch_large_finish: one block per keypoint: column pass at the resample taps.
// The item count and the prefix stay on the device; the host only supplies an upper bound of the rows
// (k_image_large_rows) to size the T' buffer.
// ---------------------------------------------------------------------------------------
#ifndef HS_LARGE_CHUNK
#define HS_LARGE_CHUNK 18   // consecutive window rows per wavefront task (a multiple of three: the three-row form below)
#endif
#ifndef HS_LARGE_NIT3
#define HS_LARGE_NIT3 2     // gathers in flight per row of the three-row form
#endif

// dynamic LDS: per wave  nrow x srow_stride floats (window rows + borders)  +  tap_stride floats (taps); blocks of 4, 2 or 1 wavefronts
// (the host picks the largest count whose rows fit the CU's 160 KB).
// nrow = 3: three consecutive window rows per step (hs_row_stream3: one tap broadcast and one column coordinate serve three rows, 123 pair
// chains on 64 lanes instead of 41 - a third fewer vector and two thirds fewer LDS instructions per row than the one-row form); nrow = 1: the
// one-row form, for windows whose three rows do not fit.  Items with P outside (p_lo, p_hi] are skipped: a batch whose largest window is far
// above the common ones runs as two launches, so that the 513..1024 windows do not live with the LDS (= occupancy) of one 2800-pixel outlier.
__global__ __launch_bounds__(256) void k_patch_large_rows(HessList hl, PatchWork pw, PatchIO io, KpTables tb, int srow_stride, int tap_stride, int nrow,
                                                          int p_lo, int p_hi)
{
   extern __shared__ __attribute__((aligned(16))) float smem[];
   const int wave = HS_TAPS_SCALAR ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
   float *srow = smem + (size_t)wave * (nrow * srow_stride + tap_stride);
   float *stap = srow + nrow * srow_stride;
   const uint32_t *pre = io.row_prefix;
   const uint32_t n_items = min(pw.bin_count[HS_NBINS - 1], pw.cap);
   if (n_items == 0) return;
   const uint32_t row_hi = pre[n_items];
   if (row_hi > io.trows_cap) {   // the host sized the buffer from an upper bound: cannot happen, but never write past it
      if (threadIdx.x == 0 && blockIdx.x == 0) *io.overflow = 1u;
      return;
   }
   const int imPitch = io.image.pitch;
   const uint32_t ntasks = (row_hi + HS_LARGE_CHUNK - 1) / HS_LARGE_CHUNK;
   const uint32_t nw = blockDim.x >> 6;   // wavefronts per block: 4, or fewer when the rows of the launch's largest window need more than a quarter of the LDS
   for (uint32_t task = blockIdx.x * nw + wave; task < ntasks; task += gridDim.x * nw) {
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
         const int P0 = pw.P0[h], P = P0 + 2;
         if (P <= p_lo || P > p_hi) { row = it_rows_end; it++; continue; }   // another launch's window (wave-uniform)
         const int b = hl.meta[h] >> 8;
         const float scale = (float)P0 / (float)HS_PATCH;
         const int K = tb.patch_tap_k[(P0 - 1) >> 1];
         const float *taps = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
         const HsPlaneBuf ib = hs_plane_buf(io.image.img(b), io.image.rows, imPitch);
         const float kx = hl.x[h], ky = hl.y[h];
         const float a11 = pw.A[4 * h], a12 = pw.A[4 * h + 1], a21 = pw.A[4 * h + 2], a22 = pw.A[4 * h + 3];
         const uint32_t first = pre[it];
         // this item's taps -> the wave's LDS tap buffer (broadcast reads in the tap loop); with HS_TAPS_SCALAR they are read from the table itself
         if (!HS_TAPS_SCALAR) {
            for (int i = threadIdx.x & 63; i < K; i += 64) stap[i] = taps[i];
            HS_WAVE_LDS_SYNC();
         }
         const hs_row_taps ktap = HS_ROW_TAPS(taps, stap);
         float *const out0 = io.trows + (size_t)row * HS_NEED;
         const int yy0 = (int)(row - first), nr = (int)(it_rows_end - row);
         int d = 0;
         if (nrow == 3) {
#pragma unroll 1
            for (; d + 3 <= nr; d += 3)
               hs_row_stream3<HS_LARGE_NIT3, hs_row_taps>(ib, kx, ky, a12, a22, P, yy0 + d, yy0 + d + 1, yy0 + d + 2, scale, nullptr, ktap, K, srow, srow_stride,
                                             out0 + (size_t)d * HS_NEED, out0 + (size_t)(d + 1) * HS_NEED, out0 + (size_t)(d + 2) * HS_NEED, 0, a11, a21);
            if (d + 2 <= nr) {
               hs_row_stream2<HS_LARGE_NIT3, hs_row_taps>(ib, kx, ky, a12, a22, P, yy0 + d, yy0 + d + 1, scale, nullptr, ktap, K, srow, srow + srow_stride,
                                             out0 + (size_t)d * HS_NEED, out0 + (size_t)(d + 1) * HS_NEED, 0, a11, a21);
               d += 2;
            }
         }
#pragma unroll 1
         for (; d < nr; d++)
            hs_row_stream<8, hs_row_taps>(ib, kx, ky, a11, a12, a21, a22, P, yy0 + d, scale, nullptr, ktap, K, srow, out0 + (size_t)d * HS_NEED);
         row = it_rows_end;
         it++;
      }
   }
}

__global__ __launch_bounds__(256) void k_patch_large_finish(PatchWork pw, PatchIO io, KpTables tb)
{
   __shared__ float s_patch[HS_PATCH_ARR];
   const uint32_t *pre = io.row_prefix;
   const uint32_t n_items = min(pw.bin_count[HS_NBINS - 1], pw.cap);
   if (n_items == 0 || pre[n_items] > io.trows_cap) return;
   for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
      const uint32_t h = pw.bin_items[(size_t)(HS_NBINS - 1) * pw.cap + it];
      const int P0 = pw.P0[h], P = P0 + 2;
      const float scale = (float)P0 / (float)HS_PATCH;
      const int K = tb.patch_tap_k[(P0 - 1) >> 1];
      const float *taps = tb.patch_taps + tb.patch_tap_off[(P0 - 1) >> 1];
      hs_resample_reduced_batched<false, hs_row_taps>(io.trows + (size_t)pre[it] * HS_NEED, P, scale, HS_ROW_TAPS(taps, taps), K >> 1, s_patch);
      __syncthreads();
      for (int i = threadIdx.x; i < HS_PATCH_PIX; i += 256) {
         float *po = io.patches + (size_t)(h - io.h_base) * HS_PATCH_PIX + i;
         if (HS_NT_PATCH) hs_store_nt(po, s_patch[i]); else *po = s_patch[i];
      }
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


// ---------------------------------------------------------------------------------------------------------------------
// k_patch_pyramid (hesaff_params.fast = 2 only; a DIFFERENT ALGORITHM from the reference's for these windows):
// normalizeAffine (affine.cpp:102-144) warps a P x P window of the ORIGINAL image at unit spacing, blurs it with
// sigma = 1.5 * P0 / 41 and takes every (P0 / 41)-th sample.  For P > 41 that is P^2 bilinear taps and a K = 0.22 P tap
// separable blur per keypoint - bins 2-4 (15 % of the keypoints) cost 60 % of the patch stage.  The scale space already
// holds the image blurred with sigma = 1.6 * 2^(octave + level / 3): this kernel takes the 41 x 41 samples straight
// from the stored level whose blur is closest (in log scale) to 1.5 * P0 / 41, at the affine-warped positions: 1681 taps,
// no blur.  What it ignores: the reference's blur is isotropic in the NORMALISED frame (anisotropic in the image for an
// elongated region), the level's blur is isotropic in the image; and the blur matches only to +-12 %.
// One 256-thread block per keypoint, grid-stride over the lists of bins first_bin .. HS_NBINS-1.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_patch_pyramid(HessList hl, PatchWork pw, PatchIO io, PlaneTab pt, int n_octaves, float pd0, int first_bin)
{
   const int tid = threadIdx.x;
   uint32_t base = 0;
   for (int bin = first_bin; bin < HS_NBINS; bin++) {
      const uint32_t cnt = min(pw.bin_count[bin], pw.cap);
      // blocks continue round-robin across the bins' lists
      for (uint32_t wi = (blockIdx.x + gridDim.x - base % gridDim.x) % gridDim.x; wi < cnt; wi += gridDim.x) {
         const uint32_t h = pw.bin_items[(size_t)bin * pw.cap + wi];
         const int b = hl.meta[h] >> 8;
         const float x = hl.x[h], y = hl.y[h];
         const float a11 = pw.A[4 * h], a12 = pw.A[4 * h + 1], a21 = pw.A[4 * h + 2], a22 = pw.A[4 * h + 3];
         const float scale = (float)pw.P0[h] / (float)HS_PATCH;
         // level with sigma closest to 1.5 * scale: sigma(o, l) = 1.6 * 2^(o + l / 3) * pd0 in pixels of the original image
         // (pd0 = 0.5 with upscaleInputImage: the stored planes are those of the 2x up-sampled image)
         const float t = 3.0f * log2f(fmaxf(1.5f * scale / (1.6f * pd0), 1.0f));        // octave * 3 + level, fractional
         int ol = (int)(t + 0.5f);
         ol = max(0, min(ol, 3 * n_octaves - 1));
         const int o = ol / 3, l = ol - 3 * o;
         const DPlane &P = pt.L[o][l];
         const float *img = P.img(b);
         const float pd = pd0 * (float)(1 << o), inv = 1.0f / pd;
         const float xmax = (float)(P.cols - 1) - 0.001f, ymax = (float)(P.rows - 1) - 0.001f;
         float *out = io.patches + (size_t)(h - io.h_base) * HS_PATCH_PIX;
         const float b11 = a11 * scale * inv, b12 = a12 * scale * inv, b21 = a21 * scale * inv, b22 = a22 * scale * inv;
         const float ox = x * inv, oy = y * inv;
#pragma unroll
         for (int q = 0; q < HS_PATCH_PIX_IT; q++) {
            const int idx = tid + 256 * q;
            if (idx < HS_PATCH_PIX) {
               const int jj = idx / HS_PATCH, ii = idx - jj * HS_PATCH;
               const float i = (float)(ii - (HS_PATCH >> 1)), j = (float)(jj - (HS_PATCH >> 1));
               // helpers.cpp:209-244 argument order: a12 / a22 multiply the row index j, a11 / a21 the column index i
               float wx = ox + j * b12 + i * b11, wy = oy + j * b22 + i * b21;
               wx = fminf(fmaxf(wx, 0.0f), xmax); wy = fminf(fmaxf(wy, 0.0f), ymax);
               const float fx = floorf(wx), fy = floorf(wy);
               wx -= fx; wy -= fy;
               const float *p = img + (long long)(int)fy * P.pitch + (int)fx;
               const float p00 = p[0], p01 = p[1], p10 = p[P.pitch], p11 = p[P.pitch + 1];
               out[idx] = (1.0f - wy) * ((1.0f - wx) * p00 + wx * p01) + wy * ((1.0f - wx) * p10 + wx * p11);
            }
         }
      }
      base += cnt;
   }
}
