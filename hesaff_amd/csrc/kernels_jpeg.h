// kernels_jpeg.h -- the device half of cv::imread (hesaff.cpp:137) for JPEG files: entropy-decoded coefficients
// (hesaff_read_jpeg_coefficients, jpeg_decode.cpp) -> the bytes libjpeg returns at imread's settings.
//
// A JPEG decoder has one part that must run in order - the Huffman bit stream - and three that are the same small
// integer function at every block or pixel: the inverse DCT (jidctint.c, JDCT_ISLOW), the chroma up-sampling
// (jdsample.c "fancy" triangle filters) and the colour conversion (jdcolor.c).  On a host thread the three cost 60 % of
// the decode time of a 4:2:0 photograph; here they run over all images of a chunk at once:
//    k_jpeg_idct    one thread per 8 x 8 block of any component of any image (coefficients staged through LDS as whole cache
//                   lines): dequantise, two 1-D passes in 32-bit integers (wrapping, like libjpeg's INT32), range limit, 8 rows of
//                   8 samples into the component plane
//    k_jpeg_pixels  one thread per 4 output pixels: up-sample the components that are not at full resolution (edge
//                   samples replicated), YCbCr -> RGB with jdcolor.c's fixed-point constants, bytes into the chunk's
//                   input slot in the layout of a raw 1- or 3-channel image - the grey conversion (hesaff.cpp:138-148) and
//                   everything after it then run exactly as for a PPM file.
// Integer arithmetic only: the bytes equal hesaff_read_jpeg's (tests/test_gpu_parity.py), which equal libjpeg-turbo's
// (tests/test_host_side.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_common.h"

struct JpegGeom {
   int W, H, nc;
   int bw[3], bh[3], cw[3], chgt[3], hx[3], vx[3], mode[3];
   unsigned long long coef_off[3];    // bytes from the start of an image's blob
   unsigned long long plane_off[3];   // bytes from the start of an image's planes
   unsigned long long blob_bytes, plane_bytes;
   unsigned int blocks[3], blocks_per_image;
};
enum { JPEG_UP_NONE = 0, JPEG_UP_H2V1 = 1, JPEG_UP_H2V2 = 2, JPEG_UP_H1V2 = 3, JPEG_UP_REPLICATE = 4 };

__device__ __forceinline__ int hs_jpeg_descale(int x, int n) { return (int)((unsigned)x + (1u << (n - 1))) >> n; }
// range_limit[x & RANGE_MASK] of libjpeg: a 10-bit window around the level shift, clamped to 0..255
__device__ __forceinline__ int hs_jpeg_limit(int x)
{
   int v = x & 1023;
   if (v >= 512) v -= 1024;
   v += 128;
   return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one 1-D pass of jidctint.c on eight values (32-bit two's complement, wrapping)
__device__ __forceinline__ void hs_jpeg_idct8(const int d[8], int out[8], int shift)
{
   constexpr int CONST_BITS = 13;
   constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                 F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                 F_2_562915447 = 20995, F_3_072711026 = 25172;
   auto mul = [](int a, int b) { return (int)((unsigned)a * (unsigned)b); };
   auto add = [](int a, int b) { return (int)((unsigned)a + (unsigned)b); };
   auto sub = [](int a, int b) { return (int)((unsigned)a - (unsigned)b); };
   int z2 = d[2], z3 = d[6];
   int z1 = mul(add(z2, z3), F_0_541196100);
   int tmp2 = add(z1, mul(z3, -F_1_847759065));
   int tmp3 = add(z1, mul(z2, F_0_765366865));
   int tmp0 = mul(add(d[0], d[4]), 1 << CONST_BITS);
   int tmp1 = mul(sub(d[0], d[4]), 1 << CONST_BITS);
   const int tmp10 = add(tmp0, tmp3), tmp13 = sub(tmp0, tmp3), tmp11 = add(tmp1, tmp2), tmp12 = sub(tmp1, tmp2);
   tmp0 = d[7]; tmp1 = d[5]; tmp2 = d[3]; tmp3 = d[1];
   z1 = add(tmp0, tmp3); z2 = add(tmp1, tmp2); z3 = add(tmp0, tmp2);
   int z4 = add(tmp1, tmp3);
   const int z5 = mul(add(z3, z4), F_1_175875602);
   tmp0 = mul(tmp0, F_0_298631336); tmp1 = mul(tmp1, F_2_053119869); tmp2 = mul(tmp2, F_3_072711026); tmp3 = mul(tmp3, F_1_501321110);
   z1 = mul(z1, -F_0_899976223); z2 = mul(z2, -F_2_562915447); z3 = mul(z3, -F_1_961570560); z4 = mul(z4, -F_0_390180644);
   z3 = add(z3, z5); z4 = add(z4, z5);
   tmp0 = add(add(tmp0, z1), z3); tmp1 = add(add(tmp1, z2), z4); tmp2 = add(add(tmp2, z2), z3); tmp3 = add(add(tmp3, z1), z4);
   out[0] = hs_jpeg_descale(add(tmp10, tmp3), shift); out[7] = hs_jpeg_descale(sub(tmp10, tmp3), shift);
   out[1] = hs_jpeg_descale(add(tmp11, tmp2), shift); out[6] = hs_jpeg_descale(sub(tmp11, tmp2), shift);
   out[2] = hs_jpeg_descale(add(tmp12, tmp1), shift); out[5] = hs_jpeg_descale(sub(tmp12, tmp1), shift);
   out[3] = hs_jpeg_descale(add(tmp13, tmp0), shift); out[4] = hs_jpeg_descale(sub(tmp13, tmp0), shift);
}

// Blocks are numbered image by image, component by component, row by row: block t of the chunk lies at
//    blobs + (t / blocks_per_image) * blob_bytes + HEADER + (t % blocks_per_image) * 128.
// A wavefront takes 64 consecutive blocks.  Their coefficients (8 KB, contiguous inside an image) are read as whole cache lines - lane l
// reads 16-byte piece l of every KB - into LDS, from where every lane takes the eight rows of its own block (a lane reading its block
// straight from memory touches 64 lines per load instruction and the same lines eight times).  Alone on the device: 0.39 ms per 32 UHD
// 4:2:0 images = 4.6 TB/s of coefficients in and samples out; in the file pipeline it usually runs beside the copy-out of the previous
// chunk's rows - on this pool a blit kernel on the high-priority copy queue - and then takes 3.0 ms (profiles/r04_notes.md).
// grid-stride over B * blocks_per_image blocks in steps of 256; block 256
#define JPEG_LDS_ROW 144   // bytes per block in LDS: 128 + 16, so that the 64 lanes' 16-byte reads spread over the banks
__global__ __launch_bounds__(256) void k_jpeg_idct(const uint8_t *__restrict__ blobs, uint8_t *__restrict__ planes, JpegGeom g, int B)
{
   __shared__ __attribute__((aligned(16))) uint8_t s_cf[4][64 * JPEG_LDS_ROW];
   const unsigned long long total = (unsigned long long)B * g.blocks_per_image;
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   uint8_t *lds = s_cf[wv];
   for (unsigned long long t0 = ((unsigned long long)blockIdx.x * 4 + wv) * 64; t0 < total; t0 += (unsigned long long)gridDim.x * 256) {
      // image and block-in-image of the wavefront's first block (wave-uniform)
      const unsigned int b0 = (unsigned int)(t0 / g.blocks_per_image);
      const unsigned int k0 = (unsigned int)(t0 - (unsigned long long)b0 * g.blocks_per_image);
#pragma unroll
      for (int r = 0; r < 8; r++) {
         const unsigned int idx = (unsigned int)(r * 8 + (lane >> 3));   // block of the wavefront this piece belongs to
         unsigned int bb = b0, kk = k0 + idx;
         while (kk >= g.blocks_per_image) { kk -= g.blocks_per_image; bb++; }   // at most once unless an image has fewer than 64 blocks
         int4 v = make_int4(0, 0, 0, 0);
         if (t0 + idx < total) v = *reinterpret_cast<const int4 *>(blobs + (unsigned long long)bb * g.blob_bytes + HESAFF_JPEG_BLOB_HEADER + (unsigned long long)kk * 128 + (lane & 7) * 16);
         *reinterpret_cast<int4 *>(lds + idx * JPEG_LDS_ROW + (lane & 7) * 16) = v;
      }
      HS_WAVE_LDS_SYNC();   // (the stores wait for their loads through the register dependence)
      const unsigned long long t = t0 + lane;
      if (t < total) {
         unsigned int b = b0, k = k0 + (unsigned int)lane;
         while (k >= g.blocks_per_image) { k -= g.blocks_per_image; b++; }
         int c = 0;
         if (k >= g.blocks[0]) { k -= g.blocks[0]; c = 1; if (k >= g.blocks[1]) { k -= g.blocks[1]; c = 2; } }
         const uint16_t *q = reinterpret_cast<const uint16_t *>(blobs + (unsigned long long)b * g.blob_bytes) + 64 * c;
         const int4 *cf = reinterpret_cast<const int4 *>(lds + lane * JPEG_LDS_ROW);
         int ws[64];
         // pass 1: columns, dequantisation inside (|int16 x uint16| < 2^31), results scaled up by PASS1_BITS
         {
            int in[64];
#pragma unroll
            for (int r = 0; r < 8; r++) {
               const int4 v = cf[r];   // eight int16 of row r
               const int4 qq = reinterpret_cast<const int4 *>(q)[r];
               const int vv[4] = {v.x, v.y, v.z, v.w}, qv[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
               for (int j = 0; j < 4; j++) {
                  in[8 * r + 2 * j] = (int)(short)(vv[j] & 0xffff) * (int)(qv[j] & 0xffff);
                  in[8 * r + 2 * j + 1] = (int)(short)((unsigned)vv[j] >> 16) * (int)((unsigned)qv[j] >> 16);
               }
            }
#pragma unroll
            for (int col = 0; col < 8; col++) {
               const int d[8] = {in[col], in[8 + col], in[16 + col], in[24 + col], in[32 + col], in[40 + col], in[48 + col], in[56 + col]};
               int o[8];
               hs_jpeg_idct8(d, o, 13 - 2);
#pragma unroll
               for (int r = 0; r < 8; r++) ws[8 * r + col] = o[r];
            }
         }
         // pass 2: rows, descale by CONST_BITS + PASS1_BITS + 3, level shift and range limit
         const int bx = (int)(k % (unsigned)g.bw[c]), by = (int)(k / (unsigned)g.bw[c]);
         const int stride = g.bw[c] * 8;
         uint8_t *dst = planes + (unsigned long long)b * g.plane_bytes + g.plane_off[c] + (unsigned long long)(by * 8) * stride + bx * 8;
#pragma unroll
         for (int r = 0; r < 8; r++) {
            const int d[8] = {ws[8 * r], ws[8 * r + 1], ws[8 * r + 2], ws[8 * r + 3], ws[8 * r + 4], ws[8 * r + 5], ws[8 * r + 6], ws[8 * r + 7]};
            int o[8];
            hs_jpeg_idct8(d, o, 13 + 2 + 3);
            uint2 w;
            w.x = (unsigned)hs_jpeg_limit(o[0]) | ((unsigned)hs_jpeg_limit(o[1]) << 8) | ((unsigned)hs_jpeg_limit(o[2]) << 16) | ((unsigned)hs_jpeg_limit(o[3]) << 24);
            w.y = (unsigned)hs_jpeg_limit(o[4]) | ((unsigned)hs_jpeg_limit(o[5]) << 8) | ((unsigned)hs_jpeg_limit(o[6]) << 16) | ((unsigned)hs_jpeg_limit(o[7]) << 24);
            *reinterpret_cast<uint2 *>(dst + (unsigned long long)r * stride) = w;
         }
      }
      HS_WAVE_LDS_SYNC();   // the next round's pieces overwrite what this round's lanes read
   }
}

// the value of an up-sampled component at output pixel (ox, oy): jdsample.c's method for the component's ratio (jpeg_decode.cpp
// upsample_*), pl = the component's plane (w x h samples used, row stride `stride`)
__device__ __forceinline__ int hs_jpeg_sample(const uint8_t *__restrict__ pl, int stride, int w, int h, int mode, int hx, int vx, int ox, int oy)
{
   if (mode == JPEG_UP_NONE) return pl[(unsigned long long)oy * stride + ox];
   if (mode == JPEG_UP_H2V1) {
      const uint8_t *s = pl + (unsigned long long)oy * stride;
      const int x = ox >> 1;
      if (ox == 0) return s[0];
      if (ox == 2 * w - 1) return s[w - 1];
      return (ox & 1) ? (s[x] * 3 + s[x + 1] + 2) >> 2 : (s[x] * 3 + s[x - 1] + 1) >> 2;
   }
   if (mode == JPEG_UP_H2V2) {
      const int y = oy >> 1;
      int yf = (oy & 1) ? y + 1 : y - 1;
      yf = yf < 0 ? 0 : (yf > h - 1 ? h - 1 : yf);
      const uint8_t *s0 = pl + (unsigned long long)(y > h - 1 ? h - 1 : y) * stride, *s1 = pl + (unsigned long long)yf * stride;
      const int x = ox >> 1;
      const int thiscol = s0[x] * 3 + s1[x];
      if (ox == 0) return (thiscol * 4 + 8) >> 4;
      if (ox == 2 * w - 1) return (thiscol * 4 + 7) >> 4;
      if (ox & 1) return (thiscol * 3 + (s0[x + 1] * 3 + s1[x + 1]) + 7) >> 4;
      return (thiscol * 3 + (s0[x - 1] * 3 + s1[x - 1]) + 8) >> 4;
   }
   if (mode == JPEG_UP_H1V2) {
      const int y = (oy >> 1) > h - 1 ? h - 1 : (oy >> 1);
      int yf = (oy & 1) ? y + 1 : y - 1;
      yf = yf < 0 ? 0 : (yf > h - 1 ? h - 1 : yf);
      return (pl[(unsigned long long)y * stride + ox] * 3 + pl[(unsigned long long)yf * stride + ox] + ((oy & 1) ? 2 : 1)) >> 2;
   }
   const int y = (oy / vx) > h - 1 ? h - 1 : (oy / vx), x = (ox / hx) > w - 1 ? w - 1 : (ox / hx);
   return pl[(unsigned long long)y * stride + x];
}

__device__ __forceinline__ int hs_clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// grid (ceil(ceil(W / 4) / 256), H, B), block 256: thread = four pixels of a row
__global__ __launch_bounds__(256) void k_jpeg_pixels(const uint8_t *__restrict__ blobs, const uint8_t *__restrict__ planes, uint8_t *__restrict__ out,
                                                     JpegGeom g, unsigned long long out_img_stride)
{
   const int x0 = 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y, b = blockIdx.z;
   if (x0 >= g.W) return;
   const uint8_t *pl = planes + (unsigned long long)b * g.plane_bytes;
   uint8_t *o = out + (unsigned long long)b * out_img_stride + ((unsigned long long)y * g.W + x0) * g.nc;
   const int n = min(4, g.W - x0);
   if (g.nc == 1) {
      for (int i = 0; i < n; i++) o[i] = pl[(unsigned long long)y * (g.bw[0] * 8) + x0 + i];
      return;
   }
   const int ycc = *reinterpret_cast<const int *>(blobs + (unsigned long long)b * g.blob_bytes + 384);
   const bool words = n == 4 && (g.W & 3) == 0 && (out_img_stride & 3) == 0;   // the thread's 12 bytes are three aligned words
   unsigned int pk[3] = {0u, 0u, 0u};
#pragma unroll
   for (int i = 0; i < 4; i++) {
      if (i >= n) break;
      int s[3];
#pragma unroll
      for (int c = 0; c < 3; c++) s[c] = hs_jpeg_sample(pl + g.plane_off[c], g.bw[c] * 8, g.cw[c], g.chgt[c], g.mode[c], g.hx[c], g.vx[c], x0 + i, y);
      int r = s[0], gg = s[1], bb = s[2];
      if (ycc) {   // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert, SCALEBITS 16
         const int Y = s[0], cb = s[1] - 128, cr = s[2] - 128;
         r = hs_clamp8(Y + ((91881 * cr + 32768) >> 16));
         gg = hs_clamp8(Y + ((-22554 * cb + 32768 + -46802 * cr) >> 16));
         bb = hs_clamp8(Y + ((116130 * cb + 32768) >> 16));
      }
      if (words) {
         const unsigned int v[3] = {(unsigned)r, (unsigned)gg, (unsigned)bb};
#pragma unroll
         for (int k = 0; k < 3; k++) { const int at = 3 * i + k; pk[at >> 2] |= v[k] << (8 * (at & 3)); }
      } else {
         o[3 * i] = (uint8_t)r; o[3 * i + 1] = (uint8_t)gg; o[3 * i + 2] = (uint8_t)bb;
      }
   }
   if (words) {
      unsigned int *ow = reinterpret_cast<unsigned int *>(o);
      ow[0] = pk[0]; ow[1] = pk[1]; ow[2] = pk[2];
   }
}
