// kernels_export.h -- exportKeypoints (hesaff.cpp:107-130) on the device: the rows of a .hesaff.sift file
// ("x y a b c d1 ... d128\n", default ostream formatting) and of the binary sidecar, produced from the ordered
// hesaff_keypoint records of a batch so that the host threads of the file pipeline only write() them.
//
//   k_text_len     one thread per row: length of the row's text (the row is formatted, the characters counted),
//                  + the sum of every 64 consecutive rows
//   k_text_scan    one block: exclusive 64-bit scan of those sums = byte offset of every block of 64 rows
//   k_text_imgoff  byte offset of the first row of every image (the host cuts the chunk's text there)
//   k_text_write   one wavefront per 64 rows: lane = row formats into LDS at the row's offset inside the block, then
//                  the block's text leaves as aligned dwords (the LDS image starts at the byte phase of its place
//                  in the output, so that aligned dwords of LDS are aligned dwords of the output)
//   k_bin_rows     148-byte rows { x, y, a, b, c, desc[128] } of the sidecar
// The arithmetic (ellipse in double, "%g" by integer division) is export_fmt.h, shared with the host writer.
#pragma once
#include "device_common.h"
#include "export_fmt.h"
#include "kernels_keypoint.h"

#define EX_ROWS 64      // rows per block of k_text_write (one wavefront)
#define EX_ROW_MAX 580  // >= 5 * 12 ("-1.23457e-38") + 4 + 128 * 4 + 1 characters
#define EX_BIN_ROW 148  // 5 floats + 128 bytes

struct KeyHead { float x, y, s, a11, a12, a21, a22; };
__device__ __forceinline__ KeyHead ex_load_head(const KeyRec *__restrict__ k)
{
   const float *p = (const float *)k;
   KeyHead h;
   h.x = p[0]; h.y = p[1]; h.s = p[2]; h.a11 = p[3]; h.a12 = p[4]; h.a21 = p[5]; h.a22 = p[6];
   return h;
}

__global__ __launch_bounds__(256) void k_text_len(const KeyRec *__restrict__ keys, uint32_t n, float mrSize, uint16_t *__restrict__ len,
                                                  uint32_t *__restrict__ blocksum)
{
   const uint32_t r = blockIdx.x * 256u + threadIdx.x;
   uint32_t L = 0;
   if (r < n) {
      const KeyHead h = ex_load_head(keys + r);
      HxCount cnt;
      hx_fmt_row_head(cnt, h.x, h.y, h.s, h.a11, h.a12, h.a21, h.a22, mrSize);
      L = (uint32_t)cnt.n + 1u + 256u;   // '\n' + a separator and one digit per descriptor byte
      const uint32_t *d = (const uint32_t *)keys[r].desc;
      for (int i = 0; i < 32; i++) {
         const uint32_t w = d[i], lo = w & 0x7f7f7f7fu;
         L += __popc(((lo + 0x76767676u) | w) & 0x80808080u);   // bytes >= 10
         L += __popc(((lo + 0x1c1c1c1cu) | w) & 0x80808080u);   // bytes >= 100
      }
      len[r] = (uint16_t)L;
   }
   uint32_t s = L;
   for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
   if ((threadIdx.x & 63u) == 0 && (r & ~63u) < n) blocksum[r >> 6] = s;
}

// exclusive scan of nblk 32-bit sums into 64-bit offsets; off[nblk] = total.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void k_text_scan(const uint32_t *__restrict__ sums, uint32_t nblk, unsigned long long *__restrict__ off)
{
   __shared__ unsigned long long part[1024];
   const uint32_t per = (nblk + 1023u) / 1024u, i0 = threadIdx.x * per, i1 = min(nblk, i0 + per);
   unsigned long long s = 0;
   for (uint32_t i = i0; i < i1; i++) s += sums[i];
   part[threadIdx.x] = s;
   __syncthreads();
   for (uint32_t d = 1; d < 1024u; d <<= 1) {
      const unsigned long long t = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
   }
   unsigned long long run = part[threadIdx.x] - s;
   for (uint32_t i = i0; i < i1; i++) { off[i] = run; run += sums[i]; }
   if (threadIdx.x == 1023u) off[nblk] = part[1023];
}

// starts[b] = first row of image b (starts[B] = n): byte offset of that row in the chunk's text
__global__ void k_text_imgoff(const int32_t *__restrict__ starts, int B, const uint16_t *__restrict__ len,
                              const unsigned long long *__restrict__ blockoff, unsigned long long *__restrict__ imgoff)
{
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b > B) return;
   const uint32_t r = (uint32_t)starts[b];
   unsigned long long o = blockoff[r >> 6];
   for (uint32_t i = r & ~63u; i < r; i++) o += len[i];
   imgoff[b] = o;
}

struct ExLds {
   char *p;
   __device__ __forceinline__ void put(char ch) { *p++ = ch; }
};

__global__ __launch_bounds__(64) void k_text_write(const KeyRec *__restrict__ keys, uint32_t n, float mrSize, const uint16_t *__restrict__ len,
                                                   const unsigned long long *__restrict__ blockoff, char *__restrict__ text)
{
   __shared__ __attribute__((aligned(16))) char buf[EX_ROWS * EX_ROW_MAX + 16];
   const uint32_t lane = threadIdx.x, r = blockIdx.x * EX_ROWS + lane;
   const uint32_t L = r < n ? len[r] : 0u;
   uint32_t incl = L;
   for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if ((int)lane >= d) incl += t;
   }
   const uint32_t off = incl - L, T = __shfl(incl, 63);
   const unsigned long long base = blockoff[blockIdx.x];
   const uint32_t pad = (uint32_t)(base & 3ull);
   if (r < n) {
      const KeyHead h = ex_load_head(keys + r);
      ExLds out;
      out.p = buf + pad + off;
      hx_fmt_row_head(out, h.x, h.y, h.s, h.a11, h.a12, h.a21, h.a22, mrSize);
      const uint32_t *d = (const uint32_t *)keys[r].desc;
      for (int i = 0; i < 32; i++) {
         const uint32_t w = d[i];
         hx_fmt_u8(out, w & 255u); hx_fmt_u8(out, (w >> 8) & 255u); hx_fmt_u8(out, (w >> 16) & 255u); hx_fmt_u8(out, w >> 24);
      }
      out.put('\n');
   }
   HS_WAVE_LDS_SYNC();
   char *g = text + base;
   const uint32_t head = min((4u - pad) & 3u, T);
   if (lane < head) g[lane] = buf[pad + lane];
   const uint32_t nd = (T - head) >> 2;
   const uint32_t *src = (const uint32_t *)(buf + pad + head);
   uint32_t *dst = (uint32_t *)(g + head);
   for (uint32_t i = lane; i < nd; i += 64u) __builtin_nontemporal_store(src[i], dst + i);
   const uint32_t done = head + 4u * nd;
   if (lane < T - done) g[done + lane] = buf[pad + done + lane];
}

// rows of the binary sidecar (hesaff_write_bin): 32 lanes per row, lane 0 also writes the five floats
__global__ __launch_bounds__(256) void k_bin_rows(const KeyRec *__restrict__ keys, uint32_t n, float mrSize, uint32_t *__restrict__ out)
{
   const uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 5, lane = threadIdx.x & 31u, stride = (gridDim.x * 256u) >> 5;
   for (uint32_t r = g; r < n; r += stride) {
      uint32_t *o = out + (size_t)r * (EX_BIN_ROW / 4);
      if (lane == 0) {
         const KeyHead h = ex_load_head(keys + r);
         float ea, eb, ec;
         hx_ellipse(h.s, h.a11, h.a12, h.a21, h.a22, mrSize, &ea, &eb, &ec);
         o[0] = __float_as_uint(h.x); o[1] = __float_as_uint(h.y); o[2] = __float_as_uint(ea); o[3] = __float_as_uint(eb); o[4] = __float_as_uint(ec);
      }
      o[5 + lane] = ((const uint32_t *)keys[r].desc)[lane];
   }
}

// test hook: the device formatter on arbitrary values, 16 bytes of text per value + its length
__global__ void k_fmt_g_test(int n, const float *__restrict__ v, char *__restrict__ out, int32_t *__restrict__ lens)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   HxPtr o;
   o.p = out + (size_t)i * 16;
   hx_fmt_g(o, v[i]);
   lens[i] = (int32_t)(o.p - (out + (size_t)i * 16));
}
