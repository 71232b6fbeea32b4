// jpeg_decode.cpp -- Huffman-coded 8-bit JPEG decoder (sequential SOF0/SOF1 and progressive SOF2) for hesaff_read_image:
// replaces cv::imread(argv[1]) (hesaff.cpp:137) for the format of the Oxford buildings / graf images.
//
// Pixels matter here: the detector's output depends on every decoded byte, and cv::imread decodes
// through libjpeg with its defaults.  This decoder restates the PUBLISHED integer algorithms of the
// Independent JPEG Group's library for exactly that configuration, so that the bytes equal
// libjpeg's (and libjpeg-turbo's, whose SIMD paths are bit-exact with the C code):
//   * dct_method = JDCT_ISLOW       -> jidctint.c, 8x8 "accurate integer" inverse DCT (CONST_BITS 13, PASS1_BITS 2),
//                                      dequantisation inside the column pass;
//   * do_fancy_upsampling = TRUE    -> jdsample.c h2v1 / h2v2 "fancy" (triangle-filter) chroma up-sampling, image-edge
//                                      rows and columns replicated; integral replication for any other sampling ratio;
//   * YCbCr -> RGB                  -> jdcolor.c fixed-point tables (SCALEBITS 16), JFIF convention;
//   * range limiting to 0..255 after the IDCT (+128 level shift) and after the colour conversion.
// tests/test_host_side.py compares every pixel with the libjpeg-turbo decoder bundled with Pillow on 4:4:4,
// 4:2:2, 4:2:0, 4:4:0, 4:1:1, grey, odd sizes, restart intervals, several qualities, sequential and progressive.
// Progressive files (SOF2: spectral selection + successive approximation, jdphuff.c) are collected scan by scan into
// coefficient arrays and transformed once at the end; a COMPLETE file gets no inter-block smoothing in libjpeg, so the
// pixels are again identical.  Not decoded (HESAFF_ERR_IO): lossless / arithmetic-coded / 12-bit / CMYK files.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/hesaff_amd.h"

namespace {

struct Huff {
   // canonical Huffman table, JPEG Annex C / F.2.2.3: codes of length l occupy [mincode[l], maxcode[l]]
   int mincode[17], maxcode[18], valptr[17];
   uint8_t vals[256];
   bool defined = false;
   uint8_t look_len[512];    // 9-bit lookahead: code length (0 = longer than 9 bits)
   uint8_t look_val[512];
   // AC tables: when the code AND the value bits that follow it fit into FAST_BITS looked-ahead bits (nearly all coefficients of a photograph),
   // one entry gives everything: (EXTENDed value << 8) | (run << 4) | bits to skip; 0: take the general path
   static constexpr int FAST_BITS = 11;   // 9: 10 % slower; 12: no faster (the two AC tables then fill the L1 cache)
   int32_t fast_ac[1 << FAST_BITS];   // indexed by FAST_BITS looked-ahead bits (the symbol table above by 9)
   // false: the code-length counts do not describe a prefix code (more codes of some length than that length has left;
   // libjpeg: JERR_BAD_HUFF_TABLE).  Such a table must be refused before the lookahead fill below, whose index
   // code << (9 - l) would leave the 512 entries.
   bool build(const uint8_t *bits /*[1..16] at index 0..15*/, const uint8_t *v, int nv)
   {
      defined = false;
      if (nv < 0 || nv > 256) return false;
      memcpy(vals, v, (size_t)nv);
      int code = 0, k = 0;
      for (int l = 1; l <= 16; l++) {
         valptr[l] = k;
         mincode[l] = code;
         code += bits[l - 1];
         k += bits[l - 1];
         if (code > (1 << l)) return false;   // over-subscribed at length l
         maxcode[l] = bits[l - 1] ? code - 1 : -1;
         code <<= 1;
      }
      if (k != nv) return false;
      maxcode[17] = 0x7fffffff;
      memset(look_len, 0, sizeof look_len);
      code = 0; k = 0;
      for (int l = 1; l <= 9; l++) {
         for (int i = 0; i < bits[l - 1]; i++, k++, code++) {
            const int lo = code << (9 - l);
            for (int f = 0; f < (1 << (9 - l)); f++) { look_len[lo + f] = (uint8_t)l; look_val[lo + f] = v[k]; }
         }
         code <<= 1;
      }
      for (int i = 0; i < (1 << FAST_BITS); i++) {
         fast_ac[i] = 0;
         const int top = i >> (FAST_BITS - 9);                      // the 9 bits of the symbol lookahead
         const int len = look_len[top];
         if (!len) continue;
         const int rs = look_val[top], run = rs >> 4, mag = rs & 15;
         if (mag == 0 || len + mag > FAST_BITS) continue;
         int k = ((i << len) & ((1 << FAST_BITS) - 1)) >> (FAST_BITS - mag);   // the mag bits after the code
         if (k < (1 << (mag - 1))) k = k - (1 << mag) + 1;          // F.2.2.1 EXTEND
         fast_ac[i] = k * 256 + run * 16 + len + mag;               // |k| < 2048; len + mag <= 11 fits the low four bits
      }
      defined = true;
      return true;
   }
};

struct BitReader {
   const uint8_t *p, *end;
   uint64_t acc = 0;     // the next bits of the stream, first bit at bit 63
   int nbits = 0;
   bool hit_marker = false;
   void fill()
   {
      // eight bytes at once while none of them is 0xFF (no stuffed byte, no marker): the common case in entropy-coded data
      if (!hit_marker && nbits <= 32 && end - p >= 8) {
         uint64_t v;
         memcpy(&v, p, 8);
         v = __builtin_bswap64(v);
         const uint64_t inv = ~v;   // a byte of v is 0xFF <=> that byte of inv is 0
         if (((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull) == 0) {
            const int take = (64 - nbits) >> 3;   // whole bytes that fit (4..8)
            acc |= (v >> (64 - 8 * take)) << (64 - nbits - 8 * take);
            p += take;
            nbits += 8 * take;
            return;
         }
      }
      while (nbits <= 56) {
         int b = 0;
         if (!hit_marker && p < end) {
            b = *p;
            if (b == 0xFF) {
               if (p + 1 < end && p[1] == 0x00) p += 2;         // stuffed zero byte
               else { hit_marker = true; b = 0; }               // a marker: feed zeros (libjpeg does the same at the end of a segment)
            } else p++;
         }
         acc |= (uint64_t)b << (56 - nbits);
         nbits += 8;
      }
   }
   int peek(int n) { if (nbits < n) fill(); return (int)(acc >> (64 - n)); }   // n in 1..16
   void skip(int n) { acc <<= n; nbits -= n; }
   int get(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
   void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int decode_huff(BitReader &br, const Huff &h)
{
   const int look = br.peek(9);
   const int l = h.look_len[look];
   if (l) { br.skip(l); return h.look_val[look]; }
   int code = br.peek(16);
   for (int len = 10; len <= 16; len++) {
      const int c = code >> (16 - len);
      if (c <= h.maxcode[len] && h.maxcode[len] >= 0 && c >= h.mincode[len]) { br.skip(len); return h.vals[h.valptr[len] + c - h.mincode[len]]; }
   }
   br.skip(16);
   return 0;   // corrupt data: treat as zero (the reference decoder warns and goes on)
}

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }   // F.2.2.1 EXTEND

const uint8_t kZigZag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Two's-complement arithmetic that wraps: what libjpeg's INT32 does on every real target.  The IDCT of a legal stream
// never leaves 32 bits; a damaged one can, and signed overflow would be undefined behaviour here.
struct W32 {
   uint32_t u;
   W32() = default;
   explicit W32(int32_t v) : u((uint32_t)v) {}
   int32_t s() const { return (int32_t)u; }
   friend W32 operator+(W32 a, W32 b) { W32 r; r.u = a.u + b.u; return r; }
   friend W32 operator-(W32 a, W32 b) { W32 r; r.u = a.u - b.u; return r; }
   friend W32 operator*(W32 a, W32 b) { W32 r; r.u = a.u * b.u; return r; }
   friend W32 operator>>(W32 a, int n) { return W32(a.s() >> n); }
};

// jidctint.c (jpeg_idct_islow): coef in natural order, quant table in natural order, output 8 rows of 8 samples
void idct_islow(const int16_t *coef, const uint16_t *quant, uint8_t *out, int out_stride)
{
   constexpr int CONST_BITS = 13, PASS1_BITS = 2;
   constexpr int32_t FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270, FIX_0_899976223 = 7373,
                     FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069, FIX_2_053119869 = 16819,
                     FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;
   auto descale = [](W32 x, int n) { return (x + W32(1 << (n - 1))) >> n; };
   W32 ws[64];
   for (int c = 0; c < 8; c++) {
      const int16_t *in = coef + c;
      const uint16_t *q = quant + c;
      W32 *w = ws + c;
      auto D = [&](int r) { return W32((int32_t)in[8 * r] * (int32_t)q[8 * r]); };   // |int16 x uint16| < 2^31
      W32 z2 = D(2), z3 = D(6);
      W32 z1 = (z2 + z3) * W32(FIX_0_541196100);
      W32 tmp2 = z1 + z3 * W32(-FIX_1_847759065);
      W32 tmp3 = z1 + z2 * W32(FIX_0_765366865);
      z2 = D(0); z3 = D(4);
      W32 tmp0 = (z2 + z3) * W32(1 << CONST_BITS);
      W32 tmp1 = (z2 - z3) * W32(1 << CONST_BITS);
      const W32 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = D(7); tmp1 = D(5); tmp2 = D(3); tmp3 = D(1);
      z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
      W32 z4 = tmp1 + tmp3;
      const W32 z5 = (z3 + z4) * W32(FIX_1_175875602);
      tmp0 = tmp0 * W32(FIX_0_298631336); tmp1 = tmp1 * W32(FIX_2_053119869); tmp2 = tmp2 * W32(FIX_3_072711026); tmp3 = tmp3 * W32(FIX_1_501321110);
      z1 = z1 * W32(-FIX_0_899976223); z2 = z2 * W32(-FIX_2_562915447); z3 = z3 * W32(-FIX_1_961570560); z4 = z4 * W32(-FIX_0_390180644);
      z3 = z3 + z5; z4 = z4 + z5;
      tmp0 = tmp0 + z1 + z3; tmp1 = tmp1 + z2 + z4; tmp2 = tmp2 + z2 + z3; tmp3 = tmp3 + z1 + z4;
      w[8 * 0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
      w[8 * 7] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
      w[8 * 1] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
      w[8 * 6] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
      w[8 * 2] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
      w[8 * 5] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
      w[8 * 3] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
      w[8 * 4] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
   }
   for (int r = 0; r < 8; r++) {
      const W32 *w = ws + 8 * r;
      uint8_t *o = out + (size_t)r * out_stride;
      W32 z2 = w[2], z3 = w[6];
      W32 z1 = (z2 + z3) * W32(FIX_0_541196100);
      W32 tmp2 = z1 + z3 * W32(-FIX_1_847759065);
      W32 tmp3 = z1 + z2 * W32(FIX_0_765366865);
      W32 tmp0 = (w[0] + w[4]) * W32(1 << CONST_BITS);
      W32 tmp1 = (w[0] - w[4]) * W32(1 << CONST_BITS);
      const W32 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
      z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
      W32 z4 = tmp1 + tmp3;
      const W32 z5 = (z3 + z4) * W32(FIX_1_175875602);
      tmp0 = tmp0 * W32(FIX_0_298631336); tmp1 = tmp1 * W32(FIX_2_053119869); tmp2 = tmp2 * W32(FIX_3_072711026); tmp3 = tmp3 * W32(FIX_1_501321110);
      z1 = z1 * W32(-FIX_0_899976223); z2 = z2 * W32(-FIX_2_562915447); z3 = z3 * W32(-FIX_1_961570560); z4 = z4 * W32(-FIX_0_390180644);
      z3 = z3 + z5; z4 = z4 + z5;
      tmp0 = tmp0 + z1 + z3; tmp1 = tmp1 + z2 + z4; tmp2 = tmp2 + z2 + z3; tmp3 = tmp3 + z1 + z4;
      constexpr int SH = CONST_BITS + PASS1_BITS + 3;
      // range_limit[(x) & RANGE_MASK] of libjpeg == clamp(x + 128, 0, 255) for every value the IDCT of legal data produces;
      // values outside the 10-bit mask window wrap there, the same masking is applied here
      auto lim = [](W32 x) { int32_t v = x.s(); v &= 1023; if (v >= 512) v -= 1024; return clamp8(v + 128); };
      o[0] = lim(descale(tmp10 + tmp3, SH)); o[7] = lim(descale(tmp10 - tmp3, SH));
      o[1] = lim(descale(tmp11 + tmp2, SH)); o[6] = lim(descale(tmp11 - tmp2, SH));
      o[2] = lim(descale(tmp12 + tmp1, SH)); o[5] = lim(descale(tmp12 - tmp1, SH));
      o[3] = lim(descale(tmp13 + tmp0, SH)); o[4] = lim(descale(tmp13 - tmp0, SH));
   }
}

struct Comp {
   int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
   int w = 0, hgt = 0;            // downsampled_width / _height: ceil(image * samp / max_samp)
   int bw = 0, bh = 0;            // blocks allocated (multiple of the sampling factors: full MCUs)
   int pred = 0;
   std::vector<uint8_t> plane;    // bw*8 x bh*8 samples
   std::vector<int16_t> coef;     // progressive (and every file when only the coefficients are wanted): bw x bh blocks of 64
                                  // coefficients in natural order, filled scan by scan
   int16_t *cf = nullptr;         // = coef.data(), or the component's place in the blob that is handed to the device
   int covered = 0;               // sequential scans seen: 0 none, 1 of this component alone (ceil(w/8) x ceil(h/8) blocks), 2 interleaved (all blocks)
   uint16_t q[64];                // coefficients-only mode: the quantisation table the pixel stage has to use for this component
};

// what decode_jpeg hands over instead of pixels (hesaff_read_jpeg_coefficients): the device transforms the blocks (kernels_jpeg.h)
// alloc (optional): where the blob comes from - a caller that decodes many files hands back blobs it is done with instead of paying
// for 25 MB of fresh zero pages per photograph; *zeroed tells whether the memory is already zero
struct CoefOut {
   hesaff_jpeg_layout *layout; uint8_t **blob; size_t *blob_bytes;
   hesaff_blob_alloc alloc; void *user;
};

inline uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

// jdsample.c, fancy (triangle filter) up-sampling; in: w x h samples with row stride `stride`
void upsample_h2v1_fancy(const uint8_t *in, int w, int h, int stride, std::vector<uint8_t> &out, int ow)
{
   out.assign((size_t)ow * h, 0);
   std::vector<uint8_t> row((size_t)2 * w + 2);
   for (int y = 0; y < h; y++) {
      const uint8_t *s = in + (size_t)y * stride;
      uint8_t *d = row.data();
      if (w == 1) { d[0] = d[1] = s[0]; }
      else {
         d[0] = s[0];
         d[1] = (uint8_t)((s[0] * 3 + s[1] + 2) >> 2);
         for (int x = 1; x < w - 1; x++) {
            const int iv = s[x] * 3;
            d[2 * x] = (uint8_t)((iv + s[x - 1] + 1) >> 2);
            d[2 * x + 1] = (uint8_t)((iv + s[x + 1] + 2) >> 2);
         }
         const int iv = s[w - 1] * 3;
         d[2 * w - 2] = (uint8_t)((iv + s[w - 2] + 1) >> 2);
         d[2 * w - 1] = s[w - 1];
      }
      memcpy(&out[(size_t)y * ow], d, (size_t)ow);
   }
}

void upsample_h2v2_fancy(const uint8_t *in, int w, int h, int stride, std::vector<uint8_t> &out, int ow, int oh)
{
   out.assign((size_t)ow * oh, 0);
   std::vector<uint8_t> row((size_t)2 * w + 2);
   for (int oy = 0; oy < oh; oy++) {
      const int y = oy >> 1;
      // nearer input row y, farther row y-1 (upper output row) or y+1 (lower); image edges replicate (jdmainct.c context rows)
      int yf = (oy & 1) ? y + 1 : y - 1;
      if (yf < 0) yf = 0;
      if (yf > h - 1) yf = h - 1;
      const uint8_t *s0 = in + (size_t)(y > h - 1 ? h - 1 : y) * stride, *s1 = in + (size_t)yf * stride;
      uint8_t *d = row.data();
      if (w == 1) {
         const int t = s0[0] * 3 + s1[0];
         d[0] = (uint8_t)((t * 4 + 8) >> 4);
         d[1] = (uint8_t)((t * 4 + 7) >> 4);
      } else {
         int thiscol = s0[0] * 3 + s1[0], nextcol = s0[1] * 3 + s1[1], lastcol;
         d[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
         d[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
         lastcol = thiscol; thiscol = nextcol;
         for (int x = 1; x < w - 1; x++) {
            nextcol = s0[x + 1] * 3 + s1[x + 1];
            d[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            d[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
         }
         d[2 * w - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
         d[2 * w - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
      }
      memcpy(&out[(size_t)oy * ow], d, (size_t)ow);
   }
}

// 4:4:0 (h1v2): vertical triangle filter, rounding bias 1 for the upper and 2 for the lower output row (libjpeg-turbo jdsample.c)
void upsample_h1v2_fancy(const uint8_t *in, int w, int h, int stride, std::vector<uint8_t> &out, int oh)
{
   out.assign((size_t)w * oh, 0);
   for (int oy = 0; oy < oh; oy++) {
      const int y = (oy >> 1) > h - 1 ? h - 1 : (oy >> 1);
      int yf = (oy & 1) ? y + 1 : y - 1;
      if (yf < 0) yf = 0;
      if (yf > h - 1) yf = h - 1;
      const int bias = (oy & 1) ? 2 : 1;
      const uint8_t *s0 = in + (size_t)y * stride, *s1 = in + (size_t)yf * stride;
      uint8_t *d = &out[(size_t)oy * w];
      for (int x = 0; x < w; x++) d[x] = (uint8_t)((s0[x] * 3 + s1[x] + bias) >> 2);
   }
}

// integral replication (jdsample.c int_upsample / h2v1_upsample / h2v2_upsample)
void upsample_replicate(const uint8_t *in, int w, int h, int stride, int hx, int vx, std::vector<uint8_t> &out, int ow, int oh)
{
   out.assign((size_t)ow * oh, 0);
   for (int oy = 0; oy < oh; oy++) {
      const int y = (oy / vx) > h - 1 ? h - 1 : (oy / vx);
      const uint8_t *s = in + (size_t)y * stride;
      uint8_t *d = &out[(size_t)oy * ow];
      for (int ox = 0; ox < ow; ox++) { const int x = ox / hx; d[ox] = s[x > w - 1 ? w - 1 : x]; }
   }
}

int decode_jpeg(const std::vector<uint8_t> &f, uint8_t **data, int *width, int *height, int *channels, const CoefOut *co = nullptr)
{
   const bool keep = co != nullptr;   // entropy decoding only: no inverse DCT, no planes, no pixels
   // a blob from the caller's allocator is the caller's whatever happens: on a failure it goes back through *co->blob, never to free()
   struct FreeDel {
      const CoefOut *co;
      void operator()(uint8_t *p) const { if (co && co->alloc) *co->blob = p; else free(p); }
   };
   std::unique_ptr<uint8_t, FreeDel> blob(nullptr, FreeDel{co});
   size_t blob_bytes = 0;
   bool blob_dirty = false;   // a recycled blob of a sequential file: blocks that no scan decoded are cleared at the end
   const size_t n = f.size();
   if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) return HESAFF_ERR_IO;
   uint16_t qt[4][64];
   bool qt_def[4] = {false, false, false, false};
   Huff hdc[4], hac[4];
   std::vector<Comp> comps;
   int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0;
   bool have_sof = false, adobe = false, done = false, progressive = false;
   int adobe_transform = 0;
   size_t pos = 2;
   while (!done) {
      // next marker
      while (pos < n && f[pos] != 0xFF) pos++;
      while (pos < n && f[pos] == 0xFF) pos++;
      if (pos >= n) break;
      const int m = f[pos++];
      if (m == 0xD9) break;                                 // EOI
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;  // TEM, stray RSTn
      if (pos + 2 > n) return HESAFF_ERR_IO;
      const int len = be16(&f[pos]);
      if (len < 2 || pos + (size_t)len > n) return HESAFF_ERR_IO;
      const uint8_t *seg = &f[pos + 2];
      const int sl = len - 2;
      switch (m) {
         case 0xDB: {   // DQT
            int o = 0;
            while (o < sl) {
               const int pq = seg[o] >> 4, tq = seg[o] & 15;
               o++;
               if (tq > 3 || o + (pq ? 128 : 64) > sl) return HESAFF_ERR_IO;
               for (int i = 0; i < 64; i++) {
                  qt[tq][kZigZag[i]] = pq ? be16(&seg[o + 2 * i]) : seg[o + i];
               }
               o += pq ? 128 : 64;
               qt_def[tq] = true;
            }
            break;
         }
         case 0xC4: {   // DHT
            int o = 0;
            while (o < sl) {
               if (o + 17 > sl) return HESAFF_ERR_IO;
               const int tc = seg[o] >> 4, th = seg[o] & 15;
               int nv = 0;
               for (int i = 0; i < 16; i++) nv += seg[o + 1 + i];
               if (th > 3 || tc > 1 || nv > 256 || o + 17 + nv > sl) return HESAFF_ERR_IO;
               if (tc == 0)   // a DC symbol is a bit count (libjpeg: JERR_BAD_HUFF_TABLE above 15)
                  for (int i = 0; i < nv; i++)
                     if (seg[o + 17 + i] > 15) return HESAFF_ERR_IO;
               if (!(tc ? hac : hdc)[th].build(&seg[o + 1], &seg[o + 17], nv)) return HESAFF_ERR_IO;
               o += 17 + nv;
            }
            break;
         }
         case 0xC0: case 0xC1: case 0xC2: {   // SOF0 baseline / SOF1 extended sequential / SOF2 progressive (all Huffman)
            if (have_sof || sl < 6) return HESAFF_ERR_IO;
            progressive = m == 0xC2;
            if (seg[0] != 8) return HESAFF_ERR_IO;          // sample precision
            H = be16(&seg[1]); W = be16(&seg[3]);
            const int nc = seg[5];
            if (W < 1 || H < 1 || (nc != 1 && nc != 3) || sl < 6 + 3 * nc) return HESAFF_ERR_IO;
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
               Comp &c = comps[i];
               c.id = seg[6 + 3 * i]; c.h = seg[7 + 3 * i] >> 4; c.v = seg[7 + 3 * i] & 15; c.tq = seg[8 + 3 * i];
               if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return HESAFF_ERR_IO;
               hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
            }
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (Comp &c : comps) {
               c.w = (W * c.h + hmax - 1) / hmax; c.hgt = (H * c.v + vmax - 1) / vmax;
               c.bw = mcux * c.h; c.bh = mcuy * c.v;
               // Refuse absurd headers before allocating: every block of a component costs at least one bit of entropy-coded
               // data (sequential: its DC symbol; progressive: its DC symbol in the first DC scan - end-of-band runs shorten
               // the AC scans only), so a file of n bytes holds at most 8 n blocks per component.  The bound below is eight
               // times more generous and still keeps a 100-byte file from asking for gigabytes of coefficients and planes.
               if ((unsigned long long)c.bw * c.bh > (unsigned long long)n * 64ull + 4096ull) return HESAFF_ERR_IO;
               if (!keep) c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 128);   // a block no scan reaches (damaged file): the transform of zeros, as on the device
               if (progressive && !keep) { c.coef.assign((size_t)c.bw * c.bh * 64, 0); c.cf = c.coef.data(); }
               memset(c.q, 0, sizeof c.q);
            }
            if (keep) {   // the blocks are decoded in place: calloc'ed pages, no second copy
               size_t blocks = 0;
               for (Comp &c : comps) blocks += (size_t)c.bw * c.bh;
               blob_bytes = HESAFF_JPEG_BLOB_HEADER + blocks * 128;
               int zeroed = 1;
               blob.reset(co->alloc ? (uint8_t *)co->alloc(blob_bytes, &zeroed, co->user) : (uint8_t *)calloc(1, blob_bytes));
               if (!blob) return HESAFF_ERR_NOMEM;
               // a sequential scan clears every block it decodes; progressive scans add to what is there, and blocks no scan reaches must read zero
               if (!zeroed) memset(blob.get(), 0, progressive ? blob_bytes : (size_t)HESAFF_JPEG_BLOB_HEADER);
               blob_dirty = !zeroed && !progressive;
               size_t at = HESAFF_JPEG_BLOB_HEADER;
               for (Comp &c : comps) { c.cf = reinterpret_cast<int16_t *>(blob.get() + at); at += (size_t)c.bw * c.bh * 128; }
            }
            have_sof = true;
            break;
         }
         case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return HESAFF_ERR_IO;   // lossless, differential, arithmetic: not decoded
         case 0xDD: if (sl < 2) return HESAFF_ERR_IO; restart = be16(seg); break;
         case 0xEE:   // APP14 "Adobe": colour transform flag
            if (sl >= 12 && memcmp(seg, "Adobe", 5) == 0) { adobe = true; adobe_transform = seg[11]; }
            break;
         case 0xDA: {   // SOS + entropy-coded data
            if (!have_sof || sl < 1) return HESAFF_ERR_IO;
            const int ns = seg[0];
            if (ns < 1 || ns > (int)comps.size() || sl < 1 + 2 * ns + 3) return HESAFF_ERR_IO;
            std::vector<Comp *> sc;
            for (int i = 0; i < ns; i++) {
               Comp *c = nullptr;
               for (Comp &q : comps) if (q.id == seg[1 + 2 * i]) c = &q;
               if (!c) return HESAFF_ERR_IO;
               c->td = seg[2 + 2 * i] >> 4; c->ta = seg[2 + 2 * i] & 15;
               if (c->td > 3 || c->ta > 3 || !qt_def[c->tq]) return HESAFF_ERR_IO;
               if (!progressive && (!hdc[c->td].defined || !hac[c->ta].defined)) return HESAFF_ERR_IO;
               c->pred = 0;
               memcpy(c->q, qt[c->tq], sizeof c->q);   // sequential: the table in force when the component's scan starts
               sc.push_back(c);
            }
            BitReader br;
            br.p = &f[pos + len]; br.end = f.data() + n;
            if (progressive) {
               // jdphuff.c: one scan = one band of coefficients (Ss..Se) at one precision step (Ah -> Al)
               const int Ss = seg[1 + 2 * ns], Se = seg[2 + 2 * ns], Ah = seg[3 + 2 * ns] >> 4, Al = seg[3 + 2 * ns] & 15;
               if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1)) return HESAFF_ERR_IO;
               for (Comp *c : sc)
                  if (Ss == 0 ? !hdc[c->td].defined && Ah == 0 : !hac[c->ta].defined) return HESAFF_ERR_IO;
               int mx, my;
               const bool inter = ns > 1;
               if (inter) { mx = (W + 8 * hmax - 1) / (8 * hmax); my = (H + 8 * vmax - 1) / (8 * vmax); }
               else { mx = (sc[0]->w + 7) / 8; my = (sc[0]->hgt + 7) / 8; }
               int to_go = restart;
               unsigned eobrun = 0;
               const int p1 = 1 << Al, m1 = -(1 << Al);
               for (int mcu = 0; mcu < mx * my; mcu++) {
                  if (restart && to_go == 0) {
                     br.reset();
                     const uint8_t *q = br.p;
                     while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                     if (q + 1 < br.end) br.p = q + 2;
                     for (Comp *c : sc) c->pred = 0;
                     eobrun = 0;
                     to_go = restart;
                  }
                  const int mr = mcu / mx, mc = mcu % mx;
                  for (Comp *c : sc) {
                     const int nbh = inter ? c->h : 1, nbv = inter ? c->v : 1;
                     for (int by = 0; by < nbv; by++)
                        for (int bx = 0; bx < nbh; bx++) {
                           const int bxx = mc * nbh + bx, byy = mr * nbv + by;
                           int16_t dummy[64];
                           int16_t *blk = (bxx < c->bw && byy < c->bh) ? c->cf + ((size_t)byy * c->bw + bxx) * 64 : dummy;
                           if (blk == dummy) memset(dummy, 0, sizeof dummy);
                           if (Ss == 0) {
                              if (Ah == 0) {   // DC, first pass: the difference coding of the sequential mode, value scaled by 2^Al
                                 const int t = decode_huff(br, hdc[c->td]);
                                 const int diff = t ? extend(br.get(t), t) : 0;
                                 c->pred = (int)((unsigned)c->pred + (unsigned)diff);
                                 blk[0] = (int16_t)((unsigned)c->pred << Al);
                              } else if (br.get(1)) blk[0] = (int16_t)(blk[0] | p1);   // DC refinement: one more bit
                           } else if (Ah == 0) {   // AC, first pass (decode_mcu_AC_first)
                              if (eobrun > 0) { eobrun--; continue; }
                              const Huff &ac = hac[c->ta];
                              for (int k = Ss; k <= Se; k++) {
                                 const int fa = ac.fast_ac[br.peek(Huff::FAST_BITS)];
                                 if (fa) {   // run, size and value in one look-up (as in the sequential scan below)
                                    k += (fa >> 4) & 15;
                                    br.skip(fa & 15);
                                    if (k > 63) break;
                                    blk[kZigZag[k]] = (int16_t)((unsigned)(fa >> 8) << Al);
                                    continue;
                                 }
                                 const int rs = decode_huff(br, ac);
                                 const int r = rs >> 4, sz = rs & 15;
                                 if (sz) {
                                    k += r;
                                    if (k > 63) break;
                                    blk[kZigZag[k]] = (int16_t)((unsigned)extend(br.get(sz), sz) << Al);
                                 } else if (r == 15) k += 15;   // ZRL: sixteen zeros
                                 else {                          // EOBr: this band ends here for 2^r + extra blocks
                                    eobrun = 1u << r;
                                    if (r) eobrun += (unsigned)br.get(r);
                                    eobrun--;
                                    break;
                                 }
                              }
                           } else {   // AC refinement (decode_mcu_AC_refine): one more bit for every coefficient already non-zero,
                                      // newly non-zero coefficients (+-2^Al) placed after runs of still-zero ones
                              int k = Ss;
                              if (eobrun == 0) {
                                 for (; k <= Se; k++) {
                                    const int rs = decode_huff(br, hac[c->ta]);
                                    int r = rs >> 4;
                                    const int sz = rs & 15;
                                    int val = 0;
                                    if (sz) val = br.get(1) ? p1 : m1;   // size is 1 in a legal stream
                                    else if (r != 15) {
                                       eobrun = 1u << r;
                                       if (r) eobrun += (unsigned)br.get(r);
                                       break;   // the rest of the block is handled as part of the run
                                    }
                                    // advance over already non-zero coefficients (refining each) and r still-zero ones
                                    for (; k <= Se; k++) {
                                       int16_t *cp = blk + kZigZag[k];
                                       if (*cp != 0) {
                                          if (br.get(1) && (*cp & p1) == 0) *cp = (int16_t)(*cp + (*cp >= 0 ? p1 : m1));
                                       } else if (--r < 0) break;
                                    }
                                    if (val && k <= 63) blk[kZigZag[k]] = (int16_t)val;
                                 }
                              }
                              if (eobrun > 0) {
                                 for (; k <= Se; k++) {
                                    int16_t *cp = blk + kZigZag[k];
                                    if (*cp != 0 && br.get(1) && (*cp & p1) == 0) *cp = (int16_t)(*cp + (*cp >= 0 ? p1 : m1));
                                 }
                                 eobrun--;
                              }
                           }
                        }
                  }
                  if (restart) to_go--;
               }
               pos = (size_t)(br.p - f.data());
               if (pos > n) pos = n;
               continue;
            }
            // MCU geometry: interleaved scan = MCUs of h x v blocks per component; a single-component scan runs over that
            // component's own blocks, ceil(size / 8) per row and column (A.2.3)
            int mx, my;
            const bool inter = ns > 1;
            if (inter) { mx = (W + 8 * hmax - 1) / (8 * hmax); my = (H + 8 * vmax - 1) / (8 * vmax); }
            else { mx = (sc[0]->w + 7) / 8; my = (sc[0]->hgt + 7) / 8; }
            int16_t blk_local[64];
            for (Comp *c : sc) c->covered = std::max(c->covered, inter ? 2 : 1);
            int to_go = restart;
            for (int mcu = 0; mcu < mx * my; mcu++) {
               if (restart && to_go == 0) {
                  // RSTn: byte-align, skip the marker, reset predictions
                  br.reset();
                  const uint8_t *q = br.p;
                  while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                  if (q + 1 < br.end) br.p = q + 2;
                  for (Comp *c : sc) c->pred = 0;
                  to_go = restart;
               }
               const int mr = mcu / mx, mc = mcu % mx;
               for (Comp *c : sc) {
                  const int nbh = inter ? c->h : 1, nbv = inter ? c->v : 1;
                  for (int by = 0; by < nbv; by++)
                     for (int bx = 0; bx < nbh; bx++) {
                        const int bxx = mc * nbh + bx, byy = mr * nbv + by;
                        const bool inside = bxx < c->bw && byy < c->bh;
                        int16_t *blk = (keep && inside) ? c->cf + ((size_t)byy * c->bw + bxx) * 64 : blk_local;
                        // (a block visited twice - a damaged file with two scans of one component - starts from zeros again)
                        memset(blk, 0, 128);
                        const int t = decode_huff(br, hdc[c->td]);
                        const int diff = t ? extend(br.get(t), t) : 0;
                        c->pred = (int)((unsigned)c->pred + (unsigned)diff);   // a damaged stream may run the predictor past 32 bits: wrap
                        blk[0] = (int16_t)c->pred;
                        const Huff &ac = hac[c->ta];
                        for (int k = 1; k < 64;) {
                           const int fa = ac.fast_ac[br.peek(Huff::FAST_BITS)];
                           if (fa) {   // run, size and value in one look-up
                              k += (fa >> 4) & 15;
                              br.skip(fa & 15);
                              if (k > 63) break;
                              blk[kZigZag[k]] = (int16_t)(fa >> 8);
                              k++;
                              continue;
                           }
                           const int rs = decode_huff(br, ac);
                           const int r = rs >> 4, s = rs & 15;
                           if (s == 0) {
                              if (r == 15) { k += 16; continue; }
                              break;   // EOB
                           }
                           k += r;
                           if (k > 63) break;
                           blk[kZigZag[k]] = (int16_t)extend(br.get(s), s);
                           k++;
                        }
                        if (inside && !keep) idct_islow(blk, qt[c->tq], &c->plane[((size_t)byy * 8) * ((size_t)c->bw * 8) + (size_t)bxx * 8], c->bw * 8);
                     }
               }
               if (restart) to_go--;
            }
            // continue scanning for markers after the entropy-coded segment
            pos = (size_t)(br.p - f.data());
            if (pos > n) pos = n;
            // back up to a marker if the reader stopped on one
            continue;
         }
         default: break;   // APPn, COM, ...
      }
      pos += (size_t)len;
   }
   if (!have_sof) return HESAFF_ERR_IO;
   if (keep) {
      const int nc = (int)comps.size();
      hesaff_jpeg_layout L;
      memset(&L, 0, sizeof L);
      L.width = W; L.height = H; L.channels = nc == 1 ? 1 : 3;
      size_t blocks = 0;
      for (int i = 0; i < nc; i++) {
         Comp &c = comps[i];
         if (hmax % c.h || vmax % c.v) return HESAFF_ERR_IO;   // fractional sampling ratios: not supported by libjpeg either
         if (progressive) {   // like the pixel path below: the tables as they stand after the last scan
            if (!qt_def[c.tq]) return HESAFF_ERR_IO;
            memcpy(c.q, qt[c.tq], sizeof c.q);
         }
         L.h[i] = c.h; L.v[i] = c.v; L.bw[i] = c.bw; L.bh[i] = c.bh; L.cw[i] = c.w; L.chgt[i] = c.hgt;
         L.hx[i] = hmax / c.h; L.vx[i] = vmax / c.v;
         if (nc == 3 && (c.w * L.hx[i] < W || c.hgt * L.vx[i] < H)) return HESAFF_ERR_IO;
         blocks += (size_t)c.bw * c.bh;
      }
      if (!blob || blob_bytes != HESAFF_JPEG_BLOB_HEADER + blocks * 128) return HESAFF_ERR_IO;
      if (blob_dirty)   // recycled memory: the blocks that no scan of this (sequential) file decoded still hold another image
         for (Comp &c : comps) {
            const int cbw = c.covered == 2 ? c.bw : (c.covered == 1 ? (c.w + 7) / 8 : 0), cbh = c.covered == 2 ? c.bh : (c.covered == 1 ? (c.hgt + 7) / 8 : 0);
            for (int by = 0; by < c.bh; by++) {
               const int x0 = by < cbh ? std::min(cbw, c.bw) : 0;
               if (x0 < c.bw) memset(c.cf + ((size_t)by * c.bw + x0) * 64, 0, (size_t)(c.bw - x0) * 128);
            }
         }
      uint8_t *b = blob.get();
      for (int i = 0; i < nc; i++) memcpy(b + (size_t)i * 128, comps[i].q, 128);
      const int32_t ycc = nc == 3 && (adobe ? adobe_transform != 0 : true) ? 1 : 0;   // JFIF / no marker: YCbCr; Adobe transform 0: RGB
      memcpy(b + 384, &ycc, 4);
      *co->layout = L; *co->blob = blob.release(); *co->blob_bytes = blob_bytes;
      return HESAFF_OK;
   }
   if (progressive)   // all scans are in: one IDCT per block (a complete file gets no inter-block smoothing in libjpeg either)
      for (Comp &c : comps) {
         if (!qt_def[c.tq]) return HESAFF_ERR_IO;
         for (int byy = 0; byy < c.bh; byy++)
            for (int bxx = 0; bxx < c.bw; bxx++)
               idct_islow(c.cf + ((size_t)byy * c.bw + bxx) * 64, qt[c.tq], &c.plane[((size_t)byy * 8) * ((size_t)c.bw * 8) + (size_t)bxx * 8], c.bw * 8);
      }
   const int nc = (int)comps.size();
   uint8_t *out = (uint8_t *)malloc((size_t)W * H * (nc == 1 ? 1 : 3));
   if (!out) return HESAFF_ERR_NOMEM;
   if (nc == 1) {
      const Comp &c = comps[0];
      for (int y = 0; y < H; y++) memcpy(out + (size_t)y * W, &c.plane[(size_t)y * c.bw * 8], (size_t)W);
      *data = out; *width = W; *height = H; *channels = 1;
      return HESAFF_OK;
   }
   // up-sample the components that are not at full resolution (jdsample.c's method selection)
   std::vector<uint8_t> full[3];
   const uint8_t *pl[3];
   int pstride[3];
   for (int i = 0; i < 3; i++) {
      Comp &c = comps[i];
      const int hx = hmax / c.h, vx = vmax / c.v;
      if (hmax % c.h || vmax % c.v) { free(out); return HESAFF_ERR_IO; }   // fractional ratios: not supported by libjpeg either
      if (hx == 1 && vx == 1) { pl[i] = c.plane.data(); pstride[i] = c.bw * 8; continue; }
      const int ow = c.w * hx, oh = c.hgt * vx;
      if (hx == 2 && vx == 1 && c.w > 2) upsample_h2v1_fancy(c.plane.data(), c.w, c.hgt, c.bw * 8, full[i], ow);
      else if (hx == 2 && vx == 2 && c.w > 2) upsample_h2v2_fancy(c.plane.data(), c.w, c.hgt, c.bw * 8, full[i], ow, oh);
      else if (hx == 1 && vx == 2) upsample_h1v2_fancy(c.plane.data(), c.w, c.hgt, c.bw * 8, full[i], oh);
      else upsample_replicate(c.plane.data(), c.w, c.hgt, c.bw * 8, hx, vx, full[i], ow, oh);
      if (ow < W || oh < H) { free(out); return HESAFF_ERR_IO; }
      pl[i] = full[i].data(); pstride[i] = ow;
   }
   // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert
   const bool ycc = adobe ? adobe_transform != 0 : true;   // JFIF / no marker: YCbCr; Adobe transform 0: RGB
   if (ycc) {
      constexpr int SCALEBITS = 16;
      constexpr int32_t ONE_HALF = 1 << (SCALEBITS - 1);
      auto FIX = [](double x) { return (int32_t)(x * (1 << SCALEBITS) + 0.5); };
      int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
      for (int i = 0; i < 256; i++) {
         const int32_t x = i - 128;
         cr_r[i] = (int32_t)((FIX(1.40200) * x + ONE_HALF) >> SCALEBITS);
         cb_b[i] = (int32_t)((FIX(1.77200) * x + ONE_HALF) >> SCALEBITS);
         cr_g[i] = (-FIX(0.71414)) * x;
         cb_g[i] = (-FIX(0.34414)) * x + ONE_HALF;
      }
      for (int y = 0; y < H; y++) {
         const uint8_t *py = pl[0] + (size_t)y * pstride[0], *pb = pl[1] + (size_t)y * pstride[1], *pr = pl[2] + (size_t)y * pstride[2];
         uint8_t *o = out + (size_t)y * W * 3;
         for (int x = 0; x < W; x++) {
            const int Y = py[x], cb = pb[x], cr = pr[x];
            o[3 * x] = clamp8(Y + cr_r[cr]);
            o[3 * x + 1] = clamp8(Y + (int)((cb_g[cb] + cr_g[cr]) >> SCALEBITS));
            o[3 * x + 2] = clamp8(Y + cb_b[cb]);
         }
      }
   } else {
      for (int y = 0; y < H; y++) {
         uint8_t *o = out + (size_t)y * W * 3;
         for (int x = 0; x < W; x++)
            for (int k = 0; k < 3; k++) o[3 * x + k] = pl[k][(size_t)y * pstride[k] + x];
      }
   }
   *data = out; *width = W; *height = H; *channels = 3;
   return HESAFF_OK;
}

} // namespace

static int read_all(const char *path, std::vector<uint8_t> &bytes)
{
   FILE *fp = fopen(path, "rb");
   if (!fp) return HESAFF_ERR_IO;
   try {
      uint8_t chunk[1 << 16];
      for (size_t k; (k = fread(chunk, 1, sizeof chunk, fp)) > 0;) bytes.insert(bytes.end(), chunk, chunk + k);
   } catch (...) {
      fclose(fp);
      throw;
   }
   fclose(fp);
   return HESAFF_OK;
}

extern "C" int hesaff_read_jpeg_coefficients(const char *path, hesaff_jpeg_layout *layout, uint8_t **blob, size_t *blob_bytes)
{
   return hesaff_read_jpeg_coefficients_alloc(path, layout, blob, blob_bytes, nullptr, nullptr);
}

extern "C" int hesaff_read_jpeg_coefficients_alloc(const char *path, hesaff_jpeg_layout *layout, uint8_t **blob, size_t *blob_bytes,
                                                   hesaff_blob_alloc alloc, void *user)
{
   if (!path || !layout || !blob || !blob_bytes) return HESAFF_ERR_ARG;
   *blob = nullptr;
   try {
      std::vector<uint8_t> bytes;
      const int rc = read_all(path, bytes);
      if (rc != HESAFF_OK) return rc;
      const CoefOut co = {layout, blob, blob_bytes, alloc, user};
      return decode_jpeg(bytes, nullptr, nullptr, nullptr, nullptr, &co);
   } catch (const std::bad_alloc &) {
      return HESAFF_ERR_NOMEM;
   } catch (...) {
      return HESAFF_ERR_IO;
   }
}

extern "C" int hesaff_read_jpeg(const char *path, uint8_t **data, int *width, int *height, int *channels)
{
   if (!path || !data || !width || !height || !channels) return HESAFF_ERR_ARG;
   FILE *fp = fopen(path, "rb");
   if (!fp) return HESAFF_ERR_IO;
   try {
      std::vector<uint8_t> bytes;
      try {
         uint8_t chunk[1 << 16];
         for (size_t k; (k = fread(chunk, 1, sizeof chunk, fp)) > 0;) bytes.insert(bytes.end(), chunk, chunk + k);
      } catch (...) {
         fclose(fp);
         throw;
      }
      fclose(fp);
      return decode_jpeg(bytes, data, width, height, channels);
   } catch (const std::bad_alloc &) {
      return HESAFF_ERR_NOMEM;
   } catch (...) {
      return HESAFF_ERR_IO;
   }
}
