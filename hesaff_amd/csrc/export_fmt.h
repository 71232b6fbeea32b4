// export_fmt.h -- the arithmetic of exportKeypoints (hesaff.cpp:107-130), written once for the host writer
// (hostio.cpp) and for the GPU formatter (kernels_export.h): the ellipse (a, b, c) of a region and the
// default operator<<(ostream&, float) print of a value ("%g", six significant digits).  Integer arithmetic
// only in the formatter, explicit IEEE double operations in the ellipse (translation units that include
// this header are compiled with -ffp-contract=off), so host and device produce the same bytes.
#pragma once
#include <math.h>
#include <stdint.h>
#include "hmath.h"

// hesaff.cpp:115-123, statement by statement:
//    float sc = mrSize * k.s;  SVD svd(A, FULL_UV);  d[i] = 1.0f/(d[i]*d[i]*sc*sc);
//    A = svd.u * Mat::diag(svd.w) * svd.u.t();   ->  a = A(0,0), b = A(0,1), c = A(1,1)
// cv::SVD (OpenCV's float Jacobi solver) is replaced by the closed-form symmetric
// eigen-decomposition of A A^T evaluated in double; u and w are then stored as float like the
// members of cv::SVD, d is the reference's float expression and the two matrix products
// accumulate in double like cv::gemm does for CV_32F.  (a,b,c) carry the 1e-4 tolerance of
// north_star; on SURVEY.md App. C's 640x480 input this form reproduces the md5 of the compiled
// reference's output file.
// A NaN entry (records no image produces: a non-finite or zero shape) is stored as x86's default NaN, the value the reference's
// SSE arithmetic gives an invalid operation and glibc prints as "-nan": the sign of a generated NaN is the one thing that differs
// between the host's and the GPU's IEEE arithmetic.
HM_HD float hx_canon_nan(float v) { return v != v ? hm_u2f(0xffc00000u) : v; }

HM_HD void hx_ellipse(float s, float fa11, float fa12, float fa21, float fa22, float mrSize, float *a, float *b, float *c)
{
   const float sc = mrSize * s;
   const double a11 = fa11, a12 = fa12, a21 = fa21, a22 = fa22;
   const double m00 = a11 * a11 + a12 * a12, m01 = a11 * a21 + a12 * a22, m11 = a21 * a21 + a22 * a22;
   const double tr = m00 + m11, df = m00 - m11;
   const double disc = sqrt(df * df + 4.0 * m01 * m01);
   const double l1 = (tr + disc) / 2.0, l2 = (tr - disc) / 2.0;
   // unit eigenvector of the larger eigenvalue: (l1 - m11, m01) or (m01, l1 - m00), the longer one
   double vx = l1 - m11, vy = m01;
   const double ux = m01, uy = l1 - m00;
   if (ux * ux + uy * uy > vx * vx + vy * vy) { vx = ux; vy = uy; }
   const double n = sqrt(vx * vx + vy * vy);
   float cu = 1.0f, su = 0.0f;
   if (n > 0) { cu = (float)(vx / n); su = (float)(vy / n); }
   float w0 = (float)sqrt(l1), w1 = (float)sqrt(l2);
   w0 = 1.0f / (w0 * w0 * sc * sc);
   w1 = 1.0f / (w1 * w1 * sc * sc);
   const float p00 = (float)((double)cu * w0), p01 = (float)(-(double)su * w1);
   const float p10 = (float)((double)su * w0), p11 = (float)((double)cu * w1);
   *a = hx_canon_nan((float)((double)p00 * cu + (double)p01 * -su));
   *b = hx_canon_nan((float)((double)p00 * su + (double)p01 * cu));
   *c = hx_canon_nan((float)((double)p10 * su + (double)p11 * cu));
}

// ---------------------------------------------------------------------------------------------------------------
// "%g" (precision 6) of a float == default operator<<(ostream&, float), the format of hesaff.cpp:125.
// A binary32 value is mant * 2^e exactly, so its six significant digits are
//    d = round-half-even(mant * 2^e / 10^(X-5)),  X = floor(log10 |v|),
// an integer division.  Two evaluations of it:
//   * 1e-22 <= |v| < 2^23, normal (every coordinate and every ellipse entry of a real image):
//     (mant * 10^(5-X)) >> -e in 128 bits;
//   * everything else - denormals, |v| < 1e-22, |v| >= 2^23 - by the same division on 256-bit integers (a 20-step
//     bisection for the quotient: rare, compact, exact).
// inf / nan print as glibc prints them ("inf", "-inf", "nan", "-nan").  tests/test_host_side.py compares the host
// build with snprintf on millions of values over the whole range; tests/test_gpu_parity.py compares the device with the host.
// A Sink receives the characters one by one: a counter (length pass) or a pointer (write pass).
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned __int128 hx_u128;

struct HxCount {
   int n = 0;
   HM_HD void put(char) { n++; }
};
struct HxPtr {
   char *p;
   HM_HD void put(char ch) { *p++ = ch; }
};

// 10^k as a 128-bit integer, k <= 27 (10^27 < 2^90)
HM_HD hx_u128 hx_pow10_128(int k)
{
   const uint64_t p10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                             10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull,
                             1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull,
                             10000000000000000000ull};
   return k <= 19 ? (hx_u128)p10[k] : (hx_u128)p10[19] * p10[k - 19];
}

// 256-bit unsigned integers, eight 32-bit limbs, only the operations the slow path needs; every loop has a constant
// trip count so that on the GPU the limbs stay in registers
struct HxBig {
   uint32_t w[8];
};
HM_HD void hx_big_set(HxBig &a, uint32_t v) { a.w[0] = v; for (int i = 1; i < 8; i++) a.w[i] = 0; }
HM_HD void hx_big_mul(HxBig &a, uint32_t m)
{
   uint64_t carry = 0;
   for (int i = 0; i < 8; i++) {
      const uint64_t t = (uint64_t)a.w[i] * m + carry;
      a.w[i] = (uint32_t)t;
      carry = t >> 32;
   }
}
HM_HD void hx_big_mul_pow10(HxBig &a, int k) { for (; k >= 9; k -= 9) hx_big_mul(a, 1000000000u); for (; k > 0; k--) hx_big_mul(a, 10u); }
HM_HD void hx_big_shl(HxBig &a, int k) { for (; k >= 31; k -= 31) hx_big_mul(a, 0x80000000u); if (k > 0) hx_big_mul(a, 1u << k); }
HM_HD int hx_big_cmp(const HxBig &a, const HxBig &b)   // -1, 0, 1
{
   int r = 0;
   for (int i = 0; i < 8; i++) r = a.w[i] != b.w[i] ? (a.w[i] < b.w[i] ? -1 : 1) : r;   // the highest differing limb decides
   return r;
}
HM_HD void hx_big_sub(HxBig &a, const HxBig &b)   // a -= b, a >= b
{
   uint64_t borrow = 0;
   for (int i = 0; i < 8; i++) {
      const uint64_t t = (uint64_t)a.w[i] - b.w[i] - borrow;
      a.w[i] = (uint32_t)t;
      borrow = (t >> 32) & 1u;
   }
}

// digits d (100000 <= d <= 999999) and decimal exponent X of |v| = mant * 2^e, any mant < 2^24, mant != 0
// (not inlined on the device: five prints per row would carry five copies of a path no real row takes; the result comes back in
//  registers - digits | exponent << 32 - because reference parameters would live in scratch memory, see hs_solve3x3)
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#else
inline
#endif
unsigned long long hx_digits_big(uint32_t mant, int e)
{
   const int msb = 31 - __builtin_clz(mant);
   int X = ((msb + e) * 1233) >> 12;   // floor(log10 |v|) or one off: floor((msb + e) * log10(2)), 1233 / 4096 = 0.30103
   unsigned d = 0;
   for (int tries = 0; tries < 4; tries++) {
      HxBig num, den;
      hx_big_set(num, mant);
      hx_big_set(den, 1u);
      if (X <= 5) hx_big_mul_pow10(num, 5 - X); else hx_big_mul_pow10(den, X - 5);
      if (e >= 0) hx_big_shl(num, e); else hx_big_shl(den, -e);
      // q = floor(num / den) if it is below 2^20, by bisection on q * den <= num
      unsigned q = 0;
      for (int bit = 19; bit >= 0; bit--) {
         HxBig t = den;
         hx_big_mul(t, q | (1u << bit));
         if (hx_big_cmp(t, num) <= 0) q |= 1u << bit;
      }
      if (q >= 1000000u) { X++; continue; }   // (also when the true quotient exceeds 2^20 - 1: q saturates there)
      if (q < 100000u) { X--; continue; }
      HxBig t = den;
      hx_big_mul(t, q);
      hx_big_sub(num, t);                      // remainder
      hx_big_mul(num, 2u);
      const int c = hx_big_cmp(num, den);
      d = q;
      if (c > 0 || (c == 0 && (d & 1u))) d++;
      if (d == 1000000u) { d = 100000u; X++; }
      break;
   }
   return (unsigned long long)d | ((unsigned long long)(uint32_t)X << 32);
}

template <class Sink> HM_HD void hx_fmt_g(Sink &out, float vf)
{
   const uint32_t bits = hm_f2u(vf);
   const uint32_t ex = (bits >> 23) & 255u, fr = bits & 0x7fffffu;
   if (bits >> 31) out.put('-');
   if (ex == 255u) {
      if (fr) { out.put('n'); out.put('a'); out.put('n'); }
      else { out.put('i'); out.put('n'); out.put('f'); }
      return;
   }
   if (ex == 0 && fr == 0) { out.put('0'); return; }
   const uint32_t mant = ex ? (fr | 0x800000u) : fr;
   const int e = ex ? (int)ex - 150 : -149;   // |v| = mant * 2^e
   unsigned d = 0;
   int X = 0;
   bool fast = ex != 0 && e < 0 && e >= -96;   // 2^-73 <= |v| < 2^23
   if (fast) {
      const int sh = -e;
      X = ((23 + e) * 1233) >> 12;
      hx_u128 q = 0, N = 0;
      int tries = 0;
      for (; tries < 3; tries++) {
         if (X > 5 || X < -22) break;
         N = (hx_u128)mant * hx_pow10_128(5 - X);   // < 2^24 * 10^27 < 2^114
         q = N >> sh;
         if (q >= 1000000u) { X++; continue; }
         if (q < 100000u) { X--; continue; }
         break;
      }
      fast = tries < 3 && X <= 5 && X >= -22 && q >= 100000u && q < 1000000u;
      if (fast) {
         const hx_u128 rem = N & ((((hx_u128)1) << sh) - 1), half = ((hx_u128)1) << (sh - 1);
         d = (unsigned)q;
         if (rem > half || (rem == half && (d & 1u))) d++;
         if (d == 1000000u) { d = 100000u; X++; }
      }
   }
   if (!fast) { const unsigned long long dx = hx_digits_big(mant, e); d = (unsigned)dx; X = (int)(uint32_t)(dx >> 32); }
   // six digits, most significant first
   char dig[6];
   for (int i = 5; i >= 0; i--) { dig[i] = (char)('0' + d % 10u); d /= 10u; }
   int nd = 6;
   for (int i = 5; i >= 1; i--) nd = (nd == i + 1 && dig[i] == '0') ? i : nd;   // %g strips trailing zeros
   if (X < -4 || X >= 6) {
      out.put(dig[0]);
      if (nd > 1) out.put('.');
      for (int i = 1; i < 6; i++) if (i < nd) out.put(dig[i]);
      out.put('e');
      int ax = X;
      if (ax < 0) { out.put('-'); ax = -ax; } else out.put('+');
      out.put((char)('0' + ax / 10));
      out.put((char)('0' + ax % 10));
   } else if (X >= 0) {
      for (int i = 0; i < 6; i++) if (i <= X) out.put(dig[i]);   // nd may be <= X: the stripped zeros belong to the integer part
      if (nd > X + 1) out.put('.');
      for (int i = 1; i < 6; i++) if (i > X && i < nd) out.put(dig[i]);
   } else {
      out.put('0'); out.put('.');
      for (int i = 0; i < 3; i++) if (i < -X - 1) out.put('0');
      for (int i = 0; i < 6; i++) if (i < nd) out.put(dig[i]);
   }
}

// " 0" .. " 255": characters of one descriptor byte in a row (separator + digits)
HM_HD int hx_u8_len(unsigned v) { return 2 + (v >= 10u) + (v >= 100u); }
template <class Sink> HM_HD void hx_fmt_u8(Sink &out, unsigned v)
{
   out.put(' ');
   const unsigned h = v / 100u, t = (v / 10u) % 10u;
   if (v >= 100u) out.put((char)('0' + h));
   if (v >= 10u) out.put((char)('0' + t));
   out.put((char)('0' + v % 10u));
}

// the five floats of a row: "x y a b c" (hesaff.cpp:125-126), without the descriptor
template <class Sink> HM_HD void hx_fmt_row_head(Sink &out, float x, float y, float s, float a11, float a12, float a21, float a22, float mrSize)
{
   float ea, eb, ec;
   hx_ellipse(s, a11, a12, a21, a22, mrSize, &ea, &eb, &ec);
   hx_fmt_g(out, x); out.put(' ');
   hx_fmt_g(out, y); out.put(' ');
   hx_fmt_g(out, ea); out.put(' ');
   hx_fmt_g(out, eb); out.put(' ');
   hx_fmt_g(out, ec);
}
