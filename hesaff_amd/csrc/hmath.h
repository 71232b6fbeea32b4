// hmath.h -- deterministic scalar math shared by the HIP kernels and the host side of
// libhesaff_amd.  Every function here is written with explicit IEEE binary32/binary64
// operations only (no libm / OCML calls, no FMA contraction: the translation units that
// include this header are compiled with -ffp-contract=off), so the device result is
// bit-identical to the host result.
//
// Why this exists: the reference calls glibc libm for atan2f (siftdesc.cpp:136) and
// powf (pyramid.cpp:196, :227).  The device math library (OCML) rounds differently, and
// a 1-ulp change flips integer descriptor bins.  The two
// functions below re-state the *published algorithms* glibc 2.35 uses:
//   hm_atan2f  : fdlibm e_atan2f.c / s_atanf.c (Sun Microsystems 1993), float-only ops.
//   hm_pow2f   : powf(2.0f, y) of glibc >= 2.28 (Szabolcs Nagy, ARM optimized-routines
//                exp2f table method, EXP2F_TABLE_BITS = 5), non-FMA evaluation.
// tests/test_hmath.py checks both against this image's libm on tens of millions of
// inputs (host build), and tests/test_gpu_hmath.py checks device == host.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HM_HD __host__ __device__ __forceinline__
#else
#define HM_HD inline
#endif

HM_HD uint32_t hm_f2u(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
HM_HD float    hm_u2f(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
HM_HD uint64_t hm_d2u(double f) { union { double f; uint64_t u; } v; v.f = f; return v.u; }
HM_HD double   hm_u2d(uint64_t u) { union { double f; uint64_t u; } v; v.u = u; return v.f; }

HM_HD float hm_fabsf(float x) { return hm_u2f(hm_f2u(x) & 0x7fffffffu); }

// ---- atanf, fdlibm algorithm (argument reduction into 5 intervals + odd/even poly) ----
HM_HD float hm_atanf(float x)
{
   const float atanhi0 = hm_u2f(0x3eed6338u), atanhi1 = hm_u2f(0x3f490fdau),
               atanhi2 = hm_u2f(0x3f7b985eu), atanhi3 = hm_u2f(0x3fc90fdau);
   const float atanlo0 = hm_u2f(0x31ac3769u), atanlo1 = hm_u2f(0x33222168u),
               atanlo2 = hm_u2f(0x33140fb4u), atanlo3 = hm_u2f(0x33a22168u);
   const float aT0 = hm_u2f(0x3eaaaaabu), aT1 = hm_u2f(0xbe4ccccdu), aT2 = hm_u2f(0x3e124925u),
               aT3 = hm_u2f(0xbde38e38u), aT4 = hm_u2f(0x3dba2e6eu), aT5 = hm_u2f(0xbd9d8795u),
               aT6 = hm_u2f(0x3d886b35u), aT7 = hm_u2f(0xbd6ef16bu), aT8 = hm_u2f(0x3d4bda59u),
               aT9 = hm_u2f(0xbd15a221u), aT10 = hm_u2f(0x3c8569d7u);
   const int32_t hx = (int32_t)hm_f2u(x);
   const int32_t ix = hx & 0x7fffffff;
   int id;
   float hi = 0.0f, lo = 0.0f;
   if (ix >= 0x4c000000) {                 // |x| >= 2^25
      if (ix > 0x7f800000) return x + x;   // NaN
      return (hx > 0) ? (atanhi3 + atanlo3) : (-atanhi3 - atanlo3);
   }
   if (ix < 0x3ee00000) {                  // |x| < 0.4375
      if (ix < 0x31000000) return x;       // |x| < 2^-29
      id = -1;
   } else {
      x = hm_fabsf(x);
      if (ix < 0x3f980000) {               // |x| < 1.1875
         if (ix < 0x3f300000) { id = 0; hi = atanhi0; lo = atanlo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
         else                 { id = 1; hi = atanhi1; lo = atanlo1; x = (x - 1.0f) / (x + 1.0f); }
      } else {
         if (ix < 0x401c0000) { id = 2; hi = atanhi2; lo = atanlo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
         else                 { id = 3; hi = atanhi3; lo = atanlo3; x = -1.0f / x; }
      }
   }
   const float z = x * x;
   const float w = z * z;
   const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
   const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
   if (id < 0) return x - x * (s1 + s2);
   const float r = hi - ((x * (s1 + s2) - lo) - x);
   return (hx < 0) ? -r : r;
}

// ---- atan2f, fdlibm algorithm ----
HM_HD float hm_atan2f(float y, float x)
{
   const float tiny = 1.0e-30f;
   const float pi_o_4 = hm_u2f(0x3f490fdbu), pi_o_2 = hm_u2f(0x3fc90fdbu),
               pi = hm_u2f(0x40490fdbu), pi_lo = hm_u2f(0xb3bbbd2eu);
   const int32_t hx = (int32_t)hm_f2u(x), hy = (int32_t)hm_f2u(y);
   const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
   if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;   // NaN
   if (hx == 0x3f800000) return hm_atanf(y);               // x == 1
   const int m = (int)(((uint32_t)hy >> 31) & 1u) | (int)(((uint32_t)hx >> 30) & 2u);
   if (iy == 0) {
      switch (m) {
         case 0: case 1: return y;
         case 2: return pi + tiny;
         default: return -pi - tiny;
      }
   }
   if (ix == 0) return (hy < 0) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);
   if (ix == 0x7f800000) {
      if (iy == 0x7f800000) {
         switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
         }
      } else {
         switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
         }
      }
   }
   if (iy == 0x7f800000) return (hy < 0) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);
   const int32_t k = (iy - ix) >> 23;
   float z;
   if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
   else if (hx < 0 && k < -60) z = 0.0f;
   else z = hm_atanf(hm_fabsf(y / x));
   switch (m) {
      case 0: return z;
      case 1: return hm_u2f(hm_f2u(z) ^ 0x80000000u);
      case 2: return pi - (z - pi_lo);
      default: return (z - pi_lo) - pi;
   }
}

// ---- powf(2.0f, y) for |y| < 126 (no overflow/underflow handling needed on this path:
//      callers pass y = b/3 with |b| <= 1.5, or 1/numberOfScales) ----
// log2(2.0f) evaluates to exactly 1.0 in glibc's log2_inline (table entry for z == 1 has
// invc = 1, logc = 0, so r = 0 and the polynomial collapses to y0 = k = 1), hence
// ylogx = (double)y and the result is (float)exp2_inline(y).
HM_HD float hm_pow2f(float y)
{
   const uint64_t T[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
   const double C0 = hm_u2d(0x3fac6af84b912394ull);   // 0x1.c6af84b912394p-5
   const double C1 = hm_u2d(0x3fcebfce50fac4f3ull);   // 0x1.ebfce50fac4f3p-3
   const double C2 = hm_u2d(0x3fe62e42ff0c52d6ull);   // 0x1.62e42ff0c52d6p-1
   const double SHIFT = hm_u2d(0x42e8000000000000ull); // 0x1.8p+52 / 32
   const double xd = (double)y;
   double kd = xd + SHIFT;
   const uint64_t ki = hm_d2u(kd);
   kd -= SHIFT;
   const double r = xd - kd;
   uint64_t t = T[ki & 31];
   t += ki << (52 - 5);
   const double s = hm_u2d(t);
#ifdef HM_POW2F_FMA
   const double z = __builtin_fma(C0, r, C1);
   const double r2 = r * r;
   double p = __builtin_fma(C2, r, 1.0);
   p = __builtin_fma(z, r2, p);
#else
   const double z = C0 * r + C1;
   const double r2 = r * r;
   double p = C2 * r + 1.0;
   p = z * r2 + p;
#endif
   p = p * s;
   return (float)p;
}
