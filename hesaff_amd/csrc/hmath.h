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
// tests/test_host_side.py (test_hmath_restatements_equal_glibc) checks both against this image's libm on a host
// build; tests/test_gpu_parity.py checks device == libm (test_device_math_matches_libm) and the kernel's range-free
// forms == the general ones (test_range_free_gradient_forms_equal_the_general_ones).
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
   // argument reduction: x' = num / den with (num, den) chosen per interval -- ONE division
   // (fdlibm writes four branches with a division each; the quotients are the same)
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
      float num, den;
      if (ix < 0x3f980000) {               // |x| < 1.1875
         if (ix < 0x3f300000) { id = 0; hi = atanhi0; lo = atanlo0; num = 2.0f * x - 1.0f; den = 2.0f + x; }
         else                 { id = 1; hi = atanhi1; lo = atanlo1; num = x - 1.0f; den = x + 1.0f; }
      } else {
         if (ix < 0x401c0000) { id = 2; hi = atanhi2; lo = atanlo2; num = x - 1.5f; den = 1.0f + 1.5f * x; }
         else                 { id = 3; hi = atanhi3; lo = atanlo3; num = -1.0f; den = x; }
      }
      x = num / den;
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

// ---- atan2f again, written with selects instead of branches for the wide-SIMD device
//      (a wavefront that diverges over fdlibm's interval branches executes every one of them,
//      each with its own division).  Same operations on the taken path, hence the same bits:
//        * atanf(|y/x|): (num, den, hi, lo) of the interval are selected, ONE division; the
//          "|x| < 0.4375" interval is num = q, den = 1, hi = lo = 0, where
//          hi - ((t - lo) - x) == x - t exactly; fdlibm's "|x| < 2^-29 -> x" shortcut returns
//          what the polynomial path returns anyway (x*x*... is below half an ulp of x);
//        * the quadrant fix-up  m = 1: -z,  m = 3: (z - pi_lo) - pi == -(pi - (z - pi_lo));
//        * y == 0 with x != 0 needs no special case (q = 0 gives +-0 / +-pi);
//        * x == 1 needs none either (y / 1 == y and atanf is odd in every interval).
//      Inputs fdlibm treats separately and gradients never produce (NaN, infinities, exponent
//      gaps above 60) leave through hm_atan2f.  Checked against libm in tests/test_host_side.py.
HM_HD float hm_atan2f_sel(float y, float x)
{
   const float pi_o_2 = hm_u2f(0x3fc90fdbu), pi = hm_u2f(0x40490fdbu), pi_lo = hm_u2f(0xb3bbbd2eu);
   const int32_t hx = (int32_t)hm_f2u(x), hy = (int32_t)hm_f2u(y);
   const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
   const int32_t k = (iy - ix) >> 23;
   if ((ix >= 0x7f800000) | (iy >= 0x7f800000) | ((iy != 0) & (ix != 0) & ((k > 60) | (k < -60)))) return hm_atan2f(y, x);
   const float q = hm_fabsf(y / x);   // NaN for 0/0, +inf for y/0: both replaced below
   const int32_t iq = (int32_t)hm_f2u(q);
   const bool i0 = iq < 0x3f300000, i1 = iq < 0x3f980000, i2 = iq < 0x401c0000, im = iq < 0x3ee00000;
   // fdlibm s_atanf.c: (2x-1)/(2+x) | (x-1)/(x+1) | (x-1.5)/(1+1.5x) | -1/x
   float num = i1 ? (i0 ? 2.0f * q - 1.0f : q - 1.0f) : (i2 ? q - 1.5f : -1.0f);
   float den = i1 ? (i0 ? 2.0f + q : q + 1.0f) : (i2 ? 1.0f + 1.5f * q : q);
   float hi = i1 ? (i0 ? hm_u2f(0x3eed6338u) : hm_u2f(0x3f490fdau)) : (i2 ? hm_u2f(0x3f7b985eu) : hm_u2f(0x3fc90fdau));
   float lo = i1 ? (i0 ? hm_u2f(0x31ac3769u) : hm_u2f(0x33222168u)) : (i2 ? hm_u2f(0x33140fb4u) : hm_u2f(0x33a22168u));
   num = im ? q : num;
   den = im ? 1.0f : den;
   hi = im ? 0.0f : hi;
   lo = im ? 0.0f : lo;
   const float xr = num / den;
   const float z = xr * xr;
   const float w = z * z;
   const float aT0 = hm_u2f(0x3eaaaaabu), aT1 = hm_u2f(0xbe4ccccdu), aT2 = hm_u2f(0x3e124925u),
               aT3 = hm_u2f(0xbde38e38u), aT4 = hm_u2f(0x3dba2e6eu), aT5 = hm_u2f(0xbd9d8795u),
               aT6 = hm_u2f(0x3d886b35u), aT7 = hm_u2f(0xbd6ef16bu), aT8 = hm_u2f(0x3d4bda59u),
               aT9 = hm_u2f(0xbd15a221u), aT10 = hm_u2f(0x3c8569d7u);
   const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
   const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
   float za = hi - ((xr * (s1 + s2) - lo) - xr);
   za = (iq >= 0x4c000000) ? hm_u2f(0x3fc90fdau) + hm_u2f(0x33a22168u) : za;   // |q| >= 2^25
   float r = (hx < 0) ? pi - (za - pi_lo) : za;
   // x == 0: +-pi/2 (pi_o_2 + tiny), or, with y == 0 too, y itself / +-pi by the sign of x
   r = (ix == 0) ? ((iy == 0) ? ((hx < 0) ? pi : 0.0f) : pi_o_2) : r;
   return hm_u2f(hm_f2u(r) ^ ((uint32_t)hy & 0x80000000u));
}


// ---- atan2f a third time, table-driven: the per-interval constants of hm_atan2f_sel come from one 8-float row
//      (HM_ATAN_TAB, staged in LDS by the device caller) instead of a tree of selects:
//         num = c1*q + c0,   den = d1*q + d0,   z = hi - ((xr*(s1+s2) - lo) - xr),   xr = num/den
//      which are fdlibm's expressions with the same roundings:
//         |q| < 0.4375 : q / 1                    (c1, c0, d1, d0) = (1, 0, 0, 1)     1*q, q + 0, 0*q + 1 are exact
//         < 0.6875     : (2q - 1) / (2 + q)                          (2, -1, 1, 2)
//         < 1.1875     : (q - 1) / (q + 1)                           (1, -1, 1, 1)
//         < 2.4375     : (q - 1.5) / (1 + 1.5q)                      (1, -1.5, 1.5, 1)
//         otherwise    : -1 / q                                      (0, -1, 1, 0)    0*q - 1 = -1, q + 0 = q
//      Row layout: c1, d1, c0, d0, hi, lo, 0, 0 (the (c1, d1) and (c0, d0) pairs feed packed operations).
//      Everything else is hm_atan2f_sel's code; checked against it and libm in tests/test_host_side.py.
#define HM_ATAN_TAB_FLOATS 40
#define HM_ATAN_TAB_INIT                                                                                           \
   {1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f,                                                               \
    2.0f, 1.0f, -1.0f, 2.0f, 0x1.dac67p-2f /*3eed6338*/, 0x1.586ed2p-28f /*31ac3769*/, 0.0f, 0.0f,                  \
    1.0f, 1.0f, -1.0f, 1.0f, 0x1.921fb4p-1f /*3f490fda*/, 0x1.4442dp-25f /*33222168*/, 0.0f, 0.0f,                  \
    1.0f, 1.5f, -1.5f, 1.0f, 0x1.f730bcp-1f /*3f7b985e*/, 0x1.281f68p-25f /*33140fb4*/, 0.0f, 0.0f,                 \
    0.0f, 1.0f, -1.0f, 0.0f, 0x1.921fb4p+0f /*3fc90fda*/, 0x1.4442dp-24f /*33a22168*/, 0.0f, 0.0f}

// the operands the select / table forms hand over (NaN, infinities, exponent gaps above 60) are rare: keep fdlibm's
// branchy general form out of line on the device so that it does not bloat every call site
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline)) inline float hm_atan2f_rare(float y, float x) { return hm_atan2f(y, x); }
#else
inline float hm_atan2f_rare(float y, float x) { return hm_atan2f(y, x); }
#endif

// IEEE division and square root for operands that need no range handling.  The compiler's correctly rounded f32
// division is  v_div_scale x 2, v_rcp, five fma / mul steps, v_div_fmas, v_div_fixup: the scale and fix-up instructions
// only act on denormal / huge operands and quotients and on zero, infinite or NaN inputs.  With both operands and the
// quotient in the normal range (or a zero numerator) the remaining steps ARE the division: same intermediate values,
// same result.  Likewise sqrtf = v_sqrt_f32 (1 ulp) + the choice between the result and its two neighbours by the
// sign of two fma residuals, framed by a 2^32 scaling for arguments below 2^-96 and a class test for 0 / inf.
// hm_div_normal with a zero divisor returns NaN, not +-inf (callers replace that lane's result anyway).
HM_HD float hm_div_normal(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
   const float r0 = __builtin_amdgcn_rcpf(b);
   const float e0 = __builtin_fmaf(-b, r0, 1.0f);
   const float r1 = __builtin_fmaf(e0, r0, r0);
   const float q0 = a * r1;
   const float e1 = __builtin_fmaf(-b, q0, a);
   const float q1 = __builtin_fmaf(e1, r1, q0);
   const float e2 = __builtin_fmaf(-b, q1, a);
   return __builtin_fmaf(e2, r1, q1);
#else
   return a / b;
#endif
}
HM_HD float hm_sqrt_normal(float x)   // x == 0 or normal, finite
{
#if defined(__HIP_DEVICE_COMPILE__)
   float s = __builtin_amdgcn_sqrtf(x);
   const float sd = hm_u2f(hm_f2u(s) - 1u), su = hm_u2f(hm_f2u(s) + 1u);
   const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
   s = (vp <= 0.0f) ? sd : s;
   s = (vs > 0.0f) ? su : s;
   return s;
#else
   return __builtin_sqrtf(x);
#endif
}

// ND = true: y and x are zero or normal numbers (never denormal): the two divisions need no range handling.  k_sift_grad's
// operands are differences of pixel values that are multiples of 2^-17 in [0, 255].
template <bool ND>
HM_HD float hm_atan2f_tab_t(float y, float x, const float *tab)
{
   const float pi_o_2 = hm_u2f(0x3fc90fdbu), pi = hm_u2f(0x40490fdbu), pi_lo = hm_u2f(0xb3bbbd2eu);
   const int32_t hx = (int32_t)hm_f2u(x), hy = (int32_t)hm_f2u(y);
   const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
   const int32_t k = (iy - ix) >> 23;
   if ((ix >= 0x7f800000) | (iy >= 0x7f800000) | ((iy != 0) & (ix != 0) & ((k > 60) | (k < -60)))) return hm_atan2f_rare(y, x);
   const float q = hm_fabsf(ND ? hm_div_normal(y, x) : y / x);   // NaN for 0/0, +inf (ND: NaN) for y/0: both replaced below
   const int32_t iq = (int32_t)hm_f2u(q);
   const int id = (int)(iq >= 0x3ee00000) + (int)(iq >= 0x3f300000) + (int)(iq >= 0x3f980000) + (int)(iq >= 0x401c0000);
   const float *row = tab + 8 * id;
   const float num = row[0] * q + row[2];
   const float den = row[1] * q + row[3];
   const float hi = row[4], lo = row[5];
   const float xr = ND ? hm_div_normal(num, den) : num / den;
   const float z = xr * xr;
   const float w = z * z;
   const float aT0 = hm_u2f(0x3eaaaaabu), aT1 = hm_u2f(0xbe4ccccdu), aT2 = hm_u2f(0x3e124925u),
               aT3 = hm_u2f(0xbde38e38u), aT4 = hm_u2f(0x3dba2e6eu), aT5 = hm_u2f(0xbd9d8795u),
               aT6 = hm_u2f(0x3d886b35u), aT7 = hm_u2f(0xbd6ef16bu), aT8 = hm_u2f(0x3d4bda59u),
               aT9 = hm_u2f(0xbd15a221u), aT10 = hm_u2f(0x3c8569d7u);
   const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
   const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
   float za = hi - ((xr * (s1 + s2) - lo) - xr);
   za = (iq >= 0x4c000000) ? hm_u2f(0x3fc90fdau) + hm_u2f(0x33a22168u) : za;   // |q| >= 2^25
   float r = (hx < 0) ? pi - (za - pi_lo) : za;
   // x == 0: +-pi/2 (pi_o_2 + tiny), or, with y == 0 too, y itself / +-pi by the sign of x
   r = (ix == 0) ? ((iy == 0) ? ((hx < 0) ? pi : 0.0f) : pi_o_2) : r;
   return hm_u2f(hm_f2u(r) ^ ((uint32_t)hy & 0x80000000u));
}
HM_HD float hm_atan2f_tab(float y, float x, const float *tab) { return hm_atan2f_tab_t<false>(y, x, tab); }
HM_HD float hm_atan2f_tab_nd(float y, float x, const float *tab) { return hm_atan2f_tab_t<true>(y, x, tab); }

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

// ---- orientation coordinate of the SIFT histogram, siftdesc.cpp:65:
//        o = float( float(orientationBins) * (ori + 2*M_PI) / (2*M_PI) ),   evaluated in double
//      The double division by the constant 2*pi is replaced by Markstein's sequence
//        q = a*y;  r = fma(-q, b, a);  q' = fma(r, y, q),   y = RN(1/b)
//      which returns the correctly rounded quotient RN(a/b) for this b (checked exhaustively
//      for every float `ori` in [-3.2, 3.2], tests/test_host_side.py keeps a sampled check).
HM_HD float hm_sift_orient_coord(float ori)
{
   const double twopi = hm_u2d(0x401921fb54442d18ull);      // 2 * M_PI
   const double inv_twopi = hm_u2d(0x3fc45f306dc9c883ull);  // RN(1 / (2 * M_PI))
   const double a = 8.0 * ((double)ori + twopi);
   const double q = a * inv_twopi;
   const double r = __builtin_fma(-q, twopi, a);
   const double q2 = __builtin_fma(r, inv_twopi, q);
   return (float)q2;
}

