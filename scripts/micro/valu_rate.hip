// valu_rate.hip -- issue rate of a few VALU instruction kinds on gfx950 (cycles per wave64 instruction per SIMD).
// hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND> __global__ void k(float *out, int iters)
{
   float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
   v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
   double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
   const float c = 1.0000001f;
   const v2f cc = {c, c};
   const double dc = 1.0000001;
   for (int i = 0; i < iters; i++) {
      if (KIND == 0) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 1) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));) }
      if (KIND == 2) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));) }
      if (KIND == 3) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));) }
      if (KIND == 4) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
      if (KIND == 5) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 6) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));) }
      if (KIND == 7) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
      if (KIND == 9) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s10", "s11");) }
      if (KIND == 10) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 11) { REP8(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
      if (KIND == 12) { REP8(asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
      if (KIND == 13) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 14) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
      if (KIND == 15) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 16) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 17) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));) }
      if (KIND == 18) { REP8(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %8\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %8\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %8\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(c));) }
      if (KIND == 19) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_fmas_f32 %1, %1, %8, %1\n v_div_fixup_f32 %2, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_fmas_f32 %4, %4, %8, %4\n v_div_fixup_f32 %5, %5, %8, %5\n v_div_fixup_f32 %6, %6, %8, %6\n v_div_fixup_f32 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
      if (KIND == 20) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
      if (KIND == 21) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
      if (KIND == 22) { REP8(asm volatile("v_cmp_gt_f32 s[10:11], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cmp_gt_f32 s[12:13], %2, %8\n v_cndmask_b32_e64 %3, %3, %8, s[12:13]\n v_cmp_gt_f32 s[10:11], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cmp_gt_f32 s[12:13], %6, %8\n v_cndmask_b32_e64 %7, %7, %8, s[12:13]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s10", "s11", "s12", "s13");) }
      if (KIND == 23) { REP8(asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 24) { REP8(asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (KIND == 8) { REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int KIND> void run(const char *name, float *d)
{
   const int iters = 2000, blocks = 256 * 8, threads = 256;   // 8 waves per SIMD
   hipEvent_t a, b;
   hipEventCreate(&a); hipEventCreate(&b);
   hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 10);
   hipEventRecord(a);
   hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters);
   hipEventRecord(b);
   hipEventSynchronize(b);
   float ms = 0;
   hipEventElapsedTime(&ms, a, b);
   const double winst = (double)blocks * (threads / 64) * iters * 64.0;   // wave-instructions
   // cycles per wave-instruction per SIMD at 2.4 GHz: time * 2.4e9 * 1024 SIMDs / wave-instructions
   printf("%-16s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024.0 / winst);
}
int main()
{
   float *d;
   hipMalloc(&d, 256 * 8 * 256 * 4);
   run<0>("v_mul_f32", d); run<5>("v_fma_f32", d); run<1>("v_pk_mul_f32", d); run<2>("v_pk_add_f32", d); run<3>("v_fma_f64", d); run<6>("v_mul_f64", d);
   run<20>("cndmask e64 vcc", d); run<21>("cmp+cndmask vcc", d); run<22>("cmp+cndmask sgpr", d); run<23>("v_max/min_f32", d); run<24>("v_bfi_b32", d);
   run<7>("v_cndmask vcc", d); run<9>("v_cndmask sgpr", d); run<4>("v_rcp_f32", d); run<8>("v_sqrt_f32", d);
   run<10>("v_add_f32", d); run<11>("v_floor_f32", d); run<12>("v_cvt_i32_f32", d); run<13>("v_and_b32", d); run<14>("v_cmp_gt_f32", d);
   run<15>("v_mad_u32_u24", d); run<16>("v_mul_lo_u32", d); run<17>("v_pk_fma_f32", d); run<18>("v_cvt_f64_f32", d); run<19>("div helpers", d);
   return 0;
}
