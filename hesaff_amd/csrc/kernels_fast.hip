// kernels_fast.hip -- the fast mode of the per-keypoint stages (hesaff_params.fast = 1).
// NOT bit-exact with the reference: this translation unit is compiled with -ffp-contract=fast and approximate f32
// division / square root, and HS_FAST selects shuffle-tree sums (second-moment matrix, photometric mean / variance),
// the device library's atan2f and a float orientation coordinate.  Everything upstream of the keypoint list (pyramid,
// extrema, localisation, ordering) and the window geometry (k_prepare_patch: double-precision rectification, border
// and window tests) stay on the parity kernels, so both modes work on the same Hessian keypoints.
// The mismatch against parity mode is measured by tools/fast_mode_report.py (profiles/r02_fast_mode.json).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

#define HS_FAST 1
namespace hsfast {
#include "kernels_sift.h"   // pulls in kernels_patch.h, kernels_keypoint.h, kernels_pyramid.h, device_common.h, hmath.h
}
#include "fast_api.h"

namespace {
template <class T> T take(const void *p, size_t sz)
{
   static_assert(std::is_trivially_copyable<T>::value, "argument structs are plain data");
   T t;
   if (sz != sizeof(T)) abort();   // the two translation units disagree about a struct: a build error, not a run-time condition
   memcpy(&t, p, sizeof(T));
   return t;
}
} // namespace

void hsfast_set_attrs(const size_t lds[4], size_t lds_large)
{
   (void)hipFuncSetAttribute((const void *)hsfast::k_patch_extract_small<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds[0]);
   (void)hipFuncSetAttribute((const void *)hsfast::k_patch_extract_small<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds[1]);
   (void)hipFuncSetAttribute((const void *)hsfast::k_patch_mid<HS_MID_PMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds[2]);
   (void)hipFuncSetAttribute((const void *)hsfast::k_patch_mid<HS_BIN3_PMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds[3]);
   if (lds_large) (void)hipFuncSetAttribute((const void *)hsfast::k_patch_large_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_large);
}

void hsfast_affine(hipStream_t st, uint32_t grid, const FastArgs &a, uint32_t h_lo, uint32_t h_hi, const uint32_t *n_ptr)
{
   hipLaunchKernelGGL(hsfast::k_affine, dim3(grid), dim3(64), 0, st, take<hsfast::PlaneTab>(a.pt, a.sz_pt), take<hsfast::HessList>(a.hl, a.sz_hl), h_lo, h_hi, n_ptr,
                      take<hsfast::KpTables>(a.tb, a.sz_tb), take<hsfast::DConsts>(a.kc, a.sz_kc), take<hsfast::AffineOut>(a.ao, a.sz_ao));
}

void hsfast_patch_bins(hipStream_t s0, hipStream_t s1, hipStream_t s2, hipStream_t s3, const FastArgs &a, const void *io_mid, const void *io_big,
                       const uint32_t grids[4], const size_t lds[4])
{
   const hsfast::HessList hl = take<hsfast::HessList>(a.hl, a.sz_hl);
   const hsfast::PatchWork pw = take<hsfast::PatchWork>(a.pw, a.sz_pw);
   const hsfast::KpTables tb = take<hsfast::KpTables>(a.tb, a.sz_tb);
   hipLaunchKernelGGL(hsfast::k_patch_extract_small<0>, dim3(grids[0]), dim3(256), lds[0], s0, hl, pw, take<hsfast::PatchIO>(a.io, a.sz_io), tb);
   hipLaunchKernelGGL(hsfast::k_patch_extract_small<1>, dim3(grids[1]), dim3(256), lds[1], s1, hl, pw, take<hsfast::PatchIO>(a.io, a.sz_io), tb);
   hipLaunchKernelGGL(hsfast::k_patch_mid<HS_MID_PMAX>, dim3(grids[2]), dim3(256), lds[2], s2, hl, pw, take<hsfast::PatchIO>(io_mid, a.sz_io), tb);
   hipLaunchKernelGGL(hsfast::k_patch_mid<HS_BIN3_PMAX>, dim3(grids[3]), dim3(256), lds[3], s3, hl, pw, take<hsfast::PatchIO>(io_big, a.sz_io), tb);
}

void hsfast_patch_large(hipStream_t st, const FastArgs &a, uint32_t *row_prefix, uint32_t gblocks, size_t lds, int srow_stride, int tap_stride, uint32_t g_finish)
{
   const hsfast::HessList hl = take<hsfast::HessList>(a.hl, a.sz_hl);
   const hsfast::PatchWork pw = take<hsfast::PatchWork>(a.pw, a.sz_pw);
   const hsfast::KpTables tb = take<hsfast::KpTables>(a.tb, a.sz_tb);
   const hsfast::PatchIO io = take<hsfast::PatchIO>(a.io, a.sz_io);
   hipLaunchKernelGGL(hsfast::k_large_prefix, dim3(1), dim3(256), 0, st, pw, row_prefix);
   hipLaunchKernelGGL(hsfast::k_patch_large_rows, dim3(gblocks), dim3(256), lds, st, hl, pw, io, tb, srow_stride, tap_stride);
   hipLaunchKernelGGL(hsfast::k_patch_large_finish, dim3(g_finish), dim3(256), 0, st, pw, io, tb);
}

void hsfast_sift(hipStream_t st, const FastArgs &a, uint32_t n, void *vo, uint32_t g_grad, uint32_t g_hist)
{
   const hsfast::SiftIO so = take<hsfast::SiftIO>(a.so, a.sz_so);
   const hsfast::KpTables tb = take<hsfast::KpTables>(a.tb, a.sz_tb);
   const uint32_t nb64 = (n + 63) / 64;
   (void)nb64; (void)vo; (void)g_hist;
   // one kernel from the patch to the 128 bytes: nothing but the descriptor leaves the chip (k_desc_fused, kernels_sift.h)
   hipLaunchKernelGGL(hsfast::k_desc_fused, dim3(std::min(n, g_grad)), dim3(256), 0, st, so, tb, take<hsfast::DConsts>(a.kc, a.sz_kc));
}

void hsfast_patch_bin0(hipStream_t st, const FastArgs &a, uint32_t grid, size_t lds)
{
   hipLaunchKernelGGL(hsfast::k_patch_extract_small<0>, dim3(grid), dim3(256), lds, st, take<hsfast::HessList>(a.hl, a.sz_hl), take<hsfast::PatchWork>(a.pw, a.sz_pw),
                      take<hsfast::PatchIO>(a.io, a.sz_io), take<hsfast::KpTables>(a.tb, a.sz_tb));
}

void hsfast_patch_pyramid(hipStream_t st, const FastArgs &a, int n_octaves, float pd0, int first_bin, uint32_t grid)
{
   hipLaunchKernelGGL(hsfast::k_patch_pyramid, dim3(grid), dim3(256), 0, st, take<hsfast::HessList>(a.hl, a.sz_hl), take<hsfast::PatchWork>(a.pw, a.sz_pw),
                      take<hsfast::PatchIO>(a.io, a.sz_io), take<hsfast::PlaneTab>(a.pt, a.sz_pt), n_octaves, pd0, first_bin);
}
