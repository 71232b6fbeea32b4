// fast_api.h -- launchers of the fast-mode kernels (kernels_fast.hip, hesaff_params.fast = 1).
// The fast translation unit compiles the SAME per-keypoint kernel sources with contraction and approximate
// division / square root allowed and HS_FAST set (shuffle-tree sums, device-library atan2f, fused mean/variance);
// its kernels live in namespace hsfast.  The argument structs are passed as untyped pointers: both translation
// units define them from the same headers, so the layouts are identical (sizes are checked on the callee side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct FastArgs {
   const void *hl, *pw, *io, *tb, *kc, *pt, *ao, *so;   // HessList, PatchWork, PatchIO, KpTables, DConsts, PlaneTab, AffineOut, SiftIO
   size_t sz_hl, sz_pw, sz_io, sz_tb, sz_kc, sz_pt, sz_ao, sz_so;
};

void hsfast_affine(hipStream_t st, uint32_t grid, const FastArgs &a, uint32_t h_lo, uint32_t h_hi, const uint32_t *n_ptr);
void hsfast_patch_bins(hipStream_t s0, hipStream_t s1, hipStream_t s2, hipStream_t s3, const FastArgs &a, const void *io_mid, const void *io_big,
                       const uint32_t grids[4], const size_t lds[4]);
void hsfast_patch_large(hipStream_t st, const FastArgs &a, uint32_t *row_prefix, uint32_t gblocks, size_t lds, int srow_stride, int tap_stride, uint32_t g_finish);
void hsfast_sift(hipStream_t st, const FastArgs &a, uint32_t n, void *vo, uint32_t g_grad, uint32_t g_hist);
void hsfast_set_attrs(const size_t lds[4], size_t lds_large);
// fast level 2: windows of bins first_bin.. sampled from the scale-space level with the matching blur (k_patch_pyramid)
void hsfast_patch_pyramid(hipStream_t st, const FastArgs &a, int n_octaves, float pd0, int first_bin, uint32_t grid);
void hsfast_patch_bin0(hipStream_t st, const FastArgs &a, uint32_t grid, size_t lds);
