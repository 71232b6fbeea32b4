// hesaff.hpp -- C++ host mirror of the reference's operator interface for the detect +
// describe path, implemented over the C ABI (include/hesaff_amd.h).  Names, argument
// meaning and outputs follow hesaff.cpp:21-131: HessianAffineParams (defaults :28-35),
// Keypoint (:41-48), AffineHessianDetector::{detectPyramidKeypoints, keys,
// exportKeypoints}.  The reference's per-keypoint virtual callbacks (pyramid.h:43-47,
// affine.h:48-58) do not exist here: the GPU runs the stages breadth-first and hands
// back the same `keys` vector, in the same order.
#pragma once
#include <cmath>
#include <cstdint>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hesaff_amd.h"

namespace hesaff_amd {

struct HessianAffineParams {   // hesaff.cpp:21-36
   float threshold;
   int max_iter;
   float desc_factor;
   int patch_size;
   bool verbose;
   HessianAffineParams()
   {
      threshold = 16.0f / 3.0f;
      max_iter = 16;
      desc_factor = 3.0f * std::sqrt(3.0f);
      patch_size = 41;
      verbose = false;
   }
};

typedef hesaff_keypoint Keypoint;   // hesaff.cpp:41-48, identical layout

struct AffineHessianDetector {
   std::vector<Keypoint> keys;      // hesaff.cpp:54
   int g_numberOfPoints = 0;        // hesaff.cpp:38
   int g_numberOfAffinePoints = 0;  // hesaff.cpp:39

   explicit AffineHessianDetector(const HessianAffineParams &par = HessianAffineParams(), int device = 0)
   {
      if (par.patch_size != 41) throw std::invalid_argument("patch_size is fixed at 41 in this build");
      // the structs carry no size field: refuse a library built from another header before passing one across
      if (hesaff_abi_version() != HESAFF_ABI_VERSION || hesaff_sizeof_params() != sizeof(hesaff_params) || hesaff_sizeof_timings() != sizeof(hesaff_timings))
         throw std::runtime_error("libhesaff_amd.so was built from a different include/hesaff_amd.h (ABI version mismatch)");
      hesaff_default_params(&p_);
      p_.threshold = par.threshold;          // hesaff.cpp:155
      p_.maxIterations = par.max_iter;       // hesaff.cpp:158
      p_.mrSize = par.desc_factor;           // hesaff.cpp:160
      p_.max_batch = 1;
      if (hesaff_create(&ctx_, &p_, device) != HESAFF_OK) throw std::runtime_error(hesaff_last_error(nullptr));
   }
   ~AffineHessianDetector() { hesaff_destroy(ctx_); }
   AffineHessianDetector(const AffineHessianDetector &) = delete;
   AffineHessianDetector &operator=(const AffineHessianDetector &) = delete;

   // == grey conversion hesaff.cpp:138-148 + detectPyramidKeypoints hesaff.cpp:167 with the
   // whole callback chain; image is what cv::imread would deliver (8-bit, 1 or 3 channels).
   void detectPyramidKeypoints(const uint8_t *image, int width, int height, int channels)
   {
      hesaff_result r;
      const int stride = width * channels;
      if (hesaff_detect_batch(ctx_, 1, &image, &width, &height, &stride, &channels, &r) != HESAFF_OK)
         throw std::runtime_error(hesaff_last_error(ctx_));
      g_numberOfPoints = r.count_hessian;
      g_numberOfAffinePoints += r.count_desc;   // the reference never resets this counter (hesaff.cpp:166)
      keys.assign(r.keys, r.keys + r.count_desc);
   }

   // hesaff.cpp:107-130
   void exportKeypoints(std::ostream &out)
   {
      char *buf = nullptr;
      size_t len = 0;
      if (hesaff_format_sift_mt(keys.data(), (int)keys.size(), p_.mrSize, 0, &buf, &len) != HESAFF_OK)   // rows on all host cores
         throw std::runtime_error("hesaff_format_sift_mt failed");
      out.write(buf, (std::streamsize)len);
      out.flush();
      hesaff_free(buf);
   }

 private:
   hesaff_params p_;
   hesaff_ctx *ctx_ = nullptr;
};

} // namespace hesaff_amd
