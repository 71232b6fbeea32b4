// host_tables.h -- host-side constant tables of libhesaff_amd (uploaded once per context).
// These are tiny setup computations the reference does in its constructors; they call the
// host libm (expf / exp) exactly like the reference, so they stay on the host.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace hesaff {

// helpers.cpp:286,293 : kernel size from sigma, made odd
inline int gauss_ksize(float sigma)
{
   int size = (int)(2.0 * 3.0 * sigma + 1.0);
   if (size % 2 == 0) size++;
   return size;
}

// Taps of cv::getGaussianKernel(n, sigma, CV_32F) (OpenCV 2.4 imgproc/smooth.cpp, sigma > 0):
// double exp, float store, double sum of the stored floats, float(cf * (1/sum)).
inline void gauss_taps(int n, float sigma, float *cf)
{
   const double sigmaX = (double)sigma;
   const double scale2X = -0.5 / (sigmaX * sigmaX);
   double sum = 0;
   for (int i = 0; i < n; i++) {
      const double x = i - (n - 1) * 0.5;
      const double t = std::exp(scale2X * x * x);
      cf[i] = (float)t;
      sum += cf[i];
   }
   sum = 1. / sum;
   for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

// helpers.cpp:104-129 computeGaussMask (size x size, separable, tails folded in)
inline void gauss_mask(int size, float *mask)
{
   const int half = size >> 1;
   const float scale = float(half) / 3.0f;
   const float scale2 = -2.0f * scale * scale;
   std::vector<float> tmp(half + 1);
   for (int i = 0; i <= half; i++) tmp[i] = expf(float(i * i) / scale2);
   const int endSize = int(ceilf(scale * 5.0f) - half);
   for (int i = 1; i < endSize; i++) tmp[half - i] += expf(float((i + half) * (i + half)) / scale2);
   for (int i = -half; i <= half; i++)
      for (int j = -half; j <= half; j++) mask[(i + half) * size + (j + half)] = tmp[i < 0 ? -i : i] * tmp[j < 0 ? -j : j];
}

// helpers.cpp:131-147 computeCircularGaussMask
inline void circ_gauss_mask(int size, float *mask)
{
   const int half = size >> 1;
   const float r2 = float(half * half);
   const float sigma2 = 0.9f * r2;
   for (int i = 0; i < size; i++)
      for (int j = 0; j < size; j++) {
         const float disq = float((i - half) * (i - half) + (j - half) * (j - half));
         mask[i * size + j] = (disq < r2) ? expf(-disq / sigma2) : 0.0f;
      }
}

// siftdesc.cpp:18-49 precomputeBinsAndWeights for patchSize 41, 4 spatial / 8 orientation bins
inline void sift_bins(int32_t *bin0, int32_t *bin1, float *w0, float *w1)
{
   const int ps = 41, sb = 4, ob = 8, half = ps >> 1;
   const float step = float(sb + 1) / (2 * half);
   for (int i = 0; i < ps; i++) {
      const float x = step * i;
      const int xi = (int)(x);
      bin0[i] = xi - 1;
      bin1[i] = xi;
      w1[i] = x - xi;
      w0[i] = 1.0f - w1[i];
      if (bin0[i] < 0) { bin0[i] = 0; w0[i] = 0; }
      if (bin0[i] >= sb) { bin0[i] = sb - 1; w0[i] = 0; }
      if (bin1[i] < 0) { bin1[i] = 0; w1[i] = 0; }
      if (bin1[i] >= sb) { bin1[i] = sb - 1; w1[i] = 0; }
      bin0[i] *= ob;
      bin1[i] *= ob;
   }
}

// The scale schedule of one octave, pyramid.cpp:224-259 (float arithmetic as written there)
struct OctaveSchedule {
   float init_sigma;        // pyramid.cpp:278  sqrt(1.6^2 - 0.5^2)
   float blur_sigma[5];     // [i], i = 1..4 : curSigma * sqrt(step^2 - 1)
   float level_sigma[5];    // [0] = initialSigma, [i] = curSigma * step
   float norm2[5];          // (sigma*sigma)^2 handed to hessianResponse (pyramid.cpp:76)
};
OctaveSchedule make_schedule(float initialSigma, bool upscale);

} // namespace hesaff
