// hesaff_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A plain, single-threaded C++ restatement of the arithmetic of perdoch/hesaff's hot path
// (Gaussian scale space + det-of-Hessian, 3x3x3 extrema + localisation, Baumberg affine
// iteration, affine patch normalisation, SIFT), written from the reference's behaviour to
// the level of float operation order.  Every function cites the reference file:line it
// follows.  It exists only so that tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can check / time the HIP path against it; nothing under hesaff_amd/
// may include, link or call it.
//
// PARITY STATUS: *parity unpinned*.  The reference ships no tests, fixtures or golden
// vectors (14 files, none of them data), and it cannot be built in this image: all five
// sources include <cv.h>/<highgui.h> of OpenCV >= 2.3.1 (README:17, Makefile:2), which is
// not installed and is not vendored in /root/reference.  The separable Gaussian
// (cv::GaussianBlur, called at helpers.cpp:287,294) is therefore restated here from
// OpenCV 2.4's published algorithm (imgproc/src/smooth.cpp getGaussianKernel +
// filter.cpp RowFilter / SymmRowSmallFilter / SymmColumnFilter, scalar == SSE order).
// libm calls (expf, powf, atan2f, sqrt) go to the host glibc exactly as the reference's
// do.  Build with -ffp-contract=off (the reference's own Makefile targets baseline x86-64,
// which has no FMA).
//
// Build: see oracle/Makefile  ->  oracle/libhesaff_oracle.so
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <malloc.h>
#include <string>
#include <vector>

namespace {

struct Plane {
   int rows = 0, cols = 0;
   std::vector<float> d;
   Plane() {}
   Plane(int r, int c) : rows(r), cols(c), d((size_t)r * c, 0.0f) {}
   float &at(int r, int c) { return d[(size_t)r * cols + c]; }
   float at(int r, int c) const { return d[(size_t)r * cols + c]; }
   const float *row(int r) const { return &d[(size_t)r * cols]; }
   float *row(int r) { return &d[(size_t)r * cols]; }
};

// ----------------------------------------------------------------------------------------
// Parameters (defaults of pyramid.h:32-40, affine.h:37-45, siftdesc.h:25-30, hesaff.cpp:28-35)
// ----------------------------------------------------------------------------------------
struct Params {
   int numberOfScales = 3;
   float initialSigma = 1.6f;
   float threshold = 16.0f / 3.0f;
   float edgeEigenValueRatio = 10.0f;
   int border = 5;
   int maxIterations = 16;
   float convergenceThreshold = 0.05f;   // affine.h:41 (double literal stored to float)
   int patchSize = 41;
   int smmWindowSize = 19;
   float mrSize = 3.0f * sqrtf(3.0f);    // hesaff.cpp:32 (std::sqrt(float))
   float affInitialSigma = 1.6f;
   int spatialBins = 4;
   int orientationBins = 8;
   float maxBinValue = 0.2f;
   int upscaleInputImage = 0;            // pyramid.h:34
};

// ----------------------------------------------------------------------------------------
// cv::GaussianBlur restatement (third-party: OpenCV 2.4.x, see header)
// ----------------------------------------------------------------------------------------
// helpers.cpp:286,293 : kernel size from sigma
int gaussKsize(float sigma)
{
   int size = (int)(2.0 * 3.0 * sigma + 1.0);
   if (size % 2 == 0) size++;
   return size;
}

// OpenCV getGaussianKernel(n, sigma, CV_32F), sigma > 0 branch
void gaussKernel(int n, float sigmaf, float *cf)
{
   const double sigmaX = (double)sigmaf;
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

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// helpers.cpp:283-295 -> cv::GaussianBlur(..., BORDER_REPLICATE), float32 single channel
// (target_clones: the same loops compiled for wider vectors where the CPU has them, chosen at load time; element-wise IEEE multiplies and adds,
//  no fused multiply-add - "avx2" does not include FMA and contraction is off - so every clone computes the same bits)
__attribute__((target_clones("avx512f", "avx2", "default")))
void gaussianBlur(const Plane &in, float sigma, Plane &out)
{
   const int n = gaussKsize(sigma);
   const int rows = in.rows, cols = in.cols;
   Plane res(rows, cols);
   if (n == 1) { res.d = in.d; out = res; return; }
   std::vector<float> k(n);
   gaussKernel(n, sigma, k.data());
   const int r = n / 2;
   Plane tmp(rows, cols);
   std::vector<float> pad;
   // row pass
   for (int y = 0; y < rows; y++) {
      const float *S = in.row(y);
      float *T = tmp.row(y);
      if (n <= 5) {
         // SymmRowSmallFilter (ksize 3 or 5, symmetric kernel)
         const float k0 = k[r], k1 = k[r + 1], k2 = (n == 5) ? k[r + 2] : 0.0f;
         for (int x = 0; x < cols; x++) {
            const float s0 = S[x];
            const float sm1 = S[clampi(x - 1, 0, cols - 1)], sp1 = S[clampi(x + 1, 0, cols - 1)];
            float v = s0 * k0 + (sm1 + sp1) * k1;
            if (n == 5) {
               const float sm2 = S[clampi(x - 2, 0, cols - 1)], sp2 = S[clampi(x + 2, 0, cols - 1)];
               v = v + (sm2 + sp2) * k2;
            }
            T[x] = v;
         }
      } else {
         // RowFilter: ascending sequential accumulation, t = k[0] * S[x - r]; t += k[j] * S[x - r + j] (j = 1 .. n-1), columns clamped
         // (BORDER_REPLICATE).  Loop order only: the row is padded once (pad[i] = S[clamp(i - r)]) and the taps run in the OUTER loop
         // over a chunk of pixels, so that the compiler can vectorise across pixels; every pixel's own chain of operations - the
         // only thing the result depends on - is the one above, in the same order.
         pad.resize((size_t)cols + 2 * r);
         for (int i = 0; i < cols + 2 * r; i++) pad[i] = S[clampi(i - r, 0, cols - 1)];
         const int CH = 1024;
         for (int x0 = 0; x0 < cols; x0 += CH) {
            const int m = std::min(CH, cols - x0);
            const float *P0 = pad.data() + x0;
            float *Tx = T + x0;
            const float k0 = k[0];
            for (int x = 0; x < m; x++) Tx[x] = k0 * P0[x];
            for (int j = 1; j < n; j++) {
               const float kj = k[j];
               const float *Pj = P0 + j;
               for (int x = 0; x < m; x++) Tx[x] += kj * Pj[x];
            }
         }
      }
   }
   // column pass: SymmColumnFilter  d = k[r] * T[y]; d += k[r + j] * (T[y + j] + T[y - j]) (j = 1 .. r), rows clamped.  Loop order only: tiles of
   // columns, so that the 2 r + 1 rows an output row reads stay in the cache for the huge windows of normalizeAffine (K up to ~1600);
   // per output element the operations and their order are unchanged.
   const int CW = 256;
   for (int x0 = 0; x0 < cols; x0 += CW) {
      const int m = std::min(CW, cols - x0);
      for (int y = 0; y < rows; y++) {
         float *D = res.row(y) + x0;
         const float *T0 = tmp.row(y) + x0;
         const float kr = k[r];
         for (int x = 0; x < m; x++) D[x] = kr * T0[x];
         for (int j = 1; j <= r; j++) {
            const float *Tp = tmp.row(clampi(y + j, 0, rows - 1)) + x0;
            const float *Tm = tmp.row(clampi(y - j, 0, rows - 1)) + x0;
            const float kj = k[r + j];
            for (int x = 0; x < m; x++) D[x] += kj * (Tp[x] + Tm[x]);
         }
      }
   }
   out = res;
}

// helpers.cpp:297-329 doubleImage.  The reference indexes the source with its byte stride (`in[input.step]`, reading four
// rows below) and never writes the last row and column of the result; it is dead code at the defaults
// (upscaleInputImage = 0) and cannot be restated literally without undefined behaviour.  This is the evident intent,
// the same float expressions with the row below / column to the right clamped at the image edge (shared definition of
// the oracle and the product; SURVEY.md 8f rank 4 "fix stride bug").
void doubleImage(const Plane &in, Plane &out)
{
   Plane n(in.rows * 2, in.cols * 2);
   for (int y = 0; y < n.rows; y++)
      for (int x = 0; x < n.cols; x++) {
         const int r = y >> 1, c = x >> 1, r1 = r + 1 < in.rows ? r + 1 : in.rows - 1, c1 = c + 1 < in.cols ? c + 1 : in.cols - 1;
         const float v00 = in.at(r, c), v01 = in.at(r, c1), v10 = in.at(r1, c), v11 = in.at(r1, c1);
         float v;
         if ((y & 1) == 0) v = (x & 1) == 0 ? v00 : 0.5f * (v00 + v01);
         else v = (x & 1) == 0 ? 0.5f * (v00 + v10) : 0.25f * (v00 + v01 + v10 + v11);
         n.at(y, x) = v;
      }
   out = n;
}

// helpers.cpp:331-339
void halfImage(const Plane &in, Plane &out)
{
   Plane n(in.rows / 2, in.cols / 2);
   for (int r = 0; r < n.rows; r++)
      for (int c = 0; c < n.cols; c++) n.at(r, c) = in.at(2 * r, 2 * c);
   out = n;
}

// pyramid.cpp:63-114 ; the 1-pixel frame is never read by the reference (left
// uninitialised there), it is written as 0 here.
void hessianResponse(const Plane &in, float norm, Plane &out)
{
   Plane o(in.rows, in.cols);
   const float norm2 = norm * norm;
   for (int r = 1; r < in.rows - 1; r++) {
      const float *pm = in.row(r - 1), *p0 = in.row(r), *pp = in.row(r + 1);
      float *q = o.row(r);
      for (int c = 1; c < in.cols - 1; c++) {
         const float v11 = pm[c - 1], v12 = pm[c], v13 = pm[c + 1];
         const float v21 = p0[c - 1], v22 = p0[c], v23 = p0[c + 1];
         const float v31 = pp[c - 1], v32 = pp[c], v33 = pp[c + 1];
         const float Lxx = (v21 - 2 * v22 + v23);
         const float Lyy = (v12 - 2 * v22 + v32);
         const float Lxy = (v13 - v11 + v31 - v33) / 4.0f;
         q[c] = (Lxx * Lyy - Lxy * Lxy) * norm2;
      }
   }
   out = o;
}

// helpers.cpp:46-88 (the file-local swap<T>(T*,T*) swaps VALUES)
inline void swapv(float &a, float &b) { const float t = a; a = b; b = t; }
void solveLinear3x3(float *A, float *b)
{
   int i = 0;
   float vp = fabsf(A[0]);
   const float tmp = fabsf(A[3]);
   if (tmp > vp) { i = 1; vp = tmp; }
   if (fabsf(A[6]) > vp) { i = 2; }
   if (i != 0) {
      swapv(A[3 * i], A[0]); swapv(A[3 * i + 1], A[1]); swapv(A[3 * i + 2], A[2]); swapv(b[i], b[0]);
   }
   vp = A[3] / A[0]; A[4] -= vp * A[1]; A[5] -= vp * A[2]; b[1] -= vp * b[0];
   vp = A[6] / A[0]; A[7] -= vp * A[1]; A[8] -= vp * A[2]; b[2] -= vp * b[0];
   if (fabsf(A[4]) < fabsf(A[7])) { swapv(A[7], A[4]); swapv(A[8], A[5]); swapv(b[2], b[1]); }
   vp = A[7] / A[4];
   A[8] -= vp * A[5];
   b[2] -= vp * b[1];
   b[2] = (b[2]) / A[8];
   b[1] = (b[1] - A[5] * b[2]) / A[4];
   b[0] = (b[0] - A[2] * b[2] - A[1] * b[1]) / A[0];
}

// helpers.cpp:90-97
void rectifyUpIsUp(float &a11, float &a12, float &a21, float &a22)
{
   const double a = a11, b = a12, c = a21, d = a22;
   const double det = sqrt(fabs(a * d - b * c));
   const double b2a2 = sqrt(b * b + a * a);
   a11 = (float)(b2a2 / det);
   a12 = 0;
   a21 = (float)((d * b + c * a) / (b2a2 * det));
   a22 = (float)(det / b2a2);
}

// helpers.cpp:104-129
void computeGaussMask(int size, float *mask)
{
   const int half = size >> 1;
   const float scale = float(half) / 3.0f;
   const float scale2 = -2.0f * scale * scale;
   std::vector<float> tmp(half + 1);
   for (int i = 0; i <= half; i++) tmp[i] = expf(float(i * i) / scale2);
   const int endSize = int(ceilf(scale * 5.0f) - half);
   for (int i = 1; i < endSize; i++) tmp[half - i] += expf(float((i + half) * (i + half)) / scale2);
   for (int i = 0; i <= half; i++)
      for (int j = 0; j <= half; j++) {
         const float v = tmp[i] * tmp[j];
         mask[(i + half) * size + (-j + half)] = v;
         mask[(-i + half) * size + (j + half)] = v;
         mask[(i + half) * size + (j + half)] = v;
         mask[(-i + half) * size + (-j + half)] = v;
      }
}

// helpers.cpp:131-147
void computeCircularGaussMask(int size, float *mask)
{
   const int half = size >> 1;
   const float r2 = float(half * half);
   const float sigma2 = 0.9f * r2;
   for (int i = 0; i < size; i++)
      for (int j = 0; j < size; j++) {
         const float disq = float((i - half) * (i - half) + (j - half) * (j - half));
         mask[i * size + j] = (disq < r2) ? expf(-disq / sigma2) : 0;
      }
}

// helpers.cpp:149-175 (double arithmetic inside)
void invSqrt(float &a, float &b, float &c, float &l1, float &l2)
{
   double t, r;
   if (b != 0) {
      r = double(c - a) / (2 * b);
      if (r >= 0) t = 1.0 / (r + sqrt(1 + r * r));
      else t = -1.0 / (-r + sqrt(1 + r * r));
      r = 1.0 / sqrt(1 + t * t);
      t = t * r;
   } else {
      r = 1;
      t = 0;
   }
   double x = 1.0 / sqrt(r * r * a - 2 * r * t * b + t * t * c);
   double z = 1.0 / sqrt(t * t * a + 2 * r * t * b + r * r * c);
   const double d = sqrt(x * z);
   x /= d;
   z /= d;
   if (x < z) { l1 = float(z); l2 = float(x); } else { l1 = float(x); l2 = float(z); }
   a = float(r * r * x + t * t * z);
   b = float(-r * t * x + t * r * z);
   c = float(t * t * x + r * r * z);
}

// helpers.cpp:177-188
bool getEigenvalues(float a, float b, float c, float d, float &l1, float &l2)
{
   const float trace = a + d;
   const float delta1 = (trace * trace - 4 * (a * d - b * c));
   if (delta1 < 0) return false;
   const float delta = sqrtf(delta1);
   l1 = (trace + delta) / 2.0f;
   l2 = (trace - delta) / 2.0f;
   return true;
}

// (int)floor(x) as the reference's x86 build evaluates it: cvttss2si returns INT_MIN for
// NaN and out-of-range values (C++ leaves it undefined).  Made explicit here so the
// oracle does not depend on the host ISA.
inline int floorToInt(float w)
{
   const float f = floorf(w);
   if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
   return (int)f;
}

// helpers.cpp:209-244 ; res is (2*halfH+1) x (2*halfW+1)
bool interpolate(const Plane &im, float ofsx, float ofsy, float a11, float a12, float a21, float a22,
                 float *res, int resRows, int resCols)
{
   bool ret = false;
   const int width = im.cols - 1, height = im.rows - 1;
   const int halfWidth = resCols >> 1, halfHeight = resRows >> 1;
   float *out = res;
   for (int j = -halfHeight; j <= halfHeight; ++j) {
      const float rx = ofsx + j * a12;
      const float ry = ofsy + j * a22;
      for (int i = -halfWidth; i <= halfWidth; ++i) {
         float wx = rx + i * a11;
         float wy = ry + i * a21;
         const int x = floorToInt(wx);
         const int y = floorToInt(wy);
         if (x >= 0 && y >= 0 && x < width && y < height) {
            wx -= x;
            wy -= y;
            *out++ = (1.0f - wy) * ((1.0f - wx) * im.at(y, x) + wx * im.at(y, x + 1)) +
                     (wy) * ((1.0f - wx) * im.at(y + 1, x) + wx * im.at(y + 1, x + 1));
         } else {
            *out++ = 0;
            ret = true;
         }
      }
   }
   return ret;
}

// helpers.cpp:191-207
bool interpolateCheckBorders(int imRows, int imCols, float ofsx, float ofsy, float a11, float a12,
                             float a21, float a22, int resRows, int resCols)
{
   const int width = imCols - 2, height = imRows - 2;
   const int halfWidth = resCols >> 1, halfHeight = resRows >> 1;
   const float x[4] = {(float)-halfWidth, (float)-halfWidth, (float)halfWidth, (float)halfWidth};
   const float y[4] = {(float)-halfHeight, (float)halfHeight, (float)-halfHeight, (float)halfHeight};
   for (int i = 0; i < 4; i++) {
      const float imx = ofsx + x[i] * a11 + y[i] * a12;
      const float imy = ofsy + x[i] * a21 + y[i] * a22;
      if (floorf(imx) <= 0 || floorf(imy) <= 0 || ceilf(imx) >= width || ceilf(imy) >= height) return true;
   }
   return false;
}

// affine.cpp:14-33 and siftdesc.cpp:123-134 (same stencil: central difference WITHOUT 1/2,
// one-sided at the frame)
inline void gradAt(const float *img, int size, int r, int c, float &gx, float &gy)
{
   if (c == 0) gx = img[r * size + c + 1] - img[r * size + c];
   else if (c == size - 1) gx = img[r * size + c] - img[r * size + c - 1];
   else gx = img[r * size + c + 1] - img[r * size + c - 1];
   if (r == 0) gy = img[(r + 1) * size + c] - img[r * size + c];
   else if (r == size - 1) gy = img[r * size + c] - img[(r - 1) * size + c];
   else gy = img[(r + 1) * size + c] - img[(r - 1) * size + c];
}

// helpers.cpp:246-281
void photometricallyNormalize(float *image, const float *mask, int size)
{
   float sum = 0, gsum = 0;
   const int n = size * size;
   for (int i = 0; i < n; i++)
      if (mask[i] > 0) { sum += image[i]; gsum++; }
   sum = sum / gsum;
   float var = 0;
   for (int i = 0; i < n; i++)
      if (mask[i] > 0) var += (sum - image[i]) * (sum - image[i]);
   var = sqrtf(var / gsum);
   if (var < 0.0001) return;
   const float fac = 50.0f / var;
   for (int i = 0; i < n; i++) {
      float v = 128 + fac * (image[i] - sum);
      if (v > 255) v = 255;
      if (v < 0) v = 0;
      image[i] = v;
   }
}

// ----------------------------------------------------------------------------------------
// Stage objects
// ----------------------------------------------------------------------------------------
struct HessKp {   // arguments of onHessianKeypointDetected (pyramid.h:46) + provenance
   float x, y, s, pd;
   int type;
   float response;
   int octave, level;   // level = index i-2 of prevBlur inside the octave (0,1,2)
   int r0, c0;          // initial extremum pixel
};

struct AffRes {   // result of findAffineShape (affine.cpp:35-100)
   int converged;
   float a11, a12, a21, a22;
   int iters;
};

struct Keypoint {   // hesaff.cpp:41-48
   float x, y, s;
   float a11, a12, a21, a22;
   float response;
   int type;
   unsigned char desc[128];
};

struct Oracle {
   Params par;
   // derived constants, pyramid.h:59-64
   float edgeScoreThreshold, finalThreshold, positiveThreshold, negativeThreshold;
   std::vector<float> smmMask;        // 19x19, affine.h:71
   std::vector<float> siftMask;       // 41x41, siftdesc.h:47
   std::vector<int> bin0, bin1;       // siftdesc.cpp:18-49
   std::vector<float> w0, w1;

   // outputs
   std::vector<HessKp> hess;
   std::vector<AffRes> aff;
   std::vector<Keypoint> keys;
   std::vector<int> keySrc;                       // index into hess for each key
   std::vector<std::vector<Plane>> blurs, resps;  // [octave][0..4], kept when keepPlanes
   bool keepPlanes = false;
   bool detectOnly = false;                       // stop after Hessian keypoints
   long nCandidates = 0;

   Oracle() { configure(); }

   // derived constants and tables from `par` (the reference builds them in its constructors:
   // pyramid.h:59-64, affine.h:63-75, siftdesc.h:40-49); called again by ho_set_params
   void configure()
   {
      edgeScoreThreshold = (par.edgeEigenValueRatio + 1.0f) * (par.edgeEigenValueRatio + 1.0f) / par.edgeEigenValueRatio;
      finalThreshold = par.threshold * par.threshold;
      positiveThreshold = (float)(0.8 * finalThreshold);
      negativeThreshold = -positiveThreshold;
      smmMask.resize(par.smmWindowSize * par.smmWindowSize);
      computeGaussMask(par.smmWindowSize, smmMask.data());
      siftMask.resize(par.patchSize * par.patchSize);
      computeCircularGaussMask(par.patchSize, siftMask.data());
      precomputeBinsAndWeights();
   }

   // siftdesc.cpp:18-49
   void precomputeBinsAndWeights()
   {
      const int ps = par.patchSize, half = ps >> 1;
      const float step = float(par.spatialBins + 1) / (2 * half);
      bin0.resize(ps); bin1.resize(ps); w0.resize(ps); w1.resize(ps);
      for (int i = 0; i < ps; i++) {
         const float x = step * i;
         const int xi = (int)(x);
         bin0[i] = xi - 1;
         bin1[i] = xi;
         w1[i] = x - xi;
         w0[i] = 1.0f - w1[i];
         if (bin0[i] < 0) { bin0[i] = 0; w0[i] = 0; }
         if (bin0[i] >= par.spatialBins) { bin0[i] = par.spatialBins - 1; w0[i] = 0; }
         if (bin1[i] < 0) { bin1[i] = 0; w1[i] = 0; }
         if (bin1[i] >= par.spatialBins) { bin1[i] = par.spatialBins - 1; w1[i] = 0; }
         bin0[i] *= par.orientationBins;
         bin1[i] *= par.orientationBins;
      }
   }

   // siftdesc.cpp:115-140 + :51-113 ; patch is modified in place like the reference's
   void computeSiftDescriptor(float *patch, float *vec /*128*/)
   {
      const int ps = par.patchSize;
      photometricallyNormalize(patch, siftMask.data(), ps);
      std::vector<float> grad(ps * ps), ori(ps * ps);
      for (int r = 0; r < ps; ++r)
         for (int c = 0; c < ps; ++c) {
            float gx, gy;
            gradAt(patch, ps, r, c, gx, gy);
            grad[r * ps + c] = sqrtf(gx * gx + gy * gy);
            ori[r * ps + c] = atan2f(gy, gx);
         }
      const int nv = par.spatialBins * par.spatialBins * par.orientationBins;
      for (int i = 0; i < nv; i++) vec[i] = 0;
      // samplePatch, siftdesc.cpp:51-81
      for (int r = 0; r < ps; ++r) {
         const int br0 = par.spatialBins * bin0[r]; const float wr0 = w0[r];
         const int br1 = par.spatialBins * bin1[r]; const float wr1 = w1[r];
         for (int c = 0; c < ps; ++c) {
            float val = siftMask[r * ps + c] * grad[r * ps + c];
            const int bc0 = bin0[c]; const float wc0 = w0[c] * val;
            const int bc1 = bin1[c]; const float wc1 = w1[c] * val;
            const float o = float(par.orientationBins) * (ori[r * ps + c] + 2 * M_PI) / (2 * M_PI);
            int bo0 = (int)o;
            const float wo1 = o - bo0;
            bo0 %= par.orientationBins;
            const int bo1 = (bo0 + 1) % par.orientationBins;
            const float wo0 = 1.0f - wo1;
            val = wr0 * wc0; if (val > 0) { vec[br0 + bc0 + bo0] += val * wo0; vec[br0 + bc0 + bo1] += val * wo1; }
            val = wr0 * wc1; if (val > 0) { vec[br0 + bc1 + bo0] += val * wo0; vec[br0 + bc1 + bo1] += val * wo1; }
            val = wr1 * wc0; if (val > 0) { vec[br1 + bc0 + bo0] += val * wo0; vec[br1 + bc0 + bo1] += val * wo1; }
            val = wr1 * wc1; if (val > 0) { vec[br1 + bc1 + bo0] += val * wo0; vec[br1 + bc1 + bo1] += val * wo1; }
         }
      }
      normalizeVec(vec, nv);
      bool changed = false;
      for (int i = 0; i < nv; i++)
         if (vec[i] > par.maxBinValue) { vec[i] = par.maxBinValue; changed = true; }
      if (changed) normalizeVec(vec, nv);
      for (int i = 0; i < nv; i++) {
         // (int) of NaN: x86 gives INT_MIN, min(...,255) keeps it, the later
         // (unsigned char) cast (hesaff.cpp:91) yields 0 -- stated explicitly.
         const float q = 512.0f * vec[i];
         int b = (q == q) ? (int)q : 0;
         if (b > 255) b = 255;
         vec[i] = float(b);
      }
   }
   // siftdesc.cpp:83-96
   static void normalizeVec(float *vec, int n)
   {
      float vectlen = 0.0f;
      for (int i = 0; i < n; i++) { const float val = vec[i]; vectlen += val * val; }
      vectlen = sqrtf(vectlen);
      const float fac = float(1.0f / vectlen);
      for (int i = 0; i < n; i++) vec[i] *= fac;
   }

   // affine.cpp:35-100
   AffRes findAffineShape(const Plane &blur, float x, float y, float s, float pixelDistance)
   {
      AffRes res = {0, 0, 0, 0, 0, 0};
      float eigen_ratio_act = 0.0f, eigen_ratio_bef = 0.0f;
      float u11 = 1.0f, u12 = 0.0f, u21 = 0.0f, u22 = 1.0f, l1 = 1.0f, l2 = 1.0f;
      const float lx = x / pixelDistance, ly = y / pixelDistance;
      const float ratio = s / (par.affInitialSigma * pixelDistance);
      const int W = par.smmWindowSize, maskPixels = W * W;
      std::vector<float> img(maskPixels);
      for (int l = 0; l < par.maxIterations; l++) {
         interpolate(blur, lx, ly, u11 * ratio, u12 * ratio, u21 * ratio, u22 * ratio, img.data(), W, W);
         float a = 0, b = 0, c = 0;
         for (int i = 0; i < maskPixels; ++i) {
            float gxx, gyy;
            gradAt(img.data(), W, i / W, i % W, gxx, gyy);
            const float v = smmMask[i];
            const float gxy = gxx * gyy;
            a += gxx * gxx * v;
            b += gxy * v;
            c += gyy * gyy * v;
         }
         a /= maskPixels; b /= maskPixels; c /= maskPixels;
         invSqrt(a, b, c, l1, l2);
         eigen_ratio_bef = eigen_ratio_act;
         eigen_ratio_act = 1 - l2 / l1;
         const float u11t = u11, u12t = u12;
         u11 = a * u11t + b * u21; u12 = a * u12t + b * u22;
         u21 = b * u11t + c * u21; u22 = b * u12t + c * u22;
         if (!getEigenvalues(u11, u12, u21, u22, l1, l2)) break;
         if ((l1 / l2 > 6) || (l2 / l1 > 6)) break;
         if (eigen_ratio_act < par.convergenceThreshold && eigen_ratio_bef < par.convergenceThreshold) {
            res.converged = 1; res.a11 = u11; res.a12 = u12; res.a21 = u21; res.a22 = u22; res.iters = l;
            return res;
         }
      }
      return res;
   }

   // affine.cpp:102-144 ; returns true when the keypoint is REJECTED (like the reference)
   bool normalizeAffine(const Plane &img, float x, float y, float s, float a11, float a12, float a21,
                        float a22, float *patch)
   {
      const int ps = par.patchSize;
      const float mrScale = ceilf(s * par.mrSize);
      int patchImageSize = 2 * int(mrScale) + 1;
      const float imageToPatchScale = float(patchImageSize) / float(ps);
      if (interpolateCheckBorders(img.rows, img.cols, x, y, a11 * imageToPatchScale, a12 * imageToPatchScale,
                                  a21 * imageToPatchScale, a22 * imageToPatchScale, ps, ps))
         return true;
      if (imageToPatchScale > 0.4) {
         patchImageSize += 2;
         Plane smoothed(patchImageSize, patchImageSize);
         if (!interpolate(img, x, y, a11, a12, a21, a22, smoothed.d.data(), patchImageSize, patchImageSize)) {
            gaussianBlur(smoothed, 1.5f * imageToPatchScale, smoothed);
            interpolate(smoothed, (float)(patchImageSize >> 1), (float)(patchImageSize >> 1),
                        imageToPatchScale, 0, 0, imageToPatchScale, patch, ps, ps);
         } else
            return true;
      } else {
         a11 *= imageToPatchScale; a12 *= imageToPatchScale;
         a21 *= imageToPatchScale; a22 *= imageToPatchScale;
         interpolate(img, x, y, a11, a12, a21, a22, patch, ps, ps);
      }
      return false;
   }

   // hesaff.cpp:72-105 : rectify, normalise, describe, store
   void onAffineShapeFound(const Plane &image, const HessKp &h, const AffRes &a, int hidx)
   {
      float a11 = a.a11, a12 = a.a12, a21 = a.a21, a22 = a.a22;
      rectifyUpIsUp(a11, a12, a21, a22);
      std::vector<float> patch(par.patchSize * par.patchSize);
      if (!normalizeAffine(image, h.x, h.y, h.s, a11, a12, a21, a22, patch.data())) {
         float vec[128];
         computeSiftDescriptor(patch.data(), vec);
         Keypoint k;
         k.x = h.x; k.y = h.y; k.s = h.s; k.a11 = a11; k.a12 = a12; k.a21 = a21; k.a22 = a22;
         k.response = h.response; k.type = h.type;
         for (int i = 0; i < 128; i++) k.desc[i] = (unsigned char)vec[i];
         keys.push_back(k);
         keySrc.push_back(hidx);
      }
   }

   // pyramid.cpp:122-204
   void localizeKeypoint(int r, int c, float curScale, float pixelDistance, const Plane &low, const Plane &cur,
                         const Plane &high, const Plane &blur, const Plane &prevBlur, Plane &octaveMap,
                         const Plane &image, int octave, int level)
   {
      const int cols = cur.cols, rows = cur.rows;
      const int r0 = r, c0 = c;
      float b[3] = {0, 0, 0};
      float val = 0;
      int nr = r, nc = c;
      for (int iter = 0; iter < 5; iter++) {
         r = nr; c = nc;
         const float dxx = cur.at(r, c - 1) - 2.0f * cur.at(r, c) + cur.at(r, c + 1);
         const float dyy = cur.at(r - 1, c) - 2.0f * cur.at(r, c) + cur.at(r + 1, c);
         const float dss = low.at(r, c) - 2.0f * cur.at(r, c) + high.at(r, c);
         const float dxy = 0.25f * (cur.at(r + 1, c + 1) - cur.at(r + 1, c - 1) - cur.at(r - 1, c + 1) + cur.at(r - 1, c - 1));
         if (0 == iter) {
            const float edgeScore = (dxx + dyy) * (dxx + dyy) / (dxx * dyy - dxy * dxy);
            if (edgeScore >= edgeScoreThreshold || edgeScore < 0) return;
         }
         const float dxs = 0.25f * (high.at(r, c + 1) - high.at(r, c - 1) - low.at(r, c + 1) + low.at(r, c - 1));
         const float dys = 0.25f * (high.at(r + 1, c) - high.at(r - 1, c) - low.at(r + 1, c) + low.at(r - 1, c));
         float A[9] = {dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss};
         const float dx = 0.5f * (cur.at(r, c + 1) - cur.at(r, c - 1));
         const float dy = 0.5f * (cur.at(r + 1, c) - cur.at(r - 1, c));
         const float ds = 0.5f * (high.at(r, c) - low.at(r, c));
         b[0] = -dx; b[1] = -dy; b[2] = -ds;
         solveLinear3x3(A, b);
         if (std::isnan(b[0]) || std::isnan(b[1]) || std::isnan(b[2])) return;
         val = cur.at(r, c) + 0.5f * (dx * b[0] + dy * b[1] + ds * b[2]);
         if (b[0] > 0.6) { if (c < cols - 3) nc++; else return; }
         if (b[1] > 0.6) { if (r < rows - 3) nr++; else return; }
         if (b[0] < -0.6) { if (c > 3) nc--; else return; }
         if (b[1] < -0.6) { if (r > 3) nr--; else return; }
         if (nr == r && nc == c) break;
      }
      if (fabsf(b[0]) > 1.5 || fabsf(b[1]) > 1.5 || fabsf(b[2]) > 1.5 || fabsf(val) < finalThreshold ||
          octaveMap.at(r, c) > 0)
         return;
      octaveMap.at(r, c) = 1;
      const float scale = curScale * powf(2.0f, b[2] / par.numberOfScales);
      // pyramid.cpp:24-37
      int type;
      if (val < 0) type = 2;
      else {
         const float *p = blur.row(r) + c;
         const float Lxx = (p[-1] - 2 * p[0] + p[1]);
         type = (Lxx < 0) ? 0 : 1;
      }
      HessKp h;
      h.x = pixelDistance * (c + b[0]); h.y = pixelDistance * (r + b[1]); h.s = pixelDistance * scale;
      h.pd = pixelDistance; h.type = type; h.response = val; h.octave = octave; h.level = level; h.r0 = r0; h.c0 = c0;
      hess.push_back(h);
      if (detectOnly) { aff.push_back(AffRes{0, 0, 0, 0, 0, 0}); return; }
      // hesaff.cpp:66-70 -> affine.cpp:35
      const AffRes a = findAffineShape(prevBlur, h.x, h.y, h.s, h.pd);
      aff.push_back(a);
      if (a.converged) onAffineShapeFound(image, h, a, (int)hess.size() - 1);
   }

   // pyramid.cpp:39-61
   static bool isMax(float val, const Plane &pix, int row, int col)
   {
      for (int r = row - 1; r <= row + 1; r++)
         for (int c = col - 1; c <= col + 1; c++)
            if (pix.at(r, c) > val) return false;
      return true;
   }
   static bool isMin(float val, const Plane &pix, int row, int col)
   {
      for (int r = row - 1; r <= row + 1; r++)
         for (int c = col - 1; c <= col + 1; c++)
            if (pix.at(r, c) < val) return false;
      return true;
   }

   // pyramid.cpp:224-259 (+ :206-222 scan)
   void detectOctave(const Plane &firstLevel, float pixelDistance, Plane &nextFirst, const Plane &image, int octave)
   {
      Plane octaveMap(firstLevel.rows, firstLevel.cols);
      const float sigmaStep = powf(2.0f, 1.0f / (float)par.numberOfScales);
      float curSigma = par.initialSigma;
      const int nl = par.numberOfScales + 2;
      std::vector<Plane> L(nl), R(nl);
      L[0] = firstLevel;
      hessianResponse(L[0], curSigma * curSigma, R[0]);
      for (int i = 1; i < nl; i++) {
         float sigma = curSigma * sqrtf(sigmaStep * sigmaStep - 1.0f);
         gaussianBlur(L[i - 1], sigma, L[i]);
         sigma = curSigma * sigmaStep;
         hessianResponse(L[i], sigma * sigma, R[i]);
         if (i >= 2) {
            const Plane &low = R[i - 2], &cur = R[i - 1], &high = R[i];
            const int rows = cur.rows, cols = cur.cols;
            for (int r = par.border; r < rows - par.border; r++)
               for (int c = par.border; c < cols - par.border; c++) {
                  const float val = cur.at(r, c);
                  if ((val > positiveThreshold && (isMax(val, cur, r, c) && isMax(val, low, r, c) && isMax(val, high, r, c))) ||
                      (val < negativeThreshold && (isMin(val, cur, r, c) && isMin(val, low, r, c) && isMin(val, high, r, c)))) {
                     nCandidates++;
                     localizeKeypoint(r, c, curSigma, pixelDistance, low, cur, high, L[i - 1], L[i - 2], octaveMap,
                                      image, octave, i - 2);
                  }
               }
         }
         if (i == par.numberOfScales) halfImage(L[i], nextFirst);
         curSigma *= sigmaStep;
      }
      if (keepPlanes) { blurs.push_back(L); resps.push_back(R); }
   }

   // pyramid.cpp:261-292 (upscaleInputImage = 0 path)
   void detect(const Plane &image)
   {
      hess.clear(); aff.clear(); keys.clear(); keySrc.clear(); blurs.clear(); resps.clear(); nCandidates = 0;
      float curSigma = 0.5f;
      float pixelDistance = 1.0f;
      Plane firstLevel = image;
      if (par.upscaleInputImage > 0) {   // pyramid.cpp:267-271
         doubleImage(image, firstLevel);
         pixelDistance *= 0.5f;
         curSigma *= 2.0f;
      }
      if (par.initialSigma > curSigma) {
         const float sigma = sqrtf(par.initialSigma * par.initialSigma - curSigma * curSigma);
         gaussianBlur(firstLevel, sigma, firstLevel);
      }
      const int minSize = 2 * par.border + 2;
      int octave = 0;
      while (firstLevel.rows > minSize && firstLevel.cols > minSize) {
         Plane next;
         detectOctave(firstLevel, pixelDistance, next, image, octave);
         pixelDistance *= 2.0;
         firstLevel = next;
         octave++;
      }
   }
};

// hesaff.cpp:107-130 (exportKeypoints), the statements in the reference's order:
//    float sc = mrSize * k.s;  SVD svd(A, FULL_UV);  d[i] = 1.0f/(d[i]*d[i]*sc*sc);
//    A = svd.u * Mat::diag(svd.w) * svd.u.t();        print A(0,0), A(0,1), A(1,1)
// cv::SVD is third-party (OpenCV, a float Jacobi solver): it is restated as the closed-form
// symmetric eigen-decomposition of A A^T in double, with u and w then stored as float like the
// members of cv::SVD; the reference's own float expression for d and the matrix product
// (cv::gemm accumulates CV_32F products in double) follow as written.  With this form the
// whole .hesaff.sift text of the SURVEY App. C inputs has the md5 the survey recorded from the
// compiled reference (scripts/check_survey_probe.py).
void ellipseOf(const Keypoint &k, float mrSize, float &ea, float &eb, float &ec)
{
   const float sc = mrSize * k.s;
   const double a11 = k.a11, a12 = k.a12, a21 = k.a21, a22 = k.a22;
   const double m00 = a11 * a11 + a12 * a12, m01 = a11 * a21 + a12 * a22, m11 = a21 * a21 + a22 * a22;
   const double tr = m00 + m11, df = m00 - m11;
   const double disc = std::sqrt(df * df + 4.0 * m01 * m01);
   const double l1 = (tr + disc) / 2.0, l2 = (tr - disc) / 2.0;
   // unit eigenvector of l1: (l1 - m11, m01) or (m01, l1 - m00), whichever is longer
   double vx = l1 - m11, vy = m01;
   const double wx = m01, wy = l1 - m00;
   if (wx * wx + wy * wy > vx * vx + vy * vy) { vx = wx; vy = wy; }
   const double n = std::sqrt(vx * vx + vy * vy);
   float cu = 1.0f, su = 0.0f;
   if (n > 0) { cu = (float)(vx / n); su = (float)(vy / n); }
   float w0 = (float)std::sqrt(l1), w1 = (float)std::sqrt(l2);
   w0 = 1.0f / (w0 * w0 * sc * sc);   // hesaff.cpp:120
   w1 = 1.0f / (w1 * w1 * sc * sc);   // hesaff.cpp:121
   // u = [cu -su; su cu];  u * diag(w) in double -> float, then * u^T in double -> float
   const float p00 = (float)((double)cu * w0), p01 = (float)(-(double)su * w1);
   const float p10 = (float)((double)su * w0), p11 = (float)((double)cu * w1);
   ea = (float)((double)p00 * cu + (double)p01 * -su);
   eb = (float)((double)p00 * su + (double)p01 * cu);
   ec = (float)((double)p10 * su + (double)p11 * cu);
}

} // namespace

// ========================================================================================
// C interface for ctypes (tests/, bench cpu_baseline, smoke) -- test infrastructure only
// ========================================================================================
extern "C" {

int ho_gauss_ksize(float sigma) { return gaussKsize(sigma); }
void ho_gauss_kernel(int n, float sigma, float *k) { gaussKernel(n, sigma, k); }

void ho_gaussian_blur(const float *in, int rows, int cols, float sigma, float *out)
{
   Plane p(rows, cols), o;
   memcpy(p.d.data(), in, sizeof(float) * rows * cols);
   gaussianBlur(p, sigma, o);
   memcpy(out, o.d.data(), sizeof(float) * rows * cols);
}
void ho_hessian_response(const float *in, int rows, int cols, float norm, float *out)
{
   Plane p(rows, cols), o;
   memcpy(p.d.data(), in, sizeof(float) * rows * cols);
   hessianResponse(p, norm, o);
   memcpy(out, o.d.data(), sizeof(float) * rows * cols);
}
void ho_double_image(const float *in, int rows, int cols, float *out)
{
   Plane p(rows, cols), o;
   memcpy(p.d.data(), in, sizeof(float) * rows * cols);
   doubleImage(p, o);
   memcpy(out, o.d.data(), sizeof(float) * o.rows * o.cols);
}
void ho_half_image(const float *in, int rows, int cols, float *out)
{
   Plane p(rows, cols), o;
   memcpy(p.d.data(), in, sizeof(float) * rows * cols);
   halfImage(p, o);
   memcpy(out, o.d.data(), sizeof(float) * o.rows * o.cols);
}
// hesaff.cpp:138-148 ; channels = 1 replicates grey like cv::imread's BGR conversion
void ho_gray_from_u8(const unsigned char *in, int n, int channels, float *out)
{
   for (int i = 0; i < n; i++) {
      const unsigned char *p = in + (size_t)i * channels;
      const unsigned char c0 = p[0], c1 = channels == 3 ? p[1] : p[0], c2 = channels == 3 ? p[2] : p[0];
      out[i] = (float(c0) + c1 + c2) / 3.0f;
   }
}
int ho_interpolate(const float *im, int rows, int cols, float ofsx, float ofsy, float a11, float a12, float a21,
                   float a22, float *res, int resRows, int resCols)
{
   Plane p(rows, cols);
   memcpy(p.d.data(), im, sizeof(float) * rows * cols);
   return interpolate(p, ofsx, ofsy, a11, a12, a21, a22, res, resRows, resCols) ? 1 : 0;
}
void ho_solve_linear3x3(float *A, float *b) { solveLinear3x3(A, b); }
void ho_inv_sqrt(float *abc, float *l) { invSqrt(abc[0], abc[1], abc[2], l[0], l[1]); }
int ho_get_eigenvalues(float a, float b, float c, float d, float *l)
{
   return getEigenvalues(a, b, c, d, l[0], l[1]) ? 1 : 0;
}
void ho_rectify(float *A) { rectifyUpIsUp(A[0], A[1], A[2], A[3]); }
void ho_gauss_mask(int size, float *mask) { computeGaussMask(size, mask); }
void ho_circ_gauss_mask(int size, float *mask) { computeCircularGaussMask(size, mask); }
void ho_sift_tables(int *b0, int *b1, float *w0, float *w1)
{
   Oracle o;
   for (int i = 0; i < o.par.patchSize; i++) { b0[i] = o.bin0[i]; b1[i] = o.bin1[i]; w0[i] = o.w0[i]; w1[i] = o.w1[i]; }
}
float ho_atan2f(float y, float x) { return atan2f(y, x); }
float ho_pow2f(float y) { return powf(2.0f, y); }

// stage-level entry points on caller-provided data
int ho_find_affine_shape(const float *blur, int rows, int cols, float x, float y, float s, float pd, float *A, int *iters)
{
   Oracle o;
   Plane p(rows, cols);
   memcpy(p.d.data(), blur, sizeof(float) * rows * cols);
   const AffRes a = o.findAffineShape(p, x, y, s, pd);
   A[0] = a.a11; A[1] = a.a12; A[2] = a.a21; A[3] = a.a22; *iters = a.iters;
   return a.converged;
}
// returns 1 if rejected (reference convention), patch = 41x41
int ho_normalize_affine(const float *img, int rows, int cols, float x, float y, float s, const float *A, float *patch)
{
   Oracle o;
   Plane p(rows, cols);
   memcpy(p.d.data(), img, sizeof(float) * rows * cols);
   return o.normalizeAffine(p, x, y, s, A[0], A[1], A[2], A[3], patch) ? 1 : 0;
}
void ho_sift(float *patch, float *vec)
{
   Oracle o;
   o.computeSiftDescriptor(patch, vec);
}

// ---- full pipeline with a handle ----
void *ho_create()
{
   // The reference warps every patch into ONE growing workspace (affine.cpp:120-124); this restatement allocates its
   // planes per call.  Keep freed blocks inside the heap instead of returning them to the kernel each time, so that the
   // timed baseline measures arithmetic and not page faults (a 3840x2160 image: system time 0.7 s instead of tens of seconds).
   static const int once = (mallopt(M_MMAP_THRESHOLD, 1 << 30), mallopt(M_TRIM_THRESHOLD, 1 << 30), 0);
   (void)once;
   return new Oracle();
}
// the stage functions above with the parameters of a handle (ho_set_params)
int ho_h_find_affine_shape(void *h, const float *blur, int rows, int cols, float x, float y, float s, float pd, float *A, int *iters)
{
   Plane p(rows, cols);
   memcpy(p.d.data(), blur, sizeof(float) * rows * cols);
   const AffRes a = ((Oracle *)h)->findAffineShape(p, x, y, s, pd);
   A[0] = a.a11; A[1] = a.a12; A[2] = a.a21; A[3] = a.a22; *iters = a.iters;
   return a.converged;
}
int ho_h_normalize_affine(void *h, const float *img, int rows, int cols, float x, float y, float s, const float *A, float *patch)
{
   Plane p(rows, cols);
   memcpy(p.d.data(), img, sizeof(float) * rows * cols);
   return ((Oracle *)h)->normalizeAffine(p, x, y, s, A[0], A[1], A[2], A[3], patch) ? 1 : 0;
}
void ho_h_sift(void *h, float *patch, float *vec) { ((Oracle *)h)->computeSiftDescriptor(patch, vec); }
void ho_destroy(void *h) { delete (Oracle *)h; }
void ho_set_keep_planes(void *h, int keep) { ((Oracle *)h)->keepPlanes = keep != 0; }
void ho_set_detect_only(void *h, int v) { ((Oracle *)h)->detectOnly = v != 0; }
// Non-default parameters, the fields hesaff_params (include/hesaff_amd.h) exposes:
// threshold pyramid.h:37 / hesaff.cpp:155, edgeEigenValueRatio pyramid.h:38, initialSigma pyramid.h:36,
// maxIterations affine.h:39 / hesaff.cpp:158, convergenceThreshold affine.h:41, mrSize affine.h:44 /
// hesaff.cpp:160, maxBinValue siftdesc.h:29, upscaleInputImage pyramid.h:34.
void ho_set_params(void *h, float threshold, float edgeEigenValueRatio, float initialSigma, int maxIterations,
                   float convergenceThreshold, float mrSize, float maxBinValue)
{
   Oracle *o = (Oracle *)h;
   o->par.threshold = threshold;
   o->par.edgeEigenValueRatio = edgeEigenValueRatio;
   o->par.initialSigma = initialSigma;
   o->par.maxIterations = maxIterations;
   o->par.convergenceThreshold = convergenceThreshold;
   o->par.mrSize = mrSize;
   o->par.maxBinValue = maxBinValue;
   o->configure();
}
void ho_set_upscale(void *h, int upscale) { ((Oracle *)h)->par.upscaleInputImage = upscale; }
void ho_detect(void *h, const float *gray, int rows, int cols)
{
   Plane p(rows, cols);
   memcpy(p.d.data(), gray, sizeof(float) * rows * cols);
   ((Oracle *)h)->detect(p);
}
int ho_num_hessian(void *h) { return (int)((Oracle *)h)->hess.size(); }
int ho_num_keys(void *h) { return (int)((Oracle *)h)->keys.size(); }
long ho_num_candidates(void *h) { return ((Oracle *)h)->nCandidates; }
int ho_num_octaves(void *h) { return (int)((Oracle *)h)->blurs.size(); }
// f[6] = x,y,s,pd,response,(unused) ; i[5] = type,octave,level,r0,c0
void ho_get_hessian(void *h, int idx, float *f, int *i)
{
   const HessKp &k = ((Oracle *)h)->hess[idx];
   f[0] = k.x; f[1] = k.y; f[2] = k.s; f[3] = k.pd; f[4] = k.response; f[5] = 0;
   i[0] = k.type; i[1] = k.octave; i[2] = k.level; i[3] = k.r0; i[4] = k.c0;
}
// f[4] = a11..a22 (un-rectified U); i[2] = converged, iters
void ho_get_affine(void *h, int idx, float *f, int *i)
{
   const AffRes &a = ((Oracle *)h)->aff[idx];
   f[0] = a.a11; f[1] = a.a12; f[2] = a.a21; f[3] = a.a22; i[0] = a.converged; i[1] = a.iters;
}
// f[8] = x,y,s,a11,a12,a21,a22,response ; i[2] = type, source hessian index ; desc[128]
void ho_get_key(void *h, int idx, float *f, int *i, unsigned char *desc)
{
   const Oracle *o = (Oracle *)h;
   const Keypoint &k = o->keys[idx];
   f[0] = k.x; f[1] = k.y; f[2] = k.s; f[3] = k.a11; f[4] = k.a12; f[5] = k.a21; f[6] = k.a22; f[7] = k.response;
   i[0] = k.type; i[1] = o->keySrc[idx];
   memcpy(desc, k.desc, 128);
}
// bulk getters (arrays sized by ho_num_keys): geom[n][8], type[n], desc[n][128]
void ho_get_keys(void *h, float *geom, int *type, unsigned char *desc)
{
   const Oracle *o = (Oracle *)h;
   for (size_t n = 0; n < o->keys.size(); n++) {
      const Keypoint &k = o->keys[n];
      float *f = geom + 8 * n;
      f[0] = k.x; f[1] = k.y; f[2] = k.s; f[3] = k.a11; f[4] = k.a12; f[5] = k.a21; f[6] = k.a22; f[7] = k.response;
      type[n] = k.type;
      memcpy(desc + 128 * n, k.desc, 128);
   }
}
void ho_plane_dims(void *h, int octave, int *rows, int *cols)
{
   const Oracle *o = (Oracle *)h;
   *rows = o->blurs[octave][0].rows; *cols = o->blurs[octave][0].cols;
}
// which = 0: blur L[level], 1: response R[level]
void ho_get_plane(void *h, int octave, int which, int level, float *out)
{
   const Oracle *o = (Oracle *)h;
   const Plane &p = which == 0 ? o->blurs[octave][level] : o->resps[octave][level];
   memcpy(out, p.d.data(), sizeof(float) * p.rows * p.cols);
}
// hesaff.cpp:107-130 text export; returns bytes written (or needed when buf == NULL)
long ho_export(void *h, char *buf, long cap)
{
   const Oracle *o = (Oracle *)h;
   std::string s;
   char line[96];   // five %g of a float: at most 5 x 13 characters + 4 blanks
   snprintf(line, sizeof line, "%d\n%zu\n", 128, o->keys.size());
   s += line;
   for (size_t n = 0; n < o->keys.size(); n++) {
      const Keypoint &k = o->keys[n];
      float ea, eb, ec;
      ellipseOf(k, o->par.mrSize, ea, eb, ec);
      snprintf(line, sizeof line, "%g %g %g %g %g", k.x, k.y, ea, eb, ec);
      s += line;
      for (int i = 0; i < 128; i++) { snprintf(line, sizeof line, " %d", (int)k.desc[i]); s += line; }
      s += "\n";
   }
   if (buf && (long)s.size() <= cap) memcpy(buf, s.data(), s.size());
   return (long)s.size();
}
void ho_ellipse(const float *geom8, float mrSize, float *abc)
{
   Keypoint k;
   k.x = geom8[0]; k.y = geom8[1]; k.s = geom8[2]; k.a11 = geom8[3]; k.a12 = geom8[4]; k.a21 = geom8[5]; k.a22 = geom8[6];
   ellipseOf(k, mrSize, abc[0], abc[1], abc[2]);
}

} // extern "C"
