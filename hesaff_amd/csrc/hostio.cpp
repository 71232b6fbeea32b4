// hostio.cpp -- host-only pieces of the drop-in boundary: PGM/PPM reader (replaces
// cv::imread at hesaff.cpp:137), ellipse closed form and the .hesaff.sift text writer
// (replaces exportKeypoints hesaff.cpp:107-130).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hesaff_amd.h"

namespace {

// skip whitespace and '#' comments of a PNM header
int pnm_next_int(FILE *f, int *out)
{
   int c = fgetc(f);
   for (;;) {
      while (c == ' ' || c == '\t' || c == '\n' || c == '\r') c = fgetc(f);
      if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
      break;
   }
   if (c < '0' || c > '9') return -1;
   long v = 0;
   while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); if (v > 100000000) return -1; c = fgetc(f); }
   *out = (int)v;   // the single whitespace after the token has been consumed
   return 0;
}

// "%g"-style (precision 6) formatting of a float == default operator<<(ostream&, float),
// the format exportKeypoints uses (hesaff.cpp:125).
inline int fmt_g(char *dst, float v) { return snprintf(dst, 32, "%g", (double)v); }

inline char *fmt_u8(char *p, unsigned v)
{
   if (v >= 100) { *p++ = (char)('0' + v / 100); v %= 100; *p++ = (char)('0' + v / 10); *p++ = (char)('0' + v % 10); }
   else if (v >= 10) { *p++ = (char)('0' + v / 10); *p++ = (char)('0' + v % 10); }
   else *p++ = (char)('0' + v);
   return p;
}

} // namespace

extern "C" {

void hesaff_free(void *p) { free(p); }

int hesaff_read_pnm(const char *path, uint8_t **data, int *width, int *height, int *channels)
{
   if (!path || !data || !width || !height || !channels) return HESAFF_ERR_ARG;
   FILE *f = fopen(path, "rb");
   if (!f) return HESAFF_ERR_IO;
   int c1 = fgetc(f), c2 = fgetc(f);
   int w = 0, h = 0, maxv = 0;
   if (c1 != 'P' || (c2 != '5' && c2 != '6') || pnm_next_int(f, &w) || pnm_next_int(f, &h) || pnm_next_int(f, &maxv) || w < 1 ||
       h < 1 || maxv != 255) {
      fclose(f);
      return HESAFF_ERR_IO;
   }
   const int ch = c2 == '5' ? 1 : 3;
   const size_t n = (size_t)w * h * ch;
   uint8_t *buf = (uint8_t *)malloc(n);
   if (!buf) { fclose(f); return HESAFF_ERR_NOMEM; }
   if (fread(buf, 1, n, f) != n) { free(buf); fclose(f); return HESAFF_ERR_IO; }
   fclose(f);
   *data = buf; *width = w; *height = h; *channels = ch;
   return HESAFF_OK;
}

// hesaff.cpp:115-123: sc = mrSize*s; SVD(A) = U W V^T; M = U diag(1/(w_i^2 sc^2)) U^T
// == (A A^T)^-1 / sc^2.  Evaluated in double, returned as float (the reference runs a
// float Jacobi SVD; agreement ~1e-6 relative, acceptance tolerance 1e-4).
void hesaff_ellipse(const hesaff_keypoint *k, float mrSize, float *a, float *b, float *c)
{
   const float sc = mrSize * k->s;
   const double a11 = k->a11, a12 = k->a12, a21 = k->a21, a22 = k->a22;
   const double m00 = a11 * a11 + a12 * a12, m01 = a11 * a21 + a12 * a22, m11 = a21 * a21 + a22 * a22;
   const double det = m00 * m11 - m01 * m01;
   const double sc2 = (double)sc * (double)sc;
   *a = (float)(m11 / det / sc2);
   *b = (float)(-m01 / det / sc2);
   *c = (float)(m00 / det / sc2);
}

int hesaff_format_sift(const hesaff_keypoint *keys, int n, float mrSize, char **out, size_t *len)
{
   if (n < 0 || (n > 0 && !keys) || !out || !len) return HESAFF_ERR_ARG;
   // worst case per row: 5 floats * 16 + 128 * 4 + 1
   const size_t cap = 64 + (size_t)n * (5 * 16 + 128 * 4 + 2);
   char *buf = (char *)malloc(cap);
   if (!buf) return HESAFF_ERR_NOMEM;
   char *p = buf;
   p += snprintf(p, 64, "%d\n%d\n", 128, n);
   for (int i = 0; i < n; i++) {
      const hesaff_keypoint &k = keys[i];
      float ea, eb, ec;
      hesaff_ellipse(&k, mrSize, &ea, &eb, &ec);
      p += fmt_g(p, k.x); *p++ = ' ';
      p += fmt_g(p, k.y); *p++ = ' ';
      p += fmt_g(p, ea); *p++ = ' ';
      p += fmt_g(p, eb); *p++ = ' ';
      p += fmt_g(p, ec);
      for (int j = 0; j < 128; j++) { *p++ = ' '; p = fmt_u8(p, k.desc[j]); }
      *p++ = '\n';
   }
   *out = buf;
   *len = (size_t)(p - buf);
   return HESAFF_OK;
}

int hesaff_write_sift(const char *path, const hesaff_keypoint *keys, int n, float mrSize)
{
   if (!path) return HESAFF_ERR_ARG;
   char *buf = nullptr;
   size_t len = 0;
   const int rc = hesaff_format_sift(keys, n, mrSize, &buf, &len);
   if (rc != HESAFF_OK) return rc;
   FILE *f = fopen(path, "wb");
   if (!f) { free(buf); return HESAFF_ERR_IO; }
   const size_t w = fwrite(buf, 1, len, f);
   const int ce = fclose(f);
   free(buf);
   return (w == len && ce == 0) ? HESAFF_OK : HESAFF_ERR_IO;
}

} // extern "C"
