// hostio.cpp -- host-only pieces of the drop-in boundary: PGM/PPM reader (replaces
// cv::imread at hesaff.cpp:137), ellipse closed form and the .hesaff.sift text writer
// (replaces exportKeypoints hesaff.cpp:107-130).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <string>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include <cerrno>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

#include "../../include/hesaff_amd.h"
#include "export_fmt.h"

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

// ---- PNG: what cv::imread(path) with its default flag returns for a PNG
// file - 8 bits per channel, alpha dropped, 16-bit samples reduced to their high byte, palette and
// 1/2/4-bit grey expanded - decoded with zlib only.  PNG is lossless, so the pixels do not depend on
// the decoder.  Adam7-interlaced files are de-interlaced pass by pass.
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int read_png_bytes(const std::vector<uint8_t> &f, uint8_t **data, int *width, int *height, int *channels)
{
   static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
   if (f.size() < 8 + 25 || memcmp(f.data(), sig, 8) != 0) return HESAFF_ERR_IO;
   size_t pos = 8;
   uint32_t W = 0, H = 0;
   int depth = 0, ctype = -1;
   bool have_ihdr = false, done = false, interlaced = false;
   std::vector<uint8_t> idat, plte;
   while (!done && pos + 12 <= f.size()) {
      const uint32_t len = be32(&f[pos]);
      if (len > f.size() - pos - 12) return HESAFF_ERR_IO;
      const uint8_t *type = &f[pos + 4], *body = &f[pos + 8];
      if (be32(body + len) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len)) return HESAFF_ERR_IO;
      if (memcmp(type, "IHDR", 4) == 0) {
         if (len != 13 || have_ihdr) return HESAFF_ERR_IO;
         W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9];
         if (body[10] != 0 || body[11] != 0 || body[12] > 1) return HESAFF_ERR_IO;   // compression, filter; interlace 0 or 1 (Adam7)
         interlaced = body[12] == 1;
         if (W < 1 || H < 1 || W > 65535u || H > 65535u) return HESAFF_ERR_IO;
         have_ihdr = true;
      } else if (memcmp(type, "PLTE", 4) == 0) plte.assign(body, body + len);
      else if (memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), body, body + len);
      else if (memcmp(type, "IEND", 4) == 0) done = true;
      pos += 12 + (size_t)len;
   }
   if (!have_ihdr || !done || idat.empty()) return HESAFF_ERR_IO;
   int nch;   // samples per pixel in the file
   switch (ctype) {
      case 0: nch = 1; break;
      case 2: nch = 3; break;
      case 3: nch = 1; break;
      case 4: nch = 2; break;
      case 6: nch = 4; break;
      default: return HESAFF_ERR_IO;
   }
   const bool depth_ok = (ctype == 0) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                         : (ctype == 3) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
   if (!depth_ok || (ctype == 3 && (plte.empty() || plte.size() % 3 != 0))) return HESAFF_ERR_IO;
   const int bpp = std::max(1, nch * depth / 8);   // filter unit
   // passes: the whole image, or the seven Adam7 sub-images (PNG specification, section 8.2): pass p holds the pixels
   // (ys + i * dy, xs + j * dx), each pass is filtered like an image of its own
   struct Pass { uint32_t xs, ys, dx, dy; };
   static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
   static const Pass whole = {0, 0, 1, 1};
   const int npass = interlaced ? 7 : 1;
   const Pass *passes = interlaced ? adam7 : &whole;
   auto pass_w = [&](const Pass &q) { return (W > q.xs) ? (W - q.xs + q.dx - 1) / q.dx : 0u; };
   auto pass_h = [&](const Pass &q) { return (H > q.ys) ? (H - q.ys + q.dy - 1) / q.dy : 0u; };
   unsigned long long rawsize = 0;
   for (int pi = 0; pi < npass; pi++) {
      const unsigned long long pw = pass_w(passes[pi]), ph = pass_h(passes[pi]);
      if (pw && ph) rawsize += ph * ((pw * nch * depth + 7) / 8 + 1);
   }
   // deflate expands by at most ~1032:1: an image the IDAT bytes cannot possibly produce is rejected before
   // anything of its claimed size is allocated (a 60-byte file may claim 65535 x 65535 RGBA16)
   if (rawsize > (unsigned long long)idat.size() * 1040ull + 4096ull) return HESAFF_ERR_IO;
   std::vector<uint8_t> raw((size_t)rawsize);
   uLongf rawlen = (uLongf)raw.size();
   if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return HESAFF_ERR_IO;
   const int och = (ctype == 0 || ctype == 4) ? 1 : 3;
   uint8_t *out = (uint8_t *)malloc((size_t)W * H * och);
   if (!out) return HESAFF_ERR_NOMEM;
   const int step = depth == 16 ? 2 : 1;   // 16-bit samples: the high byte (big-endian first)
   size_t pos_raw = 0;
   std::vector<uint8_t> zero;
   for (int pi = 0; pi < npass; pi++) {
      const Pass &q = passes[pi];
      const uint32_t pw = pass_w(q), ph = pass_h(q);
      if (!pw || !ph) continue;
      const size_t rowbytes = ((size_t)pw * nch * depth + 7) / 8;
      zero.assign(rowbytes, 0);
      for (uint32_t y = 0; y < ph; y++) {
         // unfilter in place (PNG specification, section 9)
         uint8_t *row = &raw[pos_raw + (size_t)y * (rowbytes + 1)];
         const int ft = row[0];
         uint8_t *cur = row + 1;
         const uint8_t *up = y ? cur - (rowbytes + 1) : zero.data();
         switch (ft) {
            case 0: break;
            case 1: for (size_t i = bpp; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
            case 2: for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + up[i]); break;
            case 3:
               for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + (((i >= (size_t)bpp ? cur[i - bpp] : 0) + up[i]) >> 1));
               break;
            case 4:
               for (size_t i = 0; i < rowbytes; i++) {
                  const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up[i], c = i >= (size_t)bpp ? up[i - bpp] : 0;
                  const int pq = a + b - c, pa = std::abs(pq - a), pb = std::abs(pq - b), pc = std::abs(pq - c);
                  cur[i] = (uint8_t)(cur[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
               }
               break;
            default: free(out); return HESAFF_ERR_IO;
         }
         uint8_t *orow = out + (size_t)(q.ys + y * q.dy) * W * och;
         for (uint32_t x = 0; x < pw; x++) {
            uint8_t *o = orow + (size_t)(q.xs + x * q.dx) * och;
            if (depth < 8) {   // grey or palette index, packed most significant bits first
               const int per = 8 / depth, sh = (per - 1 - (int)(x % per)) * depth;
               const int v = (cur[x / per] >> sh) & ((1 << depth) - 1);
               if (ctype == 3) {
                  if ((size_t)v * 3 + 2 >= plte.size()) { free(out); return HESAFF_ERR_IO; }
                  o[0] = plte[3 * v]; o[1] = plte[3 * v + 1]; o[2] = plte[3 * v + 2];
               } else o[0] = (uint8_t)(v * (255 / ((1 << depth) - 1)));
            } else if (ctype == 3) {
               const int v = cur[x];
               if ((size_t)v * 3 + 2 >= plte.size()) { free(out); return HESAFF_ERR_IO; }
               o[0] = plte[3 * v]; o[1] = plte[3 * v + 1]; o[2] = plte[3 * v + 2];
            } else {
               const uint8_t *px = cur + (size_t)x * nch * step;
               if (och == 1) o[0] = px[0];
               else { o[0] = px[0]; o[1] = px[step]; o[2] = px[2 * step]; }
            }
         }
      }
      pos_raw += (size_t)ph * (rowbytes + 1);
   }
   *data = out; *width = (int)W; *height = (int)H; *channels = och;
   return HESAFF_OK;
}

// "%g"-style (precision 6) formatting of a float == default operator<<(ostream&, float),
// the format exportKeypoints uses (hesaff.cpp:125).  snprintf is the definition ...
inline int fmt_g_libc(char *dst, float v) { return snprintf(dst, 32, "%g", (double)v); }

// ... and export_fmt.h (shared with the GPU formatter, kernels_export.h) produces the same bytes with exact integer
// arithmetic over the whole binary32 range.  tests/test_host_side.py compares both on millions of floats.
inline int fmt_g(char *dst, float vf)
{
   HxPtr out{dst};
   hx_fmt_g(out, vf);
   return (int)(out.p - dst);
}

// " 0" .. " 255": the separator and the digits of one descriptor byte as one 4-byte store + its length (a row is 128 of
// them: a table walk instead of 128 divisions and three-way branches)
struct U8Table {
   uint32_t w[256];
   uint8_t len[256];
   U8Table()
   {
      for (unsigned v = 0; v < 256; v++) {
         char t[4] = {' ', 0, 0, 0};
         int n = 1;
         if (v >= 100) { t[n++] = (char)('0' + v / 100); t[n++] = (char)('0' + v / 10 % 10); t[n++] = (char)('0' + v % 10); }
         else if (v >= 10) { t[n++] = (char)('0' + v / 10); t[n++] = (char)('0' + v % 10); }
         else t[n++] = (char)('0' + v);
         memcpy(&w[v], t, 4);
         len[v] = (uint8_t)n;
      }
   }
};
const U8Table kU8;

// Output buffers of tens of MB are recycled: hesaff_free() parks a big block in a small cache and
// the next formatter call takes it again.  First-touch page faults of fresh memory (tens of
// thousands per UHD image, serialised in the kernel across writer threads) otherwise cost more
// than the formatting itself.
struct BigCache {
   std::mutex mu;
   std::unordered_map<void *, size_t> live;           // blocks handed out by big_alloc
   std::vector<std::pair<void *, size_t>> parked;     // freed blocks kept for reuse
   size_t parked_bytes = 0;
   static constexpr size_t kMinBytes = (size_t)1 << 20, kMaxParked = 64, kMaxParkedBytes = (size_t)8 << 30;
   ~BigCache() { for (auto &b : parked) free(b.first); }
};
BigCache g_big;

char *big_alloc(size_t bytes)
{
   if (bytes < BigCache::kMinBytes) return (char *)malloc(bytes);
   {
      std::lock_guard<std::mutex> lk(g_big.mu);
      int best = -1;
      for (int i = 0; i < (int)g_big.parked.size(); i++)
         if (g_big.parked[i].second >= bytes && (best < 0 || g_big.parked[i].second < g_big.parked[best].second)) best = i;
      if (best >= 0) {
         const auto blk = g_big.parked[best];
         g_big.parked.erase(g_big.parked.begin() + best);
         g_big.parked_bytes -= blk.second;
         g_big.live[blk.first] = blk.second;
         return (char *)blk.first;
      }
   }
   const size_t cap = bytes + bytes / 8;   // some head-room so that the next, slightly larger image still fits
   void *q = malloc(cap);
   if (!q) return nullptr;
   std::lock_guard<std::mutex> lk(g_big.mu);
   g_big.live[q] = cap;
   return (char *)q;
}

void big_free(void *p)
{
   if (!p) return;
   {
      std::lock_guard<std::mutex> lk(g_big.mu);
      auto it = g_big.live.find(p);
      if (it != g_big.live.end()) {
         const size_t cap = it->second;
         g_big.live.erase(it);
         if (g_big.parked.size() < BigCache::kMaxParked && g_big.parked_bytes + cap <= BigCache::kMaxParkedBytes) {
            g_big.parked.push_back({p, cap});
            g_big.parked_bytes += cap;
            return;
         }
      }
   }
   free(p);
}

// worst case per row: 5 floats * 16 + 128 * 4 + newline
const size_t kRowMax = 5 * 16 + 128 * 4 + 2;

char *format_rows(const hesaff_keypoint *keys, int i0, int i1, float mrSize, char *p)
{
   for (int i = i0; i < i1; i++) {
      const hesaff_keypoint &k = keys[i];
      float ea, eb, ec;
      hesaff_ellipse(&k, mrSize, &ea, &eb, &ec);
      p += fmt_g(p, k.x); *p++ = ' ';
      p += fmt_g(p, k.y); *p++ = ' ';
      p += fmt_g(p, ea); *p++ = ' ';
      p += fmt_g(p, eb); *p++ = ' ';
      p += fmt_g(p, ec);
      for (int j = 0; j < 128; j++) {   // each store is 4 bytes wide, the next one overwrites what was not a digit
         const unsigned v = k.desc[j];
         memcpy(p, &kU8.w[v], 4);
         p += kU8.len[v];
      }
      *p++ = '\n';
   }
   return p;
}

// Runs task(0..n_tasks-1) on up to `threads` host threads (the caller is one of them).  A thread that cannot be
// created (std::system_error) is simply not used: the tasks it would have taken are picked up by the others.
template <class F> void run_tasks(int n_tasks, int threads, F task)
{
   std::atomic<int> next(0);
   auto work = [&] {
      for (int i; (i = next.fetch_add(1)) < n_tasks;) task(i);
   };
   std::vector<std::thread> th;
   try {
      th.reserve((size_t)std::max(0, threads - 1));
      for (int t = 1; t < threads; t++) th.emplace_back(work);
   } catch (...) {
   }
   work();
   for (auto &x : th) x.join();
}

} // namespace

// No exception crosses the C ABI: allocation failures (std::bad_alloc) and thread-creation failures
// (std::system_error) inside an entry point become error codes.
#define HOSTIO_TRY try {
#define HOSTIO_CATCH                                           \
   }                                                           \
   catch (const std::bad_alloc &) { return HESAFF_ERR_NOMEM; } \
   catch (...) { return HESAFF_ERR_IO; }

extern "C" {

// test hook: the two float formatters side by side (0 = identical bytes)
int hesaff_test_fmt_g(const float *v, int n)
{
   char a[40], b[40];
   int bad = 0;
   for (int i = 0; i < n; i++) {
      const int la = fmt_g(a, v[i]), lb = fmt_g_libc(b, v[i]);
      if (la != lb || memcmp(a, b, (size_t)la) != 0) bad++;
   }
   return bad;
}


void hesaff_free(void *p) { big_free(p); }

// cv::imread's PxM decoder (OpenCV 2.4 modules/highgui/src/grfmt_pxm.cpp, restated from its published source - OpenCV is not
// in this image): P1..P6, maxval 1..65535.
//   * binary 8-bit samples (P5/P6, maxval <= 255) are taken as they are, whatever maxval says;
//   * plain samples (P2/P3) above maxval are clamped to it and mapped through i * 255 / maxval (integer division);
//   * 16-bit samples (maxval > 255; big-endian in the binary forms) are reduced to their high byte, v >> 8;
//   * bitmaps (P1/P4): bit 0 = white (255), bit 1 = black (0); P4 rows are padded to whole bytes.
int hesaff_read_pnm(const char *path, uint8_t **data, int *width, int *height, int *channels)
{
   return hesaff_read_pnm_alloc(path, data, width, height, channels, nullptr, nullptr);
}

int hesaff_read_pnm_alloc(const char *path, uint8_t **data, int *width, int *height, int *channels, hesaff_blob_alloc alloc, void *user)
{
   if (!path || !data || !width || !height || !channels) return HESAFF_ERR_ARG;
   *data = nullptr;
   // a buffer from the caller's allocator is the caller's whatever happens: on a failure it is handed back through *data, never free()d
   auto drop = [&](uint8_t *q) { if (alloc) *data = q; else free(q); };
   FILE *f = fopen(path, "rb");
   if (!f) return HESAFF_ERR_IO;
   const int c1 = fgetc(f), c2 = fgetc(f);
   int w = 0, h = 0, maxv = 1;
   const bool known = c1 == 'P' && c2 >= '1' && c2 <= '6';
   const int kind = known ? c2 - '0' : 0;
   const bool bitmap = kind == 1 || kind == 4, plain = kind >= 1 && kind <= 3;
   if (!known || pnm_next_int(f, &w) || pnm_next_int(f, &h) || (!bitmap && pnm_next_int(f, &maxv)) || w < 1 || h < 1 || maxv < 1 || maxv > 65535) {
      fclose(f);
      return HESAFF_ERR_IO;
   }
   const int ch = (kind == 3 || kind == 6) ? 3 : 1;
   const size_t n = (size_t)w * h * ch;
   // a file cannot hold more samples than it has bytes (plain: a digit and a separator each; bitmaps: a bit each):
   // an absurd header is refused before anything of its claimed size is allocated
   {
      const long here = ftell(f);
      fseek(f, 0, SEEK_END);
      const long end = ftell(f);
      fseek(f, here, SEEK_SET);
      const unsigned long long avail = (here >= 0 && end >= here) ? (unsigned long long)(end - here) : 0ull;
      const unsigned long long need = kind == 4 ? (unsigned long long)h * (((unsigned long long)w + 7) / 8)
                                     : plain ? (kind == 1 ? n : 2 * n - 1) : n * (maxv > 255 ? 2ull : 1ull);
      if (avail < need) { fclose(f); return HESAFF_ERR_IO; }
   }
   int zeroed = 0;
   uint8_t *buf = alloc ? (uint8_t *)alloc(n, &zeroed, user) : (uint8_t *)malloc(n);   // every byte is written below
   if (!buf) { fclose(f); return HESAFF_ERR_NOMEM; }
   bool ok = true;
   if (kind == 5 || kind == 6) {
      if (maxv <= 255) ok = fread(buf, 1, n, f) == n;
      else {
         std::vector<uint8_t> row;
         try { row.resize((size_t)w * ch * 2); } catch (...) { drop(buf); fclose(f); return HESAFF_ERR_NOMEM; }
         for (int y = 0; y < h && ok; y++) {
            ok = fread(row.data(), 1, row.size(), f) == row.size();
            uint8_t *o = buf + (size_t)y * w * ch;
            for (size_t x = 0; ok && x < (size_t)w * ch; x++) o[x] = row[2 * x];   // big-endian: the first byte is the high one
         }
      }
   } else if (kind == 4) {
      const size_t rb = ((size_t)w + 7) / 8;
      std::vector<uint8_t> row;
      try { row.resize(rb); } catch (...) { drop(buf); fclose(f); return HESAFF_ERR_NOMEM; }
      for (int y = 0; y < h && ok; y++) {
         ok = fread(row.data(), 1, rb, f) == rb;
         for (int x = 0; ok && x < w; x++) buf[(size_t)y * w + x] = ((row[(size_t)x >> 3] >> (7 - (x & 7))) & 1) ? 0 : 255;
      }
   } else if (kind == 1) {
      // plain bitmap: the digits need no separators
      for (size_t i = 0; i < n && ok; i++) {
         int c = fgetc(f);
         for (;;) {
            while (c == ' ' || c == '\t' || c == '\n' || c == '\r') c = fgetc(f);
            if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
            break;
         }
         if (c != '0' && c != '1') ok = false;
         else buf[i] = c == '1' ? 0 : 255;
      }
   } else {
      for (size_t i = 0; i < n && ok; i++) {
         int v = 0;
         if (pnm_next_int(f, &v)) { ok = false; break; }
         if (v > maxv) v = maxv;
         buf[i] = maxv > 255 ? (uint8_t)(v >> 8) : (uint8_t)(v * 255 / maxv);
      }
   }
   fclose(f);
   if (!ok) { drop(buf); return HESAFF_ERR_IO; }
   *data = buf; *width = w; *height = h; *channels = ch;
   return HESAFF_OK;
}

int hesaff_read_png(const char *path, uint8_t **data, int *width, int *height, int *channels)
{
   if (!path || !data || !width || !height || !channels) return HESAFF_ERR_ARG;
   FILE *f = fopen(path, "rb");
   if (!f) return HESAFF_ERR_IO;
   HOSTIO_TRY
   std::vector<uint8_t> bytes;
   try {
      uint8_t chunk[1 << 16];
      for (size_t n; (n = fread(chunk, 1, sizeof chunk, f)) > 0;) bytes.insert(bytes.end(), chunk, chunk + n);
   } catch (...) {
      fclose(f);
      throw;
   }
   fclose(f);
   return read_png_bytes(bytes, data, width, height, channels);
   HOSTIO_CATCH
}


// cv::imread (hesaff.cpp:137) for Windows bitmaps, the way OpenCV's BMP decoder (grfmt_bmp.cpp) delivers them at imread's default flag:
// 1 / 4 / 8 bits per pixel through the palette (BI_RGB, BI_RLE4, BI_RLE8), 16 bits as 5-5-5 or - BI_BITFIELDS with a green mask of 0x7e0 -
// 5-6-5 with the low bits left zero (5-5-5: b = v << 3, g = (v >> 2) & ~7, r = (v >> 7) & ~7; 5-6-5: g = (v >> 3) & ~3, r = (v >> 8) & ~7), 24 bits, 32 bits
// with the fourth byte dropped; bottom-up or top-down rows, OS/2 core headers (12 bytes, 3-byte palette entries) as well.  Pixels a
// run-length stream skips (delta escapes, short lines) keep palette entry 0.  -> R, G, B order, 3 channels (1 when every palette entry the
// file can address is grey: (g + g + g) / 3.0f is g exactly, so the grey conversion of hesaff.cpp:145 sees the same image either way).
static int read_bmp_bytes(const std::vector<uint8_t> &b, uint8_t **data, int *width, int *height, int *channels)
{
   auto u16 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8); };
   auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
   if (b.size() < 26 || b[0] != 'B' || b[1] != 'M') return HESAFF_ERR_IO;
   const size_t off_bits = u32(10), hsize = u32(14);
   long long w = 0, h = 0;
   uint32_t bpp = 0, comp = 0, clr_used = 0;
   bool core = false;
   if (hsize == 12) { core = true; w = u16(18); h = u16(20); bpp = u16(24); }
   else if (hsize >= 40 && b.size() >= 14 + 40) {
      w = (int32_t)u32(18); h = (int32_t)u32(22); bpp = u16(28); comp = u32(30); clr_used = u32(46);
   } else return HESAFF_ERR_IO;
   const bool top_down = h < 0;
   if (top_down) h = -h;
   if (w < 1 || h < 1 || w > (1 << 30) / h || u16(core ? 22 : 26) != 1) return HESAFF_ERR_IO;
   if (!(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 16 || bpp == 24 || bpp == 32)) return HESAFF_ERR_IO;
   if (!(comp == 0 || (comp == 1 && bpp == 8) || (comp == 2 && bpp == 4) || (comp == 3 && (bpp == 16 || bpp == 32)))) return HESAFF_ERR_IO;
   if (top_down && (comp == 1 || comp == 2)) return HESAFF_ERR_IO;
   bool is565 = false;
   if (comp == 3) {
      const size_t mo = 14 + 40;   // the three masks follow a 40-byte header, or are its fields 40.. in the V4 / V5 headers: the same offset
      if (b.size() < mo + 12) return HESAFF_ERR_IO;
      const uint32_t rm = u32(mo), gm = u32(mo + 4), bm = u32(mo + 8);
      if (bpp == 16) {
         if (rm == 0xf800 && gm == 0x7e0 && bm == 0x1f) is565 = true;
         else if (!(rm == 0x7c00 && gm == 0x3e0 && bm == 0x1f)) return HESAFF_ERR_IO;
      } else if (!(rm == 0xff0000 && gm == 0xff00 && bm == 0xff)) return HESAFF_ERR_IO;
   }
   uint8_t pal[256][3];   // R, G, B
   memset(pal, 0, sizeof pal);
   bool grey_pal = false;
   if (bpp <= 8) {
      const size_t po = 14 + hsize, esz = core ? 3 : 4;
      size_t n = clr_used ? clr_used : ((size_t)1 << bpp);
      if (n > ((size_t)1 << bpp)) n = (size_t)1 << bpp;
      if (po + n * esz > b.size()) return HESAFF_ERR_IO;
      grey_pal = true;
      for (size_t i = 0; i < n; i++) {
         pal[i][2] = b[po + i * esz]; pal[i][1] = b[po + i * esz + 1]; pal[i][0] = b[po + i * esz + 2];
         if (pal[i][0] != pal[i][1] || pal[i][1] != pal[i][2]) grey_pal = false;
      }
   }
   const int ch = grey_pal ? 1 : 3;
   const size_t npix = (size_t)w * (size_t)h;
   uint8_t *out = (uint8_t *)malloc(npix * ch);
   if (!out) return HESAFF_ERR_NOMEM;
   auto put = [&](long long x, long long yfile, const uint8_t *rgb) {   // yfile: row in file order
      const long long y = top_down ? yfile : h - 1 - yfile;
      uint8_t *o = out + ((size_t)y * w + x) * ch;
      if (ch == 1) o[0] = rgb[0]; else { o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; }
   };
   bool ok = off_bits <= b.size();
   if (ok && comp != 1 && comp != 2) {
      const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
      if (stride * (size_t)h > b.size() - off_bits) ok = false;   // (w h <= 2^30: no overflow)
      for (long long yf = 0; ok && yf < h; yf++) {
         const uint8_t *row = b.data() + off_bits + (size_t)yf * stride;
         for (long long x = 0; x < w; x++) {
            uint8_t rgb[3];
            if (bpp <= 8) {
               const uint32_t idx = bpp == 8 ? row[x] : bpp == 4 ? ((row[x >> 1] >> ((x & 1) ? 0 : 4)) & 15) : ((row[x >> 3] >> (7 - (x & 7))) & 1);
               rgb[0] = pal[idx][0]; rgb[1] = pal[idx][1]; rgb[2] = pal[idx][2];
            } else if (bpp == 16) {
               const uint32_t v = (uint32_t)row[2 * x] | ((uint32_t)row[2 * x + 1] << 8);
               rgb[2] = (uint8_t)(v << 3);
               rgb[1] = is565 ? (uint8_t)((v >> 3) & ~3u) : (uint8_t)((v >> 2) & ~7u);
               rgb[0] = is565 ? (uint8_t)((v >> 8) & ~7u) : (uint8_t)((v >> 7) & ~7u);
            } else {
               const uint8_t *q = row + x * (bpp / 8);
               rgb[0] = q[2]; rgb[1] = q[1]; rgb[2] = q[0];
            }
            put(x, yf, rgb);
         }
      }
   } else if (ok) {
      // BI_RLE8 / BI_RLE4: (count, value) runs; 0,0 end of line; 0,1 end of bitmap; 0,2,dx,dy move; 0,n >= 3: n literal pixels, padded to 16 bits
      for (size_t i = 0; i < npix; i++) { if (ch == 1) out[i] = pal[0][0]; else { out[3 * i] = pal[0][0]; out[3 * i + 1] = pal[0][1]; out[3 * i + 2] = pal[0][2]; } }
      size_t p = off_bits;
      long long x = 0, yf = 0;
      auto pix = [&](uint32_t idx) { if (x < w && yf < h) put(x, yf, pal[idx]); x++; };
      bool done = false;
      while (!done && p + 1 < b.size()) {
         const uint32_t n = b[p], v = b[p + 1];
         p += 2;
         if (n > 0) {
            for (uint32_t k = 0; k < n; k++) pix(bpp == 8 ? v : ((k & 1) ? (v & 15) : (v >> 4)));
         } else if (v == 0) { x = 0; yf++; }
         else if (v == 1) done = true;
         else if (v == 2) { if (p + 1 >= b.size()) { ok = false; break; } x += b[p]; yf += b[p + 1]; p += 2; }
         else {
            const size_t nbytes = bpp == 8 ? v : (v + 1) / 2;
            if (p + nbytes > b.size()) { ok = false; break; }
            for (uint32_t k = 0; k < v; k++) pix(bpp == 8 ? b[p + k] : ((k & 1) ? (b[p + k / 2] & 15) : (b[p + k / 2] >> 4)));
            p += (nbytes + 1) & ~(size_t)1;
         }
      }
   }
   if (!ok) { free(out); return HESAFF_ERR_IO; }
   *data = out; *width = (int)w; *height = (int)h; *channels = ch;
   return HESAFF_OK;
}

int hesaff_read_bmp(const char *path, uint8_t **data, int *width, int *height, int *channels)
{
   if (!path || !data || !width || !height || !channels) return HESAFF_ERR_ARG;
   FILE *f = fopen(path, "rb");
   if (!f) return HESAFF_ERR_IO;
   HOSTIO_TRY
   std::vector<uint8_t> bytes;
   try {
      uint8_t chunk[1 << 16];
      for (size_t n; (n = fread(chunk, 1, sizeof chunk, f)) > 0;) bytes.insert(bytes.end(), chunk, chunk + n);
   } catch (...) {
      fclose(f);
      throw;
   }
   fclose(f);
   return read_bmp_bytes(bytes, data, width, height, channels);
   HOSTIO_CATCH
}

// cv::imread (hesaff.cpp:137) for TIFF files.  OpenCV hands 8-bit TIFFs to libtiff's RGBA interface (TIFFReadRGBAStrip / Tile) and drops the
// alpha byte; this reader restates that interface for the baseline forms - what scanners, cameras' converters and image tools write:
//   * bilevel / 2 / 4 / 8-bit grey (MinIsBlack, MinIsWhite: v * 255 / (2^bits - 1), inverted for MinIsWhite), 8-bit palette (the 16-bit colour map
//     reduced with >> 8, or taken as it is when no entry exceeds 255: libtiff's checkcmap), 8-bit RGB, RGB + one extra sample (unassociated
//     alpha is multiplied in the way libtiff does, (v * a + 127) / 255; associated alpha and unspecified extra samples are dropped);
//   * strips or tiles, chunky or planar, little- or big-endian, the first directory of the file;
//   * no compression, PackBits, LZW (both bit orders of the code stream libtiff accepts: the post-6.0 MSB-first form only), Deflate (8 / 32946),
//     horizontal differencing (predictor 2) for 8-bit samples.
// Everything else - 16-bit and floating-point samples, YCbCr / CMYK / Lab, JPEG- and fax-compressed data, BigTIFF - returns HESAFF_ERR_IO.
// -> 1 channel for grey files, 3 (R, G, B) otherwise.
namespace {
struct TiffFile {
   const std::vector<uint8_t> &b;
   bool be = false;
   explicit TiffFile(const std::vector<uint8_t> &bytes) : b(bytes) {}
   bool has(size_t o, size_t n) const { return o <= b.size() && n <= b.size() - o; }
   uint32_t u16(size_t o) const { return be ? ((uint32_t)b[o] << 8) | b[o + 1] : ((uint32_t)b[o + 1] << 8) | b[o]; }
   uint32_t u32(size_t o) const { return be ? (u16(o) << 16) | u16(o + 2) : (u16(o + 2) << 16) | u16(o); }
};

// values of one directory entry (types BYTE 1, SHORT 3, LONG 4) as 32-bit numbers; false: another type or out of the file
bool tiff_values(const TiffFile &t, size_t entry, std::vector<uint32_t> &out, size_t max_count)
{
   const uint32_t type = t.u16(entry + 2), count = t.u32(entry + 4);
   const size_t sz = type == 1 ? 1 : type == 3 ? 2 : type == 4 ? 4 : 0;
   if (sz == 0 || count == 0 || count > max_count) return false;
   size_t at = entry + 8;
   if ((size_t)count * sz > 4) { at = t.u32(entry + 8); if (!t.has(at, (size_t)count * sz)) return false; }
   out.resize(count);
   for (uint32_t i = 0; i < count; i++) out[i] = sz == 1 ? t.b[at + i] : sz == 2 ? t.u16(at + 2 * i) : t.u32(at + 4 * i);
   return true;
}

// TIFF 6.0 section 9: PackBits
bool tiff_packbits(const uint8_t *src, size_t n, uint8_t *dst, size_t want)
{
   size_t i = 0, o = 0;
   while (o < want && i < n) {
      const int c = (int8_t)src[i++];
      if (c >= 0) { const size_t m = (size_t)c + 1; if (i + m > n || o + m > want) return false; memcpy(dst + o, src + i, m); i += m; o += m; }
      else if (c != -128) { const size_t m = (size_t)(1 - c); if (i >= n || o + m > want) return false; memset(dst + o, src[i++], m); o += m; }
   }
   return o == want;
}

// TIFF 6.0 section 13: LZW, MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, "early change" of the code width
bool tiff_lzw(const uint8_t *src, size_t n, uint8_t *dst, size_t want)
{
   struct Ent { uint16_t prev; uint16_t len; uint8_t first, last; };
   std::vector<Ent> tab(4096);
   for (int i = 0; i < 256; i++) tab[(size_t)i] = {0xffff, 1, (uint8_t)i, (uint8_t)i};
   uint32_t acc = 0; int nbits = 0, width = 9; size_t i = 0, o = 0; int next = 258, prev = -1;
   auto emit = [&](int code) -> bool {
      const size_t len = tab[(size_t)code].len;
      if (o + len > want) return false;
      size_t at = o + len;
      for (int c = code; c != 0xffff; c = tab[(size_t)c].prev) dst[--at] = tab[(size_t)c].last;
      o += len;
      return true;
   };
   while (o < want) {
      while (nbits < width) { if (i >= n) return false; acc = (acc << 8) | src[i++]; nbits += 8; }
      const int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1));
      nbits -= width;
      if (code == 257) break;
      if (code == 256) { next = 258; width = 9; prev = -1; continue; }
      if (prev < 0) { if (code > 255 || !emit(code)) return false; prev = code; continue; }
      if (code < next) {
         if (!emit(code)) return false;
         if (next < 4096) { tab[(size_t)next] = {(uint16_t)prev, (uint16_t)(tab[(size_t)prev].len + 1), tab[(size_t)prev].first, tab[(size_t)code].first}; next++; }
      } else if (code == next && next < 4096) {
         tab[(size_t)next] = {(uint16_t)prev, (uint16_t)(tab[(size_t)prev].len + 1), tab[(size_t)prev].first, tab[(size_t)prev].first};
         next++;
         if (!emit(code)) return false;
      } else return false;
      prev = code;
      if (next == 511 && width == 9) width = 10;
      else if (next == 1023 && width == 10) width = 11;
      else if (next == 2047 && width == 11) width = 12;
   }
   return o == want;
}

int read_tiff_bytes(const std::vector<uint8_t> &bytes, uint8_t **data, int *width, int *height, int *channels)
{
   TiffFile t(bytes);
   if (bytes.size() < 8) return HESAFF_ERR_IO;
   if (bytes[0] == 'M' && bytes[1] == 'M') t.be = true;
   else if (!(bytes[0] == 'I' && bytes[1] == 'I')) return HESAFF_ERR_IO;
   if (t.u16(2) != 42) return HESAFF_ERR_IO;   // (43: BigTIFF)
   const size_t ifd = t.u32(4);
   if (!t.has(ifd, 2)) return HESAFF_ERR_IO;
   const uint32_t nent = t.u16(ifd);
   if (!t.has(ifd + 2, (size_t)nent * 12)) return HESAFF_ERR_IO;
   uint32_t W = 0, H = 0, comp = 1, photo = 0xffff, spp = 1, rps = 0xffffffffu, planar = 1, pred = 1, tw = 0, th = 0, fill = 1, extra = 0xffff;
   std::vector<uint32_t> bps{1}, offs, counts, cmap, v;
   bool tiled = false;
   for (uint32_t e = 0; e < nent; e++) {
      const size_t at = ifd + 2 + (size_t)e * 12;
      const uint32_t tag = t.u16(at);
      const bool ok = tiff_values(t, at, v, tag == 320 ? 3 * 256 : (size_t)1 << 24);
      switch (tag) {
         case 256: if (!ok) return HESAFF_ERR_IO; W = v[0]; break;
         case 257: if (!ok) return HESAFF_ERR_IO; H = v[0]; break;
         case 258: if (!ok) return HESAFF_ERR_IO; bps = v; break;
         case 259: if (!ok) return HESAFF_ERR_IO; comp = v[0]; break;
         case 262: if (!ok) return HESAFF_ERR_IO; photo = v[0]; break;
         case 266: if (ok) fill = v[0]; break;
         case 273: if (!ok) return HESAFF_ERR_IO; offs = v; break;
         case 277: if (!ok) return HESAFF_ERR_IO; spp = v[0]; break;
         case 278: if (ok) rps = v[0]; break;
         case 279: if (!ok) return HESAFF_ERR_IO; counts = v; break;
         case 284: if (ok) planar = v[0]; break;
         case 317: if (ok) pred = v[0]; break;
         case 320: if (!ok) return HESAFF_ERR_IO; cmap = v; break;
         case 322: if (!ok) return HESAFF_ERR_IO; tw = v[0]; tiled = true; break;
         case 323: if (!ok) return HESAFF_ERR_IO; th = v[0]; break;
         case 324: if (!ok) return HESAFF_ERR_IO; offs = v; break;
         case 325: if (!ok) return HESAFF_ERR_IO; counts = v; break;
         case 338: if (ok) extra = v[0]; break;
         case 339: if (ok) for (uint32_t f : v) if (f != 1) return HESAFF_ERR_IO; break;   // SampleFormat: unsigned integers only
         default: break;
      }
   }
   if (W < 1 || H < 1 || W > (1u << 30) / H || spp < 1 || spp > 4 || bps.size() < 1) return HESAFF_ERR_IO;
   const uint32_t bits = bps[0];
   for (uint32_t q : bps) if (q != bits) return HESAFF_ERR_IO;
   if (!(comp == 1 || comp == 5 || comp == 8 || comp == 32946 || comp == 32773) || fill != 1 || (planar != 1 && planar != 2)) return HESAFF_ERR_IO;
   if (pred != 1 && !(pred == 2 && bits == 8)) return HESAFF_ERR_IO;
   bool grey = false;
   if (photo == 0 || photo == 1) { if (spp != 1 || !(bits == 1 || bits == 2 || bits == 4 || bits == 8)) return HESAFF_ERR_IO; grey = true; }
   else if (photo == 3) { if (spp != 1 || bits != 8 || cmap.size() != 3 * 256) return HESAFF_ERR_IO; }
   else if (photo == 2) { if (!(spp == 3 || spp == 4) || bits != 8) return HESAFF_ERR_IO; }
   else return HESAFF_ERR_IO;
   if (planar == 2 && spp == 1) planar = 1;
   // chunk geometry: strips are tiles as wide as the image
   if (!tiled) { tw = W; th = std::min(rps, H); if (th == 0) return HESAFF_ERR_IO; }
   else if (tw < 1 || th < 1 || tw > (1u << 20) || th > (1u << 20)) return HESAFF_ERR_IO;
   const uint32_t tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
   const size_t planes = planar == 2 ? spp : 1, spc = planar == 2 ? 1 : spp;   // samples per pixel inside one chunk
   const size_t nchunks = (size_t)tx * ty * planes;
   if (offs.size() != nchunks || counts.size() != nchunks) return HESAFF_ERR_IO;
   const size_t row_bytes = ((size_t)tw * spc * bits + 7) / 8;
   if ((size_t)th > ((size_t)1 << 31) / std::max<size_t>(row_bytes, 1)) return HESAFF_ERR_IO;
   bool cmap16 = false;
   for (uint32_t q : cmap) if (q >= 256) cmap16 = true;   // libtiff's checkcmap: a map that stays below 256 is an 8-bit map
   const int ch = grey ? 1 : 3;
   uint8_t *out = (uint8_t *)malloc((size_t)W * H * ch);
   if (!out) return HESAFF_ERR_NOMEM;
   std::vector<uint8_t> chunk;
   std::vector<uint8_t> rgba;   // planar files: the samples of all planes gathered per pixel
   bool ok = true;
   try {
      if (planar == 2) rgba.assign((size_t)W * H * spp, 0);
      for (size_t c = 0; ok && c < nchunks; c++) {
         const size_t plane = c / ((size_t)tx * ty), ci = c % ((size_t)tx * ty), cx = ci % tx, cy = ci / tx;
         const uint32_t rows = tiled ? th : std::min<uint32_t>(th, H - (uint32_t)cy * th);
         const size_t want = row_bytes * rows;
         if (!t.has(offs[c], counts[c]) || (comp == 1 && counts[c] < want)) { ok = false; break; }   // (before the buffer is made: a header may claim gigabytes)
         chunk.resize(want);
         const uint8_t *src = bytes.data() + offs[c];
         if (comp == 1) memcpy(chunk.data(), src, want);
         else if (comp == 32773) ok = tiff_packbits(src, counts[c], chunk.data(), want);
         else if (comp == 5) ok = tiff_lzw(src, counts[c], chunk.data(), want);
         else { uLongf len = (uLongf)want; ok = uncompress(chunk.data(), &len, src, (uLong)counts[c]) == Z_OK && len == want; }
         if (!ok) break;
         if (pred == 2)
            for (uint32_t r = 0; r < rows; r++) { uint8_t *q = chunk.data() + (size_t)r * row_bytes; for (size_t k = spc; k < (size_t)tw * spc; k++) q[k] = (uint8_t)(q[k] + q[k - spc]); }
         for (uint32_t r = 0; r < rows; r++) {
            const size_t y = (size_t)cy * th + r;
            if (y >= H) break;
            const uint8_t *q = chunk.data() + (size_t)r * row_bytes;
            for (uint32_t xx = 0; xx < tw; xx++) {
               const size_t x = (size_t)cx * tw + xx;
               if (x >= W) break;
               if (planar == 2) { rgba[(y * W + x) * spp + plane] = q[xx]; continue; }
               uint8_t *o = out + (y * W + x) * ch;
               if (grey) {
                  uint32_t vv = bits == 8 ? q[xx] : bits == 4 ? ((q[xx >> 1] >> ((xx & 1) ? 0 : 4)) & 15) : bits == 2 ? ((q[xx >> 2] >> (6 - 2 * (xx & 3))) & 3) : ((q[xx >> 3] >> (7 - (xx & 7))) & 1);
                  vv = bits == 8 ? vv : vv * 255 / ((1u << bits) - 1);
                  o[0] = (uint8_t)(photo == 0 ? 255 - vv : vv);
               } else if (photo == 3) {
                  const uint32_t idx = q[xx];
                  for (int k = 0; k < 3; k++) o[k] = (uint8_t)(cmap16 ? cmap[(size_t)k * 256 + idx] >> 8 : cmap[(size_t)k * 256 + idx]);
               } else {
                  const uint8_t *px = q + (size_t)xx * spp;
                  if (spp == 4 && extra == 2) { const uint32_t a = px[3]; for (int k = 0; k < 3; k++) o[k] = (uint8_t)((px[k] * a + 127) / 255); }
                  else { o[0] = px[0]; o[1] = px[1]; o[2] = px[2]; }
               }
            }
         }
      }
      if (ok && planar == 2)
         for (size_t i = 0; i < (size_t)W * H; i++) {
            const uint8_t *px = rgba.data() + i * spp;
            uint8_t *o = out + i * 3;
            if (spp == 4 && extra == 2) { const uint32_t a = px[3]; for (int k = 0; k < 3; k++) o[k] = (uint8_t)((px[k] * a + 127) / 255); }
            else { o[0] = px[0]; o[1] = px[1]; o[2] = px[2]; }
         }
   } catch (...) { free(out); throw; }
   if (!ok) { free(out); return HESAFF_ERR_IO; }
   *data = out; *width = (int)W; *height = (int)H; *channels = ch;
   return HESAFF_OK;
}
} // namespace

int hesaff_read_tiff(const char *path, uint8_t **data, int *width, int *height, int *channels)
{
   if (!path || !data || !width || !height || !channels) return HESAFF_ERR_ARG;
   FILE *f = fopen(path, "rb");
   if (!f) return HESAFF_ERR_IO;
   HOSTIO_TRY
   std::vector<uint8_t> bytes;
   try {
      uint8_t chunk[1 << 16];
      for (size_t n; (n = fread(chunk, 1, sizeof chunk, f)) > 0;) bytes.insert(bytes.end(), chunk, chunk + n);
   } catch (...) {
      fclose(f);
      throw;
   }
   fclose(f);
   return read_tiff_bytes(bytes, data, width, height, channels);
   HOSTIO_CATCH
}

// the imread of hesaff.cpp:137 for the formats this library decodes itself: PBM/PGM/PPM, PNG, JPEG, BMP and baseline TIFF, by magic number
int hesaff_read_image(const char *path, uint8_t **data, int *width, int *height, int *channels)
{
   return hesaff_read_image_alloc(path, data, width, height, channels, nullptr, nullptr);
}

int hesaff_read_image_alloc(const char *path, uint8_t **data, int *width, int *height, int *channels, hesaff_blob_alloc alloc, void *user)
{
   if (!path || !data || !width || !height || !channels) return HESAFF_ERR_ARG;
   FILE *f = fopen(path, "rb");
   if (!f) return HESAFF_ERR_IO;
   const int c1 = fgetc(f), c2 = fgetc(f);
   fclose(f);
   if (c1 == 'P' && c2 >= '1' && c2 <= '6') return hesaff_read_pnm_alloc(path, data, width, height, channels, alloc, user);
   if (c1 == 0x89 && c2 == 'P') return hesaff_read_png(path, data, width, height, channels);
   if (c1 == 0xFF && c2 == 0xD8) return hesaff_read_jpeg(path, data, width, height, channels);
   if (c1 == 'B' && c2 == 'M') return hesaff_read_bmp(path, data, width, height, channels);
   if ((c1 == 'I' && c2 == 'I') || (c1 == 'M' && c2 == 'M')) return hesaff_read_tiff(path, data, width, height, channels);
   return HESAFF_ERR_IO;
}

// hesaff.cpp:115-123 in closed form: export_fmt.h (one definition for the host writer and the GPU formatter)
void hesaff_ellipse(const hesaff_keypoint *k, float mrSize, float *a, float *b, float *c)
{
   hx_ellipse(k->s, k->a11, k->a12, k->a21, k->a22, mrSize, a, b, c);
}

int hesaff_format_sift(const hesaff_keypoint *keys, int n, float mrSize, char **out, size_t *len)
{
   return hesaff_format_sift_mt(keys, n, mrSize, 1, out, len);
}

// Worker threads when the caller says "auto": CPUs of the affinity mask, no more than the cgroup's CPU-time limit
// (threads beyond it only time-slice: a 256-CPU host with a 16-CPU quota formats fastest on 16), 1..64.
int hesaff_host_threads(void)
{
   static const int cached = [] {
      unsigned n = std::max(1u, std::thread::hardware_concurrency());
      cpu_set_t set;
      if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)std::max(1, CPU_COUNT(&set));
      if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
         char q[32] = {0};
         double period = 0;
         if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const double cpus = atof(q) / period;
            if (cpus >= 1.0 && cpus < (double)n) n = (unsigned)cpus;
         }
         fclose(f);
      }
      return (int)std::min(std::max(n, 1u), 64u);
   }();
   return cached;
}

// hesaff_host_plan_for's stage_threads for a pool of decode_threads + write_threads (callers of hesaff_process_files pass the two counts
// of their plan, not the plan): the largest s in 1..4 with clamp((pool + s) / 4, 1, 4) == s.  (The rule is not one-to-one - 7 and 8
// CPUs both give a pool of 6 - and the larger count is taken: cpus = 8 -> 2, 16 -> 4, as hesaff_host_plan_for says.)
int hesaff_stage_threads_for_pool(int pool)
{
   int best = 1;
   for (int s = 1; s <= 4; s++)
      if (std::max(1, std::min(4, (pool + s) / 4)) == s) best = s;
   return best;
}

int hesaff_host_plan_for(int devices_sharing_host, hesaff_host_plan *out)
{
   if (!out || devices_sharing_host < 1) return HESAFF_ERR_ARG;
   const int cpus = std::max(1, hesaff_host_threads() / devices_sharing_host);
   out->cpus = cpus;
   out->stage_threads = std::max(1, std::min(4, cpus / 4));
   const int pool = std::max(2, cpus - out->stage_threads);
   out->decode_threads = std::max(1, pool / 4);
   out->write_threads = pool - out->decode_threads;
   return HESAFF_OK;
}

// Rows are formatted by `threads` workers into one buffer.
int hesaff_format_sift_mt(const hesaff_keypoint *keys, int n, float mrSize, int threads, char **out, size_t *len)
{
   if (n < 0 || (n > 0 && !keys) || !out || !len) return HESAFF_ERR_ARG;
   int T = threads > 0 ? threads : hesaff_host_threads();
   T = std::max(1, std::min(T, n / 4096 + 1));   // below ~4 k rows a thread costs more than it saves
   char head[64];
   const int hl = snprintf(head, sizeof head, "%d\n%d\n", 128, n);
   if (T == 1) {
      char *buf = big_alloc((size_t)hl + (size_t)n * kRowMax + 1);
      if (!buf) return HESAFF_ERR_NOMEM;
      memcpy(buf, head, (size_t)hl);
      char *e = format_rows(keys, 0, n, mrSize, buf + hl);
      *out = buf;
      *len = (size_t)(e - buf);
      return HESAFF_OK;
   }
   // one allocation at the worst-case row size: worker t formats its rows at their worst-case
   // offset, then the blocks are moved down over the slack (left to right, so never onto unread data)
   char *buf = big_alloc((size_t)hl + (size_t)n * kRowMax + 1);
   if (!buf) return HESAFF_ERR_NOMEM;
   memcpy(buf, head, (size_t)hl);
   std::vector<size_t> plen;
   try { plen.assign((size_t)T, 0); } catch (...) { big_free(buf); return HESAFF_ERR_NOMEM; }
   run_tasks(T, T, [&](int t) {
      const int i0 = (int)((long long)n * t / T), i1 = (int)((long long)n * (t + 1) / T);
      char *dst = buf + hl + (size_t)i0 * kRowMax;
      plen[t] = (size_t)(format_rows(keys, i0, i1, mrSize, dst) - dst);
   });
   size_t o = (size_t)hl + plen[0];
   for (int t = 1; t < T; t++) {
      const int i0 = (int)((long long)n * t / T);
      memmove(buf + o, buf + hl + (size_t)i0 * kRowMax, plen[t]);
      o += plen[t];
   }
   *out = buf;
   *len = o;
   return HESAFF_OK;
}

// a whole buffer to an open descriptor
static bool write_all(int fd, const char *q, size_t left)
{
   while (left > 0) {
      const ssize_t w = write(fd, q, left < ((size_t)1 << 30) ? left : ((size_t)1 << 30));
      if (w < 0) { if (errno == EINTR) continue; return false; }
      q += w; left -= (size_t)w;
   }
   return true;
}

// header + body with ONE system call when the kernel takes it whole (it does for a regular file with room: a UHD image's 46 MB of rows
// go out in one writev); what a short write leaves is finished by write_all
static bool write_head_body(int fd, const char *head, size_t hl, const char *body, size_t len)
{
   struct iovec iov[2] = {{(void *)head, hl}, {(void *)body, len < ((size_t)1 << 30) ? len : ((size_t)1 << 30)}};
   ssize_t w;
   do { w = writev(fd, iov, len > 0 ? 2 : 1); } while (w < 0 && errno == EINTR);
   if (w < 0) return false;
   size_t done = (size_t)w;
   if (done < hl) { if (!write_all(fd, head + done, hl - done)) return false; done = hl; }
   return write_all(fd, body + (done - hl), len - (done - hl));
}

// Every writer of this file puts its output under a temporary name next to `path` ("<path>.part.<pid>.<tid>": two writers of one
// path - a caller's mistake - never share it) and renames it when it is complete and closed: a killed run leaves no torn file under the
// final name, so a regular file that exists under its final name is whole (hesaff_set_resume relies on it).  Where that cannot work the
// writer falls back to writing `path` itself: a target that exists and is not a regular file (/dev/stdout, a FIFO, a device), or a
// directory in which no new file can be created (an existing writable file in a read-only directory).
static int open_part(const char *path, std::string &part)
{
   struct stat sb;
   if (stat(path, &sb) == 0 && !S_ISREG(sb.st_mode)) { part.clear(); return open(path, O_WRONLY | O_CLOEXEC); }
   char suffix[64];
   snprintf(suffix, sizeof suffix, ".part.%ld.%ld", (long)getpid(), (long)syscall(SYS_gettid));
   part = std::string(path) + suffix;
   const int fd = open(part.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
   if (fd >= 0) return fd;
   part.clear();
   return open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
}
static int finish_part(int fd, bool ok, const std::string &part, const char *path)
{
   if (close(fd) != 0) ok = false;
   if (part.empty()) return ok ? HESAFF_OK : HESAFF_ERR_IO;   // written in place
   if (ok && rename(part.c_str(), path) != 0) ok = false;
   if (!ok) unlink(part.c_str());
   return ok ? HESAFF_OK : HESAFF_ERR_IO;
}

static int write_file(const char *path, const char *buf, size_t len)
{
   std::string part;
   const int fd = open_part(path, part);
   if (fd < 0) return HESAFF_ERR_IO;
   return finish_part(fd, write_all(fd, buf, len), part, path);
}

int hesaff_write_sift(const char *path, const hesaff_keypoint *keys, int n, float mrSize)
{
   return hesaff_write_sift_mt(path, keys, n, mrSize, 0);
}

// threads == 1: the file is formatted and written in blocks of a few thousand rows through one buffer that stays in the
// cache (a UHD image is 40 MB of text: formatting it whole and then copying it into the page cache moves it through
// DRAM three times, block by block once).  Other thread counts: rows on `threads` workers into one buffer, one write.
int hesaff_write_sift_mt(const char *path, const hesaff_keypoint *keys, int n, float mrSize, int threads)
{
   if (!path || n < 0 || (n > 0 && !keys)) return HESAFF_ERR_ARG;
   if (threads != 1) {
      char *buf = nullptr;
      size_t len = 0;
      const int rc = hesaff_format_sift_mt(keys, n, mrSize, threads, &buf, &len);
      if (rc != HESAFF_OK) return rc;
      const int wr = write_file(path, buf, len);
      big_free(buf);
      return wr;
   }
   HOSTIO_TRY
   const int kBlockRows = 4096;
   static thread_local std::vector<char> tl_buf;
   tl_buf.resize(64 + (size_t)kBlockRows * kRowMax);
   char *buf = tl_buf.data();
   std::string part;
   const int fd = open_part(path, part);
   if (fd < 0) return HESAFF_ERR_IO;
   size_t fill = (size_t)snprintf(buf, 64, "%d\n%d\n", 128, n);
   bool ok = true;
   for (int i0 = 0; ok && (i0 < n || fill > 0); i0 += kBlockRows) {
      const int i1 = std::min(n, i0 + kBlockRows);
      char *e = i0 < i1 ? format_rows(keys, i0, i1, mrSize, buf + fill) : buf + fill;
      size_t left = (size_t)(e - buf);
      const char *q = buf;
      while (left > 0) {
         const ssize_t w = write(fd, q, left);
         if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
         q += w; left -= (size_t)w;
      }
      fill = 0;
      if (i1 >= n) break;
   }
   return finish_part(fd, ok, part, path);
   HOSTIO_CATCH
}

// Binary sidecar of the text file: the same five floats and 128 bytes per row, not printed.  Little-endian:
//   char magic[8] = "HESAFFB1"; uint32 dim = 128; uint32 count; count x { float x, y, a, b, c; uint8 desc[128] }  (148 bytes each)
int hesaff_write_bin(const char *path, const hesaff_keypoint *keys, int n, float mrSize)
{
   if (!path || n < 0 || (n > 0 && !keys)) return HESAFF_ERR_ARG;
   HOSTIO_TRY
   const size_t kRec = 5 * 4 + 128, kBlock = 8192;
   static thread_local std::vector<char> tl_bin;
   tl_bin.resize(16 + kBlock * kRec);
   char *buf = tl_bin.data();
   std::string part;
   const int fd = open_part(path, part);
   if (fd < 0) return HESAFF_ERR_IO;
   memcpy(buf, "HESAFFB1", 8);
   const uint32_t dim = 128, cnt = (uint32_t)n;
   memcpy(buf + 8, &dim, 4); memcpy(buf + 12, &cnt, 4);
   size_t fill = 16;
   bool ok = true;
   for (int i0 = 0; ok; i0 += (int)kBlock) {
      const int i1 = std::min(n, i0 + (int)kBlock);
      char *p = buf + fill;
      for (int i = i0; i < i1; i++) {
         float v[5] = {keys[i].x, keys[i].y, 0, 0, 0};
         hesaff_ellipse(&keys[i], mrSize, &v[2], &v[3], &v[4]);
         memcpy(p, v, 20); memcpy(p + 20, keys[i].desc, 128);
         p += kRec;
      }
      size_t left = (size_t)(p - buf);
      const char *q = buf;
      while (left > 0) {
         const ssize_t w = write(fd, q, left);
         if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
         q += w; left -= (size_t)w;
      }
      fill = 0;
      if (i1 >= n) break;
   }
   return finish_part(fd, ok, part, path);
   HOSTIO_CATCH
}

int hesaff_write_sift_rows(const char *path, const char *rows, size_t len, int n)
{
   if (!path || n < 0 || (len > 0 && !rows)) return HESAFF_ERR_ARG;
   HOSTIO_TRY
   std::string part;
   const int fd = open_part(path, part);
   if (fd < 0) return HESAFF_ERR_IO;
   char head[64];
   const int hl = snprintf(head, sizeof head, "%d\n%d\n", 128, n);
   return finish_part(fd, write_head_body(fd, head, (size_t)hl, rows, len), part, path);
   HOSTIO_CATCH
}

// Is `path` the complete output of an earlier run?  text (format HESAFF_OUT_TEXT): "128\n<n>\n", a body of at least n minimal rows, the
// last byte a newline (n == 0: nothing after the header) - three small reads, whatever the size of the file: every writer of this library
// renames its output into place when it is whole, so for its own files the test is exact.  With HESAFF_OUT_STRICT or-ed into `format` the
// newlines of the body are counted too (the whole file is read: about 46 MB per dense 3840 x 2160 image): a text cut at a row boundary by a
// writer that does not rename - the reference binary - is then not complete.  sidecar: magic, dim 128, size == 16 + 148 n.  -> n, or -1.
// What this cannot see: an output made with other parameters (thresholds, fast mode); resume is for re-running the SAME job.
int hesaff_output_is_complete(const char *path, int format)
{
   if (!path) return -1;
   const bool strict = (format & HESAFF_OUT_STRICT) != 0;
   format &= ~HESAFF_OUT_STRICT;
   const int fd = open(path, O_RDONLY | O_CLOEXEC);
   if (fd < 0) return -1;
   int result = -1;
   const off_t size = lseek(fd, 0, SEEK_END);
   char head[64] = {0};
   const ssize_t got = size > 0 ? pread(fd, head, sizeof head - 1, 0) : 0;
   if (format == HESAFF_OUT_BIN) {
      uint32_t dim = 0, cnt = 0;
      if (got >= 16 && memcmp(head, "HESAFFB1", 8) == 0) {
         memcpy(&dim, head + 8, 4); memcpy(&cnt, head + 12, 4);
         if (dim == 128 && cnt <= 0x7fffffffu && (unsigned long long)size == 16ull + 148ull * cnt) result = (int)cnt;
      }
   } else if (got >= 6 && memcmp(head, "128\n", 4) == 0) {
      long n = 0;
      int i = 4;
      while (i < got && head[i] >= '0' && head[i] <= '9' && n < 100000000) n = n * 10 + (head[i++] - '0');
      if (i > 4 && i < got && head[i] == '\n') {
         const off_t body0 = (off_t)(i + 1), body = size - body0;
         char last = 0;
         // a row is at least 5 * 2 + 128 * 2 characters; the text must end with the newline of its last row
         if (n == 0) { if (body == 0) result = 0; }
         else if (body >= (off_t)n * 266 && pread(fd, &last, 1, size - 1) == 1 && last == '\n' && !strict) result = (int)n;
         else if (body >= (off_t)n * 266 && last == '\n') {
            long lines = 0;
            bool ok = true;
            std::vector<char> buf;
            try { buf.resize((size_t)1 << 20); } catch (...) { ok = false; }
            for (off_t at = body0; ok && at < size && lines <= n;) {
               const ssize_t r = pread(fd, buf.data(), buf.size(), at);
               if (r <= 0) { if (r < 0 && errno == EINTR) continue; ok = false; break; }
               for (const char *q = buf.data(), *e = q + r; (q = (const char *)memchr(q, '\n', (size_t)(e - q))) != nullptr; q++) lines++;
               at += r;
            }
            if (ok && lines == n) result = (int)n;
         }
      }
   }
   close(fd);
   return result;
}

// hesaff_write_bin's file from rows that are already packed (148 bytes each)
int hesaff_write_bin_rows(const char *path, const char *rows, int n)
{
   if (!path || n < 0 || (n > 0 && !rows)) return HESAFF_ERR_ARG;
   HOSTIO_TRY
   std::string part;
   const int fd = open_part(path, part);
   if (fd < 0) return HESAFF_ERR_IO;
   char head[16];
   memcpy(head, "HESAFFB1", 8);
   const uint32_t dim = 128, cnt = (uint32_t)n;
   memcpy(head + 8, &dim, 4); memcpy(head + 12, &cnt, 4);
   return finish_part(fd, write_head_body(fd, head, 16, rows, (size_t)n * 148), part, path);
   HOSTIO_CATCH
}

// One file per image of a batch (exportKeypoints once per image, hesaff.cpp:170-176), images
// taken by `threads` workers from a shared counter; every worker formats its image on its own.
int hesaff_write_sift_batch(int n_images, const char *const *paths, const hesaff_result *results, float mrSize, int threads)
{
   if (n_images < 0 || (n_images > 0 && (!paths || !results))) return HESAFF_ERR_ARG;
   int T = threads > 0 ? threads : hesaff_host_threads();
   T = std::max(1, std::min(T, n_images));
   std::atomic<int> err(HESAFF_OK);
   run_tasks(n_images, T, [&](int i) {
      if (!paths[i]) { err = HESAFF_ERR_ARG; return; }
      const int rc = hesaff_write_sift_mt(paths[i], results[i].keys, results[i].count_desc, mrSize, 1);
      if (rc != HESAFF_OK) err = rc;
   });
   return err.load();
}

} // extern "C"
