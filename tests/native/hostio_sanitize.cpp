// Sanitizer driver for the host side of the C ABI (test infrastructure; built by tests/test_host_sanitize.py with
// g++ -fsanitize=address,undefined from hesaff_amd/csrc/hostio.cpp + jpeg_decode.cpp).
//   hostio_sanitize read <file>...     every file through hesaff_read_image (any return code is fine: the point is
//                                      that a damaged PNM / PNG / JPEG is refused without touching memory it does not own) and through
//                                      hesaff_read_jpeg_coefficients (+ _alloc with a dirty recycled blob: same bytes)
//   hostio_sanitize format <seed> <n>  n keypoints of random bit patterns (NaN, infinities, denormals included) through
//                                      hesaff_ellipse / hesaff_format_sift / hesaff_format_sift_mt; both texts must agree
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/hesaff_amd.h"

static uint64_t rng_state;
static uint32_t rnd()
{
   rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
   return (uint32_t)(rng_state >> 32);
}

int main(int argc, char **argv)
{
   if (argc < 2) return 2;
   if (!strcmp(argv[1], "read")) {
      int ok = 0, bad = 0, coef = 0;
      for (int i = 2; i < argc; i++) {
         uint8_t *data = nullptr;
         int w = 0, h = 0, ch = 0;
         const int rc = hesaff_read_image(argv[i], &data, &w, &h, &ch);
         if (rc == HESAFF_OK) {
            // touch every byte the reader says it produced
            unsigned sum = 0;
            for (size_t k = 0; k < (size_t)w * h * ch; k++) sum += data[k];
            if (sum == 0xffffffffu) puts("");
            hesaff_free(data);
            ok++;
         } else {
            if (data) return 3;   // an error must not hand out a buffer
            bad++;
         }
         // the entropy-only half of the JPEG reader (what hesaff_process_files runs): fresh memory, then a recycled blob full of another
         // image's bytes - the two blobs must be identical, and every byte the layout promises must be readable
         hesaff_jpeg_layout L1, L2;
         uint8_t *b1 = nullptr, *b2 = nullptr;
         size_t n1 = 0, n2 = 0;
         const int rc1 = hesaff_read_jpeg_coefficients(argv[i], &L1, &b1, &n1);
         if (rc1 != HESAFF_OK) { if (b1) return 4; continue; }
         size_t blocks = 0;
         for (int c = 0; c < L1.channels; c++) blocks += (size_t)L1.bw[c] * L1.bh[c];
         if (n1 != HESAFF_JPEG_BLOB_HEADER + blocks * 128) return 5;
         unsigned sum = 0;
         for (size_t k = 0; k < n1; k++) sum += b1[k];
         if (sum == 0xffffffffu) puts("");
         struct Dirty { size_t bytes; } dirty = {n1};
         auto alloc = [](size_t bytes, int *zeroed, void *user) -> void * {
            (void)user;
            void *p = malloc(bytes);
            if (p) memset(p, 0xA5, bytes);
            *zeroed = 0;
            return p;
         };
         const int rc2 = hesaff_read_jpeg_coefficients_alloc(argv[i], &L2, &b2, &n2, alloc, &dirty);
         if (rc2 != HESAFF_OK || n2 != n1 || memcmp(&L1, &L2, sizeof L1) != 0 || memcmp(b1, b2, n1) != 0) return 6;
         hesaff_free(b1); hesaff_free(b2);
         coef++;
      }
      printf("read ok=%d refused=%d coefficient blobs=%d\n", ok, bad, coef);
      return 0;
   }
   if (!strcmp(argv[1], "format") && argc >= 4) {
      rng_state = strtoull(argv[2], nullptr, 10);
      const int n = atoi(argv[3]);
      std::vector<hesaff_keypoint> keys((size_t)n);
      for (auto &k : keys) {
         uint32_t *w = reinterpret_cast<uint32_t *>(&k);
         for (size_t j = 0; j < sizeof(hesaff_keypoint) / 4; j++) {
            const uint32_t r = rnd();
            // a third of the words are raw bit patterns (NaN / inf / denormals among them), the rest ordinary floats
            if (r % 3 == 0) w[j] = rnd();
            else { float f = (float)(int32_t)rnd() / 65536.0f; memcpy(&w[j], &f, 4); }
         }
      }
      for (auto &k : keys) { float a, b, c; hesaff_ellipse(&k, 5.1961524f, &a, &b, &c); }
      char *t1 = nullptr, *t2 = nullptr;
      size_t l1 = 0, l2 = 0;
      if (hesaff_format_sift(keys.data(), n, 5.1961524f, &t1, &l1) != HESAFF_OK) return 4;
      if (hesaff_format_sift_mt(keys.data(), n, 5.1961524f, 7, &t2, &l2) != HESAFF_OK) return 5;
      const bool same = l1 == l2 && memcmp(t1, t2, l1) == 0;
      hesaff_free(t1);
      hesaff_free(t2);
      printf("format n=%d bytes=%zu same=%d\n", n, l1, (int)same);
      return same ? 0 : 6;
   }
   return 2;
}
