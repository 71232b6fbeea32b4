// hesaff_cli.cpp -- `hesaff <image>` : same command line, stdout line and output file as
// the reference's main() (hesaff.cpp:133-180); the work runs on the MI355X through
// libhesaff_amd.so.  Input: binary PGM/PPM (P5/P6) or PNG.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "hesaff.hpp"

// `hesaff --batch <list file>` (extension): one image path per line; all
// images go through one hesaff_detect_batch call (grouped by size inside the library) and one
// hesaff_write_sift_batch call; every image gets the same `<image>.hesaff.sift` the single-image
// form writes.  The reference has no such mode (it would try to open a file called "--batch").
static int run_batch_mode(const char *list_path)
{
   std::ifstream lf(list_path);
   if (!lf) { fprintf(stderr, "hesaff: cannot read list '%s'\n", list_path); return 1; }
   std::vector<std::string> names;
   for (std::string line; std::getline(lf, line);) {
      while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
      if (!line.empty() && line[0] != '#') names.push_back(line);
   }
   const int n = (int)names.size();
   std::vector<uint8_t *> data((size_t)n, nullptr);
   std::vector<int> w((size_t)n), h((size_t)n), ch((size_t)n), stride((size_t)n);
   int rc = 0;
   // decode on a few host threads (the files are independent)
   {
      std::atomic<int> next(0), bad(-1);
      auto work = [&] {
         for (int i; (i = next.fetch_add(1)) < n;)
            if (hesaff_read_image(names[i].c_str(), &data[i], &w[i], &h[i], &ch[i]) != HESAFF_OK) bad = i;
            else stride[i] = w[i] * ch[i];
      };
      std::vector<std::thread> th;
      const int T = std::max(1, std::min<int>(n, std::min<unsigned>(std::thread::hardware_concurrency(), 16u)));
      for (int t = 1; t < T; t++) th.emplace_back(work);
      work();
      for (auto &x : th) x.join();
      if (bad.load() >= 0) {
         fprintf(stderr, "hesaff: cannot read '%s' (binary PGM/PPM with maxval 255 or PNG expected)\n", names[bad.load()].c_str());
         rc = 1;
      }
   }
   hesaff_ctx *ctx = nullptr;
   if (rc == 0) {
      hesaff_params par;
      hesaff_default_params(&par);
      par.max_batch = std::max(1, std::min(n, 64));
      if (hesaff_create(&ctx, &par, 0) != HESAFF_OK) { fprintf(stderr, "hesaff: %s\n", hesaff_last_error(nullptr)); rc = 1; }
      else {
         std::vector<hesaff_result> res((size_t)n);
         const auto t1 = std::chrono::steady_clock::now();
         if (hesaff_detect_batch(ctx, n, data.data(), w.data(), h.data(), stride.data(), ch.data(), res.data()) != HESAFF_OK) {
            fprintf(stderr, "hesaff: %s\n", hesaff_last_error(ctx));
            rc = 1;
         } else {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            long long nh = 0, nd = 0;
            std::vector<std::string> outs((size_t)n);
            std::vector<const char *> outp((size_t)n);
            for (int i = 0; i < n; i++) {
               std::cout << names[i] << ": Detected " << res[i].count_hessian << " keypoints and " << res[i].count_desc << " affine shapes" << std::endl;
               nh += res[i].count_hessian; nd += res[i].count_desc;
               outs[i] = names[i] + ".hesaff.sift";
               outp[i] = outs[i].c_str();
            }
            std::cout << "Detected " << nh << " keypoints and " << nd << " affine shapes in " << n << " images in " << dt << " sec." << std::endl;
            if (hesaff_write_sift_batch(n, outp.data(), res.data(), par.mrSize, 0) != HESAFF_OK) { fprintf(stderr, "hesaff: cannot write the output files\n"); rc = 1; }
         }
         hesaff_destroy(ctx);
      }
   }
   for (uint8_t *d : data) hesaff_free(d);
   return rc;
}

int main(int argc, char **argv)
{
   if (argc > 2 && strcmp(argv[1], "--batch") == 0) return run_batch_mode(argv[2]);
   if (argc > 1) {
      uint8_t *data = nullptr;
      int w = 0, h = 0, ch = 0;
      if (hesaff_read_image(argv[1], &data, &w, &h, &ch) != HESAFF_OK) {
         fprintf(stderr, "hesaff: cannot read '%s' (binary PGM/PPM with maxval 255 or PNG expected)\n", argv[1]);
         return 1;
      }
      try {
         hesaff_amd::HessianAffineParams par;
         hesaff_amd::AffineHessianDetector detector(par);
         const auto t1 = std::chrono::steady_clock::now();
         detector.detectPyramidKeypoints(data, w, h, ch);
         const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
         std::cout << "Detected " << detector.g_numberOfPoints << " keypoints and " << detector.g_numberOfAffinePoints
                   << " affine shapes in " << dt << " sec." << std::endl;
         const std::string name = std::string(argv[1]) + ".hesaff.sift";
         std::ofstream out(name.c_str());
         if (!out) { fprintf(stderr, "hesaff: cannot write '%s'\n", name.c_str()); hesaff_free(data); return 1; }
         detector.exportKeypoints(out);
      } catch (const std::exception &e) {
         fprintf(stderr, "hesaff: %s\n", e.what());
         hesaff_free(data);
         return 1;
      }
      hesaff_free(data);
   } else {
      printf("\nUsage: hesaff image_name.ppm\nDetects Hessian Affine points and describes them using SIFT descriptor.\nThe detector assumes that the vertical orientation is preserved.\n\n");
   }
   return 0;
}
