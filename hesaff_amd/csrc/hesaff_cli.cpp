// hesaff_cli.cpp -- `hesaff <image>` : same command line, stdout line and output file as
// the reference's main() (hesaff.cpp:133-180); the work runs on the MI355X through
// libhesaff_amd.so.  Input: binary PGM/PPM (P5/P6), PNG or JPEG.
//
// Extensions (the reference has no flags; it would try to open a file called "--batch"):
//   hesaff --batch <list file> [--devices <spec>]
//       one image path per line; every image gets the `<image>.hesaff.sift` the single-image form writes.
//       --devices 0-7 | 0,2,5 | all   shards the list over several GPUs of the node: one context per device, each on
//       its own host thread, contiguous blocks of images (hesaff_shard_range), no data exchanged between devices;
//       the per-device counts are summed on the host.  A device may be named more than once.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hesaff.hpp"

namespace {

struct Image {
   uint8_t *data = nullptr;
   int w = 0, h = 0, ch = 0;
   bool ok = false;
};

// "0-7", "0,2,5", "1", "all" -> device ordinals
bool parse_devices(const char *spec, std::vector<int> &out)
{
   out.clear();
   const int ndev = hesaff_device_count();
   if (strcmp(spec, "all") == 0) {
      for (int i = 0; i < ndev; i++) out.push_back(i);
      return !out.empty();
   }
   const char *p = spec;
   while (*p) {
      char *e = nullptr;
      const long a = strtol(p, &e, 10);
      if (e == p || a < 0) return false;
      long b = a;
      p = e;
      if (*p == '-') {
         b = strtol(p + 1, &e, 10);
         if (e == p + 1 || b < a) return false;
         p = e;
      }
      for (long d = a; d <= b; d++) out.push_back((int)d);
      if (*p == ',') p++;
      else if (*p) return false;
   }
   for (int d : out)
      if (d >= ndev) return false;
   return !out.empty();
}

int run_batch_mode(const char *list_path, const char *devices_spec)
{
   std::ifstream lf(list_path);
   if (!lf) { fprintf(stderr, "hesaff: cannot read list '%s'\n", list_path); return 1; }
   std::vector<std::string> names;
   for (std::string line; std::getline(lf, line);) {
      while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
      if (!line.empty() && line[0] != '#') names.push_back(line);
   }
   std::vector<int> devices;
   if (!parse_devices(devices_spec ? devices_spec : "0", devices)) {
      fprintf(stderr, "hesaff: bad --devices '%s' (%d device(s) visible)\n", devices_spec ? devices_spec : "0", hesaff_device_count());
      return 1;
   }
   const int n_all = (int)names.size();
   std::vector<Image> imgs((size_t)n_all);
   int rc = 0;
   // decode on a few host threads (the files are independent); an unreadable file is reported and skipped,
   // the other images of the list are still processed (exit code 1 at the end)
   {
      std::atomic<int> next(0);
      auto work = [&] {
         for (int i; (i = next.fetch_add(1)) < n_all;)
            imgs[i].ok = hesaff_read_image(names[i].c_str(), &imgs[i].data, &imgs[i].w, &imgs[i].h, &imgs[i].ch) == HESAFF_OK;
      };
      std::vector<std::thread> th;
      const int T = std::max(1, std::min<int>(n_all, std::min(hesaff_host_threads(), 16)));
      for (int t = 1; t < T; t++) th.emplace_back(work);
      work();
      for (auto &x : th) x.join();
   }
   std::vector<int> good;
   for (int i = 0; i < n_all; i++) {
      if (imgs[i].ok) good.push_back(i);
      else {
         fprintf(stderr, "hesaff: cannot read '%s' (binary PGM/PPM with maxval 255, PNG or JPEG expected): skipped\n", names[i].c_str());
         rc = 1;
      }
   }
   const int n = (int)good.size();
   const int world = (int)devices.size();
   std::vector<hesaff_result> res((size_t)n);
   std::vector<long long> nh((size_t)world, 0), nd((size_t)world, 0);
   std::vector<std::string> errs((size_t)world);
   std::mutex out_mutex;
   float mrSize = 0;
   {
      hesaff_params par;
      hesaff_default_params(&par);
      mrSize = par.mrSize;
   }
   const auto t1 = std::chrono::steady_clock::now();
   // one context per device, each driven by its own host thread over its contiguous shard of the images
   auto device_worker = [&](int rank) {
      int lo = 0, hi = 0;
      hesaff_shard_range(n, rank, world, &lo, &hi);
      if (hi <= lo) return;
      const int m = hi - lo;
      hesaff_params par;
      hesaff_default_params(&par);
      par.max_batch = std::max(1, std::min(m, 64));
      hesaff_ctx *ctx = nullptr;
      if (hesaff_create(&ctx, &par, devices[rank]) != HESAFF_OK) { errs[rank] = hesaff_last_error(nullptr); return; }
      std::vector<const uint8_t *> data((size_t)m);
      std::vector<int> w((size_t)m), h((size_t)m), ch((size_t)m), stride((size_t)m);
      for (int k = 0; k < m; k++) {
         const Image &im = imgs[good[lo + k]];
         data[k] = im.data; w[k] = im.w; h[k] = im.h; ch[k] = im.ch; stride[k] = im.w * im.ch;
      }
      if (hesaff_detect_batch(ctx, m, data.data(), w.data(), h.data(), stride.data(), ch.data(), res.data() + lo) != HESAFF_OK) {
         errs[rank] = hesaff_last_error(ctx);
         hesaff_destroy(ctx);
         return;
      }
      std::vector<std::string> outs((size_t)m);
      std::vector<const char *> outp((size_t)m);
      for (int k = 0; k < m; k++) {
         nh[rank] += res[lo + k].count_hessian;
         nd[rank] += res[lo + k].count_desc;
         outs[k] = names[good[lo + k]] + ".hesaff.sift";
         outp[k] = outs[k].c_str();
      }
      // the result records live in the context's pinned memory: write this shard's files before the context goes away
      const int T = std::max(1, hesaff_host_threads() / world);
      if (hesaff_write_sift_batch(m, outp.data(), res.data() + lo, mrSize, T) != HESAFF_OK) errs[rank] = "cannot write the output files";
      {
         std::lock_guard<std::mutex> g(out_mutex);
         for (int k = 0; k < m; k++)
            std::cout << names[good[lo + k]] << ": Detected " << res[lo + k].count_hessian << " keypoints and " << res[lo + k].count_desc
                      << " affine shapes" << std::endl;
      }
      hesaff_destroy(ctx);
   };
   {
      std::vector<std::thread> th;
      for (int r = 1; r < world; r++) th.emplace_back(device_worker, r);
      device_worker(0);
      for (auto &x : th) x.join();
   }
   const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
   long long tot_h = 0, tot_d = 0;
   for (int r = 0; r < world; r++) {
      tot_h += nh[r]; tot_d += nd[r];
      if (!errs[r].empty()) { fprintf(stderr, "hesaff: device %d: %s\n", devices[r], errs[r].c_str()); rc = 1; }
   }
   std::cout << "Detected " << tot_h << " keypoints and " << tot_d << " affine shapes in " << n << " images in " << dt << " sec.";
   if (world > 1) std::cout << " (" << world << " device contexts)";
   std::cout << std::endl;
   for (Image &im : imgs) hesaff_free(im.data);
   return rc;
}

} // namespace

int main(int argc, char **argv)
{
   if (argc > 2 && strcmp(argv[1], "--batch") == 0) {
      const char *devices = nullptr;
      if (argc > 4 && strcmp(argv[3], "--devices") == 0) devices = argv[4];
      else if (argc > 3) { fprintf(stderr, "hesaff: usage: hesaff --batch <list file> [--devices 0-7|0,2|all]\n"); return 1; }
      return run_batch_mode(argv[2], devices);
   }
   if (argc > 1) {
      uint8_t *data = nullptr;
      int w = 0, h = 0, ch = 0;
      if (hesaff_read_image(argv[1], &data, &w, &h, &ch) != HESAFF_OK) {
         fprintf(stderr, "hesaff: cannot read '%s' (binary PGM/PPM with maxval 255, PNG or JPEG expected)\n", argv[1]);
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
