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
//       --schedule static | dynamic   static (default): one contiguous shard per device context.  dynamic: the list is cut into
//       blocks of 256 images and every device context takes the next block when it has finished its own - for lists whose
//       keypoint density varies strongly along the list; the output files are the same either way.
//       --fast 2                      hesaff_params.fast: windows larger than the patch sampled from the scale space (NOT bit-exact,
//       another algorithm for those keypoints); default 0 = parity mode.
//       --resume | --resume=strict    skip every image whose complete output (of the selected format) already exists (strict: the rows of an
//                                     existing text file are counted too - for outputs a non-renaming writer may have left torn); outputs are written
//       under a temporary name and renamed, so an interrupted run leaves no torn file (checkpoint / resume).
//       --host-share K                this process may use 1/K of the host's CPUs (default 1): one of K processes on a node, e.g. one per GPU.
//       Host threads per device context = hesaff_host_plan_for(devices x K): the library's one rule (include/hesaff_amd.h).
//       On a CPU-starved plan (CPUs of this process <= pool threads + 1, e.g. --host-share 8 on a 16-CPU quota) the threads the HIP runtime creates
//       while the contexts are made - its event thread spins for most of the time the device works - are moved to nice 19: inside a two-CPU share
//       the text path then writes 217 instead of 181 images/s (profiles/r06_notes.md).  --runtime-nice 0 leaves them alone, --runtime-nice 1 forces it.
//       --output text | bin | both    what every image gets: <image>.hesaff.sift (default), the binary sidecar
//       <image>.hesaff.bin (the same rows unprinted, include/hesaff_amd.h: hesaff_write_bin), or both.
#include <chrono>
#include <condition_variable>
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

#include <dirent.h>
#include <set>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "hesaff.hpp"

namespace {

std::set<long> g_tids_at_start;   // the threads this process had before it made its first library call

// the thread ids of this process (/proc/self/task)
std::set<long> list_tids()
{
   std::set<long> out;
   if (DIR *d = opendir("/proc/self/task")) {
      while (struct dirent *e = readdir(d)) {
         char *end = nullptr;
         const long t = strtol(e->d_name, &end, 10);
         if (end != e->d_name && *end == 0 && t > 0) out.insert(t);
      }
      closedir(d);
   }
   return out;
}

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
      if (e == p || a < 0 || a >= ndev) return false;   // checked before anything is expanded ("0-99999999999")
      long b = a;
      p = e;
      if (*p == '-') {
         b = strtol(p + 1, &e, 10);
         if (e == p + 1 || b < a || b >= ndev) return false;
         p = e;
      }
      if (out.size() + (size_t)(b - a + 1) > 4096) return false;
      for (long d = a; d <= b; d++) out.push_back((int)d);
      if (*p == ',') p++;
      else if (*p) return false;
   }
   return !out.empty();
}

// hesaff --batch: the list is cut into contiguous shards, one per device context (hesaff_shard_range); every shard runs
// through hesaff_process_files - decode threads -> device -> writer threads, bounded memory - on its own host thread.
int run_batch_mode(const char *list_path, const char *devices_spec, int out_format, bool dynamic, int fast, int resume, int host_share, int runtime_nice)
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
   const int n = (int)names.size();
   const int world = (int)devices.size();
   std::vector<const char *> paths((size_t)n);
   for (int i = 0; i < n; i++) paths[(size_t)i] = names[(size_t)i].c_str();
   std::vector<hesaff_file_status> status((size_t)n);
   for (auto &st : status) { st.rc = HESAFF_ERR_IO; st.stage = HESAFF_FILE_PENDING; st.count_hessian = 0; st.count_desc = 0; }
   std::vector<std::string> errs((size_t)world);
   int rc = 0;
   const auto t1 = std::chrono::steady_clock::now();
   const int kBlock = 256;
   std::atomic<int> next_block(0);
   // Threads that appear while the contexts are created and that this program did not start are the HIP runtime's helpers (its event
   // thread, which spins while the device works).  On a CPU-starved plan they go to nice 19, below the pool's nice 10: the spinning then
   // yields to the threads that write.  `own`: the device workers' ids, complete before the first context is made (the latch below).
   const std::set<long> &tids_before = g_tids_at_start;   // (taken in main: already hesaff_device_count starts the runtime)
   std::mutex own_mu;
   std::condition_variable own_cv;
   std::set<long> own;
   auto device_worker = [&](int rank) {
      {
         std::unique_lock<std::mutex> lk(own_mu);
         own.insert((long)syscall(SYS_gettid));
         own_cv.notify_all();
         own_cv.wait(lk, [&] { return (int)own.size() >= world; });
      }
      int lo = 0, hi = 0;
      hesaff_shard_range(n, rank, world, &lo, &hi);
      if (!dynamic && hi <= lo) return;
      hesaff_params par;
      hesaff_default_params(&par);
      par.max_batch = std::max(1, std::min(dynamic ? n : hi - lo, 64));
      par.fast = fast;
      hesaff_ctx *ctx = nullptr;
      if (hesaff_create(&ctx, &par, devices[(size_t)rank]) != HESAFF_OK) { errs[(size_t)rank] = hesaff_last_error(nullptr); return; }
      hesaff_set_output_format(ctx, out_format);
      hesaff_set_resume(ctx, resume);
      hesaff_host_plan hp;   // this device's share of the host: the library's one rule (include/hesaff_amd.h)
      hesaff_host_plan_for(world * host_share, &hp);
      const int wt = hp.write_threads, dt = hp.decode_threads;
      if (runtime_nice == 1 || (runtime_nice < 0 && hp.cpus <= dt + wt + 1)) {   // this device's share of the host is no larger than its pool + the caller
         {
            // one tiny batch through the context first: the runtime starts its helper threads with the first launches and events, not with the context
            std::vector<uint8_t> blank((size_t)64 * 64, 0);
            const uint8_t *img = blank.data();
            const int side = 64, chn = 1;
            hesaff_result res;
            (void)hesaff_detect_batch(ctx, 1, &img, &side, &side, &side, &chn, &res);
         }
         std::lock_guard<std::mutex> lk(own_mu);
         int moved = 0;
         for (long t : list_tids())
            if (!tids_before.count(t) && !own.count(t) && setpriority(PRIO_PROCESS, (id_t)t, 19) == 0) moved++;   // (a refusal changes nothing)
         if (runtime_nice == 1) fprintf(stderr, "hesaff: device %d: %d runtime helper thread(s) at nice 19\n", devices[(size_t)rank], moved);
      }
      for (;;) {
         if (dynamic) {   // the next block of the list nobody has taken yet
            lo = next_block.fetch_add(1) * kBlock;
            hi = std::min(n, lo + kBlock);
         }
         if (hi <= lo) break;
         if (hesaff_process_files(ctx, hi - lo, paths.data() + lo, nullptr, dt, wt, status.data() + lo) != HESAFF_OK) { errs[(size_t)rank] = hesaff_last_error(ctx); break; }
         if (!dynamic) break;
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
   int n_ok = 0, n_skipped = 0;
   for (int r = 0; r < world; r++)
      if (!errs[(size_t)r].empty()) { fprintf(stderr, "hesaff: device %d: %s\n", devices[(size_t)r], errs[(size_t)r].c_str()); rc = 1; }
   for (int i = 0; i < n; i++) {
      const hesaff_file_status &st = status[(size_t)i];
      if (st.rc == HESAFF_OK && st.stage == HESAFF_FILE_SKIPPED) {
         std::cout << names[(size_t)i] << ": output exists (" << st.count_desc << " rows), skipped" << std::endl;
         n_skipped++;
      } else if (st.rc == HESAFF_OK) {
         std::cout << names[(size_t)i] << ": Detected " << st.count_hessian << " keypoints and " << st.count_desc << " affine shapes" << std::endl;
         tot_h += st.count_hessian; tot_d += st.count_desc; n_ok++;
      } else {
         rc = 1;
         if (st.stage == HESAFF_FILE_DETECTED) fprintf(stderr, "hesaff: cannot write the output of '%s'\n", names[(size_t)i].c_str());
         else if (st.stage == HESAFF_FILE_UNREADABLE)
            fprintf(stderr, "hesaff: cannot read '%s' (PBM / PGM / PPM, PNG, JPEG, BMP or baseline TIFF expected): skipped\n", names[(size_t)i].c_str());
      }
   }
   std::cout << "Detected " << tot_h << " keypoints and " << tot_d << " affine shapes in " << n_ok << " images in " << dt << " sec.";
   if (n_skipped) std::cout << " (" << n_skipped << " more skipped: complete outputs exist)";
   if (world > 1) std::cout << " (" << world << " device contexts)";
   std::cout << std::endl;
   return rc;
}

} // namespace

int main(int argc, char **argv)
{
   g_tids_at_start = list_tids();
   if (hesaff_abi_version() != HESAFF_ABI_VERSION || hesaff_sizeof_params() != sizeof(hesaff_params)) {
      fprintf(stderr, "hesaff: libhesaff_amd.so has ABI version %d, this program was built for %d\n", hesaff_abi_version(), HESAFF_ABI_VERSION);
      return 1;
   }
   // batch mode: "--batch <list>" and the other options in any order (the reference has no options: a first argument that does
   // not start with "--" is an image, as in hesaff.cpp:133-137)
   bool batch = false;
   for (int i = 1; i < argc; i++) batch = batch || strcmp(argv[i], "--batch") == 0;
   if (batch) {
      const char *devices = nullptr, *list = nullptr;
      int out_format = HESAFF_OUT_TEXT, fast = 0, host_share = 1, runtime_nice = -1;
      bool bad = false, dynamic = false;
      int resume = 0;
      for (int i = 1; i < argc && !bad; i += 2) {
         if (strcmp(argv[i], "--resume") == 0) { resume = 1; i--; continue; }
         if (strcmp(argv[i], "--resume=strict") == 0) { resume = 2; i--; continue; }   // also count the rows of existing text outputs
         if (i + 1 >= argc) bad = true;
         else if (strcmp(argv[i], "--batch") == 0) list = argv[i + 1];
         else if (strcmp(argv[i], "--devices") == 0) devices = argv[i + 1];
         else if (strcmp(argv[i], "--host-share") == 0) { host_share = atoi(argv[i + 1]); bad = host_share < 1 || host_share > 1024; }
         else if (strcmp(argv[i], "--runtime-nice") == 0) { runtime_nice = atoi(argv[i + 1]); bad = runtime_nice < 0 || runtime_nice > 1; }
         else if (strcmp(argv[i], "--fast") == 0) {
            if (strcmp(argv[i + 1], "0") == 0 || strcmp(argv[i + 1], "2") == 0) fast = atoi(argv[i + 1]);
            else bad = true;
         } else if (strcmp(argv[i], "--schedule") == 0) {
            if (strcmp(argv[i + 1], "dynamic") == 0) dynamic = true;
            else if (strcmp(argv[i + 1], "static") != 0) bad = true;
         } else if (strcmp(argv[i], "--output") == 0) {
            if (strcmp(argv[i + 1], "text") == 0) out_format = HESAFF_OUT_TEXT;
            else if (strcmp(argv[i + 1], "bin") == 0) out_format = HESAFF_OUT_BIN;
            else if (strcmp(argv[i + 1], "both") == 0) out_format = HESAFF_OUT_TEXT | HESAFF_OUT_BIN;
            else bad = true;
         } else bad = true;
      }
      if (bad || !list) { fprintf(stderr, "hesaff: usage: hesaff --batch <list file> [--devices 0-7|0,2|all] [--output text|bin|both] [--schedule static|dynamic] [--fast 0|2] [--resume|--resume=strict] [--host-share K] [--runtime-nice 0|1]\n"); return 1; }
      return run_batch_mode(list, devices, out_format, dynamic, fast, resume, host_share, runtime_nice);
   }
   if (argc > 1) {
      uint8_t *data = nullptr;
      int w = 0, h = 0, ch = 0;
      if (hesaff_read_image(argv[1], &data, &w, &h, &ch) != HESAFF_OK) {
         fprintf(stderr, "hesaff: cannot read '%s' (PBM / PGM / PPM, PNG, JPEG, BMP or baseline TIFF expected)\n", argv[1]);
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
