// pipeline.hip -- host side of libhesaff_amd.so: context, HBM buffer plan, kernel
// orchestration of one batch, and the C ABI of include/hesaff_amd.h.
//
// The reference chains its stages depth-first through virtual callbacks, one keypoint at
// a time (hesaff.cpp:66-105).  Here a batch of B equally sized images runs breadth-first:
//   pyramid (per octave: R0, 4x blur+response, decimate)  ->  extrema + localise per octave
//   -> order by bitmask rank -> affine iteration -> rectify/bin -> patch -> SIFT -> pack.
// The reference's output order (octave, level, raster of the initial extremum) is
// reproduced by ranking survivors through a bitmask laid out in exactly that order.
//
// Nothing here reads the environment unless the library is built with -DHESAFF_TUNING
// (`make tuning`, a second .so for A/B measurements): a drop-in library must not change its
// schedule, let alone its results, because of an environment variable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <condition_variable>
#include <deque>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/hesaff_amd.h"
#include "host_tables.h"
#include "kernels_keypoint.h"
#include "kernels_patch.h"
#include "kernels_sift.h"
#include "kernels_pyramid.h"
#include "kernels_export.h"
#include "kernels_jpeg.h"
#include "chunk_engine.h"

namespace hesaff {
OctaveSchedule make_schedule(float initialSigma, bool upscale)
{
   OctaveSchedule s;
   // pyramid.cpp:227 : powf(2, 1/numberOfScales) ; hm_pow2f == glibc powf(2,.) bit for bit
   const float sigmaStep = hm_pow2f(1.0f / (float)HS_NSCALES);
   float curSigma = initialSigma;
   // pyramid.cpp:263-280: the input is taken to be blurred by 0.5 already (1.0 after the 2x up-sampling);
   // no initial blur when initialSigma does not exceed that
   const float inputSigma = upscale ? 0.5f * 2.0f : 0.5f;
   s.init_sigma = initialSigma > inputSigma ? sqrtf(initialSigma * initialSigma - inputSigma * inputSigma) : 0.0f;
   s.level_sigma[0] = curSigma;
   s.blur_sigma[0] = 0.0f;
   {
      const float n = curSigma * curSigma;
      s.norm2[0] = n * n;
   }
   for (int i = 1; i < HS_NSCALES + 2; i++) {
      s.blur_sigma[i] = curSigma * sqrtf(sigmaStep * sigmaStep - 1.0f);
      const float sigma = curSigma * sigmaStep;
      s.level_sigma[i] = sigma;
      const float n = sigma * sigma;
      s.norm2[i] = n * n;
      curSigma *= sigmaStep;
   }
   return s;
}
} // namespace hesaff

static thread_local std::string g_create_error;

#define HIP_TRY(expr)                                                                         \
   do {                                                                                       \
      hipError_t e_ = (expr);                                                                 \
      if (e_ != hipSuccess) {                                                                 \
         char buf_[512];                                                                      \
         snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
         throw HsError(HESAFF_ERR_DEVICE, buf_);                                              \
      }                                                                                       \
   } while (0)

struct HsError {
   int code;
   std::string msg;
   HsError(int c, const std::string &m) : code(c), msg(m) {}
};

// Host wait for a HIP event WITHOUT a spinning core.  hipEventSynchronize spins in this runtime even on events created with
// hipEventBlockingSync (measured in round 5 with CLOCK_THREAD_CPUTIME_ID around the call: 96 ms of CPU for a 96 ms wait, one busy core per
// context while a chunk's kernels run - half the CPU quota of an 8-GPU node).  Poll hipEventQuery instead: a few immediate queries for
// waits of microseconds, then sleeps of 20 .. 200 us.
static void hs_wait_event(hipEvent_t ev)
{
   timespec t0;
   clock_gettime(CLOCK_MONOTONIC, &t0);
   for (int tries = 0;; tries++) {
      const hipError_t e = hipEventQuery(ev);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) throw HsError(HESAFF_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(e));
      (void)hipGetLastError();   // hipErrorNotReady is not an error
      if (tries < 16) continue;
      // sleep an eighth of what has been waited so far, 20 .. 200 us: the wake-up is late by at most ~6 % of a short wait, 0.2 ms of a long one
      timespec now;
      clock_gettime(CLOCK_MONOTONIC, &now);
      const long long waited = (long long)(now.tv_sec - t0.tv_sec) * 1000000000ll + (now.tv_nsec - t0.tv_nsec);
#ifdef HESAFF_TUNING
      static const long long nap_cap = getenv("HESAFF_NAP_US") ? atoll(getenv("HESAFF_NAP_US")) * 1000ll : 200000ll;
#else
      const long long nap_cap = 200000ll;
#endif
      const timespec nap = {0, (long)std::min<long long>(std::max<long long>(waited / 8, 20000), nap_cap)};
      nanosleep(&nap, nullptr);
   }
}

static double thread_cpu_ms()   // CPU time of the calling thread (debug lines: does the caller sleep while the device works?)
{
   timespec ts;
   clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
   return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

// A device buffer that only grows.  ensure() never leaves a dangling pointer behind: the new block
// is allocated before the old one is released (when the device cannot hold both, the old block
// is released first and the allocation retried); on failure the buffer is empty (p == nullptr,
// bytes == 0) and HESAFF_ERR_NOMEM is thrown.
struct DevBuf {
   void *p = nullptr;
   size_t bytes = 0;
   void ensure(size_t need)
   {
      if (need <= bytes) return;
      void *q = nullptr;
      hipError_t e = hipMalloc(&q, need);
      if (e != hipSuccess && p) {
         (void)hipGetLastError();
         (void)hipFree(p);
         p = nullptr; bytes = 0;
         e = hipMalloc(&q, need);
      }
      if (e != hipSuccess) {
         (void)hipGetLastError();
         if (p) (void)hipFree(p);
         p = nullptr; bytes = 0;
         char buf[256];
         snprintf(buf, sizeof buf, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e));
         throw HsError(HESAFF_ERR_NOMEM, buf);
      }
      if (p) (void)hipFree(p);
      p = q;
      bytes = need;
   }
   // for buffers whose size follows the data (keypoints of a group, rows of a chunk): a new maximum is allocated with head-room, so
   // that hipFree + hipMalloc - tens of ms for the group buffers, with the device idle - become rare instead of "every denser chunk"
   void ensure_grow(size_t need)
   {
      if (need <= bytes) return;
      try { ensure(need + need / 8); }
      catch (const HsError &) { ensure(need); }
   }
   void release()
   {
      if (p) (void)hipFree(p);
      p = nullptr;
      bytes = 0;
   }
   template <class T> T *as() const { return (T *)p; }
};

struct OctGeom {
   int rows, cols, pitch;
   long long word_base;   // first bitmask word of this octave inside one image
   int words_per_row;
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

#define HS_NSIDE 4   // side streams of the patch stage (one per window-size bin 0..3)
// grids of the grid-stride list kernels (blocks of 256 threads)
#ifndef HS_GRID_LOC
#define HS_GRID_LOC 1024
#endif
#ifndef HS_GRID_DED
#define HS_GRID_DED 512
#endif
#ifndef HS_GRID_SCAT
#define HS_GRID_SCAT 1024
#endif
#ifndef HS_GRID_PACK
#define HS_GRID_PACK 3584   // 14 blocks per CU: what k_pack's 10.7 KB of LDS per block lets a CU hold
#endif
#ifndef HS_MID_CAP
#define HS_MID_CAP 6   // blocks per CU of the two row-streamed bins (at most; the occupancy query may say fewer)
#endif
#ifndef HS_BIG_CAP
#define HS_BIG_CAP 4
#endif
#ifndef HS_OVERSUB
#define HS_OVERSUB 128u   // oversubscription of the statically strided persistent grids (step at B = 128: x 1 / 32 / 128 / 256 / 2048: 443 / 438 / 433 / 434 / 447 ms)
#endif
#ifndef HS_NSLOT
#define HS_NSLOT 3   // patch / descriptor buffer slots of the group pipeline
#endif
// The HIP streams a context runs on (four compute streams, two copy streams).  They are created once per device and handed from a
// destroyed context to the next one (capi_impl.h): which hardware queues a NEW stream shares depends on everything the process has
// created before, so only reuse keeps the queue pairing of the first context.  Contexts alive at the same time get sets of their own.
struct StreamSet { hipStream_t comp[4] = {nullptr, nullptr, nullptr, nullptr}; hipStream_t h2d = nullptr, d2h = nullptr; };
static std::mutex g_sets_mu;
static std::map<int, std::vector<StreamSet>> g_idle_sets;   // per device: the sets no context is using
static bool take_stream_set(int device, StreamSet &out)
{
   std::lock_guard<std::mutex> lk(g_sets_mu);
   std::vector<StreamSet> &v = g_idle_sets[device];
   if (v.empty()) return false;
   out = v.back();
   v.pop_back();
   return true;
}
static void give_stream_set(int device, const StreamSet &s)
{
   std::lock_guard<std::mutex> lk(g_sets_mu);
   g_idle_sets[device].push_back(s);
}

struct hesaff_ctx {
   hesaff_params par;
   int device = 0;
   StreamSet sset;
   bool pooled_streams = false;
   hipStream_t stream = nullptr;
   std::string err;
   hesaff::OctaveSchedule sched;
   DConsts consts;

   // tables
   DevBuf t_smm, t_sift, t_bin0, t_bin1, t_w0, t_w1, t_pyr_taps, t_patch_taps, t_patch_off, t_patch_k;
   int pyr_K[5];          // [0] initial blur (0 = none), [1..4] octave blurs
   int pyr_tap_off[5];
   bool pyr_march = false;   // the four octave blurs have K = 9, 11, 13, 15 (default initialSigma): marching kernel
   int max_p0 = 0;        // tap table covers odd P0 <= max_p0
   int batch_max_p = 0;   // largest window side P of the current batch's huge windows (known after detection)
   int n_masked = 0;
   KpTables tables;

   // geometry of the current buffer plan (H x W: the images; the pyramid starts at (H << up) x (W << up))
   int up = 0;            // upscaleInputImage, pyramid.h:34
   int B = 0, H = 0, W = 0;
   std::vector<OctGeom> oct;
   long long words_per_image = 0;
   uint32_t cap = 0;      // keypoint capacity of a batch
   uint32_t cand_cap = 0; // candidate slots of one octave (k_extrema_march -> k_localize)
   bool map_clean = false;   // b_map holds nothing but 0xFFFFFFFF and bids of epochs above map_epoch (OctaveCtx::map_epoch, kernels_pyramid.h)
   int map_kbits = 32;       // bits of an order key (3 x pixels of the first pyramid level)
   uint32_t map_epoch = 0;   // the last epoch handed out; 0: the next pass refills the map first

   // planes
   DevBuf b_gray, b_up, b_L, b_L3, b_R, b_map, b_bitmask, b_prefix, b_blocksums, b_generic;
   std::vector<DPlane> L;   // [octave*3 + level]
   DPlane gray, upimg, L3, R[5];
   // lists
   DevBuf b_counters;       // uint32: [0] cand_count [1] rec_count [2] overflow [3] hess_total [4] desc_total [5] group end
                            //         [6] T' row overflow, [8..12] bin_count, [24..28] bin work counters,
                            //         [32..32+HS_MAX_OCTAVES) octave rec starts
   DevBuf b_cand, b_rec_f, b_rec_i, b_rec_w, b_hess_f, b_hess_i, b_aff, b_pw, b_bins, b_rank, b_desc, b_out, b_starts, b_patches, b_stage;
   DevBuf b_input;          // staging for host images (stage API)
   // hesaff_detect_batch, host entry point: chunks of max_batch images are pipelined -- pinned
   // staging + H2D of chunk i+1 and D2H of chunk i-1 run beside the kernels of chunk i
   DevBuf b_in2[2], b_outstage[2];
   struct Pinned {
      void *p = nullptr;
      size_t bytes = 0;
      void ensure(size_t need)
      {
         if (need <= bytes) return;
         if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
         hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
         if (e != hipSuccess) { p = nullptr; throw HsError(HESAFF_ERR_NOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(e)); }
         bytes = need;
      }
      // a block that is asked for a little more every other chunk (result blocks: the chunks' keypoint counts differ) grows with
      // head-room, so that hipHostFree + hipHostMalloc (hundreds of MB, device-synchronising) stop after the first chunks
      void ensure_grow(size_t need)
      {
         if (need <= bytes) return;
         ensure((need + need / 4 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1));
      }
      void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
   };
   Pinned pin_in[2];
   // Page-locked buffers the readers of hesaff_process_files fill directly (chunk_engine.h: PinHooks): handed out by size, taken back
   // when their image is on the device, kept pinned from one list to the next (pinning costs 0.1 ms per MB), released with the context.
   // At most kPinReadBytes are out or parked; a request beyond that gets nullptr (the image then takes the staging copy).
   struct PinReadCache {
      static constexpr size_t kLargest = (size_t)64 << 20;
      size_t max_bytes = (size_t)4 << 30;    // out + parked never exceed this (hesaff_set_pinned_read_budget)
      size_t keep_bytes = (size_t)1 << 30;   // parked buffers kept from one hesaff_process_files call to the next
      std::mutex mu;
      std::vector<std::pair<void *, size_t>> parked;
      size_t bytes_total = 0;   // out + parked
      int device = 0;
      // (hipHostFree / hipHostMalloc are device-synchronising and slow: never under `mu`, which FileIO reaches with its own lock held)
      void *take(size_t bytes)
      {
         if (bytes == 0 || bytes > kLargest) return nullptr;
         std::vector<std::pair<void *, size_t>> evicted;
         bool room = false;
         {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t k = 0; k < parked.size(); k++)
               if (parked[k].second == bytes) { void *q = parked[k].first; parked[k] = parked.back(); parked.pop_back(); return q; }
            // no room: parked buffers of other sizes (an earlier list's images) make way
            while (bytes_total + bytes > max_bytes && !parked.empty()) {
               evicted.push_back(parked.back());
               bytes_total -= parked.back().second;
               parked.pop_back();
            }
            room = bytes_total + bytes <= max_bytes;
            if (room) bytes_total += bytes;
         }
         for (auto &b : evicted) (void)hipHostFree(b.first);
         if (!room) return nullptr;
         void *q = nullptr;
         if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lk(mu);
            bytes_total -= bytes;
            return nullptr;
         }
         return q;
      }
      void give(void *q, size_t bytes)
      {
         std::lock_guard<std::mutex> lk(mu);
         parked.emplace_back(q, bytes);
      }
      // the end of a hesaff_process_files call: what is parked beyond `keep` goes back to the system
      void trim(size_t keep)
      {
         std::vector<std::pair<void *, size_t>> out;
         {
            std::lock_guard<std::mutex> lk(mu);
            size_t held = 0;
            for (auto &b : parked) held += b.second;
            while (held > keep && !parked.empty()) {
               out.push_back(parked.back());
               held -= parked.back().second; bytes_total -= parked.back().second;
               parked.pop_back();
            }
         }
         for (auto &b : out) (void)hipHostFree(b.first);
      }
      void release()
      {
         trim(0);
         std::lock_guard<std::mutex> lk(mu);
         bytes_total = 0;
      }
   } pin_read;
   hipEvent_t ev_h2d_blk[2] = {nullptr, nullptr};   // blocking-sync: the staging thread sleeps until a chunk's direct copies have left the readers' buffers
   std::vector<Pinned> pin_out;       // result blocks: one per chunk of the current call (hesaff_detect_batch), or a ring of three
   hesaff_engine::BlockRing ring;     // (hesaff_detect_batch_cb, hesaff_process_files: a block returns to the ring when its consumer is done with it)
   hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
   hipEvent_t ev_exp[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // profiling, per staging slot: brackets of a chunk's length pass and of its write pass (the host's waits between them - a free pinned block - are not the export's)
   float export_ms = 0.0f; int32_t export_rows = 0;
   hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_in_free[2] = {nullptr, nullptr}, ev_out_ready[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
   // Small results the host reads every batch (per-image counts, counters, byte offsets of the text rows) arrive in PAGE-LOCKED memory:
   // a hipMemcpyAsync into pageable memory is not asynchronous - the calling thread waits inside the runtime, spinning, until everything
   // before it in the stream has run (a whole batch of kernels: one busy core per context, measured in round 5) - whereas a copy into
   // pinned memory is enqueued and the host sleeps on a blocking-sync event.
   struct HostSmall {
      void *p = nullptr; size_t bytes = 0;
      void *ensure(size_t need)
      {
         if (need > bytes) {
            if (p) (void)hipHostFree(p);
            p = nullptr; bytes = 0;
            const size_t cap = std::max<size_t>(need * 2, 4096);
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { p = nullptr; throw HsError(HESAFF_ERR_NOMEM, "hipHostMalloc failed (small result block)"); }
            bytes = cap;
         }
         return p;
      }
      void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
   } h_small_end, h_small_mid, h_small_exp;
   struct HStarts {   // view of h_small_end with std::vector's two members in use
      int32_t *q = nullptr;
      int32_t *data() const { return q; }
   } h_starts;
   DevBuf t_mask_idx, t_sgrad_nb, t_sgrad_om, t_vo_rows, t_vo_src, b_rowprefix, b_trows, b_trows2, b_trows3;
   DevBuf b_jcoef[2], b_jplane;   // JPEG chunks: the images' coefficient blobs per input slot, the component planes after the inverse DCT (kernels_jpeg.h)
   DevBuf b_ex_len, b_ex_sums, b_ex_off, b_ex_imgoff, b_ex_starts;   // device export (kernels_export.h): row lengths, sums / offsets per 64 rows, offsets per image
   size_t rows_lds_set = 0;            // dynamic LDS opt-in of k_patch_large_rows on THIS device
   // persistent grids of the LDS-window kernels: exactly as many blocks as the device holds at once (CUs x resident
   // blocks per CU), so that every block takes the same share of a bin's list; queried per device at hesaff_create
   int n_cu = 256;
   uint32_t g_small0 = 256 * 6, g_small1 = 256 * 4, g_mid = HS_MID_BLOCKS, g_big = HS_BIG_BLOCKS, g_lfin = 256 * 4, g_shist = 256 * 32;
   uint32_t trows_rows = 4u << 20;     // rows of T' (82 floats each) the large-window buffer holds at least: 1.3 GB

   hesaff_timings tm;
   int profiling = 0;
   int out_format = HESAFF_OUT_TEXT;   // hesaff_set_output_format
   int resume = 0;                     // hesaff_set_resume: 0 off, 1 skip complete outputs (O(1) test), 2 strict (rows counted)
   int pool_priority = -1;             // hesaff_set_pool_priority: -1 lower the pool's priority when the plan is CPU-starved, 0 never, 1 always
   int stage_threads = 4;              // host threads that copy a chunk's pixels into pinned memory (hesaff_process_files: within its thread budget)
   hipStream_t side_streams[HS_NSIDE] = {nullptr, nullptr, nullptr, nullptr};
   hipStream_t sift_stream = nullptr, sift_stream2 = nullptr;   // descriptor kernels of even / odd groups (sift2: HESAFF_SIFT2)
   bool sift2 = true;
   hipStream_t aff_stream = nullptr;      // affine shape of image group g+1 runs beside the patch extraction of group g
   hipEvent_t ev_detect_done = nullptr, ev_batch_done = nullptr;   // blocking-sync events: the host sleeps instead of spinning
   std::vector<hipEvent_t> ev_aff;        // one per image group, grown on demand
   hipEvent_t ev_extract_done[HS_NSLOT] = {}, ev_sift_done[HS_NSLOT] = {};
   DevBuf b_patches2[HS_NSLOT], b_siftvec2[HS_NSLOT], b_meanvar2[HS_NSLOT], b_siftvo2[HS_NSLOT];
   hipEvent_t ev_fork = nullptr, ev_join[HS_NSIDE] = {nullptr, nullptr, nullptr, nullptr};
   bool fast_pyramid = false;      // hesaff_params.fast == 2: windows beyond bin 0 sampled from the scale-space level with the matching blur (not bit-exact)
   // schedule knobs: fixed in the product build, environment-driven only under -DHESAFF_TUNING
   bool no_overlap = false;        // HESAFF_OVERLAP=0: every kernel alone on the device (per-kernel profiling)
   uint32_t sift_group_kpts = 0;   // HESAFF_GROUP: keypoints per image group; 0 = by batch
   bool taper_groups = false;      // HESAFF_TAPER: small groups at both ends of a batch (pipeline fill / drain)
   int aff_blocks_per_cu = 8;      // HESAFF_AFF_BLOCKS: persistent k_affine blocks per CU (19 KB of LDS each: 8 resident).  Alone on the device
                                   // 64 / 128 blocks per CU are 4 % faster (20.7 / 20.6 vs 21.6 ms), beside the other stages' kernels they
                                   // make the step 3.5 % slower (453 vs 438 ms at B = 128): the queued blocks take every slot that frees up
   int side_mask = 15;             // HESAFF_SIDE: bit i = window-size bin i runs on its own side stream
   int force_bands = 0;            // HESAFF_BANDS: force the band count of k_blur_hess_march
   int force_exband = 0;           // HESAFF_EXBAND: rows per band of k_extrema_march (tuning)
   bool debug = false;             // HESAFF_DEBUG=1: launch geometry on stderr
   uint32_t sgrad_grid = 0;        // persistent grid of k_sift_grad (set with the device: 32 blocks per CU; HESAFF_SGRAD_GRID; 0: one block per keypoint)
   int large_stream = 0;           // HESAFF_LARGE_STREAM: 1 = the large-window kernels behind bins 0 and 1 on their stream, 0 = on the main stream (behind bin 3)
   int large_nw = 2;               // HESAFF_LARGE_NW: wavefronts per block of the three-row form at most (0: as many as fit, up to four).  Two: blocks of 40 KB find room beside the other stages' kernels where blocks of 80 KB wait (dense step 786 -> 773 ms, photographs 393 -> 383)
   int large_nrow = 3;             // HESAFF_LARGE_NROW: window rows per wavefront step of k_patch_large_rows (3, or 1: the round-5 form)
   int large_split = 1280;         // HESAFF_LARGE_SPLIT: windows up to this side in a launch of their own when the batch holds larger ones (0: one launch)
   uint32_t sift_slice = 0;        // HESAFF_SIFT_SLICE: keypoints per slice of the descriptor stage (launch_sift); 0 = a group's kernels each over the whole group
   bool sift_slice_ring = true;    // HESAFF_SLICE_RING: the slices of a group reuse one slice-sized piece of the intermediates (0: every slice its own piece)

   std::vector<hipEvent_t> ev_pool;
   size_t ev_used = 0;
};

namespace {

struct EvPair { hipEvent_t a, b; int kind; double bytes; };

hipEvent_t get_event(hesaff_ctx *c)
{
   if (c->ev_used == c->ev_pool.size()) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      c->ev_pool.push_back(e);
   }
   return c->ev_pool[c->ev_used++];
}

template <class T> void upload(DevBuf &b, const std::vector<T> &v)
{
   b.ensure(std::max<size_t>(v.size() * sizeof(T), 16));
   HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
}

void build_tables(hesaff_ctx *c)
{
   std::vector<float> smm(HS_SMM_PIX), sm(HS_PATCH_PIX), w0(HS_PATCH), w1(HS_PATCH);
   std::vector<int32_t> b0(HS_PATCH), b1(HS_PATCH);
   hesaff::gauss_mask(HS_SMM, smm.data());
   hesaff::circ_gauss_mask(HS_PATCH, sm.data());
   hesaff::sift_bins(b0.data(), b1.data(), w0.data(), w1.data());
   {
      std::vector<int32_t> midx;
      for (int i = 0; i < HS_PATCH_PIX; i++)
         if (sm[i] > 0) midx.push_back(i);
      c->n_masked = (int)midx.size();
      upload(c->t_mask_idx, midx);
      // k_sift_grad's per-pixel constants (affine.cpp:14-33 stencil convention: one-sided differences at the patch border)
      std::vector<int32_t> nb(4 * 1280, 0), om(2 * 1280, 0);
      for (size_t s = 0; s < 1280; s++) {
         const bool used = s < midx.size();
         const int i = used ? midx[s] : 0, r = i / HS_PATCH, cc = i - r * HS_PATCH;
         const bool valid = used && r < HS_PATCH - 1 && cc < HS_PATCH - 1;   // row / column 40 carry no weight in samplePatch
         if (valid) {
            nb[4 * s + 0] = 4 * (cc == 0 ? i : i - 1);
            nb[4 * s + 1] = 4 * (i + 1);
            nb[4 * s + 2] = 4 * (r == 0 ? i : i - HS_PATCH);
            nb[4 * s + 3] = 4 * (i + HS_PATCH);
         }
         om[2 * s + 0] = valid ? r * (HS_PATCH - 1) + cc : -1;
         memcpy(&om[2 * s + 1], &sm[i], 4);
      }
      upload(c->t_sgrad_nb, nb); upload(c->t_sgrad_om, om);
      // layout of the gradient pairs in HBM (kernels_sift.h: HS_VO_COMPACT), from the mask itself: per row the span of 16-byte items
      // (two pixels) that hold a pixel with weight, rows back to back; the plain layout is "every row whole"
      std::vector<int32_t> vrow(4 * HS_VO_DIM, 0);
      std::vector<uint16_t> vsrc(HS_VO_ITEMS, 0);
      int at = 0;
      for (int r = 0; r < HS_VO_DIM; r++) {
         int flo = 1, fhi = 0;
         if (HS_VO_COMPACT) {
            for (int cc = 0; cc < HS_VO_DIM; cc++)
               if (sm[r * HS_PATCH + cc] > 0) { if (fhi < flo) flo = cc / 2; fhi = cc / 2; }
         } else { flo = 0; fhi = HS_VO_DIM / 2 - 1; }
         vrow[4 * r + 0] = at - flo; vrow[4 * r + 1] = flo; vrow[4 * r + 2] = fhi;
         for (int f = flo; f <= fhi; f++, at++)
            if (at < HS_VO_ITEMS) vsrc[(size_t)at] = (uint16_t)(r * (HS_VO_DIM / 2) + f);
      }
      // the layout constants of kernels_sift.h are those of THIS mask (helpers.cpp:131-147 at patchSize 41)
      if (at != (HS_VO_COMPACT ? HS_VO_ZERO : HS_VO_ITEMS) || sm[0] > 0) throw HsError(HESAFF_ERR_ARG, "internal: gradient-pair layout does not match the circular mask");
      upload(c->t_vo_rows, vrow); upload(c->t_vo_src, vsrc);
   }
   upload(c->t_smm, smm); upload(c->t_sift, sm); upload(c->t_bin0, b0); upload(c->t_bin1, b1); upload(c->t_w0, w0); upload(c->t_w1, w1);
   c->up = c->par.upscaleInputImage > 0 ? 1 : 0;
   c->sched = hesaff::make_schedule(c->par.initialSigma, c->up != 0);
   std::vector<float> taps;
   for (int i = 0; i < 5; i++) {
      const float sigma = i == 0 ? c->sched.init_sigma : c->sched.blur_sigma[i];
      c->pyr_tap_off[i] = (int)taps.size();
      taps.resize(taps.size() + 256, 0.0f);
      if (i == 0 && !(c->sched.init_sigma > 0.0f)) { c->pyr_K[0] = 0; continue; }   // pyramid.cpp:276: no initial blur
      const int K = hesaff::gauss_ksize(sigma);
      if (K > 255) throw HsError(HESAFF_ERR_ARG, "initialSigma too large (a pyramid blur would need more than 255 taps)");
      c->pyr_K[i] = K;
      if (K == 1) taps[c->pyr_tap_off[i]] = 1.0f;
      else hesaff::gauss_taps(K, sigma, taps.data() + c->pyr_tap_off[i]);
   }
   c->pyr_march = c->pyr_K[1] == 9 && c->pyr_K[2] == 11 && c->pyr_K[3] == 13 && c->pyr_K[4] == 15;
   upload(c->t_pyr_taps, taps);
   const hesaff_params &p = c->par;
   DConsts &k = c->consts;
   // pyramid.h:59-64
   k.edgeScoreThreshold = (p.edgeEigenValueRatio + 1.0f) * (p.edgeEigenValueRatio + 1.0f) / p.edgeEigenValueRatio;
   k.finalThreshold = p.threshold * p.threshold;
   k.positiveThreshold = (float)(0.8 * k.finalThreshold);
   k.negativeThreshold = -k.positiveThreshold;
   k.convergenceThreshold = p.convergenceThreshold;
   k.affInitialSigma = 1.6f;   // AffineShapeParams::initialSigma affine.h:40 (not overridden by hesaff.cpp)
   k.mrSize = p.mrSize;
   k.maxBinValue = p.maxBinValue;
   k.maxIterations = p.maxIterations;
   k.pd0 = c->up ? 0.5f : 1.0f;   // pixelDistance of octave 0, pyramid.cpp:264,270
}

// taps of the per-keypoint patch blur (affine.cpp:129: sigma = 1.5f * P0/41) for odd P0
void ensure_patch_taps(hesaff_ctx *c, int max_p0)
{
   if (max_p0 <= c->max_p0) return;
   if ((max_p0 & 1) == 0) max_p0++;
   std::vector<float> taps;
   std::vector<int32_t> off((max_p0 + 1) / 2), kk((max_p0 + 1) / 2);
   for (int P0 = 1; P0 <= max_p0; P0 += 2) {
      const float scale = (float)P0 / (float)HS_PATCH;
      const float sigma = 1.5f * scale;
      const int K = hesaff::gauss_ksize(sigma);
      off[(P0 - 1) / 2] = (int32_t)taps.size();
      kk[(P0 - 1) / 2] = K;
      taps.resize(taps.size() + K);
      if (K == 1) taps[taps.size() - 1] = 1.0f;
      else hesaff::gauss_taps(K, sigma, taps.data() + off[(P0 - 1) / 2]);
   }
   upload(c->t_patch_taps, taps); upload(c->t_patch_off, off); upload(c->t_patch_k, kk);
   c->max_p0 = max_p0;
}

void refresh_tables_struct(hesaff_ctx *c)
{
   KpTables &t = c->tables;
   t.smm_mask = c->t_smm.as<float>(); t.sift_mask = c->t_sift.as<float>();
   t.bin0 = c->t_bin0.as<int32_t>(); t.bin1 = c->t_bin1.as<int32_t>();
   t.w0 = c->t_w0.as<float>(); t.w1 = c->t_w1.as<float>();
   t.patch_taps = c->t_patch_taps.as<float>(); t.patch_tap_off = c->t_patch_off.as<int32_t>(); t.patch_tap_k = c->t_patch_k.as<int32_t>();
   t.max_p0 = c->max_p0;
   t.mask_idx = c->t_mask_idx.as<int32_t>();
   t.sgrad_nb = c->t_sgrad_nb.as<int4>(); t.sgrad_om = c->t_sgrad_om.as<int2>();
   t.vo_rows = c->t_vo_rows.as<int4>(); t.vo_src = c->t_vo_src.as<uint16_t>();
   t.n_masked = c->n_masked;
}

DPlane make_plane(float *p, int rows, int cols, int pitch)
{
   DPlane d;
   d.p = p; d.rows = rows; d.cols = cols; d.pitch = pitch; d.img_stride = (long long)rows * pitch;
   return d;
}

template <class KERNEL> void set_dyn_lds(KERNEL kern, size_t lds)
{
   HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

size_t small_extract_lds_bytes(int bin) { return (size_t)(bin == 0 ? SmallGeom<0>::FLOATS : SmallGeom<1>::FLOATS) * 4; }
size_t mid_lds_bytes() { return (size_t)MidGeom<HS_MID_PMAX>::FLOATS * 4; }
size_t big_lds_bytes() { return (size_t)MidGeom<HS_BIN3_PMAX>::FLOATS * 4; }

// geometry of the large-window row kernel for windows up to pmax: LDS per wave = window row + replicated borders + taps
constexpr size_t HS_LDS_PER_CU = 160 * 1024;
struct LargeGeom { int srow_stride, tap_stride; size_t lds; };   // lds: bytes for a block of FOUR wavefronts
LargeGeom large_geom(int pmax)
{
   LargeGeom g;
   // window row + r replicated border samples on each side, r = K/2 <= (6 * 1.5 * P0/41 + 2) / 2
   g.srow_stride = round_up((int)(pmax * 1.23) + 16, 64);
   g.tap_stride = round_up((int)(pmax * 0.22) + 8, 64);   // K = odd(int(6 * 1.5 * P0/41 + 1))
   g.lds = (size_t)4 * (g.srow_stride + g.tap_stride) * 4;
   return g;
}

// Dynamic-LDS opt-ins are per device: applied when a context is created on its device (hesaff_create) and, for the
// large-window kernel whose need depends on the image size, in plan().
template <class KERNEL> uint32_t resident_grid(hesaff_ctx *c, KERNEL kern, int threads, size_t dyn_lds, uint32_t fallback_per_cu)
{
   int nb = 0;
   if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, threads, dyn_lds) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = (int)fallback_per_cu;
   }
   return (uint32_t)c->n_cu * (uint32_t)nb;
}

void set_kernel_attrs(hesaff_ctx *c)
{
   set_dyn_lds(k_patch_extract_small<0>, small_extract_lds_bytes(0));
   set_dyn_lds(k_patch_extract_small<1>, small_extract_lds_bytes(1));
   set_dyn_lds(k_patch_mid<HS_MID_PMAX>, mid_lds_bytes());
   set_dyn_lds(k_patch_mid<HS_BIN3_PMAX>, big_lds_bytes());
   hipDeviceProp_t prop;
   HIP_TRY(hipGetDeviceProperties(&prop, c->device));
   c->n_cu = std::max(1, prop.multiProcessorCount);
   c->g_small0 = resident_grid(c, k_patch_extract_small<0>, 256, small_extract_lds_bytes(0), 6);
   c->g_small1 = resident_grid(c, k_patch_extract_small<1>, 256, small_extract_lds_bytes(1), 4);
   // the row-streamed bins claim their items dynamically: any grid that fills the device works; one T' slot per block
   c->g_mid = std::min<uint32_t>(resident_grid(c, k_patch_mid<HS_MID_PMAX>, 256, mid_lds_bytes(), HS_MID_CAP), HS_MID_BLOCKS);
   c->g_big = std::min<uint32_t>(resident_grid(c, k_patch_mid<HS_BIN3_PMAX>, 256, big_lds_bytes(), HS_BIG_CAP), HS_BIG_BLOCKS);
   c->g_lfin = resident_grid(c, k_patch_large_finish, 256, 0, 4);
   c->g_shist = resident_grid(c, k_sift_hist, 64, 0, 32);
   // k_sift_grad: a block keeps its per-pixel tables and requests the next patch while it works on the current one; well
   // over the resident count so that the tail of a launch is short (measured: one block per keypoint 18.5 ms per 32 UHD
   // images, 6 / 16 / 32 / 64 blocks per CU 15.2 / 13.8 / 13.3 / 13.4)
   c->sgrad_grid = (uint32_t)c->n_cu * 32u;
   // The statically strided grids are launched HS_OVERSUB x oversubscribed: an item's cost varies several-fold, blocks beyond the
   // resident count start as others finish, and the hardware's block scheduler evens out what a fixed stride cannot
   // (a claim per item on an atomic counter serialises in L2 instead).  Measured per 32 UHD images, x 1 / 8 / 32:
   // k_patch_extract_small<0> 13.8 / 13.3 / 12.9 ms, <1> 6.1 / - / 5.9, k_sift_hist 13.4 / 12.0 / 11.7.
   c->g_small0 *= HS_OVERSUB; c->g_small1 *= HS_OVERSUB; c->g_shist *= HS_OVERSUB;
}

// Buffer plan for a batch of B images of H x W.
void plan_buffers(hesaff_ctx *c, int B, int H, int W);
void plan(hesaff_ctx *c, int B, int H, int W)
{
   try {
      plan_buffers(c, B, H, W);
   } catch (const HsError &e) {
      if (e.code == HESAFF_ERR_NOMEM) {
         // a plan the device cannot hold must not keep what it managed to allocate on the way (hundreds of GB for an absurd
         // capacity request): every geometry-sized buffer goes back, the next plan starts from nothing
         DevBuf *bufs[] = {&c->b_gray, &c->b_up, &c->b_L, &c->b_L3, &c->b_R, &c->b_map, &c->b_bitmask, &c->b_prefix, &c->b_blocksums, &c->b_cand,
                           &c->b_rec_f, &c->b_rec_i, &c->b_rec_w, &c->b_hess_f, &c->b_hess_i, &c->b_aff, &c->b_pw, &c->b_bins, &c->b_rank, &c->b_desc,
                           &c->b_out, &c->b_starts};
         (void)hipStreamSynchronize(c->stream);
         for (DevBuf *b : bufs) b->release();
         c->map_clean = false;
      }
      throw;
   }
}

void plan_buffers(hesaff_ctx *c, int B, int H, int W)
{
   if (B <= c->B && H == c->H && W == c->W) return;
   if (H < 1 || W < 1 || (H << c->up) > 65535 || (W << c->up) > 65535) throw HsError(HESAFF_ERR_ARG, "image size out of range (1..65535 at the first pyramid level)");
   const int PH = H << c->up, PW = W << c->up;   // first pyramid level
   // the cached geometry describes buffers that are about to be replaced: a failure below must not leave it valid
   c->B = c->H = c->W = 0;
   c->oct.clear();
   c->L.clear();
   long long words = 0;
   size_t L_floats = 0;
   {
      int r = PH, cc = PW;
      const int minSize = 2 * HS_BORDER + 2;   // pyramid.cpp:283
      while (r > minSize && cc > minSize) {
         OctGeom g;
         g.rows = r; g.cols = cc; g.pitch = round_up(cc, 64);
         g.words_per_row = (cc + 63) / 64;
         g.word_base = words;
         words += (long long)HS_NSCALES * r * g.words_per_row;
         L_floats += (size_t)3 * B * r * g.pitch;
         c->oct.push_back(g);
         r /= 2; cc /= 2;
         if ((int)c->oct.size() >= HS_MAX_OCTAVES) break;
      }
   }
   c->words_per_image = words;
   const int pitch0 = round_up(W, 64);
   const size_t plane0 = (size_t)B * H * pitch0;
   c->b_gray.ensure(plane0 * 4);
   c->gray = make_plane(c->b_gray.as<float>(), H, W, pitch0);
   c->b_L.ensure(std::max<size_t>(L_floats * 4, 16));
   {
      float *p = c->b_L.as<float>();
      for (const OctGeom &g : c->oct)
         for (int l = 0; l < 3; l++) {
            c->L.push_back(make_plane(p, g.rows, g.cols, g.pitch));
            p += (size_t)B * g.rows * g.pitch;
         }
   }
   const int ppitch0 = round_up(PW, 64);
   const size_t pplane0 = (size_t)B * PH * ppitch0;
   if (c->up) {
      c->b_up.ensure(pplane0 * 4);
      c->upimg = make_plane(c->b_up.as<float>(), PH, PW, ppitch0);
   }
   c->b_L3.ensure(pplane0 * 4);
   c->b_R.ensure(pplane0 * 4 * 5);
   {
      const void *before = c->b_map.p;
      const size_t before_bytes = c->b_map.bytes;
      c->b_map.ensure(std::max<size_t>((size_t)B * PH * PW * 4, 16));
      // a new block is filled once before its first use (run_detection).  Pointer AND size: ensure()'s out-of-memory path frees the old
      // block first, and the larger one may come back at the same address with a tail that was never filled
      if (c->b_map.p != before || c->b_map.bytes != before_bytes) c->map_clean = false;
      int kb = 1;
      while (kb < 32 && (3ull * (unsigned long long)PH * PW) > (1ull << kb)) kb++;
      if (kb != c->map_kbits) { c->map_kbits = kb; c->map_clean = false; }   // (another key width: epochs of the old one mean nothing)
   }
   const long long total_words = (long long)B * words;
   c->b_bitmask.ensure(std::max<size_t>((size_t)total_words * 8, 16));
   c->b_prefix.ensure(std::max<size_t>((size_t)(total_words + 1) * 4, 16));
   double mpx = (double)B * PH * PW / 1.0e6;   // capacity per megapixel of the first pyramid level
   double capd = mpx * (double)c->par.max_kpts_per_mpx;
   if (capd < 4096) capd = 4096;
   if (capd > 2.0e9) throw HsError(HESAFF_ERR_ARG, "batch too large for 32-bit keypoint indices");
   c->cap = ((uint32_t)capd + 63u) & ~63u;   // a multiple of 64: the arrays carved out of one buffer (cap entries each) stay 16-byte aligned
   const size_t cap = c->cap;
   const long long scan_items = std::max<long long>(total_words, (long long)cap);
   c->b_blocksums.ensure((size_t)((scan_items + SCAN_BLOCK - 1) / SCAN_BLOCK + 1) * 4);
   c->b_counters.ensure(64 * 4);
   {
      // candidate slots: the keypoint capacity + what the wavefronts of k_extrema_march may leave unused of their blocks of 64
      // (octave 0 has the most wavefronts: one per 248-column strip and 32-row band at least)
      const unsigned long long waves0 = (unsigned long long)((PW + EXM_STRIP - 1) / EXM_STRIP) * (unsigned long long)(PH / 32 + 1) * (unsigned long long)B;
      // + cap / 8: a wavefront also abandons the rest of its block whenever a ballot group does not fit (holes grow with the number of
      // blocks, not only with the number of wavefronts).  96 bytes per slot: 11 GB per 256 UHD images at the default max_kpts_per_mpx
      const unsigned long long cc = (unsigned long long)cap + (unsigned long long)cap / 8 + HS_CAND_BLOCK * waves0;
      if (cc > 0xfffffff0ull) throw HsError(HESAFF_ERR_ARG, "batch too large for 32-bit candidate indices");
      c->cand_cap = (uint32_t)cc;
      c->b_cand.ensure((size_t)cc * sizeof(CandRec));
   }
   c->b_rec_f.ensure(cap * 4 * 4);
   c->b_rec_i.ensure(cap * 4 * 4);
   c->b_rec_w.ensure(cap * 8);
   c->b_hess_f.ensure(cap * 4 * 4);
   c->b_hess_i.ensure(cap * 2 * 4);
   c->b_aff.ensure(cap * 6 * 4);
   c->b_pw.ensure(cap * 6 * 4);
   c->b_bins.ensure(cap * HS_NBINS * 4);
   c->b_rank.ensure((cap + 1) * 4);
   c->b_desc.ensure(cap * 128);
   c->b_out.ensure(cap * sizeof(KeyRec));
   c->b_starts.ensure(((size_t)(B + 1) * 3 + 2) * 4);   // hessian starts | descriptor starts | huge-window rows per image, + their largest side
   // patch taps: P <= sqrt(W*H) + small (the det-1 window must fit)
   const int max_p0 = (int)std::floor(std::sqrt((double)W * (double)H)) + 3;
   ensure_patch_taps(c, max_p0);
   refresh_tables_struct(c);
   // per-block T' slots of the row-streamed bins (persistent grids of fixed size)
   c->b_trows2.ensure((size_t)HS_MID_BLOCKS * (HS_MID_PMAX + 2 * HS_MID_RPAD) * HS_NEED * 4);
   c->b_trows3.ensure((size_t)HS_BIG_BLOCKS * (HS_BIN3_PMAX + 2 * HS_BIG_RPAD) * HS_NEED * 4);
   {
      // k_patch_large_rows keeps one window row (+ borders, + taps) per wavefront in LDS: blocks of four wavefronts while four rows of the
      // batch's largest window fit the CU's 160 KB, of two or one beyond that (run_patch_stage); a row that does not fit alone - a window
      // above ~27 900 pixels a side, i.e. an image of more than 780 Mpx - is refused here
      const LargeGeom lg = large_geom(c->max_p0 + 2);
      if (lg.lds / 4 > HS_LDS_PER_CU) throw HsError(HESAFF_ERR_ARG, "image too large for the large-window row kernel (sqrt(width x height) above about 27900)");
      // (three rows per wavefront where they fit: the launches ask for up to the whole LDS of a CU)
      const size_t want = std::min<size_t>(lg.lds + (size_t)8 * lg.srow_stride * 4, HS_LDS_PER_CU);
      if (want > c->rows_lds_set) {
         set_dyn_lds(k_patch_large_rows, want);
         c->rows_lds_set = want;
      }
   }
   c->B = B; c->H = H; c->W = W;
}

template <class LOAD> void exclusive_scan(hesaff_ctx *c, LOAD load, long long n, uint32_t *out, uint32_t *total)
{
   // out[0..n) exclusive prefix, *total = sum (device pointers)
   if (n <= 0) { HIP_TRY(hipMemsetAsync(total, 0, 4, c->stream)); return; }
   const int nb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
   uint32_t *bs = c->b_blocksums.as<uint32_t>();
   hipLaunchKernelGGL(k_scan_reduce<LOAD>, dim3(nb), dim3(256), 0, c->stream, load, n, bs);
   hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, c->stream, bs, nb, total);
   hipLaunchKernelGGL(k_scan_down<LOAD>, dim3(nb), dim3(256), 0, c->stream, load, n, bs, out);
}

// Stage timers: one HIP event pair per bracket, recorded on the stream the bracketed work is launched on.
struct StageTimer {
   hesaff_ctx *c;
   std::vector<EvPair> pairs;
   std::vector<hipStream_t> streams;
   explicit StageTimer(hesaff_ctx *ctx) : c(ctx) {}
   int begin(int kind, double bytes = 0, hipStream_t st = nullptr)
   {
      if (!c->profiling) return -1;
      if (kind >= 100 && c->profiling < 2) return -1;
      if (!st) st = c->stream;
      EvPair p; p.a = get_event(c); p.b = get_event(c); p.kind = kind; p.bytes = bytes;
      (void)hipEventRecord(p.a, st);
      pairs.push_back(p);
      streams.push_back(st);
      return (int)pairs.size() - 1;
   }
   void end(int id) { if (id >= 0) (void)hipEventRecord(pairs[id].b, streams[id]); }
};

enum { T_PYR = 0, T_DET = 1, T_AFF = 2, T_PATCH = 3, T_SIFT = 4, T_TOTAL = 5, T_PACK = 6, T_BLURHESS = 100, T_EXTREMA = 101 };

// Band height of k_blur_hess_march: 16 bands per octave is the measured optimum for 16 x 4K at every octave
// (sweeps in profiles/r01_notes.md); small batches get proportionally more bands to keep ~1000 blocks in flight.
template <int K, bool WL, bool WR, bool WH, bool WR0 = false, bool SRC8 = false>
void launch_march(hesaff_ctx *c, const DPlane &in, const DPlane &outL, const DPlane &outR, const DPlane &outHalf, const float *taps,
                  float norm2, int B, const DPlane &outR0 = DPlane(), float norm2_in = 0.0f, const GraySrc &gs = GraySrc(), const DPlane &outGray = DPlane())
{
   const int strips = (in.cols + BM_STRIP - 1) / BM_STRIP;
   const long long blocks_per_band = (long long)((strips + 3) / 4) * B;
   int best_nb = 16 * (int)std::max<long long>(1, std::min<long long>(4, 64 / std::max<long long>(1, blocks_per_band)));
   best_nb = std::max(1, std::min(best_nb, std::max(1, in.rows / 8)));
   if (c->force_bands > 0) best_nb = std::min(c->force_bands, in.rows);
   const int band = (in.rows + best_nb - 1) / best_nb;
   if (c->debug) fprintf(stderr, "[hesaff] march K=%d %dx%d B=%d bands=%d band=%d blocks=%lld\n", K, in.cols, in.rows, B, best_nb, band, blocks_per_band * best_nb);
   const dim3 grid((strips + 3) / 4, (in.rows + band - 1) / band, B);
   hipLaunchKernelGGL((k_blur_hess_march<K, WL, WR, WH, WR0, SRC8>), grid, dim3(256), 0, c->stream, in, outL, outR, outHalf, taps, norm2, band, outR0, norm2_in, gs, outGray);
}

template <bool WL, bool WR, bool WH>
void launch_blur_hess(hesaff_ctx *c, const DPlane &in, const DPlane &outL, const DPlane &outR, const DPlane &outHalf, const float *taps, int K,
                      float norm2, int B)
{
   switch (K) {
      case 9: launch_march<9, WL, WR, WH>(c, in, outL, outR, outHalf, taps, norm2, B); return;
      case 11: launch_march<11, WL, WR, WH>(c, in, outL, outR, outHalf, taps, norm2, B); return;
      case 13: launch_march<13, WL, WR, WH>(c, in, outL, outR, outHalf, taps, norm2, B); return;
      case 15: launch_march<15, WL, WR, WH>(c, in, outL, outR, outHalf, taps, norm2, B); return;
      default: break;   // non-default initialSigma
   }
   if (K <= 2 * BH_RMAX + 1) {
      // LDS-tile kernel: any tap count up to 15
      const dim3 grid((in.cols + BH_TW - 1) / BH_TW, (in.rows + BH_TH - 1) / BH_TH, B);
      hipLaunchKernelGGL((k_blur_hess_tile<WL, WR, WH>), grid, dim3(256), 0, c->stream, in, outL, outR, outHalf, taps, K, norm2);
      return;
   }
   // any larger tap count (initialSigma above ~1.9): plain two-pass blur + stand-alone response / decimation kernels
   const size_t planeF = (size_t)B * in.rows * in.pitch;
   c->b_generic.ensure(planeF * 4 * 2);
   const DPlane tmp = make_plane(c->b_generic.as<float>(), in.rows, in.cols, in.pitch);
   const DPlane blurred = WL ? outL : make_plane(c->b_generic.as<float>() + planeF, in.rows, in.cols, in.pitch);
   const dim3 grid((in.cols + 255) / 256, in.rows, B);
   hipLaunchKernelGGL(k_blur_rows_generic, grid, dim3(256), 0, c->stream, in, tmp, taps, K);
   hipLaunchKernelGGL(k_blur_cols_generic, grid, dim3(256), 0, c->stream, tmp, blurred, taps, K);
   if (WR) hipLaunchKernelGGL(k_hess, grid, dim3(256), 0, c->stream, blurred, outR, norm2);
   if (WH) hipLaunchKernelGGL(k_half, dim3((outHalf.cols + 255) / 256, outHalf.rows, B), dim3(256), 0, c->stream, blurred, outHalf);
}

struct Lists {
   CandList cl;
   RecList rl;
   HessList hl;
   AffineOut ao;
   PatchWork pw;
   uint32_t *counters;
};

Lists make_lists(hesaff_ctx *c)
{
   Lists s;
   uint32_t *cnt = c->b_counters.as<uint32_t>();
   const size_t cap = c->cap;
   s.counters = cnt;
   s.cl.count = cnt + 0; s.cl.items = c->b_cand.as<CandRec>(); s.cl.cap = c->cand_cap; s.cl.overflow = cnt + 2;
   s.rl.count = cnt + 1; s.rl.cap = c->cap;
   float *rf = c->b_rec_f.as<float>();
   s.rl.x = rf; s.rl.y = rf + cap; s.rl.s = rf + 2 * cap; s.rl.response = rf + 3 * cap;
   uint32_t *ri = c->b_rec_i.as<uint32_t>();
   s.rl.meta = (int32_t *)ri; s.rl.cell = ri + cap; s.rl.key = ri + 2 * cap; s.rl.bit = ri + 3 * cap;
   s.rl.word = c->b_rec_w.as<long long>();
   float *hf = c->b_hess_f.as<float>();
   s.hl.x = hf; s.hl.y = hf + cap; s.hl.s = hf + 2 * cap; s.hl.response = hf + 3 * cap;
   int32_t *hi = c->b_hess_i.as<int32_t>();
   s.hl.meta = hi; s.hl.r0c0 = hi + cap; s.hl.cap = c->cap;
   int32_t *ai = c->b_aff.as<int32_t>();
   s.ao.converged = ai; s.ao.iters = ai + cap; s.ao.U = (float *)(ai + 2 * cap);
   int32_t *pi = c->b_pw.as<int32_t>();
   s.pw.P0 = pi; s.pw.alive = pi + cap; s.pw.A = (float *)(pi + 2 * cap);
   s.pw.bin_count = cnt + 8; s.pw.bin_work = cnt + 24; s.pw.bin_items = c->b_bins.as<uint32_t>(); s.pw.cap = c->cap;
   return s;
}

// normalizeAffine for every keypoint k_prepare_patch left alive and binned.  Every launch is a persistent grid of
// fixed size that reads its work-list length from the device-side bin counters: the host never waits for them.
// large_rows_bound: upper bound of the large bin's T' rows in this group (from k_image_large_rows).
void run_patch_stage(hesaff_ctx *c, const Lists &s, const DPlane &image, float *patches_out, uint32_t h_base, uint32_t large_rows_bound,
                     const PlaneTab *pt = nullptr)
{
   hipStream_t st = c->stream;
   PatchIO io;
   memset(&io, 0, sizeof io);
   io.image = image;
   io.patches = patches_out;
   io.h_base = h_base;
   if (c->fast_pyramid && pt) {
      // hesaff_params.fast = 2: bin 0 (P <= 41) on the parity kernel, every larger window from the pyramid (k_patch_pyramid)
      hipStream_t s0 = st;
      const bool forked = c->side_streams[0] && !c->no_overlap;
      if (forked) {
         HIP_TRY(hipEventRecord(c->ev_fork, st));
         HIP_TRY(hipStreamWaitEvent(c->side_streams[0], c->ev_fork, 0));
         s0 = c->side_streams[0];
      }
      hipLaunchKernelGGL(k_patch_extract_small<0>, dim3(c->g_small0), dim3(256), small_extract_lds_bytes(0), s0, s.hl, s.pw, io, c->tables);
      if (forked) HIP_TRY(hipEventRecord(c->ev_join[0], s0));
      hipLaunchKernelGGL(k_patch_pyramid, dim3((uint32_t)c->n_cu * 32u), dim3(256), 0, st, s.hl, s.pw, io, *pt, (int)c->oct.size(), c->consts.pd0, 1);
      if (forked) HIP_TRY(hipStreamWaitEvent(st, c->ev_join[0], 0));
      return;
   }
   // The bins are independent (disjoint keypoints) and each kernel leaves CU resources idle
   // (LDS- or latency-bound), so they run concurrently on side streams.
   hipStream_t s0 = st, s1 = st, s2 = st, s3 = st;
   const bool forked = c->side_streams[0] && !c->no_overlap;
   if (forked) {
      HIP_TRY(hipEventRecord(c->ev_fork, st));
      for (int i = 0; i < HS_NSIDE; i++) HIP_TRY(hipStreamWaitEvent(c->side_streams[i], c->ev_fork, 0));
      if (c->side_mask & 1) s0 = c->side_streams[0];
      if (c->side_mask & 2) s1 = c->side_streams[1];
      if (c->side_mask & 4) s2 = c->side_streams[2];
      if (c->side_mask & 8) s3 = c->side_streams[3];
   }
   {
      PatchIO io2 = io;
      io2.trows = c->b_trows2.as<float>();
      PatchIO io3 = io;
      io3.trows = c->b_trows3.as<float>();
      hipLaunchKernelGGL(k_patch_extract_small<0>, dim3(c->g_small0), dim3(256), small_extract_lds_bytes(0), s0, s.hl, s.pw, io, c->tables);
      hipLaunchKernelGGL(k_patch_extract_small<1>, dim3(c->g_small1), dim3(256), small_extract_lds_bytes(1), s1, s.hl, s.pw, io, c->tables);
      hipLaunchKernelGGL(k_patch_mid<HS_MID_PMAX>, dim3(c->g_mid), dim3(256), mid_lds_bytes(), s2, s.hl, s.pw, io2, c->tables);
      hipLaunchKernelGGL(k_patch_mid<HS_BIN3_PMAX>, dim3(c->g_big), dim3(256), big_lds_bytes(), s3, s.hl, s.pw, io3, c->tables);
   }
   // the rare huge windows (P > 512): row tasks over all of them, then one block per keypoint.  The main stream shares its HIP stream
   // (= hardware queue) with bin 3, whose kernel is the longest of the bins on photographs: behind it the large-window kernels made that
   // queue the stage's critical path (8.1 + 8.8 + 1.9 ms per 32 photograph mosaics against 5.3 for the queue of bins 0 and 1).  They go
   // behind bins 0 and 1 instead (large_stream = 1); k_prepare_patch and the fork event order them after the bin counts either way.
   hipStream_t sl = (forked && c->large_stream == 1) ? s1 : st;
   if (large_rows_bound > 0) {
      const uint32_t rows_cap = std::max(large_rows_bound, c->trows_rows);
      c->b_trows.ensure((size_t)rows_cap * HS_NEED * 4);
      c->b_rowprefix.ensure(((size_t)c->cap + 1) * 4);
      // LDS per wavefront for the largest window that exists in this batch (rounded up so that few distinct launch shapes occur),
      // not for the largest the image could hold.  Three window rows per wavefront step where they fit (k_patch_large_rows); a batch whose
      // largest window is above 1024 runs as two launches - windows up to 1024 with the LDS, i.e. the occupancy, of a 1024 window, the
      // rest with that of the batch's largest.
      const int pmax = std::min(c->max_p0 + 2, std::max(HS_BIN3_PMAX + 1, (c->batch_max_p > 0 ? c->batch_max_p : c->max_p0 + 2)));
      io.trows = c->b_trows.as<float>();
      io.row_prefix = c->b_rowprefix.as<uint32_t>();
      io.trows_cap = rows_cap;
      io.overflow = s.counters + 6;
      hipLaunchKernelGGL(k_large_prefix, dim3(1), dim3(256), 0, sl, s.pw, c->b_rowprefix.as<uint32_t>());
      auto launch_rows = [&](int p_lo, int p_hi) {
         const LargeGeom lg = large_geom(std::min(c->max_p0 + 2, (p_hi + 255) / 256 * 256));
         const size_t wave1 = lg.lds / 4;                                        // one row + taps
         const size_t wave3 = wave1 + (size_t)2 * lg.srow_stride * 4;            // three rows + taps
         // the three-row form only where six wavefronts of it fit a CU (windows up to about 1700): below that occupancy the kernel
         // is all exposed gather latency (measured: 3.5x slower at two wavefronts per CU, profiles/r06_notes.md)
         const int nrow = (c->large_nrow == 3 && wave3 * 6 <= HS_LDS_PER_CU) ? 3 : 1;
         const size_t per_wave = nrow == 3 ? wave3 : wave1;
         // wavefronts per block: four while their rows fit the CU's LDS (plan_buffers made sure one row fits); blocks of two where two
         // such blocks pack the CU's LDS more tightly than one block of four
         uint32_t nw = 4;
         while (nw > 1 && per_wave * nw > HS_LDS_PER_CU) nw >>= 1;
         if (nw == 4 && (HS_LDS_PER_CU / (per_wave * 2)) * 2 > (HS_LDS_PER_CU / (per_wave * 4)) * 4) nw = 2;
         if (nrow == 3 && c->large_nw > 0) nw = std::min<uint32_t>(nw, (uint32_t)c->large_nw);
         const uint32_t gblocks = std::min<uint32_t>((large_rows_bound + nw * HS_LARGE_CHUNK - 1) / (nw * HS_LARGE_CHUNK), 256 * 16 * (4 / nw));
         hipLaunchKernelGGL(k_patch_large_rows, dim3(gblocks), dim3(64 * nw), per_wave * nw, sl, s.hl, s.pw, io, c->tables, lg.srow_stride, lg.tap_stride, nrow,
                            p_lo, std::min(p_hi, 0x7ffffff0));
      };
      if (c->large_split > 0 && pmax > c->large_split) { launch_rows(0, c->large_split); launch_rows(c->large_split, pmax); }
      else launch_rows(0, pmax);
      hipLaunchKernelGGL(k_patch_large_finish, dim3(c->g_lfin), dim3(256), 0, sl, s.pw, io, c->tables);
   }
   if (forked) {
      for (int i = 0; i < HS_NSIDE; i++) HIP_TRY(hipEventRecord(c->ev_join[i], c->side_streams[i]));
      for (int i = 0; i < HS_NSIDE; i++) HIP_TRY(hipStreamWaitEvent(st, c->ev_join[i], 0));
   }
}

// The scale-space + detection part for the current plan; fills the ordered Hessian list.
// src: device u8 images ([B][H][row_stride] with `channels` interleaved channels).
void run_detection(hesaff_ctx *c, const uint8_t *d_src, int channels, long long src_img_stride, int src_row_stride, int B,
                   const Lists &s, StageTimer &tm, bool keep_all_planes, float *planes_out)
{
   const hesaff::OctaveSchedule &sc = c->sched;
   hipStream_t st = c->stream;
   uint32_t *cnt = s.counters;
   const float *ptaps = c->t_pyr_taps.as<float>();
   HIP_TRY(hipMemsetAsync(cnt, 0, 64 * 4, st));
   HIP_TRY(hipMemsetAsync(c->b_bitmask.p, 0, std::max<size_t>((size_t)B * c->words_per_image * 8, 8), st));
   // octaveMap (pyramid.cpp:226: zeroed per octave): the order-key map is filled with "free" when it is new; every pass over an octave then bids
   // with keys of a fresh, smaller epoch (OctaveCtx::map_epoch), so what earlier passes left behind never wins - no fill and no reset per octave
   if (!c->map_clean) { HIP_TRY(hipMemsetAsync(c->b_map.p, 0xFF, c->b_map.bytes, st)); c->map_epoch = c->map_kbits < 32 ? (0xffffffffu >> c->map_kbits) : 0u; c->map_clean = true; }

   int t = tm.begin(T_PYR);
   DPlane none = make_plane(nullptr, 0, 0, 0);
   // Default parameters: grey conversion (hesaff.cpp:138-148) fused into the initial blur 0.5 -> 1.6 (pyramid.cpp:276-280,
   // K = 11): the 8-bit images are read once, the float grey plane (normalizeAffine's input) and L0 are written.
   const bool fused_gray = !c->oct.empty() && c->pyr_K[0] == 11 && !c->up;
   if (fused_gray) {
      GraySrc gs;
      gs.p = d_src; gs.channels = channels; gs.img_stride = src_img_stride; gs.row_stride = src_row_stride;
      const int tb = tm.begin(T_BLURHESS, 0);   // not one of the 58 B/px launches (bytes 0)
      launch_march<11, true, false, false, false, true>(c, c->gray, c->L[0], none, none, ptaps + c->pyr_tap_off[0], 0.0f, B, DPlane(), 0.0f, gs, c->gray);
      tm.end(tb);
   } else {
      // grey conversion; without an initial blur (initialSigma <= the input's own blur) it is the first level directly
      const bool direct = c->pyr_K[0] == 0 && !c->oct.empty();
      const dim3 grid((c->W + 255) / 256, c->H, B);
      hipLaunchKernelGGL(k_gray, grid, dim3(256), 0, st, d_src, channels, src_img_stride, src_row_stride, c->gray);
      if (c->up) {
         // pyramid.cpp:267-271: the first level is the 2x up-sampled image (doubleImage, helpers.cpp:297-329)
         const dim3 g2((c->upimg.cols + 255) / 256, c->upimg.rows, B);
         hipLaunchKernelGGL(k_double, g2, dim3(256), 0, st, c->gray, c->upimg);
      }
      const DPlane &first = c->up ? c->upimg : c->gray;
      if (direct) HIP_TRY(hipMemcpyAsync(c->L[0].p, first.p, (size_t)B * first.img_stride * 4, hipMemcpyDeviceToDevice, st));
      if (!c->oct.empty() && c->pyr_K[0] > 0) {
         // pyramid.cpp:276-280 initial blur 0.5 -> initialSigma
         const int tb = tm.begin(T_BLURHESS, 0);   // initial blur: not counted in the 12N launches (bytes 0)
         launch_blur_hess<true, false, false>(c, first, c->L[0], none, none, ptaps + c->pyr_tap_off[0], c->pyr_K[0], 0.0f, B);
         tm.end(tb);
      }
   }
   tm.end(t);
   float *pout = planes_out;
   for (size_t o = 0; o < c->oct.size(); o++) {
      const OctGeom &g = c->oct[o];
      const size_t planeF = (size_t)B * g.rows * g.pitch;
      DPlane Lo[5], Ro[5];
      for (int l = 0; l < 3; l++) Lo[l] = c->L[o * 3 + l];
      Lo[3] = make_plane(c->b_L3.as<float>(), g.rows, g.cols, g.pitch);
      Lo[4] = none;
      if (keep_all_planes) Lo[4] = make_plane(c->b_stage.as<float>(), g.rows, g.cols, g.pitch);
      for (int l = 0; l < 5; l++) Ro[l] = make_plane(c->b_R.as<float>() + l * planeF, g.rows, g.cols, g.pitch);
      t = tm.begin(T_PYR);
      // R0 = hessianResponse(L0) (pyramid.cpp:230) is fused into the first blur launch when the
      // marching kernel handles it (default sigmas: K = 9); otherwise a separate pass.
      const bool fuse_r0 = c->pyr_march;
      if (!fuse_r0) {
         const dim3 grid((g.cols + 255) / 256, g.rows, B);
         hipLaunchKernelGGL(k_hess, grid, dim3(256), 0, st, Lo[0], Ro[0], sc.norm2[0]);
      }
      const bool has_next = o + 1 < c->oct.size();
      for (int i = 1; i <= 4; i++) {
         // algorithmic bytes of this launch: 12 N, + 8 N when it also produces R0
         // (read L0 + write R0 of the stand-alone pass), + 2 N for the fused decimation
         double bytes = 12.0 * (double)B * g.rows * g.cols;
         if (i == 1 && fuse_r0) bytes += 8.0 * (double)B * g.rows * g.cols;
         if (i == 3 && has_next) bytes += 2.0 * (double)B * g.rows * g.cols;
         const float *taps = ptaps + c->pyr_tap_off[i];
         const int K = c->pyr_K[i];
         const int tb = tm.begin(T_BLURHESS, bytes);
         if (i == 1 && fuse_r0) launch_march<9, true, true, false, true>(c, Lo[0], Lo[1], Ro[1], none, taps, sc.norm2[1], B, Ro[0], sc.norm2[0]);
         else if (i < 3) launch_blur_hess<true, true, false>(c, Lo[i - 1], Lo[i], Ro[i], none, taps, K, sc.norm2[i], B);
         else if (i == 3) {
            if (has_next) launch_blur_hess<true, true, true>(c, Lo[2], Lo[3], Ro[3], c->L[(o + 1) * 3], taps, K, sc.norm2[3], B);
            else launch_blur_hess<true, true, false>(c, Lo[2], Lo[3], Ro[3], none, taps, K, sc.norm2[3], B);
         } else {
            if (keep_all_planes) launch_blur_hess<true, true, false>(c, Lo[3], Lo[4], Ro[4], none, taps, K, sc.norm2[4], B);
            else launch_blur_hess<false, true, false>(c, Lo[3], none, Ro[4], none, taps, K, sc.norm2[4], B);
         }
         tm.end(tb);
      }
      tm.end(t);
      if (planes_out) {
         // stage API (B == 1): copy L0..L4, R0..R4 tightly packed
         for (int l = 0; l < 5; l++) {
            HIP_TRY(hipMemcpy2DAsync(pout, (size_t)g.cols * 4, Lo[l].p, (size_t)g.pitch * 4, (size_t)g.cols * 4, g.rows, hipMemcpyDeviceToHost, st));
            pout += (size_t)g.rows * g.cols;
         }
         for (int l = 0; l < 5; l++) {
            HIP_TRY(hipMemcpy2DAsync(pout, (size_t)g.cols * 4, Ro[l].p, (size_t)g.pitch * 4, (size_t)g.cols * 4, g.rows, hipMemcpyDeviceToHost, st));
            pout += (size_t)g.rows * g.cols;
         }
      }
      // ---- detection on this octave ----
      t = tm.begin(T_DET);
      HIP_TRY(hipMemsetAsync(cnt + 0, 0, 4, st));
      HIP_TRY(hipMemcpyAsync(cnt + 32 + o, cnt + 1, 4, hipMemcpyDeviceToDevice, st));
      OctaveCtx oc;
      for (int l = 0; l < 5; l++) { oc.R[l] = Ro[l]; oc.L[l] = Lo[l]; oc.sigma[l] = sc.level_sigma[l]; }
      oc.pixelDistance = c->consts.pd0 * (float)(1 << o);   // pyramid.cpp:288: doubles per octave
      oc.octave = (int)o;
      oc.map = c->b_map.as<uint32_t>();
      // a fresh epoch for this pass (counting down; the all-ones epoch is the fill value): refill when they have run out
      if (c->map_epoch == 0) { HIP_TRY(hipMemsetAsync(c->b_map.p, 0xFF, c->b_map.bytes, st)); c->map_epoch = c->map_kbits < 32 ? (0xffffffffu >> c->map_kbits) : 0u; }
      if (c->map_epoch > 0) c->map_epoch--;
      oc.map_epoch = c->map_kbits < 32 ? (c->map_epoch << c->map_kbits) : 0u;
      oc.word_base = g.word_base;
      oc.words_per_image = c->words_per_image;
      oc.words_per_row = g.words_per_row;
      if (g.rows > 2 * HS_BORDER && g.cols > 2 * HS_BORDER) {
         FivePlanes fp;
         for (int l = 0; l < 5; l++) fp.R[l] = Ro[l];
         // bands of 128 rows (a band re-reads 4 rows of halo and starts with two row loads nothing overlaps: 32 / 64 / 128 / 256 rows measured
         // 25.0 / 23.4 / 22.2 / 23.5 ms for the detection stage of 256 UHD images); shorter bands when that would leave the chip short of wavefronts
         const int strips = (g.cols + EXM_STRIP - 1) / EXM_STRIP;
         auto waves_at = [&](int rows_per_band) { return (long long)strips * ((g.rows + rows_per_band - 1) / rows_per_band) * B; };
         const int band = c->force_exband > 0 ? c->force_exband : (waves_at(128) >= 4096 ? 128 : (waves_at(64) >= 4096 ? 64 : 32));
         const dim3 grid(strips, (g.rows + band - 1) / band, B);
         const int te = tm.begin(T_EXTREMA, 20.0 * (double)B * g.rows * g.cols);
         hipLaunchKernelGGL(k_extrema_march, grid, dim3(64), 0, st, fp, c->consts.positiveThreshold, c->consts.negativeThreshold, s.cl, band);
         tm.end(te);
         hipLaunchKernelGGL(k_localize, dim3(HS_GRID_LOC), dim3(256), 0, st, oc, s.cl, s.rl, c->consts);
         hipLaunchKernelGGL(k_dedupe, dim3(HS_GRID_DED), dim3(256), 0, st, oc, s.rl, (const uint32_t *)(cnt + 32 + o),
                            c->b_bitmask.as<unsigned long long>());
      }
      tm.end(t);
   }
   // ---- ordering ----
   t = tm.begin(T_DET);
   const long long total_words = (long long)B * c->words_per_image;
   LoadPopc lp; lp.p = c->b_bitmask.as<unsigned long long>();
   exclusive_scan(c, lp, total_words, c->b_prefix.as<uint32_t>(), cnt + 3);
   // the records at their ranks as 32-byte items (in the candidate buffer: its last reader, the last octave's k_localize, is done), then dealt out
   HessItem *items = reinterpret_cast<HessItem *>(c->b_cand.p);
   static_assert(sizeof(HessItem) == 32 && sizeof(CandRec) >= sizeof(HessItem), "the items fit the candidate slots (cand_cap >= cap)");
   hipLaunchKernelGGL(k_scatter_ordered, dim3(HS_GRID_SCAT), dim3(256), 0, st, s.rl, (const unsigned long long *)c->b_bitmask.p,
                      (const uint32_t *)c->b_prefix.p, items, s.hl.cap);
   hipLaunchKernelGGL(k_hess_deal, dim3(HS_GRID_SCAT), dim3(256), 0, st, (const HessItem *)items, (const uint32_t *)(cnt + 3), s.hl);
   hipLaunchKernelGGL(k_image_counts, dim3((B + 1 + 63) / 64), dim3(64), 0, st, (const uint32_t *)c->b_prefix.p,
                      c->words_per_image, B, (const uint32_t *)(cnt + 3), c->b_starts.as<int32_t>());
   // per image: upper bound of the T' rows its huge windows (P > 512) need, known from the scales alone
   HIP_TRY(hipMemsetAsync(c->b_starts.as<int32_t>() + 2 * (B + 1), 0, (size_t)(B + 2) * 4, st));
   hipLaunchKernelGGL(k_image_large_rows, dim3(512), dim3(256), 0, st, s.hl, (const uint32_t *)(cnt + 3), c->consts.mrSize,
                      c->b_starts.as<uint32_t>() + 2 * (B + 1), B);
   tm.end(t);
}

__global__ void k_desc_starts(const int32_t *__restrict__ hess_starts, int nimg, const uint32_t *__restrict__ rank,
                              const uint32_t *__restrict__ n_hess, const uint32_t *__restrict__ total_desc, int32_t *__restrict__ out)
{
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b > nimg) return;
   const uint32_t hs = (uint32_t)hess_starts[b];
   out[b] = (b == nimg || hs >= *n_hess) ? (int32_t)*total_desc : (int32_t)rank[hs];
}

void collect_timings(hesaff_ctx *c, StageTimer &tm, int B)
{
   hesaff_timings &t = c->tm;
   memset(&t, 0, sizeof t);
   for (const EvPair &p : tm.pairs) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, p.a, p.b);
      switch (p.kind) {
         case T_PYR: t.pyramid_ms += ms; break;
         case T_DET: t.detect_ms += ms; break;
         case T_AFF: t.affine_ms += ms; break;
         case T_PATCH: t.patch_ms += ms; break;
         case T_SIFT: t.sift_ms += ms; break;
         case T_PACK: t.pack_ms += ms; break;
         case T_TOTAL: t.total_ms += ms; break;
         case T_BLURHESS:
            if (p.bytes > 0) { t.blur_hess_ms += ms; t.blur_hess_launches++; t.blur_hess_bytes += p.bytes; }
            break;
         case T_EXTREMA: t.extrema_ms += ms; t.extrema_launches++; t.extrema_bytes += p.bytes; break;
      }
   }
   double sumN = 0;
   for (const OctGeom &g : c->oct) sumN += (double)g.rows * g.cols;
   t.pyramid_bytes = (double)B * (5.0 * c->H * c->W + 58.0 * sumN);
   t.export_ms = c->export_ms; t.export_rows = c->export_rows;   // (run_chunks keeps them across the batches of a list)   // (+ the up-sampling pass when upscaleInputImage is set: not counted)
}

// The descriptor kernels (kernels_sift.h) over n patches in HBM.
// sift_slice > 0: the group's keypoints in slices of that many, the four kernels back to back per slice, so that what a kernel
// writes (mean / variance, gradient pairs, histograms) and the patches the slice's first kernel fetched are still in the device's
// 256 MB memory-side cache when the next kernel of the slice reads them (sweep in profiles/r06_notes.md).
void launch_sift_range(hesaff_ctx *c, hipStream_t ss, const SiftIO &so, uint32_t n, float2 *vo)
{
   const uint32_t nb64 = (n + 63) / 64;
   hipLaunchKernelGGL(k_sift_meanvar, dim3((n + SM_KP - 1) / SM_KP), dim3(64), 0, ss, so, c->tables);
   hipLaunchKernelGGL(k_sift_grad, dim3(c->sgrad_grid ? std::min(n, c->sgrad_grid) : n), dim3(256), 0, ss, so, c->tables, vo);
   hipLaunchKernelGGL(k_sift_hist, dim3(std::min<uint32_t>((n + 3) / 4, c->g_shist)), dim3(64), 0, ss, so, c->tables, (const float2 *)vo);
   hipLaunchKernelGGL(k_sift_quantize, dim3(nb64), dim3(64), 0, ss, so, c->consts);
}

void launch_sift(hesaff_ctx *c, hipStream_t ss, const SiftIO &so, uint32_t n, float2 *vo)
{
   const uint32_t slice = c->sift_slice;
   if (slice == 0 || slice >= n) { launch_sift_range(c, ss, so, n, vo); return; }
   for (uint32_t lo = 0; lo < n; lo += slice) {
      const uint32_t m = std::min(slice, n - lo);
      SiftIO s = so;
      s.patches = so.patches + (size_t)lo * HS_PATCH_PIX;
      s.h_lo = so.h_lo + lo; s.h_hi = s.h_lo + m;
      // the intermediates are indexed relative to h_lo: in ring mode every slice uses the group buffers' first slice-sized piece
      // (the kernels of one stream run one after the other; a keypoint's zero items of the pair block are never written)
      const size_t at = c->sift_slice_ring ? 0 : lo;
      s.meanvar = so.meanvar + 2 * at;
      s.vec = so.vec + 128 * at;
      launch_sift_range(c, ss, s, m, vo + at * HS_VO_PITCH);
   }
}

// per-group patch / descriptor buffers (two slots): sized once per batch for the largest group
void ensure_group_buffers(hesaff_ctx *c, uint32_t n)
{
   // The patch buffers rotate over HS_NSLOT slots (the patch stage fills one while the descriptor stage reads the others).  The
   // descriptor stage's own intermediates - gradient pairs (10.4 KB per keypoint), histograms, mean / variance - live and die on its
   // stream: when both descriptor streams are one HIP stream (the product), one copy of them serves every group.
   const int dslots = (c->sift_stream == c->sift_stream2 && !c->no_overlap) ? 1 : HS_NSLOT;
   for (int slot = 0; slot < HS_NSLOT; slot++) {
      c->b_patches2[slot].ensure_grow((size_t)n * HS_PATCH_PIX * 4);
      if (slot >= dslots) continue;
      c->b_siftvec2[slot].ensure_grow((size_t)n * 128 * 4);
      c->b_meanvar2[slot].ensure_grow((size_t)n * 2 * 4);
      // the (mask*grad, o) pairs of pixels outside the circular mask stay (0, 0): zero-fill on (re)allocation
      const void *before = c->b_siftvo2[slot].p;
      const size_t bytes_before = c->b_siftvo2[slot].bytes;
      c->b_siftvo2[slot].ensure_grow((size_t)n * HS_VO_PITCH * 8 + 64);
      if (c->b_siftvo2[slot].p != before || c->b_siftvo2[slot].bytes != bytes_before)
         HIP_TRY(hipMemsetAsync(c->b_siftvo2[slot].p, 0, c->b_siftvo2[slot].bytes, c->stream));
   }
}

// Whole hot path on a device-resident batch.  Leaves ordered KeyRec records in b_out and
// per-image start offsets (hessian: b_starts[0..B], desc: b_starts[B+1..2B+1]).
void run_batch(hesaff_ctx *c, const uint8_t *d_src, int channels, long long src_img_stride, int src_row_stride, int B, int H, int W)
{
   plan(c, B, H, W);
   c->ev_used = 0;
   StageTimer tm(c);
   Lists s = make_lists(c);
   hipStream_t st = c->stream;
   uint32_t *cnt = s.counters;
   const int tt = tm.begin(T_TOTAL);
   run_detection(c, d_src, channels, src_img_stride, src_row_stride, B, s, tm, false, nullptr);

   int t;
   PlaneTab pt;
   memset(&pt, 0, sizeof pt);
   uint32_t n_hess_host = 0;   // Hessian keypoints of the batch (known after the host round trip below)
   for (size_t o = 0; o < c->oct.size(); o++)
      for (int l = 0; l < 3; l++) pt.L[o][l] = c->L[o * 3 + l];
   {
      // The one host round trip of a batch: per-image Hessian counts + large-window row bounds.  The bin kernels
      // only extract the 41x41 patches (to HBM); the descriptor runs as four kernels with the parallel axis each
      // part wants (kernels_sift.h).  Images are processed in groups so that the patch buffers stay bounded.
      int32_t *hs_p = (int32_t *)c->h_small_mid.ensure(((size_t)3 * (B + 1) + 1) * 4);   // pinned: see hesaff_ctx::HostSmall
      struct { int32_t *q; int32_t *data() const { return q; } int32_t &operator[](size_t i) const { return q[i]; } } hs{hs_p};
      HIP_TRY(hipMemcpyAsync(hs.data(), c->b_starts.p, ((size_t)3 * (B + 1) + 1) * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(c->ev_detect_done, st));
      const double dbg_ca = c->debug ? thread_cpu_ms() : 0.0;
      hs_wait_event(c->ev_detect_done);   // sleeps: no core spins while the detection stage runs
      if (c->debug) fprintf(stderr, "[hesaff] run_batch: caller's CPU inside the wait for the detection stage %.2f ms\n", thread_cpu_ms() - dbg_ca);
      if ((uint32_t)hs[B] > c->cap) throw HsError(HESAFF_ERR_CAPACITY, "keypoint capacity exceeded; raise hesaff_params.max_kpts_per_mpx");
      n_hess_host = (uint32_t)hs[B];
      const uint32_t *lrows = (const uint32_t *)hs.data() + 2 * (B + 1);
      c->batch_max_p = (int)lrows[B + 1];   // largest huge window of the batch (0: none)
      // image groups [h_lo, h_hi) of at most group_kpts keypoints: about 16 groups per batch keep the
      // three-stage pipeline full, between 300 k (launch overheads) and 1.2 M keypoints (buffer size);
      // the T' rows of a group's huge windows must fit the row buffer (a single image may exceed it: the buffer grows)
      // (the group size itself hardly matters: 0.6 / 0.9 / 1.2 / 1.8 / 2.4 M keypoints per group at B = 256, shuffled: 810 / 823 / 816 /
      //  818 / 816 ms; what matters is that the buffers of a group stay modest: 33 KB per keypoint of a group)
      const uint32_t group_kpts = c->sift_group_kpts ? c->sift_group_kpts : std::min<uint32_t>(std::max<uint32_t>((uint32_t)hs[B] / 16u, 300000u), 1200000u);
      struct Group { uint32_t lo, hi, large_rows; };
      std::vector<Group> groups;
      uint32_t max_n = 0;
      for (int g0 = 0; g0 < B;) {
         int g1 = g0 + 1;
         unsigned long long rows = lrows[g0];
         // tapered schedule: the first groups grow (1/8, 1/8, 1/4, 1/2 of the limit) and the last ones shrink the same way, so that
         // the pipeline's fill (affine shape of the first group alone on the device) and drain (descriptors of the last) are short
         uint32_t limit = group_kpts;
         if (c->taper_groups) {
            const uint32_t done = (uint32_t)hs[g0], left = (uint32_t)hs[B] - done;
            limit = std::min(group_kpts, std::max(group_kpts / 8u, std::min(done, left / 2u)));
         }
         while (g1 < B && (uint32_t)(hs[g1 + 1] - hs[g0]) <= limit && rows + lrows[g1] <= c->trows_rows) { rows += lrows[g1]; g1++; }
         if (rows > 0xffffffffull) throw HsError(HESAFF_ERR_NOMEM, "window rows of one image exceed 32 bits");
         if (hs[g1] > hs[g0]) {
            groups.push_back({(uint32_t)hs[g0], (uint32_t)hs[g1], (uint32_t)rows});
            max_n = std::max(max_n, (uint32_t)(hs[g1] - hs[g0]));
         }
         g0 = g1;
      }
      while (c->ev_aff.size() < groups.size()) {
         hipEvent_t e;
         HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
         c->ev_aff.push_back(e);
      }
      if (max_n) ensure_group_buffers(c, max_n);
      // Software pipeline over image groups, one stream per stage:
      //   affine shape of group g+1 (aff_stream)  |  patch extraction of group g (main + side
      //   streams, latency-bound)  |  descriptor kernels of groups g-1 and g-2 (sift_stream, sift_stream2: the
      //   HBM-bound mean / variance pass of one group beside the gradient and histogram kernels of the other).
      // Three patch/descriptor buffer slots rotate.
      hipStream_t as = c->no_overlap ? st : c->aff_stream;
      if (as != st) HIP_TRY(hipStreamWaitEvent(as, c->ev_detect_done, 0));
      auto launch_affine = [&](size_t gi) {
         const int ta = tm.begin(T_AFF, 0, as);
         const uint32_t agrid = std::min<uint32_t>((groups[gi].hi - groups[gi].lo + HS_AFFP_G - 1) / HS_AFFP_G, (uint32_t)c->n_cu * c->aff_blocks_per_cu);
         hipLaunchKernelGGL(k_affine, dim3(agrid), dim3(64), 0, as, pt, s.hl, groups[gi].lo, groups[gi].hi, (const uint32_t *)(cnt + 3), c->tables, c->consts, s.ao);
         tm.end(ta);
         if (as != st) HIP_TRY(hipEventRecord(c->ev_aff[gi], as));
      };
      if (!groups.empty()) launch_affine(0);
      bool slot_used[HS_NSLOT] = {};
      for (size_t gi = 0; gi < groups.size(); gi++) {
         const uint32_t h_lo = groups[gi].lo, h_hi = groups[gi].hi, n = h_hi - h_lo;
         if (gi + 1 < groups.size()) launch_affine(gi + 1);
         if (as != st) HIP_TRY(hipStreamWaitEvent(st, c->ev_aff[gi], 0));
         const int slot = (int)(gi % HS_NSLOT);
         if (slot_used[slot]) HIP_TRY(hipStreamWaitEvent(st, c->ev_sift_done[slot], 0));   // the slot's previous descriptors are finished
         t = tm.begin(T_PATCH);
         HIP_TRY(hipMemsetAsync(cnt + 8, 0, 24 * 4, st));   // bin counts [8..13) and work counters [24..29)
         // (the group's end travels as a kernel argument: a 4-byte copy from pageable memory would make the host wait
         //  here until the stream has drained, once per group)
         hipLaunchKernelGGL(k_prepare_patch, dim3(1024), dim3(256), 0, st, s.hl, h_lo, h_hi, (const uint32_t *)(cnt + 3), s.ao, H, W, c->consts,
                            c->tables, s.pw);
         run_patch_stage(c, s, c->gray, c->b_patches2[slot].as<float>(), h_lo, groups[gi].large_rows, &pt);
         tm.end(t);
         HIP_TRY(hipEventRecord(c->ev_extract_done[slot], st));
         hipStream_t ss = c->no_overlap ? st : ((c->sift2 && (gi & 1)) ? c->sift_stream2 : c->sift_stream);
         if (ss != st) HIP_TRY(hipStreamWaitEvent(ss, c->ev_extract_done[slot], 0));
         SiftIO so;
         const int dslot = (c->sift_stream == c->sift_stream2 && !c->no_overlap) ? 0 : slot;   // ensure_group_buffers
         so.patches = c->b_patches2[slot].as<float>(); so.alive = s.pw.alive; so.meanvar = c->b_meanvar2[dslot].as<float>();
         so.vec = c->b_siftvec2[dslot].as<float>(); so.desc = c->b_desc.as<uint8_t>(); so.h_lo = h_lo; so.h_hi = h_hi;
         const int ts = tm.begin(T_SIFT, 0, ss);
         launch_sift(c, ss, so, n, c->b_siftvo2[dslot].as<float2>());
         tm.end(ts);
         HIP_TRY(hipEventRecord(c->ev_sift_done[slot], ss));
         slot_used[slot] = true;
      }
      for (int sl = 0; sl < HS_NSLOT; sl++)
         if (slot_used[sl]) HIP_TRY(hipStreamWaitEvent(st, c->ev_sift_done[sl], 0));
   }
   t = tm.begin(T_PACK);
   // final stable compaction (hesaff.cpp:87: keys.push_back in detection order): exclusive scan of the alive flags of the batch's
   // Hessian keypoints (alive[] is rewritten for h < n_hess each batch; the host knows n_hess since the round trip after detection -
   // the scan used to run over the whole capacity, 85 M flags for 31 M keypoints, behind a kernel that cleared the tail)
   LoadFlagI32 lf; lf.p = s.pw.alive;
   exclusive_scan(c, lf, (long long)n_hess_host, c->b_rank.as<uint32_t>(), cnt + 4);
   hipLaunchKernelGGL(k_pack, dim3(HS_GRID_PACK), dim3(256), 0, st, s.hl, (const uint32_t *)(cnt + 3), s.pw, (const uint32_t *)c->b_rank.p,
                      (const uint8_t *)c->b_desc.p, c->b_out.as<KeyRec>());
   hipLaunchKernelGGL(k_desc_starts, dim3((B + 1 + 63) / 64), dim3(64), 0, st, (const int32_t *)c->b_starts.p, B,
                      (const uint32_t *)c->b_rank.p, (const uint32_t *)(cnt + 3), (const uint32_t *)(cnt + 4),
                      c->b_starts.as<int32_t>() + (B + 1));
   tm.end(t);
   tm.end(tt);
   c->h_starts.q = (int32_t *)c->h_small_end.ensure(((size_t)2 * (B + 1) + 8) * 4);
   HIP_TRY(hipMemcpyAsync(c->h_starts.data(), c->b_starts.p, (size_t)2 * (B + 1) * 4, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipMemcpyAsync(c->h_starts.data() + 2 * (B + 1), cnt, 8 * 4, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipEventRecord(c->ev_batch_done, st));
   const double dbg_cb = c->debug ? thread_cpu_ms() : 0.0;
   hs_wait_event(c->ev_batch_done);
   if (c->debug) fprintf(stderr, "[hesaff] run_batch: caller's CPU inside the wait for the end of the batch %.2f ms\n", thread_cpu_ms() - dbg_cb);
   HIP_TRY(hipGetLastError());
   if (c->profiling) collect_timings(c, tm, B);
   const int32_t *cn = c->h_starts.data() + 2 * (B + 1);
   if (cn[2] != 0 || (uint32_t)cn[1] > c->cap)
      throw HsError(HESAFF_ERR_CAPACITY, "keypoint capacity exceeded; raise hesaff_params.max_kpts_per_mpx");
   if (cn[6] != 0) throw HsError(HESAFF_ERR_NOMEM, "large-window row buffer exceeded (internal bound violated)");
}

// ---- exportKeypoints on the device (kernels_export.h) ----
// Lengths of the n rows of `keys`, their 64-row offsets, and the byte offset of every image's first row (d_starts: B + 1 row
// starts on the device).  The host waits for the stream here: it needs the byte counts to size the copy out.
unsigned long long export_text_prepare(hesaff_ctx *c, const KeyRec *keys, uint32_t n, const int32_t *d_starts, int B,
                                       std::vector<unsigned long long> &img_off)
{
   img_off.assign((size_t)B + 1, 0ull);
   if (n == 0) return 0ull;
   hipStream_t st = c->stream;
   const uint32_t nblk = (n + EX_ROWS - 1) / EX_ROWS;
   c->b_ex_len.ensure_grow(((size_t)n + 64) * 2);
   c->b_ex_sums.ensure_grow((size_t)nblk * 4);
   c->b_ex_off.ensure_grow(((size_t)nblk + 1) * 8);
   c->b_ex_imgoff.ensure(((size_t)B + 1) * 8);
   hipLaunchKernelGGL(k_text_len, dim3((n + 255) / 256), dim3(256), 0, st, keys, n, c->par.mrSize, c->b_ex_len.as<uint16_t>(), c->b_ex_sums.as<uint32_t>());
   hipLaunchKernelGGL(k_text_scan, dim3(1), dim3(1024), 0, st, (const uint32_t *)c->b_ex_sums.p, nblk, c->b_ex_off.as<unsigned long long>());
   hipLaunchKernelGGL(k_text_imgoff, dim3((B + 1 + 63) / 64), dim3(64), 0, st, d_starts, B, (const uint16_t *)c->b_ex_len.p,
                      (const unsigned long long *)c->b_ex_off.p, c->b_ex_imgoff.as<unsigned long long>());
   // into pinned memory, then a sleep on the blocking-sync event (a copy into the pageable vector would spin in the runtime)
   unsigned long long *ho = (unsigned long long *)c->h_small_exp.ensure(((size_t)B + 1) * 8);
   HIP_TRY(hipMemcpyAsync(ho, c->b_ex_imgoff.p, ((size_t)B + 1) * 8, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipEventRecord(c->ev_batch_done, st));      // (run_batch has returned: the event is free)
   hs_wait_event(c->ev_batch_done);
   HIP_TRY(hipGetLastError());
   memcpy(img_off.data(), ho, ((size_t)B + 1) * 8);
   return img_off[(size_t)B];
}

// ... and the rows themselves into d_text (export_text_prepare's byte count), on the main stream
void export_text_write(hesaff_ctx *c, const KeyRec *keys, uint32_t n, char *d_text)
{
   if (n == 0) return;
   hipLaunchKernelGGL(k_text_write, dim3((n + EX_ROWS - 1) / EX_ROWS), dim3(EX_ROWS), 0, c->stream, keys, n, c->par.mrSize, (const uint16_t *)c->b_ex_len.p,
                      (const unsigned long long *)c->b_ex_off.p, d_text);
}

// hesaff_jpeg_layout (what the host's entropy stage reports; from a caller, so checked) -> the kernels' geometry
JpegGeom make_jpeg_geom(const hesaff_jpeg_layout &L)
{
   JpegGeom g;
   memset(&g, 0, sizeof g);
   if (L.width < 1 || L.height < 1 || L.width > 65535 || L.height > 65535 || (L.channels != 1 && L.channels != 3))
      throw HsError(HESAFF_ERR_ARG, "bad JPEG layout");
   g.W = L.width; g.H = L.height; g.nc = L.channels;
   unsigned long long coef = HESAFF_JPEG_BLOB_HEADER, plane = 0, blocks = 0;
   for (int i = 0; i < g.nc; i++) {
      const long long bw = L.bw[i], bh = L.bh[i], cw = L.cw[i], chgt = L.chgt[i], hx = L.hx[i], vx = L.vx[i];
      if (bw < 1 || bh < 1 || bw > 8192 || bh > 8192 || cw < 1 || chgt < 1 || cw > bw * 8 || chgt > bh * 8 || hx < 1 || hx > 4 || vx < 1 || vx > 4 ||
          cw * hx < L.width || chgt * vx < L.height || (g.nc == 1 && (hx != 1 || vx != 1)))
         throw HsError(HESAFF_ERR_ARG, "bad JPEG layout");
      g.bw[i] = (int)bw; g.bh[i] = (int)bh; g.cw[i] = (int)cw; g.chgt[i] = (int)chgt; g.hx[i] = (int)hx; g.vx[i] = (int)vx;
      // jdsample.c's choice of method (jpeg_decode.cpp)
      g.mode[i] = (hx == 1 && vx == 1) ? JPEG_UP_NONE : (hx == 2 && vx == 1 && cw > 2) ? JPEG_UP_H2V1 : (hx == 2 && vx == 2 && cw > 2) ? JPEG_UP_H2V2
                  : (hx == 1 && vx == 2) ? JPEG_UP_H1V2 : JPEG_UP_REPLICATE;
      g.blocks[i] = (unsigned int)(bw * bh);
      g.coef_off[i] = coef; g.plane_off[i] = plane;
      coef += (unsigned long long)(bw * bh) * 128;
      plane += (unsigned long long)(bw * bh) * 64;
      blocks += (unsigned long long)(bw * bh);
   }
   if (blocks > 0x7fffffffull) throw HsError(HESAFF_ERR_ARG, "bad JPEG layout");
   g.blob_bytes = coef; g.plane_bytes = plane; g.blocks_per_image = (unsigned int)blocks;
   return g;
}

// coefficient blobs of B images of one layout (device) -> their pixels, W x H x channels bytes each, out_img_stride apart
void jpeg_pixels(hesaff_ctx *c, const uint8_t *d_blobs, const JpegGeom &g, int B, uint8_t *d_out, size_t out_img_stride, hipStream_t st)
{
   if (B < 1) return;
   c->b_jplane.ensure_grow((size_t)g.plane_bytes * (size_t)B);
   const unsigned long long total = (unsigned long long)B * g.blocks_per_image;
   hipLaunchKernelGGL(k_jpeg_idct, dim3((unsigned)std::min<unsigned long long>((total + 255) / 256, 1u << 20)), dim3(256), 0, st, d_blobs,
                      c->b_jplane.as<uint8_t>(), g, B);
   hipLaunchKernelGGL(k_jpeg_pixels, dim3(((g.W + 3) / 4 + 255) / 256, g.H, B), dim3(256), 0, st, d_blobs, (const uint8_t *)c->b_jplane.as<uint8_t>(), d_out, g,
                      (unsigned long long)out_img_stride);
}

void export_bin_rows(hesaff_ctx *c, const KeyRec *keys, uint32_t n, char *d_bin)
{
   if (n == 0) return;
   hipLaunchKernelGGL(k_bin_rows, dim3(std::min<uint32_t>((n + 7) / 8, 4096u)), dim3(256), 0, c->stream, keys, n, c->par.mrSize, (uint32_t *)d_bin);
}

} // namespace
#include "capi_impl.h"
