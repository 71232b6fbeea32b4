// chunk_engine.h -- the host-only half of the chunk engine (capi_impl.h): the chunk types, the ring of pinned result blocks
// and FileIO, the decoder / writer thread pools of hesaff_process_files.  Nothing here touches HIP, so that this code - every
// mutex, condition variable and hand-over between threads of the file pipeline - also runs under ThreadSanitizer and
// AddressSanitizer on a CPU with a mock device loop (tests/native/engine_sanitize.cpp, tests/test_host_sanitize.py).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <pthread.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/hesaff_amd.h"

namespace hesaff_engine {

// Pinned result blocks in rotation: a block is busy from the copy out of a chunk until its consumer gives it back.
struct BlockRing {
   std::mutex mu;
   std::condition_variable cv;
   std::vector<char> busy;
   void reset(int n)
   {
      std::lock_guard<std::mutex> lk(mu);
      busy.assign((size_t)(n > 0 ? n : 0), 0);
   }
   int acquire()   // waits for a free block
   {
      std::unique_lock<std::mutex> lk(mu);
      int blk = -1;
      cv.wait(lk, [&] {
         for (size_t i = 0; i < busy.size(); i++)
            if (!busy[i]) { blk = (int)i; return true; }
         return false;
      });
      busy[(size_t)blk] = 1;
      return blk;
   }
   void release(int block)
   {
      {
         std::lock_guard<std::mutex> lk(mu);
         if (block >= 0 && block < (int)busy.size()) busy[(size_t)block] = 0;
      }
      cv.notify_all();
   }
};

struct HostChunk {                // at most max_batch images of one geometry
   int W = 0, H = 0, ch = 1;
   std::vector<const uint8_t *> data;
   std::vector<size_t> stride;    // bytes between rows
   std::vector<int> index;        // the caller's image numbers
   // JPEG files travel as entropy-decoded coefficients (hesaff_read_jpeg_coefficients): data[b] is then image b's blob of blob_bytes
   // bytes, all images of the chunk have the layout `jpeg`, and the device makes the W x H x ch pixels (kernels_jpeg.h)
   size_t blob_bytes = 0;         // 0: data[b] are pixels
   hesaff_jpeg_layout jpeg = {};
   bool pinned = false;           // every data[b] is page-locked memory of the context (PinHooks): copied to the device from where it is
};

// Page-locked host buffers for the images a FileIO reads (the context's hipHostMalloc behind two plain function pointers: nothing here
// touches HIP).  alloc may return nullptr (no room, or the caller pins nothing): the image then lives in ordinary memory and travels
// through the context's pinned staging buffer.  release: the buffer goes back to the context, which keeps it pinned for the next list.
struct PinHooks {
   void *(*alloc)(size_t bytes, void *user) = nullptr;
   void (*release)(void *p, size_t bytes, void *user) = nullptr;
   void *user = nullptr;
};
inline bool same_jpeg_layout(const hesaff_jpeg_layout &a, const hesaff_jpeg_layout &b) { return memcmp(&a, &b, sizeof a) == 0; }

// what a consumer wants copied to the host for every chunk
enum { WANT_KEYS = 1,    // the hesaff_keypoint records (hesaff.cpp:41-48)
       WANT_TEXT = 2,    // the rows of the .hesaff.sift files, formatted on the device (kernels_export.h; hesaff.cpp:124-128)
       WANT_BIN = 4 };   // the 148-byte rows of the binary sidecar

struct ChunkDone {
   const HostChunk *chunk;
   const int32_t *count_hessian, *count_desc;   // per image of the chunk
   const size_t *key_off;                       // first record of each image inside keys (= first row inside bin)
   const hesaff_keypoint *keys;                 // WANT_KEYS; pinned host memory: valid until the block is released / the next call
   int block;
   const char *text = nullptr;                     // WANT_TEXT: the rows of all images of the chunk back to back (no file headers)
   const unsigned long long *text_off = nullptr;   //            image b owns text[text_off[b] .. text_off[b + 1])
   const char *bin = nullptr;                      // WANT_BIN: image b's rows start at bin + key_off[b] * 148
};

struct ChunkIO {
   virtual bool next(HostChunk &out) = 0;       // staging thread, one call at a time, in order; false: no more chunks
   virtual void staged(const HostChunk &) {}    // staging thread: the chunk's pixels are in pinned memory, its sources may go
   virtual void done(const ChunkDone &) = 0;    // caller's thread, in order
   virtual int wants() const { return WANT_KEYS; }
   // images of the largest chunk that a chunk of `this_chunk` images with this geometry may be followed by (a list that starts with
   // small chunks says so here: the device plan and the pinned input buffers are then made once, for the large chunks)
   virtual int largest_chunk(int this_chunk) const { return this_chunk; }
   // caller's thread: the device refused this chunk (rc = HESAFF_ERR_ARG: image geometry, HESAFF_ERR_CAPACITY: more keypoints than
   // planned for).  true: noted per image, go on with the next chunk; false: the whole call fails with rc
   virtual bool failed(const HostChunk &, int /*rc*/) { return false; }
   virtual ~ChunkIO() {}
};

// hesaff_detect_batch / hesaff_detect_batch_cb: chunks of a caller-supplied image list, images of equal (width, height, channels)
// grouped in input order.  The arguments are validated by the caller (capi_impl.h: validate_image_list).
struct ArrayIO : ChunkIO {
   std::vector<HostChunk> chunks;
   size_t pos = 0;
   BlockRing *ring = nullptr;
   hesaff_result *results = nullptr;             // hesaff_detect_batch: filled in place
   hesaff_chunk_sink sink = nullptr;             // hesaff_detect_batch_cb
   void *user = nullptr;
   std::atomic<int> sink_rc{0};                 // written by done() on the caller's thread, read by next() on the staging thread
   ArrayIO(BlockRing *ring_, int max_batch, int n, const uint8_t *const *images, const int *widths, const int *heights, const int *strides,
           const int *channels)
      : ring(ring_)
   {
      std::vector<char> taken((size_t)n, 0);
      for (int i = 0; i < n; i++) {
         if (taken[(size_t)i]) continue;
         const int W = widths[i], H = heights[i], ch = channels ? channels[i] : 1;
         std::vector<int> grp;
         for (int j = i; j < n; j++)
            if (!taken[(size_t)j] && widths[j] == W && heights[j] == H && (channels ? channels[j] : 1) == ch) {
               grp.push_back(j);
               taken[(size_t)j] = 1;
            }
         // chunk sizes: max_batch, except that a long run starts and ends with smaller chunks (1/4, 1/2 of it): the first chunk's
         // copy in and the last chunk's copy out are the pipeline's fill and drain - nothing overlaps them
         const size_t mb = (size_t)(max_batch > 0 ? max_batch : 1), N = grp.size();
         std::vector<size_t> sizes;
         if (N >= 4 * mb && mb >= 8) {
            sizes.push_back(mb / 4); sizes.push_back(mb / 2);
            size_t left = N - mb / 4 - mb / 2 - (mb / 2 + mb / 4);
            while (left > 0) { const size_t t = std::min(mb, left); sizes.push_back(t); left -= t; }
            sizes.push_back(mb / 2); sizes.push_back(mb / 4);
         } else {
            for (size_t left = N; left > 0;) { const size_t t = std::min(mb, left); sizes.push_back(t); left -= t; }
         }
         size_t g0 = 0;
         for (size_t sz : sizes) {
            HostChunk k;
            k.W = W; k.H = H; k.ch = ch;
            for (size_t g = g0; g < g0 + sz; g++) {
               const int j = grp[g];
               k.index.push_back(j);
               k.data.push_back(images[j]);
               k.stride.push_back(strides ? (size_t)strides[j] : (size_t)W * ch);
            }
            chunks.push_back(std::move(k));
            g0 += sz;
         }
      }
   }
   bool next(HostChunk &out) override
   {
      if (pos >= chunks.size() || sink_rc.load() != 0) return false;
      out = chunks[pos++];
      return true;
   }
   int largest_chunk(int this_chunk) const override
   {
      if (pos == 0 || pos > chunks.size()) return this_chunk;
      const HostChunk &q = chunks[pos - 1];   // the chunk next() handed out last (staging thread, like next())
      size_t m = (size_t)this_chunk;
      for (size_t i = pos; i < chunks.size() && chunks[i].W == q.W && chunks[i].H == q.H && chunks[i].ch == q.ch; i++) m = std::max(m, chunks[i].data.size());
      return (int)m;
   }
   void done(const ChunkDone &d) override
   {
      const size_t B = d.chunk->index.size();
      if (results) {
         for (size_t b = 0; b < B; b++) {
            hesaff_result &r = results[d.chunk->index[b]];
            r.count_hessian = d.count_hessian[b]; r.count_desc = d.count_desc[b]; r.keys = d.keys + d.key_off[b];
         }
         return;
      }
      std::vector<hesaff_result> tmp(B);
      for (size_t b = 0; b < B; b++) { tmp[b].count_hessian = d.count_hessian[b]; tmp[b].count_desc = d.count_desc[b]; tmp[b].keys = d.keys + d.key_off[b]; }
      if (sink_rc.load() == 0) sink_rc.store(sink(user, (int)B, d.chunk->index.data(), tmp.data()));
      ring->release(d.block);
   }
};

// hesaff_process_files: decode -> chunks -> device -> write.  ONE pool of decode_threads + write_threads host threads serves both
// ends: a worker decodes the next image whenever the look-ahead window has room (cheap for PNM, and what keeps the device fed) and
// otherwise writes a finished image.  With separate pools the decoders of a PGM list idle while the writers - one write() per
// image, tens of MB into the page cache - are the slowest stage of the whole pipeline.
struct FileIO : ChunkIO {
   BlockRing *ring;
   int max_batch;
   int n;
   const char *const *paths, *const *out_paths;
   hesaff_file_status *status;
   float mrSize;
   int fmt;
   bool device_format;   // rows arrive formatted (ChunkDone::text / bin): the writers only write()
   int resume;           // hesaff_set_resume: an image whose complete output exists is not read (2: the rows of a text output are counted)
   bool nice_pool;       // the pool's threads run at nice 10 (hesaff_set_pool_priority: a CPU-starved plan)
   bool device_jpeg;     // JPEG files: entropy decoding only on the pool's threads, the pixels are made on the device
   PinHooks pin;         // where the readers' buffers come from (see blob_alloc)
   std::unordered_map<void *, size_t> pinned;   // buffers of `pin` that are out (with an image, or in blob_pool); under mu
   struct Img {          // state: 0 pending, 1 decoded, 2 unreadable, 3 handed on
      uint8_t *data = nullptr; int w = 0, h = 0, ch = 0; int state = 0;
      size_t blob_bytes = 0; hesaff_jpeg_layout jpeg = {};   // blob_bytes != 0: data is a coefficient blob
   };
   std::vector<Img> imgs;
   std::mutex mu;
   std::condition_variable cv_work, cv_img, cv_done;   // workers: a job may be available; next(): an image changed state; wait_writers()
   int next_decode = 0, consumed = 0, window = 0, pos = 0, chunks_out = 0;
   bool stop = false;
   struct Task { int index; const hesaff_keypoint *keys; int n; int chunk; const char *text; size_t text_len; const char *bin; };
   std::deque<Task> tasks;
   struct Open { int left; int block; };
   std::vector<Open> open_chunks;
   int tasks_in_flight = 0;
   std::vector<std::thread> workers;

   FileIO(BlockRing *ring_, int max_batch_, float mrSize_, int fmt_, int n_, const char *const *p, const char *const *o, hesaff_file_status *st,
          int dec_threads, int wr_threads, bool device_format_ = false, int resume_ = 0, bool device_jpeg_ = false, PinHooks pin_ = PinHooks(),
          bool nice_pool_ = false)
      : ring(ring_), max_batch(max_batch_), n(n_), paths(p), out_paths(o), status(st), mrSize(mrSize_), fmt(fmt_), device_format(device_format_),
        resume(resume_), nice_pool(nice_pool_), device_jpeg(device_jpeg_), pin(pin_), imgs((size_t)n_)
   {
      window = 2 * max_batch + dec_threads;
      try {
         for (int t = 0; t < dec_threads + wr_threads; t++) workers.emplace_back([this] { work_loop(); });
      } catch (...) {
         if (workers.empty()) { shutdown(); throw; }   // not one thread: give up; fewer than asked for: go on with those
      }
   }
   ~FileIO() override { shutdown(); }
   void shutdown()
   {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv_work.notify_all(); cv_img.notify_all(); cv_done.notify_all();
      for (auto &t : workers) if (t.joinable()) t.join();
      for (Img &im : imgs) if (im.data) { drop_buffer(im.data); im.data = nullptr; }
      for (auto &b : blob_pool) drop_buffer(b.first);
      blob_pool.clear();
   }
   // a reader's buffer that is not needed any more: back to the context when it is one of its pinned ones, else to the allocator
   void drop_buffer(uint8_t *p)
   {
      size_t bytes = 0;
      bool is_pinned = false;
      {
         std::lock_guard<std::mutex> lk(mu_pin);
         auto it = pinned.find(p);
         if (it != pinned.end()) { is_pinned = true; bytes = it->second; pinned.erase(it); }
      }
      if (is_pinned) pin.release(p, bytes, pin.user);
      else hesaff_free(p);
   }
   std::mutex mu_pin;   // guards `pinned` (taken alone or inside mu, never the other way round)
   bool decode_ready() const { return !stop && next_decode < n && next_decode < consumed + window; }   // under mu
   void work_loop()
   {
      (void)pthread_setname_np(pthread_self(), "hs-pool");   // (shows in /proc/<pid>/task/*/comm: bench.py's per-thread CPU table)
      // The pool's threads read and write() flat out; the caller's thread needs the CPU for microseconds at a time, to launch the next kernels
      // the moment an event fires.  On a share of two CPUs it must not queue behind them: the pool then runs at a lower priority (nice is
      // per thread on Linux; lowering it needs no privilege; a refusal changes nothing but the latency of the next launch).  On a host
      // with CPUs to spare the pool keeps the priority it was started with: it is what feeds the device.
      if (nice_pool) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 10);
      for (;;) {
         int i = -1;
         Task t{};
         {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return stop || decode_ready() || !tasks.empty(); });
            // Decode first while the device could run dry (less than one chunk decoded ahead), otherwise write first: finished rows hold
            // one of the three pinned result blocks until their files are written, and a small pool that fills the whole look-ahead
            // window (two chunks) before it writes lets the device wait for a free block (2 threads: 268 -> see profiles/r05_notes.md)
            const bool can_decode = decode_ready();
            const bool fed = next_decode - consumed >= max_batch || next_decode >= n;   // a full next chunk is decoded or being decoded
            if (!tasks.empty() && (!can_decode || fed)) { t = tasks.front(); tasks.pop_front(); }
            else if (can_decode) i = next_decode++;
            else if (!tasks.empty()) { t = tasks.front(); tasks.pop_front(); }
            else return;   // stop, nothing left to write
         }
         if (i >= 0) decode_one(i); else write_one(t);
      }
   }
   std::string out_name(int i, bool bin) const
   {
      const char *o = out_paths ? out_paths[i] : nullptr;
      if (!bin) return o ? std::string(o) : std::string(paths[i]) + ".hesaff.sift";   // hesaff.cpp:170-173
      return o ? (std::string(o) + ((fmt & HESAFF_OUT_TEXT) ? ".bin" : "")) : std::string(paths[i]) + ".hesaff.bin";
   }
   // Pixel buffers and coefficient blobs come from the context's page-locked memory when it offers some (PinHooks): the reader then
   // fills the buffer the copy engine reads - no malloc'ed image, no staging copy (8 MB per UHD PGM file, 25-50 MB per UHD JPEG blob:
   // 1-4 ms of a host thread per image).  Buffers of images that have reached the device go to the next decoder instead of back to
   // the allocator (fresh zero pages for every image otherwise: a third of the entropy decoder's time, a quarter of a PGM read).
   std::vector<std::pair<uint8_t *, size_t>> blob_pool;   // under mu
   static void *blob_alloc(size_t bytes, int *zeroed, void *user)
   {
      FileIO *io = (FileIO *)user;
      {
         std::lock_guard<std::mutex> lk(io->mu);
         for (size_t k = 0; k < io->blob_pool.size(); k++)
            if (io->blob_pool[k].second == bytes) {
               uint8_t *p = io->blob_pool[k].first;
               io->blob_pool[k] = io->blob_pool.back();
               io->blob_pool.pop_back();
               *zeroed = 0;
               return p;
            }
      }
      if (io->pin.alloc) {
         if (void *p = io->pin.alloc(bytes, io->pin.user)) {
            std::lock_guard<std::mutex> lk(io->mu_pin);
            io->pinned[p] = bytes;
            *zeroed = 0;
            return p;
         }
      }
      *zeroed = 1;
      return calloc(1, bytes);
   }
   bool is_pinned(const void *p)
   {
      std::lock_guard<std::mutex> lk(mu_pin);
      return pinned.count(const_cast<void *>(p)) != 0;
   }
   static bool is_jpeg_file(const char *path)   // SOI marker, like hesaff_read_image's choice of reader
   {
      FILE *f = fopen(path, "rb");
      if (!f) return false;
      unsigned char m[2] = {0, 0};
      const bool ok = fread(m, 1, 2, f) == 2 && m[0] == 0xFF && m[1] == 0xD8;
      fclose(f);
      return ok;
   }
   void decode_one(int i)
   {
      if (resume && paths[i]) {
         int n_text = 0, n_bin = 0;
         if (fmt & HESAFF_OUT_TEXT) n_text = hesaff_output_is_complete(out_name(i, false).c_str(), HESAFF_OUT_TEXT | (resume == 2 ? HESAFF_OUT_STRICT : 0));
         if (fmt & HESAFF_OUT_BIN) n_bin = hesaff_output_is_complete(out_name(i, true).c_str(), HESAFF_OUT_BIN);
         if (n_text >= 0 && n_bin >= 0 && (fmt != (HESAFF_OUT_TEXT | HESAFF_OUT_BIN) || n_text == n_bin)) {
            {
               std::lock_guard<std::mutex> lk(mu);
               imgs[(size_t)i].state = 2;
               status[i].rc = HESAFF_OK; status[i].stage = HESAFF_FILE_SKIPPED;
               status[i].count_hessian = -1; status[i].count_desc = (fmt & HESAFF_OUT_TEXT) ? n_text : n_bin;
            }
            cv_img.notify_all();
            return;
         }
      }
      Img im;
      int rc = HESAFF_ERR_ARG;
      if (paths[i] && device_jpeg && is_jpeg_file(paths[i])) {
         rc = hesaff_read_jpeg_coefficients_alloc(paths[i], &im.jpeg, &im.data, &im.blob_bytes, blob_alloc, this);
         if (rc == HESAFF_OK) { im.w = im.jpeg.width; im.h = im.jpeg.height; im.ch = im.jpeg.channels; }
      } else if (paths[i]) {
         rc = hesaff_read_image_alloc(paths[i], &im.data, &im.w, &im.h, &im.ch, blob_alloc, this);
      }
      int stage = HESAFF_FILE_UNREADABLE;
      if (rc == HESAFF_OK && (im.w > 65535 || im.h > 65535)) {   // the device plans 16-bit pixel coordinates (plan(), pipeline.hip)
         rc = HESAFF_ERR_ARG; stage = HESAFF_FILE_REJECTED;
      }
      if (rc != HESAFF_OK && im.data) { drop_buffer(im.data); im.data = nullptr; }   // (a failed reader hands an allocator's buffer back)
      {
         std::lock_guard<std::mutex> lk(mu);
         if (rc == HESAFF_OK) { im.state = 1; imgs[(size_t)i] = im; }
         else { imgs[(size_t)i].state = 2; status[i].rc = rc; status[i].stage = stage; }
      }
      cv_img.notify_all();
   }
   // the next run of consecutive readable images of one geometry.  Like ArrayIO's chunks, a long list starts and ends with smaller
   // chunks (1/4, 1/2 of max_batch): the device waits for the first chunk's images to be read and copied in, and the writers for the
   // last chunk's rows - nothing overlaps either
   int chunk_limit() const   // under mu
   {
      const int mb = max_batch, left = n - pos, tail = mb / 2 + mb / 4;
      if (n < 4 * mb || mb < 8) return mb;
      if (chunks_out == 0) return mb / 4;
      if (chunks_out == 1) return mb / 2;
      if (left <= mb / 4) return left;
      if (left <= tail) return left - mb / 4;
      if (left < tail + mb) return left - tail;
      return mb;
   }
   int largest_chunk(int this_chunk) const override { return (n >= 4 * max_batch && max_batch >= 8) ? std::max(this_chunk, max_batch) : this_chunk; }
   bool next(HostChunk &out) override
   {
      std::unique_lock<std::mutex> lk(mu);
      const int limit = chunk_limit();
      for (;;) {
         if (stop || pos >= n) break;
         cv_img.wait(lk, [&] { return stop || imgs[(size_t)pos].state != 0; });
         if (stop) break;
         Img &im = imgs[(size_t)pos];
         if (im.state == 2) { pos++; consumed = pos; cv_work.notify_all(); continue; }
         if (out.data.empty()) { out.W = im.w; out.H = im.h; out.ch = im.ch; out.blob_bytes = im.blob_bytes; out.jpeg = im.jpeg; out.pinned = true; }
         else if (im.w != out.W || im.h != out.H || im.ch != out.ch || im.blob_bytes != out.blob_bytes ||
                  (im.blob_bytes && !same_jpeg_layout(im.jpeg, out.jpeg))) break;
         out.data.push_back(im.data);
         out.pinned = out.pinned && is_pinned(im.data);
         out.stride.push_back((size_t)im.w * im.ch);
         out.index.push_back(pos);
         im.state = 3;
         pos++;
         if ((int)out.data.size() >= limit) break;
      }
      if (!out.data.empty()) chunks_out++;
      return !out.data.empty();
   }
   void staged(const HostChunk &q) override
   {
      {
         std::lock_guard<std::mutex> lk(mu);
         for (int i : q.index) {
            Img &im = imgs[(size_t)i];
            const size_t bytes = im.blob_bytes ? im.blob_bytes : (size_t)im.w * im.h * im.ch;   // (every reader allocates exactly this)
            // the pool holds at most `window` buffers; on a list of mixed sizes a buffer of a size that is no longer asked for makes
            // room for the one just used (otherwise the pool would fill with sizes that never match again and stop recycling)
            if ((int)blob_pool.size() >= window) {
               for (size_t k = 0; k < blob_pool.size(); k++)
                  if (blob_pool[k].second != bytes) {
                     drop_buffer(blob_pool[k].first);
                     blob_pool.erase(blob_pool.begin() + (long)k);
                     break;
                  }
            }
            if ((int)blob_pool.size() < window) blob_pool.emplace_back(im.data, bytes);
            else drop_buffer(im.data);
            im.data = nullptr;
         }
         consumed = std::max(consumed, q.index.back() + 1);
      }
      cv_work.notify_all();
   }
   void done(const ChunkDone &d) override
   {
      const size_t B = d.chunk->index.size();
      {
         std::lock_guard<std::mutex> lk(mu);
         const int id = (int)open_chunks.size();
         open_chunks.push_back({(int)B, d.block});
         for (size_t b = 0; b < B; b++) {
            const int i = d.chunk->index[b];
            status[i].count_hessian = d.count_hessian[b];
            status[i].count_desc = d.count_desc[b];
            status[i].stage = HESAFF_FILE_DETECTED;
            Task t{i, d.keys ? d.keys + d.key_off[b] : nullptr, d.count_desc[b], id, nullptr, 0, nullptr};
            if (d.text) { t.text = d.text + d.text_off[b]; t.text_len = (size_t)(d.text_off[b + 1] - d.text_off[b]); }
            if (d.bin) t.bin = d.bin + d.key_off[b] * (size_t)148;
            tasks.push_back(t);
            tasks_in_flight++;
         }
      }
      cv_work.notify_all();
   }
   int wants() const override
   {
      if (!device_format) return WANT_KEYS;
      return ((fmt & HESAFF_OUT_TEXT) ? WANT_TEXT : 0) | ((fmt & HESAFF_OUT_BIN) ? WANT_BIN : 0);
   }
   bool failed(const HostChunk &q, int rc) override
   {
      std::lock_guard<std::mutex> lk(mu);
      for (int i : q.index) { status[i].rc = rc; status[i].stage = HESAFF_FILE_REJECTED; }
      return true;
   }
   void write_one(const Task &t)
   {
      int rc = HESAFF_OK;
      if (fmt & HESAFF_OUT_TEXT) {
         const std::string name = out_name(t.index, false);
         if (device_format) rc = hesaff_write_sift_rows(name.c_str(), t.text, t.text_len, t.n);
         else rc = hesaff_write_sift_mt(name.c_str(), t.keys, t.n, mrSize, 1);
      }
      if ((fmt & HESAFF_OUT_BIN) && rc == HESAFF_OK) {
         const std::string name = out_name(t.index, true);
         if (device_format) rc = hesaff_write_bin_rows(name.c_str(), t.bin, t.n);
         else rc = hesaff_write_bin(name.c_str(), t.keys, t.n, mrSize);
      }
      int blk = -1;
      {
         std::lock_guard<std::mutex> lk(mu);
         status[t.index].rc = rc;
         if (rc == HESAFF_OK) status[t.index].stage = HESAFF_FILE_WRITTEN;
         if (--open_chunks[(size_t)t.chunk].left == 0) blk = open_chunks[(size_t)t.chunk].block;
         tasks_in_flight--;
      }
      if (blk >= 0) ring->release(blk);
      cv_done.notify_all();
   }
   void wait_writers()
   {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return tasks_in_flight == 0; });
   }
};


} // namespace hesaff_engine
