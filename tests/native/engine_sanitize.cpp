// Sanitizer driver for the host-only half of the chunk engine (test infrastructure; built by tests/test_host_sanitize.py with
// g++ -fsanitize=thread and -fsanitize=address,undefined from hesaff_amd/csrc/chunk_engine.h + hostio.cpp + jpeg_decode.cpp).
// The device side of run_chunks (capi_impl.h) is replaced by a mock that keeps its threading shape - a staging thread that calls
// ChunkIO::next / staged one chunk ahead, the caller's thread that "computes" a chunk, delivers the previous one and takes a
// result block from the ring of three - and fabricates records from the pixels it was handed (so every decoded byte is read
// while the engine says it is alive, and every record is read by a writer while its block is marked busy).
//   engine_sanitize <max_batch> <decode_threads> <write_threads> <format> <file>...      (format + 4: the mock hands over rows that
//                   are already text / packed - ChunkDone::text, text_off, bin - like the device formatter of kernels_export.h;
//                   format + 8: JPEG files arrive as coefficient blobs from the pool of recycled blobs, like hesaff_process_files)
//   engine_sanitize array <max_batch> <n_images>      ArrayIO (hesaff_detect_batch_cb's chunk source: a sink called per chunk with records
//                   that live in a ring block, the sink's return code read by the staging thread) under the same mock loop; the sink
//                   fails on the last third of a second run
// prints "chunks=<images per chunk, ...>" and "files=<n> written=<w> unreadable=<u> rows=<r>"; exit code 0 unless the pipeline misbehaved.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include "../../hesaff_amd/csrc/chunk_engine.h"

using namespace hesaff_engine;

struct State {
   HostChunk q;
   std::vector<int32_t> nh, nd;
   std::vector<size_t> off;
   std::vector<uint32_t> sum;
   std::vector<unsigned long long> toff;
   int total = 0, block = -1;
};

struct SinkState { long long rows = 0; int calls = 0; int fail_after = -1; uint32_t acc = 0; };
static int array_sink(void *user, int n_images, const int *image_index, const hesaff_result *results)
{
   SinkState *s = (SinkState *)user;
   for (int i = 0; i < n_images; i++) {
      for (int r = 0; r < results[i].count_desc; r++) s->acc = s->acc * 31u + results[i].keys[r].desc[(r + image_index[i]) & 127];   // read the block
      s->rows += results[i].count_desc;
   }
   s->calls++;
   return (s->fail_after >= 0 && s->calls > s->fail_after) ? 1 : 0;
}

// ArrayIO under the mock device loop: staging thread one chunk ahead (next), caller's thread computes, delivers the previous chunk (done ->
// sink -> ring release) and takes a block
static int run_array(int max_batch, int n, int fail_after, long long *rows_out)
{
   std::vector<std::vector<uint8_t>> pix((size_t)n);
   std::vector<const uint8_t *> ptr((size_t)n);
   std::vector<int> w((size_t)n), h((size_t)n), ch((size_t)n);
   for (int i = 0; i < n; i++) {
      w[(size_t)i] = (i % 3 == 1) ? 20 : 32; h[(size_t)i] = 16; ch[(size_t)i] = (i % 5 == 4) ? 3 : 1;
      pix[(size_t)i].assign((size_t)w[(size_t)i] * h[(size_t)i] * ch[(size_t)i], (uint8_t)(i * 7));
      ptr[(size_t)i] = pix[(size_t)i].data();
   }
   BlockRing ring;
   ring.reset(3);
   std::vector<std::vector<hesaff_keypoint>> blocks(3);
   ArrayIO io(&ring, max_batch, n, ptr.data(), w.data(), h.data(), nullptr, ch.data());
   SinkState st;
   st.fail_after = fail_after;
   io.sink = array_sink; io.user = &st;
   struct AState { HostChunk q; std::vector<int32_t> nh, nd; std::vector<size_t> off; int total = 0, block = -1; };
   auto stage = [&]() -> std::unique_ptr<AState> {
      std::unique_ptr<AState> s(new AState());
      if (!io.next(s->q)) return nullptr;
      io.staged(s->q);
      return s;
   };
   std::future<std::unique_ptr<AState>> staged = std::async(std::launch::async, stage);
   std::unique_ptr<AState> prev;
   auto deliver = [&](AState &s) {
      ChunkDone d;
      d.chunk = &s.q; d.count_hessian = s.nh.data(); d.count_desc = s.nd.data(); d.key_off = s.off.data();
      d.keys = blocks[(size_t)s.block].data(); d.block = s.block;
      io.done(d);
   };
   for (;;) {
      std::unique_ptr<AState> cur = staged.get();
      if (!cur) break;
      staged = std::async(std::launch::async, stage);
      for (size_t b = 0; b < cur->q.data.size(); b++) {
         const int cnt = (cur->q.index[b] * 37) % 200;
         cur->nh.push_back(cnt + 1); cur->nd.push_back(cnt); cur->off.push_back((size_t)cur->total);
         cur->total += cnt;
      }
      if (prev) { deliver(*prev); prev.reset(); }
      cur->block = ring.acquire();
      blocks[(size_t)cur->block].assign((size_t)cur->total + 1, hesaff_keypoint());
      for (auto &k : blocks[(size_t)cur->block]) for (int j = 0; j < 128; j++) k.desc[j] = (uint8_t)(j + cur->block);
      prev = std::move(cur);
   }
   if (prev) { deliver(*prev); prev.reset(); }
   *rows_out = st.rows;
   return io.sink_rc.load();
}

int main(int argc, char **argv)
{
   if (argc == 4 && !strcmp(argv[1], "array")) {
      const int max_batch = atoi(argv[2]), n = atoi(argv[3]);
      long long rows = 0, want = 0, rows2 = 0;
      for (int i = 0; i < n; i++) want += (i * 37) % 200;
      const int rc = run_array(max_batch, n, -1, &rows);
      const int rc2 = run_array(max_batch, n, 2, &rows2);     // the sink reports failure on its third call: the run stops early
      printf("array images=%d rows=%lld want=%lld rc=%d failing_run_rc=%d rows=%lld\n", n, rows, want, rc, rc2, rows2);
      return (rc == 0 && rows == want && rc2 != 0 && rows2 < want) ? 0 : 3;
   }
   if (argc < 6) return 2;
   const int max_batch = atoi(argv[1]), dt = atoi(argv[2]), wt = atoi(argv[3]), fmt = atoi(argv[4]) & 3;
   const bool device_format = (atoi(argv[4]) & 4) != 0, device_jpeg = (atoi(argv[4]) & 8) != 0;
   const bool mock_pin = (atoi(argv[4]) & 16) != 0;   // the context's page-locked read buffers (PinHooks), here plain malloc with a small budget
   const int n = argc - 5;
   std::vector<const char *> paths((size_t)n);
   for (int i = 0; i < n; i++) paths[(size_t)i] = argv[5 + i];
   std::vector<hesaff_file_status> status((size_t)n);
   for (auto &s : status) { s.rc = HESAFF_ERR_IO; s.stage = HESAFF_FILE_PENDING; s.count_hessian = s.count_desc = 0; }
   BlockRing ring;
   ring.reset(3);
   std::vector<std::vector<hesaff_keypoint>> blocks(3);
   std::vector<std::vector<char>> text_blocks(3), bin_blocks(3);
   long long rows = 0;
   std::string chunk_sizes;
   // mock of hesaff_ctx::PinReadCache: a budget of a few buffers, so that lists mix chunks of "pinned" and ordinary images; every buffer must
   // come back (release) before the run ends, and a chunk may only call itself pinned when all its images are
   struct MockPin {
      std::mutex mu;
      std::unordered_map<void *, size_t> out;
      size_t budget = 40000, given = 0, returned = 0;
   } mp;
   PinHooks pin;
   if (mock_pin) {
      pin.user = &mp;
      pin.alloc = [](size_t bytes, void *user) -> void * {
         MockPin *m = (MockPin *)user;
         std::lock_guard<std::mutex> lk(m->mu);
         size_t live = 0;
         for (auto &e : m->out) live += e.second;
         if (live + bytes > m->budget) return nullptr;
         void *q = malloc(bytes);
         if (q) { m->out[q] = bytes; m->given++; }
         return q;
      };
      pin.release = [](void *q, size_t bytes, void *user) {
         MockPin *m = (MockPin *)user;
         std::lock_guard<std::mutex> lk(m->mu);
         auto it = m->out.find(q);
         if (it == m->out.end() || it->second != bytes) { fprintf(stderr, "release of a buffer that is not out\n"); abort(); }
         m->out.erase(it); m->returned++;
         free(q);
      };
   }
   int pinned_chunks = 0;
   {
      FileIO io(&ring, max_batch, 5.196152f, fmt, n, paths.data(), nullptr, status.data(), dt, wt, device_format, false, device_jpeg, pin);
      auto stage = [&]() -> std::unique_ptr<State> {
         std::unique_ptr<State> s(new State());
         if (!io.next(s->q)) return nullptr;
         if (s->q.pinned) {
            if (!mock_pin) { fprintf(stderr, "pinned chunk without hooks\n"); abort(); }
            std::lock_guard<std::mutex> lk(mp.mu);
            for (const uint8_t *d : s->q.data) if (!mp.out.count((void *)d)) { fprintf(stderr, "pinned chunk holds an ordinary buffer\n"); abort(); }
            pinned_chunks++;
         }
         // "copy to pinned memory": read every pixel of every image of the chunk
         for (size_t b = 0; b < s->q.data.size(); b++) {
            uint32_t acc = 0;
            const size_t bytes = s->q.blob_bytes ? s->q.blob_bytes : (size_t)s->q.W * s->q.H * s->q.ch;   // a JPEG file's coefficient blob (recycled by the pool afterwards)
            for (size_t k = 0; k < bytes; k++) acc = acc * 31u + s->q.data[b][k];
            s->sum.push_back(acc);
         }
         io.staged(s->q);
         return s;
      };
      std::future<std::unique_ptr<State>> staged = std::async(std::launch::async, stage);
      std::unique_ptr<State> prev;
      auto deliver = [&](State &s) {
         ChunkDone d;
         d.chunk = &s.q; d.count_hessian = s.nh.data(); d.count_desc = s.nd.data(); d.key_off = s.off.data();
         d.keys = blocks[(size_t)s.block].data(); d.block = s.block;
         if (device_format) {
            d.keys = nullptr;
            if (io.wants() & WANT_TEXT) { d.text = text_blocks[(size_t)s.block].data(); d.text_off = s.toff.data(); }
            if (io.wants() & WANT_BIN) d.bin = bin_blocks[(size_t)s.block].data();
         }
         io.done(d);
      };
      for (;;) {
         std::unique_ptr<State> cur = staged.get();
         if (!cur) break;
         staged = std::async(std::launch::async, stage);
         const size_t B = cur->q.data.size();
         chunk_sizes += (chunk_sizes.empty() ? "" : ",") + std::to_string(B);
         for (size_t b = 0; b < B; b++) {
            const int cnt = (int)(cur->sum[b] % 700u);     // rows of this image, 0 included
            cur->nh.push_back(cnt + 3); cur->nd.push_back(cnt); cur->off.push_back((size_t)cur->total);
            cur->total += cnt;
         }
         if (prev) { deliver(*prev); prev.reset(); }
         cur->block = ring.acquire();
         std::vector<hesaff_keypoint> &blk = blocks[(size_t)cur->block];
         blk.assign((size_t)cur->total + 1, hesaff_keypoint());     // a writer still reading this block would be a race / use after free
         size_t o = 0;
         for (size_t b = 0; b < B; b++)
            for (int r = 0; r < cur->nd[b]; r++, o++) {
               hesaff_keypoint &k = blk[o];
               k.x = (float)(cur->sum[b] % 1000u) + (float)r; k.y = (float)r * 0.5f; k.s = 2.0f + (float)(r % 7);
               k.a11 = 1.25f; k.a12 = 0.0f; k.a21 = 0.1f; k.a22 = 0.8f; k.response = 30.0f; k.type = r & 1;
               for (int j = 0; j < 128; j++) k.desc[j] = (uint8_t)(cur->sum[b] + (uint32_t)(r * 131 + j));
            }
         if (device_format) {
            // the "device formatter": every image's rows through the host formatter, headers cut off, back to back
            std::vector<char> &tb = text_blocks[(size_t)cur->block], &bb = bin_blocks[(size_t)cur->block];
            tb.assign(1, 0); bb.assign((size_t)cur->total * 148 + 1, 0);
            cur->toff.assign(B + 1, 0ull);
            size_t to = 0;
            for (size_t b = 0; b < B; b++) {
               char *txt = nullptr; size_t len = 0;
               if (hesaff_format_sift(blk.data() + cur->off[b], cur->nd[b], 5.196152f, &txt, &len) != HESAFF_OK) return 4;
               size_t skip = 0;
               for (int nl = 0; nl < 2; skip++) if (txt[skip] == '\n') nl++;
               tb.resize(to + (len - skip) + 1);
               memcpy(tb.data() + to, txt + skip, len - skip);
               cur->toff[b] = to; to += len - skip;
               hesaff_free(txt);
               for (int r = 0; r < cur->nd[b]; r++) {
                  const hesaff_keypoint &k = blk[cur->off[b] + (size_t)r];
                  float v[5] = {k.x, k.y, 0, 0, 0};
                  hesaff_ellipse(&k, 5.196152f, &v[2], &v[3], &v[4]);
                  memcpy(bb.data() + (cur->off[b] + (size_t)r) * 148, v, 20);
                  memcpy(bb.data() + (cur->off[b] + (size_t)r) * 148 + 20, k.desc, 128);
               }
            }
            cur->toff[B] = to;
         }
         rows += cur->total;
         prev = std::move(cur);
      }
      if (prev) { deliver(*prev); prev.reset(); }
      io.wait_writers();
      io.shutdown();
   }
   int written = 0, unreadable = 0, other = 0;
   for (int i = 0; i < n; i++) {
      if (status[(size_t)i].stage == HESAFF_FILE_WRITTEN && status[(size_t)i].rc == HESAFF_OK) written++;
      else if (status[(size_t)i].stage == HESAFF_FILE_UNREADABLE) unreadable++;
      else other++;
   }
   if (mock_pin) {
      std::lock_guard<std::mutex> lk(mp.mu);
      printf("pinned_chunks=%d given=%zu returned=%zu out=%zu\n", pinned_chunks, mp.given, mp.returned, mp.out.size());
      if (!mp.out.empty() || mp.given != mp.returned || mp.given == 0) return 5;
   }
   printf("chunks=%s\n", chunk_sizes.c_str());
   printf("files=%d written=%d unreadable=%d other=%d rows=%lld\n", n, written, unreadable, other, rows);
   return other == 0 ? 0 : 3;
}
