// Sanitizer driver for the host-only half of the chunk engine (test infrastructure; built by tests/test_host_sanitize.py with
// g++ -fsanitize=thread and -fsanitize=address,undefined from hesaff_amd/csrc/chunk_engine.h + hostio.cpp + jpeg_decode.cpp).
// The device side of run_chunks (capi_impl.h) is replaced by a mock that keeps its threading shape - a staging thread that calls
// ChunkIO::next / staged one chunk ahead, the caller's thread that "computes" a chunk, delivers the previous one and takes a
// result block from the ring of three - and fabricates records from the pixels it was handed (so every decoded byte is read
// while the engine says it is alive, and every record is read by a writer while its block is marked busy).
//   engine_sanitize <max_batch> <decode_threads> <write_threads> <format> <file>...
// prints "files=<n> written=<w> unreadable=<u> rows=<r>"; exit code 0 unless the pipeline misbehaved.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include "../../hesaff_amd/csrc/chunk_engine.h"

using namespace hesaff_engine;

struct State {
   HostChunk q;
   std::vector<int32_t> nh, nd;
   std::vector<size_t> off;
   std::vector<uint32_t> sum;
   int total = 0, block = -1;
};

int main(int argc, char **argv)
{
   if (argc < 6) return 2;
   const int max_batch = atoi(argv[1]), dt = atoi(argv[2]), wt = atoi(argv[3]), fmt = atoi(argv[4]);
   const int n = argc - 5;
   std::vector<const char *> paths((size_t)n);
   for (int i = 0; i < n; i++) paths[(size_t)i] = argv[5 + i];
   std::vector<hesaff_file_status> status((size_t)n);
   for (auto &s : status) { s.rc = HESAFF_ERR_IO; s.stage = HESAFF_FILE_PENDING; s.count_hessian = s.count_desc = 0; }
   BlockRing ring;
   ring.reset(3);
   std::vector<std::vector<hesaff_keypoint>> blocks(3);
   long long rows = 0;
   {
      FileIO io(&ring, max_batch, 5.196152f, fmt, n, paths.data(), nullptr, status.data(), dt, wt);
      auto stage = [&]() -> std::unique_ptr<State> {
         std::unique_ptr<State> s(new State());
         if (!io.next(s->q)) return nullptr;
         // "copy to pinned memory": read every pixel of every image of the chunk
         for (size_t b = 0; b < s->q.data.size(); b++) {
            uint32_t acc = 0;
            const size_t bytes = (size_t)s->q.W * s->q.H * s->q.ch;
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
         io.done(d);
      };
      for (;;) {
         std::unique_ptr<State> cur = staged.get();
         if (!cur) break;
         staged = std::async(std::launch::async, stage);
         const size_t B = cur->q.data.size();
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
   printf("files=%d written=%d unreadable=%d other=%d rows=%lld\n", n, written, unreadable, other, rows);
   return other == 0 ? 0 : 3;
}
