// Micro-benchmark behind DESIGN.md section 6: how fast do T threads write() fresh 46 MB files to /dev/shm from (a) ordinary memory,
// (b) memory pinned by hipHostMalloc (where the copy engine delivers the formatted rows)?
//   hipcc -O2 -o /tmp/tmpfs_write scripts/micro/tmpfs_write.cpp && /tmp/tmpfs_write
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double run(const char *src, size_t bytes, int threads, int files_per_thread, int round)
{
   auto t0 = std::chrono::steady_clock::now();
   std::vector<std::thread> th;
   for (int t = 0; t < threads; t++)
      th.emplace_back([=] {
         for (int f = 0; f < files_per_thread; f++) {
            const std::string p = "/dev/shm/tw_" + std::to_string(round) + "_" + std::to_string(t) + "_" + std::to_string(f);
            const int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
            size_t o = 0;
            while (o < bytes) { const ssize_t w = write(fd, src + o, bytes - o); if (w <= 0) break; o += (size_t)w; }
            close(fd);
         }
      });
   for (auto &x : th) x.join();
   const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   for (int t = 0; t < threads; t++)
      for (int f = 0; f < files_per_thread; f++) unlink(("/dev/shm/tw_" + std::to_string(round) + "_" + std::to_string(t) + "_" + std::to_string(f)).c_str());
   return (double)bytes * threads * files_per_thread / dt / 1e9;
}

int main()
{
   const size_t bytes = (size_t)46 << 20;
   char *plain = (char *)malloc(bytes), *pinned = nullptr;
   memset(plain, 'a', bytes);
   if (hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault) != hipSuccess) { puts("hipHostMalloc failed"); return 1; }
   memset(pinned, 'b', bytes);
   int round = 0;
   for (int threads : {1, 2, 4, 8, 16}) {
      const int fpt = 48 / threads > 0 ? 48 / threads : 1;
      const double a = run(plain, bytes, threads, fpt, round++), b = run(pinned, bytes, threads, fpt, round++);
      printf("threads %2d: ordinary memory %6.2f GB/s   pinned memory %6.2f GB/s\n", threads, a, b);
   }
   return 0;
}
