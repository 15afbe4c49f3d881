// Second host-copy probe (round 5): the questions the first one left open for
// soda_hip_run_host_box --
//   * hipHostRegister on FRESH memory each time (the first probe re-registered
//     one array and read 0 ms the second time): touched vs untouched pages;
//   * two threads registering two arrays at once;
//   * page-aligned pieces registered while the previous piece's DMA runs;
//   * hipMemcpy2DAsync device -> registered host, partial rows (the valid box
//     of an iterated stencil: columns [100, 8092) of 8192).
//   hipcc -O2 -o bin/r05_hostcopy_bench2 r05_hostcopy_bench2.cpp -lpthread
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                \
    }                                                         \
  } while (0)

static double now() {
  return std::chrono::duration<double>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

static char* fresh(size_t bytes, bool touch) {
  char* p = (char*)mmap(nullptr, bytes + 4096, PROT_READ | PROT_WRITE,
                        MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) exit(2);
  p += 16;   // like a malloc'ed block: not page aligned
  if (touch)
    for (size_t i = 0; i < bytes; i += 4096) p[i] = (char)i;
  return p;
}

int main() {
  const size_t bytes = (size_t)256 << 20;
  const double gb = bytes / 1e9;
  void* dev;
  void* dev2;
  CK(hipMalloc(&dev, bytes));
  CK(hipMalloc(&dev2, bytes));
  CK(hipMemset(dev2, 1, bytes));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  double t;

  for (int touch = 1; touch >= 0; --touch)
    for (int rep = 0; rep < 2; ++rep) {
      char* a = fresh(bytes, touch);
      t = now();
      CK(hipHostRegister(a, bytes, hipHostRegisterDefault));
      double reg = now() - t;
      t = now();
      if (touch) {
        CK(hipMemcpyAsync(dev, a, bytes, hipMemcpyHostToDevice, s0));
      } else {
        CK(hipMemcpyAsync(a, dev2, bytes, hipMemcpyDeviceToHost, s0));
      }
      CK(hipStreamSynchronize(s0));
      double cp = now() - t;
      t = now();
      CK(hipHostUnregister(a));
      double un = now() - t;
      printf("{\"what\": \"register fresh %s array\", \"rep\": %d, "
             "\"register_ms\": %.2f, \"copy_ms\": %.2f, \"unregister_ms\": %.2f}\n",
             touch ? "touched (H2D)" : "untouched (D2H)", rep, reg * 1e3,
             cp * 1e3, un * 1e3);
    }

  {  // pageable copies of fresh arrays, for comparison
    char* a = fresh(bytes, true);
    t = now();
    CK(hipMemcpy(dev, a, bytes, hipMemcpyHostToDevice));
    double h2d = now() - t;
    char* b = fresh(bytes, false);
    t = now();
    CK(hipMemcpy(b, dev2, bytes, hipMemcpyDeviceToHost));
    double d2h = now() - t;
    char* c = fresh(bytes, true);
    t = now();
    CK(hipMemcpy(c, dev2, bytes, hipMemcpyDeviceToHost));
    double d2ht = now() - t;
    printf("{\"what\": \"pageable hipMemcpy, fresh arrays\", \"h2d_ms\": %.2f, "
           "\"d2h_untouched_ms\": %.2f, \"d2h_touched_ms\": %.2f}\n", h2d * 1e3,
           d2h * 1e3, d2ht * 1e3);
  }

  {  // two threads registering at once
    char* a = fresh(bytes, true);
    char* b = fresh(bytes, true);
    t = now();
    std::thread th([&] { CK(hipHostRegister(b, bytes, hipHostRegisterDefault)); });
    CK(hipHostRegister(a, bytes, hipHostRegisterDefault));
    double mine = now() - t;
    th.join();
    double both = now() - t;
    printf("{\"what\": \"two threads register 256 MiB each\", \"first_ms\": %.2f, "
           "\"both_ms\": %.2f}\n", mine * 1e3, both * 1e3);
    CK(hipHostUnregister(a));
    CK(hipHostUnregister(b));
  }

  for (size_t piece : {(size_t)8 << 20, (size_t)32 << 20}) {
    // page-aligned pieces of a fresh array: register piece k, copy piece k
    char* a = fresh(bytes, true);
    char* end = a + bytes;
    t = now();
    char* cur = a;
    std::vector<std::pair<char*, size_t>> regs;
    while (cur < end) {
      char* lo = (char*)((uintptr_t)cur & ~(uintptr_t)4095);
      char* hi = (char*)(((uintptr_t)lo + piece));
      if (hi > end) hi = (char*)(((uintptr_t)end + 4095) & ~(uintptr_t)4095);
      CK(hipHostRegister(lo, hi - lo, hipHostRegisterDefault));
      regs.push_back({lo, (size_t)(hi - lo)});
      char* stop = hi < end ? hi : end;
      CK(hipMemcpyAsync((char*)dev + (cur - a), cur, stop - cur,
                        hipMemcpyHostToDevice, s0));
      cur = stop;
    }
    double enq = now() - t;
    CK(hipStreamSynchronize(s0));
    double all = now() - t;
    t = now();
    for (auto& r : regs) CK(hipHostUnregister(r.first));
    double un = now() - t;
    printf("{\"what\": \"fresh array, page-aligned pieces: register + async "
           "H2D\", \"piece_MiB\": %zu, \"enqueue_ms\": %.2f, \"total_ms\": %.2f, "
           "\"unregister_ms\": %.2f, \"GBs\": %.1f}\n", piece >> 20, enq * 1e3,
           all * 1e3, un * 1e3, gb / all);
  }

  {  // a helper thread registers the output while the input streams in
    char* a = fresh(bytes, true);
    char* b = fresh(bytes, true);
    const size_t piece = (size_t)16 << 20;
    t = now();
    std::thread th([&] { CK(hipHostRegister(b, bytes, hipHostRegisterDefault)); });
    char* end = a + bytes;
    char* cur = a;
    std::vector<char*> regs;
    while (cur < end) {
      char* lo = (char*)((uintptr_t)cur & ~(uintptr_t)4095);
      char* hi = lo + piece;
      if (hi > end) hi = (char*)(((uintptr_t)end + 4095) & ~(uintptr_t)4095);
      CK(hipHostRegister(lo, hi - lo, hipHostRegisterDefault));
      regs.push_back(lo);
      char* stop = hi < end ? hi : end;
      CK(hipMemcpyAsync((char*)dev + (cur - a), cur, stop - cur,
                        hipMemcpyHostToDevice, s0));
      cur = stop;
    }
    CK(hipStreamSynchronize(s0));
    double in = now() - t;
    th.join();
    double joined = now() - t;
    // the valid box of jacobi2d x 100 on 8192^2: rows and columns [100, 8092)
    const size_t pitch = 8192 * 4, width = 7992 * 4, height = 7992;
    double t2 = now();
    CK(hipMemcpy2DAsync(b + 100 * pitch + 400, pitch,
                        (char*)dev2 + 100 * pitch + 400, pitch, width, height,
                        hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    double out2d = now() - t2;
    t2 = now();
    CK(hipMemcpyAsync(b + 100 * pitch, (char*)dev2 + 100 * pitch,
                      pitch * height, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    double out1d = now() - t2;
    // the same 2-D copy in 8 bands
    t2 = now();
    for (int k = 0; k < 8; ++k) {
      size_t r0 = 100 + (size_t)k * 999, r1 = k == 7 ? 8092 : r0 + 999;
      CK(hipMemcpy2DAsync(b + r0 * pitch + 400, pitch,
                          (char*)dev2 + r0 * pitch + 400, pitch, width, r1 - r0,
                          hipMemcpyDeviceToHost, s1));
    }
    CK(hipStreamSynchronize(s1));
    double out2db = now() - t2;
    printf("{\"what\": \"input in 16 MiB pieces beside a thread registering the "
           "output\", \"input_done_ms\": %.2f, \"output_registered_ms\": %.2f, "
           "\"d2h_2d_ms\": %.2f, \"d2h_1d_same_rows_ms\": %.2f, "
           "\"d2h_2d_8_bands_ms\": %.2f}\n", in * 1e3, joined * 1e3,
           out2d * 1e3, out1d * 1e3, out2db * 1e3);
    // 2-D copy into UNREGISTERED memory
    char* c = fresh(bytes, true);
    t2 = now();
    CK(hipMemcpy2DAsync(c + 100 * pitch + 400, pitch,
                        (char*)dev2 + 100 * pitch + 400, pitch, width, height,
                        hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    printf("{\"what\": \"2-D D2H into pageable memory\", \"ms\": %.2f}\n",
           (now() - t2) * 1e3);
    for (auto r : regs) CK(hipHostUnregister(r));
    CK(hipHostUnregister(b));
  }
  return 0;
}
