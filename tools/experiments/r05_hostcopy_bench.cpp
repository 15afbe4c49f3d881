// Host <-> device copy paths on the GPU box, to size soda_hip_run_host_box's
// staging (round 5): pageable hipMemcpy (what rounds 1-4 did), hipHostRegister
// on the caller's array, and a ring of pinned staging chunks filled by worker
// threads while the DMA engines drain it.  Prints one JSON object per line.
//   hipcc -O2 -o r05_hostcopy_bench r05_hostcopy_bench.cpp -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

static double now() {
  return std::chrono::duration<double>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

static void par_memcpy(char* dst, const char* src, size_t n, int threads) {
  if (threads <= 1) {
    memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> pool;
  size_t per = (n / threads + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; ++t) {
    size_t a = (size_t)t * per, b = a + per > n ? n : a + per;
    if (a >= n) break;
    pool.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
  }
  for (auto& th : pool) th.join();
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 256) << 20;
  char* host = (char*)aligned_alloc(4096, bytes);
  char* back = (char*)aligned_alloc(4096, bytes);
  for (size_t i = 0; i < bytes; i += 4096) host[i] = (char)i, back[i] = 1;
  void* dev;
  CK(hipMalloc(&dev, bytes));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const double gb = bytes / 1e9;
  double t;

  for (int rep = 0; rep < 2; ++rep) {
    t = now();
    CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    double h2d = now() - t;
    t = now();
    CK(hipMemcpy(back, dev, bytes, hipMemcpyDeviceToHost));
    double d2h = now() - t;
    printf("{\"what\": \"pageable hipMemcpy\", \"rep\": %d, \"h2d_ms\": %.2f, "
           "\"d2h_ms\": %.2f, \"h2d_GBs\": %.1f, \"d2h_GBs\": %.1f}\n",
           rep, h2d * 1e3, d2h * 1e3, gb / h2d, gb / d2h);
  }

  // registering the caller's pages
  for (int rep = 0; rep < 2; ++rep) {
    t = now();
    CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    double reg = now() - t;
    t = now();
    CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    double h2d = now() - t;
    t = now();
    CK(hipHostUnregister(host));
    double unreg = now() - t;
    printf("{\"what\": \"hipHostRegister\", \"rep\": %d, \"register_ms\": %.2f, "
           "\"h2d_ms\": %.2f, \"unregister_ms\": %.2f, \"h2d_GBs\": %.1f}\n",
           rep, reg * 1e3, h2d * 1e3, unreg * 1e3, gb / h2d);
  }
  // registering in bands while the previous band copies
  for (size_t band : {(size_t)16 << 20, (size_t)64 << 20}) {
    t = now();
    for (size_t off = 0; off < bytes; off += band) {
      size_t n = off + band > bytes ? bytes - off : band;
      CK(hipHostRegister(host + off, n, hipHostRegisterDefault));
      CK(hipMemcpyAsync((char*)dev + off, host + off, n, hipMemcpyHostToDevice,
                        s0));
    }
    CK(hipStreamSynchronize(s0));
    double all = now() - t;
    t = now();
    for (size_t off = 0; off < bytes; off += band)
      CK(hipHostUnregister(host + off));
    double unreg = now() - t;
    printf("{\"what\": \"register bands + async copy\", \"band_MiB\": %zu, "
           "\"total_ms\": %.2f, \"unregister_ms\": %.2f, \"GBs\": %.1f}\n",
           band >> 20, all * 1e3, unreg * 1e3, gb / all);
  }

  // pinned memory: what the DMA engines do alone
  char* pin;
  char* pin2;
  CK(hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&pin2, bytes, hipHostMallocDefault));
  memset(pin, 1, bytes);
  memset(pin2, 2, bytes);
  void* dev2;
  CK(hipMalloc(&dev2, bytes));
  for (int rep = 0; rep < 2; ++rep) {
    t = now();
    CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    double h2d = now() - t;
    t = now();
    CK(hipMemcpyAsync(pin2, dev2, bytes, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    double d2h = now() - t;
    t = now();
    CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s0));
    CK(hipMemcpyAsync(pin2, dev2, bytes, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s0));
    CK(hipStreamSynchronize(s1));
    double both = now() - t;
    printf("{\"what\": \"pinned DMA\", \"rep\": %d, \"h2d_GBs\": %.1f, "
           "\"d2h_GBs\": %.1f, \"both_ms\": %.2f, \"both_GBs_each\": %.1f}\n",
           rep, gb / h2d, gb / d2h, both * 1e3, gb / both);
  }

  // host memcpy pageable -> pinned with worker threads
  for (int threads : {1, 2, 4, 8, 12, 16}) {
    par_memcpy(pin, host, bytes, threads);
    t = now();
    par_memcpy(pin, host, bytes, threads);
    double in = now() - t;
    t = now();
    par_memcpy(back, pin2, bytes, threads);
    double out = now() - t;
    printf("{\"what\": \"host memcpy\", \"threads\": %d, \"to_pinned_GBs\": %.1f, "
           "\"from_pinned_GBs\": %.1f}\n", threads, gb / in, gb / out);
  }

  // the ring: worker threads fill chunk i+1 while chunk i is in flight
  for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)32 << 20})
    for (int threads : {2, 4, 8, 16}) {
      const int slots = 4;
      std::vector<hipEvent_t> ev(slots);
      for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      // H2D
      t = now();
      int i = 0;
      for (size_t off = 0; off < bytes; off += chunk, ++i) {
        size_t n = off + chunk > bytes ? bytes - off : chunk;
        int sl = i % slots;
        if (i >= slots) CK(hipEventSynchronize(ev[sl]));
        par_memcpy(pin + (size_t)sl * chunk, host + off, n, threads);
        CK(hipMemcpyAsync((char*)dev + off, pin + (size_t)sl * chunk, n,
                          hipMemcpyHostToDevice, s0));
        CK(hipEventRecord(ev[sl], s0));
      }
      CK(hipStreamSynchronize(s0));
      double h2d = now() - t;
      // D2H: copies run ahead, the host drains behind
      t = now();
      size_t nchunks = (bytes + chunk - 1) / chunk;
      for (size_t c = 0; c < nchunks + slots - 1; ++c) {
        if (c < nchunks) {
          size_t off = c * chunk, n = off + chunk > bytes ? bytes - off : chunk;
          int sl = c % slots;
          CK(hipMemcpyAsync(pin2 + (size_t)sl * chunk, (char*)dev + off, n,
                            hipMemcpyDeviceToHost, s1));
          CK(hipEventRecord(ev[sl], s1));
        }
        if (c + 1 >= (size_t)slots) {
          size_t d = c + 1 - slots;     // drain chunk d
          size_t off = d * chunk, n = off + chunk > bytes ? bytes - off : chunk;
          CK(hipEventSynchronize(ev[d % slots]));
          par_memcpy(back + off, pin2 + (d % slots) * chunk, n, threads);
        }
      }
      double d2h = now() - t;
      printf("{\"what\": \"pinned ring\", \"chunk_MiB\": %zu, \"threads\": %d, "
             "\"h2d_ms\": %.2f, \"d2h_ms\": %.2f, \"h2d_GBs\": %.1f, "
             "\"d2h_GBs\": %.1f}\n",
             chunk >> 20, threads, h2d * 1e3, d2h * 1e3, gb / h2d, gb / d2h);
      for (auto& e : ev) CK(hipEventDestroy(e));
    }
  return 0;
}
