// One 1 GB write stream, four ways: a wave owns an 8 KB row (the kernels' pattern) or the four waves of a workgroup
// interleave 1 KB pieces of a 32 KB chunk; non-temporal or plain stores.  Also a copy (1R:1W) in both patterns.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_write_pattern.hip -o audiocodec_amd/lib/ubench_write_pattern
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                \
  do {                                                       \
    hipError_t e_ = (x);                                     \
    if (e_ != hipSuccess) {                                  \
      printf("%s: %s\n", #x, hipGetErrorString(e_));         \
      exit(1);                                               \
    }                                                        \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool INTERLEAVE, bool NT, bool COPY>
__global__ __launch_bounds__(256) void k(const float* __restrict__ S, float* __restrict__ A, long long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long chunk = blockIdx.x;                       // 4 rows = 32 KB
  if (chunk * 4 + wave >= rows) return;
  // element (in 16-byte units) of piece i of this wave
  auto at = [&](int i) { return INTERLEAVE ? chunk * 2048 + (long long)(4 * i + wave) * 64 + lane : (chunk * 4 + wave) * 512 + 64 * i + lane; };
  v4f v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = COPY ? reinterpret_cast<const v4f*>(S)[at(i)] : v4f{1.f + i, 2.f, 3.f, (float)lane};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v4f* p = reinterpret_cast<v4f*>(A) + at(i);
    if (NT) __builtin_nontemporal_store(v[i], p);
    else *p = v[i];
  }
}

// the pattern of a torch fill: every thread owns PER consecutive 16-byte pieces (64 or 128 contiguous bytes)
template <int PER, bool NT>
__global__ __launch_bounds__(256) void k_thread(float* __restrict__ A, long long n16) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  v4f* p = reinterpret_cast<v4f*>(A) + t * PER;
  if ((t + 1) * PER > n16) return;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const v4f v = {1.f + i, 2.f, 3.f, (float)threadIdx.x};
    if (NT) __builtin_nontemporal_store(v, p + i);
    else p[i] = v;
  }
}
// block-strided: thread t of block b writes pieces b * 256 * PER + i * 256 + t (each instruction of a wave = 1 KB contiguous)
template <int PER, bool NT>
__global__ __launch_bounds__(256) void k_block(float* __restrict__ A, long long n16) {
  const long long base = (long long)blockIdx.x * 256 * PER + threadIdx.x;
  if (base + 256 * (PER - 1) >= n16) return;
  v4f* p = reinterpret_cast<v4f*>(A) + base;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const v4f v = {1.f + i, 2.f, 3.f, (float)threadIdx.x};
    if (NT) __builtin_nontemporal_store(v, p + 256 * i);
    else p[256 * i] = v;
  }
}

int main() {
  const size_t GB = 1ull << 30;
  char* arena;
  CK(hipMalloc(&arena, 100 * GB));
  const long long rows = 120064;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((rows + 3) / 4);
  float* S = reinterpret_cast<float*>(arena + 2 * GB);
  auto time = [&](auto launch) {
    std::vector<float> ts;
    for (int it = 0; it < 9; ++it) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
  };
  const double bytes = (double)rows * 8192;
  for (int far = 0; far < 2; ++far) {
    float* A = reinterpret_cast<float*>(arena + (far ? 80 : 6) * GB);
    printf("destination at %d GiB (source at 2 GiB)\n", far ? 80 : 6);
    float t;
    t = time([&] { k<false, true, false><<<grid, 256>>>(S, A, rows); });  printf("  write, row per wave, nt      %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k<false, false, false><<<grid, 256>>>(S, A, rows); }); printf("  write, row per wave, plain   %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k<true, true, false><<<grid, 256>>>(S, A, rows); });   printf("  write, interleaved, nt       %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k<true, false, false><<<grid, 256>>>(S, A, rows); });  printf("  write, interleaved, plain    %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    const long long n16 = rows * 512;
    t = time([&] { k_thread<4, false><<<(unsigned)(n16 / 4 / 256), 256>>>(A, n16); });  printf("  write, 64 B per thread, plain   %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k_thread<4, true><<<(unsigned)(n16 / 4 / 256), 256>>>(A, n16); });   printf("  write, 64 B per thread, nt      %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k_thread<8, false><<<(unsigned)(n16 / 8 / 256), 256>>>(A, n16); });  printf("  write, 128 B per thread, plain  %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k_block<4, false><<<(unsigned)(n16 / 4 / 256), 256>>>(A, n16); });   printf("  write, 4 x 4 KB per block, plain %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k_block<4, true><<<(unsigned)(n16 / 4 / 256), 256>>>(A, n16); });    printf("  write, 4 x 4 KB per block, nt    %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k_block<1, false><<<(unsigned)(n16 / 256), 256>>>(A, n16); });       printf("  write, 1 x 4 KB per block, plain %.4f ms  %.0f GB/s\n", t, bytes / t / 1e6);
    t = time([&] { k<false, true, true><<<grid, 256>>>(S, A, rows); });   printf("  copy, row per wave, nt       %.4f ms  %.0f GB/s\n", t, 2 * bytes / t / 1e6);
    t = time([&] { k<false, false, true><<<grid, 256>>>(S, A, rows); });  printf("  copy, row per wave, plain    %.4f ms  %.0f GB/s\n", t, 2 * bytes / t / 1e6);
    t = time([&] { k<true, true, true><<<grid, 256>>>(S, A, rows); });    printf("  copy, interleaved, nt        %.4f ms  %.0f GB/s\n", t, 2 * bytes / t / 1e6);
    t = time([&] { k<true, false, true><<<grid, 256>>>(S, A, rows); });   printf("  copy, interleaved, plain     %.4f ms  %.0f GB/s\n", t, 2 * bytes / t / 1e6);
  }
  return 0;
}
