// One write stream, a wave owns an 8 KB row (eight 1 KB stores): the stores issued back to back against spaced out in time
// (s_sleep between them), against one store per wave (4 KB per workgroup, the fill pattern).  Design aid.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_store_spacing.hip -o audiocodec_amd/lib/ubench_store_spacing
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int SLEEP, int WAIT>
__global__ __launch_bounds__(256) void k_row(float* __restrict__ A, long long rows) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= rows) return;
  const int lane = threadIdx.x & 63;
  v4f* a = reinterpret_cast<v4f*>(A + w * 2048) + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_nontemporal_store(v4f{1.f + i, 2.f, 3.f, (float)lane}, a + 64 * i);
    if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
}
// each wave: PER stores, the workgroup covers 4 PER KB contiguous, wave w pieces w, w + 4, ...
template <int PER>
__global__ __launch_bounds__(256) void k_block(float* __restrict__ A, long long n16) {
  const long long base = (long long)blockIdx.x * 256 * PER + threadIdx.x;
  if (base + 256 * (PER - 1) >= n16) return;
  v4f* p = reinterpret_cast<v4f*>(A) + base;
#pragma unroll
  for (int i = 0; i < PER; ++i) __builtin_nontemporal_store(v4f{1.f + i, 2.f, 3.f, (float)threadIdx.x}, p + 256 * i);
}
// two waves per 8 KB row: wave w of a workgroup writes half (w & 1) -- 4 KB, four 1 KB stores -- of row 2 blockIdx + (w >> 1);
// STREAMS = 2: the same into a second tensor B right after (the fused encode's two write streams, X then thr)
template <int STREAMS>
__global__ __launch_bounds__(256) void k_half_row(float* __restrict__ A, float* __restrict__ B, long long rows) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 2 + (w >> 1);
  if (row >= rows) return;
  const long long off = row * 2048;
  v4f* a = reinterpret_cast<v4f*>(A + off) + 256 * (w & 1) + lane;
#pragma unroll
  for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(v4f{1.f + i, 2.f, 3.f, (float)lane}, a + 64 * i);
  if (STREAMS == 2) {
    v4f* b = reinterpret_cast<v4f*>(B + off) + 256 * (w & 1) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(v4f{5.f + i, 2.f, 3.f, (float)lane}, b + 64 * i);
  }
}
template <int STREAMS>
__global__ __launch_bounds__(256) void k_row2(float* __restrict__ A, float* __restrict__ B, long long rows) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= rows) return;
  const int lane = threadIdx.x & 63;
  v4f* a = reinterpret_cast<v4f*>(A + w * 2048) + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v4f{1.f + i, 2.f, 3.f, (float)lane}, a + 64 * i);
  if (STREAMS == 2) {
    v4f* b = reinterpret_cast<v4f*>(B + w * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v4f{5.f + i, 2.f, 3.f, (float)lane}, b + 64 * i);
  }
}
// persistent: each wave walks rows w, w + W, w + 2 W ... (W = waves in the grid), 8 stores per row
template <int SLEEP>
__global__ __launch_bounds__(256) void k_persist(float* __restrict__ A, long long rows) {
  const long long W = (long long)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  for (long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); w < rows; w += W) {
    v4f* a = reinterpret_cast<v4f*>(A + w * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_nontemporal_store(v4f{1.f + i, 2.f, 3.f, (float)lane}, a + 64 * i);
      if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
}
int main() {
  const long long rows = 120064;
  float *A, *B;
  CK(hipMalloc(&A, rows * 8192));
  CK(hipMalloc(&B, rows * 8192));   // (a back-to-back allocation: normally the same class of VRAM as A, DESIGN_LOG.md)
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((rows + 3) / 4);
  auto time = [&](auto launch) {
    std::vector<float> ts;
    for (int it = 0; it < 9; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
  };
  for (int i = 0; i < 300; ++i) k_row<0, 0><<<grid, 256>>>(A, rows);
  const double bytes = (double)rows * 8192;
  const long long n16 = rows * 512;
  float t;
  t = time([&] { k_block<1><<<(unsigned)(n16 / 256), 256>>>(A, n16); });      printf("one store per wave (4 KB per workgroup)      %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_block<2><<<(unsigned)(n16 / 512), 256>>>(A, n16); });      printf("two stores per wave (8 KB per workgroup)     %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_block<4><<<(unsigned)(n16 / 1024), 256>>>(A, n16); });     printf("four stores per wave (16 KB per workgroup)   %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_block<8><<<(unsigned)(n16 / 2048), 256>>>(A, n16); });     printf("eight stores per wave (32 KB per workgroup)  %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_row<0, 0><<<grid, 256>>>(A, rows); });                     printf("row per wave, back to back                   %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_row<1, 0><<<grid, 256>>>(A, rows); });                     printf("row per wave, s_sleep 1 between stores       %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_row<4, 0><<<grid, 256>>>(A, rows); });                     printf("row per wave, s_sleep 4 between stores       %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_row<16, 0><<<grid, 256>>>(A, rows); });                    printf("row per wave, s_sleep 16 between stores      %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_row<0, 1><<<grid, 256>>>(A, rows); });                     printf("row per wave, vmcnt(0) after every store     %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  // does a row split over two waves (4 stores per wave instead of 8) lift the one-class ceiling?
  t = time([&] { k_half_row<1><<<(unsigned)((rows + 1) / 2), 256>>>(A, B, rows); });  printf("two waves per row (4 KB each), one stream       %.4f ms %5.0f GB/s\n", t, bytes / t / 1e6);
  t = time([&] { k_row2<2><<<grid, 256>>>(A, B, rows); });                            printf("row per wave, two streams (A then B)           %.4f ms %5.0f GB/s\n", t, 2 * bytes / t / 1e6);
  t = time([&] { k_half_row<2><<<(unsigned)((rows + 1) / 2), 256>>>(A, B, rows); });  printf("two waves per row (4 KB each), two streams     %.4f ms %5.0f GB/s\n", t, 2 * bytes / t / 1e6);
  for (int wg : {256 * 3, 256 * 8}) {
    t = time([&] { k_persist<0><<<wg, 256>>>(A, rows); });                    printf("persistent %d workgroups, back to back      %.4f ms %5.0f GB/s\n", wg, t, bytes / t / 1e6);
    t = time([&] { k_persist<4><<<wg, 256>>>(A, rows); });                    printf("persistent %d workgroups, s_sleep 4          %.4f ms %5.0f GB/s\n", wg, t, bytes / t / 1e6);
  }
  return 0;
}
