// Two write streams of 1 GB each, written side by side by one kernel (row i of A, then row i of B, rows dealt to
// workgroups in order), against where B sits relative to A inside one 160 GiB allocation: the pure form of the placement
// effect of DESIGN_LOG.md section 9a (design aid, not product).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_two_streams.hip -o audiocodec_amd/lib/ubench_two_streams
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x)                                                \
  do {                                                       \
    hipError_t e_ = (x);                                     \
    if (e_ != hipSuccess) {                                  \
      printf("%s: %s\n", #x, hipGetErrorString(e_));         \
      exit(1);                                               \
    }                                                        \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// one wave per row of 8 KB: 8 x 16-byte stores per lane and stream; MODE 0: write A and B; 1: write A only (2 GB worth of
// rows); 2: read A, write B
template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ A, float* __restrict__ B, long long rows) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  v4f* a = reinterpret_cast<v4f*>(A + row * 2048) + lane;
  v4f* b = reinterpret_cast<v4f*>(B + row * 2048) + lane;
  v4f v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = MODE == 2 ? a[64 * i] : v4f{1.f + i, 2.f, 3.f, (float)lane};
  if (MODE != 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i], a + 64 * i);
  }
  if (MODE != 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 1.5f, b + 64 * i);
  }
}

int main() {
  const size_t GB = 1ull << 30;
  const int AG = 160;
  char* arena;
  CK(hipMalloc(&arena, AG * GB));
  const long long rows = 120064;   // 983 MB per stream
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](int mode, size_t offA, size_t offB) {
    float* A = reinterpret_cast<float*>(arena + offA);
    float* B = reinterpret_cast<float*>(arena + offB);
    std::vector<float> ts;
    for (int it = 0; it < 7; ++it) {
      CK(hipEventRecord(e0));
      const unsigned grid = (unsigned)((rows + 3) / 4);
      if (mode == 0) k<0><<<grid, 256>>>(A, B, rows);
      if (mode == 1) k<1><<<grid, 256>>>(A, B, rows);
      if (mode == 2) k<2><<<grid, 256>>>(A, B, rows);
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
  printf("one write stream (A at 2 GiB): %.4f ms = %.0f GB/s\n", run(1, 2 * GB, 4 * GB), bytes / run(1, 2 * GB, 4 * GB) / 1e6);
  for (int mode : {0, 2}) {
    printf("%s, A at 2 GiB, B at 4, 8, ... GiB (ms; %.2f GB moved):\n", mode == 0 ? "two write streams" : "read A, write B", 2 * bytes / 1e9);
    for (int o = 4; o + 1 < AG; o += 4) {
      printf(" %.3f", run(mode, 2 * GB, (size_t)o * GB));
      if ((o / 4) % 13 == 0) printf("\n");
    }
    printf("\n");
  }
  return 0;
}
