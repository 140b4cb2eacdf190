// Row-per-wave streams (8 KB rows, rows dealt to 4-wave workgroups in order) against how many waves a CU holds at a time:
// dynamic LDS caps the workgroups per CU, so the chip-wide window of rows in flight shrinks with it.  Design aid.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_window.hip -o audiocodec_amd/lib/ubench_window
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
extern __shared__ char dyn[];

// MODE 0: write row i of A.  MODE 1: read row i of A, write row i of B (copy).  MODE 2: read A, write B and C (encode's mix)
// ROT 1: wave w walks the eight 1 KB pieces of its row starting at piece w mod 8 (at any moment the chip then touches all
// eight piece positions evenly instead of the same one in every row)
template <int MODE, int ROT = 0>
__global__ __launch_bounds__(256) void k(float* __restrict__ A, float* __restrict__ B, float* __restrict__ Cc, long long rows) {
  if (threadIdx.x == 0 && rows < 0) dyn[0] = 1;   // keep the dynamic LDS allocation alive
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= rows) return;
  const int lane = threadIdx.x & 63;
  const int rot = ROT ? (int)(w & 7) : 0;
  v4f* a = reinterpret_cast<v4f*>(A + w * 2048) + lane;
  v4f v[8];
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v4f{1.f + i, 2.f, 3.f, (float)lane}, a + 64 * ((i + rot) & 7));
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a[64 * ((i + rot) & 7)];
  v4f* b = reinterpret_cast<v4f*>(B + w * 2048) + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 1.5f, b + 64 * ((i + rot) & 7));
  if (MODE == 2) {
    v4f* c = reinterpret_cast<v4f*>(Cc + w * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 2.5f, c + 64 * ((i + rot) & 7));
  }
}
int main() {
  const long long rows = 120064;
  float *A, *B, *Cc;
  CK(hipMalloc(&A, rows * 8192)); CK(hipMalloc(&B, rows * 8192)); CK(hipMalloc(&Cc, rows * 8192));
  CK(hipMemset(A, 0, rows * 8192));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((rows + 3) / 4);
  for (int m = 0; m < 3; ++m) {
    const void* f = m == 0 ? (const void*)k<0> : m == 1 ? (const void*)k<1> : (const void*)k<2>;
    CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  for (int i = 0; i < 200; ++i) k<1><<<grid, 256, 0>>>(A, B, Cc, rows);
  const char* names[] = {"write A", "read A, write B", "read A, write B and C"};
  const int wgs[] = {8, 6, 4, 3, 2, 1};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<float> tr;
    for (int it = 0; it < 9; ++it) {
      CK(hipEventRecord(e0));
      if (mode == 0) k<0, 1><<<grid, 256, 0>>>(A, B, Cc, rows);
      if (mode == 1) k<1, 1><<<grid, 256, 0>>>(A, B, Cc, rows);
      if (mode == 2) k<2, 1><<<grid, 256, 0>>>(A, B, Cc, rows);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 3) tr.push_back(ms);
    }
    std::sort(tr.begin(), tr.end());
    printf("%s, pieces rotated by wave: %.0f GB/s\n", names[mode], (mode + 1.0) * rows * 8192 / tr[tr.size() / 2] / 1e6);
    printf("%s (GB/s by workgroups of 4 waves per CU):", names[mode]);
    for (int wi = 0; wi < 6; ++wi) {
      const size_t lds = wgs[wi] >= 8 ? 0 : (size_t)(160 * 1024 / wgs[wi]) - 1024;
      std::vector<float> ts;
      for (int it = 0; it < 9; ++it) {
        CK(hipEventRecord(e0));
        if (mode == 0) k<0><<<grid, 256, lds>>>(A, B, Cc, rows);
        if (mode == 1) k<1><<<grid, 256, lds>>>(A, B, Cc, rows);
        if (mode == 2) k<2><<<grid, 256, lds>>>(A, B, Cc, rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      const double bytes = (mode + 1.0) * rows * 8192;
      printf("  %d: %.0f", wgs[wi], bytes / ts[ts.size() / 2] / 1e6);
    }
    printf("\n");
  }
  return 0;
}
