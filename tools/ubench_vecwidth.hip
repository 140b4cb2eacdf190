// Row-per-wave copy (read row i of A, write row i of B; 8 KB rows in order) with 16-byte against 8-byte accesses per lane,
// and with one row per wave against two half-size rows 4 KB apart in two different tensors (the mono kernels' shape:
// two clips per wave).  Design aid.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_vecwidth.hip -o audiocodec_amd/lib/ubench_vecwidth
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: 8 x 16 B per lane, one 8 KB row.  MODE 1: 16 x 8 B per lane, one 8 KB row.
// MODE 2: two 4 KB rows (rows i of two halves of the tensors, "two clips"), 8 x 8 B per lane each (the mono kernels).
// MODE 3: two 4 KB rows, 4 x 16 B per lane each on HALF the lanes... (lanes 0-31 row 0, lanes 32-63 row 1): 16-byte mono
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, float* __restrict__ B, long long rows) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= rows) return;
  const int lane = threadIdx.x & 63;
  if (MODE == 0) {
    const v4f* a = reinterpret_cast<const v4f*>(A + w * 2048) + lane;
    v4f* b = reinterpret_cast<v4f*>(B + w * 2048) + lane;
    v4f v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a[64 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 1.5f, b + 64 * i);
  } else if (MODE == 1) {
    const v2f* a = reinterpret_cast<const v2f*>(A + w * 2048) + lane;
    v2f* b = reinterpret_cast<v2f*>(B + w * 2048) + lane;
    v2f v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a[64 * i];
#pragma unroll
    for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(v[i] * 1.5f, b + 64 * i);
  } else if (MODE == 2) {
    const long long half = rows * 1024;   // floats per half tensor
    const v2f* a0 = reinterpret_cast<const v2f*>(A + w * 1024) + lane;
    const v2f* a1 = reinterpret_cast<const v2f*>(A + half + w * 1024) + lane;
    v2f* b0 = reinterpret_cast<v2f*>(B + w * 1024) + lane;
    v2f* b1 = reinterpret_cast<v2f*>(B + half + w * 1024) + lane;
    v2f u[8], v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = a0[64 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a1[64 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(u[i] * 1.5f, b0 + 64 * i);
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 1.5f, b1 + 64 * i);
  } else {
    const long long half = rows * 1024;
    const int h = lane >> 5, l = lane & 31;
    const v4f* a = reinterpret_cast<const v4f*>(A + h * half + w * 1024) + l;
    v4f* b = reinterpret_cast<v4f*>(B + h * half + w * 1024) + l;
    v4f v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a[32 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 1.5f, b + 32 * i);
  }
}
int main() {
  const long long rows = 120064;
  float *A, *B;
  CK(hipMalloc(&A, rows * 8192)); CK(hipMalloc(&B, rows * 8192));
  CK(hipMemset(A, 0, rows * 8192));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((rows + 3) / 4);
  for (int i = 0; i < 200; ++i) k<0><<<grid, 256>>>(A, B, rows);
  const char* names[] = {"one 8 KB row, 16 B per lane", "one 8 KB row, 8 B per lane", "two 4 KB rows, 8 B per lane (mono kernels)", "two 4 KB rows, 16 B per lane on half-waves"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      std::vector<float> ts;
      for (int it = 0; it < 9; ++it) {
        CK(hipEventRecord(e0));
        if (mode == 0) k<0><<<grid, 256>>>(A, B, rows);
        if (mode == 1) k<1><<<grid, 256>>>(A, B, rows);
        if (mode == 2) k<2><<<grid, 256>>>(A, B, rows);
        if (mode == 3) k<3><<<grid, 256>>>(A, B, rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      printf("%-48s %.4f ms  %.0f GB/s\n", names[mode], ts[ts.size() / 2], 2.0 * rows * 8192 / ts[ts.size() / 2] / 1e6);
    }
  return 0;
}
