// Does it matter which XCD touches which 4 KB pages?  Workgroups are dealt round-robin over the 8 XCDs (g mod 8 = XCD
// label).  With 4 KB per workgroup in address order (the fill pattern) XCD x only ever touches pages p = x mod 8; with a
// row of 8 KB per wave and 32 KB per workgroup every XCD touches every page residue.  Here: rows of 8 KB per wave, but
// the rows dealt so that an XCD pair (2 j, 2 j + 1) only gets rows r = j mod 4, i.e. pages {2 j, 2 j + 1} mod 8.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_affinity.hip -o audiocodec_amd/lib/ubench_affinity
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// MAP 0: row = 4 g + w.  MAP 1: XCD pair <-> row residue mod 4.  MAP 2: XCD <-> row residue mod 8 (16 KB page pairs... rows r = x mod 8)
// MODE 3: read only (row sums into Cc)
template <int MODE, int MAP>
__global__ __launch_bounds__(256) void k(float* __restrict__ A, float* __restrict__ B, float* __restrict__ Cc, long long rows, int shift) {
  const long long g = blockIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long r;
  if (MAP == 0) r = 4 * g + w;
  else if (MAP == 1) r = 4 * (8 * (g >> 3) + 2 * w + (g & 1)) + ((((int)(g & 7) >> 1) + shift) & 3);
  else r = 8 * (4 * (g >> 3) + w) + ((g + shift) & 7);
  if (r >= rows) return;
  if (MODE == 3) {
    const v4f* a = reinterpret_cast<const v4f*>(A + r * 2048) + lane;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[64 * i];
    if (acc.x == 12345.f) Cc[r] = acc.y;
    return;
  }
  v4f* a = reinterpret_cast<v4f*>(A + r * 2048) + lane;
  v4f v[8];
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v4f{1.f + i, 2.f, 3.f, (float)lane}, a + 64 * i);
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a[64 * i];
  v4f* b = reinterpret_cast<v4f*>(B + r * 2048) + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 1.5f, b + 64 * i);
  if (MODE == 2) {
    v4f* c = reinterpret_cast<v4f*>(Cc + r * 2048) + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i] * 2.5f, c + 64 * i);
  }
}
__global__ void k_xcc(int* out) {
  if (threadIdx.x == 0 && blockIdx.x < 64) {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.x] = v & 15;
  }
}
int main() {
  const long long rows = 120064;   // a multiple of 32
  float *A, *B, *Cc;
  CK(hipMalloc(&A, rows * 8192)); CK(hipMalloc(&B, rows * 8192)); CK(hipMalloc(&Cc, rows * 8192));
  printf("bases mod 32 KB: %llu %llu %llu\n", (unsigned long long)A % 32768, (unsigned long long)B % 32768, (unsigned long long)Cc % 32768);
  CK(hipMemset(A, 0, rows * 8192));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((rows + 3) / 4);
  for (int i = 0; i < 300; ++i) k<1, 0><<<grid, 256>>>(A, B, Cc, rows, 0);
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
  {   // which physical XCD does workgroup g run on?
    int* ids; CK(hipMalloc(&ids, 64 * sizeof(int)));
    k_xcc<<<64, 64>>>(ids);
    int h[64]; CK(hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\n");
  }
  const char* names[] = {"write A", "read A, write B", "read A, write B and C", "read A"};
  for (int mode = 0; mode < 4; ++mode) {
    const double bytes = (mode == 3 ? 1.0 : mode + 1.0) * rows * 8192;
    auto run = [&](int map, int shift) {
      return time([&] {
        if (mode == 0) { if (map == 0) k<0, 0><<<grid, 256>>>(A, B, Cc, rows, shift); else if (map == 1) k<0, 1><<<grid, 256>>>(A, B, Cc, rows, shift); else k<0, 2><<<grid, 256>>>(A, B, Cc, rows, shift); }
        if (mode == 1) { if (map == 0) k<1, 0><<<grid, 256>>>(A, B, Cc, rows, shift); else if (map == 1) k<1, 1><<<grid, 256>>>(A, B, Cc, rows, shift); else k<1, 2><<<grid, 256>>>(A, B, Cc, rows, shift); }
        if (mode == 2) { if (map == 0) k<2, 0><<<grid, 256>>>(A, B, Cc, rows, shift); else if (map == 1) k<2, 1><<<grid, 256>>>(A, B, Cc, rows, shift); else k<2, 2><<<grid, 256>>>(A, B, Cc, rows, shift); }
        if (mode == 3) { if (map == 0) k<3, 0><<<grid, 256>>>(A, B, Cc, rows, shift); else if (map == 1) k<3, 1><<<grid, 256>>>(A, B, Cc, rows, shift); else k<3, 2><<<grid, 256>>>(A, B, Cc, rows, shift); }
      });
    };
    printf("%-24s in order %5.0f GB/s | pair<->row mod 4, shift 0..3:", names[mode], bytes / run(0, 0) / 1e6);
    for (int sh = 0; sh < 4; ++sh) printf(" %5.0f", bytes / run(1, sh) / 1e6);
    printf(" | XCD<->row mod 8, shift 0..7:");
    for (int sh = 0; sh < 8; ++sh) printf(" %5.0f", bytes / run(2, sh) / 1e6);
    printf("\n");
  }
  return 0;
}
