// What about two write streams in one "class" of VRAM stretches (DESIGN_LOG.md section 9a) is slow?  Maps the classes of a
// 160 GiB arena against a stream A at 2 GiB (two-stream write time per 4 GiB step), picks one same-class and one other-class
// place for B, and times variants of how the two streams are interleaved (design aid, not product).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_classes.hip -o audiocodec_amd/lib/ubench_classes
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                        \
  do {                                               \
    hipError_t e_ = (x);                             \
    if (e_ != hipSuccess) {                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                       \
    }                                                \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void put_row(float* base, long long row, int lane, float s) {
  v4f* p = reinterpret_cast<v4f*>(base + row * 2048) + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v4f{1.f + i, s, 3.f, (float)lane}, p + 64 * i);
}
__device__ __forceinline__ float get_row(const float* base, long long row, int lane) {
  const v4f* p = reinterpret_cast<const v4f*>(base + row * 2048) + lane;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += p[64 * i];
  return acc.x + acc.y + acc.z + acc.w;
}

// MODE 0: wave writes row i of A, then row i of B (the encode kernel's pattern)
// MODE 1: even workgroups write rows of A, odd workgroups rows of B (both streams active, never from one wave)
// MODE 2: the first half of the grid writes all of A, the second half all of B (one stream after the other)
// MODE 3: wave writes 4 consecutive rows of A, then the same 4 rows of B
// MODE 4: A only, 2 x rows (single stream of the same size)
// MODE 5: read row i of R, write row i of A and of B (the encode kernel's three streams)
// MODE 6: wave writes row i of A and row (i + rows / 2) mod rows of B (streams half a tensor apart)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ A, float* __restrict__ B, const float* __restrict__ R,
                                         long long rows) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (MODE == 0 || MODE == 5 || MODE == 6) {
    if (w >= rows) return;
    float s = 2.f;
    if (MODE == 5) s = get_row(R, w, lane);
    put_row(A, w, lane, s);
    put_row(B, MODE == 6 ? (w + rows / 2) % rows : w, lane, s * 1.5f);
  } else if (MODE == 1) {
    const long long g = blockIdx.x >> 1, row = g * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    put_row((blockIdx.x & 1) ? B : A, row, lane, 2.f);
  } else if (MODE == 2) {
    if (w >= 2 * rows) return;
    put_row(w < rows ? A : B, w < rows ? w : w - rows, lane, 2.f);
  } else if (MODE == 3) {
    if (4 * w >= rows) return;
    for (int j = 0; j < 4; ++j) put_row(A, 4 * w + j, lane, 2.f);
    for (int j = 0; j < 4; ++j) put_row(B, 4 * w + j, lane, 3.f);
  } else if (MODE == 4) {
    if (w >= 2 * rows) return;
    put_row(A, w, lane, 2.f);
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
  auto run = [&](int mode, size_t offA, size_t offB, size_t offR) {
    float* A = reinterpret_cast<float*>(arena + offA);
    float* B = reinterpret_cast<float*>(arena + offB);
    const float* R = reinterpret_cast<const float*>(arena + offR);
    std::vector<float> ts;
    for (int it = 0; it < 9; ++it) {
      CK(hipEventRecord(e0));
      const unsigned g1 = (unsigned)((rows + 3) / 4);
      switch (mode) {
        case 0: k<0><<<g1, 256>>>(A, B, R, rows); break;
        case 1: k<1><<<2 * g1, 256>>>(A, B, R, rows); break;
        case 2: k<2><<<2 * g1, 256>>>(A, B, R, rows); break;
        case 3: k<3><<<(g1 + 3) / 4, 256>>>(A, B, R, rows); break;
        case 4: k<4><<<2 * g1, 256>>>(A, B, R, rows); break;
        case 5: k<5><<<g1, 256>>>(A, B, R, rows); break;
        case 6: k<6><<<g1, 256>>>(A, B, R, rows); break;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
  };
  // settle the device, then the map
  for (int i = 0; i < 200; ++i) k<0><<<(unsigned)((rows + 3) / 4), 256>>>((float*)(arena + 2 * GB), (float*)(arena + 6 * GB), nullptr, rows);
  CK(hipDeviceSynchronize());
  std::vector<std::pair<float, int>> map;
  printf("two write streams, A at 2 GiB, B at 6, 10, ... GiB (ms):\n");
  for (int o = 6; o + 2 < AG; o += 4) {
    const float t = run(0, 2 * GB, (size_t)o * GB, 0);
    map.push_back({t, o});
    printf(" %.3f", t);
  }
  printf("\n");
  float lo = 1e9f, hi = 0.f;
  for (auto& m : map) lo = std::min(lo, m.first), hi = std::max(hi, m.first);
  int same = -1, other = -1, other2 = -1;
  for (auto& m : map) {
    if (same < 0 && m.first > hi - 0.15f * (hi - lo)) same = m.second;
    if (m.first < lo + 0.15f * (hi - lo)) {
      if (other < 0) other = m.second;
      else other2 = m.second;
    }
  }
  printf("range %.3f .. %.3f ms; same-class place %d GiB, other-class places %d and %d GiB\n", lo, hi, same, other, other2);
  if (same < 0 || other < 0 || hi < 1.08f * lo) {
    printf("no two classes in this arena\n");
    return 0;
  }
  const double bytes = (double)rows * 8192;
  const char* names[] = {"0 wave: row of A then row of B", "1 even WGs write A, odd WGs write B", "2 all of A, then all of B",
                         "3 wave: 4 rows of A then 4 rows of B", "4 one stream of twice the size (A only)",
                         "5 read R, write A and B (R beside A)", "6 A row i with B row i + rows/2"};
  for (int mode = 0; mode < 7; ++mode) {
    const float ts = run(mode, 2 * GB, (size_t)same * GB, 4 * GB), to = run(mode, 2 * GB, (size_t)other * GB, 4 * GB);
    const double moved = (mode == 5 ? 3 : 2) * bytes;
    printf("%-44s same class %.4f ms (%5.0f GB/s)   other class %.4f ms (%5.0f GB/s)\n", names[mode], ts, moved / ts / 1e6, to,
           moved / to / 1e6);
  }
  if (other2 > 0) {
    printf("read stream in the third place (%d GiB), A and B as before:\n", other2);
    const float ts = run(5, 2 * GB, (size_t)same * GB, (size_t)other2 * GB), to = run(5, 2 * GB, (size_t)other * GB, (size_t)other2 * GB);
    printf("%-44s same class %.4f ms (%5.0f GB/s)   other class %.4f ms (%5.0f GB/s)\n", names[5], ts, 3 * bytes / ts / 1e6, to, 3 * bytes / to / 1e6);
  }
  return 0;
}
