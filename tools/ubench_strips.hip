// HBM access-pattern microbenchmark for gfx950: every wave copies 8 KB rows (one 16-byte load / store per lane, 8 per
// row) walking a strip of L consecutive rows, strips dealt to waves in order -- the access pattern of the wave-level
// MDCT kernels without their arithmetic.  Reports GB/s (read + written bytes) for 1R:1W and 1R:2W at several L,
// with and without a one-row register prefetch.   (design aid, not product)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_strips.hip -o audiocodec_amd/lib/ub/ubench_strips
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NOUT, bool PREFETCH>
__global__ __launch_bounds__(256) void k_strips(const v4f* __restrict__ in, v4f* __restrict__ o1, v4f* __restrict__ o2,
                                                long long nrows, int L, long long nstrips) {
  const int lane = threadIdx.x & 63;
  const long long strip = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (strip >= nstrips) return;
  const long long r0 = strip * L;
  const long long r1 = (r0 + L < nrows) ? r0 + L : nrows;
  v4f cur[8], nxt[8];
  if (PREFETCH) {
#pragma unroll
    for (int i = 0; i < 8; ++i) nxt[i] = in[r0 * 512 + 64 * i + lane];
  }
  for (long long r = r0; r < r1; ++r) {
    if (PREFETCH) {
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
      if (r + 1 < r1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = in[(r + 1) * 512 + 64 * i + lane];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = in[r * 512 + 64 * i + lane];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o1[r * 512 + 64 * i + lane] = cur[i] * 1.5f;
      if (NOUT == 2) o2[r * 512 + 64 * i + lane] = cur[i] + 1.0f;
    }
  }
}

// pattern B: a workgroup of 4 waves owns 4 T consecutive rows; wave w copies rows base + 4 t + w (t = 0..T-1), and
// also re-reads row base + 4 t + w - 1 (the block the MDCT fold needs again) when HALO is set
template <bool HALO, bool XCD>
__global__ __launch_bounds__(256) void k_inter(const v4f* __restrict__ in, v4f* __restrict__ o1, v4f* __restrict__ o2,
                                               long long nrows, int T, int nwg) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int g = blockIdx.x;
  if (XCD) g = (g & 7) * (nwg / 8) + (g >> 3);   // consecutive logical workgroups on one XCD (nwg multiple of 8)
  const long long base = (long long)g * 4 * T;
  for (int t = 0; t < T; ++t) {
    const long long r = base + 4 * t + w;
    if (r >= nrows) return;
    v4f cur[8], prv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = in[r * 512 + 64 * i + lane];
    if (HALO && r > 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) prv[i] = in[(r - 1) * 512 + 64 * i + lane];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) prv[i] = v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o1[r * 512 + 64 * i + lane] = cur[i] * 1.5f + prv[i];
      o2[r * 512 + 64 * i + lane] = cur[i] + 1.0f;
    }
  }
}

// pattern A with the halo row read first (what the strip kernels do) and an optional XCD-contiguous strip order
template <bool XCD>
__global__ __launch_bounds__(256) void k_strips_halo(const v4f* __restrict__ in, v4f* __restrict__ o1, v4f* __restrict__ o2,
                                                     long long nrows, int L, long long nstrips, int nwg) {
  const int lane = threadIdx.x & 63;
  int g = blockIdx.x;
  if (XCD) g = (g & 7) * (nwg / 8) + (g >> 3);
  const long long strip = (long long)g * 4 + (threadIdx.x >> 6);
  if (strip >= nstrips) return;
  const long long r0 = strip * L;
  const long long r1 = (r0 + L < nrows) ? r0 + L : nrows;
  v4f cur[8], nxt[8], acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (r0 > 0) ? in[(r0 - 1) * 512 + 64 * i + lane] : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) nxt[i] = in[r0 * 512 + 64 * i + lane];
  for (long long r = r0; r < r1; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    if (r + 1 < r1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) nxt[i] = in[(r + 1) * 512 + 64 * i + lane];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o1[r * 512 + 64 * i + lane] = cur[i] * 1.5f + acc[i];
      o2[r * 512 + 64 * i + lane] = cur[i] + 1.0f;
      acc[i] = cur[i];
    }
  }
}

// pattern C: persistent waves; wave id w of W copies rows w, w + W, w + 2 W, ... and re-reads row r - 1 (HALO);
// NT threads per workgroup, XCD: consecutive logical workgroups on one XCD
template <int NT, bool HALO, bool XCD>
__global__ __launch_bounds__(NT) void k_persist(const v4f* __restrict__ in, v4f* __restrict__ o1, v4f* __restrict__ o2,
                                                long long nrows, int nwg) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int g = blockIdx.x;
  if (XCD) g = (g & 7) * (nwg / 8) + (g >> 3);
  const long long W = (long long)nwg * (NT / 64);
  for (long long r = (long long)g * (NT / 64) + w; r < nrows; r += W) {
    v4f cur[8], prv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = in[r * 512 + 64 * i + lane];
    if (HALO && r > 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) prv[i] = in[(r - 1) * 512 + 64 * i + lane];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) prv[i] = v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o1[r * 512 + 64 * i + lane] = cur[i] * 1.5f + prv[i];
      o2[r * 512 + 64 * i + lane] = cur[i] + 1.0f;
    }
  }
}

int main() {
  const long long nrows = 256ll * 469;   // 8 KB rows: 984 MB per tensor (the bench workload's frame count)
  const size_t bytes = (size_t)nrows * 8192;
  v4f *in, *o1, *o2;
  hipMalloc(&in, bytes);
  hipMalloc(&o1, bytes);
  hipMalloc(&o2, bytes);
  hipMemset(in, 0x3c, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-6s %-10s %-10s %-10s %-10s   (GB/s of read + written bytes)\n", "L", "1R1W", "1R1W+pf", "1R2W", "1R2W+pf");
  for (int L : {1, 2, 4, 8, 15, 16, 32, 64, 469}) {
    const long long nstrips = (nrows + L - 1) / L;
    const unsigned grid = (unsigned)((nstrips + 3) / 4);
    printf("%-6d", L);
    for (int v = 0; v < 4; ++v) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        switch (v) {
          case 0: hipLaunchKernelGGL((k_strips<1, false>), dim3(grid), dim3(256), 0, 0, in, o1, o2, nrows, L, nstrips); break;
          case 1: hipLaunchKernelGGL((k_strips<1, true>), dim3(grid), dim3(256), 0, 0, in, o1, o2, nrows, L, nstrips); break;
          case 2: hipLaunchKernelGGL((k_strips<2, false>), dim3(grid), dim3(256), 0, 0, in, o1, o2, nrows, L, nstrips); break;
          case 3: hipLaunchKernelGGL((k_strips<2, true>), dim3(grid), dim3(256), 0, 0, in, o1, o2, nrows, L, nstrips); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double moved = (double)bytes * (v < 2 ? 2 : 3);
      printf(" %-10.0f", moved / best / 1e6);
    }
    printf("\n");
  }
  printf("\nuseful GB/s (3 x tensor bytes / time), 1R2W:\n%-6s %-12s %-12s %-12s %-12s %-12s %-12s\n", "L|T", "strips+halo", "same,xcd", "inter", "inter,xcd",
         "inter+halo", "same,xcd");
  for (int L : {1, 2, 3, 4, 6, 8, 12, 16}) {
    printf("%-6d", L);
    for (int v = 0; v < 6; ++v) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        const long long nstrips = (nrows + L - 1) / L;
        int nwg = (int)((nstrips + 3) / 4);
        nwg = (nwg + 7) / 8 * 8;
        hipEventRecord(e0);
        switch (v) {
          case 0: hipLaunchKernelGGL((k_strips_halo<false>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, L, nstrips, nwg); break;
          case 1: hipLaunchKernelGGL((k_strips_halo<true>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, L, nstrips, nwg); break;
          case 2: hipLaunchKernelGGL((k_inter<false, false>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, L, nwg); break;
          case 3: hipLaunchKernelGGL((k_inter<false, true>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, L, nwg); break;
          case 4: hipLaunchKernelGGL((k_inter<true, false>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, L, nwg); break;
          case 5: hipLaunchKernelGGL((k_inter<true, true>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, L, nwg); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf(" %-12.0f", (double)bytes * 3 / best / 1e6);
    }
    printf("\n");
  }
  printf("\npersistent waves, 1R2W + halo re-read, useful GB/s:\n%-28s %-10s %-10s\n", "config", "plain", "xcd");
  for (int cfg = 0; cfg < 6; ++cfg) {
    const int nwgs[6] = {256, 512, 768, 1024, 2048, 256};
    const int nts[6] = {768, 384, 256, 256, 256, 1024};
    char name[64];
    snprintf(name, sizeof(name), "%d wg x %d threads", nwgs[cfg], nts[cfg]);
    printf("%-28s", name);
    for (int x = 0; x < 2; ++x) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        const int nwg = nwgs[cfg];
        if (nts[cfg] == 768) {
          if (x) hipLaunchKernelGGL((k_persist<768, true, true>), dim3(nwg), dim3(768), 0, 0, in, o1, o2, nrows, nwg);
          else hipLaunchKernelGGL((k_persist<768, true, false>), dim3(nwg), dim3(768), 0, 0, in, o1, o2, nrows, nwg);
        } else if (nts[cfg] == 384) {
          if (x) hipLaunchKernelGGL((k_persist<384, true, true>), dim3(nwg), dim3(384), 0, 0, in, o1, o2, nrows, nwg);
          else hipLaunchKernelGGL((k_persist<384, true, false>), dim3(nwg), dim3(384), 0, 0, in, o1, o2, nrows, nwg);
        } else if (nts[cfg] == 1024) {
          if (x) hipLaunchKernelGGL((k_persist<1024, true, true>), dim3(nwg), dim3(1024), 0, 0, in, o1, o2, nrows, nwg);
          else hipLaunchKernelGGL((k_persist<1024, true, false>), dim3(nwg), dim3(1024), 0, 0, in, o1, o2, nrows, nwg);
        } else {
          if (x) hipLaunchKernelGGL((k_persist<256, true, true>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, nwg);
          else hipLaunchKernelGGL((k_persist<256, true, false>), dim3(nwg), dim3(256), 0, 0, in, o1, o2, nrows, nwg);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf(" %-10.0f", (double)bytes * 3 / best / 1e6);
    }
    printf("\n");
  }
  return 0;
}
