// How many workgroups of a given LDS size / register count does one gfx950 CU hold?  Each workgroup spins for a fixed
// number of cycles; with G = 256 CUs x k workgroups the launch takes ceil(k / resident-per-CU) spin periods.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_occupancy.hip -o audiocodec_amd/lib/ub/ubench_occupancy
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k_spin(float* out, long long cycles) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  float v = lds[(threadIdx.x + 1) % blockDim.x];
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) out[0] = v;
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const long long spin = 200000;   // cycles of the 100 MHz-or-shader-clock counter; only ratios matter
  printf("%-8s %-8s %s\n", "threads", "LDS B", "time for 256 x {1,2,3,4,6,8} workgroups, relative to 256 x 1");
  for (int threads : {256, 384}) {
    for (int ldsb : {40960, 53760, 54272, 54784, 61952, 65536, 79968, 80384, 81408, 81920, 98304}) {
      printf("%-8d %-8d", threads, ldsb);
      float base = 0;
      for (int k : {1, 2, 3, 4, 6, 8}) {
        float best = 1e30f;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k_spin, dim3(256 * k), dim3(threads), ldsb, 0, out, spin);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        if (k == 1) base = best;
        printf("  %5.2f", best / base);
      }
      printf("\n");
    }
  }
  return 0;
}
