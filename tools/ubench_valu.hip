// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction at 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o gpurun_out/ubench_valu   (design aid, not product)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <typename T> __device__ T mk(float v);
template <> __device__ float mk<float>(float v) { return v; }
template <> __device__ v2f mk<v2f>(float v) { return v2f{v, v + 1.f}; }
__device__ float fold(float v) { return v; }
__device__ float fold(v2f v) { return v.x + v.y; }
#define REP8(x) x x x x x x x x
#define BODY(T, NAME, ASM)                                                                              \
  __global__ __launch_bounds__(256) void NAME(float* out, long long* cyc, int iters) {               \
    T a0 = mk<T>(1.f + threadIdx.x), a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,   \
        a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                                  \
    T b = mk<T>(1.0001f), c = mk<T>(1e-3f);                                                   \
    long long t0 = __builtin_amdgcn_s_memtime();                                                      \
    for (int i = 0; i < iters; ++i) {                                                                 \
      REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                        "+v"(a7) : "v"(b), "v"(c));)                                                  \
    }                                                                                                 \
    long long t1 = __builtin_amdgcn_s_memtime();                                                      \
    T s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = fold(s);                                           \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;    \
  }

// each ASM string = 8 independent instructions
BODY(float, k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
BODY(v2f, k_pkfma, "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
BODY(v2f, k_pkadd, "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
BODY(float, k_add, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
            "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
BODY(float, k_max, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
            "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
BODY(float, k_log, "v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
            "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n")
BODY(float, k_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
             "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
BODY(float, k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
            "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
BODY(v2f, k_pkmov, "v_pk_mov_b32 %0, %8, %9\n v_pk_mov_b32 %1, %8, %9\n v_pk_mov_b32 %2, %8, %9\n v_pk_mov_b32 %3, %8, %9\n"
              "v_pk_mov_b32 %4, %8, %9\n v_pk_mov_b32 %5, %8, %9\n v_pk_mov_b32 %6, %8, %9\n v_pk_mov_b32 %7, %8, %9\n")
BODY(float, k_lshladd, "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
                "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n")
BODY(float, k_dpp, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")

typedef void (*kern_t)(float*, long long*, int);

int main() {
  struct { const char* name; kern_t k; } ks[] = {{"v_fma_f32", k_fma}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_add_f32", k_pkadd},
      {"v_add_f32", k_add}, {"v_max_f32", k_max}, {"v_log_f32", k_log}, {"v_sqrt_f32", k_sqrt}, {"v_mov_b32", k_mov},
      {"v_pk_mov_b32", k_pkmov}, {"v_lshl_add_u32", k_lshladd}, {"v_add_f32_dpp", k_dpp}};
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&cyc, 256 * 8 * 4 * sizeof(long long));
  const int iters = 2000;
  printf("%-16s %s\n", "instruction", "cycles per wave-instruction per SIMD at 1 / 2 / 3 / 4 waves per SIMD  (wave-visible cycles per instr)");
  for (auto& e : ks) {
    printf("%-16s", e.name);
    for (int wps = 1; wps <= 4; ++wps) {
      const int blocks = 256 * wps;   // 256-thread blocks: one wave per SIMD each; wps blocks per CU
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
      std::vector<long long> h(blocks * 4);
      hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      double s = 0;
      for (auto v : h) s += (double)v;
      const double per_wave = s / h.size() / (iters * 64.0);   // cycles seen by one wave per instruction
      printf("  %6.2f (%5.2f)", per_wave / wps, per_wave);
    }
    printf("\n");
  }
  return 0;
}
