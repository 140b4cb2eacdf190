// Band x band Toeplitz product of the masking model (sum_i Q_i g[64 - i + j], 64 Bark bands, two signals per wave)
// three ways on one wave, for accuracy and cycle counts (design aid, not product):
//   0  f32 VALU: Q broadcast through LDS, g window read per lane (the form of psy_stage in ac_fast.hip)
//   1  bf16 MFMA: v_mfma_f32_4x4x4_16b_bf16, 16 blocks = 16 column tiles of S, A (the Q rows) broadcast from block s
//      (cbsz = 4, abid = s), 16 instructions for the 64-deep contraction
//   2  split bf16 MFMA: Q = hi + lo (rows 0,1 = hi, rows 2,3 = lo of the same 4-row A tile), S = hi + lo (two B tables):
//      32 instructions, ~16 mantissa bits
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_spread.hip -o gpurun_out/ubench_mfma_spread
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v2b __attribute__((ext_vector_type(2)));

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                          \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int COPY_STRIDE = 320;           // bytes between the four shifted copies of the reversed bf16 prototype
constexpr int TAB_BYTES = 4 * COPY_STRIDE; // one table (hi or lo)

// ---- f32 VALU form -------------------------------------------------------------------------------------
__device__ __forceinline__ v2f spread_valu(v2f Q, char* buf, const float* g, int lane) {
  wave_sync();
  *reinterpret_cast<v2f*>(buf + 8 * lane) = Q;
  wave_sync();
  v2f acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
  const float* gp = g + 64 + lane;
#pragma unroll 8
  for (int i = 0; i < 64; i += 2) {
    const v4f qq = *reinterpret_cast<const v4f*>(buf + 8 * i);
    acc0 += v2f{qq.x, qq.y} * gp[-i];
    acc1 += v2f{qq.z, qq.w} * gp[-i - 1];
  }
  return acc0 + acc1;
}

// ---- MFMA forms -----------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const v2b r = __builtin_convertvector(v2f{a, b}, v2b);
  return __builtin_bit_cast(uint32_t, r);
}
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {   // lane k of every quad to the whole quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);
}

template <int S, int MODE>
struct Steps {
  static __device__ __forceinline__ void run(const v4s a, const char* bhi, const char* blo, v4f& d0, v4f& d1) {
    Steps<S - 1, MODE>::run(a, bhi, blo, d0, d1);
    const v4s b = *reinterpret_cast<const v4s*>(bhi + 8 * S);
    if (MODE == 2) {
      d0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d0, 4, S, 0);
      const v4s bl = *reinterpret_cast<const v4s*>(blo + 8 * S);
      d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, bl, d1, 4, S, 0);
    } else if (S & 1) {
      d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d1, 4, S, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d0, 4, S, 0);
    }
  }
};
template <int MODE>
struct Steps<-1, MODE> {
  static __device__ __forceinline__ void run(const v4s, const char*, const char*, v4f&, v4f&) {}
};

// tab = LDS copy of the hi table followed by the lo table
template <int MODE>
__device__ __forceinline__ v2f spread_mfma(v2f Q, const char* tab, int lane) {
  const uint32_t whi = pk_bf16(Q.x, Q.y);
  uint32_t wlo = 0;
  if (MODE == 2) {
    const float hx = __uint_as_float(whi << 16), hy = __uint_as_float(whi & 0xffff0000u);
    wlo = pk_bf16(Q.x - hx, Q.y - hy);
  }
  // quad-local 4 x 4 transpose of 16-bit values: lane 4 s + i ends up with row i (0: hi of signal 0, 1: hi of signal 1,
  // 2, 3: the lo parts) of bands 4 s .. 4 s + 3 = the A tile of step s, which cbsz / abid broadcast to all 16 blocks
  const int i = lane & 3;
  const bool lo_row = i >= 2;
  // (both broadcasts are evaluated by every lane before the select: DPP reads need the whole quad active)
  const uint32_t h0 = quad_bcast<0>(whi), h1 = quad_bcast<1>(whi), h2 = quad_bcast<2>(whi), h3 = quad_bcast<3>(whi);
  uint32_t c0 = h0, c1 = h1, c2 = h2, c3 = h3;
  if (MODE == 2) {
    const uint32_t l0 = quad_bcast<0>(wlo), l1 = quad_bcast<1>(wlo), l2 = quad_bcast<2>(wlo), l3 = quad_bcast<3>(wlo);
    c0 = lo_row ? l0 : h0, c1 = lo_row ? l1 : h1, c2 = lo_row ? l2 : h2, c3 = lo_row ? l3 : h3;
  } else {
    c0 = lo_row ? 0u : h0, c1 = lo_row ? 0u : h1, c2 = lo_row ? 0u : h2, c3 = lo_row ? 0u : h3;
  }
  const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;
  const uint32_t a01 = __builtin_amdgcn_perm(c1, c0, sel);
  const uint32_t a23 = __builtin_amdgcn_perm(c3, c2, sel);
  const uint2 au = {a01, a23};
  const v4s a = __builtin_bit_cast(v4s, au);
  // B tile of step s, lane l (column l): g[64 - 4 s - k + l], k = 0..3 = four consecutive entries of the reversed
  // prototype; copy (l & 3) of the table is shifted so that the read is 8-byte aligned
  uint32_t boff = (lane & 3) * COPY_STRIDE + 8 * (16 - (lane >> 2));
  asm volatile("" : "+v"(boff));   // keep the 16 / 32 tile reads inside the caller's loop (hoisted, they pin 32 / 64 registers)
  const char* bhi = tab + boff;
  const char* blo = bhi + TAB_BYTES;
  v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
  Steps<15, MODE>::run(a, bhi, blo, d0, d1);
  const v4f d = d0 + d1;
  return MODE == 2 ? v2f{d.x + d.z, d.y + d.w} : v2f{d.x, d.y};
}

template <int MODE>
__global__ __launch_bounds__(256) void k_spread(const v2f* __restrict__ Qin, const float* __restrict__ g,
                                                 const uint16_t* __restrict__ tabs, v2f* __restrict__ out,
                                                 long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024 + 512 + 2 * TAB_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* buf = lds + 1024 * wave;
  float* gl = reinterpret_cast<float*>(lds + 4096);
  char* tab = lds + 4096 + 512;
  for (int i = threadIdx.x; i < 128; i += blockDim.x) gl[i] = g[i];
  for (int i = threadIdx.x; i < TAB_BYTES; i += blockDim.x) reinterpret_cast<uint16_t*>(tab)[i] = tabs[i];
  __syncthreads();
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  v2f Q = Qin[w * 64 + lane];
  v2f r = {0.f, 0.f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    r = MODE == 0 ? spread_valu(Q, buf, gl, lane) : spread_mfma<MODE>(Q, tab, lane);
    if (it + 1 < iters) Q = Q + r * 1e-30f;   // dependence between iterations
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[w * 64 + lane] = r;
  if (lane == 0) cyc[w] = t1 - t0;
}

static uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  // spreading prototype at 48 kHz, 64 bands, alpha 0.6 (psychoacoustic.py:212-230)
  const int M = 64;
  const double max_bark = 6.0 * asinh(24000.0 / 600.0), alpha = 0.6;
  std::vector<float> g(128);
  for (int m = 0; m < 128; ++m) {
    const double z = -max_bark + 2.0 * max_bark * m / (2 * M - 1);
    const double f = 15.81 + 7.5 * (z + 0.474) - 17.5 * sqrt(1.0 + (z + 0.474) * (z + 0.474));
    g[m] = (float)pow(10.0, alpha * f / 10.0);
  }
  // tables: copy c, element y = rev[y - c], rev[m] = g[128 - m] (m = 1..127)
  std::vector<uint16_t> tabs(TAB_BYTES, 0);   // TAB_BYTES uint16 = hi table + lo table
  for (int c = 0; c < 4; ++c)
    for (int y = 0; y < 132; ++y) {
      const int m = y - c;
      if (m < 1 || m > 127) continue;
      const float v = g[128 - m];
      const uint16_t hi = bf16_rne(v);
      const uint16_t lo = bf16_rne(v - bf16_f(hi));
      tabs[(c * COPY_STRIDE) / 2 + y] = hi;
      tabs[(TAB_BYTES + c * COPY_STRIDE) / 2 + y] = lo;
    }
  const int blocks = 1024, waves = blocks * 4;
  std::vector<v2f> Q(waves * 64);
  srand(7);
  for (auto& q : Q) {
    const double e0 = -8.4 + 10.2 * rand() / RAND_MAX, e1 = -8.4 + 10.2 * rand() / RAND_MAX;   // (1e-14 .. 1e3)^0.6
    q = v2f{(float)pow(10.0, e0), (float)pow(10.0, e1)};
  }
  v2f *dQ, *dO;
  float* dg;
  uint16_t* dt;
  long long* dc;
  CK(hipMalloc(&dQ, Q.size() * 8));
  CK(hipMalloc(&dO, Q.size() * 8));
  CK(hipMalloc(&dg, 512));
  CK(hipMalloc(&dt, TAB_BYTES * 2));
  CK(hipMalloc(&dc, waves * 8));
  CK(hipMemcpy(dQ, Q.data(), Q.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g.data(), 512, hipMemcpyHostToDevice));
  CK(hipMemcpy(dt, tabs.data(), TAB_BYTES * 2, hipMemcpyHostToDevice));
  std::vector<v2f> O(Q.size());
  std::vector<long long> cyc(waves);
  for (int mode = 0; mode < 3; ++mode) {
    auto launch = [&](int nb, int iters) {
      if (mode == 0) k_spread<0><<<nb, 256>>>(dQ, dg, dt, dO, dc, iters);
      if (mode == 1) k_spread<1><<<nb, 256>>>(dQ, dg, dt, dO, dc, iters);
      if (mode == 2) k_spread<2><<<nb, 256>>>(dQ, dg, dt, dO, dc, iters);
      CK(hipDeviceSynchronize());
    };
    launch(blocks, 1);
    CK(hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0, rms = 0;
    for (int w = 0; w < 256; ++w)
      for (int j = 0; j < 64; ++j)
        for (int c = 0; c < 2; ++c) {
          double ref = 0;
          for (int i = 0; i < 64; ++i) ref += (double)(c ? Q[w * 64 + i].y : Q[w * 64 + i].x) * g[64 - i + j];
          const double got = c ? O[w * 64 + j].y : O[w * 64 + j].x;
          const double e = fabs(got - ref) / ref;
          worst = fmax(worst, e);
          rms += e * e;
        }
    printf("mode %d: max rel err %.3g  rms %.3g\n", mode, worst, sqrt(rms / (256 * 128)));
    // cycles per product: 1 workgroup per CU (1 wave / SIMD), then 2 and 3 per CU
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
      const int iters = 2000;
      launch(256 * per_cu, iters);
      CK(hipMemcpy(cyc.data(), dc, 256 * per_cu * 4 * 8, hipMemcpyDeviceToHost));
      double s = 0;
      for (int i = 0; i < 256 * per_cu * 4; ++i) s += cyc[i];
      printf("   %d wave(s)/SIMD: %.1f memtime ticks per product per wave\n", per_cu, s / (256 * per_cu * 4) / iters);
    }
  }
  return 0;
}
