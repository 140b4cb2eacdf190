// Internal declarations shared by the translation units of libaudiocodec_amd.so.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/audiocodec_amd.h"
#include "../../include/audiocodec_amd_testing.h"
#include "ac_tables.h"

namespace ac {

void set_error(const char* fmt, ...);

#define AC_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ac::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return AC_EHIP;                                                                        \
    }                                                                                        \
  } while (0)

#define AC_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      ac::set_error(__VA_ARGS__);  \
      return AC_EINVAL;            \
    }                              \
  } while (0)

// The kernels move rows with 16-byte (8-byte: mono, 16-bit PCM) vector accesses: tensor base addresses must be 16-byte
// aligned (any allocation is; a view that starts inside one may not be).
#define AC_REQUIRE_ALIGNED(...)                                                                                  \
  do {                                                                                                           \
    const void* ac_ptrs_[] = {__VA_ARGS__};                                                                      \
    for (const void* ac_p_ : ac_ptrs_)                                                                           \
      if ((reinterpret_cast<uintptr_t>(ac_p_) & 15) != 0) {                                                      \
        ac::set_error("tensor address %p is not 16-byte aligned (pass the start of an allocation or an aligned view)", ac_p_); \
        return AC_EINVAL;                                                                                        \
      }                                                                                                          \
  } while (0)

// Makes `device` current for the lifetime of the guard (no-op when it already is).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) {
      switched = (hipSetDevice(device) == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

extern int g_force_generic;

// ---- element-wise pieces shared by the stand-alone utilities (ac_generic.hip) and the fused encode epilogue (ac_fast.hip):
// one definition, so that fused and un-fused results agree bit for bit
// counter-based generator: a 64-bit mix (splitmix64) of (seed, pair index) -> Box-Muller, both outputs used:
// elements 2p and 2p+1 of the flattened tensor take R cos(theta) and R sin(theta)
__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
#ifdef __HIPCC__
__device__ __forceinline__ void normal_pair(uint64_t key, uint64_t pair, float& g0, float& g1) {
  const uint64_t r = mix64(key ^ pair);
  const float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  const float u2 = (float)(uint32_t)((r >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);   // [0, 1): one revolution
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
  g0 = rad * __builtin_amdgcn_cosf(u2);   // v_cos_f32 / v_sin_f32 take revolutions
  g1 = rad * __builtin_amdgcn_sinf(u2);
}
// add_noise (psychoacoustic.py:150-167): x + thr * Normal(0, 1/6), g a standard normal
__device__ __forceinline__ float noisy_of(float x, float thr, float g) { return __builtin_fmaf(thr, g * (1.0f / 6.0f), x); }
// amplitude_to_dB (psychoacoustic.py:83-85) and its [0, 1] normalisation (:98-100): 10 log10 = 3.0103 log2 (v_log_f32)
__device__ __forceinline__ float db_of(float v, int norm) {
  float dB = __builtin_fmaf(3.0102999566398120f, __builtin_amdgcn_logf(fmaxf(1e-14f, v * v)), 120.f);
  // (the reference subtracts _dB_MIN = amplitude_to_dB(eps), the same expression, so its normalised scale starts at
  // exactly 0: the clamp keeps that under the rounding of the multiply-add above)
  if (norm) dB = fmaxf((dB + 20.f) * (1.0f / 140.f), 0.f);
  return dB;
}
#endif

typedef __bf16 bf16_t;   // storage type of the AC_BF16 tensors (device code converts with v_cvt_pk_bf16_f32)
typedef _Float16 f16_t;  // storage type of the AC_F16 tensors (filter bank only)

// layout of the run-structured masking-model image and of its per-frame LDS slot (build_runs in ac_psy_mid.hip; the fields
// are those of runs::RunsParams, ac_psy_runs_dev.h)
struct RunsLayout {
  int words = 0;
  int lw = 0, kb = 0, n4 = 0, n16 = 0, n64 = 0, o4 = 0, o16 = 0, o64 = 0, oz = 0, slot = 0;
  int off_S = 0, off_bc = 0, off_bd = 0, off_lst = 0, off_bw = 0, off_idx = 0;
};

}  // namespace ac

// ---- plan objects ---------------------------------------------------------------------------

struct ac_mdct_plan {
  int N = 0, window = 0, device = 0;
  int cus = 0;                 // compute units of the device (sizes the persistent launches)
  int fast = 0;                // 1: wave-level FFT kernels available for this N
  int pre = AC_F64;            // arithmetic type the constants were computed in (the reference's precompute_dtype)
  int fold4 = 0;               // 1: the wave-level kernels run their FOLD4 form (fold blocks that are not rotations)
  int adjoint = 0;             // 1: a plan made by ac_mdct_plan_adjoint (the transposed filter bank of another plan)
  ac::FoldCoef coef;           // the plan's fold coefficients (host copy)
  float* d_coef = nullptr;     // [8][N/2]  a1 a2 a3 a4 s1 s2 s3 s4
  float* d_coefv = nullptr;    // the same coefficients in the order of the 16-byte wave kernels (k_fwd_wave_v / k_inv_wave_v), N % 4 == 0
  float* d_ctab = nullptr;     // [8N]      cos(pi i / (4N)), generic kernels
  double* d_coef64 = nullptr;  // the same two tables in float64 (AC_F64 entry points)
  double* d_ctab64 = nullptr;
  // fast-path tables (ac_fast.hip): per-FFT-element fold coefficients and twiddles
  float* d_fast = nullptr;
  size_t fast_bytes = 0;
};

struct ac_psy_plan {
  int N = 0, M = 0, device = 0;
  double sample_rate = 0, alpha = 0;
  int fast = 0;                // 1: the fused wave-level epilogue supports this (N, M, table shape)
  int pre = AC_F64;            // arithmetic type the constants were computed in (the reference's precompute_dtype)
  int spread = 0;              // AC_SPREAD_*: form of the band x band spreading product in the wave-level kernels
  ac::PsyTables host;
  int32_t* d_wb_ptr = nullptr; int32_t* d_wb_idx = nullptr; float* d_wb_val = nullptr; int wb_max = 0;
  int32_t* d_wi_ptr = nullptr; int32_t* d_wi_idx = nullptr; float* d_wi_val = nullptr; int wi_max = 0;
  // transposed walks for the backward pass: W by bin, W_inv by band
  int32_t* d_wf_ptr = nullptr; int32_t* d_wf_idx = nullptr; float* d_wf_val = nullptr;
  int32_t* d_vb_ptr = nullptr; int32_t* d_vb_idx = nullptr; float* d_vb_val = nullptr;
  float* d_S = nullptr;        // [M, M]
  float* d_quiet = nullptr;    // [M]
  float* d_beta = nullptr;     // [M]
  // float64 constants (AC_F64 entry points): CSR values, S, quiet, beta = linspace(0, max_bark, M) in float64
  double* d_wb_val64 = nullptr; double* d_wi_val64 = nullptr;
  double* d_wf_val64 = nullptr; double* d_vb_val64 = nullptr;   // ... of the transposed walks (backward in float64)
  double* d_S64 = nullptr; double* d_quiet64 = nullptr; double* d_beta64 = nullptr;
  // fast-path tables (ac_fast.hip)
  float* d_fast = nullptr;
  size_t fast_bytes = 0;
  // wave-level masking model for general band layouts (ac_psy_mid.hip): any even filter_bands_n <= 1024, <= 64 bands
  int mid = 0;
  uint32_t* d_mid = nullptr;
  int mid_words = 0;
  int mid_wi_w = 0, mid_off_S = 0, mid_off_band = 0, mid_off_wbe = 0, mid_off_wi = 0;   // layout of the image (build_mid)
  // ... in its run-structured form (ac_psy_runs_dev.h): the form every kernel of that tier runs when the plan's tables have
  // the structure (runs = 1; the band walk above stays for tables that do not)
  int runs = 0;
  uint32_t* d_runs = nullptr;
  ac::RunsLayout runs_lay;
  int cus = 0;                 // compute units of the device (sizes the launches)
};

struct ac_stream {
  const ac_mdct_plan* plan = nullptr;
  int B = 0, C = 0;
  int N = 0, device = 0;           // copies of the plan's (the stream may outlive the plan object on teardown)
  float* d_prev_block = nullptr;   // analysis state  [B, N, C]
  float* d_prev_tmp = nullptr;     // double buffer for the analysis state (written by the kernel that reads the other)
  float* d_tail = nullptr;         // synthesis state [B, C, N/2]  (u_last[h .. N-1])
  float* d_tail_tmp = nullptr;     // double buffer for the synthesis state
  // the buffers swap roles with every chunk; "home" = the pair the state is moved back to by ac_stream_settle (and by
  // ac_stream_run on entry and exit, and by ac_stream_reset), so that launches captured into a HIP graph address the
  // state where a later replay will find it
  float* d_prev_home = nullptr;
  float* d_tail_home = nullptr;
  // float64 streams (AC_F64 through ac_stream_*_typed): the same state in double, allocated by the first float64 call
  double* d_prev64 = nullptr;      // [B, N, C]
  double* d_tail64 = nullptr;      // [B, C, N/2]
  double* d_tail64_tmp = nullptr;
};

// ---- kernel launchers (each returns an AC_* status) -----------------------------------------
namespace ac {

// generic O(N^2) kernels: any even N, any C, any M
int lds_fft_tier_of(const ac_mdct_plan* p, int C);   // ac_generic.hip: 2 / 1 / 0, see ac_mdct_plan_tier
// the 16-byte kernels of the LDS-FFT tier on mono rows (ac_wave_rows.hip)
int launch_fwd_wave_mono(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin, int F,
                         hipStream_t s);
int launch_inv_wave_mono(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out, int B, int Kp,
                         int nblk, hipStream_t s);
// 16-bit PCM at the boundary of the LDS-FFT tier (filters_n 120 / 240 / 480 / 960 / 1920 / 576 / 1152, mono / stereo)
bool lds_fft_serves_pcm16(const ac_mdct_plan* p, int C);
int launch_fwd_lds_pcm16(const ac_mdct_plan* p, const int16_t* x, float* X, int B, int K, int C, hipStream_t s);
int launch_inv_lds_pcm16(const ac_mdct_plan* p, const float* X, int16_t* x, int B, int Kp, int C, hipStream_t s);
int launch_fwd_wave_mono_pcm16(const ac_mdct_plan* p, const int16_t* x, float* X, int B, int Kin, int F, hipStream_t s);
int launch_inv_wave_mono_pcm16(const ac_mdct_plan* p, const float* X, int16_t* x, int B, int Kp, int nblk, hipStream_t s);
// the fused encode of the LDS-FFT tier (ac_wave_enc.hip): the 16-byte analysis kernels with the run-structured masking model
// on the frame while it is in LDS; filters_n 108 ... 4096 with an instance, float32 mono / stereo
bool wave_encode_fuses(const ac_mdct_plan* p, const ac_psy_plan* psy, int C, const void* x, const void* X, const void* thr);
int launch_enc_wave(const ac_mdct_plan* p, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr, float drown,
                    const float* prev_block, int B, int Kin, int F, int C, hipStream_t s);
// ... and on channel pairs of any channel count / rows off the 16-byte grid (ac_wave_rows2.hip)
int launch_fwd_wave_team(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin, int F, int C,
                         hipStream_t s);
int launch_inv_wave_team(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out, int B, int Kp,
                         int nblk, int C, hipStream_t s);
int launch_fwd_wave_strided(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin, int F,
                            int C, hipStream_t s);
int launch_inv_wave_strided(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out, int B,
                            int Kp, int nblk, int C, hipStream_t s);
int launch_fwd_generic(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin,
                       int F, int C, hipStream_t s);
int launch_inv_generic(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out,
                       int B, int Kp, int nblk, int C, hipStream_t s);
int launch_tonality_generic(const ac_psy_plan* p, const float* X, float* t, int B, int F, int C, hipStream_t s);
int launch_threshold_generic(const ac_psy_plan* p, const float* X, const float* t, float drown, float* thr, int B,
                             int F, int C, hipStream_t s);

// wave-level FFT kernels (ac_fast.hip)
bool fast_mdct_supported(int N, const FoldCoef& c);
// filters_n 512 / 256 run several frames per wave (2 / 4); those kernels serve float32 mono / stereo tensors with at least
// one block (streaming state included) -- everything else at these sizes takes the LDS-FFT tier
int fast_mdct_frames_per_wave(int N);
bool fast_multi_serves(const ac_mdct_plan* p, int C, int iof, int blocks);
// ... and whether the masking model of `psy` (general band layouts, ac_psy_mid_dev.h) rides in the same launch
bool fast_multi_fuses(const ac_mdct_plan* p, const ac_psy_plan* psy, int C, int iof, int blocks);
bool fast_psy_supported(const ac_psy_plan* p);
int fast_mdct_plan_init(ac_mdct_plan* p);
int fast_psy_plan_init(ac_psy_plan* p);
// psy may be null (plain transform).  X/t/thr as in ac_encode_fused.
// iof: 0 = float32 tensors; 1 = int16 PCM on the PCM side (x), spectra float32; 2 = bfloat16 tensors throughout (every
// pointer then addresses 2-byte elements; C = 1 or 2 only).  prev_block / tail state must be null unless iof == 0
// state_out (iof == 0 only): receives block Kin-1 of every signal, the next chunk's prev_block (must not alias prev_block)
int launch_fwd_fast(const ac_mdct_plan* p, const ac_psy_plan* psy, const void* x, int iof, float* X, float* t,
                    float* thr, float drown, const float* prev_block, int B, int Kin, int F, int C, hipStream_t s,
                    float* state_out = nullptr, float* noisy = nullptr, float* dbn = nullptr, uint64_t seed = 0);
// noisy / dbn (either may be null): the element-wise epilogues of ac_encode_fused_ex, where fast_epilogue_supported()
bool fast_epilogue_supported(const ac_mdct_plan* p, const ac_psy_plan* psy, int iof, int C);
int launch_inv_fast(const ac_mdct_plan* p, const float* X, void* x, int iof, const float* tail_in, float* tail_out,
                    int B, int Kp, int nblk, int C, hipStream_t s);
int launch_psy_fast(const ac_psy_plan* p, const float* X, const float* t_in, float* t_out, float* thr, float drown,
                    int B, int F, int C, hipStream_t s, int iof = 0);
// streaming duplex: the analysis of a chunk of k_fwd blocks (psy: with the fused masking model) and the synthesis of a
// chunk of k_inv frames in one launch, where fast_duplex_serves() -- small float32 mono / stereo launches
bool fast_duplex_serves(const ac_mdct_plan* p, const ac_psy_plan* psy, int B, int C, int k_fwd, int k_inv);
int launch_duplex_fast(const ac_mdct_plan* p, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr,
                       float drown, const float* prev_block, float* state_out, int k_fwd, const float* X_inv, float* x_inv,
                       const float* tail_in, float* tail_out, int k_inv, int B, int C, hipStream_t s);

// wave-level masking model for general band layouts (ac_psy_mid.hip); mono / stereo float32
bool mid_psy_supported(const ac_psy_plan* p);
int mid_psy_plan_init(ac_psy_plan* p);
int launch_psy_mid(const ac_psy_plan* p, const float* X, const float* t_in, float* t_out, float* thr, float drown, int B,
                   int F, int C, hipStream_t s);
// the run-structured image of a model's tables, built on the host (no device needed): false when the tables lack the
// structure (ac_psy_runs_dev.h); runs_psy_plan_init uploads it
bool build_runs(const PsyTables& t, std::vector<uint32_t>* out, RunsLayout* lay);
int runs_psy_plan_init(ac_psy_plan* p);

int launch_tonality_bwd_generic(const ac_psy_plan* p, const float* X, const float* gt, float* gX, int accumulate, int B,
                                int F, int C, hipStream_t s);
int launch_threshold_bwd_generic(const ac_psy_plan* p, const float* X, const float* t, float drown, const float* gthr,
                                 float* gX, float* gt, int B, int F, int C, hipStream_t s);
// wave-level backward: tonality backward when g_thr is null, else threshold backward
int launch_psy_bwd_fast(const ac_psy_plan* p, const float* X, const float* t, float drown, const float* g_thr,
                        const float* g_t, float* g_X, float* g_t_out, int accumulate, int B, int F, int C,
                        hipStream_t s);
// compute_dtype variants (ac_generic.hip)
int launch_fwd_f64(const ac_mdct_plan* p, const double* x, double* X, int B, int Kin, int F, int C, hipStream_t s);
int launch_inv_f64(const ac_mdct_plan* p, const double* X, double* x, int B, int Kp, int nblk, int C, hipStream_t s);
// ... with the streaming state (block -1 of every signal / the aliased half of the frame before frame 0 in, of the last one out)
int launch_fwd_f64_stream(const ac_mdct_plan* p, const double* x, double* X, const double* prev_block, int B, int Kin, int F, int C,
                          hipStream_t s);
int launch_inv_f64_stream(const ac_mdct_plan* p, const double* X, double* x, const double* tail_in, double* tail_out, int B, int Kp,
                          int nblk, int C, hipStream_t s);
int launch_fwd_bf16(const ac_mdct_plan* p, const bf16_t* x, bf16_t* X, int B, int Kin, int F, int C, hipStream_t s);
int launch_inv_bf16(const ac_mdct_plan* p, const bf16_t* X, bf16_t* x, int B, int Kp, int nblk, int C, hipStream_t s);
int launch_fwd_f16(const ac_mdct_plan* p, const f16_t* x, f16_t* X, int B, int Kin, int F, int C, hipStream_t s);
int launch_inv_f16(const ac_mdct_plan* p, const f16_t* X, f16_t* x, int B, int Kp, int nblk, int C, hipStream_t s);
int launch_tonality_f64(const ac_psy_plan* p, const double* X, double* t, int B, int F, int C, hipStream_t s);
int launch_tonality_bf16(const ac_psy_plan* p, const bf16_t* X, bf16_t* t, int B, int F, int C, hipStream_t s);
int launch_threshold_f64(const ac_psy_plan* p, const double* X, const double* t, double drown, double* thr, int B, int F,
                         int C, hipStream_t s);
int launch_threshold_bf16(const ac_psy_plan* p, const bf16_t* X, const bf16_t* t, float drown, bf16_t* thr, int B, int F,
                          int C, hipStream_t s);
int launch_tonality_bwd_typed(const ac_psy_plan* p, const void* X, const void* gt, void* gX, int dtype, int B, int F, int C, hipStream_t s);
int launch_threshold_bwd_typed(const ac_psy_plan* p, const void* X, const void* t, double drown, const void* gthr, void* gX, void* gt,
                               int dtype, int B, int F, int C, hipStream_t s);
int launch_db_typed(const void* a, void* out, size_t n, int norm, int dtype, hipStream_t s);
int launch_add_noise_typed(const void* X, const void* thr, void* out, size_t n, uint64_t seed, int dtype, hipStream_t s);
int launch_db(const float* a, float* out, size_t n, int norm, hipStream_t s);
int launch_db_bwd(const float* a, const float* g, float* ga, size_t n, int norm, hipStream_t s);
int launch_add_noise(const float* X, const float* thr, float* out, size_t n, uint64_t seed, hipStream_t s);

}  // namespace ac
