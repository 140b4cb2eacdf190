// Wave-level kernels for filters_n = 1024 and 2048 on gfx950 (MI355X); the description below is for 1024
// (8 complex FFT points per lane), 2048 runs the same code with 16 (template parameter R).
//
// One 64-lane wavefront transforms one frame of two signals at a time (the two channels of a stereo clip, or two
// mono signals): both ride in the two halves of 64-bit register pairs (v2f = (s0, s1)), so twiddles, window
// coefficients, addresses and LDS traffic are shared.
//
// Data movement per frame, q = 64 i + lane being the 16-byte granule (x[2q], x[2q+1]) x (s0, s1) that lane `lane`
// loads / stores with one coalesced 16-byte access per i:
//   * analysis loads blocks n-1 and n of the PCM.  FFT element e = lane + 64 r needs, from each block, the even sample
//     of granule e + 256 (same lane, another register) and the odd sample of granule 767 - e (lane 63 - lane): only
//     the odd halves cross lanes, through one lane-reversal exchange in LDS (ds_write_b64 / ds_read_b64);
//   * window fold with two coefficients per element (Princen-Bradley windows: the 2x2 fold blocks are rotations);
//   * pre-twiddle -> 512-point complex FFT as three in-register radix-8 passes with two padded, conflict-free LDS
//     exchanges whose addresses are one per-lane base + an immediate -> post-twiddle; the output bin of
//     (lane, register k2) is lane + 64 k2, so the even coefficients X[2k] are already where the store wants
//     them and only the odd ones (X[N-1-2k]) take the lane-reversal exchange again;
//   * coalesced 16-byte stores of X; the psychoacoustic epilogue (tonality, Bark sums, spreading, threshold)
//     runs on the frame in registers; the band x band spreading product runs on the matrix cores by default
//     (spread_mfma: split-bf16 v_mfma_f32_4x4x4_16b_bf16) or as f32 multiply-adds (AC_SPREAD_F32);
//   * the same kernels take 16-bit PCM (IOF 1) or bfloat16 tensors (IOF 2): the conversion sits in the row loads / stores;
//   * synthesis carries the aliased half of a frame's DCT-IV in registers along a short strip of output blocks.
// Frames are dealt to waves in order, so the chip works on one contiguous window of every tensor (DESIGN_LOG.md section 9).
//
// Index maps and their bank behaviour are emulated lane by lane in tests/emulate_wave_fft.py.
// Reference formulas: mdctransformer.py:62-153 (closed forms in SURVEY.md App. A), psychoacoustic.py:102-210,301-331.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ac_internal.h"
#include "ac_psy_mid_dev.h"
#include "ac_psy_runs_dev.h"

namespace ac {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const float* gtab_t;   // LDS-resident table image

#ifndef AC_WAVES_PSY
#define AC_WAVES_PSY 4              // waves per workgroup, fused encode (LDS: three workgroups per CU)
#endif
#ifndef AC_WAVES
#define AC_WAVES 4                  // waves per workgroup, plain transform / inverse / stand-alone psycho
#endif
#ifndef AC_NT_X
#define AC_NT_X AC_NT_STORE         // the fused encode's two write streams separately (experiments: one temporal, one streaming)
#endif
#ifndef AC_NT_THR
#define AC_NT_THR AC_NT_STORE
#endif
#ifndef AC_WPE
#define AC_WPE 3                    // waves per SIMD the register allocator must leave room for
#endif
#ifndef AC_NT_STORE
#define AC_NT_STORE 1               // 1: streaming (non-temporal) stores of the output rows (measured +4 % on both kernels)
#endif
#ifndef AC_NT_LOAD
#define AC_NT_LOAD 0                // bit 0: block n, bit 1: block n-1 of the analysis, bit 2: frames of the synthesis
#endif
#ifndef AC_REV_DPP
#define AC_REV_DPP 0                // 1: lane reversals in registers (row_mirror DPP + v_permlane16/32_swap); 0: through LDS
#endif
constexpr int WAVE_LDS = 9216;      // bytes of LDS per wave: 576 x 16-byte elements (8 rows of 64 + 8 pad)
constexpr int S8_OFF = 8192;        // psycho: 128 chunk sums (8 bins each) behind the 8 KB intensity image
constexpr int ZERO_OFF = 9216;      // psycho: one zero slot (padding target of the gather lists)
constexpr int WAVE_LDS_PSY = 9232;
constexpr int MF_COPY_STRIDE = 288;               // bytes between the four shifted copies of the reversed bf16 prototype
constexpr int MF_TAB_BYTES = 4 * MF_COPY_STRIDE;  // one table (hi or lo parts)
constexpr float kEps = 1e-14f;      // _INTENSITY_EPS, psychoacoustic.py:56

// ---- geometry and mdct tables for R complex FFT points per lane: filters_n = 128 R (R = 8: 1024, R = 16: 2048).
// Two table images in ac_mdct_plan::d_fast (analysis at 0, synthesis at I_TOTAL floats); the kernel copies the first
// I_LDS floats of its image into LDS once per workgroup, so the frame loop touches HBM only for PCM / spectra.
template <int R>
struct Geo {
  static constexpr int FH = 64 * R;                 // complex FFT points per frame
  static constexpr int FN = 128 * R;                // filters_n
  static constexpr int I_P2 = 0;                    // [8][8]  float2  W64^(e0 k1)
  static constexpr int I_POST = I_P2 + 128;         // [R][64] float2  exp(-i pi k / N) * (1/(N sqrt 2) | 2 sqrt 2), k = lane + 64 j
  static constexpr int I_COEF = I_POST + 128 * R;   // [R][64] float2  fold (A, B)(e) | unfold (a, b)(k)
  static constexpr int I_PRE = I_COEF + 128 * R;    // [R][64] float2  exp(-i pi (e + 1/4) / N), e = lane + 64 r
  static constexpr int I_P1 = I_PRE + 128 * R;      // [R][64] float2  W_{64R}^(lane k0), pass-1 twiddles
  // the other two coefficients of a fold block that is not a rotation (float32-precomputed or rectangular windows; FOLD4
  // kernels read them from global memory: the image is L2-resident): analysis (cE, cO)(e) | synthesis (s3, s4)(k)
  static constexpr int I_COEF2 = I_P1 + 128 * R;    // [R][64] float2
  static constexpr int I_TOTAL = I_COEF2 + 128 * R; // floats per image in global memory
  // R = 8 holds the seven pass-1 twiddles of a lane in registers (LDS is the scarcer resource: 3 workgroups per CU);
  // R = 16 reads its fifteen from LDS (registers are: 64 for the frame alone)
  static constexpr bool P1_IN_REGS = (R == 8);
  static constexpr int I_LDS = P1_IN_REGS ? I_P1 : I_COEF2;   // floats that live in LDS (R = 8: 12 800 bytes)
  // a kernel short of LDS leaves the pre-twiddles out as well (I_LDS_NOPRE floats) and forms them as
  // PRE[e] = POST[e] * (exp(-i pi / (4 N)) / scale): four more packed multiply-adds per element
  static constexpr int I_LDS_NOPRE = I_PRE;
  static constexpr int TAB_LDS = I_LDS * 4;
};
// waves per SIMD the register allocator must leave room for: the strided any-channel-count variants (CMODE 1) and the
// 2048-filter kernels get the larger budget
template <int R, int CMODE, bool PSY = false, int SPREAD = 0>
constexpr int wpe() { return (R == 8 && (CMODE == 0 || (CMODE == 2 && !PSY))) ? AC_WPE : 2; }

// ---- psy image (32-bit words) in ac_psy_plan::d_fast for filter_bands_n = 128 R and 64 Bark bands; the first PL_LDS
// words are copied into LDS once per workgroup, the rest is held in registers.  The spectrum passes through the wave's
// 8 KB intensity image in NH = R / 8 halves of 1024 bins.
template <int R>
struct PsyGeo {
  static constexpr int NH = R / 8;
  static constexpr int PL_HALF = (R == 8) ? 12 : 16;          // gather-list length / 2 (per half of the spectrum)
  static constexpr int FN = 128 * R;
  static constexpr int PL_G = 0;                              // [128]         spreading prototype g
  static constexpr int PL_LST = 128;                          // [NH][PL_HALF][64]  gather lists: two 16-bit LDS byte offsets per word
  static constexpr int PL_LDS = PL_LST + NH * PL_HALF * 64;   // R = 8: 896 words = 3584 bytes
  static constexpr int PL_BAND = PL_LDS;                      // [NH + 1][64] x 4 words: per-lane (= per Bark band) constants
  //   group h < NH: edge offsets of half h (lo | hi << 16), wf, wl, quiet      group NH: beta, rho, u0, u1
  static constexpr int PL_IDX = PL_BAND + (NH + 1) * 256;     // [R / 4][64] x 4 words: byte offsets (lo | hi << 16) of the
  //   threshold entries of the two bins of granule 64 i + lane, word i
  static constexpr int P_TOTAL = PL_IDX + (R / 4) * 256;
  static constexpr int PSY_LDS = PL_LDS * 4;
  // bf16 tiles of the spreading matrix for the MFMA form of the band x band product (spread_mfma): hi table, lo table
  static constexpr int PL_MF = P_TOTAL;
  static constexpr int P_TOTAL_MF = PL_MF + 2 * (MF_TAB_BYTES / 4);
};
// bytes of LDS the MFMA tiles take behind the psy image (SPREAD 0: f32 VALU product, 1: bf16, 2: split bf16)
constexpr int mf_lds(int spread) { return spread * MF_TAB_BYTES; }

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v2b __attribute__((ext_vector_type(2)));

struct C2 {   // one complex value for both channels of the pair
  v2f re, im;
};

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in order; this only pins the compiler's ordering of
  // cross-lane communication through LDS
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ C2 cmul(const C2& x, const v2f w) {
  C2 r;
  r.re = x.re * w.x - x.im * w.y;
  r.im = x.re * w.y + x.im * w.x;
  return r;
}
// (re, -im) of x * w: the sign rides on the operand modifiers of the multiply-add instead of a separate negation
__device__ __forceinline__ C2 cmul_negim(const C2& x, const v2f w) {
  C2 r;
  r.re = x.re * w.x - x.im * w.y;
  r.im = (-x.re) * w.y - x.im * w.x;
  return r;
}
__device__ __forceinline__ C2 cadd(const C2& a, const C2& b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ C2 csub(const C2& a, const C2& b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ C2 mul_mi(const C2& a) { return {a.im, -a.re}; }   // a * (-i)

// 8-point DFT (forward sign) of the eight registers, outputs in natural order
__device__ __forceinline__ void dft8(C2 (&x)[8]) {
  constexpr float R = 0.70710678118654752440f;
  C2 a0 = cadd(x[0], x[4]), a4 = csub(x[0], x[4]);
  C2 a1 = cadd(x[1], x[5]), a5 = csub(x[1], x[5]);
  C2 a2 = cadd(x[2], x[6]), a6 = csub(x[2], x[6]);
  C2 a3 = cadd(x[3], x[7]), a7 = csub(x[3], x[7]);
  a5 = {(a5.re + a5.im) * R, (a5.im - a5.re) * R};     // * W8^1
  a6 = mul_mi(a6);                                      // * W8^2
  a7 = {(a7.im - a7.re) * R, -(a7.re + a7.im) * R};    // * W8^3
  {
    C2 c0 = cadd(a0, a2), c2 = csub(a0, a2), c1 = cadd(a1, a3), c3 = mul_mi(csub(a1, a3));
    x[0] = cadd(c0, c1);
    x[4] = csub(c0, c1);
    x[2] = cadd(c2, c3);
    x[6] = csub(c2, c3);
  }
  {
    C2 c0 = cadd(a4, a6), c2 = csub(a4, a6), c1 = cadd(a5, a7), c3 = mul_mi(csub(a5, a7));
    x[1] = cadd(c0, c1);
    x[5] = csub(c0, c1);
    x[3] = cadd(c2, c3);
    x[7] = csub(c2, c3);
  }
}

__device__ __forceinline__ void lds_put(char* p, const C2& v) {
  *reinterpret_cast<v4f*>(p) = v4f{v.re.x, v.re.y, v.im.x, v.im.y};
}
__device__ __forceinline__ C2 lds_get(const char* p) {
  const v4f t = *reinterpret_cast<const v4f*>(p);
  return {v2f{t.x, t.y}, v2f{t.z, t.w}};
}

// 16-point DFT (forward sign): two 8-point DFTs of the even / odd registers and one radix-2 stage
__device__ __forceinline__ void dft16(C2 (&x)[16]) {
  C2 e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    e[i] = x[2 * i];
    o[i] = x[2 * i + 1];
  }
  dft8(e);
  dft8(o);
  constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, R2 = 0.70710678118654752440f;
  const v2f w[8] = {v2f{1.f, 0.f}, v2f{c1, -s1}, v2f{R2, -R2}, v2f{s1, -c1},
                    v2f{0.f, -1.f}, v2f{-s1, -c1}, v2f{-R2, -R2}, v2f{-c1, -s1}};   // W16^k
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const C2 t = (k == 0) ? o[0] : (k == 4) ? mul_mi(o[4]) : cmul(o[k], w[k]);
    x[k] = cadd(e[k], t);
    x[k + 8] = csub(e[k], t);
  }
}
__device__ __forceinline__ void dft_regs(C2 (&x)[8]) { dft8(x); }
__device__ __forceinline__ void dft_regs(C2 (&x)[16]) { dft16(x); }

// 64R-point FFT of z[r] = element (lane + 64 r); result z[j] = bin lane + 64 j.
// Element index e = e0 + 8 e1 + 64 r (lane = e0 + 8 e1), bin k = k0 + R (k1 + 8 k2), k0 = 8 beta + kappa.
//   pass 1 over r -> k0 (radix R in registers), twiddle W_{64R}^(lane k0);
//   per batch beta of eight k0: exchange 1: row kappa (72 elements of 16 B: 64 + 8 pad), column lane;
//     lane (a = kappa, m0 = e0) reads e1 = 0..7 at a 72 + 8 e1 + m0;  pass 2 over e1 -> k1, twiddle W64^(e0 k1);
//   per half h of the k1 (k1 = (64/R) h + kk): exchange 2: element (k0, kk, e0) at 9 (k0 + R kk) + e0;
//     lane k0 + R kk reads its 8 consecutive e0;  pass 3 over e0 -> k2;  bin = lane + 64 (h + (R/8) k2).
// Every exchange address is one per-lane base + an immediate, and every access is bank-conflict-free under the
// gfx950 lane-group rules (tests/emulate_wave_fft.py emulates the maps for R = 8 and 16).
template <int R>
__device__ __forceinline__ void fft_wave(C2 (&z)[R], char* buf, gtab_t tab, const v2f (&p1)[R], int lane) {
  constexpr int NB = R / 8;      // batches of eight 64-point FFTs
  constexpr int Q = 64 / R;      // k1 values per exchange-2 half
  const int a = lane >> 3, m0 = lane & 7;
  dft_regs(z);
#pragma unroll
  for (int k = 1; k < R; ++k)
    z[k] = cmul(z[k], Geo<R>::P1_IN_REGS ? p1[k] : reinterpret_cast<const v2f*>(tab + Geo<R>::I_P1)[k * 64 + lane]);
#pragma unroll
  for (int beta = 0; beta < NB; ++beta) {
    C2 y[8];
    wave_sync();
    {
      char* w1 = buf + 16 * lane;
#pragma unroll
      for (int k = 0; k < 8; ++k) lds_put(w1 + 1152 * k, z[8 * beta + k]);
    }
    wave_sync();
    {
      const char* r1 = buf + 16 * (a * 72 + m0);
#pragma unroll
      for (int r = 0; r < 8; ++r) y[r] = lds_get(r1 + 128 * r);
    }
    dft8(y);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      z[8 * beta + k] = (k == 0) ? y[0] : cmul(y[k], reinterpret_cast<const v2f*>(tab + Geo<R>::I_P2)[k * 8 + m0]);
  }
  C2 out[R];
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    C2 y[8];
    wave_sync();
    {
      char* w2 = buf + 16 * (9 * a + m0);
#pragma unroll
      for (int beta = 0; beta < NB; ++beta)
#pragma unroll
        for (int kk = 0; kk < Q; ++kk) lds_put(w2 + 16 * (72 * beta + 9 * R * kk), z[8 * beta + Q * h + kk]);
    }
    wave_sync();
    {
      const char* r2 = buf + 144 * lane;
#pragma unroll
      for (int r = 0; r < 8; ++r) y[r] = lds_get(r2 + 16 * r);
    }
    dft8(y);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) out[h + NB * k2] = y[k2];
  }
#pragma unroll
  for (int j = 0; j < R; ++j) z[j] = out[j];
}

// the lane's pass-1 twiddles W_{64R}^(lane k), k = 1..R-1, from the image in global memory
template <int R>
__device__ __forceinline__ void load_p1(const float* __restrict__ image, int lane, v2f (&p1)[R]) {
  p1[0] = v2f{1.f, 0.f};
#pragma unroll
  for (int k = 1; k < R; ++k)
    p1[k] = Geo<R>::P1_IN_REGS ? reinterpret_cast<const v2f*>(image + Geo<R>::I_P1)[k * 64 + lane] : v2f{0.f, 0.f};
}

// 64-lane reversal of two registers at once, without LDS: lane index bits (b5 b4 | b3..b0).  v_permlane32_swap exchanges
// the upper half of its first operand with the lower half of its second, i.e. it transposes "which register" with b5;
// applied twice with the operands' roles swapped in between it complements b5 in both registers.  v_permlane16_swap does
// the same for b4 (odd rows of the first operand <-> even rows of the second).  row_mirror (DPP) complements b3..b0.
__device__ __forceinline__ void rev64_pair(float& a, float& b) {
  const unsigned ia = __float_as_uint(a), ib = __float_as_uint(b);
  const auto r = __builtin_amdgcn_permlane32_swap(ia, ib, false, false);
  const auto q = __builtin_amdgcn_permlane32_swap(r[1], r[0], false, false);
  const auto u = __builtin_amdgcn_permlane16_swap(q[0], q[1], false, false);
  const auto v = __builtin_amdgcn_permlane16_swap(u[1], u[0], false, false);
  a = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)v[0], 0x140, 0xf, 0xf, false));   // row_mirror
  b = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)v[1], 0x140, 0xf, 0xf, false));
}
__device__ __forceinline__ v2f rev64(v2f v) {
  float a = v.x, b = v.y;
  rev64_pair(a, b);
  return v2f{a, b};
}

// lane-reversal exchange of R (c0, c1) pairs: afterwards out[i] = in[(OFS - i) mod R] of lane 63 - lane
template <int OFS, int R>
__device__ __forceinline__ void rev_exchange(char* buf, int lane, const v2f (&in)[R], v2f (&out)[R]) {
#if AC_REV_DPP
#pragma unroll
  for (int i = 0; i < R; ++i) out[i] = rev64(in[(OFS - i) & (R - 1)]);
#else
  wave_sync();
  {
    char* w = buf + 8 * (63 - lane);
#pragma unroll
    for (int c = 0; c < R; ++c) *reinterpret_cast<v2f*>(w + 512 * c) = in[c];
  }
  wave_sync();
  {
    const char* r = buf + 8 * lane;
#pragma unroll
    for (int i = 0; i < R; ++i) out[i] = *reinterpret_cast<const v2f*>(r + 512 * ((OFS - i) & (R - 1)));
  }
#endif
}

// ---- the two signals a wave transforms side by side, and global <-> register movement of one natural-order row ----
// CMODE 0: exactly two channels: the pair is (clip b, channels 0 and 1), rows are interleaved 16-byte vectors.
// CMODE 1: any channel count: signals s = b C + c are paired in order, (2p, 2p+1), across clip boundaries when C is
//          odd (mono: two clips per wave), so no half of the packed registers idles except in one last odd pair;
//          rows are read with stride C from one base pointer per signal.
// CMODE 2: exactly one channel: as CMODE 1 with the two samples of a granule read / written as one 8-byte vector.
struct Pair {
  long long b0, b1;   // clips of the two signals
  int c0, c1;         // their channels
  bool has1;          // false: the second slot is the padding of an odd signal count
};
template <int CMODE>
__device__ __forceinline__ Pair make_pair(long long p, int C, long long nsig) {
  Pair q;
  if (CMODE == 0) {
    q.b0 = q.b1 = p;
    q.c0 = 0;
    q.c1 = 1;
    q.has1 = true;
  } else if (CMODE == 2) {   // one channel: signal = clip (no 64-bit divisions on the scalar unit)
    const long long s0 = 2 * p, s1 = s0 + 1;
    q.has1 = s1 < nsig;
    q.b0 = s0;
    q.b1 = q.has1 ? s1 : s0;
    q.c0 = q.c1 = 0;
  } else {
    const long long s0 = 2 * p, s1 = s0 + 1;
    q.has1 = s1 < nsig;
    q.b0 = s0 / C;
    q.c0 = (int)(s0 % C);
    q.b1 = q.has1 ? s1 / C : q.b0;
    q.c1 = q.has1 ? (int)(s1 % C) : q.c0;
  }
  return q;
}
// row of signal slot i of a [clips, rows_per_clip, N, C] tensor (floats per row over all channels = blk)
__device__ __forceinline__ size_t row_off(long long b, long long rows_per_clip, long long row, size_t blk, int c) {
  return ((size_t)b * (size_t)rows_per_clip + (size_t)row) * blk + (size_t)c;
}

template <int CMODE, bool NT = false, int R = 8>
__device__ __forceinline__ void load_row(const float* __restrict__ r0, const float* __restrict__ r1, int C, bool has1,
                                         int lane, v4f (&v)[R]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int q = 64 * i + lane;
      if (NT) v[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(r0) + q);
      else v[i] = reinterpret_cast<const v4f*>(r0)[q];
    }
  } else if (CMODE == 2) {
    // one wave-uniform branch for the second signal (only the last pair of an odd signal count lacks it)
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const v2f u = reinterpret_cast<const v2f*>(r0)[64 * i + lane];
      v[i] = v4f{u.x, 0.f, u.y, 0.f};
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const v2f w = reinterpret_cast<const v2f*>(r1)[64 * i + lane];
        v[i].y = w.x;
        v[i].w = w.y;
      }
    }
  } else {
    // uniform base per register + two 32-bit lane offsets shared by all registers (keeps the addresses out of VGPRs)
    const int off = 2 * lane * C;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const size_t step = (size_t)(128 * i) * C;
      v[i] = v4f{r0[step + off], 0.f, r0[step + off + C], 0.f};
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const size_t step = (size_t)(128 * i) * C;
        v[i].y = r1[step + off];
        v[i].w = r1[step + off + C];
      }
    }
  }
}

template <int CMODE, int R = 8, bool NTS = (AC_NT_STORE != 0)>
__device__ __forceinline__ void store_row(float* __restrict__ r0, float* __restrict__ r1, int C, bool has1, int lane,
                                          const v4f (&v)[R]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int q = 64 * i + lane;
      if (NTS) __builtin_nontemporal_store(v[i], reinterpret_cast<v4f*>(r0) + q);
      else reinterpret_cast<v4f*>(r0)[q] = v[i];
    }
  } else if (CMODE == 2) {
#pragma unroll
    for (int i = 0; i < R; ++i) reinterpret_cast<v2f*>(r0)[64 * i + lane] = v2f{v[i].x, v[i].z};
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) reinterpret_cast<v2f*>(r1)[64 * i + lane] = v2f{v[i].y, v[i].w};
    }
  } else {
    const int off = 2 * lane * C;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const size_t step = (size_t)(128 * i) * C;
      r0[step + off] = v[i].x;
      r0[step + off + C] = v[i].z;
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const size_t step = (size_t)(128 * i) * C;
        r1[step + off] = v[i].y;
        r1[step + off + C] = v[i].w;
      }
    }
  }
}

// ---- the same rows in 2-byte storage: 16-bit PCM (x = pcm / 32768; pcm = clamp(round(32768 x))) or bfloat16 ----------
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s2 __attribute__((ext_vector_type(2)));
constexpr float kPcmScale = 1.0f / 32768.0f;
__device__ __forceinline__ short to_pcm16(float v) {
  return (short)__float2int_rn(fminf(fmaxf(v * 32768.0f, -32768.0f), 32767.0f));
}
struct Pcm16Fmt {
  static __device__ __forceinline__ float dec(short h) { return (float)h * kPcmScale; }
  static __device__ __forceinline__ s2 enc2(float a, float b) { return s2{to_pcm16(a), to_pcm16(b)}; }
};
struct Bf16Fmt {   // storage = the upper half of the float32 pattern; stores round to nearest even (v_cvt_pk_bf16_f32)
  static __device__ __forceinline__ float dec(short h) { return __uint_as_float((uint32_t)(uint16_t)h << 16); }
  static __device__ __forceinline__ s2 enc2(float a, float b) {
    return __builtin_bit_cast(s2, __builtin_convertvector(v2f{a, b}, v2b));
  }
};

template <typename FMT, int CMODE, int R>
__device__ __forceinline__ void load_row_h(const int16_t* __restrict__ r0, const int16_t* __restrict__ r1, int C,
                                           bool has1, int lane, v4f (&v)[R]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const s4 p = reinterpret_cast<const s4*>(r0)[64 * i + lane];
      v[i] = v4f{FMT::dec(p.x), FMT::dec(p.y), FMT::dec(p.z), FMT::dec(p.w)};
    }
  } else if (CMODE == 2) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const s2 u = reinterpret_cast<const s2*>(r0)[64 * i + lane];
      v[i] = v4f{FMT::dec(u.x), 0.f, FMT::dec(u.y), 0.f};
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const s2 w = reinterpret_cast<const s2*>(r1)[64 * i + lane];
        v[i].y = FMT::dec(w.x);
        v[i].w = FMT::dec(w.y);
      }
    }
  } else {
    const int off = 2 * lane * C;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const size_t step = (size_t)(128 * i) * C;
      v[i] = v4f{FMT::dec(r0[step + off]), 0.f, FMT::dec(r0[step + off + C]), 0.f};
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const size_t step = (size_t)(128 * i) * C;
        v[i].y = FMT::dec(r1[step + off]);
        v[i].w = FMT::dec(r1[step + off + C]);
      }
    }
  }
}

template <typename FMT, int CMODE, int R>
__device__ __forceinline__ void store_row_h(int16_t* __restrict__ r0, int16_t* __restrict__ r1, int C, bool has1,
                                            int lane, const v4f (&v)[R]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const s2 lo = FMT::enc2(v[i].x, v[i].y), hi = FMT::enc2(v[i].z, v[i].w);
      reinterpret_cast<s4*>(r0)[64 * i + lane] = s4{lo.x, lo.y, hi.x, hi.y};
    }
  } else if (CMODE == 2) {
#pragma unroll
    for (int i = 0; i < R; ++i) reinterpret_cast<s2*>(r0)[64 * i + lane] = FMT::enc2(v[i].x, v[i].z);
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) reinterpret_cast<s2*>(r1)[64 * i + lane] = FMT::enc2(v[i].y, v[i].w);
    }
  } else {
    const int off = 2 * lane * C;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const size_t step = (size_t)(128 * i) * C;
      const s2 e = FMT::enc2(v[i].x, v[i].z);
      r0[step + off] = e.x;
      r0[step + off + C] = e.y;
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const size_t step = (size_t)(128 * i) * C;
        const s2 e = FMT::enc2(v[i].y, v[i].w);
        r1[step + off] = e.x;
        r1[step + off + C] = e.y;
      }
    }
  }
}
// IOF: 0 = float32 tensors; 1 = 16-bit PCM on the PCM side (spectra float32); 2 = bfloat16 tensors throughout
template <int IOF> struct RowFmt { using type = Pcm16Fmt; };
template <> struct RowFmt<2> { using type = Bf16Fmt; };

// wave-wide sum, result uniform (scalar register): xor butterflies inside each row of 16 lanes, then the two
// row broadcasts of the DPP unit; no LDS traffic
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {   // v + v[lane permuted by a DPP pattern] on the enabled rows
  const int iv = __builtin_bit_cast(int, v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]  (lane ^ 1)
  v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]  (lane ^ 2)
  v = dpp_add<0x141, 0xf>(v);   // row_half_mirror      (other quad of the 8)
  v = dpp_add<0x140, 0xf>(v);   // row_mirror           (other half of the 16): every lane holds its row's sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast15 into rows 1, 3: row 1 = r0 + r1, row 3 = r2 + r3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast31 into rows 2, 3: row 3 = total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }     // v_log_f32
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }    // v_exp_f32
__device__ __forceinline__ v2f log2v(v2f x) { return v2f{fast_log2(x.x), fast_log2(x.y)}; }
__device__ __forceinline__ v2f exp2v(v2f x) { return v2f{fast_exp2(x.x), fast_exp2(x.y)}; }
__device__ __forceinline__ v2f maxv(v2f a, float b) { return v2f{fmaxf(a.x, b), fmaxf(a.y, b)}; }

// ------------------------------------------------------------------------------------------------------
// psychoacoustic epilogue on one frame (both channels) held in natural order: xq[i] = (X[2q], X[2q+1]) x (c0, c1),
// q = 64 i + lane.  tonality: psychoacoustic.py:102-120; threshold: :122-148 with :169-210 (factorised,
// SURVEY App. A.3) and :301-331 (Bark mapping as per-band ranges / per-bin entry lookups).
// ------------------------------------------------------------------------------------------------------
struct PsyParams {
  const uint32_t* tab;   // ac_psy_plan::d_fast
  float alpha, inv_alpha, drown;
};

// per-lane (= per Bark band) constants and the lane's threshold-entry offsets, held in registers
template <int R>
struct PsyLane {
  v4f bc0[R / 8];   // per half: edge offsets (lo | hi << 16), wf, wl, quiet
  v4f bc1;          // beta, rho, u0, u1
  v4f idx[R / 4];   // byte offsets (lo | hi << 16) of the entries of the two bins of granule 64 i + lane, word i
};
// wave_base = byte offset of the wave's buffer inside the workgroup's LDS object: the packed 16-bit offsets become
// absolute, so unpacking one costs a single and / shift inside the loop
template <int R>
__device__ __forceinline__ PsyLane<R> load_psy_lane(const uint32_t* __restrict__ tab, int lane, uint32_t wave_base) {
  using P = PsyGeo<R>;
  PsyLane<R> c;
  const uint32_t both = wave_base * 0x10001u;
  auto rebase = [both](float f) { return __uint_as_float(__float_as_uint(f) + both); };
#pragma unroll
  for (int h = 0; h < P::NH; ++h) {
    c.bc0[h] = reinterpret_cast<const v4f*>(tab + P::PL_BAND)[h * 64 + lane];
    c.bc0[h].x = rebase(c.bc0[h].x);
  }
  c.bc1 = reinterpret_cast<const v4f*>(tab + P::PL_BAND)[P::NH * 64 + lane];
#pragma unroll
  for (int i = 0; i < R / 4; ++i) {
    const v4f w = reinterpret_cast<const v4f*>(tab + P::PL_IDX)[i * 64 + lane];
    c.idx[i] = v4f{rebase(w.x), rebase(w.y), rebase(w.z), rebase(w.w)};
  }
  return c;
}

// keeps the unpacking of a loop-invariant word inside the loop (hoisted, the 16 addresses would cost 16 registers)
__device__ __forceinline__ uint32_t in_loop(uint32_t w) {
  asm volatile("" : "+v"(w));
  return w;
}

// ------------------------------------------------------------------------------------------------------
// Band x band product with the Toeplitz spreading matrix on the matrix cores (BASELINE configs[3]):
//   out_j = sum_i Q_i S[i, j],  S[i, j] = g[64 - i + j]   (psychoacoustic.py:205-207)
// as 16 (MODE 1) or 32 (MODE 2) v_mfma_f32_4x4x4_16b_bf16.  The instruction's 16 blocks are the 16 column tiles of S
// (block b = bands 4 b .. 4 b + 3, so D lands with band j in lane j, the layout the epilogue continues in); step s
// contracts bands 4 s .. 4 s + 3.  The A tile of step s (4 rows x 4 bands) is the same for every block: it sits in the
// four lanes of block s and cbsz = 4 / abid = s broadcast it.  Rows 0, 1 = the two signals of the pair; MODE 1 leaves
// rows 2, 3 zero (plain bf16, ~3 significant digits: the tolerance is stated in the tests); MODE 2 splits Q = hi + lo
// (rows 2, 3 carry the lo parts) and S = hi + lo (a second B table), four partial products in f32 accumulators, ~16
// mantissa bits -- inside the 1e-4 parity bar.  B tile of step s in lane l: g[64 - 4 s - k + l], k = 0..3 = four
// consecutive entries of the reversed prototype; four copies of the table, shifted by one entry each, make the read an
// aligned 8-byte read for every lane (copy l & 3).  The compiler pairs the reads of two steps into ds_read2_b64, which
// the LDS serves in groups of 16 consecutive lanes over 32 banks: the copies sit 288 bytes = 8 banks (mod 32) apart, so
// the four copies a group touches (8 dwords each) fall on disjoint banks.
// mf = LDS copy of the hi table (MF_TAB_BYTES) followed by the lo table.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{a, b}, v2b));
}
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {   // lane K of every quad to the whole quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);
}
template <int S, int MODE>
struct MfSteps {
  static __device__ __forceinline__ void run(const v4s a, const char* bhi, v4f& d0, v4f& d1) {
    MfSteps<S - 1, MODE>::run(a, bhi, d0, d1);
    const v4s b = *reinterpret_cast<const v4s*>(bhi + 8 * S);
    if (MODE == 2) {
      d0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d0, 4, S, 0);
      const v4s bl = *reinterpret_cast<const v4s*>(bhi + MF_TAB_BYTES + 8 * S);
      d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, bl, d1, 4, S, 0);
    } else if (S & 1) {
      d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d1, 4, S, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d0, 4, S, 0);
    }
  }
};
template <int MODE>
struct MfSteps<-1, MODE> {
  static __device__ __forceinline__ void run(const v4s, const char*, v4f&, v4f&) {}
};
template <int MODE>
__device__ __forceinline__ v2f spread_mfma(v2f Q, const char* mf, int lane) {
  const uint32_t whi = pk_bf16(Q.x, Q.y);
  // quad-local 4 x 4 transpose of 16-bit values: lane 4 s + i gets row i of bands 4 s .. 4 s + 3.  Every lane
  // evaluates all broadcasts before the select (a DPP read needs its source lane active)
  const int i = lane & 3;
  const bool lo_row = i >= 2;
  uint32_t c0 = quad_bcast<0>(whi), c1 = quad_bcast<1>(whi), c2 = quad_bcast<2>(whi), c3 = quad_bcast<3>(whi);
  if (MODE == 2) {
    const float hx = __uint_as_float(whi << 16), hy = __uint_as_float(whi & 0xffff0000u);
    const uint32_t wlo = pk_bf16(Q.x - hx, Q.y - hy);
    const uint32_t l0 = quad_bcast<0>(wlo), l1 = quad_bcast<1>(wlo), l2 = quad_bcast<2>(wlo), l3 = quad_bcast<3>(wlo);
    c0 = lo_row ? l0 : c0, c1 = lo_row ? l1 : c1, c2 = lo_row ? l2 : c2, c3 = lo_row ? l3 : c3;
  } else {
    c0 = lo_row ? 0u : c0, c1 = lo_row ? 0u : c1, c2 = lo_row ? 0u : c2, c3 = lo_row ? 0u : c3;
  }
  const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;   // the signal's half of each word
  const uint2 au = {__builtin_amdgcn_perm(c1, c0, sel), __builtin_amdgcn_perm(c3, c2, sel)};
  const v4s a = __builtin_bit_cast(v4s, au);
  // (kept inside the frame loop: hoisted, the 16 / 32 tiles would pin 32 / 64 registers)
  const uint32_t boff = in_loop((uint32_t)((lane & 3) * MF_COPY_STRIDE + 8 * (16 - (lane >> 2))));
  v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
  MfSteps<15, MODE>::run(a, mf + boff, d0, d1);
  const v4f d = d0 + d1;
  return MODE == 2 ? v2f{d.x + d.z, d.y + d.w} : v2f{d.x, d.y};
}

// buf = the wave's LDS region (WAVE_LDS_PSY bytes), pimg = the workgroup's copy of the psy image
// lds0 = base of the workgroup's LDS object (the absolute offsets of PsyLane count from it)
struct NoEmit {
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void operator()(int, const v4f&) {}
};
// EMIT: begin() once the masking model has its per-entry values, then (i, threshold of granule 64 i + lane) as each granule
// of the threshold row comes out of the entry lookup (the kernels with element-wise epilogues consume it there instead of
// holding the whole row)
template <int R, bool WANT_T, bool WANT_THR, int SPREAD = 0, bool T_BF16 = false, class EMIT = NoEmit>
__device__ __forceinline__ void psy_stage(const v4f (&xq)[R], char* lds0, char* buf, const uint32_t* pimg,
                                          const PsyLane<R>& pc, const PsyParams& pp, int lane, v2f& t, v4f (&thr)[R],
                                          EMIT emit = EMIT()) {
  using P = PsyGeo<R>;
  if (WANT_T) {
    v2f slog = {0.f, 0.f}, ssq = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; ++i) {
      v4f I = xq[i] * xq[i];
      // the squares stay rounded products: left to -ffp-contract=fast, the compiler fuses one of the two squares of
      // ie + io into the sum -- which one differs between instantiations of this code (fused encode with / without
      // element-wise epilogues, 16-bit PCM input, stand-alone tonality), and with it the last bit of the tonality
      asm("" : "+v"(I));
      const v2f ie = v2f{I.x, I.y}, io = v2f{I.z, I.w};
      ssq += ie + io;
      // ln max(eps, a) + ln max(eps, b) = ln(max(eps, a) max(eps, b)): one v_log per two bins; the product stays
      // in the normal float range for |X| < 1e9 (>= 1e-28)
      slog += log2v(maxv(ie, kEps) * maxv(io, kEps));
    }
    slog.x = wave_sum(slog.x);
    slog.y = wave_sum(slog.y);
    ssq.x = wave_sum(ssq.x);
    ssq.y = wave_sum(ssq.y);
    const v2f am = ssq * (1.0f / P::FN) + kEps;
    // sfm = 10 log10(gm / am) with gm = exp(mean ln I)  ==  10 log10(2) (mean log2 I - log2 am)
    const v2f sfm = 3.0102999566398120f * (slog * (1.0f / P::FN) - log2v(am));
    const v2f tt = sfm * (-1.0f / 60.0f);
    // a frame with a NaN or an infinite intensity has a NaN tonality, as tf.maximum / reduce_mean / tf.minimum make it
    // (psychoacoustic.py:113-118): its sum of squares is not finite (v_max / v_min alone would return the other operand)
    t = v2f{(ssq.x - ssq.x == 0.0f) ? fminf(tt.x, 1.0f) : __builtin_nanf(""), (ssq.y - ssq.y == 0.0f) ? fminf(tt.y, 1.0f) : __builtin_nanf("")};
    if (T_BF16) {   // bfloat16 tensors: the threshold is computed from the tonality the caller gets
      const s2 e = Bf16Fmt::enc2(t.x, t.y);
      t = v2f{Bf16Fmt::dec(e.x), Bf16Fmt::dec(e.y)};
    }
  }
  if (!WANT_THR) return;

  // P_j = sum_f I_f W[f, j]  (:312-313): lane = Bark band; the two edge bins carry weights wf / wl, the interior
  // (weight 1) is gathered as single bins + 8-bin chunk sums through a host-built list of LDS offsets.  The spectrum
  // goes through the 8 KB image in halves of 1024 bins; an edge or list entry outside the half points at the zero slot.
  v2f P0 = {0.f, 0.f}, P1 = {0.f, 0.f};
#pragma unroll
  for (int h = 0; h < P::NH; ++h) {
    wave_sync();
    {
      // intensities in natural order: granule q = (I[2q], I[2q+1]) x (c0, c1) at byte 16 (q ^ ((q >> 4) & 3))
      const int lsw = lane ^ ((lane >> 4) & 3);
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<v4f*>(buf + 16 * lsw + 1024 * i) = xq[8 * h + i] * xq[8 * h + i];
    }
    wave_sync();
    // sums over aligned chunks of 8 bins (4 granules): lane c owns chunks c and c + 64
    {
      const int x = (lane >> 2) & 3;
      const char* cb = buf + 64 * lane;
      const int o0 = 16 * x, o1 = 16 * (1 ^ x), o2 = 16 * (2 ^ x), o3 = 16 * (3 ^ x);
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const char* c2 = cb + 4096 * i2;
        const v4f g0 = *reinterpret_cast<const v4f*>(c2 + o0), g1 = *reinterpret_cast<const v4f*>(c2 + o1),
                  g2 = *reinterpret_cast<const v4f*>(c2 + o2), g3 = *reinterpret_cast<const v4f*>(c2 + o3);
        const v4f s = (g0 + g1) + (g2 + g3);
        *reinterpret_cast<v2f*>(buf + S8_OFF + 8 * lane + 512 * i2) = v2f{s.x + s.z, s.y + s.w};
      }
    }
    wave_sync();
    const v4f bc0 = pc.bc0[h];
    const uint32_t edge = in_loop(__float_as_uint(bc0.x));
    P0 += *reinterpret_cast<const v2f*>(lds0 + (edge & 0xffffu)) * bc0.y;
    P1 += *reinterpret_cast<const v2f*>(lds0 + (edge >> 16)) * bc0.z;
#pragma unroll
    for (int hlf = 0; hlf < P::PL_HALF; ++hlf) {
      const uint32_t w = pimg[P::PL_LST + (h * P::PL_HALF + hlf) * 64 + lane];
      P0 += *reinterpret_cast<const v2f*>(buf + (w & 0xffffu));
      P1 += *reinterpret_cast<const v2f*>(buf + (w >> 16));
    }
  }
  const v2f Pj = P0 + P1;
  v2f Q = exp2v(pp.alpha * log2v(maxv(Pj, kEps)));   // max(eps, P)^alpha  (:206)
  // a band with a NaN intensity stays NaN (tf.maximum): through the band x band product it poisons every band of its frame and
  // signal, as the reference's dense einsum does (:205-207)
  Q = v2f{Pj.x == Pj.x ? Q.x : Pj.x, Pj.y == Pj.y ? Q.y : Pj.y};
  v2f acc;
  if (SPREAD == 0) {
    wave_sync();
    *reinterpret_cast<v2f*>(buf + 8 * lane) = Q;
    wave_sync();
    // sum_i Q_i S[i, j], S[i, j] = g[64 - i + j]  (:205-207 with the offset factor pulled out of the sum)
    v2f acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
    const float* gp = reinterpret_cast<const float*>(pimg + P::PL_G) + 64 + lane;
#pragma unroll 8
    for (int i = 0; i < 64; i += 2) {
      const v4f qq = *reinterpret_cast<const v4f*>(buf + 8 * i);   // Q_i, Q_{i+1} (broadcast read)
      acc0 += v2f{qq.x, qq.y} * gp[-i];
      acc1 += v2f{qq.z, qq.w} * gp[-i - 1];
    }
    acc = acc0 + acc1;
  } else {
    acc = spread_mfma<SPREAD>(Q, reinterpret_cast<const char*>(pimg) + P::PSY_LDS, lane);
  }
  const v4f bc1 = pc.bc1;
  const v2f offset = (1.0f - pp.drown) * (t * bc1.x + 9.0f * t + 5.5f);                        // (:185-191)
  const v2f fac = exp2v(offset * (-pp.alpha * 0.33219280948873623f));                          // 10^(-alpha O / 10)
  const v2f T = exp2v(pp.inv_alpha * log2v(maxv(fac * acc, kEps)));                             // (:208)
  v2f G = maxv(T, pc.bc0[0].w);                                                                 // (:144)
  {   // NaN where the reference has NaN: a poisoned product or a NaN tonality (the clamps -- v_max -- would drop it)
    const float px = acc.x + t.x, py = acc.y + t.y;
    G = v2f{px == px ? G.x : px, py == py ? G.y : py};
  }
  v2f Gn;
  Gn.x = __shfl_down(G.x, 1, 64);   // G of band j + 1
  Gn.y = __shfl_down(G.y, 1, 64);
  // thr of the bins of band j: interior bins sqrt(max(eps, G_j rho_j)); the bin shared with band j+1
  // sqrt(max(eps, G_j u0 + G_{j+1} u1))  (:330-331) -- one value per entry, not per bin
  const v2f A0 = maxv(G * bc1.y, kEps), A1 = maxv(G * bc1.z + Gn * bc1.w, kEps);
  wave_sync();
  // v_sqrt_f32 (1 ulp); arguments are >= 1e-14, far from the denormal range
  // (a poisoned frame: every band's G is NaN, and so is every entry)
  *reinterpret_cast<v4f*>(buf + 16 * lane) = v4f{G.x == G.x ? __builtin_amdgcn_sqrtf(A0.x) : G.x, G.y == G.y ? __builtin_amdgcn_sqrtf(A0.y) : G.y,
                                                 G.x == G.x ? __builtin_amdgcn_sqrtf(A1.x) : G.x, G.y == G.y ? __builtin_amdgcn_sqrtf(A1.y) : G.y};   // entry e at byte 8 e
  wave_sync();
  emit.begin();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const v4f ww = pc.idx[i >> 2];
    const float wf = (i & 3) == 0 ? ww.x : (i & 3) == 1 ? ww.y : (i & 3) == 2 ? ww.z : ww.w;
    const uint32_t w = in_loop(__float_as_uint(wf));
    const v2f a0 = *reinterpret_cast<const v2f*>(lds0 + (w & 0xffffu));
    const v2f a1 = *reinterpret_cast<const v2f*>(lds0 + (w >> 16));
    if constexpr (std::is_same<EMIT, NoEmit>::value) thr[i] = v4f{a0.x, a0.y, a1.x, a1.y};
    else emit(i, v4f{a0.x, a0.y, a1.x, a1.y});
  }
}

// copies the table image (and the psy image) into the workgroup's LDS behind the wave buffers; every thread takes part
template <int NW, int WSTRIDE, int TABF, int PSYW, int PSY_TOTAL = 0, int MFB = 0>
__device__ __forceinline__ void load_tables(char* lds, const float* __restrict__ image, const uint32_t* psy_tab) {
  if (image) {
    v4f* dst = reinterpret_cast<v4f*>(lds + NW * WSTRIDE);
    const v4f* src = reinterpret_cast<const v4f*>(image);
    for (int i = threadIdx.x; i < TABF / 4; i += NW * 64) dst[i] = src[i];
  }
  if (psy_tab) {
    uint4* pd = reinterpret_cast<uint4*>(lds + NW * WSTRIDE + (image ? TABF * 4 : 0));
    const uint4* ps = reinterpret_cast<const uint4*>(psy_tab);
    for (int i = threadIdx.x; i < PSYW / 4; i += NW * 64) pd[i] = ps[i];
    if (MFB > 0) {   // the bf16 tiles of spread_mfma, behind the psy image
      const uint4* ms = reinterpret_cast<const uint4*>(psy_tab + PSY_TOTAL);
      for (int i = threadIdx.x; i < MFB / 16; i += NW * 64) pd[PSYW / 4 + i] = ms[i];
    }
  }
  __syncthreads();
}

// EMIT of the fused encode with AC_EMIT_NOISY (stereo float32 rows)
template <int R>
struct NoisyEmit {
  const v4f* X_row;   // this lane's granules of the spectrum row the wave has just stored
  v4f* thr_row;
  v4f* noisy_row;
  uint64_t i4base, key;
  v4f xr[R];
  __device__ __forceinline__ void begin() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's own stores of X have landed
#pragma unroll
    for (int i = 0; i < R; ++i) xr[i] = X_row[64 * i];
  }
  __device__ __forceinline__ void operator()(int i, const v4f& th_i) {
    __builtin_nontemporal_store(th_i, thr_row + 64 * i);
    float g0, g1, g2, g3;
    const uint64_t i4 = i4base + 64u * i;
    normal_pair(key, 2 * i4, g0, g1);
    normal_pair(key, 2 * i4 + 1, g2, g3);
    __builtin_nontemporal_store(v4f{noisy_of(xr[i].x, th_i.x, g0), noisy_of(xr[i].y, th_i.y, g1), noisy_of(xr[i].z, th_i.z, g2),
                                    noisy_of(xr[i].w, th_i.w, g3)}, noisy_row + 64 * i);
    __builtin_amdgcn_sched_barrier(0);   // one granule at a time: interleaved, the sixteen draws of a row spill registers
  }
};

// ------------------------------------------------------------------------------------------------------
// analysis (+ fused epilogue)
// ------------------------------------------------------------------------------------------------------
struct FwdArgs {
  const void* x;             // [B, Kin*N, C] float32, or int16 PCM for the PCM16 kernels
  float* X;                  // [B, F, N, C]
  float* t;                  // [B, F, 1, C]   (PSY)
  float* thr;                // [B, F, N, C]   (PSY)
  const float* prev_block;   // [B, N, C] or null: block -1 of every signal (streaming analysis state)
  float* state_out;          // [B, N, C] or null: receives block Kin-1 of every signal (the next chunk's prev_block;
                             // a different buffer than prev_block: other waves still read that one)
  const float* tab;          // mdct tables (analysis image)
  PsyParams psy;
  // optional element-wise epilogues of the fused encode (EPI kernels): X + thr * Normal(0, 1/6) and amplitude_to_dB_norm(X),
  // both [B, F, N, C]; either may be null
  float* noisy;
  float* dbn;
  uint64_t noise_key;        // mix64(seed) of ac_add_noise
  int B, Kin, F, C;
  long long npairs, nsig;    // wave tasks per frame index (see Pair) and B * C
  int xcd;                   // 1: consecutive logical workgroups share an XCD (gridDim.x is a multiple of 8)
  int T;                     // > 0: workgroup g owns frames [g NW T, (g+1) NW T), wave w takes g NW T + w + NW t;
                             // 0: persistent waves, wave w of W takes frames w, w + W, ...
  long long nframes;         // npairs * F
  float pre_re, pre_im;      // PRE[e] / POST[e] of the analysis image (kernels that keep no pre-twiddles in LDS)
};

// Analysis is frame-independent: frame n of a channel pair needs blocks n-1 and n of the PCM, and a wave that loads
// both needs nothing from its neighbours.  Frames are dealt out in order -- workgroup g owns frames [g NW T, (g+1) NW T)
// and its wave w takes g NW T + w + NW t (or, with T = 0, persistent waves take w, w + W, ...) -- so at any moment the
// chip reads one contiguous window of the PCM and writes one contiguous window of each output tensor, which is what HBM
// rewards (tools/ubench_strips.hip: 10-15 % over per-wave strips); block n-1 is the block the neighbouring wave loads as
// its block n, so the second read is an L2 hit.
//
// Element e = lane + 64 r of the FFT input takes, from each block, the even sample of granule e + 256 (this lane,
// register (r + 4) & 7) and the odd sample of granule 767 - e (lane 63 - lane, register (3 - r) & 7).
// With (A, B) = COEF[e]:  carried part (block n-1) = B xe + A xo;  current part (block n) = B xo - A xe (r < 4),
// A xe - B xo (r >= 4)   (Princen-Bradley windows make the 2x2 fold blocks rotations).
// one LDS object: [NW wave buffers | table image | psy image | bf16 tiles of the spreading matrix (SPREAD > 0)]
// the matrix-core spreading kernels at 8 points per lane pay for their bf16 tiles with the pre-twiddle table, which
// they rebuild from the post-twiddles: the workgroup stays under 53 760 B, three to a CU
template <int R, bool PSY, int SPREAD>
constexpr bool fwd_nopre() { return (R == 8) && PSY && SPREAD > 0; }
template <int R, bool PSY, int NW, int SPREAD>
constexpr int fwd_lds_bytes() {
  return NW * (PSY ? WAVE_LDS_PSY : WAVE_LDS) + (fwd_nopre<R, PSY, SPREAD>() ? Geo<R>::I_LDS_NOPRE : Geo<R>::I_LDS) * 4 +
         (PSY ? PsyGeo<R>::PSY_LDS + mf_lds(SPREAD) : 0);
}
// the kernel's body: workgroup `bid` of `nblocks` (the kernel below passes blockIdx.x / gridDim.x; the streaming duplex
// kernel runs it on the first part of its grid), lds = fwd_lds_bytes() bytes of LDS, 16-byte aligned
template <int R, int CMODE, bool PSY, int NW, int IOF = 0, int SPREAD = 0, bool EPI = false>
__device__ __forceinline__ void fwd_fast_body(const FwdArgs& a, char* lds, const int bid, const int nblocks) {
  static_assert(!EPI || (PSY && CMODE == 0 && IOF == 0), "the element-wise epilogues ride on the stereo float32 fused encode");
  using G = Geo<R>;
  constexpr int WSTRIDE = PSY ? WAVE_LDS_PSY : WAVE_LDS;
  constexpr bool NOPRE = fwd_nopre<R, PSY, SPREAD>();
  constexpr int TABF = NOPRE ? G::I_LDS_NOPRE : G::I_LDS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  load_tables<NW, WSTRIDE, TABF, PsyGeo<R>::PL_LDS, PsyGeo<R>::PL_MF, mf_lds(SPREAD)>(lds, a.tab, PSY ? a.psy.tab : nullptr);
  char* buf = lds + wave * WSTRIDE;
  gtab_t tab = reinterpret_cast<const float*>(lds + NW * WSTRIDE);
  const uint32_t* pimg = reinterpret_cast<const uint32_t*>(lds + NW * WSTRIDE + TABF * 4);
  if (PSY) *reinterpret_cast<v2f*>(buf + ZERO_OFF) = v2f{0.f, 0.f};   // the gather lists' padding slot
  // (the kernels with element-wise epilogues are short of registers at the end of the masking model: they fetch the
  // lane's pass-1 twiddles per frame, beside the frame's PCM, instead of holding them across the loop)
  v2f p1[R];
  if constexpr (!EPI) load_p1<R>(a.tab, lane, p1);
  PsyLane<R> pc;
  if (PSY) pc = load_psy_lane<R>(a.psy.tab, lane, (uint32_t)(wave * WSTRIDE));
  int g = bid;
  if (a.xcd) g = (g & 7) * (nblocks >> 3) + (g >> 3);
  const int C = a.C;
  const size_t blk = (size_t)G::FN * C;   // floats per block / frame row over all channels
  // frame f = (pair, n); everything about it is wave-uniform and lives in scalar registers, advanced without divisions
  const long long stride = a.T > 0 ? (long long)NW : (long long)nblocks * NW;
  const long long f0 = (a.T > 0 ? (long long)g * NW * a.T : (long long)g * NW) + wave;
  int left = a.T > 0 ? a.T : 0x7fffffff;
  const long long dpair = stride / a.F;
  const int dn = (int)(stride % a.F);
  long long pair = f0 / a.F;
  int n = (int)(f0 % a.F);
  const long long npairs = a.npairs;
  using pcm_t = typename std::conditional<IOF != 0, int16_t, float>::type;
  const pcm_t* __restrict__ xin = static_cast<const pcm_t*>(a.x);
  const pcm_t* __restrict__ xstate = IOF != 0 ? nullptr : reinterpret_cast<const pcm_t*>(a.prev_block);
  // bfloat16 tensors (IOF 2): the streaming state stays float32 (a bfloat16 block is exact in it)
  const float* __restrict__ fstate = IOF == 2 ? a.prev_block : nullptr;

  // loads block fn (WHICH 0) or block fn-1 (WHICH 1) of frame (pr, fn) in natural order.  A missing block (before the
  // first / after the last) is loaded from a neighbouring, valid address and zeroed when it is consumed (returns false),
  // so that the loads stay unconditional and nothing waits for them at the point of issue.
  auto issue_loads = [&](auto which, long long pr, int fn, v4f (&dst)[R]) -> bool {
    const Pair q = make_pair<CMODE>(pr, C, a.nsig);
    const pcm_t *s0, *s1;
    bool ok;
    if (decltype(which)::value == 0) {
      ok = fn < a.Kin;
      const int blkidx = ok ? fn : (a.Kin > 0 ? a.Kin - 1 : 0);
      s0 = xin + row_off(q.b0, a.Kin, blkidx, blk, q.c0);
      s1 = xin + row_off(q.b1, a.Kin, blkidx, blk, q.c1);
    } else {
      if constexpr (IOF == 2) {
        if (fn < 1 && fstate) {   // block -1 = the stored float32 state
          load_row<CMODE, false, R>(fstate + row_off(q.b0, 1, 0, blk, q.c0), fstate + row_off(q.b1, 1, 0, blk, q.c1), C, q.has1, lane, dst);
          return true;
        }
      }
      ok = (fn >= 1) || xstate;
      if (fn >= 1 || !xstate) {
        const int blkidx = fn >= 1 ? fn - 1 : 0;
        s0 = xin + row_off(q.b0, a.Kin, blkidx, blk, q.c0);
        s1 = xin + row_off(q.b1, a.Kin, blkidx, blk, q.c1);
      } else {
        s0 = xstate + row_off(q.b0, 1, 0, blk, q.c0);
        s1 = xstate + row_off(q.b1, 1, 0, blk, q.c1);
      }
    }
    if (a.Kin == 0 && !(decltype(which)::value == 1 && fn == 0 && xstate)) {   // no PCM at all: any mapped address
      s0 = s1 = reinterpret_cast<const pcm_t*>(a.X);
      ok = false;
    }
    if constexpr (IOF != 0) load_row_h<typename RowFmt<IOF>::type, CMODE, R>(s0, s1, C, q.has1, lane, dst);
    else load_row<CMODE, ((AC_NT_LOAD >> decltype(which)::value) & 1) != 0, R>(s0, s1, C, q.has1, lane, dst);
    return ok;
  };
  constexpr std::integral_constant<int, 0> kCur{};
  constexpr std::integral_constant<int, 1> kPrv{};
  auto zero_row = [](v4f (&v)[R]) {
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = v4f{0.f, 0.f, 0.f, 0.f};
  };

  while (pair < npairs && left > 0) {
    const Pair pq = make_pair<CMODE>(pair, C, a.nsig);
    C2 z[R];
    if (R == 8) {
      // both blocks in flight together; one lane-reversal exchange for the odd halves of both:
      // previous block in [0, 4 KB), current block in [4 KB, 8 KB)
      v4f cb[R], pb[R];
      const bool cur_ok = issue_loads(kCur, pair, n, cb);
      if constexpr (EPI) load_p1<R>(a.tab, lane, p1);
      const bool prv_ok = issue_loads(kPrv, pair, n, pb);
      if (!cur_ok) zero_row(cb);   // edge frames only (wave-uniform)
      if (!prv_ok) zero_row(pb);
      if constexpr (IOF != 1) {
        if (a.state_out && n == a.Kin - 1)   // streaming: the chunk's last block is the next chunk's block -1
          store_row<CMODE, R>(a.state_out + row_off(pq.b0, 1, 0, blk, pq.c0), a.state_out + row_off(pq.b1, 1, 0, blk, pq.c1),
                              C, pq.has1, lane, cb);
      }
#if AC_REV_DPP
      v2f xop_r[R], xoc_r[R];   // odd halves of the two blocks, from lane 63 - lane
#pragma unroll
      for (int c = 0; c < R; ++c) {
        xop_r[c] = rev64(v2f{pb[c].z, pb[c].w});
        xoc_r[c] = rev64(v2f{cb[c].z, cb[c].w});
      }
#else
      wave_sync();
      {
        char* w = buf + 8 * (63 - lane);
#pragma unroll
        for (int c = 0; c < R; ++c) {
          *reinterpret_cast<v2f*>(w + 512 * c) = v2f{pb[c].z, pb[c].w};
          *reinterpret_cast<v2f*>(w + 4096 + 512 * c) = v2f{cb[c].z, cb[c].w};
        }
      }
      wave_sync();
      const char* rd = buf + 8 * lane;
#endif
#pragma unroll
      for (int r = 0; r < R; ++r) {
#if AC_REV_DPP
        const v2f xop = xop_r[(R / 2 - 1 - r) & (R - 1)], xoc = xoc_r[(R / 2 - 1 - r) & (R - 1)];
#else
        const v2f xop = *reinterpret_cast<const v2f*>(rd + 512 * ((R / 2 - 1 - r) & (R - 1)));
        const v2f xoc = *reinterpret_cast<const v2f*>(rd + 4096 + 512 * ((R / 2 - 1 - r) & (R - 1)));
#endif
        const v4f& gp = pb[(r + R / 2) & (R - 1)];
        const v4f& gc = cb[(r + R / 2) & (R - 1)];
        const v2f xep = v2f{gp.x, gp.y}, xec = v2f{gc.x, gc.y};
        const v2f ab = reinterpret_cast<const v2f*>(tab + G::I_COEF)[r * 64 + lane];
        const v2f carry = ab.y * xep + ab.x * xop;
        const v2f cur = (r < R / 2) ? (ab.y * xoc - ab.x * xec) : (ab.x * xec - ab.y * xoc);
        // element e = lane + 64 r: v[2e] + i v[N-1-2e]; for e < N/4 the real part comes from the previous block
        const C2 v = (r < R / 2) ? C2{carry, cur} : C2{cur, carry};
        if constexpr (NOPRE) {
          const C2 v0 = {v.re * a.pre_re - v.im * a.pre_im, v.re * a.pre_im + v.im * a.pre_re};
          z[r] = cmul(v0, reinterpret_cast<const v2f*>(tab + G::I_POST)[r * 64 + lane]);
        } else {
          z[r] = cmul(v, reinterpret_cast<const v2f*>(tab + G::I_PRE)[r * 64 + lane]);
        }
      }
    } else {
      // larger frames: one block at a time (registers), each with its own lane-reversal exchange
      v2f carry[R];
      {
        v4f pb[R];
        const bool prv_ok = issue_loads(kPrv, pair, n, pb);
        if (!prv_ok) zero_row(pb);
        v2f xo_in[R], xo[R];
#pragma unroll
        for (int c = 0; c < R; ++c) xo_in[c] = v2f{pb[c].z, pb[c].w};
        rev_exchange<R / 2 - 1, R>(buf, lane, xo_in, xo);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const v4f& gp = pb[(r + R / 2) & (R - 1)];
          const v2f ab = reinterpret_cast<const v2f*>(tab + G::I_COEF)[r * 64 + lane];
          carry[r] = ab.y * v2f{gp.x, gp.y} + ab.x * xo[r];
        }
      }
      v4f cb[R];
      const bool cur_ok = issue_loads(kCur, pair, n, cb);
      if (!cur_ok) zero_row(cb);
      if constexpr (IOF != 1) {
        if (a.state_out && n == a.Kin - 1)
          store_row<CMODE, R>(a.state_out + row_off(pq.b0, 1, 0, blk, pq.c0), a.state_out + row_off(pq.b1, 1, 0, blk, pq.c1),
                              C, pq.has1, lane, cb);
      }
      v2f xo_in[R], xo[R];
#pragma unroll
      for (int c = 0; c < R; ++c) xo_in[c] = v2f{cb[c].z, cb[c].w};
      rev_exchange<R / 2 - 1, R>(buf, lane, xo_in, xo);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const v4f& gc = cb[(r + R / 2) & (R - 1)];
        const v2f xec = v2f{gc.x, gc.y};
        const v2f ab = reinterpret_cast<const v2f*>(tab + G::I_COEF)[r * 64 + lane];
        const v2f cur = (r < R / 2) ? (ab.y * xo[r] - ab.x * xec) : (ab.x * xec - ab.y * xo[r]);
        const C2 v = (r < R / 2) ? C2{carry[r], cur} : C2{cur, carry[r]};
        z[r] = cmul(v, reinterpret_cast<const v2f*>(tab + G::I_PRE)[r * 64 + lane]);
      }
    }
    fft_wave<R>(z, buf, tab, p1, lane);
    v4f row[R];
    {
      // bin k = lane + 64 j: X[2k] = Re (granule k, this lane), X[N-1-2k] = -Im (granule N/2-1-k, lane 63 - lane)
      v2f xe[R], xo_in[R], xo[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const C2 r = cmul_negim(z[j], reinterpret_cast<const v2f*>(tab + G::I_POST)[j * 64 + lane]);
        xe[j] = r.re;
        xo_in[j] = r.im;
      }
      rev_exchange<R - 1, R>(buf, lane, xo_in, xo);
#pragma unroll
      for (int i = 0; i < R; ++i) row[i] = v4f{xe[i].x, xe[i].y, xo[i].x, xo[i].y};
    }
    const size_t o0 = row_off(pq.b0, a.F, n, blk, pq.c0), o1 = row_off(pq.b1, a.F, n, blk, pq.c1);
    const size_t t0 = ((size_t)pq.b0 * a.F + (size_t)n) * C + pq.c0, t1 = ((size_t)pq.b1 * a.F + (size_t)n) * C + pq.c1;
    if constexpr (IOF == 2) {
      int16_t* Xh = reinterpret_cast<int16_t*>(a.X);
      store_row_h<Bf16Fmt, CMODE, R>(Xh + o0, Xh + o1, C, pq.has1, lane, row);
    } else {
      store_row<CMODE, R, (AC_NT_X != 0)>(a.X + o0, a.X + o1, C, pq.has1, lane, row);
    }
    if constexpr (EPI) {
      if (a.dbn) {   // amplitude_to_dB_norm of the coefficients (psychoacoustic.py:87-100), from the registers
        v4f d[R];
#pragma unroll
        for (int i = 0; i < R; ++i) d[i] = v4f{db_of(row[i].x, 1), db_of(row[i].y, 1), db_of(row[i].z, 1), db_of(row[i].w, 1)};
        store_row<CMODE, R>(a.dbn + o0, a.dbn + o1, C, pq.has1, lane, d);
      }
    }
    // next frame of this wave
    pair += dpair;
    n += dn;
    if (n >= a.F) {
      n -= a.F;
      ++pair;
    }
    --left;
    if constexpr (PSY) {
      v2f tt;
      v4f th[R];
      if constexpr (IOF == 2) {
        // the masking model sees the spectrum the caller gets: the bfloat16-rounded coefficients
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const s2 lo = Bf16Fmt::enc2(row[i].x, row[i].y), hi = Bf16Fmt::enc2(row[i].z, row[i].w);
          row[i] = v4f{Bf16Fmt::dec(lo.x), Bf16Fmt::dec(lo.y), Bf16Fmt::dec(hi.x), Bf16Fmt::dec(hi.y)};
        }
      }
      bool emitted = false;
      if constexpr (EPI) {
        if (a.noisy) {
          // add_noise (psychoacoustic.py:150-167) on the frame: the coefficients are no longer in registers (the masking
          // model needed them all), so the row the wave stored a moment ago comes back from L2; each granule of the
          // threshold row is stored and turned into its noisy coefficients as it comes out of the entry lookup.  Element
          // pairs of the flattened tensor share one Box-Muller draw, exactly as in ac_add_noise (granule i4 = elements
          // 4 i4 .. 4 i4 + 3)
          NoisyEmit<R> emit;
          emit.X_row = reinterpret_cast<const v4f*>(a.X + o0) + lane;
          emit.thr_row = reinterpret_cast<v4f*>(a.thr + o0) + lane;
          emit.noisy_row = reinterpret_cast<v4f*>(a.noisy + o0) + lane;
          emit.i4base = (uint64_t)(o0 >> 2) + (uint64_t)lane;
          emit.key = a.noise_key;
          psy_stage<R, true, true, SPREAD, false, NoisyEmit<R>&>(row, lds, buf, pimg, pc, a.psy, lane, tt, th, emit);
          emitted = true;
        }
      }
      if (!emitted) psy_stage<R, true, true, SPREAD, IOF == 2>(row, lds, buf, pimg, pc, a.psy, lane, tt, th);
      if constexpr (IOF == 2) {
        int16_t* th_h = reinterpret_cast<int16_t*>(a.thr);
        int16_t* t_h = reinterpret_cast<int16_t*>(a.t);
        store_row_h<Bf16Fmt, CMODE, R>(th_h + o0, th_h + o1, C, pq.has1, lane, th);
        if (lane == 0) {
          const s2 e = Bf16Fmt::enc2(tt.x, tt.y);
          t_h[t0] = e.x;
          if (pq.has1) t_h[t1] = e.y;
        }
      } else {
        if (!emitted) store_row<CMODE, R, (AC_NT_THR != 0)>(a.thr + o0, a.thr + o1, C, pq.has1, lane, th);
        if (lane == 0) {
          a.t[t0] = tt.x;
          if (pq.has1) a.t[t1] = tt.y;
        }
      }
    }
  }
}

template <int R, int CMODE, bool PSY, int NW, int IOF = 0, int SPREAD = 0, bool EPI = false>
__global__ __launch_bounds__(NW * 64, (wpe<R, CMODE, PSY, SPREAD>())) void k_fwd_fast(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[fwd_lds_bytes<R, PSY, NW, SPREAD>()];
  fwd_fast_body<R, CMODE, PSY, NW, IOF, SPREAD, EPI>(a, lds, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------------
// synthesis
// ------------------------------------------------------------------------------------------------------
struct InvArgs {
  const float* X;          // [B, Kp, N, C]
  void* x;                 // [B, nblk*N, C] float32, or int16 PCM for the PCM16 kernels
  const float* tail_in;    // [B, C, N/2] or null
  float* tail_out;         // [B, C, N/2] or null
  const float* tab;
  int B, Kp, nblk, C, seglen, nseg;
  int rev;                 // 1: workgroups walk the spectrum from its end (AC_INV_REV tuning hook)
  long long npairs, nsig;  // see Pair
  long long ntasks;        // npairs * nseg
};

// DCT-IV of one frame held in natural order: returns (now, nxt) per output element k = lane + 64 j.
// Element e = lane + 64 r is X[2e] (granule e, this lane) + i X[N-1-2e] (granule N/2-1-e, lane 63 - lane).
template <int R>
__device__ __forceinline__ void idct_frame(const v4f (&frm)[R], char* buf, gtab_t tab, const v2f (&p1)[R], int lane,
                                           v2f (&now)[R], v2f (&nxt)[R]) {
  using G = Geo<R>;
  C2 z[R];
  {
    v2f xo_in[R], xo[R];
#pragma unroll
    for (int c = 0; c < R; ++c) xo_in[c] = v2f{frm[c].z, frm[c].w};
    rev_exchange<R - 1, R>(buf, lane, xo_in, xo);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const C2 v = {v2f{frm[r].x, frm[r].y}, xo[r]};
      z[r] = cmul(v, reinterpret_cast<const v2f*>(tab + G::I_PRE)[r * 64 + lane]);
    }
  }
  fft_wave<R>(z, buf, tab, p1, lane);
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const C2 r = cmul_negim(z[j], reinterpret_cast<const v2f*>(tab + G::I_POST)[j * 64 + lane]);
    // u[2k] = Re, u[N-1-2k] = -Im; k < N/4 (j < R/2): u[2k] belongs to this block, u[N-1-2k] to the next
    if (j < R / 2) {
      now[j] = r.re;
      nxt[j] = r.im;
    } else {
      now[j] = r.im;
      nxt[j] = r.re;
    }
  }
}

template <int R, int CMODE, int NW, int IOF = 0>
__device__ __forceinline__ void inv_fast_body(const InvArgs& a, char* lds, const int bid, const int nblocks) {
  using G = Geo<R>;
  constexpr int FH = G::FH;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  load_tables<NW, WAVE_LDS, G::I_LDS, 0>(lds, a.tab + G::I_TOTAL, nullptr);
  char* buf = lds + wave * WAVE_LDS;
  gtab_t tab = reinterpret_cast<const float*>(lds + NW * WAVE_LDS);
  v2f p1[R];
  load_p1<R>(a.tab + G::I_TOTAL, lane, p1);
  const int C = a.C;
  const size_t blk = (size_t)G::FN * C;
  // Synthesis block n overlap-adds the DCT-IV of frames n and n-1, so a wave walks a short strip of consecutive output
  // blocks with the aliased half of the last frame carried in registers.  The waves of a workgroup own consecutive
  // strips: the first block of a strip is finished last, when the neighbouring wave hands over the aliased half of
  // its last frame through LDS -- only the first wave of a workgroup pays an extra DCT-IV (of the frame before its
  // strip).  Workgroups are dispatched in order, which keeps the window of memory in flight contiguous.
  constexpr bool COOP = (R == 8);   // hand-over between waves (the 2048-filter kernel has no registers to spare)
  const long long task_raw = (long long)(a.rev ? nblocks - 1 - bid : bid) * NW + wave;
  const bool valid = task_raw < a.ntasks;
  const long long task = valid ? task_raw : a.ntasks - 1;   // idle waves of the last workgroup only join the barriers
  const int sgm = (int)(task % a.nseg);
  const Pair pq = make_pair<CMODE>(task / a.nseg, C, a.nsig);
  const bool has1 = pq.has1;
  const int n0 = sgm * a.seglen;
  const int n1 = min(a.nblk, n0 + a.seglen);
  using spec_t = typename std::conditional<IOF == 2, int16_t, float>::type;   // storage of the spectrum
  const spec_t* X0 = reinterpret_cast<const spec_t*>(a.X) + row_off(pq.b0, a.Kp, 0, blk, pq.c0);   // frame 0 of the two signals
  const spec_t* X1 = reinterpret_cast<const spec_t*>(a.X) + row_off(pq.b1, a.Kp, 0, blk, pq.c1);
  auto load_frame = [&](const spec_t* r0, const spec_t* r1, v4f (&dst)[R]) {
    if constexpr (IOF == 2) load_row_h<Bf16Fmt, CMODE, R>(r0, r1, C, has1, lane, dst);
    else load_row<CMODE, (AC_NT_LOAD & 4) != 0, R>(r0, r1, C, has1, lane, dst);
  };
  const size_t ts0 = ((size_t)pq.b0 * C + pq.c0) * FH, ts1 = ((size_t)pq.b1 * C + pq.c1) * FH;   // stream state rows
  constexpr bool AHEAD = (R == 8);   // the next frame in flight while the current one is transformed
  const bool left = valid && n0 >= 1;              // block n0 needs the aliased half of frame n0-1 ...
  const bool deferred = COOP && left && wave > 0;  // ... which the previous wave (strip sgm-1 of the same signals) hands over

  // block n from the current frame's half (now) and the previous frame's aliased half (cin):
  // with (a, b) = COEF[k]: o1 = a now + b cin -> out[j], o2 = b now - a cin -> out[N-1-j]  
  // k < N/4: j = N/2-1 - 2k (odd: granule N/4-1-k, lane 63 - lane), N-1-j = N/2 + 2k (even: granule N/4+k, this lane)
  // else     j = 2k - N/2 (even: granule k - N/4, this lane),       N-1-j = 3N/2-1 - 2k (odd: granule 3N/4-1-k)
  auto emit = [&](int n, const v2f (&now)[R], const v2f (&cin)[R]) {
    v4f row[R];
    v2f xe[R], xo_in[R], xo[R];
#pragma unroll
    for (int j2 = 0; j2 < R; ++j2) {
      const v2f ab = reinterpret_cast<const v2f*>(tab + G::I_COEF)[j2 * 64 + lane];
      const v2f o1 = ab.x * now[j2] + ab.y * cin[j2];
      const v2f o2 = ab.y * now[j2] - ab.x * cin[j2];
      xe[(j2 + R / 2) & (R - 1)] = (j2 < R / 2) ? o2 : o1;
      xo_in[j2] = (j2 < R / 2) ? o1 : o2;
    }
    rev_exchange<R / 2 - 1, R>(buf, lane, xo_in, xo);
#pragma unroll
    for (int i = 0; i < R; ++i) row[i] = v4f{xe[i].x, xe[i].y, xo[i].x, xo[i].y};
    const size_t o0 = row_off(pq.b0, a.nblk, n, blk, pq.c0), o1 = row_off(pq.b1, a.nblk, n, blk, pq.c1);
    if constexpr (IOF != 0)
      store_row_h<typename RowFmt<IOF>::type, CMODE, R>(static_cast<int16_t*>(a.x) + o0, static_cast<int16_t*>(a.x) + o1, C, has1, lane, row);
    else store_row<CMODE, R>(static_cast<float*>(a.x) + o0, static_cast<float*>(a.x) + o1, C, has1, lane, row);
  };

  v2f carry[R], now0[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    carry[r] = v2f{0.f, 0.f};
    now0[r] = v2f{0.f, 0.f};
  }
  if (valid) {
    v4f ahead[R];
    if (left && !deferred) {
      // aliased half of frame n0-1 (always an existing frame: n0-1 < Kp)
      v4f row[R];
      load_frame(X0 + (size_t)(n0 - 1) * blk, X1 + (size_t)(n0 - 1) * blk, row);
      if (AHEAD && n0 < a.Kp) load_frame(X0 + (size_t)n0 * blk, X1 + (size_t)n0 * blk, ahead);
      v2f dummy[R];
      idct_frame<R>(row, buf, tab, p1, lane, dummy, carry);
    } else {
      if (AHEAD && n0 < a.Kp) load_frame(X0 + (size_t)n0 * blk, X1 + (size_t)n0 * blk, ahead);
      if (!left && a.tail_in) {
#pragma unroll
        for (int j2 = 0; j2 < R; ++j2) {
          const int k = lane + 64 * j2;
          const int j = (j2 < R / 2) ? (FH - 1 - 2 * k) : (2 * k - FH);
          carry[j2].x = a.tail_in[ts0 + j];
          carry[j2].y = has1 ? a.tail_in[ts1 + j] : 0.f;
        }
      }
    }

    for (int n = n0; n < n1; ++n) {
      v2f now[R], nxt[R];
      if (n < a.Kp) {
        if (!AHEAD) load_frame(X0 + (size_t)n * blk, X1 + (size_t)n * blk, ahead);
        idct_frame<R>(ahead, buf, tab, p1, lane, now, nxt);
        if (AHEAD && n + 1 < n1 && n + 1 < a.Kp)
          load_frame(X0 + (size_t)(n + 1) * blk, X1 + (size_t)(n + 1) * blk, ahead);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          now[r] = v2f{0.f, 0.f};
          nxt[r] = v2f{0.f, 0.f};
        }
      }
      if (deferred && n == n0) {
#pragma unroll
        for (int r = 0; r < R; ++r) now0[r] = now[r];   // finished after the hand-over
      } else {
        emit(n, now, carry);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) carry[r] = nxt[r];
    }

    if (a.tail_out && n1 == a.nblk) {
#pragma unroll
      for (int j2 = 0; j2 < R; ++j2) {
        const int k = lane + 64 * j2;
        const int j = (j2 < R / 2) ? (FH - 1 - 2 * k) : (2 * k - FH);
        a.tail_out[ts0 + j] = carry[j2].x;
        if (has1) a.tail_out[ts1 + j] = carry[j2].y;
      }
    }
  }

  if (COOP) {
    // hand the aliased half of the strip's last frame to the wave that owns the next strip
    wave_sync();
    if (valid) {
#pragma unroll
      for (int r = 0; r < R; ++r) *reinterpret_cast<v2f*>(buf + 8 * lane + 512 * r) = carry[r];
    }
    __syncthreads();
    v2f cin[R];
    if (deferred) {
#pragma unroll
      for (int r = 0; r < R; ++r) cin[r] = *reinterpret_cast<const v2f*>(buf - WAVE_LDS + 8 * lane + 512 * r);
    }
    __syncthreads();   // every hand-over has been read: the buffers may be reused for the last exchange
    if (deferred) emit(n0, now0, cin);
  }
}

template <int R, int CMODE, int NW, int IOF = 0>
__global__ __launch_bounds__(NW * 64, ((IOF == 2 || (IOF == 1 && !(R == 8 && CMODE != 1))) ? 2 : wpe<R, CMODE>())) void k_inv_fast(InvArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[NW * WAVE_LDS + Geo<R>::TAB_LDS];
  inv_fast_body<R, CMODE, NW, IOF>(a, lds, (int)blockIdx.x, (int)gridDim.x);
}

// Streaming duplex (BASELINE configs[4]): the analysis of chunk i + 1 and the synthesis of chunk i in ONE launch -- the
// first nfwd workgroups run the analysis body, the others the synthesis body.  A chunk of one clip is a few hundred wave
// tasks: two dependent launches of ~8 us each are latency, not bandwidth, and the two halves are independent.
template <int R, int CMODE, bool PSY, int NW, int SPREAD>
__global__ __launch_bounds__(NW * 64, (CMODE == 0 ? wpe<R, CMODE, PSY, SPREAD>() : 2)) void k_duplex_fast(FwdArgs fa, InvArgs ia, int nfwd) {
  constexpr int LF = fwd_lds_bytes<R, PSY, NW, SPREAD>(), LI = NW * WAVE_LDS + Geo<R>::TAB_LDS;
  __shared__ __attribute__((aligned(16))) char lds[LF > LI ? LF : LI];
  const int b = (int)blockIdx.x;   // (uniform per workgroup: the barriers inside the bodies stay consistent)
  if (b < nfwd) fwd_fast_body<R, CMODE, PSY, NW, 0, SPREAD, false>(fa, lds, b, nfwd);
  else inv_fast_body<R, CMODE, NW, 0>(ia, lds, b - nfwd, (int)gridDim.x - nfwd);
}

// ------------------------------------------------------------------------------------------------------
// stand-alone tonality / threshold on a spectrum in HBM: one wave per (b, frame, channel pair)
// ------------------------------------------------------------------------------------------------------
struct PsyArgs {
  const float* X;
  const float* t_in;
  float* t_out;
  float* thr;
  PsyParams psy;
  int C, F;
  long long nsig;     // B * C
  long long ntasks;   // npairs * F
};

template <int R, int CMODE, bool WANT_T, bool WANT_THR, int NW, int SPREAD = 0, int IOF = 0>
__global__ __launch_bounds__(NW * 64, (R == 8 ? AC_WPE : 2)) void k_psy_fast(PsyArgs a) {
  using P = PsyGeo<R>;
  __shared__ __attribute__((aligned(16))) char lds[NW * WAVE_LDS_PSY + P::PSY_LDS + mf_lds(SPREAD)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* pimg = reinterpret_cast<const uint32_t*>(lds + NW * WAVE_LDS_PSY);
  if (WANT_THR) load_tables<NW, WAVE_LDS_PSY, 0, P::PL_LDS, P::PL_MF, mf_lds(SPREAD)>(lds, nullptr, a.psy.tab);
  const long long task = (long long)blockIdx.x * NW + wave;
  if (task >= a.ntasks) return;
  char* buf = lds + wave * WAVE_LDS_PSY;
  if (WANT_THR) *reinterpret_cast<v2f*>(buf + ZERO_OFF) = v2f{0.f, 0.f};
  const int f = (int)(task % a.F);
  const int C = a.C;
  const Pair pq = make_pair<CMODE>(task / a.F, C, a.nsig);
  const bool has1 = pq.has1;
  const size_t blk = (size_t)P::FN * C;
  const size_t o0 = row_off(pq.b0, a.F, f, blk, pq.c0), o1 = row_off(pq.b1, a.F, f, blk, pq.c1);
  const size_t t0 = ((size_t)pq.b0 * a.F + (size_t)f) * C + pq.c0, t1 = ((size_t)pq.b1 * a.F + (size_t)f) * C + pq.c1;
  v4f row[R], th[R];
  if constexpr (IOF == 2) {
    const int16_t* Xh = reinterpret_cast<const int16_t*>(a.X);
    load_row_h<Bf16Fmt, CMODE, R>(Xh + o0, Xh + o1, C, has1, lane, row);
  } else {
    load_row<CMODE, false, R>(a.X + o0, a.X + o1, C, has1, lane, row);
  }
  v2f tt = {0.f, 0.f};
  if (!WANT_T) {
    if constexpr (IOF == 2) {
      const int16_t* th_in = reinterpret_cast<const int16_t*>(a.t_in);
      tt.x = Bf16Fmt::dec(th_in[t0]);
      tt.y = has1 ? Bf16Fmt::dec(th_in[t1]) : 0.f;
    } else {
      tt.x = a.t_in[t0];
      tt.y = has1 ? a.t_in[t1] : 0.f;
    }
  }
  PsyLane<R> pc;
  if (WANT_THR) pc = load_psy_lane<R>(a.psy.tab, lane, (uint32_t)(wave * WAVE_LDS_PSY));
  psy_stage<R, WANT_T, WANT_THR, SPREAD, IOF == 2>(row, lds, buf, pimg, pc, a.psy, lane, tt, th);
  if constexpr (IOF == 2) {
    if (WANT_T && lane == 0) {
      int16_t* t_h = reinterpret_cast<int16_t*>(a.t_out);
      const s2 e = Bf16Fmt::enc2(tt.x, tt.y);
      t_h[t0] = e.x;
      if (has1) t_h[t1] = e.y;
    }
    if (WANT_THR) {
      int16_t* th_h = reinterpret_cast<int16_t*>(a.thr);
      store_row_h<Bf16Fmt, CMODE, R>(th_h + o0, th_h + o1, C, has1, lane, th);
    }
  } else {
    if (WANT_T && lane == 0) {
      a.t_out[t0] = tt.x;
      if (has1) a.t_out[t1] = tt.y;
    }
    if (WANT_THR) store_row<CMODE, R>(a.thr + o0, a.thr + o1, C, has1, lane, th);
  }
}

// ------------------------------------------------------------------------------------------------------
// backward passes of the masking model at wave level (the adjoints of psy_stage): one wave per (frame, signal pair)
// ------------------------------------------------------------------------------------------------------
struct PsyBwdArgs {
  const float* X;
  const float* t;        // tonality used by the forward pass (threshold backward)
  const float* g_thr;    // [B,F,N,C] (threshold backward)
  const float* g_t;      // [B,F,1,C] (tonality backward)
  float* g_X;            // [B,F,N,C]
  float* g_t_out;        // [B,F,1,C] (threshold backward)
  PsyParams psy;
  int C, F, accumulate;
  long long nsig, ntasks;
};

// sums of an image of N per-bin values over the bins of each Bark band (lane = band), through the same chunk sums,
// edge offsets and gather lists as the forward Bark mapping: returns w_first v[f0] + w_last v[f1] + sum over the
// interior bins, the two edge weights given per lane.
template <int R>
__device__ __forceinline__ v2f band_sums(const v4f (&v)[R], char* lds0, char* buf, const uint32_t* pimg,
                                         const PsyLane<R>& pc, v2f w_first, v2f w_last, v2f w_inner, int lane) {
  using P = PsyGeo<R>;
  v2f e0 = {0.f, 0.f}, e1 = {0.f, 0.f}, in0 = {0.f, 0.f}, in1 = {0.f, 0.f};
#pragma unroll
  for (int h = 0; h < P::NH; ++h) {
    wave_sync();
    {
      const int lsw = lane ^ ((lane >> 4) & 3);
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<v4f*>(buf + 16 * lsw + 1024 * i) = v[8 * h + i];
    }
    wave_sync();
    {
      const int x = (lane >> 2) & 3;
      const char* cb = buf + 64 * lane;
      const int o0 = 16 * x, o1 = 16 * (1 ^ x), o2 = 16 * (2 ^ x), o3 = 16 * (3 ^ x);
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const char* c2 = cb + 4096 * i2;
        const v4f g0 = *reinterpret_cast<const v4f*>(c2 + o0), g1 = *reinterpret_cast<const v4f*>(c2 + o1),
                  g2 = *reinterpret_cast<const v4f*>(c2 + o2), g3 = *reinterpret_cast<const v4f*>(c2 + o3);
        const v4f s = (g0 + g1) + (g2 + g3);
        *reinterpret_cast<v2f*>(buf + S8_OFF + 8 * lane + 512 * i2) = v2f{s.x + s.z, s.y + s.w};
      }
    }
    wave_sync();
    const uint32_t edge = in_loop(__float_as_uint(pc.bc0[h].x));
    e0 += *reinterpret_cast<const v2f*>(lds0 + (edge & 0xffffu));   // zero slot when the edge bin is in another half
    e1 += *reinterpret_cast<const v2f*>(lds0 + (edge >> 16));
#pragma unroll
    for (int hlf = 0; hlf < P::PL_HALF; ++hlf) {
      const uint32_t w = pimg[P::PL_LST + (h * P::PL_HALF + hlf) * 64 + lane];
      in0 += *reinterpret_cast<const v2f*>(buf + (w & 0xffffu));
      in1 += *reinterpret_cast<const v2f*>(buf + (w >> 16));
    }
  }
  return w_first * e0 + w_last * e1 + w_inner * (in0 + in1);
}

// per-bin values from two per-band entry values (entry 2j: bins of band j alone, entry 2j+1: the bin shared by bands j
// and j+1), gathered through the lane's entry offsets: out[i] = (entry(2q), entry(2q+1)) for granule q = 64 i + lane
template <int R>
__device__ __forceinline__ void entry_gather(v2f own, v2f shared, char* lds0, char* buf, const PsyLane<R>& pc, int lane,
                                             v4f (&out)[R]) {
  wave_sync();
  *reinterpret_cast<v4f*>(buf + 16 * lane) = v4f{own.x, own.y, shared.x, shared.y};   // entry e at byte 8 e
  wave_sync();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const v4f ww = pc.idx[i >> 2];
    const float wf = (i & 3) == 0 ? ww.x : (i & 3) == 1 ? ww.y : (i & 3) == 2 ? ww.z : ww.w;
    const uint32_t w = in_loop(__float_as_uint(wf));
    const v2f a0 = *reinterpret_cast<const v2f*>(lds0 + (w & 0xffffu));
    const v2f a1 = *reinterpret_cast<const v2f*>(lds0 + (w >> 16));
    out[i] = v4f{a0.x, a0.y, a1.x, a1.y};
  }
}

// band x band product with the Toeplitz spreading matrix: FORWARD: out_j = sum_i v_i S[i, j] = sum_i v_i g[64 - i + j];
// otherwise the transposed product out_i = sum_j S[i, j] v_j = sum_j v_j g[64 - i + j]  (lane = the output index)
template <bool FORWARD>
__device__ __forceinline__ v2f spread(v2f v, char* buf, const uint32_t* pimg_g, int lane) {
  wave_sync();
  *reinterpret_cast<v2f*>(buf + 8 * lane) = v;
  wave_sync();
  v2f acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
  const float* gp = reinterpret_cast<const float*>(pimg_g) + 64 + (FORWARD ? lane : -lane);
#pragma unroll 8
  for (int i = 0; i < 64; i += 2) {
    const v4f qq = *reinterpret_cast<const v4f*>(buf + 8 * i);   // v_i, v_{i+1} (broadcast read)
    acc0 += v2f{qq.x, qq.y} * (FORWARD ? gp[-i] : gp[i]);
    acc1 += v2f{qq.z, qq.w} * (FORWARD ? gp[-i - 1] : gp[i + 1]);
  }
  return acc0 + acc1;
}

// d thr / d X and d thr / d t  (the chain of k_threshold_bwd_generic in ac_generic.hip, psychoacoustic.py:122-148 with
// 169-210, 301-331), and d t / d X (psychoacoustic.py:102-120) when TONALITY
template <int R, int CMODE, bool TONALITY, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_psy_bwd_fast(PsyBwdArgs a) {
  using P = PsyGeo<R>;
  __shared__ __attribute__((aligned(16))) char lds[NW * WAVE_LDS_PSY + P::PSY_LDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* pimg = reinterpret_cast<const uint32_t*>(lds + NW * WAVE_LDS_PSY);
  if (!TONALITY) load_tables<NW, WAVE_LDS_PSY, 0, P::PL_LDS>(lds, nullptr, a.psy.tab);
  const long long task = (long long)blockIdx.x * NW + wave;
  if (task >= a.ntasks) return;
  char* buf = lds + wave * WAVE_LDS_PSY;
  const int f = (int)(task % a.F);
  const int C = a.C;
  const Pair pq = make_pair<CMODE>(task / a.F, C, a.nsig);
  const bool has1 = pq.has1;
  const size_t blk = (size_t)P::FN * C;
  const size_t o0 = row_off(pq.b0, a.F, f, blk, pq.c0), o1 = row_off(pq.b1, a.F, f, blk, pq.c1);
  const size_t t0 = ((size_t)pq.b0 * a.F + (size_t)f) * C + pq.c0, t1 = ((size_t)pq.b1 * a.F + (size_t)f) * C + pq.c1;
  v4f x[R];
  load_row<CMODE, false, R>(a.X + o0, a.X + o1, C, has1, lane, x);

  if (TONALITY) {
    // t = min(c' [mean ln max(eps, I) - ln(mean I + eps)], 1), c' = (10 / ln 10) / (-60):
    // d t / d X_f = c' / N ([I_f > eps] / I_f - 1 / (mean I + eps)) 2 X_f   where the clamp is inactive
    v2f slog = {0.f, 0.f}, ssq = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const v4f I = x[i] * x[i];
      const v2f ie = v2f{I.x, I.y}, io = v2f{I.z, I.w};
      ssq += ie + io;
      slog += log2v(maxv(ie, kEps) * maxv(io, kEps));
    }
    slog.x = wave_sum(slog.x);
    slog.y = wave_sum(slog.y);
    ssq.x = wave_sum(ssq.x);
    ssq.y = wave_sum(ssq.y);
    const v2f am = ssq * (1.0f / P::FN) + kEps;
    const v2f tt = (3.0102999566398120f / -60.0f) * (slog * (1.0f / P::FN) - log2v(am));
    const float cc = (10.0f / 2.302585092994046f) / -60.0f;
    v2f g = v2f{a.g_t[t0], has1 ? a.g_t[t1] : 0.f} * (cc / P::FN);
    g.x = tt.x < 1.0f ? g.x : 0.f;
    g.y = tt.y < 1.0f ? g.y : 0.f;
    const v2f inv_am = v2f{1.0f / am.x, 1.0f / am.y};
    v4f gx[R];
    if (a.accumulate) load_row<CMODE, false, R>(a.g_X + o0, a.g_X + o1, C, has1, lane, gx);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const v4f I = x[i] * x[i];
      v4f r;
      r.x = (I.x > kEps ? 1.0f / I.x : 0.f) - inv_am.x;
      r.y = (I.y > kEps ? 1.0f / I.y : 0.f) - inv_am.y;
      r.z = (I.z > kEps ? 1.0f / I.z : 0.f) - inv_am.x;
      r.w = (I.w > kEps ? 1.0f / I.w : 0.f) - inv_am.y;
      const v4f d = r * v4f{g.x, g.y, g.x, g.y} * 2.0f * x[i];
      gx[i] = a.accumulate ? gx[i] + d : d;
    }
    store_row<CMODE, R>(a.g_X + o0, a.g_X + o1, C, has1, lane, gx);
    return;
  } else {
    *reinterpret_cast<v2f*>(buf + ZERO_OFF) = v2f{0.f, 0.f};
    const PsyLane<R> pc = load_psy_lane<R>(a.psy.tab, lane, (uint32_t)(wave * WAVE_LDS_PSY));
    const PsyParams& pp = a.psy;
    const v2f t = v2f{a.t[t0], has1 ? a.t[t1] : 0.f};
    // true edge weights of the forward Bark mapping (only one half carries each)
    float wfirst = 0.f, wlast = 0.f;
#pragma unroll
    for (int h = 0; h < P::NH; ++h) {
      wfirst += pc.bc0[h].y;
      wlast += pc.bc0[h].z;
    }
    const float quiet = pc.bc0[0].w, beta = pc.bc1.x, rho = pc.bc1.y, u0 = pc.bc1.z, u1 = pc.bc1.w;
    // ---- forward recompute: P, Q, A, fac, Y, T, G per band (lane) ----
    v4f I[R];
#pragma unroll
    for (int i = 0; i < R; ++i) I[i] = x[i] * x[i];
    const v2f one = {1.f, 1.f};
    const v2f Pj = band_sums<R>(I, lds, buf, pimg, pc, one * wfirst, one * wlast, one, lane);
    const v2f Q = exp2v(pp.alpha * log2v(maxv(Pj, kEps)));
    const v2f A = spread<true>(Q, buf, pimg + P::PL_G, lane);
    const v2f dOdt = one * ((1.0f - pp.drown) * (beta + 9.0f));
    const v2f offset = (1.0f - pp.drown) * (t * beta + 9.0f * t + 5.5f);
    const v2f fac = exp2v(offset * (-pp.alpha * 0.33219280948873623f));
    const v2f Y = fac * A;
    const v2f T = exp2v(pp.inv_alpha * log2v(maxv(Y, kEps)));
    const v2f G = maxv(T, quiet);
    v2f Gn;
    Gn.x = __shfl_down(G.x, 1, 64);
    Gn.y = __shfl_down(G.y, 1, 64);
    // ---- E per entry, thr and d L / d E per bin ----
    const v2f E0 = G * rho, E1 = G * u0 + Gn * u1;
    v4f Eb[R];
    entry_gather<R>(E0, E1, lds, buf, pc, lane, Eb);
    v4f gE[R];
    {
      v4f g[R];
      load_row<CMODE, false, R>(a.g_thr + o0, a.g_thr + o1, C, has1, lane, g);
#pragma unroll
      for (int i = 0; i < R; ++i) {
        gE[i].x = Eb[i].x > kEps ? 0.5f * g[i].x * __builtin_amdgcn_rsqf(Eb[i].x) : 0.f;
        gE[i].y = Eb[i].y > kEps ? 0.5f * g[i].y * __builtin_amdgcn_rsqf(Eb[i].y) : 0.f;
        gE[i].z = Eb[i].z > kEps ? 0.5f * g[i].z * __builtin_amdgcn_rsqf(Eb[i].z) : 0.f;
        gE[i].w = Eb[i].w > kEps ? 0.5f * g[i].w * __builtin_amdgcn_rsqf(Eb[i].w) : 0.f;
      }
    }
    // ---- d L / d G_j = sum_f gE_f Winv[j, f]: first bin u1 of the band below when shared, last bin u0 when shared ----
    const float u1_below = __shfl_up(u1, 1, 64);
    const float bw_first = (lane > 0 && u1_below != 0.f) ? u1_below : rho;
    const float bw_last = (u0 != 0.f) ? u0 : rho;
    const v2f gG = band_sums<R>(gE, lds, buf, pimg, pc, one * bw_first, one * bw_last, one * rho, lane);
    v2f gT, gY;
    gT.x = T.x > quiet ? gG.x : 0.f;
    gT.y = T.y > quiet ? gG.y : 0.f;
    gY.x = Y.x > kEps ? gT.x * T.x / (pp.alpha * Y.x) : 0.f;
    gY.y = Y.y > kEps ? gT.y * T.y / (pp.alpha * Y.y) : 0.f;
    const v2f gA = gY * fac;
    // d fac / d t = fac (-alpha ln 10 / 10) d O / d t
    v2f gt = gY * A * fac * (-pp.alpha * 0.2302585092994046f) * dOdt;
    gt.x = wave_sum(gt.x);
    gt.y = wave_sum(gt.y);
    if (lane == 0) {
      a.g_t_out[t0] = gt.x;
      if (has1) a.g_t_out[t1] = gt.y;
    }
    // ---- d L / d Q_i = sum_j S[i, j] gA_j, d L / d P_i ----
    const v2f gQ = spread<false>(gA, buf, pimg + P::PL_G, lane);
    v2f gP;
    gP.x = Pj.x > kEps ? gQ.x * pp.alpha * Q.x / Pj.x : 0.f;
    gP.y = Pj.y > kEps ? gQ.y * pp.alpha * Q.y / Pj.y : 0.f;
    // ---- d L / d I_f = sum_i W[f, i] gP_i: bins of one band gP_i, the shared bin wl_i gP_i + wf_{i+1} gP_{i+1} ----
    v2f gPn;
    gPn.x = __shfl_down(gP.x, 1, 64);
    gPn.y = __shfl_down(gP.y, 1, 64);
    const float wfirst_above = __shfl_down(wfirst, 1, 64);
    const v2f B1 = gP * wlast + gPn * wfirst_above;
    v4f gI[R];
    entry_gather<R>(gP, B1, lds, buf, pc, lane, gI);
#pragma unroll
    for (int i = 0; i < R; ++i) gI[i] = 2.0f * x[i] * gI[i];
    store_row<CMODE, R>(a.g_X + o0, a.g_X + o1, C, has1, lane, gI);
  }
}

// ------------------------------------------------------------------------------------------------------
// Several short frames per wave: filters_n = 512, 256 (the reference's own test sizes,
// audiocodec/tests/test_mdctransformer.py:23) and 128.  A short frame keeps the wave-level scheme when NFR = 64 / LB
// frames share a wave, each on a group of LB consecutive lanes with eight complex points per lane:
//   filters_n = 512: NFR = 2 frames x 256 points on LB = 32 lanes;   256: NFR = 4 frames x 128 points on 16;
//   128: NFR = 8 frames x 64 points on 8 lanes (pass 2 below is then the identity: 64 = 8 x 8).
// With the frame index in the TOP lane bits (lane = l + LB f) the 8 LB-point FFT is 8 x (8 / NFR) x 8 with exactly the
// two LDS exchanges of the 512-point transform: element e = l + LB r; pass 1 over r (radix 8, twiddle W_{8 LB}^(l k0));
// exchange 1 hands lane (a = k0, m0) the eight values m0 + 8 e1 of row k0, and e1 = e1' + (8 / NFR) f, so pass 2 is NFR
// independent transforms of 8 / NFR points (one per frame, twiddle W_LB^(m0 k1')); exchange 2 and pass 3 (radix 8 over
// e0) are unchanged, and lane l + LB f ends up with the bins l + LB j of frame f: the layout the row loads / stores and
// the fold want, with LB in the place of 64.  Lane reversals stay inside a group: row_mirror DPP for LB = 16, row_mirror
// + two v_permlane16_swap per register pair for LB = 32 -- no LDS.  Table images have the Geo<8> layout, every entry
// replicated to the 64 lanes by the host (index r * 64 + lane as in the one-frame kernels).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dft4(C2& x0, C2& x1, C2& x2, C2& x3) {
  const C2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_mi(csub(x1, x3));
  x0 = cadd(t0, t2);
  x2 = csub(t0, t2);
  x1 = cadd(t1, t3);
  x3 = csub(t1, t3);
}
__device__ __forceinline__ void dft2(C2& x0, C2& x1) {
  const C2 s = cadd(x0, x1), d = csub(x0, x1);
  x0 = s;
  x1 = d;
}

template <int NFR>
__device__ __forceinline__ void fft_wave_multi(C2 (&z)[8], char* buf, gtab_t tab, const v2f (&p1)[8], int lane) {
  constexpr int Q2 = NFR >= 8 ? 1 : 8 / NFR;   // points of pass 2 per frame; 1: the pass is the identity, the exchanges remain
  static_assert(NFR == 2 || NFR == 4 || NFR == 8 || NFR == 16, "frames per wave");
  const int a = lane >> 3, m0 = lane & 7;
  if (NFR == 16) {   // two frames of four points each (see the 64-filter layout below)
    dft4(z[0], z[1], z[2], z[3]);
    dft4(z[4], z[5], z[6], z[7]);
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (k != 4) z[k] = cmul(z[k], p1[k]);
  } else {
    dft8(z);
#pragma unroll
    for (int k = 1; k < 8; ++k) z[k] = cmul(z[k], p1[k]);
  }
  C2 y[8];
  wave_sync();
  {
    char* w1 = buf + 16 * lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) lds_put(w1 + 1152 * k, z[k]);
  }
  wave_sync();
  {
    const char* r1 = buf + 16 * (a * 72 + m0);
#pragma unroll
    for (int r = 0; r < 8; ++r) y[r] = lds_get(r1 + 128 * r);
  }
  if (NFR == 2) {
    dft4(y[0], y[1], y[2], y[3]);
    dft4(y[4], y[5], y[6], y[7]);
  } else if (NFR == 4) {
    dft2(y[0], y[1]);
    dft2(y[2], y[3]);
    dft2(y[4], y[5]);
    dft2(y[6], y[7]);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    z[k] = (k % Q2 == 0) ? y[k] : cmul(y[k], reinterpret_cast<const v2f*>(tab + Geo<8>::I_P2)[k * 8 + m0]);
  wave_sync();
  {
    char* w2 = buf + 16 * (9 * a + m0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) lds_put(w2 + 16 * 72 * kk, z[kk]);
  }
  wave_sync();
  {
    const char* r2 = buf + 144 * lane;
#pragma unroll
    for (int r = 0; r < 8; ++r) y[r] = lds_get(r2 + 16 * r);
  }
  dft8(y);
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = y[j];
}

// lane reversal inside a group of LB lanes (lane -> lane ^ (LB - 1)), both halves of a (c0, c1) pair
template <int LB>
__device__ __forceinline__ v2f rev_group(v2f v) {
  unsigned a = __float_as_uint(v.x), b = __float_as_uint(v.y);
  if (LB == 32) {   // complement lane bit 4: v_permlane16_swap twice, the operands' roles exchanged in between
    const auto u = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const auto w = __builtin_amdgcn_permlane16_swap(u[1], u[0], false, false);
    a = w[0];
    b = w[1];
  }
  constexpr int MIRROR = (LB == 4) ? 0x1b : (LB == 8) ? 0x141 : 0x140;   // quad_perm [3,2,1,0] / row_half_mirror / row_mirror
  return v2f{__uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)a, MIRROR, 0xf, 0xf, false)),
             __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)b, MIRROR, 0xf, 0xf, false))};
}
// out[i] = in[(OFS - i) mod 8] of the mirrored lane of the group
template <int LB, int OFS>
__device__ __forceinline__ void rev_exchange_g(const v2f (&in)[8], v2f (&out)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = rev_group<LB>(in[(OFS - i) & 7]);
}

// one natural-order row of a short frame: lane l of its group moves granules l + LB i, i = 0..7
// CMODE 0: two channels, 16-byte interleaved granules; CMODE 2: one channel, two signals side by side, 8-byte granules
template <int CMODE, int LB>
__device__ __forceinline__ void load_rowm(const float* r0, const float* r1, bool has1, int l, v4f (&v)[8]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const v4f*>(r0)[LB * i + l];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const v2f u = reinterpret_cast<const v2f*>(r0)[LB * i + l];
      v[i] = v4f{u.x, 0.f, u.y, 0.f};
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v2f w = reinterpret_cast<const v2f*>(r1)[LB * i + l];
        v[i].y = w.x;
        v[i].w = w.y;
      }
    }
  }
}
template <int CMODE, int LB>
__device__ __forceinline__ void store_rowm(float* r0, float* r1, bool has1, int l, const v4f (&v)[8]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#if AC_NT_STORE
      __builtin_nontemporal_store(v[i], reinterpret_cast<v4f*>(r0) + LB * i + l);
#else
      reinterpret_cast<v4f*>(r0)[LB * i + l] = v[i];
#endif
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<v2f*>(r0)[LB * i + l] = v2f{v[i].x, v[i].z};
    if (has1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) reinterpret_cast<v2f*>(r1)[LB * i + l] = v2f{v[i].y, v[i].w};
    }
  }
}

// filters_n = 64: a frame is a 32-point transform = 4 x 8.  INPUT rows sit on 8 lanes x 4 registers, two frames in the
// register halves (lane = l8 + 8 g, register 4 fb + i4 holds granule l8 + 8 i4 of frame 2 g + fb); pass 1 is a radix-4
// over i4, the exchanges are the usual ones, pass 3 the radix-8 over l8, and the OUTPUT lands on 4 lanes x 8 registers
// (lane = l4 + 4 f, register j holds bin l4 + 4 j of frame f = 2 g + fb): the LB = 4 form of the layouts above.
template <int CMODE>
__device__ __forceinline__ void load_half(const float* r0, const float* r1, bool has1, int l8, v4f* v) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = reinterpret_cast<const v4f*>(r0)[8 * i + l8];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const v2f u = reinterpret_cast<const v2f*>(r0)[8 * i + l8];
      const v2f w = has1 ? reinterpret_cast<const v2f*>(r1)[8 * i + l8] : v2f{0.f, 0.f};
      v[i] = v4f{u.x, w.x, u.y, w.y};
    }
  }
}
template <int CMODE>
__device__ __forceinline__ void store_half(float* r0, float* r1, bool has1, int l8, const v4f* v) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<v4f*>(r0)[8 * i + l8] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      reinterpret_cast<v2f*>(r0)[8 * i + l8] = v2f{v[i].x, v[i].z};
      if (has1) reinterpret_cast<v2f*>(r1)[8 * i + l8] = v2f{v[i].y, v[i].w};
    }
  }
}

// the same movers for 16-bit PCM rows (x = pcm / 32768 on the way in, clamp(round(32768 x)) on the way out)
template <int CMODE, int LB>
__device__ __forceinline__ void load_rowm(const int16_t* r0, const int16_t* r1, bool has1, int l, v4f (&v)[8]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const s4 p = reinterpret_cast<const s4*>(r0)[LB * i + l];
      v[i] = v4f{Pcm16Fmt::dec(p.x), Pcm16Fmt::dec(p.y), Pcm16Fmt::dec(p.z), Pcm16Fmt::dec(p.w)};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const s2 u = reinterpret_cast<const s2*>(r0)[LB * i + l];
      const s2 w = has1 ? reinterpret_cast<const s2*>(r1)[LB * i + l] : s2{0, 0};
      v[i] = v4f{Pcm16Fmt::dec(u.x), Pcm16Fmt::dec(w.x), Pcm16Fmt::dec(u.y), Pcm16Fmt::dec(w.y)};
    }
  }
}
template <int CMODE, int LB>
__device__ __forceinline__ void store_rowm(int16_t* r0, int16_t* r1, bool has1, int l, const v4f (&v)[8]) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const s2 lo = Pcm16Fmt::enc2(v[i].x, v[i].y), hi = Pcm16Fmt::enc2(v[i].z, v[i].w);
      reinterpret_cast<s4*>(r0)[LB * i + l] = s4{lo.x, lo.y, hi.x, hi.y};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      reinterpret_cast<s2*>(r0)[LB * i + l] = Pcm16Fmt::enc2(v[i].x, v[i].z);
      if (has1) reinterpret_cast<s2*>(r1)[LB * i + l] = Pcm16Fmt::enc2(v[i].y, v[i].w);
    }
  }
}
template <int CMODE>
__device__ __forceinline__ void load_half(const int16_t* r0, const int16_t* r1, bool has1, int l8, v4f* v) {
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const s4 p = reinterpret_cast<const s4*>(r0)[8 * i + l8];
      v[i] = v4f{Pcm16Fmt::dec(p.x), Pcm16Fmt::dec(p.y), Pcm16Fmt::dec(p.z), Pcm16Fmt::dec(p.w)};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const s2 u = reinterpret_cast<const s2*>(r0)[8 * i + l8];
      const s2 w = has1 ? reinterpret_cast<const s2*>(r1)[8 * i + l8] : s2{0, 0};
      v[i] = v4f{Pcm16Fmt::dec(u.x), Pcm16Fmt::dec(w.x), Pcm16Fmt::dec(u.y), Pcm16Fmt::dec(w.y)};
    }
  }
}

struct FwdMArgs {
  const void* x;     // [B, Kin*N, C]  float32, or 16-bit PCM (IOF 1)
  float* X;          // [B, F, N, C]
  const float* prev_block;   // [B, N, C] or null: block -1 of every signal (streaming analysis state)
  float* state_out;          // [B, N, C] or null: receives block Kin-1 (another buffer than prev_block)
  const float* tab;  // analysis image (Geo<8> layout, lane-replicated)
  int Kin, F, C;
  int cpp;           // chunks of NFR consecutive frames per signal pair: ceil(F / NFR)
  int T;             // chunks per wave: workgroup g owns chunks [g NW T, (g+1) NW T), wave w takes g NW T + w + NW t
  long long nsig, ntasks;   // B * C and npairs * cpp
  // fused masking model (PSY kernels): tonality [B, F, 1, C], threshold [B, F, N, C], the image of ac_psy_plan::d_runs
  float* t;
  float* thr;
  const uint32_t* psy_img;
  runs::RunsParams rp;
};

// LDS of the several-frames-per-wave analysis kernels: [NW wave buffers | table image | masking-model image (PSY)].
// The table images in global memory carry every entry replicated to the 64 lanes (index r * 64 + lane); the PSY kernels,
// short of LDS, keep one period of each row only -- TS = max(LB, 8) entries (8: the 64-filter kernels read their input-side
// tables by lane mod 8) -- and, at two frames per wave (filters_n = 512, where the masking model's image is largest), read
// the fold coefficients and the pre-twiddles from the (L2-resident) global image: 53.6 KB per workgroup, three to a CU.
template <int NFR, bool PSY> constexpr int multi_ts() { return PSY ? ((64 / NFR) > 8 ? (64 / NFR) : 8) : 64; }
template <int NFR, bool PSY> constexpr bool multi_pre_global() { return PSY && NFR == 2; }
template <int NFR, bool PSY> constexpr int multi_tab_bytes() {   // (PRE_GLOBAL: fold coefficients and pre-twiddles both stay in global memory)
  return PSY ? (128 + (multi_pre_global<NFR, PSY>() ? 1 : 3) * 16 * multi_ts<NFR, PSY>()) * 4 : Geo<8>::TAB_LDS;
}
// The fused masking model (ac_psy_runs_dev.h) works on FB frames side by side, each in a slot of its own: intensities,
// their partial sums, later G and the threshold entries.  multi_slot(): the largest slot build_runs lays out for
// filters_n = FN (all four levels), a compile-time stride so that a frame's displacement is an immediate of its LDS
// accesses.  The wave lays the spectra of one group of FB frames at a time straight into the slots (the lanes of the later
// groups keep theirs in registers meanwhile) and the intensities overwrite them: FB slots per wave, no staging area.
template <int NFR> constexpr int multi_slot() { return runs::runs_slot_max(1024 / NFR); }
template <int NFR> constexpr int multi_fb() { return (1024 / NFR) >= 512 ? 2 : 4; }
template <int NFR, bool PSY> constexpr int multi_wave_bytes() {
  if (!PSY) return WAVE_LDS;
  const int need = multi_fb<NFR>() * multi_slot<NFR>();
  return need > WAVE_LDS ? need : WAVE_LDS;
}

// analysis: the lanes of group f transform frame NFR c + f of the wave's signal pair (a group whose frame index is past
// the last frame idles); fold and twiddles as in k_fwd_fast with LB in the place of 64
// PSY: the masking model of ac_psy_mid_dev.h on the frames just transformed (the fused encode at filters_n 64 ... 512): the
// wave lays its NFR spectra out in natural order in its LDS buffer (8 KB: NFR frames x N bins x two signals), then walks
// them one frame at a time with all 64 lanes exactly as k_psy_mid does on a row loaded from HBM -- same device function on
// the same values, so X, tonality and threshold equal transform -> k_psy_mid bit for bit, and X is not read back from HBM.
// A frame's intensities overwrite its own slot, and so do its 64 G_j after them.
template <int NFR, int CMODE, int NW, int IOF = 0, bool FOLD4 = false, bool PSY = false>
__global__ __launch_bounds__(NW * 64, AC_WPE) void k_fwd_multi(FwdMArgs a) {
  using pcm_t = typename std::conditional<IOF == 1, int16_t, float>::type;   // (streaming state: float32 only, IOF 0)
  using G = Geo<8>;
  constexpr int LB = 64 / NFR;
  constexpr int TS = multi_ts<NFR, PSY>();                  // entries per register row of the LDS tables
  constexpr bool PRE_GLOBAL = multi_pre_global<NFR, PSY>();
  constexpr int TABB = multi_tab_bytes<NFR, PSY>();
  // (LDS offsets of the tables in floats: the global image's own when it is copied whole)
  constexpr int L_POST = PSY ? 128 : G::I_POST, L_COEF = PSY ? 128 + 16 * TS : G::I_COEF, L_PRE = PSY ? 128 + 32 * TS : G::I_PRE;
  constexpr int WSTR = multi_wave_bytes<NFR, PSY>();            // bytes of LDS per wave
  extern __shared__ __attribute__((aligned(16))) char lds[];   // NW * WSTR + TABB (+ the masking-model image)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (PSY) {
    float* dst = reinterpret_cast<float*>(lds + NW * WSTR);
    for (int i = threadIdx.x; i < 128; i += NW * 64) dst[G::I_P2 + i] = a.tab[G::I_P2 + i];
    for (int i = threadIdx.x; i < 8 * TS; i += NW * 64) {     // one period of every row of POST / COEF / PRE
      const int src = (i / TS) * 64 + (i % TS);
      reinterpret_cast<v2f*>(dst + L_POST)[i] = reinterpret_cast<const v2f*>(a.tab + G::I_POST)[src];
      if (!PRE_GLOBAL) {
        reinterpret_cast<v2f*>(dst + L_COEF)[i] = reinterpret_cast<const v2f*>(a.tab + G::I_COEF)[src];
        reinterpret_cast<v2f*>(dst + L_PRE)[i] = reinterpret_cast<const v2f*>(a.tab + G::I_PRE)[src];
      }
    }
    uint4* pd = reinterpret_cast<uint4*>(lds + NW * WSTR + TABB);
    for (int i = threadIdx.x; i < a.rp.lds_words / 4; i += NW * 64) pd[i] = reinterpret_cast<const uint4*>(a.psy_img)[i];
    __syncthreads();
  } else {
    load_tables<NW, WAVE_LDS, G::I_LDS, 0>(lds, a.tab, nullptr);
  }
  char* buf = lds + wave * WSTR;
  gtab_t tab = reinterpret_cast<const float*>(lds + NW * WSTR);
  const uint32_t* pimg = reinterpret_cast<const uint32_t*>(lds + NW * WSTR + TABB);
  // masking model: constants of band / edge bin `lane`, the lane's per-bin entry offsets (registers: at most four words)
  constexpr int RPM = (1024 / NFR) >= 128 ? (1024 / NFR) / 128 : 1;
  runs::RegIdx<RPM> ridx = {};
  if constexpr (PSY) ridx.load(a.psy_img, a.rp, lane);
  const int tl = lane & (TS - 1);   // column of the lane in a table row
  // the fused kernels whose later groups of frames wait in registers for the model (NFR > FB) fetch the pass-1 twiddles per
  // chunk (L2-resident) instead of holding them across it: 14 registers less where the pressure peaks
  constexpr bool P1_PER_CHUNK = PSY && NFR > multi_fb<NFR>();
  v2f p1[8];
  if (!P1_PER_CHUNK) load_p1<8>(a.tab, lane, p1);
  const int f = lane / LB, l = lane & (LB - 1);
  const int C = a.C;
  const size_t blk = (size_t)(16 * LB) * C;   // floats per block / frame row over all channels
  long long task = (long long)blockIdx.x * NW * a.T + wave;
  long long pair = task / a.cpp;          // (one 64-bit division per wave; the task index then advances without)
  int c = (int)(task - pair * a.cpp);
  for (int t = 0; t < a.T && task < a.ntasks; ++t, task += NW, c += NW) {
    if (P1_PER_CHUNK) load_p1<8>(a.tab, (int)in_loop((uint32_t)lane), p1);
    while (c >= a.cpp) {
      c -= a.cpp;
      ++pair;
    }
    const Pair pq = make_pair<CMODE>(pair, C, a.nsig);
    // (tables read from global memory are fetched per chunk: hoisted out of the loop they would hold 32 registers)
    const uint32_t glane = PRE_GLOBAL ? in_loop((uint32_t)lane) : (uint32_t)lane;
    const int n = c * NFR + f;
    const bool frame_ok = n < a.F;
    const pcm_t* x0 = static_cast<const pcm_t*>(a.x) + row_off(pq.b0, a.Kin, 0, blk, pq.c0);
    const pcm_t* x1 = static_cast<const pcm_t*>(a.x) + row_off(pq.b1, a.Kin, 0, blk, pq.c1);
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    v4f cb[8], pb[8];
    // rows of frame m: the current block m and the block before it (streaming: block -1 is the stored state); a missing
    // block (before the first / after the last) is read from a neighbouring valid row and zeroed afterwards
    auto rows_of = [&](int m, const pcm_t*& c0p, const pcm_t*& c1p, const pcm_t*& p0p, const pcm_t*& p1p, bool& cur_ok,
                       bool& prv_ok) {
      const bool from_state = IOF == 0 && a.prev_block != nullptr && m == 0;
      cur_ok = m < a.Kin;
      prv_ok = (m >= 1 && m <= a.Kin) || from_state;
      const int bc = cur_ok ? m : a.Kin - 1, bp = (m >= 1 && m <= a.Kin) ? m - 1 : 0;
      c0p = x0 + (size_t)bc * blk;
      c1p = x1 + (size_t)bc * blk;
      p0p = x0 + (size_t)bp * blk;
      p1p = x1 + (size_t)bp * blk;
      if constexpr (IOF == 0) {
        if (from_state) {
          p0p = a.prev_block + row_off(pq.b0, 1, 0, blk, pq.c0);
          p1p = a.prev_block + row_off(pq.b1, 1, 0, blk, pq.c1);
        }
      }
    };
    C2 z[8];
    if (NFR == 16) {
      const int l8 = lane & 7, g = lane >> 3;
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        const int m = c * NFR + 2 * g + fb;
        const pcm_t *c0p, *c1p, *p0p, *p1p;
        bool cur_ok, prv_ok;
        rows_of(m, c0p, c1p, p0p, p1p, cur_ok, prv_ok);
        load_half<CMODE>(c0p, c1p, pq.has1, l8, cb + 4 * fb);
        load_half<CMODE>(p0p, p1p, pq.has1, l8, pb + 4 * fb);
        if constexpr (IOF == 0) {
          if (a.state_out && m == a.Kin - 1)   // streaming: the chunk's last block is the next chunk's block -1
            store_half<CMODE>(a.state_out + row_off(pq.b0, 1, 0, blk, pq.c0), a.state_out + row_off(pq.b1, 1, 0, blk, pq.c1),
                              pq.has1, l8, cb + 4 * fb);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          cb[4 * fb + i] = cur_ok ? cb[4 * fb + i] : zero;
          pb[4 * fb + i] = prv_ok ? pb[4 * fb + i] : zero;
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int fb4 = r & 4, r4 = r & 3;
        const v4f& go_p = pb[fb4 + ((1 - r4) & 3)];
        const v4f& go_c = cb[fb4 + ((1 - r4) & 3)];
        const v2f xop = rev_group<8>(v2f{go_p.z, go_p.w}), xoc = rev_group<8>(v2f{go_c.z, go_c.w});
        const v4f& gp = pb[fb4 + ((r4 + 2) & 3)];
        const v4f& gc = cb[fb4 + ((r4 + 2) & 3)];
        const v2f xep = v2f{gp.x, gp.y}, xec = v2f{gc.x, gc.y};
        const v2f ab = (PRE_GLOBAL ? reinterpret_cast<const v2f*>(a.tab + G::I_COEF)[r * 64 + glane] : reinterpret_cast<const v2f*>(tab + L_COEF)[r * TS + tl]);
        const v2f carry = ab.y * xep + ab.x * xop;
        v2f cur;
        if constexpr (FOLD4) {   // a fold block that is not a rotation: its own two coefficients for the current block
          const v2f ce = reinterpret_cast<const v2f*>(a.tab + G::I_COEF2)[r * 64 + lane];
          cur = ce.x * xec + ce.y * xoc;
        } else {
          cur = (r4 < 2) ? (ab.y * xoc - ab.x * xec) : (ab.x * xec - ab.y * xoc);
        }
        const C2 v = (r4 < 2) ? C2{carry, cur} : C2{cur, carry};
        z[r] = cmul(v, PRE_GLOBAL ? reinterpret_cast<const v2f*>(a.tab + G::I_PRE)[r * 64 + glane]
                                   : reinterpret_cast<const v2f*>(tab + L_PRE)[r * TS + tl]);
      }
    } else {
      const pcm_t *c0p, *c1p, *p0p, *p1p;
      bool cur_ok, prv_ok;
      rows_of(n, c0p, c1p, p0p, p1p, cur_ok, prv_ok);
      load_rowm<CMODE, LB>(c0p, c1p, pq.has1, l, cb);
      load_rowm<CMODE, LB>(p0p, p1p, pq.has1, l, pb);
      if constexpr (IOF == 0) {
        if (a.state_out && n == a.Kin - 1)   // streaming: the chunk's last block is the next chunk's block -1
          store_rowm<CMODE, LB>(a.state_out + row_off(pq.b0, 1, 0, blk, pq.c0), a.state_out + row_off(pq.b1, 1, 0, blk, pq.c1),
                                pq.has1, l, cb);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        cb[i] = cur_ok ? cb[i] : zero;
        pb[i] = prv_ok ? pb[i] : zero;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const v4f& go_p = pb[(3 - r) & 7];
        const v4f& go_c = cb[(3 - r) & 7];
        const v2f xop = rev_group<LB>(v2f{go_p.z, go_p.w}), xoc = rev_group<LB>(v2f{go_c.z, go_c.w});
        const v4f& gp = pb[(r + 4) & 7];
        const v4f& gc = cb[(r + 4) & 7];
        const v2f xep = v2f{gp.x, gp.y}, xec = v2f{gc.x, gc.y};
        const v2f ab = (PRE_GLOBAL ? reinterpret_cast<const v2f*>(a.tab + G::I_COEF)[r * 64 + glane] : reinterpret_cast<const v2f*>(tab + L_COEF)[r * TS + tl]);
        const v2f carry = ab.y * xep + ab.x * xop;
        v2f cur;
        if constexpr (FOLD4) {
          const v2f ce = reinterpret_cast<const v2f*>(a.tab + G::I_COEF2)[r * 64 + lane];
          cur = ce.x * xec + ce.y * xoc;
        } else {
          cur = (r < 4) ? (ab.y * xoc - ab.x * xec) : (ab.x * xec - ab.y * xoc);
        }
        const C2 v = (r < 4) ? C2{carry, cur} : C2{cur, carry};
        z[r] = cmul(v, PRE_GLOBAL ? reinterpret_cast<const v2f*>(a.tab + G::I_PRE)[r * 64 + glane]
                                   : reinterpret_cast<const v2f*>(tab + L_PRE)[r * TS + tl]);
      }
    }
    fft_wave_multi<NFR>(z, buf, tab, p1, lane);
    v4f row[8];
    {
      v2f xe[8], xo_in[8], xo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const C2 r = cmul_negim(z[j], reinterpret_cast<const v2f*>(tab + L_POST)[j * TS + tl]);
        xe[j] = r.re;
        xo_in[j] = r.im;
      }
      rev_exchange_g<LB, 7>(xo_in, xo);
#pragma unroll
      for (int i = 0; i < 8; ++i) row[i] = v4f{xe[i].x, xe[i].y, xo[i].x, xo[i].y};
    }
    if (frame_ok) {
      const int nn = n;
      store_rowm<CMODE, LB>(a.X + row_off(pq.b0, a.F, nn, blk, pq.c0), a.X + row_off(pq.b1, a.F, nn, blk, pq.c1), pq.has1, l, row);
    }
    if constexpr (PSY) {
      constexpr int FN = 16 * LB;                      // filters_n
      constexpr int RP = RPM;                          // granule registers per lane when 64 lanes share one frame
      constexpr int FB = multi_fb<NFR>();              // frames side by side (ac_psy_runs_dev.h)
      constexpr int SLOT = multi_slot<NFR>();
      static_assert(NFR % FB == 0, "whole groups");
      const runs::RunsLane lc = runs::load_lane(pimg, lane);   // (per chunk: held across the FFT it would cost seven registers)
#pragma unroll 1
      for (int g0 = 0; g0 < NFR; g0 += FB) {
        if (c * NFR + g0 >= a.F) break;
        wave_sync();   // the FFT's (or the previous group's) accesses of the slots are done
        if (f >= g0 && f < g0 + FB) {
          char* fb = buf + (f - g0) * SLOT + 16 * l;   // granule q = l + LB i of the group's frame at byte 16 q of its slot
#pragma unroll
          for (int i = 0; i < 8; ++i) *reinterpret_cast<v4f*>(fb + 16 * LB * i) = row[i];
        }
        wave_sync();
        const char* ib = buf;
        constexpr int STG = SLOT;
        char* slots = buf;
        v4f xq[FB][RP];
        bool ok[FB];
        size_t o0[FB], o1[FB];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
          const int nn = c * NFR + g0 + fb;
          ok[fb] = nn < a.F;   // (a row past the last frame holds the zero spectrum of an idle group of lanes: computed, not stored)
          o0[fb] = row_off(pq.b0, a.F, ok[fb] ? nn : 0, blk, pq.c0);
          o1[fb] = row_off(pq.b1, a.F, ok[fb] ? nn : 0, blk, pq.c1);
#pragma unroll
          for (int i = 0; i < RP; ++i)
            xq[fb][i] = runs::in_frame<RP>(a.rp, i, lane) ? *reinterpret_cast<const v4f*>(ib + fb * STG + 16 * (64 * i + lane))
                                                          : v4f{0.f, 0.f, 0.f, 0.f};
        }
        v2f t[FB];
        runs::prep_frames<RP, FB, true, true>(xq, a.rp, slots, SLOT, lane, t);
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
          if (ok[fb] && lane == 0) {
            const int nn = c * NFR + g0 + fb;
            a.t[((size_t)pq.b0 * a.F + (size_t)nn) * C + pq.c0] = t[fb].x;
            if (pq.has1) a.t[((size_t)pq.b1 * a.F + (size_t)nn) * C + pq.c1] = t[fb].y;
          }
        wave_sync();
        runs::threshold_frames<RP, FB, FN>(t, a.rp, lc, pimg, slots, SLOT, lane, ridx, [&](int fb, int i, const v4f& th) {
          if (!ok[fb]) return;
          if (CMODE == 0) {
            __builtin_nontemporal_store(th, reinterpret_cast<v4f*>(a.thr + o0[fb]) + 64 * i + lane);
          } else {
            reinterpret_cast<v2f*>(a.thr + o0[fb])[64 * i + lane] = v2f{th.x, th.z};
            if (pq.has1) reinterpret_cast<v2f*>(a.thr + o1[fb])[64 * i + lane] = v2f{th.y, th.w};
          }
        });
        wave_sync();   // the group's reads of its slots are done before the next group's intensities (or the next chunk's FFT) land
      }
    }
  }
}

struct InvMArgs {
  const float* X;    // [B, Kp, N, C]
  void* x;           // [B, nblk*N, C]  float32, or 16-bit PCM (IOF 1)
  const float* tail_in;   // [B, C, N/2] or null: aliased half of the frame before frame 0 (streaming synthesis state)
  float* tail_out;        // [B, C, N/2] or null: receives the aliased half of frame nblk - 1
  const float* tab;  // analysis image; the synthesis image follows at Geo<8>::I_TOTAL floats
  int Kp, nblk, C;
  int cpp;           // chunks of NFR consecutive output blocks per signal pair: ceil(nblk / NFR)
  int spc;           // chunks per strip
  int nstrips;       // strips per signal pair
  long long nsig, ntasks;   // B * C and npairs * nstrips
};

// synthesis: a wave walks a strip of chunks; in a chunk the lanes of group f transform frame n = NFR c + f and finish
// output block n from it and the aliased half of frame n - 1, which the group below has (one shift by LB lanes through
// LDS); group 0 takes it from the chunk before (kept in registers), and a strip that does not start a signal begins with
// the DCT-IV of the chunk before it.  Unfold as in k_inv_fast.
template <int NFR, int CMODE, int NW, int IOF = 0, bool FOLD4 = false>
__global__ __launch_bounds__(NW * 64, AC_WPE) void k_inv_multi(InvMArgs a) {
  using pcm_t = typename std::conditional<IOF == 1, int16_t, float>::type;
  using G = Geo<8>;
  constexpr int LB = 64 / NFR;
  __shared__ __attribute__((aligned(16))) char lds[NW * WAVE_LDS + G::TAB_LDS];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  load_tables<NW, WAVE_LDS, G::I_LDS, 0>(lds, a.tab + G::I_TOTAL, nullptr);
  char* buf = lds + wave * WAVE_LDS;
  gtab_t tab = reinterpret_cast<const float*>(lds + NW * WAVE_LDS);
  v2f p1[8];
  load_p1<8>(a.tab + G::I_TOTAL, lane, p1);
  const long long task = (long long)blockIdx.x * NW + wave;
  if (task >= a.ntasks) return;   // (no workgroup barrier below)
  const int f = lane / LB, l = lane & (LB - 1);
  const int C = a.C;
  const size_t blk = (size_t)(16 * LB) * C;
  const long long pair = task / a.nstrips;
  const int strip = (int)(task - pair * a.nstrips);
  const Pair pq = make_pair<CMODE>(pair, C, a.nsig);
  const int c0 = strip * a.spc, c1 = min(a.cpp, c0 + a.spc);
  const float* X0 = a.X + row_off(pq.b0, a.Kp, 0, blk, pq.c0);
  const float* X1 = a.X + row_off(pq.b1, a.Kp, 0, blk, pq.c1);
  const v4f zero = {0.f, 0.f, 0.f, 0.f};

  // DCT-IV of the frames of chunk c: (now, nxt) per output element k = l + LB j of the group's frame
  auto dct_chunk = [&](int c, v2f (&now)[8], v2f (&nxt)[8]) {
    v4f frm[8];
    v2f xo[8];
    if (NFR == 16) {   // input on 8 lanes x 4 registers, two frames in the register halves
      const int l8 = lane & 7, g = lane >> 3;
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        const int m = c * NFR + 2 * g + fb;
        const bool ok = m >= 0 && m < a.Kp;
        const int fr = ok ? m : 0;
        load_half<CMODE>(X0 + (size_t)fr * blk, X1 + (size_t)fr * blk, pq.has1, l8, frm + 4 * fb);
#pragma unroll
        for (int i = 0; i < 4; ++i) frm[4 * fb + i] = ok ? frm[4 * fb + i] : zero;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const v4f& s = frm[(r & 4) + (3 - (r & 3))];
        xo[r] = rev_group<8>(v2f{s.z, s.w});
      }
    } else {
      const int n = c * NFR + f;
      const bool ok = n >= 0 && n < a.Kp;
      const int fr = ok ? n : 0;
      load_rowm<CMODE, LB>(X0 + (size_t)fr * blk, X1 + (size_t)fr * blk, pq.has1, l, frm);
#pragma unroll
      for (int i = 0; i < 8; ++i) frm[i] = ok ? frm[i] : zero;
      v2f xo_in[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) xo_in[q] = v2f{frm[q].z, frm[q].w};
      rev_exchange_g<LB, 7>(xo_in, xo);
    }
    C2 z[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const C2 v = {v2f{frm[r].x, frm[r].y}, xo[r]};
      z[r] = cmul(v, reinterpret_cast<const v2f*>(tab + G::I_PRE)[r * 64 + lane]);
    }
    fft_wave_multi<NFR>(z, buf, tab, p1, lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const C2 r = cmul_negim(z[j], reinterpret_cast<const v2f*>(tab + G::I_POST)[j * 64 + lane]);
      if (j < 4) {
        now[j] = r.re;
        nxt[j] = r.im;
      } else {
        now[j] = r.im;
        nxt[j] = r.re;
      }
    }
  };
  // every lane receives the value of the lane LB below (the lowest group: of the highest group)
  auto shift_up = [&](const v2f (&v)[8], v2f (&out)[8]) {
    wave_sync();
    {
      char* w = buf + 8 * ((lane + LB) & 63);
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<v2f*>(w + 512 * j) = v[j];
    }
    wave_sync();
    {
      const char* r = buf + 8 * lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) out[j] = *reinterpret_cast<const v2f*>(r + 512 * j);
    }
  };

  constexpr int FHs = 8 * LB;   // outputs per block half
  const size_t ts0 = ((size_t)pq.b0 * C + pq.c0) * FHs, ts1 = ((size_t)pq.b1 * C + pq.c1) * FHs;   // stream state rows
  v2f pend[8];   // in the lanes of group 0: the aliased half of the frame before the next chunk
#pragma unroll
  for (int j = 0; j < 8; ++j) pend[j] = v2f{0.f, 0.f};
  if (c0 == 0 && a.tail_in) {
#pragma unroll
    for (int j2 = 0; j2 < 8; ++j2) {
      const int k = l + LB * j2;
      const int j = (j2 < 4) ? (FHs - 1 - 2 * k) : (2 * k - FHs);
      pend[j2].x = a.tail_in[ts0 + j];
      pend[j2].y = pq.has1 ? a.tail_in[ts1 + j] : 0.f;
    }
  }
  if (c0 > 0) {
    v2f now[8], nxt[8];
    dct_chunk(c0 - 1, now, nxt);
    shift_up(nxt, pend);
  }
  for (int c = c0; c < c1; ++c) {
    v2f now[8], nxt[8], sh[8];
    dct_chunk(c, now, nxt);
    shift_up(nxt, sh);
    const int n = c * NFR + f;
    v4f row[8];
    {
      v2f xe[8], xo_in[8], xo[8];
#pragma unroll
      for (int j2 = 0; j2 < 8; ++j2) {
        const v2f cin = (f == 0) ? pend[j2] : sh[j2];
        const v2f ab = reinterpret_cast<const v2f*>(tab + G::I_COEF)[j2 * 64 + lane];
        const v2f o1 = ab.x * now[j2] + ab.y * cin;
        v2f o2;
        if constexpr (FOLD4) {   // F^-1 of a block that is not a rotation: (s3, s4) on their own
          const v2f cd = reinterpret_cast<const v2f*>(a.tab + G::I_TOTAL + G::I_COEF2)[j2 * 64 + lane];
          o2 = cd.x * now[j2] + cd.y * cin;
        } else {
          o2 = ab.y * now[j2] - ab.x * cin;
        }
        xe[(j2 + 4) & 7] = (j2 < 4) ? o2 : o1;
        xo_in[j2] = (j2 < 4) ? o1 : o2;
      }
      rev_exchange_g<LB, 3>(xo_in, xo);
#pragma unroll
      for (int i = 0; i < 8; ++i) row[i] = v4f{xe[i].x, xe[i].y, xo[i].x, xo[i].y};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) pend[j] = sh[j];
    if (a.tail_out && n == a.nblk - 1) {   // streaming: the last frame's aliased half is the next chunk's state
#pragma unroll
      for (int j2 = 0; j2 < 8; ++j2) {
        const int k = l + LB * j2;
        const int j = (j2 < 4) ? (FHs - 1 - 2 * k) : (2 * k - FHs);
        a.tail_out[ts0 + j] = nxt[j2].x;
        if (pq.has1) a.tail_out[ts1 + j] = nxt[j2].y;
      }
    }
    if (n < a.nblk)
      store_rowm<CMODE, LB>(static_cast<pcm_t*>(a.x) + row_off(pq.b0, a.nblk, n, blk, pq.c0),
                            static_cast<pcm_t*>(a.x) + row_off(pq.b1, a.nblk, n, blk, pq.c1), pq.has1, l, row);
  }
}

// synthesis strips: short, so that the strips in flight cover a nearly contiguous window of memory (HBM rewards that:
// 0.42 ms at 15 blocks per strip, 0.38 ms at 4 with an extra DCT-IV per strip, 0.355-0.365 ms at 3 with the hand-over
// between the waves of a workgroup; B = 256, K = 468)
// (re-measured on well-placed tensors, DESIGN_LOG.md 9a: stereo N = 1024 0.343 ms at 2 blocks per strip against 0.352 at 3;
// N = 2048 0.373 at 3 against 0.396 at 2; mono N = 1024 0.190 at 3 against 0.197 at 2)
int pick_seglen(long long pairs, int frames, int preferred) {
  static const int fixed = [] {
    const char* e = getenv("AC_SEGLEN");   // tuning hook
    return e ? atoi(e) : 0;
  }();
  int s = fixed > 0 ? fixed : preferred;
  if (s > frames) s = frames;
  if (s < 1) s = 1;
  return s;
}

PsyParams psy_params(const ac_psy_plan* p, float drown) {
  PsyParams pp;
  pp.tab = reinterpret_cast<const uint32_t*>(p->d_fast);
  pp.alpha = (float)p->alpha;
  pp.inv_alpha = (float)(1.0 / p->alpha);
  pp.drown = drown;
  return pp;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
template <int R>
static bool build_mdct_fast_R(int N, const FoldCoef& c, std::vector<float>* out) {
  using G = Geo<R>;
  if (N != G::FN) return false;
  const int h = N / 2;
  std::vector<float> t(2 * G::I_TOTAL, 0.f);
  float* tf = t.data();                 // analysis image
  float* ti = t.data() + G::I_TOTAL;    // synthesis image
  const double pi = 3.14159265358979323846;
  auto put2 = [](float* base, int i, double re, double im) {
    base[2 * i] = (float)re;
    base[2 * i + 1] = (float)im;
  };
  // the reference's lower-right quadrant (1 - w[N+j] w[N-1-j]) / w[j] carries ~1e-10 of fp64 cancellation noise
  auto same = [](double x, double y) { return std::fabs(x - y) <= 1e-8; };
  for (int r = 0; r < R; ++r) {
    for (int l = 0; l < 64; ++l) {
      const int i = r * 64 + l;
      const int e = l + 64 * r;                                   // input element of (lane, register)
      const int k = l + 64 * r;                                   // output bin of (lane, register)
      double ang = -pi * (e + 0.25) / N;
      put2(tf + G::I_PRE, i, std::cos(ang), std::sin(ang));
      put2(ti + G::I_PRE, i, std::cos(ang), std::sin(ang));
      ang = -2.0 * pi * (double)(l * r) / (double)G::FH;          // W_{64R}^(lane k0), k0 = r
      put2(tf + G::I_P1, i, std::cos(ang), std::sin(ang));
      put2(ti + G::I_P1, i, std::cos(ang), std::sin(ang));
      if (l < 8 && r < 8) {
        ang = -2.0 * pi * (double)(l * r) / 64.0;                 // [k1 = r][e0 = l]
        put2(tf + G::I_P2, r * 8 + l, std::cos(ang), std::sin(ang));
        put2(ti + G::I_P2, r * 8 + l, std::cos(ang), std::sin(ang));
      }
      ang = -pi * (double)k / N;
      const double sf = 1.0 / (N * std::sqrt(2.0)), si = 2.0 * std::sqrt(2.0);
      put2(tf + G::I_POST, i, std::cos(ang) * sf, std::sin(ang) * sf);
      put2(ti + G::I_POST, i, std::cos(ang) * si, std::sin(ang) * si);
      // analysis fold of element e (see k_fwd_fast): current-frame part cE xe + cO xo, carried part kE xe + kO xo
      double cE, cO, kE, kO;
      if (e < h / 2) {   // samples N/2+2e (even) / N/2-1-2e (odd); current part = v[N-1-2e], carry = v'[2e]
        const int jc = h - 1 - 2 * e, jk = 2 * e;
        cE = c.a2[jc]; cO = c.a1[jc]; kE = c.a4[jk]; kO = c.a3[jk];
        if (!same(cE, -kO) || !same(cO, kE)) return false;        // (-A, B, B, A)
      } else {           // samples 2p (even) / N-1-2p (odd), p = e-N/4: current part = v[2e], carry = v'[N-1-2e]
        const int pidx = e - h / 2;
        const int jc = 2 * pidx, jk = h - 1 - 2 * pidx;
        cE = c.a1[jc]; cO = c.a2[jc]; kE = c.a3[jk]; kO = c.a4[jk];
        if (!same(cE, kO) || !same(cO, -kE)) return false;        // (A, -B, B, A)
      }
      put2(tf + G::I_COEF, i, kO, kE);                             // (A, B)
      // synthesis unfold of output element k (see k_inv_fast): o1 = s1 now + s2 carry, o2 = s3 now + s4 carry
      const int j = (k < h / 2) ? (h - 1 - 2 * k) : (2 * k - h);
      if (!same(c.s3[j], c.s2[j]) || !same(c.s4[j], -c.s1[j])) return false;   // (a, b, b, -a)
      put2(ti + G::I_COEF, i, c.s1[j], c.s2[j]);
    }
  }
  if (out) *out = t;
  return true;
}

// Table images of the several-frames-per-wave kernels (filters_n = 16 LB, LB = 32 or 16 lanes per frame): the Geo<8>
// layout with every entry replicated to the 64 lanes, l = lane mod LB taking the place of the lane.
// *fold4 (may be null): set when some fold block is not a rotation (float32-precomputed constants, mdctransformer.py:218-221
// in float32; the rectangular window, :209-211): the kernels then take the FOLD4 form, which reads the block's other two
// coefficients from I_COEF2.  A caller that passes no fold4 gets false for such tables.
static bool build_mdct_multi(int N, const FoldCoef& c, std::vector<float>* out, bool* fold4) {
  using G = Geo<8>;
  if (N != 512 && N != 256 && N != 128 && N != 64) return false;
  const int LB = N / 16, Q2 = LB >= 8 ? LB / 8 : 1, h = N / 2, FH = 8 * LB;
  const bool two_halves = N == 64;   // input side on 8 lanes x 4 registers (see load_half), output side on LB = 4 lanes
  bool general = false;
  std::vector<float> t(2 * G::I_TOTAL, 0.f);
  float* tf = t.data();
  float* ti = t.data() + G::I_TOTAL;
  const double pi = 3.14159265358979323846;
  auto put2 = [](float* base, int i, double re, double im) {
    base[2 * i] = (float)re;
    base[2 * i + 1] = (float)im;
  };
  auto same = [](double x, double y) { return std::fabs(x - y) <= 1e-8; };
  for (int r = 0; r < 8; ++r) {
    for (int lane = 0; lane < 64; ++lane) {
      const int l = lane % LB, i = r * 64 + lane;
      const int le = two_halves ? lane % 8 : l, k0 = two_halves ? r % 4 : r;   // input side: lane of the group, pass-1 index
      const int e = two_halves ? le + 8 * (r % 4) : l + LB * r, k = l + LB * r;
      double ang = -pi * (e + 0.25) / N;
      put2(tf + G::I_PRE, i, std::cos(ang), std::sin(ang));
      put2(ti + G::I_PRE, i, std::cos(ang), std::sin(ang));
      ang = -2.0 * pi * (double)(le * k0) / (double)FH;             // pass 1: W_{8 LB}^(l k0)
      put2(tf + G::I_P1, i, std::cos(ang), std::sin(ang));
      put2(ti + G::I_P1, i, std::cos(ang), std::sin(ang));
      if (lane < 8) {
        ang = -2.0 * pi * (double)(lane * (r % Q2)) / (double)LB;   // pass 2: [k = r][m0 = lane]  W_LB^(m0 k1'), k1' = k mod Q2
        put2(tf + G::I_P2, r * 8 + lane, std::cos(ang), std::sin(ang));
        put2(ti + G::I_P2, r * 8 + lane, std::cos(ang), std::sin(ang));
      }
      ang = -pi * (double)k / N;
      const double sf = 1.0 / (N * std::sqrt(2.0)), si = 2.0 * std::sqrt(2.0);
      put2(tf + G::I_POST, i, std::cos(ang) * sf, std::sin(ang) * sf);
      put2(ti + G::I_POST, i, std::cos(ang) * si, std::sin(ang) * si);
      double cE, cO, kE, kO;
      if (e < h / 2) {
        const int jc = h - 1 - 2 * e, jk = 2 * e;
        cE = c.a2[jc]; cO = c.a1[jc]; kE = c.a4[jk]; kO = c.a3[jk];
        if (!same(cE, -kO) || !same(cO, kE)) general = true;
      } else {
        const int pidx = e - h / 2;
        const int jc = 2 * pidx, jk = h - 1 - 2 * pidx;
        cE = c.a1[jc]; cO = c.a2[jc]; kE = c.a3[jk]; kO = c.a4[jk];
        if (!same(cE, kO) || !same(cO, -kE)) general = true;
      }
      put2(tf + G::I_COEF, i, kO, kE);
      put2(tf + G::I_COEF2, i, cE, cO);
      const int j = (k < h / 2) ? (h - 1 - 2 * k) : (2 * k - h);
      if (!same(c.s3[j], c.s2[j]) || !same(c.s4[j], -c.s1[j])) general = true;
      put2(ti + G::I_COEF, i, c.s1[j], c.s2[j]);
      put2(ti + G::I_COEF2, i, c.s3[j], c.s4[j]);
    }
  }
  if (general && !fold4) return false;
  if (fold4) *fold4 = general;
  if (out) *out = t;
  return true;
}

// Builds the two table images; false when the size is not served (filters_n 1024 and 2048 are) or the window's fold
// blocks are not rotations (the rectangular "window", mdctransformer.py:209-211), which the two-coefficient fold
// cannot express.
static bool build_mdct_fast(int N, const FoldCoef& c, std::vector<float>* out, bool* fold4) {
  if (fold4) *fold4 = false;
  if (N == Geo<8>::FN) return build_mdct_fast_R<8>(N, c, out);
  if (N == Geo<16>::FN) return build_mdct_fast_R<16>(N, c, out);
  if (N == 512 || N == 256 || N == 128 || N == 64) return build_mdct_multi(N, c, out, fold4);
  return false;
}

// frames per wave of the plan's kernels: 1 (filters_n 1024 / 2048), 2 (512), 4 (256), 8 (128) or 16 (64)
int fast_mdct_frames_per_wave(int N) { return N == 512 ? 2 : N == 256 ? 4 : N == 128 ? 8 : N == 64 ? 16 : 1; }
// what the several-frames-per-wave kernels serve: float32 tensors or 16-bit PCM on the PCM side, mono or stereo, at
// least one block
bool fast_multi_serves(const ac_mdct_plan* p, int C, int iof, int blocks) {
  // (the FOLD4 kernels -- fold blocks that are not rotations -- are instantiated for float32 tensors only)
  return fast_mdct_frames_per_wave(p->N) > 1 && (C == 1 || C == 2) && (iof == 0 || (iof == 1 && !p->fold4)) && blocks >= 1;
}

// the fused encode of the several-frames-per-wave kernels: float32 mono / stereo tensors, rotation fold blocks, and a
// masking model the general-layout wave-level code serves at this size
bool fast_multi_fuses(const ac_mdct_plan* p, const ac_psy_plan* psy, int C, int iof, int blocks) {
  return psy != nullptr && psy->runs && !p->fold4 && iof == 0 && psy->N == p->N && fast_multi_serves(p, C, iof, blocks) &&
         (size_t)AC_WAVES * 16384 + 12800 + (size_t)psy->runs_lay.off_idx * 4 <= 160 * 1024;
}

bool fast_mdct_supported(int N, const FoldCoef& c) {
  bool fold4;
  return build_mdct_fast(N, c, nullptr, &fold4);
}

int fast_mdct_plan_init(ac_mdct_plan* p) {
  std::vector<float> t;
  bool fold4 = false;
  if (!build_mdct_fast(p->N, p->coef, &t, &fold4)) {
    set_error("internal: wave-level kernels not supported for this configuration");
    return AC_EUNSUPPORTED;
  }
  p->fold4 = fold4 ? 1 : 0;
  p->fast_bytes = t.size() * sizeof(float);
  AC_HIP_CHECK(hipMalloc((void**)&p->d_fast, p->fast_bytes));
  AC_HIP_CHECK(hipMemcpy(p->d_fast, t.data(), p->fast_bytes, hipMemcpyHostToDevice));
  return AC_OK;
}

// The wave-level epilogue needs: N = 128 R (1024 or 2048), 64 Bark bands (lane = band), every band a contiguous bin
// range whose interior weights are exactly 1, every bin overlapping at most two (adjacent) bands, a per-band constant
// W_inv on the bins that belong to one band only, and per-half gather lists of at most 24 entries.
template <int R>
static bool build_psy_fast_R(const ac_psy_plan* p, std::vector<uint32_t>* out) {
  using P = PsyGeo<R>;
  const PsyTables& t = p->host;
  const int N = t.N, M = t.M;
  if (N != P::FN || M != 64) return false;
  auto Wf = [&](int f, int j) { return (float)t.W[(size_t)f * M + j]; };
  auto Vf = [&](int j, int f) { return (float)t.W_inv[(size_t)j * N + f]; };
  std::vector<uint32_t> w(P::P_TOTAL_MF, 0u);
  auto putf = [&](int idx, float v) { uint32_t u; memcpy(&u, &v, 4); w[idx] = u; };
  auto band = [](int group, int j, int word) { return P::PL_BAND + 4 * (group * 64 + j) + word; };
  // LDS byte offset of I[f] in the wave buffer while half f / 1024 is staged (granule swizzle of psy_stage)
  auto addrI = [](int f) { const int q = (f & 1023) >> 1; return (uint32_t)(16 * (q ^ ((q >> 4) & 3)) + 8 * (f & 1)); };
  for (int j = 0; j < M; ++j) {
    int f0 = -1, f1 = -1;
    for (int f = 0; f < N; ++f)
      if (Wf(f, j) != 0.f) {
        if (f0 < 0) f0 = f;
        f1 = f;
      }
    if (f0 < 0) return false;
    for (int f = f0; f <= f1; ++f) {
      if (Wf(f, j) == 0.f) return false;
      if (f > f0 && f < f1 && Wf(f, j) != 1.0f) return false;
    }
    for (int h = 0; h < P::NH; ++h) {
      const int lo = 1024 * h, hi = lo + 1023;   // bins of this half
      const bool has0 = f0 >= lo && f0 <= hi, has1 = f1 > f0 && f1 >= lo && f1 <= hi;
      w[band(h, j, 0)] = (has0 ? addrI(f0) : (uint32_t)ZERO_OFF) | ((has1 ? addrI(f1) : (uint32_t)ZERO_OFF) << 16);
      putf(band(h, j, 1), has0 ? Wf(f0, j) : 0.f);
      putf(band(h, j, 2), has1 ? Wf(f1, j) : 0.f);
      putf(band(h, j, 3), (float)t.quiet[j]);
      // interior bins f0+1 .. f1-1 (weight 1) inside this half: single bins up to an 8-aligned boundary, whole
      // chunks, single bins
      std::vector<uint32_t> lst;
      const int a = std::max(f0 + 1, lo), b = std::min(f1 - 1, hi);
      for (int f = a; f <= b;) {
        if ((f & 7) == 0 && f + 7 <= b) {
          lst.push_back((uint32_t)(S8_OFF + 8 * ((f & 1023) >> 3)));
          f += 8;
        } else {
          lst.push_back(addrI(f));
          f += 1;
        }
      }
      if ((int)lst.size() > 2 * P::PL_HALF) return false;
      lst.resize(2 * P::PL_HALF, (uint32_t)ZERO_OFF);
      for (int hlf = 0; hlf < P::PL_HALF; ++hlf)
        w[P::PL_LST + (h * P::PL_HALF + hlf) * 64 + j] = lst[2 * hlf] | (lst[2 * hlf + 1] << 16);
    }
    putf(band(P::NH, j, 0), t.beta[j]);
  }
  // bins -> entries; nnz pattern of W and W_inv is identical (same overlap)
  std::vector<int> entry(N, -1);
  std::vector<float> rho(M, 0.f), u0(M, 0.f), u1(M, 0.f);
  std::vector<bool> have_rho(M, false);
  for (int f = 0; f < N; ++f) {
    int cnt = 0, jf = -1;
    for (int j = 0; j < M; ++j)
      if (Vf(j, f) != 0.f) {
        if (cnt == 0) jf = j;
        ++cnt;
      }
    if (cnt == 1) {
      const float v = Vf(jf, f);
      if (!have_rho[jf]) {
        rho[jf] = v;
        have_rho[jf] = true;
      } else if (std::fabs(v - rho[jf]) > 1e-6f * rho[jf]) {
        return false;
      }
      entry[f] = 2 * jf;
    } else if (cnt == 2 && jf + 1 < M && Vf(jf + 1, f) != 0.f) {
      if (u0[jf] != 0.f || u1[jf] != 0.f) return false;   // one shared bin per band boundary
      u0[jf] = Vf(jf, f);
      u1[jf] = Vf(jf + 1, f);
      entry[f] = 2 * jf + 1;
    } else {
      return false;
    }
  }
  for (int j = 0; j < M; ++j) {
    putf(band(P::NH, j, 1), rho[j]);
    putf(band(P::NH, j, 2), u0[j]);
    putf(band(P::NH, j, 3), u1[j]);
  }
  // threshold entry e lives at byte 8 e of the wave buffer; word i of lane l = offsets of bins 2q, 2q+1, q = 64 i + l
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < R; ++i) {
      const int q = 64 * i + l;
      const uint32_t e0 = 8u * (uint32_t)entry[2 * q], e1 = 8u * (uint32_t)entry[2 * q + 1];
      w[P::PL_IDX + 4 * ((i >> 2) * 64 + l) + (i & 3)] = e0 | (e1 << 16);
    }
  for (int i = 0; i < 128; ++i) putf(P::PL_G + i, (float)t.g[i]);
  // bf16 tiles for spread_mfma: copy c, entry y = rev[y - c], rev[m] = g[128 - m] (m = 1 .. 127); hi parts, then lo parts
  {
    auto bf16_rne = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    auto bf16_val = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    uint16_t* tb = reinterpret_cast<uint16_t*>(w.data() + P::PL_MF);
    for (int c = 0; c < 4; ++c)
      for (int y = 0; y < 132; ++y) {
        const int m = y - c;
        if (m < 1 || m > 127) continue;
        const float v = (float)t.g[128 - m];
        const uint16_t hi = bf16_rne(v);
        tb[(c * MF_COPY_STRIDE) / 2 + y] = hi;
        tb[(MF_TAB_BYTES + c * MF_COPY_STRIDE) / 2 + y] = bf16_rne(v - bf16_val(hi));
      }
  }
  if (out) *out = w;
  return true;
}

static bool build_psy_fast(const ac_psy_plan* p, std::vector<uint32_t>* out) {
  if (p->host.N == PsyGeo<8>::FN) return build_psy_fast_R<8>(p, out);
  if (p->host.N == PsyGeo<16>::FN) return build_psy_fast_R<16>(p, out);
  return false;
}

bool fast_psy_supported(const ac_psy_plan* p) { return build_psy_fast(p, nullptr); }

int fast_psy_plan_init(ac_psy_plan* p) {
  std::vector<uint32_t> w;
  if (!build_psy_fast(p, &w)) {
    set_error("internal: fused epilogue not supported for this configuration");
    return AC_EUNSUPPORTED;
  }
  p->fast_bytes = w.size() * sizeof(uint32_t);
  AC_HIP_CHECK(hipMalloc((void**)&p->d_fast, p->fast_bytes));
  AC_HIP_CHECK(hipMemcpy(p->d_fast, w.data(), p->fast_bytes, hipMemcpyHostToDevice));
  return AC_OK;
}

static int grid_for(long long ntasks, int nw, unsigned* grid) {
  const long long g = (ntasks + nw - 1) / nw;
  if (g > 2147483647ll) {
    set_error("problem too large for one launch (%lld workgroups)", g);
    return AC_EINVAL;
  }
  *grid = (unsigned)g;
  return AC_OK;
}

// workgroups of a persistent launch: enough to fill every CU at the kernel's occupancy, a multiple of 8 (XCDs)
static unsigned persistent_grid(int cus, int wg_per_cu, long long ntasks, int nw) {
  long long g = (long long)cus * wg_per_cu;
  const long long need = (ntasks + nw - 1) / nw;
  if (g > need) g = need;
  g = (g + 7) / 8 * 8;
  return (unsigned)g;
}

template <int NFR>
static int launch_fwd_multi_N(const FwdMArgs& a, int iof, bool fold4, bool psy, int C, unsigned grid, hipStream_t s) {
  const dim3 blk(AC_WAVES * 64);
  auto go = [&](auto kernel, size_t lds) -> int {
    if (lds > 64 * 1024)
      AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), blk, lds, s, a);
    return AC_OK;
  };
  if (psy) {   // fused masking model: float32 tensors, rotation fold blocks (the caller checked)
    const size_t lds = (size_t)AC_WAVES * multi_wave_bytes<NFR, true>() + multi_tab_bytes<NFR, true>() + (size_t)a.rp.lds_words * 4;
    if (C == 2) return go(k_fwd_multi<NFR, 0, AC_WAVES, 0, false, true>, lds);
    return go(k_fwd_multi<NFR, 2, AC_WAVES, 0, false, true>, lds);
  }
  const size_t lds = (size_t)AC_WAVES * WAVE_LDS + Geo<8>::TAB_LDS;
  if (fold4) {
    if (C == 2) return go(k_fwd_multi<NFR, 0, AC_WAVES, 0, true>, lds);
    return go(k_fwd_multi<NFR, 2, AC_WAVES, 0, true>, lds);
  }
  if (iof == 1) {
    if (C == 2) return go(k_fwd_multi<NFR, 0, AC_WAVES, 1>, lds);
    return go(k_fwd_multi<NFR, 2, AC_WAVES, 1>, lds);
  }
  if (C == 2) return go(k_fwd_multi<NFR, 0, AC_WAVES>, lds);
  return go(k_fwd_multi<NFR, 2, AC_WAVES>, lds);
}
// psy (may be null): the plan of the masking model for general band layouts, fused into the launch (fast_multi_fuses())
static int launch_fwd_multi(const ac_mdct_plan* p, const ac_psy_plan* psy, const void* x, int iof, float* X, float* t, float* thr,
                            float drown, const float* prev_block, float* state_out, int B, int Kin, int F, int C, hipStream_t s) {
  const int nfr = fast_mdct_frames_per_wave(p->N);
  FwdMArgs a;
  a.x = x;
  a.X = X;
  a.prev_block = prev_block;
  a.state_out = state_out;
  a.tab = p->d_fast;
  a.Kin = Kin;
  a.F = F;
  a.C = C;
  a.cpp = (F + nfr - 1) / nfr;
  a.nsig = (long long)B * C;
  a.t = t;
  a.thr = thr;
  a.psy_img = psy ? psy->d_runs : nullptr;
  if (psy) a.rp = runs_params(psy, drown, false);
  else a.rp = runs::RunsParams{};
  const long long npairs = (C == 2) ? (long long)B : (a.nsig + 1) / 2;
  a.ntasks = npairs * a.cpp;
  // chunks per wave: the table copy (and the masking model's image) is paid once per workgroup
  static const int tper = [] { const char* e = getenv("AC_FWD_T"); return e ? atoi(e) : 4; }();
  static const int tper_psy = [] { const char* e = getenv("AC_FWD_T_PSY"); return e ? atoi(e) : 4; }();   // (2 ... 8 measure alike)
  int T = psy ? (tper_psy > 0 ? tper_psy : 4) : (tper > 0 ? tper : 4);
  while (T > 1 && a.ntasks < (long long)AC_WAVES * T * p->cus * 2) T >>= 1;
  a.T = T;
  unsigned grid;
  int st = grid_for(a.ntasks, AC_WAVES * T, &grid);
  if (st) return st;
  const bool f4 = p->fold4 != 0;
  if (nfr == 2) st = launch_fwd_multi_N<2>(a, iof, f4, psy != nullptr, C, grid, s);
  else if (nfr == 4) st = launch_fwd_multi_N<4>(a, iof, f4, psy != nullptr, C, grid, s);
  else if (nfr == 8) st = launch_fwd_multi_N<8>(a, iof, f4, psy != nullptr, C, grid, s);
  else st = launch_fwd_multi_N<16>(a, iof, f4, psy != nullptr, C, grid, s);
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

template <int NFR>
static void launch_inv_multi_N(const InvMArgs& a, int iof, bool fold4, int C, unsigned grid, hipStream_t s) {
  const dim3 blk(AC_WAVES * 64);
  if (fold4) {
    if (C == 2) hipLaunchKernelGGL((k_inv_multi<NFR, 0, AC_WAVES, 0, true>), dim3(grid), blk, 0, s, a);
    else hipLaunchKernelGGL((k_inv_multi<NFR, 2, AC_WAVES, 0, true>), dim3(grid), blk, 0, s, a);
    return;
  }
  if (iof == 1) {
    if (C == 2) hipLaunchKernelGGL((k_inv_multi<NFR, 0, AC_WAVES, 1>), dim3(grid), blk, 0, s, a);
    else hipLaunchKernelGGL((k_inv_multi<NFR, 2, AC_WAVES, 1>), dim3(grid), blk, 0, s, a);
    return;
  }
  if (C == 2) hipLaunchKernelGGL((k_inv_multi<NFR, 0, AC_WAVES>), dim3(grid), blk, 0, s, a);
  else hipLaunchKernelGGL((k_inv_multi<NFR, 2, AC_WAVES>), dim3(grid), blk, 0, s, a);
}
static int launch_inv_multi(const ac_mdct_plan* p, const float* X, void* x, int iof, const float* tail_in, float* tail_out,
                            int B, int Kp, int nblk, int C, hipStream_t s) {
  const int nfr = fast_mdct_frames_per_wave(p->N);
  InvMArgs a;
  a.X = X;
  a.x = x;
  a.tail_in = tail_in;
  a.tail_out = tail_out;
  a.tab = p->d_fast;
  a.Kp = Kp;
  a.nblk = nblk;
  a.C = C;
  a.cpp = (nblk + nfr - 1) / nfr;
  a.nsig = (long long)B * C;
  const long long npairs = (C == 2) ? (long long)B : (a.nsig + 1) / 2;
  // chunks per strip: every strip but a signal's first pays one more DCT-IV pass for the frame before it
  static const int spc_env = [] { const char* e = getenv("AC_SPC"); return e ? atoi(e) : 0; }();
  int spc = spc_env > 0 ? spc_env : 8;
  while (spc > 1 && npairs * ((a.cpp + spc - 1) / spc) < (long long)AC_WAVES * p->cus * 2) spc >>= 1;
  a.spc = spc;
  a.nstrips = (a.cpp + spc - 1) / spc;
  a.ntasks = npairs * a.nstrips;
  unsigned grid;
  const int st = grid_for(a.ntasks, AC_WAVES, &grid);
  if (st) return st;
  const bool f4 = p->fold4 != 0;
  if (nfr == 2) launch_inv_multi_N<2>(a, iof, f4, C, grid, s);
  else if (nfr == 4) launch_inv_multi_N<4>(a, iof, f4, C, grid, s);
  else if (nfr == 8) launch_inv_multi_N<8>(a, iof, f4, C, grid, s);
  else launch_inv_multi_N<16>(a, iof, f4, C, grid, s);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

// the element-wise epilogues (EPI kernels) serve stereo float32 input at 8 points per lane (filters_n = 1024)
bool fast_epilogue_supported(const ac_mdct_plan* p, const ac_psy_plan* psy, int iof, int C) {
  // (the plain-bf16 matrix-core form of the spreading product would spill three registers here: it takes the un-fused path)
  return psy != nullptr && p->N == Geo<8>::FN && iof == 0 && C == 2 && psy->spread != 1;
}

template <int R, int IOF>
static void launch_fwd_R(const FwdArgs& a, bool psy, int spread, int C, unsigned grid, hipStream_t s) {
  constexpr bool PCM16 = IOF == 1;
  if constexpr (R == 8 && IOF == 0) {
    if (psy && C == 2 && (a.noisy || a.dbn)) {
      const dim3 blk(AC_WAVES_PSY * 64);
      if (spread == 2) hipLaunchKernelGGL((k_fwd_fast<R, 0, true, AC_WAVES_PSY, 0, 2, true>), dim3(grid), blk, 0, s, a);
      else hipLaunchKernelGGL((k_fwd_fast<R, 0, true, AC_WAVES_PSY, 0, 0, true>), dim3(grid), blk, 0, s, a);
      return;
    }
  }
  if constexpr (IOF == 2) {
    // bfloat16 tensors: stereo and mono kernels (other channel counts are served by the LDS-FFT tier, see ac_api.hip)
    if (psy) {
      const dim3 blk(AC_WAVES_PSY * 64);
      if (C == 2) hipLaunchKernelGGL((k_fwd_fast<R, 0, true, AC_WAVES_PSY, 2>), dim3(grid), blk, 0, s, a);
      else if constexpr (R == 8) hipLaunchKernelGGL((k_fwd_fast<R, 2, true, AC_WAVES_PSY, 2>), dim3(grid), blk, 0, s, a);
      return;
    }
    const dim3 blk(AC_WAVES * 64);
    if (C == 2) hipLaunchKernelGGL((k_fwd_fast<R, 0, false, AC_WAVES, 2>), dim3(grid), blk, 0, s, a);
    else hipLaunchKernelGGL((k_fwd_fast<R, 2, false, AC_WAVES, 2>), dim3(grid), blk, 0, s, a);
    return;
  } else {
  if (psy) {
    const dim3 blk(AC_WAVES_PSY * 64);
    // the matrix-core forms of the spreading product serve the stereo kernels (float32 or 16-bit PCM input); the others
    // keep the f32 product
    if (C == 2 && spread == 1) hipLaunchKernelGGL((k_fwd_fast<R, 0, true, AC_WAVES_PSY, IOF, 1>), dim3(grid), blk, 0, s, a);
    else if (C == 2 && spread == 2) hipLaunchKernelGGL((k_fwd_fast<R, 0, true, AC_WAVES_PSY, IOF, 2>), dim3(grid), blk, 0, s, a);
    else if (C == 2) hipLaunchKernelGGL((k_fwd_fast<R, 0, true, AC_WAVES_PSY, IOF>), dim3(grid), blk, 0, s, a);
    else if (C == 1) {
      // (R = 16: the caller runs transform and masking model as two launches, see encode_fused in ac_api.hip)
      if constexpr (R == 8) {
        if (spread == 1) hipLaunchKernelGGL((k_fwd_fast<R, 2, true, AC_WAVES_PSY, IOF, 1>), dim3(grid), blk, 0, s, a);
        else if (spread == 2) hipLaunchKernelGGL((k_fwd_fast<R, 2, true, AC_WAVES_PSY, IOF, 2>), dim3(grid), blk, 0, s, a);
        else hipLaunchKernelGGL((k_fwd_fast<R, 2, true, AC_WAVES_PSY, IOF>), dim3(grid), blk, 0, s, a);
      }
    } else {
      if constexpr (R == 8 || !PCM16) hipLaunchKernelGGL((k_fwd_fast<R, 1, true, AC_WAVES_PSY, IOF>), dim3(grid), blk, 0, s, a);
    }
    return;
  }
  const dim3 blk(AC_WAVES * 64);
  if (C == 2) hipLaunchKernelGGL((k_fwd_fast<R, 0, false, AC_WAVES, IOF>), dim3(grid), blk, 0, s, a);
  else if (C == 1) hipLaunchKernelGGL((k_fwd_fast<R, 2, false, AC_WAVES, IOF>), dim3(grid), blk, 0, s, a);
  else hipLaunchKernelGGL((k_fwd_fast<R, 1, false, AC_WAVES, IOF>), dim3(grid), blk, 0, s, a);
  }
}

// arguments and grid of the one-frame-per-wave analysis kernels (filters_n 1024 / 2048)
static int prep_fwd_fast(const ac_mdct_plan* p, const ac_psy_plan* psy, const void* x, int iof, float* X, float* t,
                         float* thr, float drown, const float* prev_block, int B, int Kin, int F, int C, float* state_out,
                         float* noisy, float* dbn, uint64_t seed, FwdArgs& a, unsigned& grid) {
  // combinations no kernel is instantiated for (ac_api.hip routes them elsewhere; refuse rather than launch nothing)
  if ((iof == 2 && C > 2) || (psy && p->N == Geo<16>::FN && (C == 1 || (iof == 1 && C > 2)))) {
    set_error("internal: no wave-level analysis kernel for filters_n = %d, %d channels, io format %d%s", p->N, C, iof,
              psy ? ", fused masking model" : "");
    return AC_EUNSUPPORTED;
  }
  a.x = x;
  a.X = X;
  a.t = t;
  a.thr = thr;
  a.prev_block = prev_block;
  a.state_out = state_out;
  a.noisy = noisy;
  a.dbn = dbn;
  a.noise_key = mix64(seed);
  if ((noisy || dbn) && !fast_epilogue_supported(p, psy, iof, C)) {
    set_error("internal: no fused element-wise epilogue for this configuration");
    return AC_EUNSUPPORTED;
  }
  a.tab = p->d_fast;
  if (psy) a.psy = psy_params(psy, drown);
  else a.psy = PsyParams{nullptr, 0.f, 0.f, 0.f};
  a.B = B;
  a.Kin = Kin;
  a.F = F;
  a.C = C;
  a.nsig = (long long)B * C;
  a.npairs = (C == 2) ? (long long)B : (a.nsig + 1) / 2;
  a.nframes = a.npairs * F;
  {
    const double ang = -3.14159265358979323846 / (4.0 * p->N), sc = (double)p->N * 1.4142135623730951;   // 1 / (1 / (N sqrt 2))
    a.pre_re = (float)(std::cos(ang) * sc);
    a.pre_im = (float)(std::sin(ang) * sc);
  }
  // tuning hooks (read once): AC_XCD=1 groups consecutive workgroups per XCD; AC_FWD_T = frames per wave, workgroups
  // dispatched in order (default 4: measured 0.603 ms against 0.615 ms for persistent waves, B = 256, K = 468 -- fresh
  // workgroups keep the window of memory in flight contiguous); AC_FWD_T=0 = persistent waves, AC_WG_PER_CU per CU
  static const int xcd = [] { const char* e = getenv("AC_XCD"); return e ? atoi(e) : 0; }();
  static const int wgcu = [] { const char* e = getenv("AC_WG_PER_CU"); return e ? atoi(e) : 3; }();
  static const int tper = [] { const char* e = getenv("AC_FWD_T"); return e ? atoi(e) : 4; }();
  a.xcd = xcd;
  const int nw = psy ? AC_WAVES_PSY : AC_WAVES;
  // small launches (a streaming chunk of one clip): fewer frames per wave, so that the frames spread over the chip
  // instead of queueing behind each other in a few workgroups
  int tper_eff = tper;
  while (tper_eff > 1 && a.nframes < (long long)nw * tper_eff * p->cus * 2) tper_eff >>= 1;
  a.T = tper_eff;
  if (tper > 0) {
    const long long per = (long long)nw * tper_eff;
    long long g = (a.nframes + per - 1) / per;
    g = (g + 7) / 8 * 8;
    if (g > 2147483647ll) {
      set_error("problem too large for one launch (%lld workgroups)", g);
      return AC_EINVAL;
    }
    grid = (unsigned)g;
  } else {
    grid = persistent_grid(p->cus, wgcu, a.nframes, nw);
  }
  return AC_OK;
}

int launch_fwd_fast(const ac_mdct_plan* p, const ac_psy_plan* psy, const void* x, int iof, float* X, float* t,
                    float* thr, float drown, const float* prev_block, int B, int Kin, int F, int C, hipStream_t s,
                    float* state_out, float* noisy, float* dbn, uint64_t seed) {
  if (B <= 0 || C <= 0 || F <= 0) return AC_OK;
  if (fast_mdct_frames_per_wave(p->N) > 1) {
    if (noisy || dbn || !fast_multi_serves(p, C, iof, Kin) || (psy && !(fast_multi_fuses(p, psy, C, iof, Kin) && t && thr))) {
      set_error("internal: no wave-level analysis kernel for filters_n = %d, %d channels, io format %d here", p->N, C, iof);
      return AC_EUNSUPPORTED;
    }
    return launch_fwd_multi(p, psy, x, iof, X, psy ? t : nullptr, psy ? thr : nullptr, drown, prev_block, state_out, B, Kin, F, C, s);
  }
  FwdArgs a;
  unsigned grid;
  const int st = prep_fwd_fast(p, psy, x, iof, X, t, thr, drown, prev_block, B, Kin, F, C, state_out, noisy, dbn, seed, a, grid);
  if (st) return st;
  const int spread = psy ? psy->spread : 0;
  if (p->N == Geo<8>::FN) {
    if (iof == 2) launch_fwd_R<8, 2>(a, psy != nullptr, spread, C, grid, s);
    else if (iof == 1) launch_fwd_R<8, 1>(a, psy != nullptr, spread, C, grid, s);
    else launch_fwd_R<8, 0>(a, psy != nullptr, spread, C, grid, s);
  }
#ifndef AC_NO_R16
  else if (iof == 2) launch_fwd_R<16, 2>(a, psy != nullptr, spread, C, grid, s);
  else if (iof == 1) launch_fwd_R<16, 1>(a, psy != nullptr, spread, C, grid, s);
  else launch_fwd_R<16, 0>(a, psy != nullptr, spread, C, grid, s);
#endif
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

template <int R, int IOF>
static void launch_inv_R(const InvArgs& a, int C, unsigned grid, hipStream_t s) {
  const dim3 blk(AC_WAVES * 64);
  if (C == 2) hipLaunchKernelGGL((k_inv_fast<R, 0, AC_WAVES, IOF>), dim3(grid), blk, 0, s, a);
  else if (C == 1) hipLaunchKernelGGL((k_inv_fast<R, 2, AC_WAVES, IOF>), dim3(grid), blk, 0, s, a);
  else if constexpr (IOF != 2) hipLaunchKernelGGL((k_inv_fast<R, 1, AC_WAVES, IOF>), dim3(grid), blk, 0, s, a);
}

// arguments and grid of the one-frame-per-wave synthesis kernels (filters_n 1024 / 2048)
static int prep_inv_fast(const ac_mdct_plan* p, const float* X, void* x, int iof, const float* tail_in, float* tail_out,
                         int B, int Kp, int nblk, int C, InvArgs& a, unsigned& grid) {
  if (iof == 2 && C > 2) {
    set_error("internal: no wave-level synthesis kernel for bfloat16 tensors with %d channels", C);
    return AC_EUNSUPPORTED;
  }
  a.X = X;
  a.x = x;
  a.tail_in = tail_in;
  a.tail_out = tail_out;
  a.tab = p->d_fast;
  a.B = B;
  a.Kp = Kp;
  a.nblk = nblk;
  a.C = C;
  a.nsig = (long long)B * C;
  a.npairs = (C == 2) ? (long long)B : (a.nsig + 1) / 2;
  a.seglen = pick_seglen(a.npairs, nblk, (p->N == Geo<8>::FN && C == 2) ? 2 : 3);
  // small launches (a streaming chunk of one clip): one block per strip, so that the blocks spread over the chip
  if (a.npairs * ((nblk + a.seglen - 1) / a.seglen) < (long long)AC_WAVES * p->cus * 2) a.seglen = pick_seglen(a.npairs, nblk, 1);
  a.nseg = (nblk + a.seglen - 1) / a.seglen;
  a.ntasks = a.npairs * a.nseg;
#ifndef AC_INV_REV_DEFAULT
#define AC_INV_REV_DEFAULT 0
#endif
  static const int inv_rev = [] { const char* e = getenv("AC_INV_REV"); return e ? atoi(e) : AC_INV_REV_DEFAULT; }();
  a.rev = inv_rev;
  // one strip per wave, workgroups dispatched in order (persistent waves drift apart and measured slower here)
  const long long need = (a.ntasks + AC_WAVES - 1) / AC_WAVES;
  if (need > 2147483647ll) {
    set_error("problem too large for one launch (%lld workgroups)", need);
    return AC_EINVAL;
  }
  grid = (unsigned)need;
  return AC_OK;
}

int launch_inv_fast(const ac_mdct_plan* p, const float* X, void* x, int iof, const float* tail_in, float* tail_out,
                    int B, int Kp, int nblk, int C, hipStream_t s) {
  if (B <= 0 || C <= 0 || nblk <= 0) return AC_OK;
  if (fast_mdct_frames_per_wave(p->N) > 1) {
    if (!fast_multi_serves(p, C, iof, Kp)) {
      set_error("internal: no wave-level synthesis kernel for filters_n = %d, %d channels, io format %d here", p->N, C, iof);
      return AC_EUNSUPPORTED;
    }
    return launch_inv_multi(p, X, x, iof, tail_in, tail_out, B, Kp, nblk, C, s);
  }
  InvArgs a;
  unsigned grid;
  const int st = prep_inv_fast(p, X, x, iof, tail_in, tail_out, B, Kp, nblk, C, a, grid);
  if (st) return st;
  if (p->N == Geo<8>::FN) {
    if (iof == 2) launch_inv_R<8, 2>(a, C, grid, s);
    else if (iof == 1) launch_inv_R<8, 1>(a, C, grid, s);
    else launch_inv_R<8, 0>(a, C, grid, s);
  }
#ifndef AC_NO_R16
  else if (iof == 2) launch_inv_R<16, 2>(a, C, grid, s);
  else if (iof == 1) launch_inv_R<16, 1>(a, C, grid, s);
  else launch_inv_R<16, 0>(a, C, grid, s);
#endif
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

// ---- streaming duplex: analysis of one chunk and synthesis of another in one launch (k_duplex_fast) ----
// Served: float32, mono / stereo, filters_n 1024 / 2048, without the masking model or (1024, stereo) with the fused
// one in its f32 / split-bf16 spreading forms -- and only launches that leave the chip partly idle on their own: up to
// 256 wave tasks per CU over both halves (64 stereo streams in chunks of 256 blocks: 116 us per chunk against 124 for
// the chain; one stream: 11.6 against 18 -- a chunk's two dependent launches are latency there).  Beyond that each launch
// fills the chip by itself.  AC_DUPLEX_MAX_TASKS overrides the limit (tuning hook).
bool fast_duplex_serves(const ac_mdct_plan* p, const ac_psy_plan* psy, int B, int C, int k_fwd, int k_inv) {
  static const int off = [] { const char* e = getenv("AC_NO_DUPLEX"); return e ? atoi(e) : 0; }();   // tuning hook
  if (off || fast_mdct_frames_per_wave(p->N) != 1 || (C != 1 && C != 2) || k_fwd < 1 || k_inv < 1) return false;
  if (psy && !(p->N == Geo<8>::FN && C == 2 && psy->fast && psy->spread != 1)) return false;
  const long long pairs = (C == 2) ? B : (B + 1) / 2;
  static const long long max_tasks = [] { const char* e = getenv("AC_DUPLEX_MAX_TASKS"); return e ? atoll(e) : 0ll; }();
  // (the mono instantiations are compiled for two waves per SIMD: beyond latency-bound sizes the chain's kernels win)
  return pairs * ((long long)k_fwd + k_inv) <= (max_tasks > 0 ? max_tasks : (long long)p->cus * (C == 2 ? 256 : 8));
}

template <int R>
static void launch_duplex_R(const FwdArgs& fa, const InvArgs& ia, unsigned gf, unsigned gi, bool psy, int spread, int C,
                            hipStream_t s) {
  const dim3 grid(gf + gi), blk(AC_WAVES * 64);
  static_assert(AC_WAVES == AC_WAVES_PSY, "one workgroup shape for both halves");
  if constexpr (R == 8) {
    if (psy) {
      if (spread == 2) hipLaunchKernelGGL((k_duplex_fast<8, 0, true, AC_WAVES, 2>), grid, blk, 0, s, fa, ia, (int)gf);
      else hipLaunchKernelGGL((k_duplex_fast<8, 0, true, AC_WAVES, 0>), grid, blk, 0, s, fa, ia, (int)gf);
      return;
    }
  }
  if (C == 2) hipLaunchKernelGGL((k_duplex_fast<R, 0, false, AC_WAVES, 0>), grid, blk, 0, s, fa, ia, (int)gf);
  else hipLaunchKernelGGL((k_duplex_fast<R, 2, false, AC_WAVES, 0>), grid, blk, 0, s, fa, ia, (int)gf);
}

int launch_duplex_fast(const ac_mdct_plan* p, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr,
                       float drown, const float* prev_block, float* state_out, int k_fwd, const float* X_inv, float* x_inv,
                       const float* tail_in, float* tail_out, int k_inv, int B, int C, hipStream_t s) {
  if (!fast_duplex_serves(p, psy, B, C, k_fwd, k_inv)) {
    set_error("internal: the streaming duplex kernel does not serve this configuration");
    return AC_EUNSUPPORTED;
  }
  FwdArgs fa;
  InvArgs ia;
  unsigned gf, gi;
  int st = prep_fwd_fast(p, psy, x, 0, X, psy ? t : nullptr, psy ? thr : nullptr, drown, prev_block, B, k_fwd, k_fwd, C,
                         state_out, nullptr, nullptr, 0, fa, gf);
  if (!st) st = prep_inv_fast(p, X_inv, x_inv, 0, tail_in, tail_out, B, k_inv, k_inv, C, ia, gi);
  if (st) return st;
  const int spread = psy ? psy->spread : 0;
  if (p->N == Geo<8>::FN) launch_duplex_R<8>(fa, ia, gf, gi, psy != nullptr, spread, C, s);
#ifndef AC_NO_R16
  else launch_duplex_R<16>(fa, ia, gf, gi, false, 0, C, s);
#endif
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

template <int R, int CMODE, int SPREAD>
static void launch_psy_thr(const PsyArgs& a, bool want_t, unsigned grid, hipStream_t s) {
  const dim3 blk(AC_WAVES * 64);
  if (want_t) hipLaunchKernelGGL((k_psy_fast<R, CMODE, true, true, AC_WAVES, SPREAD>), dim3(grid), blk, 0, s, a);
  else hipLaunchKernelGGL((k_psy_fast<R, CMODE, false, true, AC_WAVES, SPREAD>), dim3(grid), blk, 0, s, a);
}
template <int R, int CMODE>
static void launch_psy_bf16(const PsyArgs& a, bool want_t, bool want_thr, unsigned grid, hipStream_t s) {
  const dim3 blk(AC_WAVES * 64);
  if (want_t && !want_thr) hipLaunchKernelGGL((k_psy_fast<R, CMODE, true, false, AC_WAVES, 0, 2>), dim3(grid), blk, 0, s, a);
  else if (!want_t && want_thr) hipLaunchKernelGGL((k_psy_fast<R, CMODE, false, true, AC_WAVES, 0, 2>), dim3(grid), blk, 0, s, a);
  else if (want_t && want_thr) hipLaunchKernelGGL((k_psy_fast<R, CMODE, true, true, AC_WAVES, 0, 2>), dim3(grid), blk, 0, s, a);
}
template <int R, int CMODE>
static void launch_psy_R(const PsyArgs& a, bool want_t, bool want_thr, int spread, unsigned grid, hipStream_t s) {
  const dim3 blk(AC_WAVES * 64);
  if (CMODE == 0 && want_thr && spread == 1) return launch_psy_thr<R, 0, 1>(a, want_t, grid, s);
  if (CMODE == 0 && want_thr && spread == 2) return launch_psy_thr<R, 0, 2>(a, want_t, grid, s);
  if constexpr (CMODE == 2) {   // mono: the same two forms (two clips ride in the pair)
    if (want_thr && spread == 1) return launch_psy_thr<R, 2, 1>(a, want_t, grid, s);
    if (want_thr && spread == 2) return launch_psy_thr<R, 2, 2>(a, want_t, grid, s);
  }
  if (want_t && !want_thr) hipLaunchKernelGGL((k_psy_fast<R, CMODE, true, false, AC_WAVES>), dim3(grid), blk, 0, s, a);
  else if (!want_t && want_thr) hipLaunchKernelGGL((k_psy_fast<R, CMODE, false, true, AC_WAVES>), dim3(grid), blk, 0, s, a);
  else if (want_t && want_thr) hipLaunchKernelGGL((k_psy_fast<R, CMODE, true, true, AC_WAVES>), dim3(grid), blk, 0, s, a);
}

int launch_psy_fast(const ac_psy_plan* p, const float* X, const float* t_in, float* t_out, float* thr, float drown,
                    int B, int F, int C, hipStream_t s, int iof) {
  if (B <= 0 || C <= 0 || F <= 0) return AC_OK;
  PsyArgs a;
  a.X = X;
  a.t_in = t_in;
  a.t_out = t_out;
  a.thr = thr;
  a.psy = psy_params(p, drown);
  a.C = C;
  a.F = F;
  a.nsig = (long long)B * C;
  a.ntasks = ((C == 2) ? (long long)B : (a.nsig + 1) / 2) * F;
  unsigned grid;
  int st = grid_for(a.ntasks, AC_WAVES, &grid);
  if (st) return st;
  const bool want_t = (t_out != nullptr), want_thr = (thr != nullptr);
  const int cmode = (C == 2) ? 0 : (C == 1) ? 2 : 1;
  if (iof == 2 && C > 2) {
    set_error("internal: no wave-level masking-model kernel for bfloat16 tensors with %d channels", C);
    return AC_EUNSUPPORTED;
  }
  if (iof == 2) {   // bfloat16 tensors: stereo and mono
    if (p->N == PsyGeo<8>::FN) {
      if (cmode == 0) launch_psy_bf16<8, 0>(a, want_t, want_thr, grid, s);
      else launch_psy_bf16<8, 2>(a, want_t, want_thr, grid, s);
    } else {
#ifndef AC_NO_R16
      if (cmode == 0) launch_psy_bf16<16, 0>(a, want_t, want_thr, grid, s);
      else launch_psy_bf16<16, 2>(a, want_t, want_thr, grid, s);
#endif
    }
    AC_HIP_CHECK(hipGetLastError());
    return AC_OK;
  }
  if (p->N == PsyGeo<8>::FN) {
    if (cmode == 0) launch_psy_R<8, 0>(a, want_t, want_thr, p->spread, grid, s);
    else if (cmode == 2) launch_psy_R<8, 2>(a, want_t, want_thr, p->spread, grid, s);
    else launch_psy_R<8, 1>(a, want_t, want_thr, p->spread, grid, s);
  } else {
#ifndef AC_NO_R16
    if (cmode == 0) launch_psy_R<16, 0>(a, want_t, want_thr, p->spread, grid, s);
    else if (cmode == 2) launch_psy_R<16, 2>(a, want_t, want_thr, p->spread, grid, s);
    else launch_psy_R<16, 1>(a, want_t, want_thr, p->spread, grid, s);
#endif
  }
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

template <int R, int CMODE>
static void launch_psy_bwd_R(const PsyBwdArgs& a, bool tonality, unsigned grid, hipStream_t s) {
  const dim3 blk(AC_WAVES * 64);
  if (tonality) hipLaunchKernelGGL((k_psy_bwd_fast<R, CMODE, true, AC_WAVES>), dim3(grid), blk, 0, s, a);
  else hipLaunchKernelGGL((k_psy_bwd_fast<R, CMODE, false, AC_WAVES>), dim3(grid), blk, 0, s, a);
}

// tonality backward when g_thr is null (g_t -> g_X, optionally accumulated), else threshold backward
int launch_psy_bwd_fast(const ac_psy_plan* p, const float* X, const float* t, float drown, const float* g_thr,
                        const float* g_t, float* g_X, float* g_t_out, int accumulate, int B, int F, int C,
                        hipStream_t s) {
  if (B <= 0 || C <= 0 || F <= 0) return AC_OK;
  PsyBwdArgs a;
  a.X = X;
  a.t = t;
  a.g_thr = g_thr;
  a.g_t = g_t;
  a.g_X = g_X;
  a.g_t_out = g_t_out;
  a.psy = psy_params(p, drown);
  a.C = C;
  a.F = F;
  a.accumulate = accumulate;
  a.nsig = (long long)B * C;
  a.ntasks = ((C == 2) ? (long long)B : (a.nsig + 1) / 2) * F;
  unsigned grid;
  int st = grid_for(a.ntasks, AC_WAVES, &grid);
  if (st) return st;
  const bool tonality = (g_thr == nullptr);
  const int cmode = (C == 2) ? 0 : (C == 1) ? 2 : 1;
  if (p->N == PsyGeo<8>::FN) {
    if (cmode == 0) launch_psy_bwd_R<8, 0>(a, tonality, grid, s);
    else if (cmode == 2) launch_psy_bwd_R<8, 2>(a, tonality, grid, s);
    else launch_psy_bwd_R<8, 1>(a, tonality, grid, s);
  } else {
#ifndef AC_NO_R16
    if (cmode == 0) launch_psy_bwd_R<16, 0>(a, tonality, grid, s);
    else if (cmode == 2) launch_psy_bwd_R<16, 2>(a, tonality, grid, s);
    else launch_psy_bwd_R<16, 1>(a, tonality, grid, s);
#endif
  }
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

}  // namespace ac
