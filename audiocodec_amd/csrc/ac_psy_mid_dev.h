// Device code of the wave-level masking model for general band layouts, shared by the stand-alone kernel (k_psy_mid,
// ac_psy_mid.hip) and the fused encode of the several-frames-per-wave MDCT kernels (k_fwd_multi, ac_fast.hip): ONE
// definition of the per-frame arithmetic, so that the fused and the un-fused encode agree bit for bit.  gfx950 only.
//
// One frame (both signals of a pair) per call, all 64 lanes: the frame's granules xq[i] = (X[2q], X[2q+1]) x (s0, s1),
// q = 64 i + lane, are in registers; tonality comes from DPP wave sums, the intensities go through an LDS image, lane
// j < M owns Bark band j and walks its run of (bin, weight) entries (W "by band": psychoacoustic.py:301-315), the band x
// band spreading product runs against the lane's column of S held in registers (psychoacoustic.py:205-207, tonality
// offset pulled out of the sum: SURVEY App. A.3; S is Toeplitz, S[i][j] = g[M - i + j] (:223-228), so the image holds
// the 2 M prototype values instead of M x M), and every bin gathers its <= wi_w (band, weight) entries of W_inv from a
// fixed-width table stored entry-major (psychoacoustic.py:317-331).
#pragma once
#include "ac_internal.h"

namespace ac {
namespace mid {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr float kEps = 1e-14f;   // _INTENSITY_EPS, psychoacoustic.py:56

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int iv = __builtin_bit_cast(int, v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {   // as in ac_fast.hip: xor butterflies per row of 16, two row broadcasts
  v = dpp_add<0xB1, 0xf>(v);
  v = dpp_add<0x4E, 0xf>(v);
  v = dpp_add<0x141, 0xf>(v);
  v = dpp_add<0x140, 0xf>(v);
  v = dpp_add<0x142, 0xa>(v);
  v = dpp_add<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ v2f log2v(v2f x) { return v2f{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)}; }
__device__ __forceinline__ v2f exp2v(v2f x) { return v2f{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ v2f maxv(v2f a, float b) { return v2f{fmaxf(a.x, b), fmaxf(a.y, b)}; }

// acc += q * s.x (lo) / q * s.y (hi) on both halves of q: one packed multiply-add with the scalar taken from one half of
// a register pair through op_sel (the compiler would duplicate the scalar into a pair of its own: 128 registers for S)
__device__ __forceinline__ void pk_fma_lo(v2f& acc, v2f q, v2f s) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(q), "v"(s));
}
__device__ __forceinline__ void pk_fma_hi(v2f& acc, v2f q, v2f s) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(q), "v"(s));
}


// image layout (32-bit words), one copy in LDS per workgroup:
//   off_S:    gp[0 .. 128): S[i][j] = gp[64 + j - i] = g[M - i + j], 0 where |j - i| >= M   (psychoacoustic.py:223-228)
//   off_band: per band j: {first entry, count | first bin << 16, quiet, beta}   4 words
//   off_wbe:  W by band: the weights of the band's bins first bin, first bin + 1, ...  (a band's bins are contiguous: both
//             loads of a step have addresses that depend on nothing loaded before)
//   off_wi:   entry-major: [e < wi_w][bin f] {byte offset of G[band] in the wave's G area, weight}  (weight 0 pads)
struct MidParams {
  int img_words;
  int N, M;
  int wi_w;              // entries per bin in the fixed-width W_inv table
  int off_S, off_band, off_wbe, off_wi;   // word offsets inside the image
  float alpha, inv_alpha, drown;
  float inv_n;           // 1 / N
};

// the lane's column of the spreading matrix, S[i][lane] = gp[64 + lane - i], as 32 register pairs (S[2 i][lane],
// S[2 i + 1][lane]) (the product below always runs over 64 rows: rows beyond the M bands meet Q_i = 0, lanes beyond them
// are not read)
__device__ __forceinline__ void load_scol(const uint32_t* img, const MidParams& a, int lane, v2f (&Scol)[32]) {
  const float* gp = reinterpret_cast<const float*>(img + a.off_S) + 64 + lane;
#pragma unroll
  for (int i = 0; i < 32; ++i) Scol[i] = v2f{gp[-2 * i], gp[-2 * i - 1]};
}

// granules past the frame (q >= N / 2) read as zero and are never stored
template <int R>
__device__ __forceinline__ bool in_frame(const MidParams& a, int i, int lane) { return R * 128 == a.N || 64 * i + lane < (a.N >> 1); }

// tonality of the frame (psychoacoustic.py:102-120; the arithmetic of psy_stage in ac_fast.hip)
template <int R>
__device__ __forceinline__ v2f tonality_frame(const v4f (&xq)[R], const MidParams& a, int lane) {
  v2f slog = {0.f, 0.f}, ssq = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < R; ++i) {
    v4f I = xq[i] * xq[i];
    asm("" : "+v"(I));   // the squares stay rounded products (see psy_stage)
    const v2f ie = v2f{I.x, I.y}, io = v2f{I.z, I.w};
    ssq += ie + io;
    const v2f lg = log2v(maxv(ie, kEps) * maxv(io, kEps));
    slog += in_frame<R>(a, i, lane) ? lg : v2f{0.f, 0.f};
  }
  slog.x = wave_sum(slog.x);
  slog.y = wave_sum(slog.y);
  ssq.x = wave_sum(ssq.x);
  ssq.y = wave_sum(ssq.y);
  const v2f am = ssq * a.inv_n + kEps;
  const v2f sfm = 3.0102999566398120f * (slog * a.inv_n - log2v(am));
  const v2f tt = sfm * (-1.0f / 60.0f);
  return v2f{fminf(tt.x, 1.0f), fminf(tt.y, 1.0f)};
}

// masking threshold of the frame: th[i] = thresholds of granule 64 i + lane.
// ibuf: 8 N bytes of LDS for the frame's intensities (bin f at byte 8 f: (s0, s1)) -- may be the very bytes the caller read
// xq from; Qb, Gb: 64 v2f each; img: the LDS copy of the image.  The caller orders its earlier accesses to these buffers
// before the call (wave_sync) and may reuse them after the return.
template <int R>
__device__ __forceinline__ void threshold_frame(const v4f (&xq)[R], v2f t, const MidParams& a, const uint32_t* img, char* ibuf,
                                                v2f* Qb, v2f* Gb, const v2f (&Scol)[32], int lane, v4f (&th)[R]) {
  const int half = a.N >> 1, M = a.M;
  // intensities in natural order: bin f at byte 8 f (c0, c1)
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (in_frame<R>(a, i, lane)) *reinterpret_cast<v4f*>(ibuf + 16 * (64 * i + lane)) = xq[i] * xq[i];
  wave_sync();
  const uint32_t* band = img + a.off_band;
  float quiet = 0.f, beta = 0.f;
  if (lane < M) {   // P_j = sum_f I_f W[f, j]  (:312-313)
    const uint4 bw = reinterpret_cast<const uint4*>(band)[lane];
    quiet = __uint_as_float(bw.z);
    beta = __uint_as_float(bw.w);
    const float* wt = reinterpret_cast<const float*>(img + a.off_wbe) + bw.x;
    const v2f* Ib = reinterpret_cast<const v2f*>(ibuf) + (bw.y >> 16);
    v2f P0 = {0.f, 0.f}, P1 = {0.f, 0.f};
    const int cnt = (int)(bw.y & 0xffffu);
    int k = 0;
#pragma unroll 4
    for (; k + 1 < cnt; k += 2) {
      P0 += Ib[k] * wt[k];
      P1 += Ib[k + 1] * wt[k + 1];
    }
    if (k < cnt) P0 += Ib[k] * wt[k];
    Qb[lane] = exp2v(a.alpha * log2v(maxv(P0 + P1, kEps)));   // max(eps, P)^alpha  (:206)
  } else {
    Qb[lane] = v2f{0.f, 0.f};
  }
  wave_sync();
  {   // sum_i Q_i S[i, j], offset factor outside the sum  (:185-208); even rows into acc0, odd rows into acc1
    v2f acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 64; i += 2) {
      const v4f qq = *reinterpret_cast<const v4f*>(Qb + i);   // Q_i, Q_{i+1} (broadcast read)
      pk_fma_lo(acc0, v2f{qq.x, qq.y}, Scol[i / 2]);
      pk_fma_hi(acc1, v2f{qq.z, qq.w}, Scol[i / 2]);
      if ((i & 14) == 14) __builtin_amdgcn_sched_barrier(0);   // eight broadcast reads in flight at a time, not thirty-two
    }
    const v2f offset = (1.0f - a.drown) * (t * beta + 9.0f * t + 5.5f);
    const v2f fac = exp2v(offset * (-a.alpha * 0.33219280948873623f));                 // 10^(-alpha O / 10)
    const v2f T = exp2v(a.inv_alpha * log2v(maxv(fac * (acc0 + acc1), kEps)));          // (:208)
    Gb[lane] = maxv(T, quiet);                                                          // (:144)
  }
  wave_sync();
  // thr_f = sqrt(max(eps, sum_j G_j W_inv[j, f]))  (:330-331)
  const uint4* wi = reinterpret_cast<const uint4*>(img + a.off_wi);   // [e][granule q]: the entries of bins 2 q, 2 q + 1
  const int W = a.wi_w;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int q = 64 * i + lane;
    v2f s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
    for (int e = 0; e < W && in_frame<R>(a, i, lane); ++e) {
      const uint4 en = wi[(size_t)e * half + q];
      s0 += *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(Gb) + en.x) * __uint_as_float(en.y);
      s1 += *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(Gb) + en.z) * __uint_as_float(en.w);
    }
    s0 = maxv(s0, kEps);
    s1 = maxv(s1, kEps);
    th[i] = v4f{__builtin_amdgcn_sqrtf(s0.x), __builtin_amdgcn_sqrtf(s0.y), __builtin_amdgcn_sqrtf(s1.x), __builtin_amdgcn_sqrtf(s1.y)};
  }
}

}  // namespace mid

// the launch-time parameters of a plan's image (ac_psy_mid.hip)
mid::MidParams mid_params(const ac_psy_plan* p, float drown);

}  // namespace ac
