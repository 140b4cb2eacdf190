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

// ---- band x band product with the Toeplitz spreading matrix on the matrix cores ------------------------------------------
//   out_j = sum_i Q_i S[i, j],  S[i, j] = g[64 - i + j]   (psychoacoustic.py:205-207, 223-228)
// as 32 v_mfma_f32_4x4x4_16b_bf16 on split-bfloat16 operands (the scheme of spread_mfma in ac_fast.hip, BASELINE
// configs[3]): the instruction's 16 blocks are the 16 column tiles of S, so band j of the result lands in lane j; step s
// contracts bands 4 s .. 4 s + 3; its A tile (rows 0, 1 = the two signals' Q rounded to bfloat16, rows 2, 3 = the
// remainders Q - hi) is built by a quad-local transpose in the four lanes of block s and broadcast with cbsz = 4 /
// abid = s; S = hi + lo likewise (two B tables), four partial products in float32 accumulators: ~16 mantissa bits, the
// thresholds within 1e-5 of an all-float32 product.  The B tiles of a lane -- 16 steps x (hi, lo) x four consecutive
// entries of the reversed prototype -- do not depend on the frame: a step's tiles are read from LDS once for the FB
// frames of a group (16 KB of LDS reads per group; the float32 form read every frame's Q back 32 times: 32 KB per frame).
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v2b __attribute__((ext_vector_type(2)));
constexpr int MF_COPY_STRIDE = 288;               // bytes between the four shifted copies of the reversed bf16 prototype
constexpr int MF_TAB_BYTES = 4 * MF_COPY_STRIDE;  // one table (hi or lo parts)
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{a, b}, v2b));
}
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {   // lane K of every quad to the whole quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);
}
// the A tile of one frame: rows 0, 1 = (Q.x, Q.y) rounded to bfloat16, rows 2, 3 = the remainders, of bands 4 s .. 4 s + 3 in
// the four lanes of block s
__device__ __forceinline__ v4s spread_a_tile(v2f Q, int lane) {
  const uint32_t whi = pk_bf16(Q.x, Q.y);
  const float hx = __uint_as_float(whi << 16), hy = __uint_as_float(whi & 0xffff0000u);
  const uint32_t wlo = pk_bf16(Q.x - hx, Q.y - hy);
  // quad-local 4 x 4 transpose of 16-bit values: lane 4 s + i gets row i of bands 4 s .. 4 s + 3 (every lane evaluates all
  // broadcasts before the select: a DPP read needs its source lane active)
  const int i = lane & 3;
  const bool lo_row = i >= 2;
  uint32_t c0 = quad_bcast<0>(whi), c1 = quad_bcast<1>(whi), c2 = quad_bcast<2>(whi), c3 = quad_bcast<3>(whi);
  const uint32_t l0 = quad_bcast<0>(wlo), l1 = quad_bcast<1>(wlo), l2 = quad_bcast<2>(wlo), l3 = quad_bcast<3>(wlo);
  c0 = lo_row ? l0 : c0, c1 = lo_row ? l1 : c1, c2 = lo_row ? l2 : c2, c3 = lo_row ? l3 : c3;
  const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;   // the signal's half of each word
  const uint2 au = {__builtin_amdgcn_perm(c1, c0, sel), __builtin_amdgcn_perm(c3, c2, sel)};
  return __builtin_bit_cast(v4s, au);
}
// steps 0 .. S of the product for FB frames: a step's two B tiles (hi, lo parts of S) are read from LDS once and meet every
// frame's A tile
template <int S, int FB>
struct TileSteps {
  static __device__ __forceinline__ void run(const v4s (&a)[FB], const char* b, v4f (&d0)[FB], v4f (&d1)[FB]) {
    TileSteps<S - 1, FB>::run(a, b, d0, d1);
    const v4s bh = *reinterpret_cast<const v4s*>(b + 8 * S), bl = *reinterpret_cast<const v4s*>(b + MF_TAB_BYTES + 8 * S);
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      d0[fb] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[fb], bh, d0[fb], 4, S, 0);
      d1[fb] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[fb], bl, d1[fb], 4, S, 0);
    }
  }
};
template <int FB>
struct TileSteps<-1, FB> {
  static __device__ __forceinline__ void run(const v4s (&)[FB], const char*, v4f (&)[FB], v4f (&)[FB]) {}
};
// out[fb]_j = sum_i Q[fb]_i S[i, j] in lane j; mf = the LDS copy of the tiles (hi table, then lo table)
template <int FB>
__device__ __forceinline__ void spread_tiles(const v2f (&Q)[FB], const char* mf, int lane, v2f (&out)[FB]) {
  v4s a[FB];
  v4f d0[FB], d1[FB];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    a[fb] = spread_a_tile(Q[fb], lane);
    d0[fb] = v4f{0.f, 0.f, 0.f, 0.f};
    d1[fb] = v4f{0.f, 0.f, 0.f, 0.f};
  }
  TileSteps<15, FB>::run(a, mf + (lane & 3) * MF_COPY_STRIDE + 8 * (16 - (lane >> 2)), d0, d1);
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const v4f d = d0[fb] + d1[fb];
    out[fb] = v2f{d.x + d.z, d.y + d.w};
  }
}

// image layout (32-bit words), one copy in LDS per workgroup:
//   off_S:    the bfloat16 tiles of the spreading matrix for spread_tiles: four shifted copies of the reversed prototype
//             rev[m] = gp[128 - m], gp[64 + d] = g[M + d] (0 where |d| >= M: S[i][j] = gp[64 + j - i], psychoacoustic.py:
//             223-228), hi parts (MF_TAB_BYTES) then lo parts
//   off_band: per band j: {first entry, count | first bin << 16, quiet, beta}   4 words
//   off_wbe:  W by band: the weights of the band's bins first bin, first bin + 1, ..., padded with zeros to a multiple of
//             four and 16-byte aligned (a band's bins are contiguous: the loads of a step have addresses that depend on
//             nothing loaded before)
//   off_wi:   entry-major: [e < wi_w][bin f] {byte offset of G[band] in the wave's G area, weight}  (weight 0 pads)
struct MidParams {
  int img_words;         // the whole image
  int lds_words;         // its part that is copied to LDS: all of it, or everything before the W_inv entries (filter_bands_n > 1024)
  int N, M;
  int wi_w;              // entries per bin in the fixed-width W_inv table
  int off_S, off_band, off_wbe, off_wi;   // word offsets inside the image
  float alpha, inv_alpha, drown;
  float inv_n;           // 1 / N
};

// granules past the frame (q >= N / 2) read as zero and are never stored
template <int R>
__device__ __forceinline__ bool in_frame(const MidParams& a, int i, int lane) { return R * 128 == a.N || 64 * i + lane < (a.N >> 1); }

// The per-frame arithmetic below runs on FB frames side by side: the frames are independent, so their dependent chains
// (DPP wave sums, LDS round trips, transcendentals, the 16-deep MFMA accumulations) interleave -- a single frame leaves a
// wave waiting on its own latencies most of the time -- and the constant tables (band descriptors, weights, the W_inv
// entries) are read once for all of them.  Every frame sees exactly the operations, in the order, it would see alone:
// results do not depend on FB or on which frames share a group.

// tonality of FB frames (psychoacoustic.py:102-120; the arithmetic of psy_stage in ac_fast.hip)
template <int R, int FB>
__device__ __forceinline__ void tonality_frames(const v4f (&xq)[FB][R], const MidParams& a, int lane, v2f (&t)[FB]) {
  v2f slog[FB], ssq[FB];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    slog[fb] = v2f{0.f, 0.f};
    ssq[fb] = v2f{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; ++i) {
      v4f I = xq[fb][i] * xq[fb][i];
      asm("" : "+v"(I));   // the squares stay rounded products (see psy_stage)
      const v2f ie = v2f{I.x, I.y}, io = v2f{I.z, I.w};
      ssq[fb] += ie + io;
      const v2f lg = log2v(maxv(ie, kEps) * maxv(io, kEps));
      slog[fb] += in_frame<R>(a, i, lane) ? lg : v2f{0.f, 0.f};
    }
  }
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    slog[fb].x = wave_sum(slog[fb].x);
    slog[fb].y = wave_sum(slog[fb].y);
    ssq[fb].x = wave_sum(ssq[fb].x);
    ssq[fb].y = wave_sum(ssq[fb].y);
  }
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const v2f am = ssq[fb] * a.inv_n + kEps;
    const v2f sfm = 3.0102999566398120f * (slog[fb] * a.inv_n - log2v(am));
    const v2f tt = sfm * (-1.0f / 60.0f);
    // (a NaN or an infinite intensity: NaN, as the reference's tf.maximum / tf.minimum propagate it)
    t[fb] = v2f{(ssq[fb].x - ssq[fb].x == 0.0f) ? fminf(tt.x, 1.0f) : __builtin_nanf(""), (ssq[fb].y - ssq[fb].y == 0.0f) ? fminf(tt.y, 1.0f) : __builtin_nanf("")};
  }
}

// masking thresholds of FB frames; emit(fb, i, th) receives the thresholds of granule 64 i + lane of frame fb as they come
// out (lanes with in_frame(i) only).
// ibuf + fb * istride: 8 N bytes of LDS per frame for its intensities (bin f at byte 8 f: (s0, s1)) -- may be the very bytes
// the caller read xq from; the frame's G_j (64 v2f) later takes the first 512 bytes of the same slot.  img: the LDS copy of
// the image.  The caller orders its earlier accesses to the slots before the call (wave_sync) and its later ones after it.
// wi: the W_inv entry table -- img + a.off_wi when the whole image sits in LDS, or the plan's copy in global memory (frames
// above 512 bins: the table alone would take 16 N bytes of LDS; its reads are coalesced 16-byte rows and L2-resident).
template <int R, int FB, class EMIT>
__device__ __forceinline__ void threshold_frames(const v4f (&xq)[FB][R], const v2f (&t)[FB], const MidParams& a, const uint32_t* img,
                                                 const uint4* wi, char* ibuf, int istride, int lane, EMIT emit) {
  const int half = a.N >> 1, M = a.M;
  // intensities in natural order: bin f at byte 8 f (c0, c1)
#pragma unroll
  for (int fb = 0; fb < FB; ++fb)
#pragma unroll
    for (int i = 0; i < R; ++i)
      if (in_frame<R>(a, i, lane)) *reinterpret_cast<v4f*>(ibuf + fb * istride + 16 * (64 * i + lane)) = xq[fb][i] * xq[fb][i];
  wave_sync();
  const uint32_t* band = img + a.off_band;
  float quiet = 0.f, beta = 0.f;
  v2f Q[FB];   // (lanes beyond the M bands keep 0: rows of S they would meet do not exist)
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) Q[fb] = v2f{0.f, 0.f};
  if (lane < M) {   // P_j = sum_f I_f W[f, j]  (:312-313)
    const uint4 bw = reinterpret_cast<const uint4*>(band)[lane];
    quiet = __uint_as_float(bw.z);
    beta = __uint_as_float(bw.w);
    const float* wt = reinterpret_cast<const float*>(img + a.off_wbe) + bw.x;
    const char* Ib = ibuf + 8 * (bw.y >> 16);
    v2f P0[FB], P1[FB];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      P0[fb] = v2f{0.f, 0.f};
      P1[fb] = v2f{0.f, 0.f};
    }
    // four bins per step (the host pads a band's weights to a multiple of four with zeros, 16-byte aligned; a padded step
    // re-reads the band's last bin, so nothing outside the frame's intensities is touched): even bins into P0, odd into P1
    const int cnt = (int)(bw.y & 0xffffu), last = cnt - 1;
    for (int k = 0; k < cnt; k += 4) {
      const v4f w4 = *reinterpret_cast<const v4f*>(wt + k);
      const int k1 = min(k + 1, last), k2 = min(k + 2, last), k3 = min(k + 3, last);
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const v2f* I = reinterpret_cast<const v2f*>(Ib + fb * istride);
        P0[fb] += I[k] * w4.x;
        P1[fb] += I[k1] * w4.y;
        P0[fb] += I[k2] * w4.z;
        P1[fb] += I[k3] * w4.w;
      }
    }
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const v2f Pj = P0[fb] + P1[fb];
      const v2f q = exp2v(a.alpha * log2v(maxv(Pj, kEps)));   // max(eps, P)^alpha  (:206); a NaN intensity stays NaN (tf.maximum)
      Q[fb] = v2f{Pj.x == Pj.x ? q.x : Pj.x, Pj.y == Pj.y ? q.y : Pj.y};
    }
  }
  wave_sync();   // every lane is done with the intensities: the head of each slot takes the frame's G
  v2f acc[FB];   // sum_i Q_i S[i, j] on the matrix cores, offset factor outside the sum  (:185-208)
  spread_tiles<FB>(Q, reinterpret_cast<const char*>(img + a.off_S), lane, acc);
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const v2f offset = (1.0f - a.drown) * (t[fb] * beta + 9.0f * t[fb] + 5.5f);
    const v2f fac = exp2v(offset * (-a.alpha * 0.33219280948873623f));                 // 10^(-alpha O / 10)
    const v2f T = exp2v(a.inv_alpha * log2v(maxv(fac * acc[fb], kEps)));                // (:208)
    const v2f Gq = maxv(T, quiet);                                                       // (:144)
    const float px = acc[fb].x + t[fb].x, py = acc[fb].y + t[fb].y;                      // NaN where the reference has NaN
    reinterpret_cast<v2f*>(ibuf + fb * istride)[lane] = v2f{px == px ? Gq.x : px, py == py ? Gq.y : py};
  }
  wave_sync();
  // thr_f = sqrt(max(eps, sum_j G_j W_inv[j, f]))  (:330-331)
  const int W = a.wi_w;   // wi: [e][granule q]: the entries of bins 2 q, 2 q + 1
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int q = 64 * i + lane;
    v2f s0[FB], s1[FB];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      s0[fb] = v2f{0.f, 0.f};
      s1[fb] = v2f{0.f, 0.f};
    }
    const uint4 none = {0u, 0u, 0u, 0u};   // (offset 0, weight 0: adds nothing)
    if (in_frame<R>(a, i, lane)) {
      // two entries per step, so that their loads -- the entry, then the G it points at -- overlap instead of queueing
      for (int e = 0; e < W; e += 2) {
        const uint4 en = wi[(size_t)e * half + q];
        const uint4 en2 = e + 1 < W ? wi[(size_t)(e + 1) * half + q] : none;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
          const char* Gb = ibuf + fb * istride;
          const v2f g0 = *reinterpret_cast<const v2f*>(Gb + en.x), g1 = *reinterpret_cast<const v2f*>(Gb + en.z);
          const v2f h0 = *reinterpret_cast<const v2f*>(Gb + en2.x), h1 = *reinterpret_cast<const v2f*>(Gb + en2.z);
          s0[fb] += g0 * __uint_as_float(en.y);
          s1[fb] += g1 * __uint_as_float(en.w);
          s0[fb] += h0 * __uint_as_float(en2.y);
          s1[fb] += h1 * __uint_as_float(en2.w);
        }
      }
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const v2f u0 = maxv(s0[fb], kEps), u1 = maxv(s1[fb], kEps);
        emit(fb, i, v4f{s0[fb].x == s0[fb].x ? __builtin_amdgcn_sqrtf(u0.x) : s0[fb].x, s0[fb].y == s0[fb].y ? __builtin_amdgcn_sqrtf(u0.y) : s0[fb].y,
                        s1[fb].x == s1[fb].x ? __builtin_amdgcn_sqrtf(u1.x) : s1[fb].x, s1[fb].y == s1[fb].y ? __builtin_amdgcn_sqrtf(u1.y) : s1[fb].y});
      }
    }
  }
}

}  // namespace mid

// the launch-time parameters of a plan's image (ac_psy_mid.hip)
mid::MidParams mid_params(const ac_psy_plan* p, float drown);

}  // namespace ac
