// Device code of the wave-level masking model for general band layouts in its run-structured form (round 4): the model
// of ac_psy_mid_dev.h with the structure every Bark mapping of the reference has (psychoacoustic.py:257-299) taken out of
// the frame loop at plan time.  Shared by the stand-alone kernel (k_psy_mid, ac_psy_mid.hip), the fused encode of the
// several-frames-per-wave MDCT kernels (k_fwd_multi, ac_fast.hip) and the fused encode of the LDS-FFT tier (k_fwd_wave_v,
// ac_generic.hip): ONE definition of the per-frame arithmetic, so the fused and the un-fused encode agree bit for bit.
//
// What the structure is (checked by build_runs on the host; a plan that does not have it keeps the band walk):
//   * W (bins -> bands, :301-315): the bins of band j are one contiguous run f0 .. f1; the interior bins f0+1 .. f1-1 carry
//     weight exactly 1, only the two edge bins a fraction.  So P_j = w0 I[f0] + w1 I[f1] + (sum of the interior), and the
//     interior is covered by a host-built list of aligned partial sums over 4, 16 (and 64) bins plus single bins: at most
//     8 ... 17 list entries for the widest band instead of a walk over its up to 270 bins (the old form walked with the
//     longest band deciding the trip count of every lane).
//   * W_inv (bands -> bins, :317-331): a bin inside one band sees G_j rho_j with one rho per band; a bin that holds a
//     band edge is the only one with its combination.  There are at most M - 1 such bins, so a frame has at most 2 M
//     distinct threshold values ("entries"): lane j forms the interior entry of band j and the entry of edge bin j --
//     two square roots per lane -- and every bin then only looks its entry up (the old form summed <= wi_w weighted
//     terms and took a square root per bin).
//   * T = max(eps, fac acc)^(1/alpha) with fac = 10^(-alpha O / 10) (:185-208) is evaluated in the log domain:
//     exp2(max(log2 acc - alpha O log2(10) / 10, log2 eps) / alpha): one v_log + one v_exp per band and signal instead of
//     three transcendentals.
// gfx950 only.
#pragma once
#include "ac_psy_mid_dev.h"

namespace ac {
namespace runs {

using mid::v2f;
using mid::v4f;
using mid::kEps;
using mid::wave_sync;
using mid::log2v;
using mid::exp2v;
using mid::maxv;

// The per-frame LDS slot (byte offsets from its base; every list entry of the image is such an offset, 16 bits):
//   [0, 8 N)            intensities, bin f at 8 f: (s0, s1)          -- later reused: G_j at 8 j (512 bytes), threshold
//                       entries at 512 + 16 j: (interior entry of band j, entry of edge bin j), each (s0, s1)
//   [o4, ...)           sums over aligned runs of 4 bins, 8 bytes each; then runs of 16; then (when the plan uses them) of 64
//   [oz, oz + 8)        zeros: the padding target of the lists (written once per slot by runs::slot_init)
// The first part is at least 1536 bytes (G and the entries need them when 8 N is less).
struct RunsParams {
  int img_words;                     // the whole image
  int lds_words;                     // its part that is copied to LDS: all of it, or everything before the per-bin entry offsets
                                     // (kernels with up to 8 granule registers per lane hold those in registers)
  int N, M;
  int lw;                            // list words per band (two 16-bit offsets each)
  int kb;                            // terms of an edge-bin entry (bands that meet in one bin)
  int n4, n16, n64;                  // aligned runs per level (n64 = 0: level not used)
  int o4, o16, o64, oz;              // slot offsets (bytes)
  int slot;                          // bytes per slot (a multiple of 16)
  int off_S, off_bc, off_bd, off_lst, off_bw, off_idx;   // word offsets inside the image
  float alpha, inv_alpha;
  float omd;                         // 1 - drown
  float inv_n;                       // 1 / N
};
// image (32-bit words):
//   off_S:   bfloat16 tiles of the spreading matrix (mid::spread_tiles): hi table, lo table
//   off_bc:  [64] x 4: {edge offsets lo | hi << 16, w0, w1, quiet}
//   off_bd:  [64] x 4: {beta + 9, rho, byte offset of the first G of edge bin `lane`'s entry, 0}
//   off_lst: [lw][64]: list words of band `lane`
//   off_bw:  [kb][64]: weights of edge bin `lane`'s entry (0 pads)
//   off_idx: [R][64]:  byte offsets (lo | hi << 16) of the entries of the two bins of granule 64 i + lane

constexpr float kLog2Eps = -46.506993328423076f;    // log2(1e-14)
constexpr float kLog2_10_10 = 0.33219280948873623f; // log2(10) / 10

// constants of band / edge bin `lane`, loop-invariant
struct RunsLane {
  uint32_t edge;
  float w0, w1, quiet;
  float c1, rho;
  uint32_t goff;
};
__device__ __forceinline__ RunsLane load_lane(const RunsParams& a, const uint32_t* img, int lane) {
  const uint4 bc = reinterpret_cast<const uint4*>(img + a.off_bc)[lane];
  const uint4 bd = reinterpret_cast<const uint4*>(img + a.off_bd)[lane];
  RunsLane c;
  c.edge = bc.x;
  c.w0 = __uint_as_float(bc.y);
  c.w1 = __uint_as_float(bc.z);
  c.quiet = __uint_as_float(bc.w);
  c.c1 = __uint_as_float(bd.x);
  c.rho = __uint_as_float(bd.y);
  c.goff = bd.z;
  return c;
}

// once per slot, before its first frame (the zero word is never written again)
__device__ __forceinline__ void slot_init(const RunsParams& a, char* slot0, int nslots, int slot_bytes, int lane) {
  if (lane < nslots) *reinterpret_cast<v2f*>(slot0 + lane * slot_bytes + a.oz) = v2f{0.f, 0.f};
}

template <int R>
__device__ __forceinline__ bool in_frame(const RunsParams& a, int i, int lane) { return R * 128 == a.N || 64 * i + lane < (a.N >> 1); }

// tonality of FB frames (psychoacoustic.py:102-120; the arithmetic of mid::tonality_frames) and, STORE_I, their
// intensities into the slots (the squares are formed once for both)
template <int R, int FB, bool WANT_T, bool STORE_I>
__device__ __forceinline__ void prep_frames(const v4f (&xq)[FB][R], const RunsParams& a, char* slot0, int slot_bytes, int lane, v2f (&t)[FB]) {
  v2f slog[FB], ssq[FB];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    slog[fb] = v2f{0.f, 0.f};
    ssq[fb] = v2f{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; ++i) {
      v4f I = xq[fb][i] * xq[fb][i];
      asm("" : "+v"(I));   // the squares stay rounded products (see psy_stage in ac_fast.hip)
      if (STORE_I && in_frame<R>(a, i, lane)) *reinterpret_cast<v4f*>(slot0 + fb * slot_bytes + 16 * (64 * i + lane)) = I;
      if (WANT_T) {
        const v2f ie = v2f{I.x, I.y}, io = v2f{I.z, I.w};
        ssq[fb] += ie + io;
        const v2f lg = log2v(maxv(ie, kEps) * maxv(io, kEps));
        slog[fb] += in_frame<R>(a, i, lane) ? lg : v2f{0.f, 0.f};
      }
    }
  }
  if (!WANT_T) return;
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    slog[fb].x = mid::wave_sum(slog[fb].x);
    slog[fb].y = mid::wave_sum(slog[fb].y);
    ssq[fb].x = mid::wave_sum(ssq[fb].x);
    ssq[fb].y = mid::wave_sum(ssq[fb].y);
  }
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const v2f am = ssq[fb] * a.inv_n + kEps;
    const v2f sfm = 3.0102999566398120f * (slog[fb] * a.inv_n - log2v(am));
    const v2f tt = sfm * (-1.0f / 60.0f);
    t[fb] = v2f{fminf(tt.x, 1.0f), fminf(tt.y, 1.0f)};
  }
}

// one level of partial sums: n runs, run c = the sum of the 32 bytes (four v2f) at src + 32 c, written to dst + 8 c.
// The lane's two 16-byte reads are taken in the order that keeps a group of 16 lanes on 16 different bank quads
// ((c >> 3) & 1 picks the half read first; the sum is the same either way).
template <int FB>
__device__ __forceinline__ void level_sums(char* slot0, int slot_bytes, int src, int dst, int n, int lane) {
  for (int c = lane; c < n; c += 64) {
    const int first = ((c >> 3) & 1) * 16;
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const char* s = slot0 + fb * slot_bytes + src + 32 * c;
      const v4f g0 = *reinterpret_cast<const v4f*>(s + first), g1 = *reinterpret_cast<const v4f*>(s + (16 - first));
      const v4f g = g0 + g1;
      *reinterpret_cast<v2f*>(slot0 + fb * slot_bytes + dst + 8 * c) = v2f{g.x + g.z, g.y + g.w};
    }
  }
}

// masking thresholds of FB frames whose intensities are in their slots (prep_frames, or the caller's own stores: bin f
// at 8 f); emit(fb, i, th) receives the thresholds of granule 64 i + lane of frame fb (lanes with in_frame(i) only).
// img: the LDS copy of the image.  The caller orders its stores of the intensities before the call (wave_sync) and its
// next use of the slots after it.
// idx(i): the entry-offset word of granule 64 i + lane (LdsIdx: read from the LDS image; or the caller's registers)
struct LdsIdx {
  const uint32_t* p;   // img + off_idx + lane
  __device__ __forceinline__ uint32_t operator()(int i) const { return p[64 * i]; }
};
template <int R>
struct RegIdx {
  uint32_t w[R];
  __device__ __forceinline__ void load(const uint32_t* gimg, const RunsParams& a, int lane) {   // gimg: the image in global memory
#pragma unroll
    for (int i = 0; i < R; ++i) w[i] = gimg[a.off_idx + 64 * i + lane];
  }
  __device__ __forceinline__ uint32_t operator()(int i) const { return w[i]; }
};
template <int R, int FB, class IDX, class EMIT>
__device__ __forceinline__ void threshold_frames(const v2f (&t)[FB], const RunsParams& a, const RunsLane& c, const uint32_t* img,
                                                 char* slot0, int slot_bytes, int lane, const IDX& idx, EMIT emit) {
  // partial sums over aligned runs of 4, 16 (64) bins
  level_sums<FB>(slot0, slot_bytes, 0, a.o4, a.n4, lane);
  wave_sync();
  level_sums<FB>(slot0, slot_bytes, a.o4, a.o16, a.n16, lane);
  if (a.n64 > 0) {
    wave_sync();
    level_sums<FB>(slot0, slot_bytes, a.o16, a.o64, a.n64, lane);
  }
  wave_sync();
  // P_j = sum_f I_f W[f, j]  (:312-313): lane = band; two weighted edge bins, the interior through the list
  v2f P0[FB], P1[FB];
  {
    const uint32_t e0 = c.edge & 0xffffu, e1 = c.edge >> 16;
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const char* s = slot0 + fb * slot_bytes;
      P0[fb] = *reinterpret_cast<const v2f*>(s + e0) * c.w0;
      P1[fb] = *reinterpret_cast<const v2f*>(s + e1) * c.w1;
    }
    const uint32_t* lst = img + a.off_lst + lane;
    for (int k = 0; k < a.lw; ++k) {
      const uint32_t w = lst[64 * k];
      const uint32_t u0 = w & 0xffffu, u1 = w >> 16;
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const char* s = slot0 + fb * slot_bytes;
        P0[fb] += *reinterpret_cast<const v2f*>(s + u0);
        P1[fb] += *reinterpret_cast<const v2f*>(s + u1);
      }
    }
  }
  v2f Q[FB];   // max(eps, P)^alpha (:206); lanes beyond the M bands keep 0: the rows of S they would meet do not exist
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const v2f q = exp2v(a.alpha * log2v(maxv(P0[fb] + P1[fb], kEps)));
    Q[fb] = lane < a.M ? q : v2f{0.f, 0.f};
  }
  v2f acc[FB];   // sum_i Q_i S[i, j] on the matrix cores, offset factor outside the sum  (:185-208)
  mid::spread_tiles<FB>(Q, reinterpret_cast<const char*>(img + a.off_S), lane, acc);
  wave_sync();   // every lane is done with the intensities and their sums: the head of each slot takes G and the entries
  v2f G[FB];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const v2f offset = a.omd * (t[fb] * c.c1 + 5.5f);                                          // (1 - drown) (t beta + 9 t + 5.5)
    const v2f y = maxv(log2v(acc[fb]) - (a.alpha * kLog2_10_10) * offset, kLog2Eps);           // log2 max(eps, fac acc)
    G[fb] = maxv(exp2v(a.inv_alpha * y), c.quiet);                                             // (:208, :144)
    *reinterpret_cast<v2f*>(slot0 + fb * slot_bytes + 8 * lane) = G[fb];
  }
  wave_sync();
  // entries: interior bins of band `lane` sqrt(max(eps, G rho)); edge bin `lane` sqrt(max(eps, sum_k G_{j0+k} u_k))  (:330-331)
  {
    v2f s[FB];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) s[fb] = v2f{0.f, 0.f};
    const float* bw = reinterpret_cast<const float*>(img + a.off_bw) + lane;
    for (int k = 0; k < a.kb; ++k) {
      const float u = bw[64 * k];
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) s[fb] += *reinterpret_cast<const v2f*>(slot0 + fb * slot_bytes + c.goff + 8 * k) * u;
    }
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const v2f A0 = maxv(G[fb] * c.rho, kEps), A1 = maxv(s[fb], kEps);
      *reinterpret_cast<v4f*>(slot0 + fb * slot_bytes + 512 + 16 * lane) =
          v4f{__builtin_amdgcn_sqrtf(A0.x), __builtin_amdgcn_sqrtf(A0.y), __builtin_amdgcn_sqrtf(A1.x), __builtin_amdgcn_sqrtf(A1.y)};
    }
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (in_frame<R>(a, i, lane)) {
      const uint32_t w = idx(i);
      const uint32_t u0 = w & 0xffffu, u1 = w >> 16;
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const char* s = slot0 + fb * slot_bytes;
        const v2f a0 = *reinterpret_cast<const v2f*>(s + u0), a1 = *reinterpret_cast<const v2f*>(s + u1);
        emit(fb, i, v4f{a0.x, a0.y, a1.x, a1.y});
      }
    }
  }
}

}  // namespace runs

// the launch-time parameters of a plan's run-structured image (ac_psy_mid.hip); runs_supported: the plan has one
bool runs_supported(const ac_psy_plan* p);
// idx_in_lds: the kernel reads the per-bin entry offsets from the LDS image (else it holds them in registers)
runs::RunsParams runs_params(const ac_psy_plan* p, float drown, bool idx_in_lds);

}  // namespace ac
