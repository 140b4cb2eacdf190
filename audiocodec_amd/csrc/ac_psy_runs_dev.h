// Device code of the wave-level masking model for general band layouts in its run-structured form (round 4): the model
// of ac_psy_mid_dev.h with the structure every Bark mapping of the reference has (psychoacoustic.py:257-299) taken out of
// the frame loop at plan time.  Shared by the stand-alone kernel (k_psy_runs, ac_psy_mid.hip), the fused encode of the
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
//     distinct threshold values ("entries"): the interior entry of band j and the entry of edge bin j -- two square roots
//     per band -- and every bin then only looks its entry up (the old form summed <= wi_w weighted terms and took a square
//     root per bin).
//   * T = max(eps, fac acc)^(1/alpha) with fac = 10^(-alpha O / 10) (:185-208) is evaluated in the log domain:
//     exp2(max(log2 acc - alpha O log2(10) / 10, log2 eps) / alpha): one v_log + one v_exp per band and signal instead of
//     three transcendentals.
//   * the wave sums of the tonality (:102-120) of all the frames of a group are formed together: a reduction over lanes
//     that halves the number of live registers with every exchange (4 FB values end as one register, lane & 15 = which
//     value), one vector evaluation of the flatness formula for all of them.
//   * the band x band product with the Toeplitz spreading matrix (:205-207, 223-228) runs on the matrix cores on
//     split-bfloat16 operands (mid::spread_tiles: v_mfma_f32_4x4x4_16b_bf16, a step's B tiles read once for the frames of a
//     group), lane = band before and after.
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
using mid::MF_COPY_STRIDE;
using mid::MF_TAB_BYTES;

// The per-frame LDS slot (byte offsets from its base; every list entry of the image is such an offset, 16 bits):
//   [0, ipart)          intensities, bin f at 8 f: (s0, s1); ipart = max(8 N, 1536) -- later reused: G_j at 8 j (512
//                       bytes), threshold entries at 512 + 16 j: (interior entry of band j, entry of edge bin j), each (s0, s1)
//   [o4, ...)           sums over aligned runs of 4 bins, 8 bytes each; then runs of 16; then (when the plan uses them) of 64
//   [oz, oz + 8)        zeros: the padding target of the lists (runs::slot_init)
// What depends on filter_bands_n alone is RunsGeo (a compile-time constant in the kernels that know the size):
struct RunsGeo {
  int n4, n16, o4, o16, o64;   // aligned runs per level; slot offsets of the levels (o64: where the third level starts if used)
};
__host__ __device__ constexpr int a16(int v) { return (v + 15) / 16 * 16; }
__host__ __device__ constexpr RunsGeo runs_geo(int N) {
  RunsGeo g = {N / 4, N / 16, (8 * N > 1536 ? 8 * N : 1536), 0, 0};
  g.o16 = a16(g.o4 + 8 * g.n4);
  g.o64 = a16(g.o16 + 8 * g.n16);
  return g;
}
// the largest slot build_runs can lay out for a size (all levels): a compile-time stride for kernels that know the size
__host__ __device__ constexpr int runs_slot_max(int N) { return a16(a16(runs_geo(N).o64 + 8 * (N / 64)) + 8); }

// what depends on the plan
struct RunsParams {
  int N, M;
  int lds_words;                     // the part of the image that is copied to LDS: all of it, or everything before the per-bin
                                     // entry offsets (kernels with up to 8 granule registers per lane hold those in registers)
  int lw;                            // list words per band (two 16-bit offsets each)
  int kb;                            // terms of an edge-bin entry (bands that meet in one bin)
  int n64;                           // aligned runs of 64 bins (0: level not used)
  int oz;                            // slot offset of the zero word
  int slot;                          // bytes per slot (a multiple of 16; <= runs_slot_max(N))
  float alpha, inv_alpha;
  float omd;                         // 1 - drown
  float inv_n;                       // 1 / N
};
// image (32-bit words):
//   OFF_S:   bfloat16 tiles of the spreading matrix: four shifted copies of the reversed prototype rev[m] = gp[128 - m],
//            gp[64 + d] = g[M + d] (0 where |d| >= M: S[i][j] = gp[64 + j - i]); hi parts (MF_TAB_BYTES) then lo parts
//   OFF_BC:  [64] x 4: {edge offsets lo | hi << 16, w0, w1, byte offset of the first G of edge bin `lane`'s entry}
//   OFF_BD:  [64] x 4: per band {beta + 9, quiet, rho, 0}
//   OFF_LST: [lw][64]: list words of band `lane`
//   off_bw:  [kb][64]: weights of edge bin `lane`'s entry (0 pads)
//   off_idx: [R][64]:  byte offsets (lo | hi << 16) of the entries of the two bins of granule 64 i + lane
constexpr int OFF_S = 0, OFF_BC = 2 * (MF_TAB_BYTES / 4), OFF_BD = OFF_BC + 256, OFF_LST = OFF_BD + 256;
__host__ __device__ constexpr int off_bw(int lw) { return OFF_LST + 64 * lw; }
__host__ __device__ constexpr int off_idx(int lw, int kb) { return OFF_LST + 64 * (lw + kb); }

constexpr float kLog2Eps = -46.506993328423076f;    // log2(1e-14)
constexpr float kLog2_10_10 = 0.33219280948873623f; // log2(10) / 10

// constants of band / edge bin `lane` for the list walk and the edge entries
struct RunsLane {
  uint32_t edge;
  float w0, w1;
  uint32_t goff;
};
__device__ __forceinline__ RunsLane load_lane(const uint32_t* img, int lane) {
  const uint4 bc = reinterpret_cast<const uint4*>(img + OFF_BC)[lane];
  return RunsLane{bc.x, __uint_as_float(bc.y), __uint_as_float(bc.z), bc.w};
}

// the zero word of each slot (threshold_frames writes them anew for every group: G and the entries of a group run over them in small slots)
__device__ __forceinline__ void slot_init(const RunsParams& a, char* slot0, int nslots, int slot_bytes, int lane) {
  if (lane < nslots) *reinterpret_cast<v2f*>(slot0 + lane * slot_bytes + a.oz) = v2f{0.f, 0.f};
}

template <int R>
__device__ __forceinline__ bool in_frame(const RunsParams& a, int i, int lane) { return R * 128 == a.N || 64 * i + lane < (a.N >> 1); }

// ---- wave totals of V = 4, 8 or 16 values at once -------------------------------------------------------------------------
// Four exchanges inside a row of 16 lanes -- partners lane ^ 15 (row_mirror), ^ 7 (row_half_mirror), ^ 3 and ^ 1 (quad_perm) --
// then the four rows.  While more than one register is live an exchange halves them: a lane keeps the value its bit of the
// exchange selects, hands the other to its partner and adds what it is handed.  Every value meets the same partners in the
// same order whatever V and whichever register it started in, so a frame's sums do not depend on the group it shares.
// Afterwards v[0] of lane l holds the total of value number l & 15 (V = 16), (l >> 1) & 7 (V = 8) or (l >> 2) & 3 (V = 4).
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL, int N>
__device__ __forceinline__ void exchange(float (&v)[16], bool sel) {   // N live registers -> N / 2 (N == 1: stays one)
  if constexpr (N == 1) {
    v[0] += dpp_get<CTRL>(v[0]);
  } else {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const float keep = sel ? v[i + N / 2] : v[i], give = sel ? v[i] : v[i + N / 2];
      v[i] = keep + dpp_get<CTRL>(give);
    }
  }
}
template <int V>
__device__ __forceinline__ float wave_totals(float (&v)[16], int lane) {
  static_assert(V == 4 || V == 8 || V == 16, "4 values per frame, 1 / 2 / 4 frames");
  exchange<0x140, V>(v, (lane & 8) != 0);                       // row_mirror
  exchange<0x141, (V >= 2 ? V / 2 : 1)>(v, (lane & 4) != 0);    // row_half_mirror
  exchange<0x1B, (V >= 4 ? V / 4 : 1)>(v, (lane & 2) != 0);     // quad_perm [3, 2, 1, 0]
  exchange<0xB1, (V >= 8 ? V / 8 : 1)>(v, (lane & 1) != 0);     // quad_perm [1, 0, 3, 2]
  // rows: v_permlane32_swap(x, x) = (lanes 0..31 twice, lanes 32..63 twice); v_permlane16_swap(y, y) = (even rows twice, odd
  // rows twice)
  const unsigned x = __float_as_uint(v[0]);
  const auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  const float y = __uint_as_float(h[0]) + __uint_as_float(h[1]);
  const unsigned yu = __float_as_uint(y);
  const auto r = __builtin_amdgcn_permlane16_swap(yu, yu, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// tonality of FB frames (psychoacoustic.py:102-120) and, STORE_I, their intensities into the slots (the squares are formed
// once for both).  Value number of (frame fb, kind = 0 sum of logs / 1 sum of squares, signal ch) in wave_totals:
// FB = 4: 4 fb + 2 kind + ch; FB = 2: 4 fb + 2 ch + kind; FB = 1: 2 kind + ch -- so that the lane holding a frame's sum of logs
// finds the matching sum of squares at lane ^ 2 (FB = 4, 2: quad_perm) or lane ^ 8 (FB = 1: row_ror 8).
// ISRC: isrc(fb, i) = the intensities of granule 64 i + lane of frame fb, (I[2q], I[2q+1]) x (s0, s1), zero past the frame
// (squares formed in registers, or read back where a kernel keeps them in LDS).  A lane adds its granules in tonality_ways(R)
// interleaved accumulators (i = way, way + ways, ...: one accumulator up to 1024 bins, two up to 2048, four above), the
// accumulators in order, then the lanes (wave_totals): one association for every kernel that calls this, whatever holds the
// spectrum -- and one that the waves of a frame dealt to two or four waves can form side by side (a wave per accumulator).
__host__ __device__ constexpr int tonality_ways(int R) { return R >= 32 ? 4 : R >= 16 ? 2 : 1; }
// one accumulator of a lane: {sum of log2, s0 | s1, sum of squares, s0 | s1} over the granules i = first, first + step, ... < R.
// CHUNK (0: none): a scheduling fence after every CHUNK granules -- for sources that read LDS, where the compiler would otherwise
// issue all the reads ahead and hold four registers each.
template <int R, int CHUNK, class ISRC1>
__device__ __forceinline__ v4f lane_sums(ISRC1 isrc1, const RunsParams& a, int lane, int first, int step) {
  v2f slog = {0.f, 0.f}, ssq = {0.f, 0.f};
  int k = 0;
#pragma unroll
  for (int i = first; i < R; i += step, ++k) {
    if (CHUNK > 0 && k > 0 && k % CHUNK == 0) __builtin_amdgcn_sched_barrier(0);
    const v4f I = isrc1(i);
    const v2f ie = v2f{I.x, I.y}, io = v2f{I.z, I.w};
    ssq += ie + io;
    const v2f lg = log2v(maxv(ie, kEps) * maxv(io, kEps));
    slog += in_frame<R>(a, i, lane) ? lg : v2f{0.f, 0.f};
  }
  return v4f{slog.x, slog.y, ssq.x, ssq.y};
}
// from the lanes' sums {slog.x, slog.y, ssq.x, ssq.y} of FB frames to their tonalities
template <int FB>
__device__ __forceinline__ void tonality_finish(const v4f (&acc)[FB], const RunsParams& a, int lane, v2f (&t)[FB]) {
  float v[16];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    if (FB == 4) {
      v[4 * fb + 0] = acc[fb].x, v[4 * fb + 1] = acc[fb].y, v[4 * fb + 2] = acc[fb].z, v[4 * fb + 3] = acc[fb].w;
    } else if (FB == 2) {
      v[4 * fb + 0] = acc[fb].x, v[4 * fb + 1] = acc[fb].z, v[4 * fb + 2] = acc[fb].y, v[4 * fb + 3] = acc[fb].w;
    } else {
      v[0] = acc[fb].x, v[1] = acc[fb].y, v[2] = acc[fb].z, v[3] = acc[fb].w;
    }
  }
  const float tot = wave_totals<4 * FB>(v, lane);
  // in the lanes that hold a sum of logs: the frame's sum of squares from the partner lane, then
  // sfm = 10 log10(gm / am) = 10 log10(2) (mean log2 I - log2 am), t = min(sfm / -60, 1)   (the other lanes compute on without use)
  const float sq = FB == 1 ? dpp_get<0x128>(tot) : dpp_get<0x4E>(tot);   // row_ror 8 | quad_perm [2, 3, 0, 1]
  const float am = sq * a.inv_n + kEps;
  const float sfm = 3.0102999566398120f * (tot * a.inv_n - __builtin_amdgcn_logf(am));
  // a frame with a NaN or an infinite intensity has a NaN tonality, as tf.maximum / reduce_mean / tf.minimum make it
  // (psychoacoustic.py:113-118): its sum of squares is not finite (v_max / v_min alone would return the other operand)
  const float tt = (sq - sq == 0.0f) ? fminf(sfm * (-1.0f / 60.0f), 1.0f) : __builtin_nanf("");
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const int l0 = FB == 4 ? 4 * fb : FB == 2 ? 8 * fb : 0, l1 = FB == 4 ? 4 * fb + 1 : FB == 2 ? 8 * fb + 4 : 4;
    t[fb] = v2f{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), l0)),
                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), l1))};
  }
}
template <int R, int FB, int CHUNK = 0, class ISRC>
__device__ __forceinline__ void tonality_from(ISRC isrc, const RunsParams& a, int lane, v2f (&t)[FB]) {
  constexpr int WAYS = tonality_ways(R);
  v4f acc[FB];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    acc[fb] = lane_sums<R, CHUNK>([&](int i) { return isrc(fb, i); }, a, lane, 0, WAYS);
#pragma unroll
    for (int w = 1; w < WAYS; ++w) acc[fb] += lane_sums<R, CHUNK>([&](int i) { return isrc(fb, i); }, a, lane, w, WAYS);
  }
  tonality_finish<FB>(acc, a, lane, t);
}
// the intensities of a granule: rounded products (left to -ffp-contract=fast the compiler fuses one of the two squares of
// ie + io into the sum of squares -- which one differs between instantiations; see psy_stage in ac_fast.hip)
__device__ __forceinline__ v4f squares(v4f x) {
  v4f I = x * x;
  asm("" : "+v"(I));
  return I;
}
// on a spectrum in registers: xq[fb][i] = granule 64 i + lane of frame fb, (X[2q], X[2q+1]) x (s0, s1), zero past the frame
template <int R, int FB, bool WANT_T, bool STORE_I>
__device__ __forceinline__ void prep_frames(const v4f (&xq)[FB][R], const RunsParams& a, char* slot0, int slot_bytes, int lane, v2f (&t)[FB]) {
  if (WANT_T) {
    tonality_from<R, FB, 0>([&](int fb, int i) {
      const v4f I = squares(xq[fb][i]);
      if (STORE_I && in_frame<R>(a, i, lane)) *reinterpret_cast<v4f*>(slot0 + fb * slot_bytes + 16 * (64 * i + lane)) = I;
      return I;
    }, a, lane, t);
  } else if (STORE_I) {
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (in_frame<R>(a, i, lane)) *reinterpret_cast<v4f*>(slot0 + fb * slot_bytes + 16 * (64 * i + lane)) = squares(xq[fb][i]);
  }
}

// one level of partial sums: n runs, run c = the sum of the 32 bytes (four v2f) at src + 32 c, written to dst + 8 c.
// The lane's two 16-byte reads are taken in the order that keeps a group of 16 lanes on 16 different bank quads
// ((c >> 3) & 1 picks the half read first; the sum is the same either way).
// tid / nt: the lane's number among the nt lanes that share the work (one wave: lane / 64; more where several waves hold a frame)
template <int FB>
__device__ __forceinline__ void level_sums(char* slot0, int slot_bytes, int src, int dst, int n, int tid, int nt = 64) {
  for (int c = tid; c < n; c += nt) {
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
    for (int i = 0; i < R; ++i) w[i] = gimg[off_idx(a.lw, a.kb) + 64 * i + lane];
  }
  __device__ __forceinline__ uint32_t operator()(int i) const { return w[i]; }
};

// band_sums: P_j of FB frames whose intensities and partial sums are in their slots -- lane = band.  One wave.  The caller
// orders the partial sums before the call (wave_sync, or a workgroup barrier where other waves formed them).
template <int FB>
__device__ __forceinline__ void band_sums(const RunsParams& a, const RunsLane& lc, const uint32_t* img, char* slot0, int slot_bytes, int lane,
                                          v2f (&P)[FB]) {
  static_assert(FB == 1 || FB == 2 || FB == 4, "frames side by side");
  slot_init(a, slot0, FB, slot_bytes, lane);   // (the group before, or the caller's own use, ran over the zero words)
  wave_sync();
  // P_j = sum_f I_f W[f, j]  (:312-313): lane = band; two weighted edge bins, the interior through the list
  v2f P0[FB], P1[FB];
  {
    const uint32_t e0 = lc.edge & 0xffffu, e1 = lc.edge >> 16;
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const char* s = slot0 + fb * slot_bytes;
      P0[fb] = *reinterpret_cast<const v2f*>(s + e0) * lc.w0;
      P1[fb] = *reinterpret_cast<const v2f*>(s + e1) * lc.w1;
    }
    const uint32_t* lst = img + OFF_LST + lane;
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
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) P[fb] = P0[fb] + P1[fb];
}

// band_tail: from the band intensities to the threshold entries -- afterwards slot fb holds its frame's entries (the interior
// entry of band j at 512 + 16 j, the entry of edge bin j at 512 + 16 j + 8); the slots need 1536 bytes each, nothing in them is read.  One wave; the caller orders the look-ups after the call.
template <int FB>
__device__ __forceinline__ void band_tail4(const v2f (&P)[FB], const v2f (&t)[FB], const RunsParams& a, const RunsLane& lc, const uint32_t* img,
                                           char* slot0, int slot_bytes, int lane) {
  v2f Q[FB];   // max(eps, P)^alpha (:206); lanes beyond the M bands keep 0: the rows of S they would meet do not exist
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    v2f q = exp2v(a.alpha * log2v(maxv(P[fb], kEps)));
    // a band with a NaN intensity stays NaN (tf.maximum): through the matrix product it poisons every band of its frame and
    // signal, as the reference's dense einsum does (psychoacoustic.py:205-207)
    q = v2f{P[fb].x == P[fb].x ? q.x : P[fb].x, P[fb].y == P[fb].y ? q.y : P[fb].y};
    Q[fb] = lane < a.M ? q : v2f{0.f, 0.f};
  }
  // sum_i Q_i S[i, j] on the matrix cores, lane = band in and out, offset factor outside the sum  (:185-208), with the
  // 4 x 4 x 4 tiles of ac_psy_mid_dev.h (the wave-level kernels' form)
  v2f acc[FB];
  mid::spread_tiles<FB>(Q, reinterpret_cast<const char*>(img + OFF_S), lane, acc);
  wave_sync();   // every lane is done with the intensities and their sums: the head of the slots takes G and the entries
  {
    const v4f bd = reinterpret_cast<const v4f*>(img + OFF_BD)[lane];                              // {beta + 9, quiet, rho}
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      char* sf = slot0 + fb * slot_bytes;
      const v2f offset = a.omd * (t[fb] * bd.x + 5.5f);                                         // (1 - drown) (t beta + 9 t + 5.5)
      const v2f y = maxv(log2v(acc[fb]) - (a.alpha * kLog2_10_10) * offset, kLog2Eps);          // log2 max(eps, fac acc)
      v2f G = maxv(exp2v(a.inv_alpha * y), bd.y);                                               // (:208, :144)
      // NaN where the reference has NaN: a poisoned product (above) or a NaN tonality (the clamps -- v_max -- would drop it)
      const float poison_x = acc[fb].x + t[fb].x, poison_y = acc[fb].y + t[fb].y;
      G = v2f{poison_x == poison_x ? G.x : poison_x, poison_y == poison_y ? G.y : poison_y};
      const v2f A0 = maxv(G * bd.z, kEps);                                                      // interior bins of band `lane`  (:330-331)
      *reinterpret_cast<v2f*>(sf + 8 * lane) = G;
      *reinterpret_cast<v2f*>(sf + 512 + 16 * lane) =
          v2f{G.x == G.x ? __builtin_amdgcn_sqrtf(A0.x) : G.x, G.y == G.y ? __builtin_amdgcn_sqrtf(A0.y) : G.y};
    }
  }
  wave_sync();
  // edge bin `lane`: sqrt(max(eps, sum_k G_{j0+k} u_k))  (:330-331)
  {
    v2f s[FB];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) s[fb] = v2f{0.f, 0.f};
    const float* bw = reinterpret_cast<const float*>(img + off_bw(a.lw)) + lane;
    for (int k = 0; k < a.kb; ++k) {
      const float u = bw[64 * k];
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) s[fb] += *reinterpret_cast<const v2f*>(slot0 + fb * slot_bytes + lc.goff + 8 * k) * u;
    }
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const v2f A1 = maxv(s[fb], kEps);
      *reinterpret_cast<v2f*>(slot0 + fb * slot_bytes + 512 + 16 * lane + 8) =
          v2f{s[fb].x == s[fb].x ? __builtin_amdgcn_sqrtf(A1.x) : s[fb].x, s[fb].y == s[fb].y ? __builtin_amdgcn_sqrtf(A1.y) : s[fb].y};
    }
  }
}
// ---- the same product as 16 x 16 x 16 tiles (v_mfma_f32_16x16x16_bf16_1k) for up to four frames at once --------------------
//   acc_j = sum_i Q_i S[i, j],  S[i, j] = gp[64 + j - i]
// (NOT v_mfma_f32_16x16x32_bf16, which took half the instructions: with several waves per SIMD a wave issuing it made OTHER waves'
// vector arithmetic return wrong values -- 0.5 - 2 % of the frames of a bench-sized launch, none with one workgroup per CU or
// with the instruction replaced by s_sleep of the same length; DESIGN_LOG.md, round 4.  It is not used anywhere in this library.)
// D = A B: rows of A = (frame fb, part, signal) -- row 4 fb + 2 part + ch, part 0 = Q rounded to bfloat16, part 1 = the
// remainder -- K = band i (four steps of 16), columns = band j (four tiles of 16); S = hi + lo likewise (two B tables).  A goes
// through LDS once (lane = band writes its rows' entries, 2 bytes each; lane (g, n) reads A[row n][16 s + 4 g .. + 3] as 8 bytes,
// rows of 160 bytes); with fewer than four frames the rows repeat (row mod 4 FB).  B[k][col] for lane (g, n), tile c, step s is
// rev[m0 .. m0 + 3], m0 = 64 - 16 (c - s) - n + 4 g: four consecutive entries of the reversed prototype, 8-byte aligned in
// copy n & 3 of the table; tiles with equal s - c are the same registers.  D of tile c: lane (g, n), register 2 part + ch =
// row 4 g + 2 part + ch, column 16 c + n.
constexpr int A_ROW = 160;
template <int FB>
__device__ __forceinline__ void spread16k(const v2f (&Q)[FB], const uint32_t* img, char* abuf, int lane, v4f (&D)[4]) {
  typedef short s4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    const uint32_t whi = mid::pk_bf16(Q[fb].x, Q[fb].y);
    const float hx = __uint_as_float(whi << 16), hy = __uint_as_float(whi & 0xffff0000u);
    const uint32_t wlo = mid::pk_bf16(Q[fb].x - hx, Q[fb].y - hy);
    char* w = abuf + (4 * fb) * A_ROW + 2 * lane;
    *reinterpret_cast<uint16_t*>(w) = (uint16_t)whi;
    *reinterpret_cast<uint16_t*>(w + A_ROW) = (uint16_t)(whi >> 16);
    *reinterpret_cast<uint16_t*>(w + 2 * A_ROW) = (uint16_t)wlo;
    *reinterpret_cast<uint16_t*>(w + 3 * A_ROW) = (uint16_t)(wlo >> 16);
  }
  wave_sync();
  const int g = lane >> 4, n = lane & 15;
  const char* ar = abuf + ((lane & 15) & (4 * FB - 1)) * A_ROW + 8 * g;
  s4v av[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) av[s] = *reinterpret_cast<const s4v*>(ar + 32 * s);
  const char* bt = reinterpret_cast<const char*>(img + OFF_S) + (n & 3) * mid::MF_COPY_STRIDE + 2 * (64 - (n & ~3) + 4 * g);
#pragma unroll
  for (int c = 0; c < 4; ++c) D[c] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = -3; e <= 3; ++e) {   // e = s - c: tiles in the order of their B registers
    const s4v bh = *reinterpret_cast<const s4v*>(bt + 32 * e), bl = *reinterpret_cast<const s4v*>(bt + mid::MF_TAB_BYTES + 32 * e);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = s - e;
      if (c < 0 || c > 3) continue;
      D[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[s], bh, D[c], 0, 0, 0);
      D[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[s], bl, D[c], 0, 0, 0);
    }
  }
}
template <int FB>
__device__ __forceinline__ void band_tail16(const v2f (&P)[FB], const v2f (&t)[FB], const RunsParams& a, const RunsLane& lc, const uint32_t* img,
                                            char* slot0, int slot_bytes, int lane) {
  v2f Q[FB];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    v2f q = exp2v(a.alpha * log2v(maxv(P[fb], kEps)));
    q = v2f{P[fb].x == P[fb].x ? q.x : P[fb].x, P[fb].y == P[fb].y ? q.y : P[fb].y};
    Q[fb] = lane < a.M ? q : v2f{0.f, 0.f};
  }
  wave_sync();   // every lane is done with the intensities and their sums: the head of the slots takes A, then G and the entries
  v4f D[4];
  spread16k<FB>(Q, img, slot0, lane, D);
  wave_sync();
  // the rest of the per-band arithmetic in the accumulator's layout: lane (g, n) holds frame g % FB, bands 16 c + n of the FB
  // tiles c = (g / FB) FB + ci
  {
    const int g = lane >> 4, n = lane & 15;
    const int f = g & (FB - 1), cg = (g / FB) * FB;
    // the frame's tonality (each candidate through an opaque copy: left alone the compiler turns the selection into an indexed
    // load of t[] from scratch)
    auto opaque = [](v2f x) { asm("" : "+v"(x)); return x; };
    v2f tg = t[0];
    if (FB >= 2) tg = (f & 1) ? opaque(t[1 % FB]) : tg;
    if (FB == 4) tg = f == 2 ? opaque(t[2 % FB]) : f == 3 ? opaque(t[3 % FB]) : tg;
    char* sf = slot0 + f * slot_bytes;
#pragma unroll
    for (int ci = 0; ci < FB; ++ci) {
      v4f d;
      if (FB == 4) d = D[ci];
      else if (FB == 2) d = cg ? D[2 + ci] : D[ci];
      else d = g == 0 ? D[0] : g == 1 ? D[1] : g == 2 ? D[2] : D[3];
      const v2f acc = v2f{d.x + d.z, d.y + d.w};
      const int b = 16 * (cg + ci) + n;
      const v4f bd = reinterpret_cast<const v4f*>(img + OFF_BD)[b];                              // {beta + 9, quiet, rho}
      const v2f offset = a.omd * (tg * bd.x + 5.5f);                                            // (1 - drown) (t beta + 9 t + 5.5)
      const v2f y = maxv(log2v(acc) - (a.alpha * kLog2_10_10) * offset, kLog2Eps);              // log2 max(eps, fac acc)
      v2f G = maxv(exp2v(a.inv_alpha * y), bd.y);                                               // (:208, :144)
      // NaN where the reference has NaN: a poisoned product (above) or a NaN tonality (the clamps -- v_max -- would drop it)
      const float poison_x = acc.x + tg.x, poison_y = acc.y + tg.y;
      G = v2f{poison_x == poison_x ? G.x : poison_x, poison_y == poison_y ? G.y : poison_y};
      const v2f A0 = maxv(G * bd.z, kEps);                                                      // interior bins of band b  (:330-331)
      *reinterpret_cast<v2f*>(sf + 8 * b) = G;
      *reinterpret_cast<v2f*>(sf + 512 + 16 * b) =
          v2f{G.x == G.x ? __builtin_amdgcn_sqrtf(A0.x) : G.x, G.y == G.y ? __builtin_amdgcn_sqrtf(A0.y) : G.y};
    }
  }
  wave_sync();
  // edge bin `lane`: sqrt(max(eps, sum_k G_{j0+k} u_k))  (:330-331)
  {
    v2f s[FB];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) s[fb] = v2f{0.f, 0.f};
    const float* bw = reinterpret_cast<const float*>(img + off_bw(a.lw)) + lane;
    for (int k = 0; k < a.kb; ++k) {
      const float u = bw[64 * k];
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) s[fb] += *reinterpret_cast<const v2f*>(slot0 + fb * slot_bytes + lc.goff + 8 * k) * u;
    }
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const v2f A1 = maxv(s[fb], kEps);
      *reinterpret_cast<v2f*>(slot0 + fb * slot_bytes + 512 + 16 * lane + 8) =
          v2f{s[fb].x == s[fb].x ? __builtin_amdgcn_sqrtf(A1.x) : s[fb].x, s[fb].y == s[fb].y ? __builtin_amdgcn_sqrtf(A1.y) : s[fb].y};
    }
  }
}
// the form in use: 16 x 16 x 16 tiles (-DAC_SPREAD_4X4X4: the 4 x 4 x 4 tiles; both pass tools/scale_check.py at bench size; B =
// 256 stereo, fused encode ms, 4x4x4 -> 16x16x16: 512 0.589 -> 0.573, 256 0.570 -> 0.549, 128 0.740 -> 0.682, 64 1.153 -> 1.048)
template <int FB>
__device__ __forceinline__ void band_tail(const v2f (&P)[FB], const v2f (&t)[FB], const RunsParams& a, const RunsLane& lc, const uint32_t* img,
                                          char* slot0, int slot_bytes, int lane) {
#ifdef AC_SPREAD_4X4X4
  band_tail4<FB>(P, t, a, lc, img, slot0, slot_bytes, lane);
#else
  band_tail16<FB>(P, t, a, lc, img, slot0, slot_bytes, lane);
#endif
}
template <int FB>
__device__ __forceinline__ void band_stage(const v2f (&t)[FB], const RunsParams& a, const RunsLane& lc, const uint32_t* img, char* slot0,
                                           int slot_bytes, int lane) {
  v2f P[FB];
  band_sums<FB>(a, lc, img, slot0, slot_bytes, lane, P);
  band_tail<FB>(P, t, a, lc, img, slot0, slot_bytes, lane);
}

// the threshold of granule q (bins 2 q, 2 q + 1) of the frame in `slot` from its entries; w = the granule's entry-offset word
__device__ __forceinline__ v4f entry_lookup(const char* slot, uint32_t w) {
  const v2f a0 = *reinterpret_cast<const v2f*>(slot + (w & 0xffffu)), a1 = *reinterpret_cast<const v2f*>(slot + (w >> 16));
  return v4f{a0.x, a0.y, a1.x, a1.y};
}

// masking thresholds of FB frames whose intensities are in their slots (prep_frames, or the caller's own stores: bin f
// at 8 f); emit(fb, i, th) receives the thresholds of granule 64 i + lane of frame fb (lanes with in_frame(i) only).
// img: the LDS copy of the image.  NC: filter_bands_n where the kernel knows it at compile time (0: a.N).  The caller orders
// its stores of the intensities before the call (wave_sync) and its next use of the slots after it.
template <int R, int FB, int NC, class IDX, class EMIT>
__device__ __forceinline__ void threshold_frames(const v2f (&t)[FB], const RunsParams& a, const RunsLane& lc, const uint32_t* img,
                                                 char* slot0, int slot_bytes, int lane, const IDX& idx, EMIT emit) {
  const RunsGeo geo = runs_geo(NC ? NC : a.N);
  // partial sums over aligned runs of 4, 16 (64) bins
  level_sums<FB>(slot0, slot_bytes, 0, geo.o4, geo.n4, lane);
  wave_sync();
  level_sums<FB>(slot0, slot_bytes, geo.o4, geo.o16, geo.n16, lane);
  if (a.n64 > 0) {
    wave_sync();
    level_sums<FB>(slot0, slot_bytes, geo.o16, geo.o64, a.n64, lane);
  }
  band_stage<FB>(t, a, lc, img, slot0, slot_bytes, lane);
  wave_sync();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (in_frame<R>(a, i, lane)) {
      const uint32_t w = idx(i);
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) emit(fb, i, entry_lookup(slot0 + fb * slot_bytes, w));
    }
  }
}

}  // namespace runs

// the launch-time parameters of a plan's run-structured image (ac_psy_mid.hip); runs_supported: the plan has one
bool runs_supported(const ac_psy_plan* p);
// idx_in_lds: the kernel reads the per-bin entry offsets from the LDS image (else it holds them in registers)
runs::RunsParams runs_params(const ac_psy_plan* p, float drown, bool idx_in_lds);

}  // namespace ac
