// Wave-level masking model for the configurations the fused epilogue of ac_fast.hip does not serve: any even
// filter_bands_n up to 4096 (a frame is up to R = 1, 2, 4, 8, 16 or 32 granule registers per lane, the last ones partly filled
// when filter_bands_n is not a multiple of 128: 960, 576, 480 ...; above 512 the W_inv entries stay in global memory) with any Bark-band count up to 64 and any band layout (a bin may overlap several bands, bands may share
// bins freely) -- e.g. the models beside the several-frames-per-wave MDCT kernels (filters_n 256 / 512), where the
// O(N)-per-workgroup generic kernels ran at 0.5-0.8 TB/s.  gfx950 only.
//
// The per-frame arithmetic lives in ac_psy_mid_dev.h (shared with the fused encode of the several-frames-per-wave MDCT
// kernels); this file holds the stand-alone kernel, the host-side image builder and the launcher.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ac_internal.h"
#include "ac_psy_mid_dev.h"
#include "ac_psy_runs_dev.h"

namespace ac {
namespace {

using namespace mid;
typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));   // two floats on the 4-byte grid

// Granule registers per lane from which the W_inv entries are read from global memory instead of the LDS image: from 8 (frames
// above 512 bins), where the table is 15 ... 64 KB and keeping it out of LDS doubles the resident waves (B = 256 stereo, the
// un-fused encode: 960 0.912 -> 0.843 ms, 1000 0.921 -> 0.829, 768 unchanged; mono 960 0.736 -> 0.560).
#ifndef AC_MID_WI_GLOBAL_R
#define AC_MID_WI_GLOBAL_R 8
#endif

struct MidArgs {
  const float* X;
  const float* t_in;
  float* t_out;
  float* thr;
  const uint32_t* img;   // ac_psy_plan::d_mid
  MidParams p;
  int C, F;
  int T;                 // frames per wave: workgroup g owns tasks [g nw T, (g + 1) nw T), wave w takes g nw T + w + nw t
  long long nsig, ntasks;
};

// One 64-lane wave per (frame, channel pair) as in k_psy_fast, FB frames at a time (their rows are loaded together: FB
// times the bytes in flight per wave, and the per-frame arithmetic of ac_psy_mid_dev.h interleaves the frames' dependent
// chains): rows move with coalesced 16-byte (stereo) or 8-byte (mono, two signals side by side) accesses.  All constant
// tables sit in one image copied to LDS per workgroup; a wave walks T frames so that the copy is paid once per 4 T frames.
// wave buffer: FB slots of [N] v2f intensities (c0, c1), the head of a slot reused for the frame's 64 G_j
template <int R, int CMODE, bool WANT_T, bool WANT_THR, int FB>
__global__ __launch_bounds__(256, (R >= 32 ? 1 : R >= 16 ? 2 : (WANT_THR && (FB > 1 || R >= 8)) ? 3 : 4)) void k_psy_mid(MidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.p.N;
  const int SLOT = N >= 64 ? 8 * N : 512, WAVE_BYTES = FB * SLOT;   // (a slot also takes the frame's 64 G_j: 512 bytes)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t* img = reinterpret_cast<uint32_t*>(smem);
  if (WANT_THR) {
    for (int i = threadIdx.x; i < a.p.lds_words / 4; i += blockDim.x)
      reinterpret_cast<uint4*>(img)[i] = reinterpret_cast<const uint4*>(a.img)[i];
    __syncthreads();
  }
  char* buf = smem + (size_t)a.p.lds_words * 4 + (size_t)wave * WAVE_BYTES;
  // the W_inv entries: in the LDS image, or (frames above 512 bins) read where the plan keeps them
  const uint4* wi = R >= AC_MID_WI_GLOBAL_R ? reinterpret_cast<const uint4*>(a.img + a.p.off_wi) : reinterpret_cast<const uint4*>(img + a.p.off_wi);
  const int C = a.C;
  const long long task0 = (long long)blockIdx.x * nw * a.T + wave;
  for (int tt = 0; tt < a.T && task0 + (long long)tt * nw < a.ntasks; tt += FB) {   // (no workgroup barrier inside)
    wave_sync();   // the previous group's reads of the wave's buffers are done
    auto in = [&](int i) { return in_frame<R>(a.p, i, lane); };
    bool ok[FB], has1[FB];
    size_t o0[FB], o1[FB], t0[FB], t1[FB];
    v4f xq[FB][R];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const long long task = task0 + (long long)(tt + fb) * nw;
      ok[fb] = tt + fb < a.T && task < a.ntasks;
      const long long tk = ok[fb] ? task : task0 + (long long)tt * nw;   // (a frame past the end re-reads the group's first row)
      const size_t blk = (size_t)N * C;
      if (CMODE == 1) {
        // channels c, c + 1 of clip b (any channel count): rows strided by C, the pair's two values adjacent -- one 8-byte access
        // on the 4-byte grid per bin; the half-empty last pair of an odd count reads (c - 1, c) and keeps the second value.
        // The pairs of one frame are neighbouring tasks: the waves of a workgroup share the cache lines the pairs share.
        const int CP = (C + 1) / 2;
        const long long rest = tk / CP;
        const int c = 2 * (int)(tk - rest * CP), f = (int)(rest % a.F);
        const long long b = rest / a.F;
        has1[fb] = c + 1 < C;
        o0[fb] = ((size_t)b * a.F + (size_t)f) * blk + c;
        o1[fb] = o0[fb];
        t0[fb] = ((size_t)b * a.F + (size_t)f) * C + c;
        t1[fb] = t0[fb] + 1;
        const v2u* row = reinterpret_cast<const v2u*>(a.X + o0[fb] - (has1[fb] ? 0 : 1));
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const int q = 64 * i + lane;
          const v2u u = in(i) ? *reinterpret_cast<const v2u*>(reinterpret_cast<const float*>(row) + (size_t)(2 * q) * C) : v2u{0.f, 0.f};
          const v2u w = in(i) ? *reinterpret_cast<const v2u*>(reinterpret_cast<const float*>(row) + (size_t)(2 * q + 1) * C) : v2u{0.f, 0.f};
          xq[fb][i] = has1[fb] ? v4f{u.x, u.y, w.x, w.y} : v4f{u.y, u.y, w.y, w.y};
        }
        continue;
      }
      // the two signals of the wave: stereo = the two channels of clip p; mono = clips 2 p and 2 p + 1
      const int f = (int)(tk % a.F);
      const long long p = tk / a.F;
      has1[fb] = CMODE == 0 ? true : (2 * p + 1 < a.nsig);
      const long long b0 = CMODE == 0 ? p : 2 * p, b1 = CMODE == 0 ? p : (has1[fb] ? 2 * p + 1 : 2 * p);
      o0[fb] = ((size_t)b0 * a.F + (size_t)f) * blk;
      o1[fb] = ((size_t)b1 * a.F + (size_t)f) * blk;
      t0[fb] = ((size_t)b0 * a.F + (size_t)f) * C;
      t1[fb] = CMODE == 0 ? t0[fb] + 1 : ((size_t)b1 * a.F + (size_t)f) * C;
      // granule q = lane + 64 i: (X[2q], X[2q+1]) x (s0, s1); granules past the frame (q >= N/2) read as zero
      if (CMODE == 0) {
#pragma unroll
        for (int i = 0; i < R; ++i)
          xq[fb][i] = in(i) ? reinterpret_cast<const v4f*>(a.X + o0[fb])[64 * i + lane] : v4f{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const v2f u = in(i) ? reinterpret_cast<const v2f*>(a.X + o0[fb])[64 * i + lane] : v2f{0.f, 0.f};
          // (no branch on has1: the half-empty last pair of an odd batch reads its one signal twice -- b1 above -- and
          // never stores the second; a conditional load would hold the wave at the join)
          const v2f w = in(i) ? reinterpret_cast<const v2f*>(a.X + o1[fb])[64 * i + lane] : v2f{0.f, 0.f};
          xq[fb][i] = v4f{u.x, w.x, u.y, w.y};
        }
      }
    }
    v2f t[FB];
    if (WANT_T) {
      tonality_frames<R, FB>(xq, a.p, lane, t);
#pragma unroll
      for (int fb = 0; fb < FB; ++fb)
        if (ok[fb] && lane == 0) {
          a.t_out[t0[fb]] = t[fb].x;
          if (has1[fb]) a.t_out[t1[fb]] = t[fb].y;
        }
    } else {
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) t[fb] = v2f{a.t_in[t0[fb]], has1[fb] ? a.t_in[t1[fb]] : 0.f};
    }
    if (!WANT_THR) continue;
    threshold_frames<R, FB>(xq, t, a.p, img, wi, buf, SLOT, lane, [&](int fb, int i, const v4f& th) {
      if (!ok[fb]) return;
      if (CMODE == 0) {
        __builtin_nontemporal_store(th, reinterpret_cast<v4f*>(a.thr + o0[fb]) + 64 * i + lane);
      } else if (CMODE == 1) {
        float* r0 = a.thr + o0[fb] + (size_t)(2 * (64 * i + lane)) * C;
        if (has1[fb]) {
          *reinterpret_cast<v2u*>(r0) = v2u{th.x, th.y};
          *reinterpret_cast<v2u*>(r0 + C) = v2u{th.z, th.w};
        } else {
          r0[0] = th.x;
          r0[C] = th.z;
        }
      } else {
        __builtin_nontemporal_store(v2f{th.x, th.z}, reinterpret_cast<v2f*>(a.thr + o0[fb]) + 64 * i + lane);
        if (has1[fb]) __builtin_nontemporal_store(v2f{th.y, th.w}, reinterpret_cast<v2f*>(a.thr + o1[fb]) + 64 * i + lane);
      }
    });
  }   // groups of frames of the wave
}

// frames a wave handles side by side: as many as keep the frames' registers (4 R each) within the budget
constexpr int mid_fb(int R) { return R >= 8 ? 1 : R == 4 ? 2 : 4; }
constexpr int mid_r(int N) { return N <= 128 ? 1 : N <= 256 ? 2 : N <= 512 ? 4 : N <= 1024 ? 8 : N <= 2048 ? 16 : 32; }   // granule registers per lane

struct MidLayout {
  int wi_w = 0, off_S = 0, off_band = 0, off_wbe = 0, off_wi = 0, words = 0;
};

bool build_mid(const ac_psy_plan* p, std::vector<uint32_t>* out, MidLayout* lay) {
  const PsyTables& t = p->host;
  const int N = t.N, M = t.M;
  if (N < 2 || N > 4096 || (N & 1) || M < 1 || M > 64) return false;
  SparseRows wb, wi;
  w_by_band(t, wb);
  winv_by_bin(t, wi);
  if (wi.max_row < 1 || wi.max_row > 16) return false;   // (filters_n = 64 at 48 kHz: a 375 Hz bin spans nine Bark bands)
  MidLayout L;
  L.wi_w = wi.max_row;
  L.off_S = 0;
  L.off_band = L.off_S + 2 * (MF_TAB_BYTES / 4);
  L.off_band = (L.off_band + 3) / 4 * 4;                  // 16-byte aligned rows of four words
  L.off_wbe = L.off_band + 4 * 64;                        // (16-byte aligned: off_band is, and so is every band's start)
  // a band's weights cover the run of bins from its first to its last non-zero (zeros in between, if any, stay zeros)
  std::vector<int> first(M, 0), count(M, 0), start(M, 0);
  int total = 0;
  for (int j = 0; j < M; ++j) {
    if (wb.ptr[j + 1] > wb.ptr[j]) {
      int lo = N, hi = -1;
      for (int e = wb.ptr[j]; e < wb.ptr[j + 1]; ++e) {
        lo = std::min(lo, (int)wb.idx[e]);
        hi = std::max(hi, (int)wb.idx[e]);
      }
      first[j] = lo;
      count[j] = hi - lo + 1;
    }
    start[j] = total;
    total += (count[j] + 3) / 4 * 4;   // (a band's weights: a multiple of four, zero-padded)
  }
  if (total > 4 * N + 4 * M) return false;   // (bands that each span most of the spectrum: not a Bark mapping; other tiers)
  L.off_wi = L.off_wbe + total;
  L.off_wi = (L.off_wi + 3) / 4 * 4;                      // 16-byte reads: the entries of two adjacent bins
  L.words = L.off_wi + 2 * N * L.wi_w;
  L.words = (L.words + 3) / 4 * 4;
  std::vector<uint32_t> w((size_t)L.words, 0u);
  auto putf = [&](int i, float v) { uint32_t u; memcpy(&u, &v, 4); w[(size_t)i] = u; };
  // S is Toeplitz by construction; the kernel reads it through the prototype.  Refuse anything else.
  if ((int)t.g.size() != 2 * M) return false;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j)
      if ((float)t.S[(size_t)i * M + j] != (float)t.g[(size_t)(M - i + j)]) return false;
  {
    // bf16 tiles for spread_tiles: copy c, entry y = rev[y - c], rev[m] = gp[128 - m] (m = 1 .. 127), with
    // gp[64 + d] = g[M + d] where |d| < M, else 0; hi parts, then lo parts (the layout of spread_mfma in ac_fast.hip)
    auto gp = [&](int k) { const int d = k - 64; return (d > -M && d < M) ? (float)t.g[(size_t)(M + d)] : 0.f; };
    auto bf16_rne = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    auto bf16_val = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    uint16_t* tb = reinterpret_cast<uint16_t*>(w.data() + L.off_S);
    for (int c = 0; c < 4; ++c)
      for (int y = 0; y < 132; ++y) {
        const int m = y - c;
        if (m < 1 || m > 127) continue;
        const float v = gp(128 - m);
        const uint16_t hi = bf16_rne(v);
        tb[(c * MF_COPY_STRIDE) / 2 + y] = hi;
        tb[(MF_TAB_BYTES + c * MF_COPY_STRIDE) / 2 + y] = bf16_rne(v - bf16_val(hi));
      }
  }
  for (int j = 0; j < M; ++j) {
    w[(size_t)L.off_band + 4 * j + 0] = (uint32_t)start[j];
    w[(size_t)L.off_band + 4 * j + 1] = (uint32_t)count[j] | ((uint32_t)first[j] << 16);
    putf(L.off_band + 4 * j + 2, (float)t.quiet[j]);
    putf(L.off_band + 4 * j + 3, t.beta[j]);
    for (int e = wb.ptr[j]; e < wb.ptr[j + 1]; ++e) putf(L.off_wbe + start[j] + (wb.idx[e] - first[j]), wb.val[e]);
  }
  for (int f = 0; f < N; ++f) {
    int k = 0;
    for (int e = wi.ptr[f]; e < wi.ptr[f + 1]; ++e, ++k) {
      w[(size_t)L.off_wi + 2 * ((size_t)k * N + f)] = (uint32_t)(8 * wi.idx[e]);   // byte offset of G[band]
      putf(L.off_wi + 2 * (k * N + f) + 1, wi.val[e]);
    }
  }
  if (out) *out = w;
  if (lay) *lay = L;
  return true;
}

size_t mid_lds_bytes(int N, int lds_words, int nw, int fb) { return (size_t)lds_words * 4 + (size_t)nw * fb * (N >= 64 ? 8 * (size_t)N : 512); }

template <int R, int CMODE>
int launch_mid_R(const MidArgs& a, bool want_t, bool want_thr, unsigned grid, int nw, size_t lds, hipStream_t s) {
  const dim3 blk(64 * nw);
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), blk, lds, s, a);
    return AC_OK;
  };
  if (want_t && want_thr) return go(k_psy_mid<R, CMODE, true, true, mid_fb(R)>);
  if (want_thr) return go(k_psy_mid<R, CMODE, false, true, mid_fb(R)>);
  return go(k_psy_mid<R, CMODE, true, false, mid_fb(R)>);
}


// ---- the run-structured form (ac_psy_runs_dev.h) -----------------------------------------------------------------------
struct RunsArgs {
  const float* X;
  const float* t_in;
  float* t_out;
  float* thr;
  const uint32_t* img;   // ac_psy_plan::d_runs
  runs::RunsParams p;
  int C, F;
  int T;                 // frames per wave, as in MidArgs
  long long nsig, ntasks;
};

// rows of FB tasks of a wave: offsets, flags and the granules (the loader of k_psy_mid, shared by both forms)
template <int R, int CMODE, int FB>
struct RowSet {
  bool ok[FB], has1[FB];
  size_t o0[FB], o1[FB], t0[FB], t1[FB];
  template <class INF>
  __device__ __forceinline__ void load(const float* X, int N, int C, int F, long long nsig, long long ntasks, long long task0, int tt, int T,
                                       int nw, int lane, INF in, v4f (&xq)[FB][R]) {
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const long long task = task0 + (long long)(tt + fb) * nw;
      ok[fb] = tt + fb < T && task < ntasks;
      const long long tk = ok[fb] ? task : task0 + (long long)tt * nw;   // (a frame past the end re-reads the group's first row)
      const size_t blk = (size_t)N * C;
      if (CMODE == 1) {
        const int CP = (C + 1) / 2;
        const long long rest = tk / CP;
        const int c = 2 * (int)(tk - rest * CP), f = (int)(rest % F);
        const long long b = rest / F;
        has1[fb] = c + 1 < C;
        o0[fb] = ((size_t)b * F + (size_t)f) * blk + c;
        o1[fb] = o0[fb];
        t0[fb] = ((size_t)b * F + (size_t)f) * C + c;
        t1[fb] = t0[fb] + 1;
        const v2u* row = reinterpret_cast<const v2u*>(X + o0[fb] - (has1[fb] ? 0 : 1));
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const int q = 64 * i + lane;
          const v2u u = in(i) ? *reinterpret_cast<const v2u*>(reinterpret_cast<const float*>(row) + (size_t)(2 * q) * C) : v2u{0.f, 0.f};
          const v2u w = in(i) ? *reinterpret_cast<const v2u*>(reinterpret_cast<const float*>(row) + (size_t)(2 * q + 1) * C) : v2u{0.f, 0.f};
          xq[fb][i] = has1[fb] ? v4f{u.x, u.y, w.x, w.y} : v4f{u.y, u.y, w.y, w.y};
        }
        continue;
      }
      const int f = (int)(tk % F);
      const long long p = tk / F;
      has1[fb] = CMODE == 0 ? true : (2 * p + 1 < nsig);
      const long long b0 = CMODE == 0 ? p : 2 * p, b1 = CMODE == 0 ? p : (has1[fb] ? 2 * p + 1 : 2 * p);
      o0[fb] = ((size_t)b0 * F + (size_t)f) * blk;
      o1[fb] = ((size_t)b1 * F + (size_t)f) * blk;
      t0[fb] = ((size_t)b0 * F + (size_t)f) * C;
      t1[fb] = CMODE == 0 ? t0[fb] + 1 : ((size_t)b1 * F + (size_t)f) * C;
      if (CMODE == 0) {
#pragma unroll
        for (int i = 0; i < R; ++i)
          xq[fb][i] = in(i) ? reinterpret_cast<const v4f*>(X + o0[fb])[64 * i + lane] : v4f{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const v2f u = in(i) ? reinterpret_cast<const v2f*>(X + o0[fb])[64 * i + lane] : v2f{0.f, 0.f};
          const v2f w = in(i) ? reinterpret_cast<const v2f*>(X + o1[fb])[64 * i + lane] : v2f{0.f, 0.f};
          xq[fb][i] = v4f{u.x, w.x, u.y, w.y};
        }
      }
    }
  }
  __device__ __forceinline__ void store_thr(float* thr, int C, int fb, int i, int lane, const v4f& th) const {
    if (!ok[fb]) return;
    if (CMODE == 0) {
      __builtin_nontemporal_store(th, reinterpret_cast<v4f*>(thr + o0[fb]) + 64 * i + lane);
    } else if (CMODE == 1) {
      float* r0 = thr + o0[fb] + (size_t)(2 * (64 * i + lane)) * C;
      if (has1[fb]) {
        *reinterpret_cast<v2u*>(r0) = v2u{th.x, th.y};
        *reinterpret_cast<v2u*>(r0 + C) = v2u{th.z, th.w};
      } else {
        r0[0] = th.x;
        r0[C] = th.z;
      }
    } else {
      __builtin_nontemporal_store(v2f{th.x, th.z}, reinterpret_cast<v2f*>(thr + o0[fb]) + 64 * i + lane);
      if (has1[fb]) __builtin_nontemporal_store(v2f{th.y, th.w}, reinterpret_cast<v2f*>(thr + o1[fb]) + 64 * i + lane);
    }
  }
};

// the stand-alone kernel on the run-structured image: tasks, rows and the frame loop as in k_psy_mid; per wave FB slots of
// a.p.slot bytes behind the image
template <int R, int CMODE, bool WANT_T, bool WANT_THR, int FB>
__global__ __launch_bounds__(256, (R >= 32 ? 1 : R >= 16 ? 2 : 3)) void k_psy_runs(RunsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t* img = reinterpret_cast<uint32_t*>(smem);
  const int slot = a.p.slot;
  char* buf = smem + (size_t)a.p.lds_words * 4 + (size_t)wave * FB * slot;
  constexpr bool IDX_REGS = R <= 8;   // (the launcher sizes lds_words to match)
  if (WANT_THR) {
    for (int i = threadIdx.x; i < a.p.lds_words / 4; i += blockDim.x)
      reinterpret_cast<uint4*>(img)[i] = reinterpret_cast<const uint4*>(a.img)[i];
    __syncthreads();
  }
  runs::RunsLane lc = {};
  runs::RegIdx<IDX_REGS ? R : 1> ridx = {};
  if (WANT_THR) {
    lc = runs::load_lane(img, lane);
    if (IDX_REGS) ridx.load(a.img, a.p, lane);
  }
  const long long task0 = (long long)blockIdx.x * nw * a.T + wave;
  for (int tt = 0; tt < a.T && task0 + (long long)tt * nw < a.ntasks; tt += FB) {   // (no workgroup barrier inside)
    wave_sync();   // the previous group's reads of the wave's slots are done
    RowSet<R, CMODE, FB> rs;
    v4f xq[FB][R];
    rs.load(a.X, a.p.N, a.C, a.F, a.nsig, a.ntasks, task0, tt, a.T, nw, lane, [&](int i) { return runs::in_frame<R>(a.p, i, lane); }, xq);
    v2f t[FB];
    runs::prep_frames<R, FB, WANT_T, WANT_THR>(xq, a.p, buf, slot, lane, t);
    if (WANT_T) {
#pragma unroll
      for (int fb = 0; fb < FB; ++fb)
        if (rs.ok[fb] && lane == 0) {
          a.t_out[rs.t0[fb]] = t[fb].x;
          if (rs.has1[fb]) a.t_out[rs.t1[fb]] = t[fb].y;
        }
    } else {
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) t[fb] = v2f{a.t_in[rs.t0[fb]], rs.has1[fb] ? a.t_in[rs.t1[fb]] : 0.f};
    }
    if (!WANT_THR) continue;
    wave_sync();
    auto emit = [&](int fb, int i, const v4f& th) { rs.store_thr(a.thr, a.C, fb, i, lane, th); };
    if constexpr (IDX_REGS) runs::threshold_frames<R, FB, 0>(t, a.p, lc, img, buf, slot, lane, ridx, emit);
    else runs::threshold_frames<R, FB, 0>(t, a.p, lc, img, buf, slot, lane, runs::LdsIdx{img + runs::off_idx(a.p.lw, a.p.kb) + lane}, emit);
  }
}

// ---- channel counts other than one and two with whole cache lines ("team" form, as k_fwd_wave_c of ac_generic.hip) ----------
// A workgroup = the CP = ceil(C / 2) waves that take the channel pairs of ONE frame.  The row [N, C] comes in and the threshold
// row goes out in 16-byte pieces, consecutive lanes on consecutive addresses: wave w moves piece w (8 N bytes) between HBM and
// its own slot, and every wave picks its pair's bins out of (puts its thresholds into) the row image that the CP slots hold
// together -- on the way in at the slots' heads (where the intensities go afterwards), on the way out behind the threshold
// entries (bytes 1536 ...: intensities and partial sums are dead by then).  Four workgroup barriers per frame; the arithmetic is
// k_psy_runs' (same device functions): equal results bit for bit.  filter_bands_n 258 ... 2048 (4, 8 or 16 granule registers per
// lane; the slots are at least 1536 + 8 N bytes here).
template <int R, bool WANT_T>
__global__ __launch_bounds__(256, (R >= 16 ? 2 : 3)) void k_psy_runs_c(RunsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* img = reinterpret_cast<uint32_t*>(smem);
  const int slot = a.p.slot, N = a.p.N, C = a.C;
  char* slots = smem + (size_t)a.p.lds_words * 4;
  char* buf = slots + (size_t)wave * slot;
  constexpr bool IDX_REGS = R <= 8;
  for (int i = threadIdx.x; i < a.p.lds_words / 4; i += blockDim.x)
    reinterpret_cast<uint4*>(img)[i] = reinterpret_cast<const uint4*>(a.img)[i];
  __syncthreads();
  const runs::RunsLane lc = runs::load_lane(img, lane);
  runs::RegIdx<IDX_REGS ? R : 1> ridx = {};
  if (IDX_REGS) ridx.load(a.img, a.p, lane);
  const int c0 = 2 * wave;
  const bool has1 = c0 + 1 < C, codd = (C & 1) != 0;
  const int PN = 2 * N;                          // floats of the row per piece
  const int adj = slot - 4 * PN;                 // bytes a piece's data sits further on than in the contiguous image
  const int nch = N * C / 4, pch = N / 2;        // 16-byte pieces of a row / of a slot's share
  // byte offset (from `slots`) of float v of the row image: piece v / PN at its slot
  auto image_at = [&](int v) {
    int pc = v >= PN ? 1 : 0;
    pc = v >= 2 * PN ? 2 : pc;
    pc = v >= 3 * PN ? 3 : pc;
    return 4 * v + pc * adj;
  };
  const long long frame0 = (long long)blockIdx.x * a.T;
  for (int tt = 0; tt < a.T && frame0 + tt < a.ntasks; ++tt) {   // (a.ntasks: frames B F; the loop is the workgroup's)
    const long long fr = frame0 + tt;
    const size_t ro = (size_t)fr * N * C;
    // (a lane's image offsets are formed per frame: hoisted out of the loop -- they are invariant -- the 4 R of them spill)
    int vb = 2 * lane * C + c0;
    asm volatile("" : "+v"(vb));
    // the row in: wave w's share, whole 16-byte pieces (no branch around a load; a piece past the row's end -- the half-empty
    // last share of an odd channel count -- re-reads the row's last)
    {
      v4f raw[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int j = min(wave * pch + 64 * i + lane, nch - 1);
        raw[i] = *reinterpret_cast<const v4f*>(a.X + ro + 4 * (size_t)j);
      }
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (runs::in_frame<R>(a.p, i, lane)) *reinterpret_cast<v4f*>(buf + 16 * (64 * i + lane)) = raw[i];
    }
    __syncthreads();
    v4f xq[1][R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int v0 = vb + 128 * i * C;
      v2u u = {0.f, 0.f}, w = {0.f, 0.f};
      if (runs::in_frame<R>(a.p, i, lane)) {
        if (codd) {   // (an odd channel count: a pair's two floats may lie either side of a piece boundary)
          u = v2u{*reinterpret_cast<const float*>(slots + image_at(v0)), *reinterpret_cast<const float*>(slots + image_at(v0 + 1))};
          w = v2u{*reinterpret_cast<const float*>(slots + image_at(v0 + C)), *reinterpret_cast<const float*>(slots + image_at(v0 + C + 1))};
        } else {
          u = *reinterpret_cast<const v2u*>(slots + image_at(v0));
          w = *reinterpret_cast<const v2u*>(slots + image_at(v0 + C));
        }
      }
      xq[0][i] = has1 ? v4f{u.x, u.y, w.x, w.y} : v4f{u.x, u.x, w.x, w.x};
    }
    __syncthreads();   // every wave has its bins: the slots take the intensities
    v2f t[1];
    runs::prep_frames<R, 1, WANT_T, true>(xq, a.p, buf, slot, lane, t);
    const size_t to = (size_t)fr * C + c0;
    if (WANT_T) {
      if (lane == 0) {
        a.t_out[to] = t[0].x;
        if (has1) a.t_out[to + 1] = t[0].y;
      }
    } else {
      t[0] = v2f{a.t_in[to], has1 ? a.t_in[to + 1] : 0.f};
    }
    wave_sync();
    {   // threshold_frames up to the entries (ac_psy_runs_dev.h)
      const runs::RunsGeo geo = runs::runs_geo(N);
      runs::level_sums<1>(buf, slot, 0, geo.o4, geo.n4, lane);
      wave_sync();
      runs::level_sums<1>(buf, slot, geo.o4, geo.o16, geo.n16, lane);
      if (a.p.n64 > 0) {
        wave_sync();
        runs::level_sums<1>(buf, slot, geo.o16, geo.o64, a.p.n64, lane);
      }
      runs::band_stage<1>(t, a.p, lc, img, buf, slot, lane);
    }
    __syncthreads();   // every slot's intensities and sums are dead: bytes 1536 ... take the threshold row's image
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (runs::in_frame<R>(a.p, i, lane)) {
        const uint32_t wd = IDX_REGS ? ridx(i) : img[runs::off_idx(a.p.lw, a.p.kb) + 64 * i + lane];
        const v4f th = runs::entry_lookup(buf, wd);
        const int v0 = vb + 128 * i * C;
        char* o0 = slots + 1536 + image_at(v0);
        char* o1 = slots + 1536 + image_at(v0 + C);
        if (has1 && codd) {
          *reinterpret_cast<float*>(o0) = th.x;
          *reinterpret_cast<float*>(slots + 1536 + image_at(v0 + 1)) = th.y;
          *reinterpret_cast<float*>(o1) = th.z;
          *reinterpret_cast<float*>(slots + 1536 + image_at(v0 + C + 1)) = th.w;
        } else if (has1) {
          *reinterpret_cast<v2u*>(o0) = v2u{th.x, th.y};
          *reinterpret_cast<v2u*>(o1) = v2u{th.z, th.w};
        } else {
          *reinterpret_cast<float*>(o0) = th.x;
          *reinterpret_cast<float*>(o1) = th.z;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int jj = 64 * i + lane, j = wave * pch + jj;
      if (runs::in_frame<R>(a.p, i, lane) && j < nch)
        __builtin_nontemporal_store(*reinterpret_cast<const v4f*>(buf + 1536 + 16 * jj), reinterpret_cast<v4f*>(a.thr + ro) + j);
    }
    // (the next frame's share lands at the head of the wave's own slot, its image is read after the next barrier)
  }
}

constexpr int runs_fb(int R) { return R >= 8 ? 1 : R == 4 ? 2 : 4; }

template <int R, int CMODE>
int launch_runs_R(const RunsArgs& a, bool want_t, bool want_thr, unsigned grid, int nw, size_t lds, hipStream_t s) {
  const dim3 blk(64 * nw);
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), blk, lds, s, a);
    return AC_OK;
  };
  if (want_t && want_thr) return go(k_psy_runs<R, CMODE, true, true, runs_fb(R)>);
  if (want_thr) return go(k_psy_runs<R, CMODE, false, true, runs_fb(R)>);
  return go(k_psy_runs<R, CMODE, true, false, runs_fb(R)>);
}

int launch_psy_runs(const ac_psy_plan* p, const float* X, const float* t_in, float* t_out, float* thr, float drown, int B, int F,
                    int C, hipStream_t s) {
  RunsArgs a;
  a.X = X;
  a.t_in = t_in;
  a.t_out = t_out;
  a.thr = thr;
  a.img = p->d_runs;
  const int R = mid_r(p->N), fb = runs_fb(R);
  a.p = runs_params(p, drown, R > 8);
  a.C = C;
  a.F = F;
  a.nsig = (long long)B * C;
  a.ntasks = (C == 2 ? (long long)B : C == 1 ? (a.nsig + 1) / 2 : (long long)B * ((C + 1) / 2)) * F;
  const bool want_t = t_out != nullptr, want_thr = thr != nullptr;
  // more than two channels with whole rows, where the shape has a team form and it measured faster (AC_PSY_NOTEAM=1: the
  // strided channel pairs; AC_PSY_TEAM_ALWAYS=1: wherever the shape fits -- tests)
  // (measured, B = 84 ... 170 clips of 10 s, team / strided ms: six channels 960 0.385 / 0.70, 1024 0.377 / 0.71, 512 0.41 / 0.65,
  // 480 0.52 / 0.64, 600 0.44 / 0.66, eight channels 960 0.59 / 0.77, three channels 960 0.46 / 0.58, four channels 640 0.45 / 0.51;
  // not at 4 granule registers with fewer than five channels: 512 x 3 0.50 / 0.47, 300 x 4 0.62 / 0.50)
  const bool team_pays = R >= 8 || (R == 4 && C >= 5 && p->N >= 448);
  if (C > 2 && want_thr && (C + 1) / 2 <= 4 && (R == 4 || R == 8 || R == 16) && (team_pays || getenv("AC_PSY_TEAM_ALWAYS")) &&
      !getenv("AC_PSY_NOTEAM") &&
      !((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(thr)) & 15)) {
    const int CP = (C + 1) / 2;
    // (a slot also holds a share of the threshold row's image behind the entries: 1536 + 8 N bytes; the plan's slot is
    // smaller below filters_n 640)
    const int slot_c = std::max(a.p.slot, (1536 + 8 * p->N + 15) & ~15);
    const size_t lds_c = (size_t)a.p.lds_words * 4 + (size_t)CP * slot_c;
    if (lds_c <= 160 * 1024) {
      RunsArgs c = a;
      c.p.slot = slot_c;
      c.ntasks = (long long)B * F;   // frames: a workgroup takes all the channel pairs of a frame
      const int cus_c = p->cus > 0 ? p->cus : 256;
      int Tc = 8;
      while (Tc > 1 && c.ntasks < (long long)Tc * cus_c * 6) Tc >>= 1;
      c.T = Tc;
      const long long g = (c.ntasks + Tc - 1) / Tc;
      if (g > 2147483647ll) {
        set_error("problem too large for one launch (%lld workgroups)", g);
        return AC_EINVAL;
      }
      auto go = [&](auto kernel) -> int {
        if (lds_c > 64 * 1024)
          AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        hipLaunchKernelGGL(kernel, dim3((unsigned)g), dim3(64 * CP), lds_c, s, c);
        AC_HIP_CHECK(hipGetLastError());
        return AC_OK;
      };
      if (R == 4) return want_t ? go(k_psy_runs_c<4, true>) : go(k_psy_runs_c<4, false>);
      if (R == 8) return want_t ? go(k_psy_runs_c<8, true>) : go(k_psy_runs_c<8, false>);
      return want_t ? go(k_psy_runs_c<16, true>) : go(k_psy_runs_c<16, false>);
    }
  }
  // waves per workgroup: the size that leaves the most waves resident (image + FB slots per wave; 160 KB of LDS per CU, the
  // kernels' register budget allows 12 / 8 / 4 waves per CU)
  const int wave_cap = R >= 32 ? 4 : R >= 16 ? 8 : 12;
  int nw = 4;
  {
    long best = -1;
    for (int w : {4, 3, 2, 1}) {
      const size_t b = (size_t)a.p.lds_words * 4 + (size_t)w * fb * a.p.slot;
      if (b > 160 * 1024) continue;
      const long res = std::min<long>((long)std::min<size_t>(8, 160 * 1024 / b) * w, wave_cap);
      if (res > best) {
        best = res;
        nw = w;
      }
    }
    if (best < 0) {
      set_error("internal: masking-model slots too large for LDS");
      return AC_EUNSUPPORTED;
    }
  }
  const size_t lds = want_thr ? (size_t)a.p.lds_words * 4 + (size_t)nw * fb * a.p.slot : 0;
  const int cus = p->cus > 0 ? p->cus : 256;
  int T = want_thr ? 8 : fb;
  while (T > fb && a.ntasks < (long long)nw * T * cus * 6) T >>= 1;
  a.T = T;
  const long long per = (long long)nw * T;
  const long long g = (a.ntasks + per - 1) / per;
  if (g > 2147483647ll) {
    set_error("problem too large for one launch (%lld workgroups)", g);
    return AC_EINVAL;
  }
  const unsigned grid = (unsigned)g;
  int st;
#define AC_RUNS_R(CM)                                                            \
  (R == 1 ? launch_runs_R<1, CM>(a, want_t, want_thr, grid, nw, lds, s)          \
   : R == 2 ? launch_runs_R<2, CM>(a, want_t, want_thr, grid, nw, lds, s)        \
   : R == 4 ? launch_runs_R<4, CM>(a, want_t, want_thr, grid, nw, lds, s)        \
   : R == 8 ? launch_runs_R<8, CM>(a, want_t, want_thr, grid, nw, lds, s)        \
   : R == 16 ? launch_runs_R<16, CM>(a, want_t, want_thr, grid, nw, lds, s)      \
             : launch_runs_R<32, CM>(a, want_t, want_thr, grid, nw, lds, s))
  if (C > 2) st = AC_RUNS_R(1);
  else if (C == 2) st = AC_RUNS_R(0);
  else st = AC_RUNS_R(2);
#undef AC_RUNS_R
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

}  // namespace

bool mid_psy_supported(const ac_psy_plan* p) { return build_mid(p, nullptr, nullptr); }

mid::MidParams mid_params(const ac_psy_plan* p, float drown) {
  mid::MidParams m;
  m.img_words = p->mid_words;
  m.lds_words = mid_r(p->N) >= AC_MID_WI_GLOBAL_R ? p->mid_off_wi : p->mid_words;   // (off_wi is 16-byte aligned)
  m.N = p->N;
  m.M = p->M;
  m.wi_w = p->mid_wi_w;
  m.off_S = p->mid_off_S;
  m.off_band = p->mid_off_band;
  m.off_wbe = p->mid_off_wbe;
  m.off_wi = p->mid_off_wi;
  m.alpha = (float)p->alpha;
  m.inv_alpha = (float)(1.0 / p->alpha);
  m.drown = drown;
  m.inv_n = 1.0f / (float)p->N;
  return m;
}

int mid_psy_plan_init(ac_psy_plan* p) {
  std::vector<uint32_t> w;
  MidLayout L;
  if (!build_mid(p, &w, &L)) {
    set_error("internal: wave-level masking model (general band layout) not supported for this configuration");
    return AC_EUNSUPPORTED;
  }
  p->mid_words = L.words;
  p->mid_wi_w = L.wi_w;
  p->mid_off_S = L.off_S;
  p->mid_off_band = L.off_band;
  p->mid_off_wbe = L.off_wbe;
  p->mid_off_wi = L.off_wi;
  AC_HIP_CHECK(hipMalloc((void**)&p->d_mid, w.size() * sizeof(uint32_t)));
  AC_HIP_CHECK(hipMemcpy(p->d_mid, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  return AC_OK;
}


// The run-structured image (ac_psy_runs_dev.h).  Host only: nothing here touches the device.
bool build_runs(const PsyTables& t, std::vector<uint32_t>* out, RunsLayout* lay) {
  const int N = t.N, M = t.M;
  if (N < 2 || N > 4096 || (N & 1) || M < 1 || M > 64) return false;
  if ((int)t.g.size() != 2 * M) return false;
  for (int i = 0; i < M; ++i)   // S is Toeplitz by construction; the kernel reads it through the prototype
    for (int j = 0; j < M; ++j)
      if ((float)t.S[(size_t)i * M + j] != (float)t.g[(size_t)(M - i + j)]) return false;
  auto Wf = [&](int f, int j) { return (float)t.W[(size_t)f * M + j]; };
  auto Vf = [&](int j, int f) { return (float)t.W_inv[(size_t)j * N + f]; };
  // bands: one contiguous run of bins, interior weights exactly 1
  std::vector<int> f0(M, -1), f1(M, -1);
  for (int j = 0; j < M; ++j) {
    for (int f = 0; f < N; ++f)
      if (Wf(f, j) != 0.f) {
        if (f0[j] < 0) f0[j] = f;
        f1[j] = f;
      }
    if (f0[j] < 0) return false;
    for (int f = f0[j]; f <= f1[j]; ++f) {
      if (Wf(f, j) == 0.f) return false;
      if (f > f0[j] && f < f1[j] && Wf(f, j) != 1.0f) return false;
    }
  }
  RunsLayout L;
  const runs::RunsGeo geo = runs::runs_geo(N);   // (the kernels derive the same numbers from filter_bands_n)
  auto a16 = [](int v) { return runs::a16(v); };
  L.n4 = geo.n4;
  L.n16 = geo.n16;
  L.o4 = geo.o4;
  L.o16 = geo.o16;
  // interior of band j as aligned runs of 64 (when use64) / 16 / 4 bins and single bins, greedily from the left
  auto list_of = [&](int j, bool use64, int o64, int oz, std::vector<uint32_t>& lst) {
    lst.clear();
    (void)oz;
    for (int f = f0[j] + 1; f <= f1[j] - 1;) {
      if (use64 && (f & 63) == 0 && f + 63 <= f1[j] - 1) {
        lst.push_back((uint32_t)(o64 + 8 * (f >> 6)));
        f += 64;
      } else if ((f & 15) == 0 && f + 15 <= f1[j] - 1) {
        lst.push_back((uint32_t)(L.o16 + 8 * (f >> 4)));
        f += 16;
      } else if ((f & 3) == 0 && f + 3 <= f1[j] - 1) {
        lst.push_back((uint32_t)(L.o4 + 8 * (f >> 2)));
        f += 4;
      } else {
        lst.push_back((uint32_t)(8 * f));
        f += 1;
      }
    }
  };
  const int o64c = geo.o64;
  int lmax[2] = {0, 0};
  std::vector<uint32_t> lst;
  for (int u = 0; u < 2; ++u)
    for (int j = 0; j < M; ++j) {
      list_of(j, u == 1, o64c, 0, lst);
      lmax[u] = std::max(lmax[u], (int)lst.size());
    }
  const bool use64 = lmax[1] + 2 <= lmax[0];   // a fourth level only where it shortens the longest list by a word
  L.n64 = use64 ? N / 64 : 0;
  L.o64 = o64c;
  L.oz = use64 ? a16(L.o64 + 8 * L.n64) : o64c;
  L.slot = a16(L.oz + 8);
  if (L.slot > 65535) return false;
  L.lw = (lmax[use64 ? 1 : 0] + 1) / 2;
  // bins -> entries
  std::vector<int> entry(N, -1), ej0(64, 0), ecnt(64, 0), ebin(64, -1);
  std::vector<float> rho(M, 0.f);
  std::vector<bool> have_rho(M, false);
  int nb = 0, kb = 1;
  for (int f = 0; f < N; ++f) {
    int cnt = 0, jf = -1, jl = -1;
    for (int j = 0; j < M; ++j)
      if (Vf(j, f) != 0.f) {
        if (cnt == 0) jf = j;
        jl = j;
        ++cnt;
      }
    if (cnt == 0 || jl - jf + 1 != cnt) return false;   // (a bin meets a run of consecutive bands)
    if (cnt == 1) {
      const float v = Vf(jf, f);
      if (!have_rho[jf]) {
        rho[jf] = v;
        have_rho[jf] = true;
      } else if (std::fabs(v - rho[jf]) > 1e-6f * rho[jf]) {
        return false;
      }
      entry[f] = 512 + 16 * jf;
    } else {
      if (nb >= 64) return false;
      ej0[nb] = jf;
      ecnt[nb] = cnt;
      ebin[nb] = f;
      kb = std::max(kb, cnt);
      entry[f] = 512 + 16 * nb + 8;
      ++nb;
    }
  }
  if (kb > 64) return false;
  L.kb = kb;
  const int R = mid_r(N);
  L.off_S = runs::OFF_S;
  L.off_bc = runs::OFF_BC;
  L.off_bd = runs::OFF_BD;
  L.off_lst = runs::OFF_LST;
  L.off_bw = runs::off_bw(L.lw);
  L.off_idx = runs::off_idx(L.lw, L.kb);   // (every part a multiple of 64 words: 16-byte aligned)
  static_assert(runs::OFF_BC % 4 == 0, "16-byte rows");
  if (L.slot > runs::runs_slot_max(N)) return false;   // (cannot happen: the compile-time strides of the fused kernels rely on it)
  L.words = (L.off_idx + 64 * R + 3) / 4 * 4;
  std::vector<uint32_t> w((size_t)L.words, 0u);
  auto putf = [&](int i, float v) { uint32_t u; memcpy(&u, &v, 4); w[(size_t)i] = u; };
  {
    auto gp = [&](int k) { const int d = k - 64; return (d > -M && d < M) ? (float)t.g[(size_t)(M + d)] : 0.f; };
    auto bf16_rne = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    auto bf16_val = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    uint16_t* tb = reinterpret_cast<uint16_t*>(w.data() + L.off_S);
    for (int c = 0; c < 4; ++c)
      for (int y = 0; y < 132; ++y) {
        const int m = y - c;
        if (m < 1 || m > 127) continue;
        const float v = gp(128 - m);
        const uint16_t hi = bf16_rne(v);
        tb[(c * MF_COPY_STRIDE) / 2 + y] = hi;
        tb[(MF_TAB_BYTES + c * MF_COPY_STRIDE) / 2 + y] = bf16_rne(v - bf16_val(hi));
      }
  }
  for (int l = 0; l < 64; ++l) {
    const int bc = L.off_bc + 4 * l, bd = L.off_bd + 4 * l;
    if (l < M) {
      const bool two = f1[l] > f0[l];
      w[(size_t)bc] = (uint32_t)(8 * f0[l]) | ((uint32_t)(two ? 8 * f1[l] : L.oz) << 16);
      putf(bc + 1, Wf(f0[l], l));
      putf(bc + 2, two ? Wf(f1[l], l) : 0.f);
      putf(bd + 0, t.beta[l] + 9.0f);
      putf(bd + 1, (float)t.quiet[l]);
      putf(bd + 2, rho[l]);
      list_of(l, use64, L.o64, L.oz, lst);
    } else {
      w[(size_t)bc] = (uint32_t)L.oz | ((uint32_t)L.oz << 16);
      lst.clear();
    }
    lst.resize((size_t)2 * L.lw, (uint32_t)L.oz);
    for (int k = 0; k < L.lw; ++k) w[(size_t)L.off_lst + 64 * k + l] = lst[2 * k] | (lst[2 * k + 1] << 16);
    // edge bin l: its bands' G are read from a window of kb that stays inside the 64 G of the slot
    if (l < nb) {
      const int js = std::min(ej0[l], 64 - L.kb);
      w[(size_t)bc + 3] = (uint32_t)(8 * js);
      for (int k = 0; k < ecnt[l]; ++k) putf(L.off_bw + 64 * (ej0[l] - js + k) + l, Vf(ej0[l] + k, ebin[l]));
    }
  }
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < R; ++i) {
      const int q = 64 * i + l;
      if (2 * q + 1 < N) w[(size_t)L.off_idx + 64 * i + l] = (uint32_t)entry[2 * q] | ((uint32_t)entry[2 * q + 1] << 16);
    }
  if (out) *out = w;
  if (lay) *lay = L;
  return true;
}

bool runs_supported(const ac_psy_plan* p) { return p->runs != 0; }

runs::RunsParams runs_params(const ac_psy_plan* p, float drown, bool idx_in_lds) {
  const RunsLayout& L = p->runs_lay;
  runs::RunsParams m;
  m.N = p->N;
  m.M = p->M;
  m.lds_words = idx_in_lds ? L.words : L.off_idx;   // (off_idx is a multiple of four words: build_runs)
  m.lw = L.lw;
  m.kb = L.kb;
  m.n64 = L.n64;
  m.oz = L.oz;
  m.slot = L.slot;
  m.alpha = (float)p->alpha;
  m.inv_alpha = (float)(1.0 / p->alpha);
  m.omd = 1.0f - drown;
  m.inv_n = 1.0f / (float)p->N;
  return m;
}

int runs_psy_plan_init(ac_psy_plan* p) {
  std::vector<uint32_t> w;
  if (!build_runs(p->host, &w, &p->runs_lay)) return AC_EUNSUPPORTED;   // (not an error: the plan keeps the band walk)
  AC_HIP_CHECK(hipMalloc((void**)&p->d_runs, w.size() * sizeof(uint32_t)));
  AC_HIP_CHECK(hipMemcpy(p->d_runs, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  p->runs = 1;
  return AC_OK;
}

// t_out != null: tonality (from X); thr != null: threshold (from t_out when given, else from t_in)
int launch_psy_mid(const ac_psy_plan* p, const float* X, const float* t_in, float* t_out, float* thr, float drown, int B,
                   int F, int C, hipStream_t s) {
  if (B <= 0 || C <= 0 || F <= 0) return AC_OK;
  static const bool no_runs = getenv("AC_NO_RUNS") != nullptr;   // (A/B hook: the band walk on a plan that has both forms)
  if (p->runs && !no_runs) return launch_psy_runs(p, X, t_in, t_out, thr, drown, B, F, C, s);
  // (the layout of the image was fixed when the plan was built: a launch only fills in arguments)
  MidLayout L;
  L.words = p->mid_words;
  MidArgs a;
  a.X = X;
  a.t_in = t_in;
  a.t_out = t_out;
  a.thr = thr;
  a.img = p->d_mid;
  a.p = mid_params(p, drown);
  a.C = C;
  a.F = F;
  a.nsig = (long long)B * C;
  a.ntasks = (C == 2 ? (long long)B : C == 1 ? (a.nsig + 1) / 2 : (long long)B * ((C + 1) / 2)) * F;   // pairs of rows x frames
  const bool want_t = t_out != nullptr, want_thr = thr != nullptr;
  const int R = mid_r(p->N);
  const int fb = mid_fb(R);
  // four waves per workgroup when the image and the wave buffers fit three workgroups to a CU, else two; frames above
  // 1024 bins (32 KB of intensities per wave at 4096): the workgroup size that leaves the most waves resident
  int nw = 4;
  if (mid_lds_bytes(p->N, a.p.lds_words, nw, fb) > 53 * 1024) nw = 2;
  if (R >= AC_MID_WI_GLOBAL_R) {
    long best = 0;
    for (int w : {4, 2, 1}) {
      const size_t b = mid_lds_bytes(p->N, a.p.lds_words, w, fb);
      const long res = b > 160 * 1024 ? 0 : (long)std::min<size_t>(8, 160 * 1024 / b) * w;
      if (res > best) {
        best = res;
        nw = w;
      }
    }
  }
  const size_t lds = want_thr ? mid_lds_bytes(p->N, a.p.lds_words, nw, fb) : 0;
  if (lds > 160 * 1024) {
    set_error("internal: masking-model tables too large for LDS (%zu bytes)", lds);
    return AC_EUNSUPPORTED;
  }
  // frames per wave: as many as keep every CU supplied with a few workgroups (the image copy is paid per workgroup)
  const int cus = p->cus > 0 ? p->cus : 256;
  int T = want_thr ? 8 : fb;
  while (T > fb && a.ntasks < (long long)nw * T * cus * 6) T >>= 1;
  a.T = T;
  const long long per = (long long)nw * T;
  const long long g = (a.ntasks + per - 1) / per;
  if (g > 2147483647ll) {
    set_error("problem too large for one launch (%lld workgroups)", g);
    return AC_EINVAL;
  }
  const unsigned grid = (unsigned)g;
  int st;
  if (C > 2) st = R == 1 ? launch_mid_R<1, 1>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 2 ? launch_mid_R<2, 1>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 4 ? launch_mid_R<4, 1>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 8 ? launch_mid_R<8, 1>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 16 ? launch_mid_R<16, 1>(a, want_t, want_thr, grid, nw, lds, s)
                           : launch_mid_R<32, 1>(a, want_t, want_thr, grid, nw, lds, s);
  else if (C == 2) st = R == 1 ? launch_mid_R<1, 0>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 2 ? launch_mid_R<2, 0>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 4 ? launch_mid_R<4, 0>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 8 ? launch_mid_R<8, 0>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 16 ? launch_mid_R<16, 0>(a, want_t, want_thr, grid, nw, lds, s)
                           : launch_mid_R<32, 0>(a, want_t, want_thr, grid, nw, lds, s);
  else st = R == 1 ? launch_mid_R<1, 2>(a, want_t, want_thr, grid, nw, lds, s)
            : R == 2 ? launch_mid_R<2, 2>(a, want_t, want_thr, grid, nw, lds, s)
            : R == 4 ? launch_mid_R<4, 2>(a, want_t, want_thr, grid, nw, lds, s)
            : R == 8 ? launch_mid_R<8, 2>(a, want_t, want_thr, grid, nw, lds, s)
            : R == 16 ? launch_mid_R<16, 2>(a, want_t, want_thr, grid, nw, lds, s)
                      : launch_mid_R<32, 2>(a, want_t, want_thr, grid, nw, lds, s);
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

}  // namespace ac
