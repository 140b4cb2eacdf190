// Wave-level masking model for the configurations the fused epilogue of ac_fast.hip does not serve: any even
// filter_bands_n up to 1024 (a frame is up to R = 1, 2, 4 or 8 granule registers per lane, the last ones partly filled
// when filter_bands_n is not a multiple of 128: 960, 576, 480 ...) with any Bark-band count up to 64 and any band layout (a bin may overlap several bands, bands may share
// bins freely) -- e.g. the models beside the several-frames-per-wave MDCT kernels (filters_n 256 / 512), where the
// O(N)-per-workgroup generic kernels ran at 0.5-0.8 TB/s.  gfx950 only.
//
// One 64-lane wave per (frame, channel pair) as in k_psy_fast: the row is loaded with coalesced 16-byte (stereo) or
// 8-byte (mono, two signals side by side) accesses, tonality comes from DPP wave sums, the intensities go through an LDS
// image, lane j < M owns Bark band j and walks its list of (bin, weight) entries (W "by band": psychoacoustic.py:301-315),
// the band x band spreading product reads S from LDS with the Q_i broadcast (psychoacoustic.py:205-207, tonality offset
// pulled out of the sum: SURVEY App. A.3; S is Toeplitz, S[i][j] = g[M - i + j] (:223-228), so the image holds the 2 M
// prototype values instead of M x M), and every bin gathers its <= WI (band, weight) entries of W_inv from a fixed-width
// table stored entry-major (psychoacoustic.py:317-331).  All constant tables sit in one image copied to LDS per
// workgroup; a wave walks T frames so that the copy is paid once per 4 T frames.
#include <algorithm>
#include <cstring>
#include <vector>

#include "ac_internal.h"

namespace ac {
namespace {

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

struct MidArgs {
  const float* X;
  const float* t_in;
  float* t_out;
  float* thr;
  const uint32_t* img;   // ac_psy_plan::d_mid
  int img_words;
  int N, M, C, F;
  int T;                 // frames per wave: workgroup g owns tasks [g nw T, (g + 1) nw T), wave w takes g nw T + w + nw t
  int wi_w;              // entries per bin in the fixed-width W_inv table
  int off_S, off_band, off_wbe, off_wi;   // word offsets inside the image
  float alpha, inv_alpha, drown;
  float inv_n;           // 1 / N
  long long nsig, ntasks;
};

// image layout (32-bit words):
//   off_S:    gp[0 .. 128): S[i][j] = gp[64 + j - i] = g[M - i + j], 0 where |j - i| >= M   (psychoacoustic.py:223-228)
//   off_band: per band j: {first entry, count | first bin << 16, quiet, beta}   4 words
//   off_wbe:  W by band: the weights of the band's bins first bin, first bin + 1, ...  (a band's bins are contiguous: both
//             loads of a step have addresses that depend on nothing loaded before)
//   off_wi:   entry-major: [e < wi_w][bin f] {byte offset of G[band] in the wave's G area, weight}  (weight 0 pads)
// wave buffer: [N] v2f intensities (c0, c1) | [64] v2f Q | [64] v2f G
template <int R, int CMODE, bool WANT_T, bool WANT_THR>
__global__ __launch_bounds__(256, (R == 8 ? 3 : 4)) void k_psy_mid(MidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, half = N >> 1;          // bins per frame (even, <= 128 R), granules per frame
  const int WAVE_BYTES = 8 * N + 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t* img = reinterpret_cast<uint32_t*>(smem);
  if (WANT_THR) {
    for (int i = threadIdx.x; i < a.img_words / 4; i += blockDim.x)
      reinterpret_cast<uint4*>(img)[i] = reinterpret_cast<const uint4*>(a.img)[i];
    __syncthreads();
  }
  char* buf = smem + (size_t)a.img_words * 4 + (size_t)wave * WAVE_BYTES;
  const int C = a.C, M = a.M;
  // the lane's column of the spreading matrix, S[i][lane] = gp[64 + lane - i], in registers for all the wave's frames
  // (the product below always runs over 64 rows: rows beyond the M bands meet Q_i = 0, lanes beyond them are not read)
  v2f Scol[32];   // (S[2 i][lane], S[2 i + 1][lane])
  if (WANT_THR) {
    const float* gp = reinterpret_cast<const float*>(img + a.off_S) + 64 + lane;
#pragma unroll
    for (int i = 0; i < 32; ++i) Scol[i] = v2f{gp[-2 * i], gp[-2 * i - 1]};
  }
  long long task = (long long)blockIdx.x * nw * a.T + wave;
  for (int tt = 0; tt < a.T && task < a.ntasks; ++tt, task += nw) {   // (no workgroup barrier inside)
  wave_sync();   // the previous frame's reads of the wave's buffers are done
  const int f = (int)(task % a.F);
  const long long p = task / a.F;
  // the two signals of the wave: stereo = the two channels of clip p; mono = clips 2 p and 2 p + 1
  const bool has1 = CMODE == 0 ? true : (2 * p + 1 < a.nsig);
  const long long b0 = CMODE == 0 ? p : 2 * p, b1 = CMODE == 0 ? p : (has1 ? 2 * p + 1 : 2 * p);
  const size_t blk = (size_t)N * C;
  const size_t o0 = ((size_t)b0 * a.F + (size_t)f) * blk, o1 = ((size_t)b1 * a.F + (size_t)f) * blk;
  const size_t t0 = ((size_t)b0 * a.F + (size_t)f) * C, t1 = CMODE == 0 ? t0 + 1 : ((size_t)b1 * a.F + (size_t)f) * C;

  // granule q = lane + 64 i: (X[2q], X[2q+1]) x (s0, s1); granules past the frame (q >= N/2) read as zero
  auto in = [&](int i) { return R * 128 == N || 64 * i + lane < half; };
  v4f xq[R];
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) xq[i] = in(i) ? reinterpret_cast<const v4f*>(a.X + o0)[64 * i + lane] : v4f{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const v2f u = in(i) ? reinterpret_cast<const v2f*>(a.X + o0)[64 * i + lane] : v2f{0.f, 0.f};
      xq[i] = v4f{u.x, 0.f, u.y, 0.f};
    }
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const v2f w = in(i) ? reinterpret_cast<const v2f*>(a.X + o1)[64 * i + lane] : v2f{0.f, 0.f};
        xq[i].y = w.x;
        xq[i].w = w.y;
      }
    }
  }
  v2f t = {0.f, 0.f};
  if (WANT_T) {   // psychoacoustic.py:102-120 (the arithmetic of psy_stage in ac_fast.hip)
    v2f slog = {0.f, 0.f}, ssq = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R; ++i) {
      v4f I = xq[i] * xq[i];
      asm("" : "+v"(I));   // the squares stay rounded products (see psy_stage)
      const v2f ie = v2f{I.x, I.y}, io = v2f{I.z, I.w};
      ssq += ie + io;
      const v2f lg = log2v(maxv(ie, kEps) * maxv(io, kEps));
      slog += in(i) ? lg : v2f{0.f, 0.f};
    }
    slog.x = wave_sum(slog.x);
    slog.y = wave_sum(slog.y);
    ssq.x = wave_sum(ssq.x);
    ssq.y = wave_sum(ssq.y);
    const v2f am = ssq * a.inv_n + kEps;
    const v2f sfm = 3.0102999566398120f * (slog * a.inv_n - log2v(am));
    const v2f tt = sfm * (-1.0f / 60.0f);
    t = v2f{fminf(tt.x, 1.0f), fminf(tt.y, 1.0f)};
    if (lane == 0) {
      a.t_out[t0] = t.x;
      if (has1) a.t_out[t1] = t.y;
    }
  } else {
    t.x = a.t_in[t0];
    t.y = has1 ? a.t_in[t1] : 0.f;
  }
  if (!WANT_THR) continue;

  // intensities in natural order: bin f at byte 8 f (c0, c1)
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (in(i)) *reinterpret_cast<v4f*>(buf + 16 * (64 * i + lane)) = xq[i] * xq[i];
  wave_sync();
  v2f* Qb = reinterpret_cast<v2f*>(buf + 8 * N);
  v2f* Gb = Qb + 64;
  const uint32_t* band = img + a.off_band;
  float quiet = 0.f, beta = 0.f;
  if (lane < M) {   // P_j = sum_f I_f W[f, j]  (:312-313)
    const uint4 bw = reinterpret_cast<const uint4*>(band)[lane];
    quiet = __uint_as_float(bw.z);
    beta = __uint_as_float(bw.w);
    const float* wt = reinterpret_cast<const float*>(img + a.off_wbe) + bw.x;
    const v2f* Ib = reinterpret_cast<const v2f*>(buf) + (bw.y >> 16);
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
  v4f th[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int q = 64 * i + lane;
    v2f s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
    for (int e = 0; e < W && in(i); ++e) {
      const uint4 en = wi[(size_t)e * half + q];
      s0 += *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(Gb) + en.x) * __uint_as_float(en.y);
      s1 += *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(Gb) + en.z) * __uint_as_float(en.w);
    }
    s0 = maxv(s0, kEps);
    s1 = maxv(s1, kEps);
    th[i] = v4f{__builtin_amdgcn_sqrtf(s0.x), __builtin_amdgcn_sqrtf(s0.y), __builtin_amdgcn_sqrtf(s1.x), __builtin_amdgcn_sqrtf(s1.y)};
  }
  if (CMODE == 0) {
#pragma unroll
    for (int i = 0; i < R; ++i)
      if (in(i)) __builtin_nontemporal_store(th[i], reinterpret_cast<v4f*>(a.thr + o0) + 64 * i + lane);
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i)
      if (in(i)) reinterpret_cast<v2f*>(a.thr + o0)[64 * i + lane] = v2f{th[i].x, th[i].z};
    if (has1) {
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (in(i)) reinterpret_cast<v2f*>(a.thr + o1)[64 * i + lane] = v2f{th[i].y, th[i].w};
    }
  }
  }   // frames of the wave
}

struct MidLayout {
  int wi_w = 0, off_S = 0, off_band = 0, off_wbe = 0, off_wi = 0, words = 0;
};

bool build_mid(const ac_psy_plan* p, std::vector<uint32_t>* out, MidLayout* lay) {
  const PsyTables& t = p->host;
  const int N = t.N, M = t.M;
  if (N < 2 || N > 1024 || (N & 1) || M < 1 || M > 64) return false;
  SparseRows wb, wi;
  w_by_band(t, wb);
  winv_by_bin(t, wi);
  if (wi.max_row < 1 || wi.max_row > 6) return false;
  MidLayout L;
  L.wi_w = wi.max_row;
  L.off_S = 0;
  L.off_band = L.off_S + 128;
  L.off_band = (L.off_band + 3) / 4 * 4;                  // 16-byte aligned rows of four words
  L.off_wbe = L.off_band + 4 * 64;
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
    total += count[j];
  }
  if (total > 4 * N) return false;   // (bands that each span most of the spectrum: not a Bark mapping; other tiers)
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
  for (int d = -(M - 1); d <= M - 1; ++d) putf(L.off_S + 64 + d, (float)t.g[(size_t)(M + d)]);
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

size_t mid_lds_bytes(int N, int words, int nw) { return (size_t)words * 4 + (size_t)nw * (8 * (size_t)N + 1024); }

template <int R, int CMODE>
int launch_mid_R(const MidArgs& a, bool want_t, bool want_thr, unsigned grid, int nw, size_t lds, hipStream_t s) {
  const dim3 blk(64 * nw);
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      AC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), blk, lds, s, a);
    return AC_OK;
  };
  if (want_t && want_thr) return go(k_psy_mid<R, CMODE, true, true>);
  if (want_thr) return go(k_psy_mid<R, CMODE, false, true>);
  return go(k_psy_mid<R, CMODE, true, false>);
}

}  // namespace

bool mid_psy_supported(const ac_psy_plan* p) { return build_mid(p, nullptr, nullptr); }

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

// t_out != null: tonality (from X); thr != null: threshold (from t_out when given, else from t_in)
int launch_psy_mid(const ac_psy_plan* p, const float* X, const float* t_in, float* t_out, float* thr, float drown, int B,
                   int F, int C, hipStream_t s) {
  if (B <= 0 || C <= 0 || F <= 0) return AC_OK;
  if (C != 1 && C != 2) {
    set_error("internal: the wave-level masking model serves mono and stereo tensors");
    return AC_EUNSUPPORTED;
  }
  // (the layout of the image was fixed when the plan was built: a launch only fills in arguments)
  MidLayout L;
  L.words = p->mid_words;
  L.wi_w = p->mid_wi_w;
  L.off_S = p->mid_off_S;
  L.off_band = p->mid_off_band;
  L.off_wbe = p->mid_off_wbe;
  L.off_wi = p->mid_off_wi;
  MidArgs a;
  a.X = X;
  a.t_in = t_in;
  a.t_out = t_out;
  a.thr = thr;
  a.img = p->d_mid;
  a.img_words = L.words;
  a.N = p->N;
  a.M = p->M;
  a.C = C;
  a.F = F;
  a.wi_w = L.wi_w;
  a.off_S = L.off_S;
  a.off_band = L.off_band;
  a.off_wbe = L.off_wbe;
  a.off_wi = L.off_wi;
  a.alpha = (float)p->alpha;
  a.inv_alpha = (float)(1.0 / p->alpha);
  a.drown = drown;
  a.inv_n = 1.0f / (float)p->N;
  a.nsig = (long long)B * C;
  a.ntasks = ((C == 2) ? (long long)B : (a.nsig + 1) / 2) * F;
  const bool want_t = t_out != nullptr, want_thr = thr != nullptr;
  // four waves per workgroup when the image and the wave buffers fit three workgroups to a CU, else two
  int nw = 4;
  if (mid_lds_bytes(p->N, L.words, nw) > 53 * 1024) nw = 2;
  const size_t lds = want_thr ? mid_lds_bytes(p->N, L.words, nw) : 0;
  if (lds > 160 * 1024) {
    set_error("internal: masking-model tables too large for LDS (%zu bytes)", lds);
    return AC_EUNSUPPORTED;
  }
  // frames per wave: as many as keep every CU supplied with a few workgroups (the image copy is paid per workgroup)
  const int cus = p->cus > 0 ? p->cus : 256;
  int T = want_thr ? 8 : 1;
  while (T > 1 && a.ntasks < (long long)nw * T * cus * 6) T >>= 1;
  a.T = T;
  const long long per = (long long)nw * T;
  const long long g = (a.ntasks + per - 1) / per;
  if (g > 2147483647ll) {
    set_error("problem too large for one launch (%lld workgroups)", g);
    return AC_EINVAL;
  }
  const unsigned grid = (unsigned)g;
  int st;
  const int R = p->N <= 128 ? 1 : p->N <= 256 ? 2 : p->N <= 512 ? 4 : 8;   // granule registers per lane
  if (C == 2) st = R == 1 ? launch_mid_R<1, 0>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 2 ? launch_mid_R<2, 0>(a, want_t, want_thr, grid, nw, lds, s)
                 : R == 4 ? launch_mid_R<4, 0>(a, want_t, want_thr, grid, nw, lds, s)
                          : launch_mid_R<8, 0>(a, want_t, want_thr, grid, nw, lds, s);
  else st = R == 1 ? launch_mid_R<1, 2>(a, want_t, want_thr, grid, nw, lds, s)
            : R == 2 ? launch_mid_R<2, 2>(a, want_t, want_thr, grid, nw, lds, s)
            : R == 4 ? launch_mid_R<4, 2>(a, want_t, want_thr, grid, nw, lds, s)
                     : launch_mid_R<8, 2>(a, want_t, want_thr, grid, nw, lds, s);
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

}  // namespace ac
