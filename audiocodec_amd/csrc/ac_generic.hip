// Generic kernels: any even N, any channel count, any Bark-band count.  O(N^2) direct DCT-IV.
// They are the path for sizes the wave-level FFT kernels (ac_fast.hip) do not cover and an
// independent on-device cross-check for them.  gfx950 only.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "ac_internal.h"
#include "ac_psy_runs_dev.h"

namespace ac {

static constexpr int kThreads = 256;
static constexpr float kEps = 1e-14f;   // _INTENSITY_EPS, psychoacoustic.py:56

// ---- compute_dtype variants: the kernels below are templates over the storage type TIO of the
// tensors -- float (the primary path), double (everything in fp64, constants included: the on-device oracle) and
// bfloat16 (storage only: arithmetic in float32).  TC = the type of the arithmetic and of the constant tables.
template <typename TIO> struct Compute { using type = float; };
template <> struct Compute<double> { using type = double; };
__device__ __forceinline__ float ldv(const float* p) { return *p; }
__device__ __forceinline__ double ldv(const double* p) { return *p; }
__device__ __forceinline__ float ldv(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ float ldv(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv(double* p, double v) { *p = v; }
__device__ __forceinline__ void stv(bf16_t* p, float v) { *p = (bf16_t)v; }   // round to nearest even
__device__ __forceinline__ void stv(f16_t* p, float v) { *p = (f16_t)v; }     // round to nearest even (overflow: infinity, as a cast)
extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

// ------------------------------------------------------------------------------------------------
// analysis: fold (mdctransformer.py:118,349-368 in closed form) + DCT-IV (:311-347) + scale (:125)
// one workgroup per (signal = b*C + c, frame n)
// ------------------------------------------------------------------------------------------------
template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(kThreads) void k_fwd_generic(const TIO* __restrict__ x, TIO* __restrict__ X,
                                                          const TIO* __restrict__ prev_block,
                                                          const TC* __restrict__ coef,
                                                          const TC* __restrict__ ctab, int Kin, int F, int C,
                                                          int N) {
  TC* v = reinterpret_cast<TC*>(smem_raw);  // [N]
  const int h = N >> 1;
  const long long wg = blockIdx.x;
  const int n = (int)(wg % F);
  const long long sig = wg / F;
  const int c = (int)(sig % C);
  const long long b = sig / C;
  const TC* a1 = coef;
  const TC* a2 = coef + h;
  const TC* a3 = coef + 2 * h;
  const TC* a4 = coef + 3 * h;

  const bool has_cur = n < Kin;
  const TIO* xc = x + ((size_t)b * Kin + (size_t)n) * N * C + c;              // block n
  const TIO* xp = nullptr;                                                     // block n-1
  if (n >= 1) xp = x + ((size_t)b * Kin + (size_t)(n - 1)) * N * C + c;
  else if (prev_block) xp = prev_block + (size_t)b * N * C + c;

  for (int j = threadIdx.x; j < h; j += kThreads) {
    TC vc = 0, vp = 0;
    if (has_cur) vc = a1[j] * ldv(xc + (size_t)j * C) + a2[j] * ldv(xc + (size_t)(N - 1 - j) * C);
    if (xp) vp = a3[j] * ldv(xp + (size_t)(h - 1 - j) * C) + a4[j] * ldv(xp + (size_t)(h + j) * C);
    v[h + j] = vc;
    v[j] = vp;
  }
  __syncthreads();

  const unsigned mod = 8u * (unsigned)N;
  const double scale = 1.0 / ((double)N * 1.4142135623730951);   // 1/sqrt(4N) * sqrt(2/N)
  TIO* Xo = X + (((size_t)b * F + (size_t)n) * N) * C + c;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const unsigned step = (unsigned)((2ull * (2ull * k + 1ull)) % mod);
    unsigned idx = (unsigned)((2ull * k + 1ull) % mod);   // (2m+1)(2k+1) at m = 0
    double acc = 0.0;
    for (int m = 0; m < N; ++m) {
      acc += (double)v[m] * (double)ctab[idx];
      idx += step;
      if (idx >= mod) idx -= mod;
    }
    stv(Xo + (size_t)k * C, (TC)(acc * scale));
  }
}

// ------------------------------------------------------------------------------------------------
// synthesis: scale (mdctransformer.py:145) + DCT-IV + unfold/overlap-add (:148 in closed form)
// one workgroup per (signal, output block n); block n = nblk only writes the new stream state
// ------------------------------------------------------------------------------------------------
template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(kThreads) void k_inv_generic(const TIO* __restrict__ X, TIO* __restrict__ x,
                                                          const TC* __restrict__ tail_in,
                                                          TC* __restrict__ tail_out,
                                                          const TC* __restrict__ coef,
                                                          const TC* __restrict__ ctab, int Kp, int nblk,
                                                          int nwg_per_sig, int C, int N) {
  TC* Xn = reinterpret_cast<TC*>(smem_raw);   // [N] frame n
  TC* Xm = Xn + N;                            // [N] frame n-1
  const int h = N >> 1;
  const long long wg = blockIdx.x;
  const int n = (int)(wg % nwg_per_sig);
  const long long sig = wg / nwg_per_sig;
  const int c = (int)(sig % C);
  const long long b = sig / C;
  const TC* s1 = coef + 4 * h;
  const TC* s2 = coef + 5 * h;
  const TC* s3 = coef + 6 * h;
  const TC* s4 = coef + 7 * h;

  const bool has_n = n < Kp && n < nblk;   // the virtual state block (n == nblk) has no current frame
  const bool has_m = n >= 1;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    Xn[k] = has_n ? (TC)ldv(X + (((size_t)b * Kp + (size_t)n) * N + k) * C + c) : (TC)0;
    Xm[k] = has_m ? (TC)ldv(X + (((size_t)b * Kp + (size_t)(n - 1)) * N + k) * C + c) : (TC)0;
  }
  __syncthreads();

  const unsigned mod = 8u * (unsigned)N;
  const double scale = 2.0 * 1.4142135623730951;   // sqrt(4N) * sqrt(2/N)
  for (int j = threadIdx.x; j < h; j += kThreads) {
    double a = 0.0, bb = 0.0;
    if (has_n) {   // u_n[h-1-j]
      const unsigned long long mm = 2ull * (unsigned)(h - 1 - j) + 1ull;
      const unsigned step = (unsigned)((2ull * mm) % mod);
      unsigned idx = (unsigned)(mm % mod);
      for (int k = 0; k < N; ++k) {
        a += (double)Xn[k] * (double)ctab[idx];
        idx += step;
        if (idx >= mod) idx -= mod;
      }
      a *= scale;
    }
    if (has_m) {   // u_{n-1}[h+j]
      const unsigned long long mm = 2ull * (unsigned)(h + j) + 1ull;
      const unsigned step = (unsigned)((2ull * mm) % mod);
      unsigned idx = (unsigned)(mm % mod);
      for (int k = 0; k < N; ++k) {
        bb += (double)Xm[k] * (double)ctab[idx];
        idx += step;
        if (idx >= mod) idx -= mod;
      }
      bb *= scale;
    } else if (tail_in) {
      bb = (double)tail_in[((size_t)b * C + c) * h + j];
    }
    if (n < nblk) {
      TIO* xo = x + (((size_t)b * nblk + (size_t)n) * N) * C + c;
      stv(xo + (size_t)j * C, (TC)((double)s1[j] * a + (double)s2[j] * bb));
      stv(xo + (size_t)(N - 1 - j) * C, (TC)((double)s3[j] * a + (double)s4[j] * bb));
    } else if (tail_out) {
      tail_out[((size_t)b * C + c) * h + j] = (TC)bb;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Middle tier: any filters_n from 16 to 4096 whose half is 5-smooth (2^a 3^b 5^c: the powers of two, and the 120 / 240 /
// 480 / 960 and 192 / 576 families of the speech and music codecs) that the wave-level kernels do not serve, with any
// window, the rectangular one included.  Same O(N) fold / unfold as above, the DCT-IV as an N/2-point complex FFT in LDS
// (mixed-radix Stockham -- radix 4 while it divides, then 2, 3, 5 -- fp32), one group of threads per (signal, frame).
// ------------------------------------------------------------------------------------------------
// Two channels of a clip ride side by side (c0, c0 + 1; the last one alone when C is odd): every value is a float2
// over the pair, a complex value a cpair.
__device__ __forceinline__ float2 cis_neg(const float* __restrict__ ctab, int idx, int N) {
  // exp(-i pi idx / (4 N)), 0 <= idx < 8 N, from ctab[i] = cos(pi i / (4 N)):  sin(x) = cos(x - pi/2)
  const int s = idx - 2 * N;
  return make_float2(ctab[idx], -ctab[s < 0 ? -s : s]);
}
struct alignas(16) cpair {   // (16-byte aligned: one ds_read_b128 / ds_write_b128 per value)
  float2 re, im;   // (c0, c1)
};
// (the two channels of a pair as one 2-vector: the compiler then emits packed v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 --
// half the vector-ALU instructions of the same arithmetic written on .x / .y)
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk(float2 v) { return pk2{v.x, v.y}; }
__device__ __forceinline__ float2 unpk(pk2 v) { return make_float2(v.x, v.y); }
__device__ __forceinline__ cpair cmulw(cpair a, float2 w) {
  const pk2 re = pk(a.re), im = pk(a.im);
  cpair r;
  r.re = unpk(re * w.x - im * w.y);
  r.im = unpk(re * w.y + im * w.x);
  return r;
}
__device__ __forceinline__ float2 ld2(const float* p, int C, bool has1) {   // the pair's two samples at one index
  if (C == 2) return *reinterpret_cast<const float2*>(p);                     // stereo: one 8-byte access
  return make_float2(p[0], has1 ? p[1] : 0.f);
}
__device__ __forceinline__ void st2(float* p, float2 v, int C, bool has1) {
  if (C == 2) {
    *reinterpret_cast<float2*>(p) = v;
    return;
  }
  p[0] = v.x;
  if (has1) p[1] = v.y;
}
// bfloat16 storage: the stereo pair is one 4-byte access
typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld2(const bf16_t* p, int C, bool has1) {
  if (C == 2) {
    const f2v v = __builtin_convertvector(*reinterpret_cast<const bf16x2_t*>(p), f2v);
    return make_float2(v.x, v.y);
  }
  return make_float2((float)p[0], has1 ? (float)p[1] : 0.f);
}
__device__ __forceinline__ void st2(bf16_t* p, float2 v, int C, bool has1) {
  if (C == 2) {
    *reinterpret_cast<bf16x2_t*>(p) = __builtin_convertvector(f2v{v.x, v.y}, bf16x2_t);
    return;
  }
  p[0] = (bf16_t)v.x;
  if (has1) p[1] = (bf16_t)v.y;
}
// float16 storage (MDCTransformer(compute_dtype=float16): mdctransformer.py:327-344 up-casts such tensors to float32 inside
// the DCT-IV; here all the arithmetic is float32)
typedef f16_t f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld2(const f16_t* p, int C, bool has1) {
  if (C == 2) {
    const f2v v = __builtin_convertvector(*reinterpret_cast<const f16x2_t*>(p), f2v);
    return make_float2(v.x, v.y);
  }
  return make_float2((float)p[0], has1 ? (float)p[1] : 0.f);
}
__device__ __forceinline__ void st2(f16_t* p, float2 v, int C, bool has1) {
  if (C == 2) {
    *reinterpret_cast<f16x2_t*>(p) = __builtin_convertvector(f2v{v.x, v.y}, f16x2_t);
    return;
  }
  p[0] = (f16_t)v.x;
  if (has1) p[1] = (f16_t)v.y;
}

__device__ __forceinline__ cpair cadd(cpair a, cpair b) { return {unpk(pk(a.re) + pk(b.re)), unpk(pk(a.im) + pk(b.im))}; }
__device__ __forceinline__ cpair csub(cpair a, cpair b) { return {unpk(pk(a.re) - pk(b.re)), unpk(pk(a.im) - pk(b.im))}; }
__device__ __forceinline__ cpair cscale(cpair a, float s) { return {unpk(pk(a.re) * s), unpk(pk(a.im) * s)}; }
__device__ __forceinline__ cpair cmul_mi(cpair a) {   // a * (-i)
  return {a.im, unpk(-pk(a.re))};
}
// radix of the next Stockham pass over what is left of the transform length (4 while it divides, then 2, 3, 5)
static inline __host__ __device__ int next_radix(int rem) { return rem % 4 == 0 ? 4 : rem % 2 == 0 ? 2 : rem % 3 == 0 ? 3 : 5; }

// v[N] (LDS, float2 per entry) -> y[k] = sum_m v[m] cos(pi/N (m + 1/2)(k + 1/2)) written back into v;
// A, B: N/2 cpairs each (LDS), tw[k] = exp(-2 pi i k / (N/2)), k < N/2.  Executed by a group of nt threads (tid = index
// inside the group); every group of the workgroup runs it at the same time on its own buffers (the barriers are
// workgroup-wide).  Stockham autosort, decimation in time: a pass of radix r joins r transforms of length L into one of
// length r L -- butterfly j = (p, q), q < L: x_s = src[q + L (p + s m)] W_{rL}^{q s}, m = H / (r L);
// dst[q + L (r p + t)] = sum_s x_s w_r^{s t}.
// B may BE v (the analysis kernel at filters_n > 2048, where a third buffer would leave one workgroup per CU): the
// pre-twiddled input then goes through registers into the buffer from which the passes end in A, so that the last step
// reads A and writes v.
// (ALIAS is a template parameter, not a run-time test: the staging registers of the aliased form cost the other one
// a third of its speed when both share a body)
constexpr int kAliasPerThread = 8;   // N/2 values over 256 threads, N <= 4096
template <bool ALIAS = false>
__device__ void dct4_lds(float2* v, cpair* A, cpair* B, const float* __restrict__ ctab, const float2* __restrict__ tw,
                         int N, int tid, int nt) {
  const int H = N >> 1;
  cpair* src = A;
  cpair* dst = B;
  if constexpr (ALIAS) {
    int passes = 0;
    for (int rem0 = H; rem0 > 1; rem0 /= next_radix(rem0)) ++passes;
    if (passes & 1) {   // an odd number of passes starts in B (= v) and ends in A
      src = B;
      dst = A;
    }
    cpair held[kAliasPerThread];
#pragma unroll
    for (int i = 0; i < kAliasPerThread; ++i) {
      const int n = tid + i * nt;
      if (n < H) {
        cpair t;
        t.re = v[2 * n];
        t.im = v[N - 1 - 2 * n];
        held[i] = cmulw(t, cis_neg(ctab, 4 * n + 1, N));
      }
    }
    __syncthreads();   // every value of v has been read: its bytes may now serve as B
#pragma unroll
    for (int i = 0; i < kAliasPerThread; ++i) {
      const int n = tid + i * nt;
      if (n < H) src[n] = held[i];
    }
  } else {
    for (int n = tid; n < H; n += nt) {
      cpair t;
      t.re = v[2 * n];
      t.im = v[N - 1 - 2 * n];
      A[n] = cmulw(t, cis_neg(ctab, 4 * n + 1, N));   // exp(-i pi (n + 1/4) / N)
    }
  }
  __syncthreads();
  int rem = H;
  for (int L = 1; L < H;) {
    const int r = next_radix(rem);
    const int m = H / (r * L);   // exp(-2 pi i q s / (r L)) = exp(-2 pi i (q s m) / H) = tw[q s m]
    for (int j = tid; j < H / r; j += nt) {
      const int p = j / L, q = j - p * L;
      const cpair* in = src + q + L * p;
      cpair* out = dst + q + L * r * p;
      const cpair x0 = in[0];
      if (r == 4) {
        const cpair x1 = cmulw(in[L * m], tw[q * m]), x2 = cmulw(in[2 * L * m], tw[2 * q * m]);
        const cpair x3 = cmulw(in[3 * L * m], tw[3 * q * m]);
        const cpair t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = cmul_mi(csub(x1, x3));
        out[0] = cadd(t0, t2);
        out[L] = cadd(t1, t3);
        out[2 * L] = csub(t0, t2);
        out[3 * L] = csub(t1, t3);
      } else if (r == 2) {
        const cpair x1 = cmulw(in[L * m], tw[q * m]);
        out[0] = cadd(x0, x1);
        out[L] = csub(x0, x1);
      } else if (r == 3) {
        const cpair x1 = cmulw(in[L * m], tw[q * m]), x2 = cmulw(in[2 * L * m], tw[2 * q * m]);
        const cpair sm = cadd(x1, x2), m1 = csub(x0, cscale(sm, 0.5f));
        const cpair m2 = cscale(cmul_mi(csub(x1, x2)), 0.86602540378443865f);   // -i sin(2 pi / 3) (x1 - x2)
        out[0] = cadd(x0, sm);
        out[L] = cadd(m1, m2);
        out[2 * L] = csub(m1, m2);
      } else {   // 5
        const cpair x1 = cmulw(in[L * m], tw[q * m]), x2 = cmulw(in[2 * L * m], tw[2 * q * m]);
        const cpair x3 = cmulw(in[3 * L * m], tw[3 * q * m]), x4 = cmulw(in[4 * L * m], tw[4 * q * m]);
        const cpair a1 = cadd(x1, x4), a2 = cadd(x2, x3), b1 = csub(x1, x4), b2 = csub(x2, x3);
        constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;   // cos(2 pi / 5), cos(4 pi / 5)
        constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;    // sin(2 pi / 5), sin(4 pi / 5)
        const cpair e1 = cadd(x0, cadd(cscale(a1, c1), cscale(a2, c2))), e2 = cadd(x0, cadd(cscale(a1, c2), cscale(a2, c1)));
        const cpair d1 = cmul_mi(cadd(cscale(b1, s1), cscale(b2, s2))), d2 = cmul_mi(csub(cscale(b1, s2), cscale(b2, s1)));
        out[0] = cadd(x0, cadd(a1, a2));
        out[L] = cadd(e1, d1);
        out[2 * L] = cadd(e2, d2);
        out[3 * L] = csub(e2, d2);
        out[4 * L] = csub(e1, d1);
      }
    }
    __syncthreads();
    cpair* t = src;
    src = dst;
    dst = t;
    L *= r;
    rem /= r;
  }
  for (int k = tid; k < H; k += nt) {
    const cpair r = cmulw(src[k], cis_neg(ctab, 4 * k, N));   // exp(-i pi k / N)
    v[2 * k] = r.re;
    v[N - 1 - 2 * k] = make_float2(-r.im.x, -r.im.y);
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// The same transform with one group of <= 64 lanes INSIDE ONE WAVE per frame (filters_n <= 2048): no workgroup barrier
// anywhere in a frame (LDS operations of a wave execute in order; wave_sync only pins the compiler), two or three radix
// stages per LDS round trip -- a pass has a super-radix R = R1 R2 <= 16 (16 = 4 x 4, 15 = 3 x 5, 12 = 4 x 3, 10 = 2 x 5,
// 9 = 3 x 3, 8 = 4 x 2, 6 = 2 x 3, or a plain 5 / 4 / 3 / 2), computed in registers with compile-time inner twiddles, so
// filters_n = 960 takes 3 round trips instead of 5, 480 two -- and padded buffers: element i of a buffer lives at
// i + (i >> 4), which spreads the stride-R writes of the first pass (and every other power-of-two stride) over the banks.
// The fold buffer shares the bytes of the second FFT buffer: 17 N bytes of LDS per frame, 9 frames resident per CU at
// filters_n = 960 (the three-buffer workgroup form above: 6).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// one 16-byte element of padding per 2^ps elements: ps = 4, but 2 at the two sizes where a survey of ps = 2 ... 5 over every
// instance (B = 64 stereo) found another value faster on a second look (120: 0.106 / 0.095 -> 0.089 / 0.087 ms, 36: +5 %;
// four other sizes of the first pass were noise)
#define AC_PAD_SHIFT 4
static inline __host__ __device__ constexpr int pad_shift_ct(int N) { return N == 120 || N == 36 ? 2 : AC_PAD_SHIFT; }
__device__ __forceinline__ int pad16(int i, int ps = AC_PAD_SHIFT) { return i + (i >> ps); }
static inline __host__ __device__ constexpr int padded_len(int n, int ps = AC_PAD_SHIFT) { return n + (n >> ps) + 1; }

// compile-time cos / sin of 2 pi e / R (Taylor series on the angle reduced to [-pi, pi])
constexpr double c_series(double x, bool sine) {
  double term = sine ? x : 1.0, sum = term;
  for (int k = 1; k < 16; ++k) {
    const double a = sine ? 2.0 * k : 2.0 * k - 1.0;
    term *= -x * x / (a * (a + 1.0));
    sum += term;
  }
  return sum;
}
constexpr double c_angle(int e, int R) {
  const int m = ((e % R) + R) % R;
  const double a = 6.283185307179586476925 * (double)m / (double)R;
  return a > 3.14159265358979323846 ? a - 6.283185307179586476925 : a;
}
template <int E, int R> struct Wc {   // exp(-2 pi i E / R)
  static constexpr float re = (float)c_series(c_angle(E, R), false);
  static constexpr float im = (float)(-c_series(c_angle(E, R), true));
};

// small DFTs (forward sign) on R cpairs in natural order, in place
template <int R> __device__ __forceinline__ void dft_small(cpair* a);
template <> __device__ __forceinline__ void dft_small<1>(cpair*) {}
template <> __device__ __forceinline__ void dft_small<2>(cpair* a) {
  const cpair s = cadd(a[0], a[1]), d = csub(a[0], a[1]);
  a[0] = s;
  a[1] = d;
}
template <> __device__ __forceinline__ void dft_small<3>(cpair* a) {
  const cpair sm = cadd(a[1], a[2]), m1 = csub(a[0], cscale(sm, 0.5f));
  const cpair m2 = cscale(cmul_mi(csub(a[1], a[2])), 0.86602540378443865f);   // -i sin(2 pi / 3) (x1 - x2)
  a[0] = cadd(a[0], sm);
  a[1] = cadd(m1, m2);
  a[2] = csub(m1, m2);
}
template <> __device__ __forceinline__ void dft_small<4>(cpair* a) {
  const cpair t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]), t2 = cadd(a[1], a[3]), t3 = cmul_mi(csub(a[1], a[3]));
  a[0] = cadd(t0, t2);
  a[1] = cadd(t1, t3);
  a[2] = csub(t0, t2);
  a[3] = csub(t1, t3);
}
template <> __device__ __forceinline__ void dft_small<5>(cpair* a) {
  const cpair a1 = cadd(a[1], a[4]), a2 = cadd(a[2], a[3]), b1 = csub(a[1], a[4]), b2 = csub(a[2], a[3]);
  constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;   // cos(2 pi / 5), cos(4 pi / 5)
  constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;    // sin(2 pi / 5), sin(4 pi / 5)
  const cpair e1 = cadd(a[0], cadd(cscale(a1, c1), cscale(a2, c2))), e2 = cadd(a[0], cadd(cscale(a1, c2), cscale(a2, c1)));
  const cpair d1 = cmul_mi(cadd(cscale(b1, s1), cscale(b2, s2))), d2 = cmul_mi(csub(cscale(b1, s2), cscale(b2, s1)));
  a[0] = cadd(a[0], cadd(a1, a2));
  a[1] = cadd(e1, d1);
  a[2] = cadd(e2, d2);
  a[3] = csub(e2, d2);
  a[4] = csub(e1, d1);
}
// inner twiddles W_R^(n2 k1) of the two-stage form, applied row by row with compile-time constants
template <int R1, int R2, int N2, int K1>
struct TwRow {
  static __device__ __forceinline__ void run(cpair* g) {   // g[k1], k1 = 0 .. R1 - 1, for a fixed n2 = N2
    TwRow<R1, R2, N2, K1 - 1>::run(g);
    if constexpr (K1 > 0 && N2 > 0) g[K1] = cmulw(g[K1], make_float2(Wc<N2 * K1, R1 * R2>::re, Wc<N2 * K1, R1 * R2>::im));
  }
};
template <int R1, int R2, int N2>
struct TwRow<R1, R2, N2, -1> {
  static __device__ __forceinline__ void run(cpair*) {}
};
template <int R1, int R2, int N2>
struct Stage1 {   // for n2 = 0 .. N2: DFT_R1 over n1 of x[n1 R2 + n2] (fetched by `load`), times W_R^(n2 k1) -> y[k1 R2 + n2]
  template <class LOAD>
  static __device__ __forceinline__ void run(const LOAD& load, cpair* y) {
    Stage1<R1, R2, N2 - 1>::run(load, y);
    cpair g[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) g[n1] = load(n1 * R2 + N2);
    dft_small<R1>(g);
    TwRow<R1, R2, N2, R1 - 1>::run(g);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) y[k1 * R2 + N2] = g[k1];
  }
};
template <int R1, int R2>
struct Stage1<R1, R2, -1> {
  template <class LOAD>
  static __device__ __forceinline__ void run(const LOAD&, cpair*) {}
};

// one Stockham pass of super-radix R = R1 R2 joining R transforms of length L (see dct4_lds), by the nt <= 64 lanes of a group
// inside one wave; src / dst padded (pad16).  The R-point DFT runs in registers as R2 DFTs of R1 points (inputs fetched column
// by column), compile-time inner twiddles, R1 DFTs of R2 points (outputs stored row by row):  n = n1 R2 + n2,  k = k1 + R1 k2.
// first: the inputs are the folded frame v itself, element n = v[2 n] + i v[N - 1 - 2 n], times the pre-twiddle
// exp(-i pi (n + 1/4) / N) (no separate pre-twiddle round trip); last_to_v: the outputs go to v in their final form,
// y[2 k] = Re, y[N - 1 - 2 k] = -Im of out[k] exp(-i pi k / N) (no separate post-twiddle round trip).
struct WaveTabs {
  const float2* tw;    // exp(-2 pi i k / (N/2))
  const float2* pre;   // exp(-i pi (n + 1/4) / N)
  const float2* post;  // exp(-i pi k / N)
};
template <int R1, int R2, bool first, bool last_to_v, bool CT = false>
__device__ __forceinline__ void wave_pass(const cpair* __restrict__ src, cpair* __restrict__ dst, float2* v, int N, int L, int H,
                                          const WaveTabs& tb, int tid, int nt, int ps = AC_PAD_SHIFT, int ps_dst = -1) {
  if (ps_dst < 0) ps_dst = ps;   // (ps: the padding of src; ps_dst: of dst, when the two buffers are padded differently)
  constexpr int R = R1 * R2;
  const int m = H / (R * L), nb = H / R;
  const unsigned invL = 0xFFFFFFFFu / (unsigned)L + 1u;   // j / L for j < 2^16 as a multiply-high
  auto butterfly = [&](int j) {
    const int p = L == 1 ? j : (int)__umulhi((unsigned)j, invL), q = j - p * L;
    const int base = q + L * p, tq = q * m, ob = q + L * R * p, Lm = L * m;
    auto load = [&](int s2) {
      const int n = base + Lm * s2;
      if constexpr (first) {   // (L = 1, q = 0: no pass twiddle)
        cpair t;
        t.re = v[2 * n];
        t.im = v[N - 1 - 2 * n];
        return cmulw(t, tb.pre[n]);
      } else {
        const cpair x = src[pad16(n, ps)];
        return s2 > 0 ? cmulw(x, tb.tw[tq * s2]) : x;
      }
    };
    auto store = [&](int t, const cpair& val) {
      const int k = ob + L * t;
      if constexpr (last_to_v) {
        const cpair r = cmulw(val, tb.post[k]);
        v[2 * k] = r.re;
        v[N - 1 - 2 * k] = make_float2(-r.im.x, -r.im.y);
      } else {
        dst[pad16(k, ps_dst)] = val;
      }
    };
    if constexpr (R2 == 1) {
      cpair g[R1];
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) g[n1] = load(n1);
      dft_small<R1>(g);
#pragma unroll
      for (int t = 0; t < R1; ++t) store(t, g[t]);
    } else {
      cpair y[R];
      Stage1<R1, R2, R2 - 1>::run(load, y);
#pragma unroll
      for (int k1 = 0; k1 < R1; ++k1) {
        cpair h[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) h[n2] = y[k1 * R2 + n2];
        dft_small<R2>(h);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) store(k1 + R1 * k2, h[k2]);
      }
    }
  };
  if constexpr (CT) {   // N and nt are compile-time constants of the caller: the rounds unroll, the strides fold
    const int rounds = (nb + nt - 1) / nt;
#pragma unroll
    for (int rd = 0; rd < rounds; ++rd) {
      const int j = tid + rd * nt;
      if (j < nb) butterfly(j);
    }
  } else {
    for (int j = tid; j < nb; j += nt) butterfly(j);
  }
}

// largest super-radix compiled in: 16 / 15 / 12 take two passes off some sizes but push the kernels past 256 registers
#ifndef AC_WAVE_MAX_RADIX
#define AC_WAVE_MAX_RADIX 10
#endif
// the super-radices of a size, chosen on the host (lds_wave_plan): their product is N / 2
struct WavePlan {
  int n;
  unsigned char r[6];
  int nt;   // lanes per frame (a power of two <= 64; 64 / nt frames share a wave)
};

// v[N] (LDS, float2 per entry; its bytes are ALSO buffer Bp) -> DCT-IV written back into v, as dct4_lds.  Ap, Bp: padded_len(N/2)
// cpairs each.  Called by all the lanes of a wave; tid = lane inside its group of nt.
// Buffers: pass 1 reads v (pre-twiddle fused) and writes Ap; the passes then alternate Ap -> Bp -> Ap ...  An even number of
// passes ends with a pass that reads Ap and writes v in final form (post-twiddle fused: v's bytes are Bp's, free by then); an
// odd number ends in Ap, and a separate post-twiddle step writes v.
__device__ void dct4_wave(float2* v, cpair* Ap, cpair* Bp, const WaveTabs& tb, int N, int tid, int nt, const WavePlan& wp) {
  const int H = N >> 1;
  const bool even = (wp.n & 1) == 0;
  cpair* src = Bp;   // (unused by the first pass)
  cpair* dst = Ap;
  int L = 1;
  for (int ps = 0; ps < wp.n; ++ps) {
    const int r = wp.r[ps];
    const bool first = ps == 0, lastv = even && ps == wp.n - 1;
#define AC_WAVE_PASS(A, B)                                                                   \
  if (first) wave_pass<A, B, true, false>(src, dst, v, N, L, H, tb, tid, nt);                \
  else if (lastv) wave_pass<A, B, false, true>(src, dst, v, N, L, H, tb, tid, nt);           \
  else wave_pass<A, B, false, false>(src, dst, v, N, L, H, tb, tid, nt);                     \
  break
    switch (r) {
#if AC_WAVE_MAX_RADIX >= 16
      case 16: AC_WAVE_PASS(4, 4);
      case 15: AC_WAVE_PASS(3, 5);
      case 12: AC_WAVE_PASS(4, 3);
#endif
      case 10: AC_WAVE_PASS(2, 5);
      case 9: AC_WAVE_PASS(3, 3);
      case 8: AC_WAVE_PASS(4, 2);
      case 6: AC_WAVE_PASS(2, 3);
      case 5: AC_WAVE_PASS(5, 1);
      case 4: AC_WAVE_PASS(4, 1);
      case 3: AC_WAVE_PASS(3, 1);
      default: AC_WAVE_PASS(2, 1);
    }
#undef AC_WAVE_PASS
    wave_sync_lds();
    cpair* t = (ps == 0) ? Bp : src;   // after pass 1 the data is in Ap and Bp (= v, read out) is free
    src = dst;
    dst = t;
    L *= r;
  }
  if (!even) {
    for (int k = tid; k < H; k += nt) {
      const cpair r = cmulw(src[pad16(k)], tb.post[k]);   // src == Ap here
      v[2 * k] = r.re;
      v[N - 1 - 2 * k] = make_float2(-r.im.x, -r.im.y);
    }
    wave_sync_lds();
  }
}

// the same with the size, the lanes per frame and the super-radices known at compile time (R3 / R2 = 0: three / two passes):
// strides, rounds and buffer offsets fold into immediates
template <int R> struct RadixSplit { static constexpr int A = R, B = 1; };
template <> struct RadixSplit<16> { static constexpr int A = 4, B = 4; };
template <> struct RadixSplit<15> { static constexpr int A = 3, B = 5; };
template <> struct RadixSplit<12> { static constexpr int A = 4, B = 3; };
template <> struct RadixSplit<10> { static constexpr int A = 2, B = 5; };
template <> struct RadixSplit<9> { static constexpr int A = 3, B = 3; };
template <> struct RadixSplit<8> { static constexpr int A = 4, B = 2; };
template <> struct RadixSplit<6> { static constexpr int A = 2, B = 3; };
template <int NC, int NTC, int R0, int R1, int R2, int R3>
__device__ __forceinline__ void dct4_wave_ct(float2* v, cpair* Ap, cpair* Bp, const WaveTabs& tb, int tid) {
  constexpr int H = NC / 2, NP = 1 + (R1 > 0) + (R2 > 0) + (R3 > 0);
  static_assert(R0 * (R1 ? R1 : 1) * (R2 ? R2 : 1) * (R3 ? R3 : 1) == H, "the super-radices multiply to N / 2");
  constexpr bool even = (NP & 1) == 0;
  constexpr int PS = pad_shift_ct(NC);
  // The first pass writes its outputs R0 elements apart from lane to lane.  One element of padding per 16 spreads a stride
  // that is a multiple of four over the banks; a stride of 5, 6, 9 or 10 elements already visits all sixteen 16-byte bank
  // groups in eight consecutive lanes, and the padding only folds them onto each other (filters_n = 960, R0 = 10: lanes
  // 0 .. 7 land on groups 0, 10, 5, 15, 10, 5, 15, 10).  So buffer A -- the first pass's target -- is padded only where the
  // first radix asks for it; buffer B (and the later passes' writes: runs of R0 consecutive elements) keeps the padding.
  // Measured (128 stereo clips of 10 s, base -> this): filters_n 600 transform 0.249 -> 0.212 ms, 540 0.244 -> 0.216, 648
  // 0.218 -> 0.200, 360 inverse 0.220 -> 0.202, the other sizes of 32 and 64 lanes per frame within the noise; the frames
  // of 8 and 16 lanes (108, 160) lost 8-15 % on the inverse and keep the padding.  LDS bank-conflict cycles at 960: 35 % of
  // the LDS-active cycles -> 26 %.
  constexpr int PSA = (R0 % 4 == 0 || PS != AC_PAD_SHIFT || NTC < 32) ? PS : 30, PSB = PS;
  // pass 1: v -> Ap; then Ap -> Bp -> Ap ...; an even count ends in v (= Bp's bytes) in final form
  wave_pass<RadixSplit<R0>::A, RadixSplit<R0>::B, true, false, true>(Bp, Ap, v, NC, 1, H, tb, tid, NTC, PSB, PSA);
  wave_sync_lds();
  if constexpr (NP >= 2) {
    wave_pass<RadixSplit<R1>::A, RadixSplit<R1>::B, false, NP == 2, true>(Ap, Bp, v, NC, R0, H, tb, tid, NTC, PSA, PSB);
    wave_sync_lds();
  }
  if constexpr (NP >= 3) {
    wave_pass<RadixSplit<R2>::A, RadixSplit<R2>::B, false, false, true>(Bp, Ap, v, NC, R0 * R1, H, tb, tid, NTC, PSB, PSA);
    wave_sync_lds();
  }
  if constexpr (NP >= 4) {
    wave_pass<RadixSplit<R3>::A, RadixSplit<R3>::B, false, true, true>(Ap, Bp, v, NC, R0 * R1 * R2, H, tb, tid, NTC, PSA, PSB);
    wave_sync_lds();
  }
  if constexpr (!even) {
#pragma unroll
    for (int rd = 0; rd < (H + NTC - 1) / NTC; ++rd) {
      const int k = tid + rd * NTC;
      if (k < H) {
        const cpair r = cmulw(Ap[pad16(k, PSA)], tb.post[k]);
        v[2 * k] = r.re;
        v[NC - 1 - 2 * k] = make_float2(-r.im.x, -r.im.y);
      }
    }
    wave_sync_lds();
  }
}

// ---- the same passes for a frame dealt to NTC = 128 / 256 lanes (two / four waves: filters_n above 1024), IN PLACE in one
// padded buffer that shares the bytes of v: a pass loads and transforms all its butterflies in registers (one or two per
// lane), the group synchronises, then the outputs go back.  8.5 N bytes of LDS per frame; the pre-twiddles are formed from
// the post-twiddle table (exp(-i pi (n + 1/4) / N) = post[n] exp(-i pi / (4 N))): two tables beside the frame, so that
// filters_n = 4096 keeps two workgroups (eight waves) per CU.
template <int NTC>
__device__ __forceinline__ void group_sync() {
  if constexpr (NTC > 64) __syncthreads();
  else wave_sync_lds();
}
template <int NC, int NTC, int L, int R, bool first, bool last_to_v>
__device__ __forceinline__ void group_pass(cpair* buf, float2* v, const WaveTabs& tb, float2 pre0, int tid) {
  constexpr int R1 = RadixSplit<R>::A, R2 = RadixSplit<R>::B;
  constexpr int H = NC / 2, m = H / (R * L), nb = H / R, Lm = L * m, rounds = (nb + NTC - 1) / NTC, PS = pad_shift_ct(NC);
  cpair out[rounds][R];
#pragma unroll
  for (int rd = 0; rd < rounds; ++rd) {
    const int j = tid + rd * NTC;
    if (j < nb) {
      const int p = j / L, q = j - p * L;
      const int base = q + L * p, tq = q * m;
      auto load = [&](int s2) {
        const int n = base + Lm * s2;
        if constexpr (first) {
          cpair t;
          t.re = v[2 * n];
          t.im = v[NC - 1 - 2 * n];
          const float2 w = tb.post[n];
          return cmulw(t, make_float2(w.x * pre0.x - w.y * pre0.y, w.x * pre0.y + w.y * pre0.x));
        } else {
          const cpair x = buf[pad16(n, PS)];
          return s2 > 0 ? cmulw(x, tb.tw[tq * s2]) : x;
        }
      };
      if constexpr (R2 == 1) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) out[rd][n1] = load(n1);
        dft_small<R1>(out[rd]);
      } else {
        cpair y[R];
        Stage1<R1, R2, R2 - 1>::run(load, y);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
          cpair h[R2];
#pragma unroll
          for (int n2 = 0; n2 < R2; ++n2) h[n2] = y[k1 * R2 + n2];
          dft_small<R2>(h);
#pragma unroll
          for (int k2 = 0; k2 < R2; ++k2) out[rd][k1 + R1 * k2] = h[k2];
        }
      }
    }
  }
  group_sync<NTC>();
#pragma unroll
  for (int rd = 0; rd < rounds; ++rd) {
    const int j = tid + rd * NTC;
    if (j < nb) {
      const int p = j / L, q = j - p * L, ob = q + L * R * p;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const int k = ob + L * t;
        if constexpr (last_to_v) {
          const cpair r = cmulw(out[rd][t], tb.post[k]);
          v[2 * k] = r.re;
          v[NC - 1 - 2 * k] = make_float2(-r.im.x, -r.im.y);
        } else {
          buf[pad16(k, PS)] = out[rd][t];
        }
      }
    }
  }
  group_sync<NTC>();
}
template <int NC, int NTC, int R0, int R1, int R2, int R3>
__device__ __forceinline__ void dct4_group_ct(float2* v, cpair* buf, const WaveTabs& tb, float2 pre0, int tid) {
  constexpr int H = NC / 2, NP = 1 + (R1 > 0) + (R2 > 0) + (R3 > 0);
  static_assert(R0 * (R1 ? R1 : 1) * (R2 ? R2 : 1) * (R3 ? R3 : 1) == H, "the super-radices multiply to N / 2");
  static_assert(NP >= 2, "at least two passes (the first reads v, the last writes it)");
  group_pass<NC, NTC, 1, R0, true, false>(buf, v, tb, pre0, tid);
  group_pass<NC, NTC, R0, R1, false, NP == 2>(buf, v, tb, pre0, tid);
  if constexpr (NP >= 3) group_pass<NC, NTC, R0 * R1, R2, false, NP == 3>(buf, v, tb, pre0, tid);
  if constexpr (NP >= 4) group_pass<NC, NTC, R0 * R1 * R2, R3, false, true>(buf, v, tb, pre0, tid);
}

// LDS floats per frame of the wave form: Bp (= v) and Ap, padded; of the in-place form above: one buffer
// frames of at least AC_WAVE_INPLACE_LANES lanes (and two passes or more) are transformed in place in one buffer -- half the LDS
// per frame, twice the frames resident; a frame on several waves always is.  For frames inside a wave it is OFF: built with
// -DAC_WAVE_INPLACE_LANES=4 (tools/build_variant.sh) the tier ran within +-3 % of the two-buffer form at every size measured
// (B = 256 stereo, 20 sizes 24 ... 1024: 960 0.467 / 0.479 -> 0.461 / 0.456 ms, 600 0.420 -> 0.462, 720 0.447 -> 0.428): with 9
// or 18 frames resident the kernels run at the rate of a device copy of the same tensors (0.38 ms), so residency is not the limit.
#ifndef AC_WAVE_INPLACE_LANES
#define AC_WAVE_INPLACE_LANES 65
#endif
static inline __host__ __device__ constexpr bool wave_in_place(int ntc, int r1) { return ntc > 64 || (ntc >= AC_WAVE_INPLACE_LANES && r1 > 0); }
static inline __host__ __device__ constexpr int wave_floats_per_group(int N, int ps = AC_PAD_SHIFT) { return 2 * 4 * padded_len(N / 2, ps); }
static inline __host__ __device__ constexpr int group_floats_per_frame(int N, int ps = AC_PAD_SHIFT) { return 4 * padded_len(N / 2, ps); }

// the FFT's twiddles exp(-2 pi i k / (N/2)), k < N/2, once per workgroup into LDS (every thread takes part; the caller
// synchronises before the first use)
__device__ __forceinline__ void fill_twiddles(float2* tw, const float* __restrict__ ctab, int N) {
  for (int k = threadIdx.x; k < N / 2; k += kThreads) tw[k] = cis_neg(ctab, 16 * k, N);   // exp(-i pi (16 k) / (4 N))
}

// threads per group: a power of two (a workgroup holds kThreads / nt groups) near the N/8 butterflies of a radix-4
// stage, at least one wave
static inline __host__ __device__ int lds_group_threads(int N) {
  int nt = 64;
  while (nt < kThreads && nt < N / 8) nt <<= 1;
  return nt;
}

// LDS floats per group of the analysis kernel: v [N float2] + A + B; beyond filters_n 2048 B shares v's bytes (measured:
// N = 4096 0.679 -> 0.495 ms, two workgroups per CU instead of one; at smaller sizes the third buffer is faster)
#ifndef AC_LDS_ALIAS_ABOVE
#define AC_LDS_ALIAS_ABOVE 2048
#endif
static inline __host__ __device__ int lds_fwd_floats_per_group(int N) { return N > AC_LDS_ALIAS_ABOVE ? 4 * N : 6 * N; }

// one group per (clip, channel pair, frame); ALIAS: the form for filters_n > 2048 (a kernel of its own: see dct4_lds)
template <typename TIO, bool ALIAS = false>
static __global__ __launch_bounds__(kThreads) void k_fwd_lds(const TIO* __restrict__ x, TIO* __restrict__ X,
                                                      const TIO* __restrict__ prev_block,
                                                      const float* __restrict__ coef,
                                                      const float* __restrict__ ctab, int Kin, int F, int C, int CP,
                                                      int N, long long ntasks) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int nt = lds_group_threads(N), grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  const int per = ALIAS ? 4 * N : 6 * N;                   // (= lds_fwd_floats_per_group(N)) second FFT buffer aliased to v or not
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)(kThreads / nt) * per);   // [N/2] behind the groups' buffers
  fill_twiddles(tw, ctab, N);
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);             // [N]
  cpair* A = reinterpret_cast<cpair*>(base + 2 * N);       // [N/2]
  cpair* Bf = ALIAS ? reinterpret_cast<cpair*>(base) : reinterpret_cast<cpair*>(base + 4 * N);   // [N/2]
  const int h = N >> 1;
  const long long task_raw = (long long)blockIdx.x * (kThreads / nt) + grp;
  const bool valid = task_raw < ntasks;
  const long long wg = valid ? task_raw : ntasks - 1;      // idle groups of the last workgroup only join the barriers
  const int n = (int)(wg % F);
  const long long sig = wg / F;
  const int c = 2 * (int)(sig % CP);
  const bool has1 = c + 1 < C;
  const long long b = sig / CP;
  const float* a1 = coef;
  const float* a2 = coef + h;
  const float* a3 = coef + 2 * h;
  const float* a4 = coef + 3 * h;
  const bool has_cur = n < Kin;
  const TIO* xc = x + ((size_t)b * Kin + (size_t)n) * N * C + c;
  const TIO* xp = nullptr;
  if (n >= 1) xp = x + ((size_t)b * Kin + (size_t)(n - 1)) * N * C + c;
  else if (prev_block) xp = prev_block + (size_t)b * N * C + c;
  for (int j = tid; j < h; j += nt) {
    float2 vc = make_float2(0.f, 0.f), vp = make_float2(0.f, 0.f);
    if (has_cur) {
      const float2 p = ld2(xc + (size_t)j * C, C, has1), q = ld2(xc + (size_t)(N - 1 - j) * C, C, has1);
      vc = make_float2(a1[j] * p.x + a2[j] * q.x, a1[j] * p.y + a2[j] * q.y);
    }
    if (xp) {
      const float2 p = ld2(xp + (size_t)(h - 1 - j) * C, C, has1), q = ld2(xp + (size_t)(h + j) * C, C, has1);
      vp = make_float2(a3[j] * p.x + a4[j] * q.x, a3[j] * p.y + a4[j] * q.y);
    }
    v[h + j] = vc;
    v[j] = vp;
  }
  __syncthreads();
  dct4_lds<ALIAS>(v, A, Bf, ctab, tw, N, tid, nt);
  if (!valid) return;
  const float scale = (float)(1.0 / ((double)N * 1.4142135623730951));   // 1/sqrt(4N) * sqrt(2/N)
  TIO* Xo = X + (((size_t)b * F + (size_t)n) * N) * C + c;
  for (int k = tid; k < N; k += nt) st2(Xo + (size_t)k * C, make_float2(v[k].x * scale, v[k].y * scale), C, has1);
}

// one group per (clip, channel pair, strip of `seg` output blocks): the aliased half of the previous frame's DCT-IV
// stays in LDS along the strip, so a strip of T blocks costs T + 1 transforms; the block index nblk (one past the
// last) only writes the new stream state.  Every group runs the same number of transforms (barriers are workgroup-wide).
template <typename TIO>
static __global__ __launch_bounds__(kThreads) void k_inv_lds(const TIO* __restrict__ X, TIO* __restrict__ x,
                                                      const float* __restrict__ tail_in, float* __restrict__ tail_out,
                                                      const float* __restrict__ coef, const float* __restrict__ ctab,
                                                      int Kp, int nblk, int seg, int nseg, int C, int CP, int N,
                                                      long long ntasks) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int nt = lds_group_threads(N), grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)(kThreads / nt) * 7 * N);   // [N/2] behind the groups' buffers
  fill_twiddles(tw, ctab, N);
  float* base = smem + (size_t)grp * 7 * N;
  float2* v = reinterpret_cast<float2*>(base);             // [N]
  cpair* A = reinterpret_cast<cpair*>(base + 2 * N);       // [N/2]
  cpair* Bf = reinterpret_cast<cpair*>(base + 4 * N);      // [N/2]
  float2* um = reinterpret_cast<float2*>(base + 6 * N);    // [N/2]  u_{n-1}[h + j]
  const int h = N >> 1;
  const long long task_raw = (long long)blockIdx.x * (kThreads / nt) + grp;
  const bool valid = task_raw < ntasks;
  const long long wg = valid ? task_raw : ntasks - 1;
  const int sgm = (int)(wg % nseg);
  const long long sig = wg / nseg;
  const int c = 2 * (int)(sig % CP);
  const bool has1 = c + 1 < C;
  const long long b = sig / CP;
  const float* s1 = coef + 4 * h;
  const float* s2 = coef + 5 * h;
  const float* s3 = coef + 6 * h;
  const float* s4 = coef + 7 * h;
  const float scale = 2.0f * 1.4142135623730951f;   // sqrt(4N) * sqrt(2/N)
  const int nlast = nblk + (tail_out ? 1 : 0);      // blocks incl. the virtual state block
  const int n0 = sgm * seg;
  const size_t ts = ((size_t)b * C + c) * h;        // stream state rows of the pair: ts, ts + h
  // aliased half before the strip: frame n0 - 1, the stream state, or zero
  {
    const bool halo = n0 >= 1;
    const TIO* Xi = X + (((size_t)b * Kp + (size_t)(halo ? n0 - 1 : 0)) * N) * C + c;
    for (int k = tid; k < N; k += nt) v[k] = halo ? ld2(Xi + (size_t)k * C, C, has1) : make_float2(0.f, 0.f);
    __syncthreads();
    dct4_lds(v, A, Bf, ctab, tw, N, tid, nt);
    for (int j = tid; j < h; j += nt) {
      if (halo) um[j] = make_float2(v[h + j].x * scale, v[h + j].y * scale);
      else um[j] = tail_in ? make_float2(tail_in[ts + j], has1 ? tail_in[ts + h + j] : 0.f) : make_float2(0.f, 0.f);
    }
    __syncthreads();
  }
  for (int t = 0; t < seg; ++t) {
    const int n = n0 + t;
    const bool live = valid && n < nlast;    // this group still has a block to write
    const bool has_n = n < Kp && n < nblk;   // the virtual state block (n == nblk) has no current frame
    {
      const TIO* Xi = X + (((size_t)b * Kp + (size_t)(has_n ? n : 0)) * N) * C + c;
      for (int k = tid; k < N; k += nt) v[k] = has_n ? ld2(Xi + (size_t)k * C, C, has1) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    dct4_lds(v, A, Bf, ctab, tw, N, tid, nt);
    if (live) {
      for (int j = tid; j < h; j += nt) {
        const float2 a = make_float2(v[h - 1 - j].x * scale, v[h - 1 - j].y * scale);   // u_n[h-1-j]
        const float2 bb = um[j];                                                        // u_{n-1}[h+j]
        if (n < nblk) {
          TIO* xo = x + (((size_t)b * nblk + (size_t)n) * N) * C + c;
          st2(xo + (size_t)j * C, make_float2(s1[j] * a.x + s2[j] * bb.x, s1[j] * a.y + s2[j] * bb.y), C, has1);
          st2(xo + (size_t)(N - 1 - j) * C, make_float2(s3[j] * a.x + s4[j] * bb.x, s3[j] * a.y + s4[j] * bb.y), C, has1);
        } else if (tail_out) {
          tail_out[ts + j] = bb.x;
          if (has1) tail_out[ts + h + j] = bb.y;
        }
      }
    }
    __syncthreads();
    for (int j = tid; j < h; j += nt) um[j] = make_float2(v[h + j].x * scale, v[h + j].y * scale);
    __syncthreads();
  }
}

// ---- the wave form of the two kernels above (filters_n <= 2048): a group of wp.nt lanes inside one wave per frame / strip,
// several tasks per group so that the twiddle table is built once per workgroup; no workgroup barrier after that
template <typename TIO>
static __global__ __launch_bounds__(kThreads, 2) void k_fwd_wave(const TIO* __restrict__ x, TIO* __restrict__ X,
                                                       const TIO* __restrict__ prev_block, const float* __restrict__ coef,
                                                       const float* __restrict__ ctab, int Kin, int F, int C, int CP, int N,
                                                       long long ntasks, int T, WavePlan wp) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int nt = wp.nt, gpw = (int)blockDim.x / nt, grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  const int per = wave_floats_per_group(N), h = N >> 1;
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)gpw * per);   // three tables of N/2 behind the groups' buffers
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    tw[k] = cis_neg(ctab, 16 * k, N);            // exp(-2 pi i k / (N/2))
    tw[h + k] = cis_neg(ctab, 4 * k + 1, N);     // exp(-i pi (k + 1/4) / N)
    tw[2 * h + k] = cis_neg(ctab, 4 * k, N);     // exp(-i pi k / N)
  }
  __syncthreads();
  const WaveTabs tb = {tw, tw + h, tw + 2 * h};
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);                                   // [N], sharing the bytes of Bp
  cpair* Bp = reinterpret_cast<cpair*>(base);
  cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h));
  const float *a1 = coef, *a2 = coef + h, *a3 = coef + 2 * h, *a4 = coef + 3 * h;
  const float scale = (float)(1.0 / ((double)N * 1.4142135623730951));   // 1/sqrt(4N) * sqrt(2/N)
  for (int t = 0; t < T; ++t) {
    const long long wg = ((long long)blockIdx.x * T + t) * gpw + grp;
    if (wg >= ntasks) return;
    const int n = (int)(wg % F);
    const long long sig = wg / F;
    const int c = 2 * (int)(sig % CP);
    const bool has1 = c + 1 < C;
    const long long b = sig / CP;
    const bool has_cur = n < Kin;
    const TIO* xc = x + ((size_t)b * Kin + (size_t)n) * N * C + c;
    const TIO* xp = nullptr;
    if (n >= 1) xp = x + ((size_t)b * Kin + (size_t)(n - 1)) * N * C + c;
    else if (prev_block) xp = prev_block + (size_t)b * N * C + c;
    // the frame's PCM in batches of eight steps: all the loads of a batch are in flight together (a wave that waits for
    // each step's loads in turn keeps a few hundred bytes in flight, and the tier ran at a quarter of the memory rate)
    for (int j0 = tid; j0 < h; j0 += 8 * nt) {
      float2 pc[8], qc[8], pp[8], qp[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = j0 + i * nt;
        pc[i] = qc[i] = pp[i] = qp[i] = make_float2(0.f, 0.f);
        if (j < h) {
          if (has_cur) {
            pc[i] = ld2(xc + (size_t)j * C, C, has1);
            qc[i] = ld2(xc + (size_t)(N - 1 - j) * C, C, has1);
          }
          if (xp) {
            pp[i] = ld2(xp + (size_t)(h - 1 - j) * C, C, has1);
            qp[i] = ld2(xp + (size_t)(h + j) * C, C, has1);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = j0 + i * nt;
        if (j < h) {
          v[h + j] = make_float2(a1[j] * pc[i].x + a2[j] * qc[i].x, a1[j] * pc[i].y + a2[j] * qc[i].y);
          v[j] = make_float2(a3[j] * pp[i].x + a4[j] * qp[i].x, a3[j] * pp[i].y + a4[j] * qp[i].y);
        }
      }
    }
    wave_sync_lds();
    dct4_wave(v, Ap, Bp, tb, N, tid, nt, wp);
    TIO* Xo = X + (((size_t)b * F + (size_t)n) * N) * C + c;
    for (int k = tid; k < N; k += nt) st2(Xo + (size_t)k * C, make_float2(v[k].x * scale, v[k].y * scale), C, has1);
    wave_sync_lds();   // v is read out before the next task folds into it
  }
}

template <typename TIO>
static __global__ __launch_bounds__(kThreads, 2) void k_inv_wave(const TIO* __restrict__ X, TIO* __restrict__ x,
                                                       const float* __restrict__ tail_in, float* __restrict__ tail_out,
                                                       const float* __restrict__ coef, const float* __restrict__ ctab,
                                                       int Kp, int nblk, int seg, int nseg, int C, int CP, int N,
                                                       long long ntasks, WavePlan wp) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int nt = wp.nt, gpw = (int)blockDim.x / nt, grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  const int h = N >> 1, per = wave_floats_per_group(N) + N;   // + um [N/2] float2
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)gpw * per);
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    tw[k] = cis_neg(ctab, 16 * k, N);
    tw[h + k] = cis_neg(ctab, 4 * k + 1, N);
    tw[2 * h + k] = cis_neg(ctab, 4 * k, N);
  }
  __syncthreads();
  const WaveTabs tb = {tw, tw + h, tw + 2 * h};
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);
  cpair* Bp = reinterpret_cast<cpair*>(base);
  cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h));
  float2* um = reinterpret_cast<float2*>(base + wave_floats_per_group(N));   // u_{n-1}[h + j]
  const long long wg = (long long)blockIdx.x * gpw + grp;
  if (wg >= ntasks) return;
  const int sgm = (int)(wg % nseg);
  const long long sig = wg / nseg;
  const int c = 2 * (int)(sig % CP);
  const bool has1 = c + 1 < C;
  const long long b = sig / CP;
  const float *s1 = coef + 4 * h, *s2 = coef + 5 * h, *s3 = coef + 6 * h, *s4 = coef + 7 * h;
  const float scale = 2.0f * 1.4142135623730951f;   // sqrt(4N) * sqrt(2/N)
  const int nlast = nblk + (tail_out ? 1 : 0);      // blocks incl. the virtual state block
  const int n0 = sgm * seg;
  const size_t ts = ((size_t)b * C + c) * h;        // stream state rows of the pair: ts, ts + h
  // aliased half before the strip: the stream state or zero for a signal's first strip, else frame n0 - 1 (step t = -1 of
  // the loop below: one call site of the transform)
  if (n0 == 0) {
    for (int j = tid; j < h; j += nt)
      um[j] = tail_in ? make_float2(tail_in[ts + j], has1 ? tail_in[ts + h + j] : 0.f) : make_float2(0.f, 0.f);
    wave_sync_lds();
  }
  for (int t = (n0 >= 1 ? -1 : 0); t < seg; ++t) {
    const int n = n0 + t;
    if (t >= 0 && n >= nlast) break;         // nothing left to write
    const bool has_n = t < 0 || (n < Kp && n < nblk);   // the virtual state block (n == nblk) has no current frame
    {
      const TIO* Xi = X + (((size_t)b * Kp + (size_t)(has_n ? n : 0)) * N) * C + c;
      for (int k0 = tid; k0 < N; k0 += 16 * nt) {   // sixteen loads in flight per lane (see k_fwd_wave)
        float2 r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = k0 + i * nt;
          r[i] = (has_n && k < N) ? ld2(Xi + (size_t)k * C, C, has1) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = k0 + i * nt;
          if (k < N) v[k] = r[i];
        }
      }
    }
    wave_sync_lds();
    if (has_n) dct4_wave(v, Ap, Bp, tb, N, tid, nt, wp);   // (the DCT-IV of a zero frame is zero)
    if (t >= 0) {
      for (int j = tid; j < h; j += nt) {
        const float2 a = make_float2(v[h - 1 - j].x * scale, v[h - 1 - j].y * scale);   // u_n[h-1-j]
        const float2 bb = um[j];                                                        // u_{n-1}[h+j]
        if (n < nblk) {
          TIO* xo = x + (((size_t)b * nblk + (size_t)n) * N) * C + c;
          st2(xo + (size_t)j * C, make_float2(s1[j] * a.x + s2[j] * bb.x, s1[j] * a.y + s2[j] * bb.y), C, has1);
          st2(xo + (size_t)(N - 1 - j) * C, make_float2(s3[j] * a.x + s4[j] * bb.x, s3[j] * a.y + s4[j] * bb.y), C, has1);
        } else if (tail_out) {
          tail_out[ts + j] = bb.x;
          if (has1) tail_out[ts + h + j] = bb.y;
        }
      }
      wave_sync_lds();
    }
    for (int j = tid; j < h; j += nt) um[j] = make_float2(v[h + j].x * scale, v[h + j].y * scale);
    wave_sync_lds();
  }
}

// ---- the 16-byte kernels of the tier (float32, N % 4 == 0, N <= 16 nt): wide accesses and every PCM block read ONCE.
// A lane owns the sample pairs j = 2 i, 2 i + 1 (i = tid + s nt, s < 4) and their mirrors N - 1 - j.  Block n enters frame n
// through (a1, a2) (second half of the fold) and frame n + 1 through (a3, a4) (first half): with j' = h - 1 - j the second
// is  v[h - 1 - j] = a3[h - 1 - j] x[j] + a4[h - 1 - j] x[N - 1 - j],  i.e. the SAME two samples the lane already holds, so
// it is formed at once and carried in registers to the next frame of the strip.  coefv (ac_mdct_plan::d_coefv) holds the
// coefficients in that order, 16 bytes per lane and step.  The next block's loads are issued before the transform of the
// current frame and land while it runs.
// The two rows a complex pair carries through the transform (LAY): 0 the channels of a stereo signal (one 16-byte access
// per two samples); 1 two mono signals b, b + 1 (8 bytes each; the last pair of an odd batch is half empty); 2 the channels
// c, c + 1 of three or more channels (8-byte accesses on the 4-byte grid; the last pair of an odd count is half empty).  bfloat16 tensors and filters_n % 4 == 2 run the 8-byte kernels above.
typedef float v4f_t __attribute__((ext_vector_type(4)));
// instances of the 16-byte kernels that form the lane's LDS offsets per frame from opaque copies instead of holding them, hoisted,
// across the frame loop (-DAC_WAVE_REBASE: all of them): ~130 instead of ~200 registers, slower at most sizes (+2 ... +13 %, 18
// sizes on the same tensors), faster at these: transform 1152 0.490 -> 0.434 ms, 2160 0.508 -> 0.457, 2304 0.537 -> 0.502;
// inverse 2304 0.501 -> 0.479, 7680 0.603 -> 0.545 (B = 256 stereo)
static inline __host__ __device__ constexpr bool wave_rebase(int N, bool inverse) {
  return inverse ? (N == 2304 || N == 7680) : (N == 1152 || N == 2160 || N == 2304);
}
__device__ __forceinline__ float ola2(float a, float x, float b, float y) { return __builtin_fmaf(a, x, b * y); }   // a x + b y, one rounding order
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v2u_t __attribute__((ext_vector_type(2), aligned(4)));   // two floats on the 4-byte grid
constexpr int kWaveVSteps = 4;
template <int LAY>
struct RowPair {
  bool has1;
  int C;   // (LAY 2) floats between successive samples
  // samples m, m + 1 (m even) of the two rows starting at a (and b): (row0[m], row1[m], row0[m+1], row1[m+1]).  No branch on
  // has1: a conditional load would make the wave wait for it at the join, before the transform it is meant to overlap; a
  // half-empty pair reads its one row twice (pair_geo) and never stores the second.
  __device__ __forceinline__ v4f_t load2(const float* a, const float* b, int m) const {
    if constexpr (LAY == 0) {
      return *reinterpret_cast<const v4f_t*>(a + 2 * m);
    } else if constexpr (LAY == 1) {
      const v2f_t fa = *reinterpret_cast<const v2f_t*>(a + m);
      const v2f_t fb = *reinterpret_cast<const v2f_t*>(b + m);
      return v4f_t{fa.x, fb.x, fa.y, fb.y};
    } else {
      // the pair's two channels are adjacent: one 8-byte access on the 4-byte grid per sample (gfx950 takes multi-dword global
      // accesses at dword alignment); the half-empty last pair of an odd channel count reads (c - 1, c) instead of (c, c + 1)
      int o = m * C - (has1 ? 0 : 1);   // (formed where it is used: hoisted out of the frame loop, the addresses of a lane's accesses spill)
      asm volatile("" : "+v"(o));
      const v2u_t s0 = *reinterpret_cast<const v2u_t*>(a + o), s1 = *reinterpret_cast<const v2u_t*>(a + o + C);
      return v4f_t{has1 ? s0.x : s0.y, s0.y, has1 ? s1.x : s1.y, s1.y};
    }
  }
  // the same rows as 16-bit PCM (x = pcm / 32768 on the way in, clamp(round(32768 x)) on the way out, as the wave-level
  // kernels of ac_fast.hip do): 8 / 4 bytes per access
  __device__ __forceinline__ v4f_t load2(const int16_t* a, const int16_t* b, int m) const {
    static_assert(LAY <= 1, "16-bit PCM: stereo or mono rows");
    typedef short s4_t __attribute__((ext_vector_type(4)));
    typedef short s2_t __attribute__((ext_vector_type(2)));
    constexpr float k = 1.0f / 32768.0f;
    if constexpr (LAY == 0) {
      const s4_t q = *reinterpret_cast<const s4_t*>(a + 2 * m);
      return v4f_t{(float)q.x * k, (float)q.y * k, (float)q.z * k, (float)q.w * k};
    } else {
      const s2_t qa = *reinterpret_cast<const s2_t*>(a + m), qb = *reinterpret_cast<const s2_t*>(b + m);
      return v4f_t{(float)qa.x * k, (float)qb.x * k, (float)qa.y * k, (float)qb.y * k};
    }
  }
  __device__ __forceinline__ void store2(int16_t* a, int16_t* b, int m, v4f_t v) const {
    static_assert(LAY <= 1, "16-bit PCM: stereo or mono rows");
    typedef short s4_t __attribute__((ext_vector_type(4)));
    typedef short s2_t __attribute__((ext_vector_type(2)));
    auto enc = [](float f) { return (short)__float2int_rn(fminf(fmaxf(f * 32768.0f, -32768.0f), 32767.0f)); };
    if constexpr (LAY == 0) {
      __builtin_nontemporal_store(s4_t{enc(v.x), enc(v.y), enc(v.z), enc(v.w)}, reinterpret_cast<s4_t*>(a + 2 * m));
    } else {
      __builtin_nontemporal_store(s2_t{enc(v.x), enc(v.z)}, reinterpret_cast<s2_t*>(a + m));
      if (has1) __builtin_nontemporal_store(s2_t{enc(v.y), enc(v.w)}, reinterpret_cast<s2_t*>(b + m));
    }
  }
  __device__ __forceinline__ void store2(float* a, float* b, int m, v4f_t v) const {
    if constexpr (LAY == 0) {
      __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(a + 2 * m));
    } else if constexpr (LAY == 1) {
      __builtin_nontemporal_store(v2f_t{v.x, v.z}, reinterpret_cast<v2f_t*>(a + m));
      if (has1) __builtin_nontemporal_store(v2f_t{v.y, v.w}, reinterpret_cast<v2f_t*>(b + m));
    } else {
      int o = m * C;
      asm volatile("" : "+v"(o));
      float* p0 = a + o;
      if (has1) {
        *reinterpret_cast<v2u_t*>(p0) = v2u_t{v.x, v.y};
        *reinterpret_cast<v2u_t*>(p0 + C) = v2u_t{v.z, v.w};
      } else {
        p0[0] = v.x;
        p0[C] = v.z;
      }
    }
  }
};
// pair p of a tensor [B, blocks_per_signal * N, C]: float offsets of its row(s), floats between successive blocks, the
// first of its two stream state rows ([B * C][N / 2]), and whether its second row exists
struct PairGeo {
  size_t off_a, off_b, block_stride;
  long long row0;
  bool has1;
};
template <int LAY>
__device__ __forceinline__ PairGeo pair_geo(long long p, int N, int B, int C, size_t blocks_per_signal) {
  PairGeo g;
  if constexpr (LAY == 2) {
    const int CP = (C + 1) / 2;
    const long long b0 = p / CP;
    const int c = 2 * (int)(p - b0 * CP);
    g.has1 = c + 1 < C;
    g.block_stride = (size_t)N * C;
    g.off_a = (size_t)b0 * blocks_per_signal * g.block_stride + c;
    g.off_b = g.off_a + (g.has1 ? 1 : 0);
    g.row0 = b0 * C + c;
  } else {
    g.has1 = LAY == 0 || 2 * p + 1 < B;
    g.block_stride = (size_t)N * (LAY == 1 ? 1 : 2);
    g.off_a = (size_t)(LAY == 1 ? 2 * p : p) * blocks_per_signal * g.block_stride;
    g.off_b = g.off_a + (g.has1 ? blocks_per_signal * g.block_stride : 0);   // (LAY 1: the next signal, or the same one again)
    g.row0 = 2 * p;
  }
  return g;
}

template <int NC, int NTC, int R0, int R1, int R2, int R3, int LAY, typename TX = float>   // TX: float, or int16_t = 16-bit PCM in
static __global__ __launch_bounds__((NTC > kThreads ? NTC : kThreads), 2) void k_fwd_wave_v(const TX* __restrict__ x, float* __restrict__ X,
                                                          const float* __restrict__ prev_block, const v4f_t* __restrict__ coefv,
                                                          const float* __restrict__ ctab, int Kin, int F, int N_rt, long long ntasks,
                                                          int T, int nstrip, int B, int C, int adj, WavePlan wp) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int N = NC ? NC : N_rt, nt = NC ? NTC : wp.nt, gpw = (int)blockDim.x / nt, grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  constexpr bool GRP = wave_in_place(NTC, R1);   // the frame transformed in place (a frame on several waves: one frame per workgroup of NTC lanes)
  const int ps = NC ? pad_shift_ct(NC) : AC_PAD_SHIFT;
  const int per = GRP ? group_floats_per_frame(N, ps) : wave_floats_per_group(N, ps), h = N >> 1, q = N >> 2;
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)gpw * per);
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    tw[k] = cis_neg(ctab, 16 * k, N);
    tw[h + k] = cis_neg(ctab, 4 * k, N);
    if constexpr (!GRP) tw[2 * h + k] = cis_neg(ctab, 4 * k + 1, N);
  }
  __syncthreads();
  const WaveTabs tb = {tw, tw + 2 * h, tw + h};   // (the in-place form has no pre-twiddle table)
  const float2 pre0 = cis_neg(ctab, 1, N);        // exp(-i pi / (4 N))
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);
  cpair* Bp = reinterpret_cast<cpair*>(base);
  cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
  const float scale = (float)(1.0 / ((double)N * 1.4142135623730951));
  const long long wg = (long long)blockIdx.x * gpw + grp;
  if (wg >= ntasks) return;
  // task -> (pair, strip); LAY 2 with several frames per workgroup (adj): the channel pairs of one signal and strip sit in one
  // workgroup, so that the cache lines they share (a pair uses 8 of every 4 C bytes) come through one L1 (six channels,
  // N = 120: 2.2 -> 3.4 TB/s; a frame per workgroup measured slower that way: pairs of a signal then stay far apart)
  int sp;
  long long pr;
  if (LAY == 2 && adj) {
    const int CP = (C + 1) / 2;
    const long long rest = wg / CP;
    sp = (int)(rest % nstrip);
    pr = (rest / nstrip) * CP + (wg - rest * CP);
  } else {
    sp = (int)(wg % nstrip);
    pr = wg / nstrip;
  }
  const PairGeo gx = pair_geo<LAY>(pr, N, B, C, (size_t)Kin), gX = pair_geo<LAY>(pr, N, B, C, (size_t)F),
                gp = pair_geo<LAY>(pr, N, B, C, 1);
  const RowPair<LAY> rp = {gx.has1, C};
  const int n0 = sp * T, n1 = min(n0 + T, F);
  v4f_t d0[kWaveVSteps], d1[kWaveVSteps], cy[kWaveVSteps];
  auto load_block = [&](auto xa, auto xb) {
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        d0[s] = rp.load2(xa, xb, 2 * i);           // samples 2 i, 2 i + 1
        d1[s] = rp.load2(xa, xb, N - 2 - 2 * i);   // samples N - 2 - 2 i, N - 1 - 2 i
      }
    }
  };
  // a x + b y with one fixed rounding order (a product, then one fused multiply-add): the carry is formed at two places -- at
  // a strip's start and inside the frame loop -- and a frame must not depend on which one served it (chunked = one-shot, bit
  // for bit); left to the compiler's contraction the two sites may fuse the other product
  auto fold2 = [](float a, float x, float b, float y) { return __builtin_fmaf(a, x, b * y); };
  auto carry_of = [&](int s, int i) {   // (v[h - 2 - 2 i], v[h - 1 - 2 i]) of the NEXT frame
    const v4f_t g = coefv[2 * i + 1];
    return v4f_t{fold2(g.z, d0[s].z, g.w, d1[s].x), fold2(g.z, d0[s].w, g.w, d1[s].y), fold2(g.x, d0[s].x, g.y, d1[s].z),
                 fold2(g.x, d0[s].y, g.y, d1[s].w)};
  };
  {
    const bool have = n0 >= 1 || prev_block != nullptr;
    if (n0 >= 1) load_block(x + gx.off_a + (size_t)(n0 - 1) * gx.block_stride, x + gx.off_b + (size_t)(n0 - 1) * gx.block_stride);
    else if (prev_block) load_block(prev_block + gp.off_a, prev_block + gp.off_b);
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      cy[s] = (have && i < q) ? carry_of(s, i) : v4f_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (n0 < Kin) load_block(x + gx.off_a + (size_t)n0 * gx.block_stride, x + gx.off_b + (size_t)n0 * gx.block_stride);
#ifdef AC_WAVE_REBASE
  constexpr bool REBASE_W = NC != 0;
#else
  constexpr bool REBASE_W = wave_rebase(NC, false);
#endif
  const int tid_outer = tid;
  for (int n = n0; n < n1; ++n) {
    int boff = grp * per, tid_l = tid_outer;   // (see k_enc_wave_v: per-frame offsets from opaque copies)
    if constexpr (REBASE_W) asm volatile("" : "+v"(boff), "+v"(tid_l));
    const int tid = tid_l;
    float* base = smem + boff;
    float2* v = reinterpret_cast<float2*>(base);
    cpair* Bp = reinterpret_cast<cpair*>(base);
    cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
    const bool has_cur = n < Kin;
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        v4f_t hi = {0.f, 0.f, 0.f, 0.f};
        if (has_cur) {
          const v4f_t f = coefv[2 * i];
          hi = v4f_t{fold2(f.x, d0[s].x, f.y, d1[s].z), fold2(f.x, d0[s].y, f.y, d1[s].w), fold2(f.z, d0[s].z, f.w, d1[s].x),
                     fold2(f.z, d0[s].w, f.w, d1[s].y)};
        }
        *reinterpret_cast<v4f_t*>(v + h + 2 * i) = hi;
        *reinterpret_cast<v4f_t*>(v + h - 2 - 2 * i) = cy[s];
        if (has_cur) cy[s] = carry_of(s, i);
      }
    }
    if (n + 1 < n1 && n + 1 < Kin)   // lands during the transform
      load_block(x + gx.off_a + (size_t)(n + 1) * gx.block_stride, x + gx.off_b + (size_t)(n + 1) * gx.block_stride);
    group_sync<NTC>();
    if constexpr (GRP) dct4_group_ct<NC, NTC, R0, R1, R2, R3>(v, Bp, tb, pre0, tid);
    else if constexpr (NC != 0) dct4_wave_ct<NC, NTC, R0, R1, R2, R3>(v, Ap, Bp, tb, tid);
    else dct4_wave(v, Ap, Bp, tb, N, tid, nt, wp);
    float* Xa = X + gX.off_a + (size_t)n * gX.block_stride;
    float* Xb = X + gX.off_b + (size_t)n * gX.block_stride;
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < h) rp.store2(Xa, Xb, 2 * i, *reinterpret_cast<const v4f_t*>(v + 2 * i) * scale);
    }
    group_sync<NTC>();
  }
}

// ---- the fused encode of the LDS-FFT tier: k_fwd_wave_v with the masking model (run-structured form, ac_psy_runs_dev.h) in
// the same launch -- X is not read back from HBM; same device code on the same values as the stand-alone kernel (k_psy_runs),
// so X, tonality and threshold equal transform -> tonality -> threshold bit for bit.  In two phases per strip of frames:
//   1. frame by frame, while the spectrum is in LDS: X out, the intensities in its place (bin f at 8 f: the model's slot, the
//      partial sums where the transform's second buffer was), tonality and the 64 band intensities P_j -- the part of the model
//      that needs every bin.  t goes to its tensor, P to the head of the frame's (not yet written) threshold row: 512 bytes that
//      come back from L2 in phase 2.
//   2. after the strip's last frame, when the transform's registers are dead: four frames at a time side by side -- P and t back
//      in, spreading product on the matrix cores, threshold entries, and every bin's look-up straight into the threshold rows.
// The transform's loop keeps its registers and occupancy; the per-band arithmetic runs where it interleaves four frames.
//   up to 64 lanes per frame: the wave works on its 64 / NTC frames (one strip each) together -- it needs all 64 lanes (= bands)
//     on every frame, so every group of lanes walks the same number of steps;
//   a frame on NTC / 64 waves (filters_n above 1024): all waves store X, square and form the partial sums and their share of the
//     tonality sums (one interleaved accumulator each: runs::tonality_ways); the first wave finishes t and P; in phase 2 the
//     waves take groups of four frames in turn.
struct WaveEncArgs {
  const uint32_t* img;   // ac_psy_plan::d_runs
  runs::RunsParams rp;
  float* t;              // [B, F, 1, C]
  float* thr;            // as X
};
static inline __host__ __device__ constexpr bool enc_rebase(int N) {
  return N == 540 || N == 576 || N == 2160 || N == 2304 || N == 3200 || N == 3240 || N == 3600;
}
static inline __host__ __device__ constexpr int enc_r(int N) { return N <= 128 ? 1 : N <= 256 ? 2 : N <= 512 ? 4 : N <= 1024 ? 8 : N <= 2048 ? 16 : 32; }
constexpr int kEncSlot2 = 1536;   // bytes per frame in phase 2: G (512) + the threshold entries (1024)
// floats of LDS per frame: what the transform needs, or the model's largest slot; a frame on several waves: and room for every
// wave's four slots of phase 2
static inline __host__ __device__ constexpr int enc_floats_per_frame(int N, int nt, int ps) {
  const int fft = nt > 64 ? group_floats_per_frame(N, ps) : wave_floats_per_group(N, ps), slot = runs::runs_slot_max(N) / 4;
  const int ph2 = nt > 64 ? (nt / 64) * 4 * kEncSlot2 / 4 : 0;
  const int m = fft > slot ? fft : slot;
  return m > ph2 ? m : ph2;
}
template <int NC, int NTC, int R0, int R1, int R2, int R3, int LAY>
static __global__ __launch_bounds__((NTC > kThreads ? NTC : kThreads), 2) void k_enc_wave_v(const float* __restrict__ x, float* __restrict__ X,
                                                          const float* __restrict__ prev_block, const v4f_t* __restrict__ coefv,
                                                          const float* __restrict__ ctab, int Kin, int F, long long ntasks,
                                                          int T, int nstrip, int B, WaveEncArgs pa) {
  static_assert(NC != 0 && LAY <= 1, "instances only; stereo or mono rows");
  using runs::v2f;
  using runs::v4f;
  float* smem = reinterpret_cast<float*>(smem_raw);
  constexpr int N = NC, nt = NTC;
  constexpr bool GRP = NTC > 64;   // one frame per workgroup of NTC lanes, transformed in place
  constexpr int ps = pad_shift_ct(NC);
  constexpr int per = enc_floats_per_frame(NC, NTC, ps), h = N >> 1, q = N >> 2;
  constexpr int RQ = enc_r(NC);                                  // granule registers per lane of a 64-lane pass over a frame
  constexpr int FPW = GRP ? 1 : 64 / NTC;                        // frames per wave
  constexpr int NW = GRP ? NTC / 64 : 1;                         // waves per frame
  constexpr int FB = GRP ? 1 : (NTC == 64 ? 1 : NTC == 32 ? 2 : 4);   // frames side by side in phase 1
  static_assert(!GRP || NW == runs::tonality_ways(RQ), "a wave per tonality accumulator");
  const int gpw = (int)blockDim.x / nt, grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  const int lane = threadIdx.x & 63;
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)gpw * per);
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    tw[k] = cis_neg(ctab, 16 * k, N);
    tw[h + k] = cis_neg(ctab, 4 * k, N);
    if constexpr (!GRP) tw[2 * h + k] = cis_neg(ctab, 4 * k + 1, N);
  }
  uint32_t* pimg = reinterpret_cast<uint32_t*>(tw + (GRP ? 2 : 3) * h);   // the masking model's image (without the per-bin entry offsets)
  for (int i = threadIdx.x; i < pa.rp.lds_words / 4; i += blockDim.x) reinterpret_cast<uint4*>(pimg)[i] = reinterpret_cast<const uint4*>(pa.img)[i];
  __syncthreads();
  const WaveTabs tb = {tw, tw + 2 * h, tw + h};   // (the in-place form has no pre-twiddle table)
  const float2 pre0 = cis_neg(ctab, 1, N);        // exp(-i pi / (4 N))
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);
  cpair* Bp = reinterpret_cast<cpair*>(base);
  cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
  const float scale = (float)(1.0 / ((double)N * 1.4142135623730951));
  const long long wg0 = (long long)blockIdx.x * gpw + grp;
  // (a group past the last task stays: its lanes are bands of the wave's other frames; it works on task 0 and stores nothing)
  const bool live = wg0 < ntasks;
  if (GRP && !live) return;
  const long long wg = live ? wg0 : 0;
  const int sp = (int)(wg % nstrip);
  const long long pr = wg / nstrip;
  const PairGeo gx = pair_geo<LAY>(pr, N, B, LAY == 0 ? 2 : 1, (size_t)Kin), gX = pair_geo<LAY>(pr, N, B, LAY == 0 ? 2 : 1, (size_t)F),
                gp = pair_geo<LAY>(pr, N, B, LAY == 0 ? 2 : 1, 1);
  const RowPair<LAY> rp = {gx.has1, LAY == 0 ? 2 : 1};
  const int n0 = sp * T, n1 = live ? min(n0 + T, F) : n0;
  const size_t t_a = LAY == 0 ? (size_t)pr * F * 2 : (size_t)(2 * pr) * F, t_step = LAY == 0 ? 2 : 1;   // tonality of frame n, signal 0
  const size_t t_b = LAY == 0 ? 1 : (size_t)F;                                                           // ... signal 1, from there
  v4f_t d0[kWaveVSteps], d1[kWaveVSteps], cy[kWaveVSteps];
  auto load_block = [&](auto xa, auto xb) {
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        d0[s] = rp.load2(xa, xb, 2 * i);           // samples 2 i, 2 i + 1
        d1[s] = rp.load2(xa, xb, N - 2 - 2 * i);   // samples N - 2 - 2 i, N - 1 - 2 i
      }
    }
  };
  auto fold2 = [](float a, float x, float b, float y) { return __builtin_fmaf(a, x, b * y); };   // (one rounding order: see k_fwd_wave_v)
  auto carry_of = [&](int s, int i) {   // (v[h - 2 - 2 i], v[h - 1 - 2 i]) of the NEXT frame
    const v4f_t g = coefv[2 * i + 1];
    return v4f_t{fold2(g.z, d0[s].z, g.w, d1[s].x), fold2(g.z, d0[s].w, g.w, d1[s].y), fold2(g.x, d0[s].x, g.y, d1[s].z),
                 fold2(g.x, d0[s].y, g.y, d1[s].w)};
  };
  {
    const bool have = n0 >= 1 || prev_block != nullptr;
    if (n0 >= 1) load_block(x + gx.off_a + (size_t)(n0 - 1) * gx.block_stride, x + gx.off_b + (size_t)(n0 - 1) * gx.block_stride);
    else if (prev_block) load_block(prev_block + gp.off_a, prev_block + gp.off_b);
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      cy[s] = (have && i < q) ? carry_of(s, i) : v4f_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (n0 < Kin) load_block(x + gx.off_a + (size_t)n0 * gx.block_stride, x + gx.off_b + (size_t)n0 * gx.block_stride);
  // the wave's first frame region = its first slot; a frame's region is its slot (bytes `per` * 4 apart)
  char* wslot0 = reinterpret_cast<char*>(smem + (size_t)(GRP ? grp : (threadIdx.x >> 6) * FPW) * per);
  char* myslot = reinterpret_cast<char*>(base);
  constexpr int SLOT = per * 4;
  const runs::RunsGeo geo = runs::runs_geo(NC);
  // ---- phase 1
  // The lane's LDS offsets are invariant and get hoisted out of the frame loop -- dozens of registers (the instances stand at
  // 204 ... 256, two waves per SIMD).  Formed per frame from opaque copies of the lane's index and buffer offset the kernels take
  // 130 ... 170 registers, but most of them run SLOWER (B = 256 stereo, fused encode, 52 sizes: +3 ... +20 %); the sizes where it
  // measured faster (-5 ... -24 %) take that form: enc_rebase().
#ifdef AC_ENC_REBASE_ALL
  constexpr bool REBASE = true;
#else
  constexpr bool REBASE = enc_rebase(NC);
#endif
  const int tid_outer = tid;
  for (int it = 0; it < T; ++it) {   // (every group of a wave walks T steps: the model needs all 64 lanes on each)
    int boff = grp * per, tid_l = tid_outer;   // (opaque OFFSETS: an opaque pointer would lose its address space)
    if constexpr (REBASE) asm volatile("" : "+v"(boff), "+v"(tid_l));
    const int tid = tid_l;
    float* base = smem + boff;
    float2* v = reinterpret_cast<float2*>(base);
    cpair* Bp = reinterpret_cast<cpair*>(base);
    cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
    char* myslot = reinterpret_cast<char*>(base);
    const int n = n0 + it;
    const bool fr = n < n1;          // this group has a frame in this step
    if (GRP && !fr) break;
    const bool has_cur = n < Kin;
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        v4f_t hi = {0.f, 0.f, 0.f, 0.f};
        if (has_cur) {
          const v4f_t f = coefv[2 * i];
          hi = v4f_t{fold2(f.x, d0[s].x, f.y, d1[s].z), fold2(f.x, d0[s].y, f.y, d1[s].w), fold2(f.z, d0[s].z, f.w, d1[s].x),
                     fold2(f.z, d0[s].w, f.w, d1[s].y)};
        }
        *reinterpret_cast<v4f_t*>(v + h + 2 * i) = hi;
        *reinterpret_cast<v4f_t*>(v + h - 2 - 2 * i) = cy[s];
        if (has_cur) cy[s] = carry_of(s, i);
      }
    }
    if (n + 1 < n1 && n + 1 < Kin)   // lands during the transform (issued after it: 0.758 -> 0.808 ms at 960, 0.864 -> 0.950 at 4096)
      load_block(x + gx.off_a + (size_t)(n + 1) * gx.block_stride, x + gx.off_b + (size_t)(n + 1) * gx.block_stride);
    group_sync<NTC>();
    if constexpr (GRP) dct4_group_ct<NC, NTC, R0, R1, R2, R3>(v, Bp, tb, pre0, tid);
    else dct4_wave_ct<NC, NTC, R0, R1, R2, R3>(v, Ap, Bp, tb, tid);
    const size_t nn = fr ? (size_t)n : 0;
    float* Xa = X + gX.off_a + nn * gX.block_stride;
    float* Xb = X + gX.off_b + nn * gX.block_stride;
    float* Ta = pa.thr + gX.off_a + nn * gX.block_stride;
    float* Tb = pa.thr + gX.off_b + nn * gX.block_stride;
    // X out; the frame's intensities take its place (bin f at 8 f: the model's slot).  One wave per frame: the lane's granules
    // tid + 64 s are the ones the tonality sums take from it, so their intensities stay in registers for that
    v4f Ireg[NTC == 64 ? 2 * kWaveVSteps : 1];
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      v4f I = {0.f, 0.f, 0.f, 0.f};
      if (i < h) {
        const v4f_t r = *reinterpret_cast<const v4f_t*>(v + 2 * i) * scale;
        if (fr) rp.store2(Xa, Xb, 2 * i, r);
        I = runs::squares(r);
        *reinterpret_cast<v4f*>(v + 2 * i) = I;
      }
      if constexpr (NTC == 64) Ireg[s] = I;
    }
    group_sync<NTC>();
    auto isrc_of = [&](const char* slot) {
      return [=](int i) {
        return (RQ * 128 == NC || 64 * i + lane < h) ? *reinterpret_cast<const v4f*>(slot + 16 * (64 * i + lane)) : v4f{0.f, 0.f, 0.f, 0.f};
      };
    };
    if constexpr (GRP) {
      // a wave per tonality accumulator; the partial sums of 4, 16 (64) bins by all lanes
      const int w = tid >> 6;
      const v4f part = runs::lane_sums<RQ, 4>(isrc_of(myslot), pa.rp, lane, w, NW);
      runs::level_sums<1>(myslot, SLOT, 0, geo.o4, geo.n4, tid, nt);
      __syncthreads();
      runs::level_sums<1>(myslot, SLOT, geo.o4, geo.o16, geo.n16, tid, nt);
      if (pa.rp.n64 > 0) {
        __syncthreads();
        runs::level_sums<1>(myslot, SLOT, geo.o16, geo.o64, pa.rp.n64, tid, nt);
      }
      // the accumulators of waves 1 .. NW - 1 meet the first wave's behind the image (the launcher sizes it)
      v4f* xch = reinterpret_cast<v4f*>(pimg + pa.rp.lds_words);   // [NW - 1][64] v4f behind the image (the launcher sizes it)
      if (w > 0) xch[(w - 1) * 64 + lane] = part;
      __syncthreads();
      if (w == 0) {
        v4f acc1[1] = {part};
#pragma unroll
        for (int k = 1; k < NW; ++k) acc1[0] += xch[(k - 1) * 64 + lane];
        v2f t1[1], P1[1];
        runs::tonality_finish<1>(acc1, pa.rp, lane, t1);
        runs::band_sums<1>(pa.rp, runs::load_lane(pimg, lane), pimg, myslot, SLOT, lane, P1);
        if (lane == 0) {
          pa.t[t_a + nn * t_step] = t1[0].x;
          if (rp.has1) pa.t[t_a + nn * t_step + t_b] = t1[0].y;
        }
        // P_j as "bin j" of the threshold row: lanes j, j + 1 (j even) make one 16-byte granule
        const float px = __shfl_down(P1[0].x, 1, 64), py = __shfl_down(P1[0].y, 1, 64);
        if ((lane & 1) == 0) rp.store2(Ta, Tb, lane, v4f_t{P1[0].x, P1[0].y, px, py});
      }
    } else {
#pragma unroll 1
      for (int g0 = 0; g0 < FPW; g0 += FB) {
        char* slots = wslot0 + g0 * SLOT;
        v2f tf[FB], Pf[FB];
        if constexpr (NTC == 64) {
          static_assert(RQ == 2 * kWaveVSteps, "eight granules per lane");
          runs::tonality_from<RQ, FB, 0>([&](int, int i) { return Ireg[i]; }, pa.rp, lane, tf);
        } else {
          runs::tonality_from<RQ, FB, 4>([&](int fb, int i) { return isrc_of(slots + fb * SLOT)(i); }, pa.rp, lane, tf);
        }
        runs::level_sums<FB>(slots, SLOT, 0, geo.o4, geo.n4, lane);
        wave_sync_lds();
        runs::level_sums<FB>(slots, SLOT, geo.o4, geo.o16, geo.n16, lane);
        if (pa.rp.n64 > 0) {
          wave_sync_lds();
          runs::level_sums<FB>(slots, SLOT, geo.o16, geo.o64, pa.rp.n64, lane);
        }
        runs::band_sums<FB>(pa.rp, runs::load_lane(pimg, lane), pimg, slots, SLOT, lane, Pf);
        // P (lane = band) and t of frame fb to the lanes of its group: through the head of its slot (the intensities are done with)
        wave_sync_lds();
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
          *reinterpret_cast<v2f*>(slots + fb * SLOT + 8 * lane) = Pf[fb];
          if (lane == 0) *reinterpret_cast<v2f*>(slots + fb * SLOT + 512) = tf[fb];
        }
      }
      wave_sync_lds();
      if (fr) {
        if (tid == 0) {
          const v2f tm = *reinterpret_cast<const v2f*>(myslot + 512);
          pa.t[t_a + nn * t_step] = tm.x;
          if (rp.has1) pa.t[t_a + nn * t_step + t_b] = tm.y;
        }
        for (int i = tid; i < 32; i += nt) rp.store2(Ta, Tb, 2 * i, *reinterpret_cast<const v4f_t*>(myslot + 16 * i));   // P_j as "bin j" of the threshold row
      }
    }
    group_sync<NTC>();
  }
  // ---- phase 2: the strip's frames four at a time -- slots of kEncSlot2 bytes at the head of the wave's region
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores of P and t have landed (L2) before anything reads them back
    if constexpr (GRP) __syncthreads();
    const runs::RunsLane lc = runs::load_lane(pimg, lane);
    const int total = GRP ? n1 - n0 : T * FPW;         // frames of the wave (of the workgroup: GRP), in the order m = step * FPW + group
    const int w = GRP ? (tid >> 6) : 0;
    char* slots2 = GRP ? myslot + w * (4 * kEncSlot2) : wslot0;
    const int mygroup = GRP ? 0 : lane / NTC;
    // entry-offset words of the granules this lane stores: i = tid + s nt (a frame on several waves: i = lane + 64 s, read per use)
    uint32_t ew[GRP ? 1 : 2 * kWaveVSteps];
    if constexpr (!GRP) {
#pragma unroll
      for (int s = 0; s < 2 * kWaveVSteps; ++s) {
        const int i = tid + s * nt;
        ew[s] = i < h ? pa.img[runs::off_idx(pa.rp.lw, pa.rp.kb) + i] : 0u;
      }
    }
    for (int m0 = 4 * w; m0 < total; m0 += 4 * NW) {
      wave_sync_lds();   // the look-ups of the four frames before are done
      // P and t back in: the lanes of a frame's group read its row's head (L2) into the frame's slot.  A slot without a frame --
      // past the strip, or a group without a task -- keeps what it held: finite stand-ins (a frame's arithmetic touches only its
      // own rows of the matrix product and its own slot), nothing of it is stored
      bool mine[4];
      float* Ta[4];
      float* Tb[4];
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) {
        const int m = m0 + fb, g = GRP ? 0 : m % FPW, n = n0 + (GRP ? m : m / FPW);
        mine[fb] = g == mygroup && m < total && n < n1;
        const size_t nn = mine[fb] ? (size_t)n : 0;
        Ta[fb] = pa.thr + gX.off_a + nn * gX.block_stride;
        Tb[fb] = pa.thr + gX.off_b + nn * gX.block_stride;
        char* sl = slots2 + fb * kEncSlot2;
        if (mine[fb]) {
          const int l = GRP ? lane : tid;
          for (int i = l; i < 32; i += (GRP ? 64 : nt)) *reinterpret_cast<v4f_t*>(sl + 16 * i) = rp.load2(Ta[fb], Tb[fb], 2 * i);
          if (l == 0) *reinterpret_cast<v2f*>(sl + 512) = v2f{pa.t[t_a + nn * t_step], rp.has1 ? pa.t[t_a + nn * t_step + t_b] : 0.f};
        }
      }
      wave_sync_lds();
      v2f P4[4], t4[4];
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) {
        const char* sl = slots2 + fb * kEncSlot2;
        P4[fb] = *reinterpret_cast<const v2f*>(sl + 8 * lane);
        t4[fb] = *reinterpret_cast<const v2f*>(sl + 512);
      }
      runs::band_tail<4>(P4, t4, pa.rp, lc, pimg, slots2, kEncSlot2, lane);
      wave_sync_lds();
      if constexpr (GRP) {
#pragma unroll 2
        for (int i = lane; i < h; i += 64) {
          const uint32_t wi = pa.img[runs::off_idx(pa.rp.lw, pa.rp.kb) + i];
#pragma unroll
          for (int fb = 0; fb < 4; ++fb)
            if (mine[fb]) rp.store2(Ta[fb], Tb[fb], 2 * i, runs::entry_lookup(slots2 + fb * kEncSlot2, wi));
        }
      } else {
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
          if (!mine[fb]) continue;
#pragma unroll
          for (int s = 0; s < 2 * kWaveVSteps; ++s) {
            const int i = tid + s * nt;
            if (i < h) rp.store2(Ta[fb], Tb[fb], 2 * i, runs::entry_lookup(slots2 + fb * kEncSlot2, ew[s]));
          }
        }
      }
    }
  }
}

// the synthesis in the same form: wide spectrum loads (the next frame's issued before the overlap-add of this one), the two
// output samples j, N - 1 - j of a lane's pairs as two wide stores, the aliased half of the previous frame in registers
template <int NC, int NTC, int R0, int R1, int R2, int R3, int LAY, typename TX = float>   // TX: float, or int16_t = 16-bit PCM out
static __global__ __launch_bounds__((NTC > kThreads ? NTC : kThreads), 2) void k_inv_wave_v(const float* __restrict__ X, TX* __restrict__ x,
                                                          const float* __restrict__ tail_in, float* __restrict__ tail_out,
                                                          const v4f_t* __restrict__ coefv, const float* __restrict__ ctab, int Kp,
                                                          int nblk, int seg, int nseg, int N_rt, long long ntasks, int B,
                                                          int C, int adj, WavePlan wp) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int N = NC ? NC : N_rt, nt = NC ? NTC : wp.nt, gpw = (int)blockDim.x / nt, grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  constexpr bool GRP = wave_in_place(NTC, R1);
  const int ps = NC ? pad_shift_ct(NC) : AC_PAD_SHIFT;
  const int h = N >> 1, q = N >> 2, per = GRP ? group_floats_per_frame(N, ps) : wave_floats_per_group(N, ps);
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)gpw * per);
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    tw[k] = cis_neg(ctab, 16 * k, N);
    tw[h + k] = cis_neg(ctab, 4 * k, N);
    if constexpr (!GRP) tw[2 * h + k] = cis_neg(ctab, 4 * k + 1, N);
  }
  __syncthreads();
  const WaveTabs tb = {tw, tw + 2 * h, tw + h};
  const float2 pre0 = cis_neg(ctab, 1, N);
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);
  cpair* Bp = reinterpret_cast<cpair*>(base);
  cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
  const long long wg = (long long)blockIdx.x * gpw + grp;
  if (wg >= ntasks) return;
  int sgm;
  long long pr;
  if (LAY == 2 && adj) {   // (channel pairs of one signal and strip side by side: see k_fwd_wave_v)
    const int CP = (C + 1) / 2;
    const long long rest = wg / CP;
    sgm = (int)(rest % nseg);
    pr = (rest / nseg) * CP + (wg - rest * CP);
  } else {
    sgm = (int)(wg % nseg);
    pr = wg / nseg;
  }
  const PairGeo gX = pair_geo<LAY>(pr, N, B, C, (size_t)Kp), gx = pair_geo<LAY>(pr, N, B, C, (size_t)nblk);
  const RowPair<LAY> rp = {gx.has1, C};
  const v4f_t* cv = coefv + h;   // the synthesis half of the table
  const float scale = 2.0f * 1.4142135623730951f;
  const int nlast = nblk + (tail_out ? 1 : 0);
  const int n0 = sgm * seg;
  const size_t ts = (size_t)gx.row0 * h;   // stream state rows of the pair: ts, ts + h
  // the aliased half u_{n-1}[h + 2 i], [h + 2 i + 1] of the lane's pairs stays in registers from frame to frame
  v4f_t um[kWaveVSteps];
#pragma unroll
  for (int s = 0; s < kWaveVSteps; ++s) {
    const int i = tid + s * nt;
    um[s] = v4f_t{0.f, 0.f, 0.f, 0.f};
    if (n0 == 0 && tail_in && i < q) {
      um[s].x = tail_in[ts + 2 * i];
      um[s].z = tail_in[ts + 2 * i + 1];
      if (rp.has1) {
        um[s].y = tail_in[ts + h + 2 * i];
        um[s].w = tail_in[ts + h + 2 * i + 1];
      }
    }
  }
  v4f_t r[2 * kWaveVSteps];
  auto frame_ok = [&](int t) { const int n = n0 + t; return t < 0 || (n < Kp && n < nblk); };
  auto load_frame = [&](int t) {
    const float* Xa = X + gX.off_a + (size_t)(n0 + t) * gX.block_stride;
    const float* Xb = X + gX.off_b + (size_t)(n0 + t) * gX.block_stride;
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < h) r[s] = rp.load2(Xa, Xb, 2 * i);
    }
  };
  const int t0 = n0 >= 1 ? -1 : 0;
  if (frame_ok(t0)) load_frame(t0);
#ifdef AC_WAVE_REBASE
  constexpr bool REBASE_W = NC != 0;
#else
  constexpr bool REBASE_W = wave_rebase(NC, true);
#endif
  const int tid_outer = tid;
  for (int t = t0; t < seg; ++t) {
    int boff = grp * per, tid_l = tid_outer;
    if constexpr (REBASE_W) asm volatile("" : "+v"(boff), "+v"(tid_l));
    const int tid = tid_l;
    float* base = smem + boff;
    float2* v = reinterpret_cast<float2*>(base);
    cpair* Bp = reinterpret_cast<cpair*>(base);
    cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
    const int n = n0 + t;
    if (t >= 0 && n >= nlast) break;
    const bool has_n = frame_ok(t);
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < h) *reinterpret_cast<v4f_t*>(v + 2 * i) = has_n ? r[s] : v4f_t{0.f, 0.f, 0.f, 0.f};
    }
    {   // the next frame's loads land during the transform and the overlap-add
      const int tn = t + 1;
      if (tn < seg && n0 + tn < nlast && frame_ok(tn)) load_frame(tn);
    }
    group_sync<NTC>();
    if (has_n) {
      if constexpr (GRP) dct4_group_ct<NC, NTC, R0, R1, R2, R3>(v, Bp, tb, pre0, tid);
      else if constexpr (NC != 0) dct4_wave_ct<NC, NTC, R0, R1, R2, R3>(v, Ap, Bp, tb, tid);
      else dct4_wave(v, Ap, Bp, tb, N, tid, nt, wp);
    }
    if (t >= 0) {
      if (n < nblk) {
        TX* xa = x + gx.off_a + (size_t)n * gx.block_stride;
        TX* xb = x + gx.off_b + (size_t)n * gx.block_stride;
#pragma unroll
        for (int s = 0; s < kWaveVSteps; ++s) {
          const int i = tid + s * nt;
          if (i < q) {
            const v4f_t A = *reinterpret_cast<const v4f_t*>(v + h - 2 - 2 * i) * scale;   // u_n[h-2-2i], u_n[h-1-2i]
            const v4f_t Bm = um[s];                                                         // u_{n-1}[h+2i], [h+2i+1]
            const v4f_t c0 = cv[2 * i], c1 = cv[2 * i + 1];   // (s1, s2)(2i), (s1, s2)(2i+1) | (s3, s4)(2i), (s3, s4)(2i+1)
            // (one fixed rounding order, as the analysis kernels' fold2: the team form of this kernel returns the same bits)
            const v4f_t o0 = {ola2(c0.x, A.z, c0.y, Bm.x), ola2(c0.x, A.w, c0.y, Bm.y), ola2(c0.z, A.x, c0.w, Bm.z), ola2(c0.z, A.y, c0.w, Bm.w)};
            const v4f_t o1 = {ola2(c1.z, A.x, c1.w, Bm.z), ola2(c1.z, A.y, c1.w, Bm.w), ola2(c1.x, A.z, c1.y, Bm.x), ola2(c1.x, A.w, c1.y, Bm.y)};
            rp.store2(xa, xb, 2 * i, o0);
            rp.store2(xa, xb, N - 2 - 2 * i, o1);
          }
        }
      } else if (tail_out) {
#pragma unroll
        for (int s = 0; s < kWaveVSteps; ++s) {
          const int i = tid + s * nt;
          if (i < q) {
            tail_out[ts + 2 * i] = um[s].x;
            tail_out[ts + 2 * i + 1] = um[s].z;
            if (rp.has1) {
              tail_out[ts + h + 2 * i] = um[s].y;
              tail_out[ts + h + 2 * i + 1] = um[s].w;
            }
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) um[s] = *reinterpret_cast<const v4f_t*>(v + h + 2 * i) * scale;
    }
    group_sync<NTC>();
  }
}

// ---- the 16-byte kernels for channel counts other than one and two ("team" form): whole cache lines on both sides ---------
// A row of [filters_n, C] floats interleaves the channels; a channel pair's 8 bytes per sample are a third of each cache line
// at six channels, and the strided form above (LAY 2) pays for it at the L1's request rate (2 - 3 TB/s).  Here the CP = ceil(C / 2)
// groups of lanes that transform the channel pairs of ONE signal and strip form a team: the team moves whole rows between HBM
// and LDS in 16-byte pieces, consecutive lanes on consecutive addresses, and every group picks its pair's samples out of (puts
// them into) the row image in LDS -- the image takes the bytes of the team's transform buffers, which are dead between frames:
//   block n + 1: global -> registers while frame n transforms (as the stereo kernel's prefetch), registers -> row image,
//                barrier, (channel pair, sample pair) reads into the fold's registers, barrier, fold into the group's buffer;
//   spectrum n:  group's buffer -> registers, barrier, 8-byte writes into the row image, barrier, 16-byte reads -> global, barrier.
// The arithmetic is the strided form's on the same values: results equal it bit for bit.  A workgroup holds TPW teams that
// step through their strips together (the barriers are the workgroup's; a team past its last frame idles through them).
constexpr int kTeamDeclined = -12346;   // launch_*_wave_team: the shape has no team form, the caller takes the strided one
constexpr int kTeamChunks = 8;   // 16-byte pieces of a row per lane: N C / (4 CP lanes per frame) <= 4 C / CP <= 8
template <int NC, int NTC>
struct TeamIO {
  float* stage;     // the team's row image on the way in (and transform buffers)
  float* stage_out; // ... on the way out: behind the other where the team's buffers hold two images (the wave form), else the same
  int C, c0, u, TL, NCH, NCHL;   // NCH: 16-byte pieces of a row (0: a group outside every team), NCHL: the same for the loads
  bool has1, member;   // member: the group belongs to a team (the groups a workgroup has left over after its last whole team do not)
  v4f_t raw[kTeamChunks];
  // (a lane's offsets are formed where they are used: hoisted out of the frame loop -- they are loop invariant -- they spill)
  static __device__ __forceinline__ int here(int v) {
    asm volatile("" : "+v"(v));
    return v;
  }
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < kTeamChunks; ++k) raw[k] = v4f_t{0.f, 0.f, 0.f, 0.f};
  }
  // (no branch around a load: at the join the wave would wait for the loads before it -- the row would come in one piece at a
  // time; a lane past the row's end reads its last piece again)
  __device__ __forceinline__ void load_row(const float* row) {
    const int u0 = here(u), last = NCHL - 1;
#pragma unroll
    for (int k = 0; k < kTeamChunks; ++k) {
      const int j = min(u0 + k * TL, last);
      raw[k] = *reinterpret_cast<const v4f_t*>(row + 4 * (size_t)j);
    }
  }
  __device__ __forceinline__ void row_to_image() const {
    const int u0 = here(u);
#pragma unroll
    for (int k = 0; k < kTeamChunks; ++k) {
      const int j = u0 + k * TL;
      if (j < NCH) *reinterpret_cast<v4f_t*>(stage + 4 * j) = raw[k];
    }
  }
  __device__ __forceinline__ void image_to_global(float* row, bool act) const {
    const int u0 = here(u);
#pragma unroll
    for (int k = 0; k < kTeamChunks; ++k) {
      const int j = u0 + k * TL;
      if (j < NCH && act) __builtin_nontemporal_store(*reinterpret_cast<const v4f_t*>(stage_out + 4 * j), reinterpret_cast<v4f_t*>(row + 4 * (size_t)j));
    }
  }
  // samples m, m + 1 of the group's channel pair: (c0[m], c1[m], c0[m+1], c1[m+1]); a half-empty pair carries its one channel twice
  __device__ __forceinline__ v4f_t get2(int m) const {
    const float* p = stage + here(m * C + c0);
    const v2u_t a = *reinterpret_cast<const v2u_t*>(p), b = *reinterpret_cast<const v2u_t*>(p + C);
    return v4f_t{a.x, has1 ? a.y : a.x, b.x, has1 ? b.y : b.x};
  }
  __device__ __forceinline__ void put2(int m, v4f_t o) const {
    float* p = stage_out + here(m * C + c0);
    if (!member) return;
    if (has1) {
      *reinterpret_cast<v2u_t*>(p) = v2u_t{o.x, o.y};
      *reinterpret_cast<v2u_t*>(p + C) = v2u_t{o.z, o.w};
    } else {
      p[0] = o.x;
      p[C] = o.z;
    }
  }
};
template <int NC, int NTC, int R0, int R1, int R2, int R3>
static __global__ __launch_bounds__(512, 2) void k_fwd_wave_c(const float* __restrict__ x, float* __restrict__ X,
                                                          const float* __restrict__ prev_block, const v4f_t* __restrict__ coefv,
                                                          const float* __restrict__ ctab, int Kin, int F, long long nteams, int T,
                                                          int nstrip, int C, int CP, int TPW) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  constexpr int N = NC, nt = NTC, ps = pad_shift_ct(NC), h = N >> 1, q = N >> 2;
  constexpr bool GRP = NTC > 64;
  constexpr int per = GRP ? group_floats_per_frame(N, ps) : wave_floats_per_group(N, ps);
  const int gpw = (int)blockDim.x / nt, grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)gpw * per);
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    tw[k] = cis_neg(ctab, 16 * k, N);
    tw[h + k] = cis_neg(ctab, 4 * k, N);
    if constexpr (!GRP) tw[2 * h + k] = cis_neg(ctab, 4 * k + 1, N);
  }
  __syncthreads();
  const WaveTabs tb = {tw, tw + 2 * h, tw + h};
  const float2 pre0 = cis_neg(ctab, 1, N);
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);
  cpair* Bp = reinterpret_cast<cpair*>(base);
  cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
  const float scale = (float)(1.0 / ((double)N * 1.4142135623730951));
  const int team = min(grp / CP, max(TPW - 1, 0)), gi = grp - (grp / CP) * CP;
  const long long tk = (long long)blockIdx.x * TPW + grp / CP;
  const bool valid = grp / CP < TPW && tk < nteams;   // (no early exit: every lane takes part in the workgroup's barriers)
  const int sp = valid ? (int)(tk % nstrip) : 0;
  const long long b0 = valid ? tk / nstrip : 0;
  const size_t RS = (size_t)N * C;   // floats per row
  const float* xs = x + (size_t)b0 * Kin * RS;
  float* Xs = X + (size_t)b0 * F * RS;
  TeamIO<NC, NTC> io;
  io.stage = smem + (size_t)team * CP * per;
  io.stage_out = io.stage + (GRP ? 0 : (CP * per / 2) & ~3);   // (two images of N C <= 2 N CP floats in CP per >= 4.25 N CP)
  io.C = C;
  io.c0 = 2 * gi;
  io.has1 = io.c0 + 1 < C;
  io.member = grp / CP < TPW;
  io.NCH = io.member ? N * C / 4 : 0;
  io.u = gi * nt + tid;
  io.TL = CP * nt;
  io.NCHL = N * C / 4;
  io.clear();
  const int n0 = sp * T, n1 = valid ? min(n0 + T, F) : n0;
  v4f_t d0[kWaveVSteps], d1[kWaveVSteps], cy[kWaveVSteps];
  auto image_to_d = [&]() {
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        d0[s] = io.get2(2 * i);
        d1[s] = io.get2(N - 2 - 2 * i);
      }
    }
  };
  auto fold2 = [](float a, float x, float b, float y) { return __builtin_fmaf(a, x, b * y); };   // (as k_fwd_wave_v)
  auto carry_of = [&](int s, int i) {
    const v4f_t g = coefv[2 * i + 1];
    return v4f_t{fold2(g.z, d0[s].z, g.w, d1[s].x), fold2(g.z, d0[s].w, g.w, d1[s].y), fold2(g.x, d0[s].x, g.y, d1[s].z),
                 fold2(g.x, d0[s].y, g.y, d1[s].w)};
  };
  {
    const bool have = valid && (n0 >= 1 || prev_block != nullptr);
    if (have) io.load_row(n0 >= 1 ? xs + (size_t)(n0 - 1) * RS : prev_block + (size_t)b0 * RS);
    io.row_to_image();
    __syncthreads();
    image_to_d();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      cy[s] = (have && i < q) ? carry_of(s, i) : v4f_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (valid && n0 < Kin) io.load_row(xs + (size_t)n0 * RS);
  for (int it = 0; it < T; ++it) {
    const int n = n0 + it;
    const bool act = n < n1, has_cur = act && n < Kin;
    io.row_to_image();
    __syncthreads();
    image_to_d();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        v4f_t hi = {0.f, 0.f, 0.f, 0.f};
        if (has_cur) {
          const v4f_t f = coefv[2 * i];
          hi = v4f_t{fold2(f.x, d0[s].x, f.y, d1[s].z), fold2(f.x, d0[s].y, f.y, d1[s].w), fold2(f.z, d0[s].z, f.w, d1[s].x),
                     fold2(f.z, d0[s].w, f.w, d1[s].y)};
        }
        *reinterpret_cast<v4f_t*>(v + h + 2 * i) = hi;
        *reinterpret_cast<v4f_t*>(v + h - 2 - 2 * i) = cy[s];
        if (has_cur) cy[s] = carry_of(s, i);
      }
    }
    if (n + 1 < n1 && n + 1 < Kin) io.load_row(xs + (size_t)(n + 1) * RS);   // lands during the transform
    group_sync<NTC>();
    if (act) {   // (a frame on several waves: TPW = 1, so `act` is the workgroup's)
      if constexpr (GRP) dct4_group_ct<NC, NTC, R0, R1, R2, R3>(v, Bp, tb, pre0, tid);
      else dct4_wave_ct<NC, NTC, R0, R1, R2, R3>(v, Ap, Bp, tb, tid);
    }
    v4f_t o[2 * kWaveVSteps];
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < h) o[s] = *reinterpret_cast<const v4f_t*>(v + 2 * i) * scale;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < h) io.put2(2 * i, o[s]);
    }
    __syncthreads();
    io.image_to_global(Xs + (size_t)n * RS, act);
    if constexpr (GRP) __syncthreads();   // (the wave form's two images are apart: the next row may come in while this one goes out)
  }
}
template <int NC, int NTC, int R0, int R1, int R2, int R3>
static __global__ __launch_bounds__(512, 2) void k_inv_wave_c(const float* __restrict__ X, float* __restrict__ x,
                                                          const float* __restrict__ tail_in, float* __restrict__ tail_out,
                                                          const v4f_t* __restrict__ coefv, const float* __restrict__ ctab, int Kp,
                                                          int nblk, int seg, int nseg, long long nteams, int C, int CP, int TPW) {
  float* smem = reinterpret_cast<float*>(smem_raw);
  constexpr int N = NC, nt = NTC, ps = pad_shift_ct(NC), h = N >> 1, q = N >> 2;
  constexpr bool GRP = NTC > 64;
  constexpr int per = GRP ? group_floats_per_frame(N, ps) : wave_floats_per_group(N, ps);
  const int gpw = (int)blockDim.x / nt, grp = threadIdx.x / nt, tid = threadIdx.x - grp * nt;
  float2* tw = reinterpret_cast<float2*>(smem + (size_t)gpw * per);
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    tw[k] = cis_neg(ctab, 16 * k, N);
    tw[h + k] = cis_neg(ctab, 4 * k, N);
    if constexpr (!GRP) tw[2 * h + k] = cis_neg(ctab, 4 * k + 1, N);
  }
  __syncthreads();
  const WaveTabs tb = {tw, tw + 2 * h, tw + h};
  const float2 pre0 = cis_neg(ctab, 1, N);
  float* base = smem + (size_t)grp * per;
  float2* v = reinterpret_cast<float2*>(base);
  cpair* Bp = reinterpret_cast<cpair*>(base);
  cpair* Ap = reinterpret_cast<cpair*>(base + 4 * padded_len(h, ps));
  const int team = min(grp / CP, max(TPW - 1, 0)), gi = grp - (grp / CP) * CP;
  const long long tk = (long long)blockIdx.x * TPW + grp / CP;
  const bool valid = grp / CP < TPW && tk < nteams;
  const int sgm = valid ? (int)(tk % nseg) : 0;
  const long long b0 = valid ? tk / nseg : 0;
  const size_t RS = (size_t)N * C;
  const float* Xs = X + (size_t)b0 * Kp * RS;
  float* xs = x + (size_t)b0 * nblk * RS;
  TeamIO<NC, NTC> io;
  io.stage = smem + (size_t)team * CP * per;
  io.stage_out = io.stage + (GRP ? 0 : (CP * per / 2) & ~3);   // (two images of N C <= 2 N CP floats in CP per >= 4.25 N CP)
  io.C = C;
  io.c0 = 2 * gi;
  io.has1 = io.c0 + 1 < C;
  io.member = grp / CP < TPW;
  io.NCH = io.member ? N * C / 4 : 0;
  io.u = gi * nt + tid;
  io.TL = CP * nt;
  io.NCHL = N * C / 4;
  io.clear();
  const v4f_t* cv = coefv + h;
  const float scale = 2.0f * 1.4142135623730951f;
  const int nlast = nblk + (tail_out ? 1 : 0);
  const int n0 = sgm * seg;
  const size_t ts = (size_t)(b0 * C + io.c0) * h;   // stream state rows of the pair: ts, ts + h
  v4f_t um[kWaveVSteps];
#pragma unroll
  for (int s = 0; s < kWaveVSteps; ++s) {
    const int i = tid + s * nt;
    um[s] = v4f_t{0.f, 0.f, 0.f, 0.f};
    if (valid && n0 == 0 && tail_in && i < q) {
      um[s].x = tail_in[ts + 2 * i];
      um[s].z = tail_in[ts + 2 * i + 1];
      if (io.has1) {
        um[s].y = tail_in[ts + h + 2 * i];
        um[s].w = tail_in[ts + h + 2 * i + 1];
      }
    }
  }
  auto frame_ok = [&](int t) { const int n = n0 + t; return t < 0 || (n < Kp && n < nblk); };
  // every team walks t = -1 ... seg - 1 (the barriers are the workgroup's); a signal's first strip idles through t = -1
  auto runs_at = [&](int t) { return valid && (t >= 0 ? n0 + t < nlast : n0 >= 1); };
  if (runs_at(-1) && frame_ok(-1)) io.load_row(Xs + (size_t)(n0 - 1) * RS);
  else if (runs_at(0) && n0 == 0 && frame_ok(0)) io.load_row(Xs + (size_t)n0 * RS);
  for (int t = -1; t < seg; ++t) {
    const int n = n0 + t;
    const bool act = runs_at(t), has_n = act && frame_ok(t);
    v4f_t r[2 * kWaveVSteps];
    io.row_to_image();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < h) r[s] = io.get2(2 * i);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2 * kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < h) *reinterpret_cast<v4f_t*>(v + 2 * i) = has_n ? r[s] : v4f_t{0.f, 0.f, 0.f, 0.f};
    }
    {
      const int tn = t + 1;
      // (a first strip's frame 0 was loaded before the loop and passes through t = -1 in the registers)
      if (tn < seg && runs_at(tn) && frame_ok(tn) && !(t == -1 && n0 == 0)) io.load_row(Xs + (size_t)(n0 + tn) * RS);
    }
    group_sync<NTC>();
    if (has_n) {
      if constexpr (GRP) dct4_group_ct<NC, NTC, R0, R1, R2, R3>(v, Bp, tb, pre0, tid);
      else dct4_wave_ct<NC, NTC, R0, R1, R2, R3>(v, Ap, Bp, tb, tid);
    }
    const bool out = act && t >= 0 && n < nblk;
    v4f_t o0[kWaveVSteps], o1[kWaveVSteps];
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        const v4f_t A = *reinterpret_cast<const v4f_t*>(v + h - 2 - 2 * i) * scale;
        const v4f_t Bm = um[s];
        const v4f_t c0 = cv[2 * i], c1 = cv[2 * i + 1];
        o0[s] = v4f_t{ola2(c0.x, A.z, c0.y, Bm.x), ola2(c0.x, A.w, c0.y, Bm.y), ola2(c0.z, A.x, c0.w, Bm.z), ola2(c0.z, A.y, c0.w, Bm.w)};
        o1[s] = v4f_t{ola2(c1.z, A.x, c1.w, Bm.z), ola2(c1.z, A.y, c1.w, Bm.w), ola2(c1.x, A.z, c1.y, Bm.x), ola2(c1.x, A.w, c1.y, Bm.y)};
      }
    }
    if (act && t >= 0 && n >= nblk && tail_out) {
#pragma unroll
      for (int s = 0; s < kWaveVSteps; ++s) {
        const int i = tid + s * nt;
        if (i < q) {
          tail_out[ts + 2 * i] = um[s].x;
          tail_out[ts + 2 * i + 1] = um[s].z;
          if (io.has1) {
            tail_out[ts + h + 2 * i] = um[s].y;
            tail_out[ts + h + 2 * i + 1] = um[s].w;
          }
        }
      }
    }
    if (act) {
#pragma unroll
      for (int s = 0; s < kWaveVSteps; ++s) {
        const int i = tid + s * nt;
        if (i < q) um[s] = *reinterpret_cast<const v4f_t*>(v + h + 2 * i) * scale;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kWaveVSteps; ++s) {
      const int i = tid + s * nt;
      if (i < q) {
        io.put2(2 * i, o0[s]);
        io.put2(N - 2 - 2 * i, o1[s]);
      }
    }
    __syncthreads();
    io.image_to_global(xs + (size_t)n * RS, out);
    if constexpr (GRP) __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// tonality (psychoacoustic.py:102-120), one workgroup per (b, frame, c)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ inline T block_sum(T v, T* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T s = 0;
  for (int w = 0; w < kThreads / 64; ++w) s += red[w];
  return s;
}
// the reference's functions in the compute type
__device__ __forceinline__ float m_log(float x) { return logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_pow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double m_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
// maximum / minimum as tf.maximum / tf.minimum (psychoacoustic.py:113-116, 205-208, 331): a NaN operand gives NaN (fmax / fmin
// would return the other operand and turn a poisoned frame into finite numbers)
__device__ __forceinline__ float m_max(float a, float b) { return (a != a || b != b) ? a + b : fmaxf(a, b); }
__device__ __forceinline__ double m_max(double a, double b) { return (a != a || b != b) ? a + b : fmax(a, b); }
__device__ __forceinline__ float m_min(float a, float b) { return (a != a || b != b) ? a + b : fminf(a, b); }
__device__ __forceinline__ double m_min(double a, double b) { return (a != a || b != b) ? a + b : fmin(a, b); }

template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(kThreads) void k_tonality_generic(const TIO* __restrict__ X, TIO* __restrict__ t,
                                                               int C, int N) {
  __shared__ TC red[kThreads / 64];
  const TC eps = (TC)1e-14;
  const long long wg = blockIdx.x;   // (b*F + f)*C + c
  const int c = (int)(wg % C);
  const long long bf = wg / C;
  const TIO* Xi = X + (size_t)bf * N * C + c;
  TC slog = 0, ssq = 0;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const TC a = ldv(Xi + (size_t)k * C);
    const TC I = a * a;
    slog += m_log(m_max(eps, I));
    ssq += I;
  }
  slog = block_sum(slog, red);
  ssq = block_sum(ssq, red);
  if (threadIdx.x == 0) {
    const TC gm = m_exp(slog / (TC)N);
    const TC am = ssq / (TC)N + eps;
    const TC sfm = (TC)10 * m_log(gm / am) / (TC)2.302585092994046;
    stv(t + wg, m_min(sfm / (TC)-60, (TC)1));
  }
}

// ------------------------------------------------------------------------------------------------
// global masking threshold (psychoacoustic.py:122-148 with :169-210, :301-331), factorised form
// one workgroup per (b, frame, c)
// ------------------------------------------------------------------------------------------------
template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(kThreads) void k_threshold_generic(
    const TIO* __restrict__ X, const TIO* __restrict__ t, TIO* __restrict__ thr, TC drown, TC alpha,
    const int32_t* __restrict__ wb_ptr, const int32_t* __restrict__ wb_idx, const TC* __restrict__ wb_val,
    const int32_t* __restrict__ wi_ptr, const int32_t* __restrict__ wi_idx, const TC* __restrict__ wi_val,
    const TC* __restrict__ S, const TC* __restrict__ quiet, const TC* __restrict__ beta, int C, int N,
    int M) {
  TC* I = reinterpret_cast<TC*>(smem_raw);   // [N]
  TC* Q = I + N;                             // [M]
  TC* G = Q + M;                             // [M]
  const TC eps = (TC)1e-14;
  const long long wg = blockIdx.x;
  const int c = (int)(wg % C);
  const long long bf = wg / C;
  const TIO* Xi = X + (size_t)bf * N * C + c;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const TC a = ldv(Xi + (size_t)k * C);
    I[k] = a * a;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < M; j += kThreads) {   // _to_bark_intensity (:301-315)
    TC P = 0;
    for (int e = wb_ptr[j]; e < wb_ptr[j + 1]; ++e) P += I[wb_idx[e]] * wb_val[e];
    Q[j] = m_pow(m_max(eps, P), alpha);               // (:206)
  }
  __syncthreads();
  const TC tt = ldv(t + wg);
  for (int j = threadIdx.x; j < M; j += kThreads) {   // _masking_intensity_in_bark (:169-210)
    TC acc = 0;
    for (int i = 0; i < M; ++i) acc += Q[i] * S[(size_t)i * M + j];
    const TC offset = ((TC)1 - drown) * (tt * beta[j] + (TC)9 * tt + (TC)5.5);     // (:185-191)
    const TC fac = m_pow((TC)10, -alpha * offset / (TC)10);                         // (:197)
    const TC T = m_pow(m_max(eps, fac * acc), (TC)1 / alpha);                       // (:208)
    G[j] = m_max(T, quiet[j]);                                                      // (:144)
  }
  __syncthreads();
  TIO* out = thr + (size_t)bf * N * C + c;
  for (int k = threadIdx.x; k < N; k += kThreads) {   // _bark_intensity_to_freq_ampl (:317-331)
    TC acc = 0;
    for (int e = wi_ptr[k]; e < wi_ptr[k + 1]; ++e) acc += G[wi_idx[e]] * wi_val[e];
    stv(out + (size_t)k * C, m_sqrt(m_max(eps, acc)));
  }
}

// ------------------------------------------------------------------------------------------------
// backward passes of the masking model (adjoints of the two kernels above); one workgroup per (b, frame, c)
// ------------------------------------------------------------------------------------------------
// t = min(c' [mean_f ln max(eps, I_f) - ln(mean_f I_f + eps)], 1), c' = (10 / ln 10) / (-60), I_f = X_f^2:
// d t / d X_f = c' (1/N) ([I_f > eps] / I_f - 1 / (mean I + eps)) 2 X_f   where the clamp is inactive
// TIO: the tensors' element type (float, double, bfloat16); TC: the arithmetic and the constant tables (float, double)
template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(kThreads) void k_tonality_bwd_generic(const TIO* __restrict__ X,
                                                                   const TIO* __restrict__ gt,
                                                                   TIO* __restrict__ gX, int accumulate, int C,
                                                                   int N) {
  __shared__ TC red[kThreads / 64];
  const TC eps = (TC)1e-14;
  const long long wg = blockIdx.x;   // (b*F + f)*C + c
  const int c = (int)(wg % C);
  const long long bf = wg / C;
  const TIO* Xi = X + (size_t)bf * N * C + c;
  TIO* gi = gX + (size_t)bf * N * C + c;
  TC slog = 0, ssq = 0;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const TC a = ldv(Xi + (size_t)k * C);
    slog += m_log(a * a > eps ? a * a : eps);
    ssq += a * a;
  }
  slog = block_sum(slog, red);
  ssq = block_sum(ssq, red);
  const TC am = ssq / (TC)N + eps;
  const TC cc = ((TC)10 / (TC)2.302585092994046) / (TC)-60;
  const TC tt = cc * (slog / (TC)N - m_log(am));
  const TC g = (tt < (TC)1) ? (TC)ldv(gt + wg) * cc / (TC)N : (TC)0;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const TC a = ldv(Xi + (size_t)k * C);
    const TC I = a * a;
    const TC d = g * ((I > eps ? (TC)1 / I : (TC)0) - (TC)1 / am) * (TC)2 * a;
    stv(gi + (size_t)k * C, accumulate ? (TC)ldv(gi + (size_t)k * C) + d : d);
  }
}

// thr_f = sqrt(max(eps, E_f)), E_f = sum_j G_j Winv[j,f], G_j = max(T_j, quiet_j), T_j = max(eps, Y_j)^(1/alpha),
// Y_j = fac_j A_j, A_j = sum_i Q_i S[i,j], Q_i = max(eps, P_i)^alpha, P_i = sum_f X_f^2 W[f,i],
// fac_j = 10^(-alpha O_j / 10), O_j = (1 - drown)(t beta_j + 9 t + 5.5).  The adjoint walks the chain backwards;
// every max() passes the gradient to its active branch.
template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(kThreads) void k_threshold_bwd_generic(
    const TIO* __restrict__ X, const TIO* __restrict__ t, const TIO* __restrict__ gthr, TIO* __restrict__ gX,
    TIO* __restrict__ gt, TC drown, TC alpha,
    const int32_t* __restrict__ wb_ptr, const int32_t* __restrict__ wb_idx, const TC* __restrict__ wb_val,
    const int32_t* __restrict__ wi_ptr, const int32_t* __restrict__ wi_idx, const TC* __restrict__ wi_val,
    const int32_t* __restrict__ wf_ptr, const int32_t* __restrict__ wf_idx, const TC* __restrict__ wf_val,
    const int32_t* __restrict__ vb_ptr, const int32_t* __restrict__ vb_idx, const TC* __restrict__ vb_val,
    const TC* __restrict__ S, const TC* __restrict__ quiet, const TC* __restrict__ beta, int C, int N,
    int M, int s_in_lds) {
  TC* smem = reinterpret_cast<TC*>(smem_raw);
  __shared__ TC red[kThreads / 64];
  const TC kEps = (TC)1e-14;
  TC* xs = smem;        // [N] X
  TC* gE = xs + N;      // [N] d L / d E
  TC* P = gE + N;       // [M]
  TC* Q = P + M;        // [M]
  TC* A = Q + M;        // [M]
  TC* G = A + M;        // [M]
  TC* gA = G + M;       // [M]
  TC* gP = gA + M;      // [M]
  TC* part = gP + M;    // [4][M] partial sums of the band x band products
  TC* Ss = part + 4 * M;   // [M][M] spreading matrix (when it fits)
  const long long wg = blockIdx.x;
  const int c = (int)(wg % C);
  const long long bf = wg / C;
  const TIO* Xi = X + (size_t)bf * N * C + c;
  const TIO* gi = gthr + (size_t)bf * N * C + c;
  for (int k = threadIdx.x; k < N; k += kThreads) xs[k] = ldv(Xi + (size_t)k * C);
  if (s_in_lds)
    for (int k = threadIdx.x; k < M * M; k += kThreads) Ss[k] = S[k];
  const TC* Sm = s_in_lds ? Ss : S;
  __syncthreads();
  // the band loops run on four groups of 64 threads: group q takes every fourth term, partial sums meet in LDS
  const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int j0 = 0; j0 < M; j0 += 64) {
    const int j = j0 + l;
    TC p = 0.f;
    if (j < M)
      for (int e = wb_ptr[j] + q; e < wb_ptr[j + 1]; e += 4) p += xs[wb_idx[e]] * xs[wb_idx[e]] * wb_val[e];
    if (j < M) part[q * M + j] = p;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < M; j += kThreads) {
    const TC p = (part[j] + part[M + j]) + (part[2 * M + j] + part[3 * M + j]);
    P[j] = p;
    Q[j] = m_pow(p > kEps ? p : kEps, alpha);
  }
  __syncthreads();
  for (int j0 = 0; j0 < M; j0 += 64) {
    const int j = j0 + l;
    TC acc = 0.f;
    if (j < M)
      for (int i = q; i < M; i += 4) acc += Q[i] * Sm[(size_t)i * M + j];
    if (j < M) part[q * M + j] = acc;
  }
  __syncthreads();
  const TC tt = t[wg];
  for (int j = threadIdx.x; j < M; j += kThreads) {
    const TC acc = (part[j] + part[M + j]) + (part[2 * M + j] + part[3 * M + j]);
    const TC fac = m_pow((TC)10, -alpha * ((TC)1 - drown) * (tt * beta[j] + (TC)9 * tt + (TC)5.5) / (TC)10);
    A[j] = acc;
    {
      const TC Tj = m_pow(fac * acc > kEps ? fac * acc : kEps, (TC)1 / alpha);
      G[j] = Tj > quiet[j] ? Tj : quiet[j];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < N; k += kThreads) {
    TC E = 0.f;
    for (int e = wi_ptr[k]; e < wi_ptr[k + 1]; ++e) E += G[wi_idx[e]] * wi_val[e];
    gE[k] = (E > kEps) ? (TC)ldv(gi + (size_t)k * C) * (TC)0.5 / m_sqrt(E) : (TC)0;
  }
  __syncthreads();
  for (int j0 = 0; j0 < M; j0 += 64) {
    const int j = j0 + l;
    TC gG = 0.f;
    if (j < M)
      for (int e = vb_ptr[j] + q; e < vb_ptr[j + 1]; e += 4) gG += gE[vb_idx[e]] * vb_val[e];
    if (j < M) part[q * M + j] = gG;
  }
  __syncthreads();
  TC gt_part = 0.f;
  for (int j = threadIdx.x; j < M; j += kThreads) {
    const TC gG = (part[j] + part[M + j]) + (part[2 * M + j] + part[3 * M + j]);
    const TC fac = m_pow((TC)10, -alpha * ((TC)1 - drown) * (tt * beta[j] + (TC)9 * tt + (TC)5.5) / (TC)10);
    const TC Y = fac * A[j];
    const TC T = m_pow(Y > kEps ? Y : kEps, (TC)1 / alpha);
    const TC gT = (T > quiet[j]) ? gG : (TC)0;
    const TC gY = (Y > kEps) ? gT * T / (alpha * Y) : (TC)0;
    gA[j] = gY * fac;
    // d fac / d t = fac (-alpha ln 10 / 10) (1 - drown) (beta_j + 9)
    gt_part += gY * A[j] * fac * (-alpha * (TC)0.2302585092994046) * ((TC)1 - drown) * (beta[j] + (TC)9);
  }
  gt_part = block_sum(gt_part, red);   // (contains the barriers that publish gA)
  if (threadIdx.x == 0) stv(gt + wg, gt_part);
  for (int i0 = 0; i0 < M; i0 += 64) {
    const int i = i0 + l;
    TC gQ = 0.f;
    if (i < M)
      for (int j = q; j < M; j += 4) gQ += Sm[(size_t)i * M + j] * gA[j];
    if (i < M) part[q * M + i] = gQ;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < M; i += kThreads) {
    const TC gQ = (part[i] + part[M + i]) + (part[2 * M + i] + part[3 * M + i]);
    gP[i] = (P[i] > kEps) ? gQ * alpha * Q[i] / P[i] : (TC)0;
  }
  __syncthreads();
  TIO* go = gX + (size_t)bf * N * C + c;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    TC gI = 0.f;
    for (int e = wf_ptr[k]; e < wf_ptr[k + 1]; ++e) gI += gP[wf_idx[e]] * wf_val[e];
    stv(go + (size_t)k * C, (TC)2 * xs[k] * gI);
  }
}

// ------------------------------------------------------------------------------------------------
// element-wise utilities
// ------------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));

// (db_of, normal_pair, noisy_of: ac_internal.h, shared with the fused encode epilogue)
// 16-byte vectors (n4 of them), one per thread, workgroups in address order (4 KB per workgroup: the store pattern the
// memory system rewards most, tools/ubench_write_pattern.hip) + a scalar tail; a, out 16-byte aligned when n4 > 0
static __global__ __launch_bounds__(256) void k_db(const float* __restrict__ a, float* __restrict__ out, size_t n, size_t n4,
                                            int norm) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    const f4 v = reinterpret_cast<const f4*>(a)[i];
    __builtin_nontemporal_store(f4{db_of(v.x, norm), db_of(v.y, norm), db_of(v.z, norm), db_of(v.w, norm)},
                                reinterpret_cast<f4*>(out) + i);
  } else {
    const size_t j = 4 * n4 + (i - n4);
    if (j < n) out[j] = db_of(a[j], norm);
  }
}

// d amplitude_to_dB / d a = (20 / ln 10) / a where a^2 > eps, else 0 (the clamp); the normalised form scales by 1 / 140
static __global__ __launch_bounds__(256) void k_db_bwd(const float* __restrict__ a, const float* __restrict__ g,
                                                float* __restrict__ ga, size_t n, int norm) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float c = norm ? (8.685889638065035f / 140.f) : 8.685889638065035f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = a[i];
    ga[i] = (v * v > kEps) ? g[i] * (c / v) : 0.f;
  }
}

// add_noise (psychoacoustic.py:150-167): out = X + thr * Normal(0, 1/6); X == nullptr stands for zeros (the gradient of
// add_noise with respect to the threshold is add_noise(0, grad_out) under the same seed)
static __global__ __launch_bounds__(256) void k_add_noise(const float* __restrict__ X, const float* __restrict__ thr,
                                                   float* __restrict__ out, size_t n, size_t n4, uint64_t seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t key = mix64(seed);
  if (i < n4) {
    const f4 x = X ? reinterpret_cast<const f4*>(X)[i] : f4{0.f, 0.f, 0.f, 0.f}, t = reinterpret_cast<const f4*>(thr)[i];
    float g0, g1, g2, g3;
    normal_pair(key, 2 * i, g0, g1);
    normal_pair(key, 2 * i + 1, g2, g3);
    __builtin_nontemporal_store(f4{noisy_of(x.x, t.x, g0), noisy_of(x.y, t.y, g1), noisy_of(x.z, t.z, g2), noisy_of(x.w, t.w, g3)},
                                reinterpret_cast<f4*>(out) + i);
  } else {
    const size_t j = 4 * n4 + (i - n4);
    if (j < n) {
      float g0, g1;
      normal_pair(key, j >> 1, g0, g1);
      out[j] = noisy_of(X ? X[j] : 0.f, thr[j], (j & 1) ? g1 : g0);
    }
  }
}

// the same two utilities for the other storage types (double: fp64 arithmetic; bfloat16: float32 arithmetic), one
// element per thread and iteration; the noise stream is the float32 one (same seed, same normals)
template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(256) void k_db_typed(const TIO* __restrict__ a, TIO* __restrict__ out, size_t n, int norm) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const TC v = ldv(a + i);
    TC dB = (TC)10 * m_log(m_max((TC)1e-14, v * v)) / (TC)2.302585092994046 + (TC)120;
    if (norm) dB = (dB + (TC)20) / (TC)140;
    stv(out + i, dB);
  }
}
template <typename TIO, typename TC = typename Compute<TIO>::type>
static __global__ __launch_bounds__(256) void k_add_noise_typed(const TIO* __restrict__ X, const TIO* __restrict__ thr,
                                                         TIO* __restrict__ out, size_t n, uint64_t seed) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const uint64_t key = mix64(seed);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float g0, g1;
    normal_pair(key, i >> 1, g0, g1);
    stv(out + i, (X ? (TC)ldv(X + i) : (TC)0) + (TC)ldv(thr + i) * ((TC)((i & 1) ? g1 : g0) / (TC)6));
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// even, from 16 to 4096, half of it 5-smooth (8 N floats of LDS <= 128 KB of the CU's 160 KB): the LDS-FFT middle tier applies
static bool lds_fft_ok(int N) {
  if (N < 16 || N > 4096 || (N & 1)) return false;
  int h = N / 2;
  for (int r : {2, 3, 5})
    while (h % r == 0) h /= r;
  return h == 1;
}
// dynamic LDS beyond the default 64 KB cap must be requested once per kernel and device: remembered, so that a launch in a
// streaming chain does not pay a driver call each time
template <typename K>
static int allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return AC_OK;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  int dev = 0;
  AC_HIP_CHECK(hipGetDevice(&dev));
  const std::pair<const void*, int> key(reinterpret_cast<const void*>(kernel), dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = granted.find(key);
  if (it != granted.end() && it->second >= bytes) return AC_OK;
  AC_HIP_CHECK(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  granted[key] = bytes;
  return AC_OK;
}

static int check_grid(long long n) {
  if (n <= 0) return 1;
  if (n > 2147483647ll) {
    set_error("problem too large for one launch (%lld workgroups)", n);
    return AC_EINVAL;
  }
  return 0;
}

// ---- the wave form of the LDS-FFT tier (filters_n <= 2048) ------------------------------------------------------------
#ifndef AC_LDS_WAVE_MAX
#define AC_LDS_WAVE_MAX 2048   // (0: the workgroup form everywhere, for A/B measurements)
#endif
// Which of the two forms serves a size is measured, not derived (profiles/r3/lds_fft_tier_sizes.txt, B = 64
// stereo, 10 s, same process).  float32 stereo rows with filters_n % 4 == 0 up to 1024 take the 16-byte wave kernels wherever
// the tier applies (every such size at 2.2 - 6.0 TB/s; the workgroup form 1.3 - 2.0).  Other layouts (mono, more channels,
// bfloat16) run the 8-byte wave kernels where they measured faster: the analysis up to filters_n = 1024 except where a frame
// gets 16 lanes and three passes (N / 2 from 97 to 127), the synthesis up to 1536.  AC_LDS_WAVE_MAX (tuning hook; 0: the
// workgroup form everywhere) caps both.
static bool lds_wave_ct_size(int N);
static bool lds_wave_vec_shape(int N, int C, bool f32) {
  return f32 && C >= 1 && N % 4 == 0 && (N <= 1024 || lds_wave_ct_size(N));   // (above 1024: the in-place instances only)
}
static int lds_wave_max() {
  static const int wave_max = [] { const char* e = getenv("AC_LDS_WAVE_MAX"); return e ? atoi(e) : AC_LDS_WAVE_MAX; }();
  return wave_max;
}
static bool lds_wave_ok(int N, bool synthesis, int C, bool f32) {
  const int wave_max = lds_wave_max();
  if (wave_max <= 0) return false;
  if (lds_wave_vec_shape(N, C, f32) && (lds_fft_ok(N) || lds_wave_ct_size(N))) return true;   // (instances reach 8192, the tier's other forms 4096)
  if (!lds_fft_ok(N) || N > wave_max) return false;
  const int H = N / 2;
  return synthesis ? N <= 1536 : (N <= 1024 && !(H > 96 && H < 128));
}
// Sizes with compile-time instances of the 16-byte kernels: filters_n, lanes per frame, super-radices (every filters_n % 4 == 0
// with a 5-smooth half up to 8192 -- the powers of two as well: the wave-level kernels of ac_fast.hip leave them the
// rectangular window and, below 1024, more than two channels; the plan the search below would pick, with lanes >= N / 16).  Strides, round counts and buffer offsets fold into immediates: 960 runs 0.156 -> 0.103 ms against the run-time form of
// the same kernel.  lds_wave_plan returns these plans, so the launch geometry and the instance agree by construction; any
// other size runs the run-time form.
// Plans re-measured against alternatives on the same tensors (tools/plan_ab.py, B = 256 stereo, transform / inverse ms): 576
// (8,6,6) 0.393 / 0.446 -> (6,8,6) 0.389 / 0.398; 7680 (10,8,8,6) 0.574 / 0.589 -> (8,8,10,6) 0.472 / 0.603; at 800, 1152, 2304,
// 2880 and 6144 four other orders and splits each ran within 2 % of (or behind) the plan listed: what keeps those sizes at
// 3.7 - 4.0 TB/s where 960 runs at 4.4 is not the split (profiles/r4/lds_fft_plan_ab.txt).
#ifndef AC_WAVE_CT_SIZES   // (a build for inspection may bring a shorter list)
#define AC_WAVE_CT_SIZES \
  AC_WAVE_CT(16, 4, 8, 0, 0, 0) \
  AC_WAVE_CT(20, 4, 10, 0, 0, 0) \
  AC_WAVE_CT(24, 4, 4, 3, 0, 0) \
  AC_WAVE_CT(32, 4, 4, 4, 0, 0) \
  AC_WAVE_CT(36, 4, 6, 3, 0, 0) \
  AC_WAVE_CT(40, 4, 5, 4, 0, 0) \
  AC_WAVE_CT(48, 4, 6, 4, 0, 0) \
  AC_WAVE_CT(60, 4, 10, 3, 0, 0) \
  AC_WAVE_CT(72, 8, 9, 4, 0, 0) \
  AC_WAVE_CT(80, 8, 8, 5, 0, 0) \
  AC_WAVE_CT(96, 8, 8, 6, 0, 0) \
  AC_WAVE_CT(100, 8, 10, 5, 0, 0) \
  AC_WAVE_CT(108, 8, 9, 6, 0, 0) \
  AC_WAVE_CT(120, 8, 10, 6, 0, 0) \
  AC_WAVE_CT(144, 16, 9, 8, 0, 0) \
  AC_WAVE_CT(160, 16, 10, 8, 0, 0) \
  AC_WAVE_CT(180, 16, 10, 9, 0, 0) \
  AC_WAVE_CT(192, 16, 8, 6, 2, 0) \
  AC_WAVE_CT(200, 16, 10, 10, 0, 0) \
  AC_WAVE_CT(216, 16, 9, 4, 3, 0) \
  AC_WAVE_CT(240, 16, 8, 5, 3, 0) \
  AC_WAVE_CT(288, 32, 6, 6, 4, 0) \
  AC_WAVE_CT(300, 32, 6, 5, 5, 0) \
  AC_WAVE_CT(320, 32, 8, 5, 4, 0) \
  AC_WAVE_CT(324, 32, 9, 6, 3, 0) \
  AC_WAVE_CT(360, 32, 6, 6, 5, 0) \
  AC_WAVE_CT(384, 32, 8, 6, 4, 0) \
  AC_WAVE_CT(400, 32, 8, 5, 5, 0) \
  AC_WAVE_CT(432, 32, 9, 8, 3, 0) \
  AC_WAVE_CT(480, 32, 10, 8, 3, 0) \
  AC_WAVE_CT(500, 32, 10, 5, 5, 0) \
  AC_WAVE_CT(540, 64, 9, 6, 5, 0) \
  AC_WAVE_CT(576, 64, 6, 8, 6, 0) \
  AC_WAVE_CT(600, 64, 10, 6, 5, 0) \
  AC_WAVE_CT(640, 64, 8, 8, 5, 0) \
  AC_WAVE_CT(648, 64, 9, 6, 6, 0) \
  AC_WAVE_CT(720, 64, 10, 6, 6, 0) \
  AC_WAVE_CT(768, 64, 8, 8, 6, 0) \
  AC_WAVE_CT(800, 64, 10, 8, 5, 0) \
  AC_WAVE_CT(864, 64, 9, 8, 6, 0) \
  AC_WAVE_CT(900, 64, 10, 9, 5, 0) \
  AC_WAVE_CT(960, 64, 10, 8, 6, 0) \
  AC_WAVE_CT(972, 64, 9, 9, 6, 0) \
  AC_WAVE_CT(1000, 64, 10, 10, 5, 0) \
  AC_WAVE_CT(1080, 128, 10, 9, 6, 0) \
  AC_WAVE_CT(1152, 128, 9, 8, 8, 0) \
  AC_WAVE_CT(1200, 128, 10, 10, 6, 0) \
  AC_WAVE_CT(1280, 128, 10, 8, 8, 0) \
  AC_WAVE_CT(1296, 128, 9, 9, 8, 0) \
  AC_WAVE_CT(1440, 128, 10, 9, 8, 0) \
  AC_WAVE_CT(1500, 128, 6, 5, 5, 5) \
  AC_WAVE_CT(1536, 128, 8, 8, 6, 2) \
  AC_WAVE_CT(1600, 128, 10, 10, 8, 0) \
  AC_WAVE_CT(1620, 128, 10, 9, 9, 0) \
  AC_WAVE_CT(1728, 128, 9, 8, 4, 3) \
  AC_WAVE_CT(1800, 128, 10, 10, 9, 0) \
  AC_WAVE_CT(1920, 128, 8, 8, 5, 3) \
  AC_WAVE_CT(1944, 128, 9, 9, 4, 3) \
  AC_WAVE_CT(2000, 128, 10, 10, 10, 0) \
  AC_WAVE_CT(2160, 256, 6, 6, 6, 5) \
  AC_WAVE_CT(2304, 256, 8, 6, 6, 4) \
  AC_WAVE_CT(2400, 256, 8, 6, 5, 5) \
  AC_WAVE_CT(2500, 256, 10, 5, 5, 5) \
  AC_WAVE_CT(2560, 256, 8, 8, 5, 4) \
  AC_WAVE_CT(2592, 256, 6, 6, 6, 6) \
  AC_WAVE_CT(2700, 256, 9, 6, 5, 5) \
  AC_WAVE_CT(2880, 256, 8, 6, 6, 5) \
  AC_WAVE_CT(2916, 256, 9, 9, 6, 3) \
  AC_WAVE_CT(3000, 256, 10, 6, 5, 5) \
  AC_WAVE_CT(3072, 256, 8, 8, 6, 4) \
  AC_WAVE_CT(3200, 256, 8, 8, 5, 5) \
  AC_WAVE_CT(3240, 256, 9, 9, 5, 4) \
  AC_WAVE_CT(3456, 256, 9, 8, 8, 3) \
  AC_WAVE_CT(3600, 256, 9, 8, 5, 5) \
  AC_WAVE_CT(3840, 256, 10, 8, 8, 3) \
  AC_WAVE_CT(3888, 256, 9, 9, 8, 3) \
  AC_WAVE_CT(4000, 256, 10, 8, 5, 5) \
  AC_WAVE_CT(4096, 256, 8, 8, 8, 4) \
  AC_WAVE_CT(4320, 512, 9, 8, 6, 5) \
  AC_WAVE_CT(4500, 512, 10, 9, 5, 5) \
  AC_WAVE_CT(4608, 512, 8, 8, 6, 6) \
  AC_WAVE_CT(4800, 512, 10, 8, 6, 5) \
  AC_WAVE_CT(4860, 512, 9, 9, 6, 5) \
  AC_WAVE_CT(5000, 512, 10, 10, 5, 5) \
  AC_WAVE_CT(5120, 512, 8, 8, 8, 5) \
  AC_WAVE_CT(5184, 512, 9, 8, 6, 6) \
  AC_WAVE_CT(5400, 512, 10, 9, 6, 5) \
  AC_WAVE_CT(5760, 512, 10, 8, 6, 6) \
  AC_WAVE_CT(5832, 512, 9, 9, 6, 6) \
  AC_WAVE_CT(6000, 512, 10, 10, 6, 5) \
  AC_WAVE_CT(6144, 512, 8, 8, 8, 6) \
  AC_WAVE_CT(6400, 512, 10, 8, 8, 5) \
  AC_WAVE_CT(6480, 512, 9, 9, 8, 5) \
  AC_WAVE_CT(6912, 512, 9, 8, 8, 6) \
  AC_WAVE_CT(7200, 512, 10, 9, 8, 5) \
  AC_WAVE_CT(7680, 512, 8, 8, 10, 6) \
  AC_WAVE_CT(7776, 512, 9, 9, 8, 6) \
  AC_WAVE_CT(8000, 512, 10, 10, 8, 5) \
  AC_WAVE_CT(8100, 512, 10, 9, 9, 5) \
  AC_WAVE_CT(8192, 512, 8, 8, 8, 8) \
  AC_WAVE_CT(64, 4, 8, 4, 0, 0) \
  AC_WAVE_CT(128, 8, 8, 8, 0, 0) \
  AC_WAVE_CT(256, 16, 8, 8, 2, 0) \
  AC_WAVE_CT(512, 32, 8, 8, 4, 0) \
  AC_WAVE_CT(1024, 64, 8, 8, 8, 0) \
  AC_WAVE_CT(2048, 128, 8, 8, 8, 2)
#endif
static bool lds_wave_ct_size(int N) {
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3) \
  if (N == NC) return true;
  AC_WAVE_CT_SIZES
#undef AC_WAVE_CT
  return false;
}
static bool wave_ct_off() {
  static const int off = [] { const char* e = getenv("AC_LDS_WAVE_NOCT"); return e ? atoi(e) : 0; }();   // (A/B measurements)
  return off != 0;
}
// super-radices of N / 2 (a pass of radix r runs (N / 2) / r butterflies of r points in registers): the factorisation with
// the least estimated work -- every pass costs a round trip through LDS, a butterfly ~ r (log2 r + 3) operations, and the
// butterflies of a pass are dealt to nt lanes
static WavePlan lds_wave_plan(int N, bool groups = true) {   // groups: frames dealt to more than one wave allowed (the 16-byte kernels)
  const int H = N / 2;
  WavePlan best{};
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                \
  if (N == NC && (groups || NTC <= 64)) {                                  \
    best.n = 1 + (R1 > 0) + (R2 > 0) + (R3 > 0);                           \
    best.r[0] = R0, best.r[1] = R1, best.r[2] = R2, best.r[3] = R3;        \
    best.nt = NTC;                                                         \
    return best;                                                           \
  }
  AC_WAVE_CT_SIZES
#undef AC_WAVE_CT
  int nt = 4;
  while (nt < 64 && nt < H / 8) nt <<= 1;
  best.nt = nt;
  static const int cand[] = {16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2};
  double best_cost = 1e300;
  int cur[6];
  auto cost_of = [&](int n) {
    double c = 0;
    for (int i = 0; i < n; ++i) {
      const int r = cur[i], nb = H / r;
      int lg = 0;
      while ((1 << lg) < r) ++lg;
      c += (double)((nb + nt - 1) / nt) * r * (lg + 3) + 24.0;
    }
    return c;
  };
  // depth-first over non-increasing factor sequences (the largest radix first: its pass has the stride-r writes the padding absorbs)
  struct Rec {
    static void go(int rem, int depth, int maxr, int* cur, const int* cand, double& best_cost, WavePlan& best,
                   const decltype(cost_of)& cost) {
      if (rem == 1) {
        const double c = cost(depth);
        if (c < best_cost) {
          best_cost = c;
          best.n = depth;
          for (int i = 0; i < depth; ++i) best.r[i] = (unsigned char)cur[i];
        }
        return;
      }
      if (depth == 6) return;
      for (int i = 0; i < 11; ++i) {
        const int r = cand[i];
        if (r > maxr || r > AC_WAVE_MAX_RADIX || rem % r) continue;
        cur[depth] = r;
        go(rem / r, depth + 1, r, cur, cand, best_cost, best, cost);
      }
    }
  };
  Rec::go(H, 0, 16, cur, cand, best_cost, best, cost_of);
  return best;
}
// waves per workgroup that leave the most waves resident per CU (160 KB of LDS)
static int lds_wave_block(int N, const WavePlan& wp, int extra_floats, size_t* lds_bytes, int ps = AC_PAD_SHIFT) {
  int best_w = 1;
  long best_res = 0;
  for (int w = 1; w <= 4; ++w) {
    const size_t lds = ((size_t)(64 / wp.nt) * w * (wave_floats_per_group(N, ps) + extra_floats) + 3 * N) * sizeof(float);
    const long res = (long)std::min<size_t>(8, 160 * 1024 / std::max<size_t>(lds, 1)) * w;
    if (lds <= 160 * 1024 && res >= best_res) {
      best_res = res;
      best_w = w;
    }
  }
  *lds_bytes = ((size_t)(64 / wp.nt) * best_w * (wave_floats_per_group(N, ps) + extra_floats) + 3 * N) * sizeof(float);   // + the three tables
  return best_w;
}

// the 16-byte kernels serve float32 stereo rows whose lanes cover a frame's sample pairs in four steps
static bool lds_wave_vec_ok(const ac_mdct_plan* p, const WavePlan& wp, int C) {
  static const int off = [] { const char* e = getenv("AC_LDS_WAVE_NOVEC"); return e ? atoi(e) : 0; }();   // (A/B measurements)
  return !off && lds_wave_vec_shape(p->N, C, true) && p->d_coefv && p->N / 4 <= kWaveVSteps * wp.nt && !(wp.nt > 64 && wave_ct_off());
}
// which rows a complex pair carries (RowPair): by channel count (the C ABI takes 16-byte aligned tensors; a pointer that is
// not -- an internal caller's -- gets the 4-byte layout)
static int wave_v_layout(int C, std::initializer_list<const void*> ptrs) {
  uintptr_t bits = 0;
  for (const void* q : ptrs) bits |= reinterpret_cast<uintptr_t>(q);
  if (C == 2 && !(bits & 15)) return 0;
  if (C == 1) return (bits & 7) ? -1 : 1;   // (a mono tensor off the 8-byte grid: the 8-byte kernels of the run-time forms)
  return 2;
}
// waves per workgroup, frames per workgroup and LDS bytes of the 16-byte kernels: the wave form packs frames as lds_wave_block
// says; a frame on more than one wave (in place) is a workgroup of its own with two tables behind its buffer
static void wave_v_geometry(int N, const WavePlan& wp, int* w, int* gpw, size_t* lds) {
  const int ps = (lds_wave_ct_size(N) && !wave_ct_off()) ? pad_shift_ct(N) : AC_PAD_SHIFT;   // (as the kernel that will run pads)
  const bool ct = lds_wave_ct_size(N) && !wave_ct_off();
  if (wp.nt > 64) {
    *w = wp.nt / 64;
    *gpw = 1;
    *lds = ((size_t)group_floats_per_frame(N, ps) + 2 * (size_t)N) * sizeof(float);
  } else if (ct && wave_in_place(wp.nt, wp.r[1])) {   // (in place inside a wave: half the floats per frame, two tables)
    int best_w = 1;
    long best_res = 0;
    for (int ww = 1; ww <= 4; ++ww) {
      const size_t bytes = ((size_t)(64 / wp.nt) * ww * group_floats_per_frame(N, ps) + 2 * (size_t)N) * sizeof(float);
      const long res = (long)std::min<size_t>(8, 160 * 1024 / std::max<size_t>(bytes, 1)) * ww;
      if (bytes <= 160 * 1024 && res >= best_res) {
        best_res = res;
        best_w = ww;
      }
    }
    *w = best_w;
    *gpw = best_w * (64 / wp.nt);
    *lds = ((size_t)*gpw * group_floats_per_frame(N, ps) + 2 * (size_t)N) * sizeof(float);
  } else {
    *w = lds_wave_block(N, wp, 0, lds, ps);
    *gpw = *w * (64 / wp.nt);
  }
}
// frames per strip of the 16-byte kernels: a strip pays `extra` frames' worth of work before its first frame (the block /
// the transform before it), a launch runs in rounds of as many workgroups as are resident; the least rounds x (frames + extra)
static int wave_strip(long long pairs, int per_sig, int gpw, int w, size_t lds, int cus, double extra) {
  static const int forced = [] { const char* e = getenv("AC_LDS_WAVE_STRIP"); return e ? atoi(e) : 0; }();   // (A/B measurements, tests)
  if (forced > 0) return std::min(forced, std::max(per_sig, 1));
  const int t_max = 32;
  const long resident = (long)cus * std::max<long>(1, std::min<long>(160 * 1024 / (long)std::max<size_t>(lds, 1), 8 / w));
  static const int cand[] = {32, 24, 16, 12, 8, 6, 4, 3, 2, 1};
  int best = 1;
  double best_cost = 1e300;
  for (int seg : cand) {
    if (seg > t_max && seg > 1) continue;
    const int len = std::min(seg, per_sig);
    const long long wgs = (pairs * ((per_sig + len - 1) / len) + gpw - 1) / gpw;
    const double rounds = wgs <= 4 * resident ? (double)((wgs + resident - 1) / resident) : (double)wgs / (double)resident;
    const double cost = rounds * (len + extra);
    if (cost < best_cost) {
      best_cost = cost;
      best = len;
    }
  }
  return best;
}
// sizes with instances on 16-bit PCM rows (stereo / mono): the frame lengths of the speech and music codecs this tier is for
#define AC_WAVE_PCM_SIZES             \
  AC_WAVE_CT(120, 8, 10, 6, 0, 0)    \
  AC_WAVE_CT(240, 16, 8, 5, 3, 0)    \
  AC_WAVE_CT(480, 32, 10, 8, 3, 0)   \
  AC_WAVE_CT(960, 64, 10, 8, 6, 0)   \
  AC_WAVE_CT(1920, 128, 8, 8, 5, 3)  \
  AC_WAVE_CT(576, 64, 6, 8, 6, 0)    \
  AC_WAVE_CT(1152, 128, 9, 8, 8, 0)
template <int LAY, typename TX = float>
static int launch_fwd_wave_v(const ac_mdct_plan* p, const TX* x, float* X, const float* prev_block, int B, int Kin, int F,
                             int C, hipStream_t s) {
  constexpr bool PCM = !std::is_same<TX, float>::value;
  const WavePlan wp = lds_wave_plan(p->N);
  size_t lds = 0;
  int w = 1, gpw = 1;
  wave_v_geometry(p->N, wp, &w, &gpw, &lds);
  const long long pairs = LAY == 0 ? (long long)B : LAY == 1 ? ((long long)B + 1) / 2 : (long long)B * ((C + 1) / 2);
  const int adj = LAY == 2 && gpw >= (C + 1) / 2;   // (see the kernel: channel pairs of a signal in one workgroup)
  const int T = wave_strip(pairs, F, gpw, w, lds, p->cus, 0.25);   // (every strip reads one block more than it has frames)
  const int nstrip = (F + T - 1) / T;
  const long long ntasks = pairs * nstrip;
  const long long g = (ntasks + gpw - 1) / gpw;
  const int st2 = check_grid(g);
  if (st2) return st2 < 0 ? st2 : AC_OK;
  int st = AC_OK;
  bool done = false;
  if constexpr (PCM) {
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                                                                   \
  if (!done && p->N == NC) {                                                                                                   \
    done = true;                                                                                                               \
    st = allow_lds(k_fwd_wave_v<NC, NTC, R0, R1, R2, R3, LAY, TX>, lds);                                                       \
    if (!st)                                                                                                                   \
      hipLaunchKernelGGL((k_fwd_wave_v<NC, NTC, R0, R1, R2, R3, LAY, TX>), dim3((unsigned)g), dim3(64 * w), lds, s, x, X,      \
                         prev_block, reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kin, F, p->N, ntasks, T, nstrip,   \
                         B, C, adj, wp);                                                                                       \
  }
    AC_WAVE_PCM_SIZES
#undef AC_WAVE_CT
    if (!done) return AC_EUNSUPPORTED;
    if (st) return st;
    AC_HIP_CHECK(hipGetLastError());
    return AC_OK;
  } else {
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                                                                   \
  if (!done && p->N == NC && !wave_ct_off()) {                                                                                 \
    done = true;                                                                                                               \
    st = allow_lds(k_fwd_wave_v<NC, NTC, R0, R1, R2, R3, LAY>, lds);                                                          \
    if (!st)                                                                                                                   \
      hipLaunchKernelGGL((k_fwd_wave_v<NC, NTC, R0, R1, R2, R3, LAY>), dim3((unsigned)g), dim3(64 * w), lds, s, x, X,         \
                         prev_block, reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kin, F, p->N, ntasks, T, nstrip,   \
                         B, C, adj, wp);                                                                                       \
  }
  AC_WAVE_CT_SIZES
#undef AC_WAVE_CT
  if (!done) {
    st = allow_lds(k_fwd_wave_v<0, 0, 0, 0, 0, 0, LAY>, lds);
    if (!st)
      hipLaunchKernelGGL((k_fwd_wave_v<0, 0, 0, 0, 0, 0, LAY>), dim3((unsigned)g), dim3(64 * w), lds, s, x, X, prev_block,
                         reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kin, F, p->N, ntasks, T, nstrip, B, C, adj, wp);
  }
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
  }
}
template <int LAY, typename TX = float>
static int launch_inv_wave_v(const ac_mdct_plan* p, const float* X, TX* x, const float* tail_in, float* tail_out, int B,
                             int Kp, int nblk, int C, hipStream_t s) {
  constexpr bool PCM = !std::is_same<TX, float>::value;
  const WavePlan wp = lds_wave_plan(p->N);
  size_t lds = 0;
  int w = 1, gpw = 1;
  wave_v_geometry(p->N, wp, &w, &gpw, &lds);
  const int per_sig = nblk + (tail_out ? 1 : 0);
  const long long pairs = LAY == 0 ? (long long)B : LAY == 1 ? ((long long)B + 1) / 2 : (long long)B * ((C + 1) / 2);
  const int adj = LAY == 2 && gpw >= (C + 1) / 2;
  const int seg = wave_strip(pairs, per_sig, gpw, w, lds, p->cus, 1.0);   // (every strip but a signal's first transforms one frame more)
  const int nseg = (per_sig + seg - 1) / seg;
  const long long ntasks = pairs * nseg;
  const long long g = (ntasks + gpw - 1) / gpw;
  const int st2 = check_grid(g);
  if (st2) return st2 < 0 ? st2 : AC_OK;
  int st = AC_OK;
  bool done = false;
  if constexpr (PCM) {
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                                                                   \
  if (!done && p->N == NC) {                                                                                                   \
    done = true;                                                                                                               \
    st = allow_lds(k_inv_wave_v<NC, NTC, R0, R1, R2, R3, LAY, TX>, lds);                                                       \
    if (!st)                                                                                                                   \
      hipLaunchKernelGGL((k_inv_wave_v<NC, NTC, R0, R1, R2, R3, LAY, TX>), dim3((unsigned)g), dim3(64 * w), lds, s, X, x,      \
                         tail_in, tail_out, reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kp, nblk, seg, nseg, p->N,  \
                         ntasks, B, C, adj, wp);                                                                               \
  }
    AC_WAVE_PCM_SIZES
#undef AC_WAVE_CT
    if (!done) return AC_EUNSUPPORTED;
    if (st) return st;
    AC_HIP_CHECK(hipGetLastError());
    return AC_OK;
  } else {
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                                                                   \
  if (!done && p->N == NC && !wave_ct_off()) {                                                                                 \
    done = true;                                                                                                               \
    st = allow_lds(k_inv_wave_v<NC, NTC, R0, R1, R2, R3, LAY>, lds);                                                          \
    if (!st)                                                                                                                   \
      hipLaunchKernelGGL((k_inv_wave_v<NC, NTC, R0, R1, R2, R3, LAY>), dim3((unsigned)g), dim3(64 * w), lds, s, X, x,         \
                         tail_in, tail_out, reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kp, nblk, seg, nseg, p->N,  \
                         ntasks, B, C, adj, wp);                                                                               \
  }
  AC_WAVE_CT_SIZES
#undef AC_WAVE_CT
  if (!done) {
    st = allow_lds(k_inv_wave_v<0, 0, 0, 0, 0, 0, LAY>, lds);
    if (!st)
      hipLaunchKernelGGL((k_inv_wave_v<0, 0, 0, 0, 0, 0, LAY>), dim3((unsigned)g), dim3(64 * w), lds, s, X, x, tail_in, tail_out,
                         reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kp, nblk, seg, nseg, p->N, ntasks, B, C, adj, wp);
  }
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
  }
}

#ifdef AC_WAVE_ROWS_TU
// ---- this file compiled again as ac_wave_rows.hip (AC_WAVE_ROWS_TU = 1: mono rows), ac_wave_rows2.hip (= 2: channel
// pairs of any channel count) and ac_wave_enc.hip (= 3: the fused encode): only the instances of the 16-byte kernels for
// that row layout / form and their launchers (the instances of one layout take a minute to compile: translation units of
// their own)
#if AC_WAVE_ROWS_TU == 3
// the fused encode (k_enc_wave_v): sizes with an instance whose frame region holds a slot of the model -- filters_n 108 ...
// 4096 (the masking model's range ends there; below 108 a frame's LDS is smaller than the model's smallest slot)
static bool enc_size(int N) { return N >= 108 && N <= 4096 && lds_wave_ct_size(N); }
// ... and where the one launch measured faster than transform + masking kernel on an MI355X (ratio <= 0.98 over B = 256 stereo
// clips of 10 s, profiles/r4/lds_fft_fused_encode_sweep.txt: 0.73 - 0.98; the instances left out ran 0.99 - 1.31 x -- the ones
// that spill registers, and the small sizes, where the per-frame part of the model outweighs the second read of X)
static bool enc_pays(int N) {
  static const int sizes[] = {540, 576, 640, 720, 768, 800, 864, 900, 960, 1000, 1152, 1200, 1296, 1440, 1500, 1536, 1728, 2160, 2304,
                              2400, 2500, 2560, 2592, 2700, 2880, 2916, 3000, 3072, 3200, 3240, 3456, 3600, 4096};
  for (int n : sizes)
    if (n == N) return true;
  return false;
}
template <int LAY>
static int launch_enc_wave_v(const ac_mdct_plan* p, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr,
                             float drown, const float* prev_block, int B, int Kin, int F, hipStream_t s) {
  const WavePlan wp = lds_wave_plan(p->N);
  const int N = p->N, ps = pad_shift_ct(N);
  WaveEncArgs pa;
  pa.img = psy->d_runs;
  pa.rp = runs_params(psy, drown, false);
  pa.t = t;
  pa.thr = thr;
  const int per = enc_floats_per_frame(N, wp.nt, ps);
  const size_t fixed = ((size_t)(wp.nt > 64 ? 2 : 3) * N + (size_t)pa.rp.lds_words) * sizeof(float) +
                       (wp.nt > 64 ? (size_t)(wp.nt / 64 - 1) * 1024 : 0);   // (... and the tonality accumulators of a frame's other waves)
  int w = 1, gpw = 1;
  size_t lds = 0;
  if (wp.nt > 64) {
    w = wp.nt / 64;
    lds = (size_t)per * sizeof(float) + fixed;
  } else {   // waves per workgroup that leave the most waves resident (the tables are paid per workgroup)
    long best = -1;
    for (int ww = 1; ww <= 4; ++ww) {
      const size_t b = (size_t)(64 / wp.nt) * ww * per * sizeof(float) + fixed;
      const long res = b > 160 * 1024 ? -1 : (long)std::min<size_t>(8 / ww, 160 * 1024 / b) * ww;
      if (res >= best) {
        best = res;
        w = ww;
        lds = b;
      }
    }
    if (best < 0) return AC_EUNSUPPORTED;
    gpw = w * (64 / wp.nt);
  }
  if (lds > 160 * 1024) return AC_EUNSUPPORTED;
  const long long pairs = LAY == 0 ? (long long)B : ((long long)B + 1) / 2;
  const int T = wave_strip(pairs, F, gpw, w, lds, p->cus, 0.25);
  const int nstrip = (F + T - 1) / T;
  const long long ntasks = pairs * nstrip;
  const long long g = (ntasks + gpw - 1) / gpw;
  const int st2 = check_grid(g);
  if (st2) return st2 < 0 ? st2 : AC_OK;
  int st = AC_OK;
  bool done = false;
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                                                                   \
  if constexpr (NC >= 108 && NC <= 4096) {                                                                                     \
    if (!done && N == NC) {                                                                                                    \
      done = true;                                                                                                             \
      st = allow_lds(k_enc_wave_v<NC, NTC, R0, R1, R2, R3, LAY>, lds);                                                         \
      if (!st)                                                                                                                 \
        hipLaunchKernelGGL((k_enc_wave_v<NC, NTC, R0, R1, R2, R3, LAY>), dim3((unsigned)g), dim3(64 * w), lds, s, x, X,        \
                           prev_block, reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kin, F, ntasks, T, nstrip, B, pa); \
    }                                                                                                                          \
  }
  AC_WAVE_CT_SIZES
#undef AC_WAVE_CT
  if (!done) {
    set_error("internal: no fused-encode instance for filters_n = %d", N);
    return AC_EUNSUPPORTED;
  }
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
// whether encode() on these tensors is the one fused launch of the LDS-FFT tier: float32 mono / stereo rows on the 8- / 16-byte
// grid, a size with an instance, a masking model with the run structure
bool wave_encode_fuses(const ac_mdct_plan* p, const ac_psy_plan* psy, int C, const void* x, const void* X, const void* thr) {
  const char* e = getenv("AC_LDS_WAVE_NOFUSE");   // (A/B measurements: read per call, so that one process can time both forms)
  const int off = e ? atoi(e) : 0;
  if (off == 1 || g_force_generic || !p || !psy || !psy->runs || p->N != psy->N || !enc_size(p->N) || wave_ct_off()) return false;
  if (off != 2 && !enc_pays(p->N)) return false;   // (2: every instance, as the tests run them)
  if (C < 1 || C > 2 || !lds_wave_vec_ok(p, lds_wave_plan(p->N), C)) return false;
  const int lay = wave_v_layout(C, {x, X, thr});
  return lay == 0 || lay == 1;
}
int launch_enc_wave(const ac_mdct_plan* p, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr, float drown,
                    const float* prev_block, int B, int Kin, int F, int C, hipStream_t s) {
  return C == 2 ? launch_enc_wave_v<0>(p, psy, x, X, t, thr, drown, prev_block, B, Kin, F, s)
                : launch_enc_wave_v<1>(p, psy, x, X, t, thr, drown, prev_block, B, Kin, F, s);
}
#elif AC_WAVE_ROWS_TU == 1
int launch_fwd_wave_mono(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin, int F,
                         hipStream_t s) {
  return launch_fwd_wave_v<1>(p, x, X, prev_block, B, Kin, F, 1, s);
}
int launch_inv_wave_mono(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out, int B, int Kp,
                         int nblk, hipStream_t s) {
  return launch_inv_wave_v<1>(p, X, x, tail_in, tail_out, B, Kp, nblk, 1, s);
}
int launch_fwd_wave_mono_pcm16(const ac_mdct_plan* p, const int16_t* x, float* X, int B, int Kin, int F, hipStream_t s) {
  return launch_fwd_wave_v<1, int16_t>(p, x, X, nullptr, B, Kin, F, 1, s);
}
int launch_inv_wave_mono_pcm16(const ac_mdct_plan* p, const float* X, int16_t* x, int B, int Kp, int nblk, hipStream_t s) {
  return launch_inv_wave_v<1, int16_t>(p, X, x, nullptr, nullptr, B, Kp, nblk, 1, s);
}
#elif AC_WAVE_ROWS_TU == 4
// the team form (k_fwd_wave_c / k_inv_wave_c): waves per workgroup, teams per workgroup and LDS bytes that keep the most channel
// pairs resident per CU; false when the shape has no place in it (a team is at most a workgroup of 1024 lanes; a lane moves at
// most kTeamChunks 16-byte pieces of a row)
// ... and where it measured faster than the strided channel pairs on an MI355X (geometric mean over C = 3, 4, 6 of team /
// strided <= 0.97, profiles/r4/lds_fft_team_sweep.txt: 0.57 - 0.97; the sizes left out ran 0.97 - 1.33 x -- the team form moves
// every byte once (PMC: reads 0.65 x, writes 0.62 x of the strided form's) but keeps six or eight waves per CU where the
// strided form keeps eight or nine, and the tier is bound by latency per wave, not by the memory side)
static bool team_pays(int N, bool inverse) {
  static const int fwd[] = {64, 80, 96, 100, 108, 120, 128, 144, 160, 180, 192, 200, 216, 240, 256, 324, 384, 400, 432, 480, 500, 512,
                            576, 600, 640, 648, 720, 768, 800, 864, 900, 960, 972, 1000, 1024, 1152, 1200, 1280, 1296, 1440, 1500, 1536,
                            1600, 1620, 1728, 1800, 1920, 2000, 2048, 3072, 3456, 3840, 3888, 4096};
  static const int inv[] = {64, 80, 96, 120, 128, 144, 160, 180, 192, 200, 240, 256, 320, 384, 432, 512, 640, 720, 768, 800, 1024, 1280,
                            1536, 2048, 3072, 4096};
  if (inverse) {
    for (int n : inv)
      if (n == N) return true;
  } else {
    for (int n : fwd)
      if (n == N) return true;
  }
  return false;
}
static bool team_geometry(int N, const WavePlan& wp, int C, bool inverse, int* w, int* tpw, size_t* lds) {
  const char* e = getenv("AC_LDS_WAVE_NOTEAM");   // (read per call -- A/B measurements, tests: 1 never, 2 wherever the shape fits)
  const int mode = e ? atoi(e) : 0;
  const int CP = (C + 1) / 2, ps = pad_shift_ct(N);
  if (mode == 1 || C < 3 || wave_ct_off() || !lds_wave_ct_size(N)) return false;
  if (mode != 2 && !team_pays(N, inverse)) return false;
  if ((long long)N * C > (long long)4 * kTeamChunks * CP * wp.nt) return false;
  if (wp.nt > 64) {
    if (CP * wp.nt > 512) return false;   // (the kernels' launch bound)
    *w = CP * wp.nt / 64;
    *tpw = 1;
    *lds = ((size_t)CP * group_floats_per_frame(N, ps) + 2 * (size_t)N) * sizeof(float);
    return *lds <= 160 * 1024;
  }
  const int gw = 64 / wp.nt;
  long best = 0;
  for (int ww = 1; ww <= 8; ++ww) {
    const int gpw = ww * gw, teams = gpw / CP;
    const size_t bytes = ((size_t)gpw * wave_floats_per_group(N, ps) + 3 * (size_t)N) * sizeof(float);
    if (teams < 1 || bytes > 160 * 1024) continue;
    const long resident = std::min<long>(160 * 1024 / (long)bytes, 8 / ww);   // (two waves per SIMD: the kernels take ~200 registers)
    const long useful = resident * teams * CP;
    if (useful > best) {
      best = useful;
      *w = ww;
      *tpw = teams;
      *lds = bytes;
    }
  }
  return best > 0;
}
int launch_fwd_wave_team(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin, int F, int C,
                         hipStream_t s) {
  const WavePlan wp = lds_wave_plan(p->N);
  size_t lds = 0;
  int w = 1, tpw = 1;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(prev_block)) & 15) ||
      !team_geometry(p->N, wp, C, false, &w, &tpw, &lds))
    return kTeamDeclined;
  const int CP = (C + 1) / 2;
  const int T = wave_strip((long long)B * CP, F, tpw * CP, w, lds, p->cus, 0.25);
  const int nstrip = (F + T - 1) / T;
  const long long nteams = (long long)B * nstrip, g = (nteams + tpw - 1) / tpw;
  const int st2 = check_grid(g);
  if (st2) return st2 < 0 ? st2 : AC_OK;
  int st = AC_OK;
  bool done = false;
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                                                                   \
  if (!done && p->N == NC) {                                                                                                   \
    done = true;                                                                                                               \
    st = allow_lds(k_fwd_wave_c<NC, NTC, R0, R1, R2, R3>, lds);                                                                \
    if (!st)                                                                                                                   \
      hipLaunchKernelGGL((k_fwd_wave_c<NC, NTC, R0, R1, R2, R3>), dim3((unsigned)g), dim3(64 * w), lds, s, x, X, prev_block,   \
                         reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kin, F, nteams, T, nstrip, C, CP, tpw);        \
  }
  AC_WAVE_CT_SIZES
#undef AC_WAVE_CT
  if (!done) return kTeamDeclined;
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
int launch_inv_wave_team(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out, int B, int Kp,
                         int nblk, int C, hipStream_t s) {
  const WavePlan wp = lds_wave_plan(p->N);
  size_t lds = 0;
  int w = 1, tpw = 1;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(X)) & 15) || !team_geometry(p->N, wp, C, true, &w, &tpw, &lds))
    return kTeamDeclined;
  const int CP = (C + 1) / 2;
  const int per_sig = nblk + (tail_out ? 1 : 0);
  const int seg = wave_strip((long long)B * CP, per_sig, tpw * CP, w, lds, p->cus, 1.0);
  const int nseg = (per_sig + seg - 1) / seg;
  const long long nteams = (long long)B * nseg, g = (nteams + tpw - 1) / tpw;
  const int st2 = check_grid(g);
  if (st2) return st2 < 0 ? st2 : AC_OK;
  int st = AC_OK;
  bool done = false;
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3)                                                                                   \
  if (!done && p->N == NC) {                                                                                                   \
    done = true;                                                                                                               \
    st = allow_lds(k_inv_wave_c<NC, NTC, R0, R1, R2, R3>, lds);                                                                \
    if (!st)                                                                                                                   \
      hipLaunchKernelGGL((k_inv_wave_c<NC, NTC, R0, R1, R2, R3>), dim3((unsigned)g), dim3(64 * w), lds, s, X, x, tail_in,      \
                         tail_out, reinterpret_cast<const v4f_t*>(p->d_coefv), p->d_ctab, Kp, nblk, seg, nseg, nteams, C, CP,  \
                         tpw);                                                                                                 \
  }
  AC_WAVE_CT_SIZES
#undef AC_WAVE_CT
  if (!done) return kTeamDeclined;
  if (st) return st;
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
#else
int launch_fwd_wave_strided(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin, int F,
                            int C, hipStream_t s) {
  return launch_fwd_wave_v<2>(p, x, X, prev_block, B, Kin, F, C, s);
}
int launch_inv_wave_strided(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out, int B,
                            int Kp, int nblk, int C, hipStream_t s) {
  return launch_inv_wave_v<2>(p, X, x, tail_in, tail_out, B, Kp, nblk, C, s);
}
#endif
#else

// returned by launch_fwd_wave / launch_inv_wave when the tensors at hand are not for the 16-byte kernels (rows off the 16-byte
// grid, tuning hooks) and the size is past the 8-byte wave kernels' range: the caller goes on to the next tier
constexpr int kWaveDeclined = -12345;
static bool wave_8_byte_range(int N) { return lds_fft_ok(N) && N <= lds_wave_max(); }
template <typename TIO>
static int launch_fwd_wave(const ac_mdct_plan* p, const TIO* x, TIO* X, const TIO* prev_block, int B, int Kin, int F, int C,
                           hipStream_t s) {
  const WavePlan wp0 = lds_wave_plan(p->N);
  if constexpr (std::is_same<TIO, float>::value)
    if (lds_wave_vec_ok(p, wp0, C)) {
      const int lay = wave_v_layout(C, {x, X, prev_block});
      if (lay >= 0) return lay == 0 ? launch_fwd_wave_v<0>(p, x, X, prev_block, B, Kin, F, 2, s)
             : lay == 1 ? launch_fwd_wave_mono(p, x, X, prev_block, B, Kin, F, s)
                        : [&] {   // whole cache lines where the shape has a team form, the strided channel pairs elsewhere
                            const int st = launch_fwd_wave_team(p, x, X, prev_block, B, Kin, F, C, s);
                            return st != kTeamDeclined ? st : launch_fwd_wave_strided(p, x, X, prev_block, B, Kin, F, C, s);
                          }();
    }
  if (!wave_8_byte_range(p->N)) return kWaveDeclined;
  const WavePlan wp = lds_wave_plan(p->N, false);
  size_t lds = 0;
  const int w = lds_wave_block(p->N, wp, 0, &lds);
  const int CP = (C + 1) / 2, gpw = w * (64 / wp.nt);
  const long long ntasks = (long long)B * CP * F;
  int T = 4;
  while (T > 1 && ntasks < (long long)gpw * T * p->cus * 4) T >>= 1;
  const int st = allow_lds(k_fwd_wave<TIO>, lds);
  if (st) return st;
  const long long g = (ntasks + (long long)gpw * T - 1) / ((long long)gpw * T);
  const int st2 = check_grid(g);
  if (st2) return st2 < 0 ? st2 : AC_OK;
  hipLaunchKernelGGL(k_fwd_wave<TIO>, dim3((unsigned)g), dim3(64 * w), lds, s, x, X, prev_block, p->d_coef, p->d_ctab, Kin, F, C,
                     CP, p->N, ntasks, T, wp);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
template <typename TIO>
static int launch_inv_wave(const ac_mdct_plan* p, const TIO* X, TIO* x, const float* tail_in, float* tail_out, int B, int Kp,
                           int nblk, int C, hipStream_t s) {
  const WavePlan wp0 = lds_wave_plan(p->N);
  if constexpr (std::is_same<TIO, float>::value)
    if (lds_wave_vec_ok(p, wp0, C)) {
      const int lay = wave_v_layout(C, {X, x});
      if (lay >= 0) return lay == 0 ? launch_inv_wave_v<0>(p, X, x, tail_in, tail_out, B, Kp, nblk, 2, s)
             : lay == 1 ? launch_inv_wave_mono(p, X, x, tail_in, tail_out, B, Kp, nblk, s)
                        : [&] {
                            const int st = launch_inv_wave_team(p, X, x, tail_in, tail_out, B, Kp, nblk, C, s);
                            return st != kTeamDeclined ? st : launch_inv_wave_strided(p, X, x, tail_in, tail_out, B, Kp, nblk, C, s);
                          }();
    }
  if (!wave_8_byte_range(p->N)) return kWaveDeclined;
  const WavePlan wp = lds_wave_plan(p->N, false);
  size_t lds = 0;
  const int w = lds_wave_block(p->N, wp, p->N, &lds);
  const int per_sig = nblk + (tail_out ? 1 : 0);
  const int CP = (C + 1) / 2, gpw = w * (64 / wp.nt);
  // blocks per strip: every strip but a signal's first pays one more transform for the frame before it
  int seg = 8;
  while (seg > 1 && (long long)B * CP * ((per_sig + seg - 1) / seg) < (long long)gpw * p->cus * 4) seg >>= 1;
  const int nseg = (per_sig + seg - 1) / seg;
  const long long ntasks = (long long)B * CP * nseg;
  const int st = allow_lds(k_inv_wave<TIO>, lds);
  if (st) return st;
  const long long g = (ntasks + gpw - 1) / gpw;
  const int st2 = check_grid(g);
  if (st2) return st2 < 0 ? st2 : AC_OK;
  hipLaunchKernelGGL(k_inv_wave<TIO>, dim3((unsigned)g), dim3(64 * w), lds, s, X, x, tail_in, tail_out, p->d_coef, p->d_ctab, Kp,
                     nblk, seg, nseg, C, CP, p->N, ntasks, wp);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

// 16-bit PCM at the boundary on the instances of AC_WAVE_PCM_SIZES (mono / stereo): AC_EUNSUPPORTED elsewhere
static bool wave_pcm_size(int N) {
#define AC_WAVE_CT(NC, NTC, R0, R1, R2, R3) \
  if (N == NC) return true;
  AC_WAVE_PCM_SIZES
#undef AC_WAVE_CT
  return false;
}
bool lds_fft_serves_pcm16(const ac_mdct_plan* p, int C) {
  return !g_force_generic && !wave_ct_off() && p->d_coefv && (C == 1 || C == 2) && wave_pcm_size(p->N);
}
int launch_fwd_lds_pcm16(const ac_mdct_plan* p, const int16_t* x, float* X, int B, int K, int C, hipStream_t s) {
  return C == 2 ? launch_fwd_wave_v<0, int16_t>(p, x, X, nullptr, B, K, K + 1, 2, s) : launch_fwd_wave_mono_pcm16(p, x, X, B, K, K + 1, s);
}
int launch_inv_lds_pcm16(const ac_mdct_plan* p, const float* X, int16_t* x, int B, int Kp, int C, hipStream_t s) {
  return C == 2 ? launch_inv_wave_v<0, int16_t>(p, X, x, nullptr, nullptr, B, Kp, Kp + 1, 2, s)
                : launch_inv_wave_mono_pcm16(p, X, x, B, Kp, Kp + 1, s);
}

// which LDS-FFT form serves float32 tensors of C channels at this plan's size: 2 = a compile-time instance of the 16-byte
// kernels, 1 = the run-time forms of the tier, 0 = none (the O(N^2) kernels)
int lds_fft_tier_of(const ac_mdct_plan* p, int C) {
  if (g_force_generic) return 0;
  if (lds_wave_ok(p->N, false, C, true) && lds_wave_vec_ok(p, lds_wave_plan(p->N), C)) return lds_wave_ct_size(p->N) && !wave_ct_off() ? 2 : 1;
  return lds_fft_ok(p->N) ? 1 : 0;
}

int launch_fwd_generic(const ac_mdct_plan* p, const float* x, float* X, const float* prev_block, int B, int Kin,
                       int F, int C, hipStream_t s) {
  const long long nwg = (long long)B * C * F;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  if (lds_wave_ok(p->N, false, C, true) && !g_force_generic) {
    const int r = launch_fwd_wave<float>(p, x, X, prev_block, B, Kin, F, C, s);
    if (r != kWaveDeclined) return r;
  }
  if (lds_fft_ok(p->N) && !g_force_generic) {
    const int CP = (C + 1) / 2, gpw = kThreads / lds_group_threads(p->N);
    const long long ntasks = (long long)B * CP * F;
    const size_t lds2 = ((size_t)gpw * lds_fwd_floats_per_group(p->N) + p->N) * sizeof(float);
    const bool alias = lds_fwd_floats_per_group(p->N) == 4 * p->N;
    const int st2 = alias ? allow_lds(k_fwd_lds<float, true>, lds2) : allow_lds(k_fwd_lds<float, false>, lds2);
    if (st2) return st2;
    const dim3 grid((unsigned)((ntasks + gpw - 1) / gpw));
    if (alias)
      hipLaunchKernelGGL((k_fwd_lds<float, true>), grid, dim3(kThreads), lds2, s, x, X, prev_block, p->d_coef, p->d_ctab, Kin, F,
                         C, CP, p->N, ntasks);
    else
      hipLaunchKernelGGL((k_fwd_lds<float, false>), grid, dim3(kThreads), lds2, s, x, X, prev_block, p->d_coef, p->d_ctab, Kin, F,
                         C, CP, p->N, ntasks);
    AC_HIP_CHECK(hipGetLastError());
    return AC_OK;
  }
  const size_t lds = (size_t)p->N * sizeof(float);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the generic kernel", p->N);
  hipLaunchKernelGGL((k_fwd_generic<float, float>), dim3((unsigned)nwg), dim3(kThreads), lds, s, x, X, prev_block, p->d_coef,
                     p->d_ctab, Kin, F, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_inv_generic(const ac_mdct_plan* p, const float* X, float* x, const float* tail_in, float* tail_out,
                       int B, int Kp, int nblk, int C, hipStream_t s) {
  const int per_sig = nblk + (tail_out ? 1 : 0);
  const long long nwg = (long long)B * C * per_sig;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  if (lds_wave_ok(p->N, true, C, true) && !g_force_generic) {
    const int r = launch_inv_wave<float>(p, X, x, tail_in, tail_out, B, Kp, nblk, C, s);
    if (r != kWaveDeclined) return r;
  }
  if (lds_fft_ok(p->N) && !g_force_generic) {
    const int seg = 8, CP = (C + 1) / 2, gpw = kThreads / lds_group_threads(p->N);
    const int nseg = (per_sig + seg - 1) / seg;
    const long long ntasks = (long long)B * CP * nseg;
    const size_t lds2 = ((size_t)gpw * 7 * p->N + p->N) * sizeof(float);
    const int st2 = allow_lds(k_inv_lds<float>, lds2);
    if (st2) return st2;
    hipLaunchKernelGGL(k_inv_lds<float>, dim3((unsigned)((ntasks + gpw - 1) / gpw)), dim3(kThreads), lds2, s, X, x, tail_in,
                       tail_out, p->d_coef, p->d_ctab, Kp, nblk, seg, nseg, C, CP, p->N, ntasks);
    AC_HIP_CHECK(hipGetLastError());
    return AC_OK;
  }
  const size_t lds = 2 * (size_t)p->N * sizeof(float);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the generic kernel", p->N);
  hipLaunchKernelGGL((k_inv_generic<float, float>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, x, tail_in, tail_out,
                     p->d_coef, p->d_ctab, Kp, nblk, per_sig, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_tonality_generic(const ac_psy_plan* p, const float* X, float* t, int B, int F, int C, hipStream_t s) {
  const long long nwg = (long long)B * F * C;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  hipLaunchKernelGGL((k_tonality_generic<float, float>), dim3((unsigned)nwg), dim3(kThreads), 0, s, X, t, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_threshold_generic(const ac_psy_plan* p, const float* X, const float* t, float drown, float* thr, int B,
                             int F, int C, hipStream_t s) {
  const long long nwg = (long long)B * F * C;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  const size_t lds = ((size_t)p->N + 2 * (size_t)p->M) * sizeof(float);
  AC_REQUIRE(lds <= 64 * 1024, "filter_bands_n = %d / bark_bands_n = %d too large for the generic kernel", p->N,
             p->M);
  hipLaunchKernelGGL((k_threshold_generic<float, float>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, t, thr, drown,
                     (float)p->alpha, p->d_wb_ptr, p->d_wb_idx, p->d_wb_val, p->d_wi_ptr, p->d_wi_idx, p->d_wi_val,
                     p->d_S, p->d_quiet, p->d_beta, C, p->N, p->M);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

template <typename TIO>
static int launch_tonality_bwd_T(const ac_psy_plan* p, const TIO* X, const TIO* gt, TIO* gX, int accumulate, int B, int F, int C,
                                 hipStream_t s) {
  const long long nwg = (long long)B * F * C;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  hipLaunchKernelGGL((k_tonality_bwd_generic<TIO>), dim3((unsigned)nwg), dim3(kThreads), 0, s, X, gt, gX, accumulate, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
int launch_tonality_bwd_generic(const ac_psy_plan* p, const float* X, const float* gt, float* gX, int accumulate, int B,
                                int F, int C, hipStream_t s) {
  return launch_tonality_bwd_T<float>(p, X, gt, gX, accumulate, B, F, C, s);
}

// TC tables of the plan: float (float32 / bfloat16 tensors) or the float64 set
template <typename TIO, typename TC>
static int launch_threshold_bwd_T(const ac_psy_plan* p, const TIO* X, const TIO* t, TC drown, const TIO* gthr, TIO* gX, TIO* gt,
                                  const TC* wb_val, const TC* wi_val, const TC* wf_val, const TC* vb_val, const TC* S, const TC* quiet,
                                  const TC* beta, int B, int F, int C, hipStream_t s) {
  const long long nwg = (long long)B * F * C;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  size_t lds = (2 * (size_t)p->N + 10 * (size_t)p->M) * sizeof(TC);
  AC_REQUIRE(lds <= 64 * 1024, "filter_bands_n = %d / bark_bands_n = %d too large for the backward kernel", p->N, p->M);
  const size_t with_s = lds + (size_t)p->M * p->M * sizeof(TC);
  const int s_in_lds = with_s <= 64 * 1024;
  if (s_in_lds) lds = with_s;
  hipLaunchKernelGGL((k_threshold_bwd_generic<TIO, TC>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, t, gthr, gX, gt, drown,
                     (TC)p->alpha, p->d_wb_ptr, p->d_wb_idx, wb_val, p->d_wi_ptr, p->d_wi_idx, wi_val, p->d_wf_ptr, p->d_wf_idx,
                     wf_val, p->d_vb_ptr, p->d_vb_idx, vb_val, S, quiet, beta, C, p->N, p->M, s_in_lds);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
int launch_threshold_bwd_generic(const ac_psy_plan* p, const float* X, const float* t, float drown, const float* gthr,
                                 float* gX, float* gt, int B, int F, int C, hipStream_t s) {
  return launch_threshold_bwd_T<float, float>(p, X, t, drown, gthr, gX, gt, p->d_wb_val, p->d_wi_val, p->d_wf_val, p->d_vb_val, p->d_S,
                                              p->d_quiet, p->d_beta, B, F, C, s);
}
// compute_dtype float64 (everything in double) / bfloat16 (bfloat16 tensors, float32 arithmetic and tables)
int launch_tonality_bwd_typed(const ac_psy_plan* p, const void* X, const void* gt, void* gX, int dtype, int B, int F, int C, hipStream_t s) {
  if (dtype == AC_F64) return launch_tonality_bwd_T<double>(p, (const double*)X, (const double*)gt, (double*)gX, 0, B, F, C, s);
  return launch_tonality_bwd_T<bf16_t>(p, (const bf16_t*)X, (const bf16_t*)gt, (bf16_t*)gX, 0, B, F, C, s);
}
int launch_threshold_bwd_typed(const ac_psy_plan* p, const void* X, const void* t, double drown, const void* gthr, void* gX, void* gt,
                               int dtype, int B, int F, int C, hipStream_t s) {
  if (dtype == AC_F64)
    return launch_threshold_bwd_T<double, double>(p, (const double*)X, (const double*)t, drown, (const double*)gthr, (double*)gX, (double*)gt,
                                                  p->d_wb_val64, p->d_wi_val64, p->d_wf_val64, p->d_vb_val64, p->d_S64, p->d_quiet64,
                                                  p->d_beta64, B, F, C, s);
  return launch_threshold_bwd_T<bf16_t, float>(p, (const bf16_t*)X, (const bf16_t*)t, (float)drown, (const bf16_t*)gthr, (bf16_t*)gX,
                                               (bf16_t*)gt, p->d_wb_val, p->d_wi_val, p->d_wf_val, p->d_vb_val, p->d_S, p->d_quiet,
                                               p->d_beta, B, F, C, s);
}

int launch_db(const float* a, float* out, size_t n, int norm, hipStream_t s) {
  if (n == 0) return AC_OK;
  // 16-byte vectors when both pointers allow it, else element by element
  const bool al = (((uintptr_t)a | (uintptr_t)out) & 15) == 0;
  const size_t n4 = al ? n / 4 : 0;
  const size_t nblk = (n4 + (n - 4 * n4) + 255) / 256;   // one thread per 16-byte vector, then one per tail element
  if (nblk > 2147483647ull) {
    set_error("tensor too large for one launch (%zu elements)", n);
    return AC_EINVAL;
  }
  hipLaunchKernelGGL(k_db, dim3((unsigned)nblk), dim3(256), 0, s, a, out, n, n4, norm);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_db_bwd(const float* a, const float* g, float* ga, size_t n, int norm, hipStream_t s) {
  if (n == 0) return AC_OK;
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 16384);
  hipLaunchKernelGGL(k_db_bwd, dim3(grid), dim3(256), 0, s, a, g, ga, n, norm);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_add_noise(const float* X, const float* thr, float* out, size_t n, uint64_t seed, hipStream_t s) {
  if (n == 0) return AC_OK;
  const bool al = (((uintptr_t)X | (uintptr_t)thr | (uintptr_t)out) & 15) == 0;
  const size_t n4 = al ? n / 4 : 0;
  const size_t nblk = (n4 + (n - 4 * n4) + 255) / 256;
  if (nblk > 2147483647ull) {
    set_error("tensor too large for one launch (%zu elements)", n);
    return AC_EINVAL;
  }
  hipLaunchKernelGGL(k_add_noise, dim3((unsigned)nblk), dim3(256), 0, s, X, thr, out, n, n4, seed);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

// ---- compute_dtype variants: double = O(N^2) kernels in fp64 with fp64 tables; bfloat16 = bfloat16 tensors, float32
// arithmetic, the LDS-FFT middle tier where filters_n / 2 is 5-smooth (O(N^2) kernels otherwise) ----
int launch_fwd_f64(const ac_mdct_plan* p, const double* x, double* X, int B, int Kin, int F, int C, hipStream_t s) {
  const long long nwg = (long long)B * C * F;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  const size_t lds = (size_t)p->N * sizeof(double);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the float64 kernel", p->N);
  hipLaunchKernelGGL((k_fwd_generic<double, double>), dim3((unsigned)nwg), dim3(kThreads), lds, s, x, X,
                     (const double*)nullptr, p->d_coef64, p->d_ctab64, Kin, F, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_inv_f64(const ac_mdct_plan* p, const double* X, double* x, int B, int Kp, int nblk, int C, hipStream_t s) {
  const long long nwg = (long long)B * C * nblk;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  const size_t lds = 2 * (size_t)p->N * sizeof(double);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the float64 kernel", p->N);
  hipLaunchKernelGGL((k_inv_generic<double, double>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, x,
                     (const double*)nullptr, (double*)nullptr, p->d_coef64, p->d_ctab64, Kp, nblk, nblk, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_fwd_f64_stream(const ac_mdct_plan* p, const double* x, double* X, const double* prev_block, int B, int Kin, int F, int C,
                          hipStream_t s) {
  const long long nwg = (long long)B * C * F;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  const size_t lds = (size_t)p->N * sizeof(double);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the float64 kernel", p->N);
  hipLaunchKernelGGL((k_fwd_generic<double, double>), dim3((unsigned)nwg), dim3(kThreads), lds, s, x, X, prev_block, p->d_coef64,
                     p->d_ctab64, Kin, F, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
int launch_inv_f64_stream(const ac_mdct_plan* p, const double* X, double* x, const double* tail_in, double* tail_out, int B, int Kp,
                          int nblk, int C, hipStream_t s) {
  const int per_sig = nblk + (tail_out ? 1 : 0);
  const long long nwg = (long long)B * C * per_sig;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  const size_t lds = 2 * (size_t)p->N * sizeof(double);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the float64 kernel", p->N);
  hipLaunchKernelGGL((k_inv_generic<double, double>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, x, tail_in, tail_out,
                     p->d_coef64, p->d_ctab64, Kp, nblk, per_sig, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

// 2-byte tensors (bfloat16, float16), float32 arithmetic: the 8-byte wave kernels / the workgroup form of the LDS-FFT tier, else O(N^2)
template <typename T16>
static int launch_fwd_16(const ac_mdct_plan* p, const T16* x, T16* X, int B, int Kin, int F, int C, hipStream_t s) {
  const long long nwg = (long long)B * C * F;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  if (lds_wave_ok(p->N, false, C, false) && !g_force_generic) {
    const int r = launch_fwd_wave<T16>(p, x, X, (const T16*)nullptr, B, Kin, F, C, s);
    if (r != kWaveDeclined) return r;
  }
  if (lds_fft_ok(p->N) && !g_force_generic) {
    const int CP = (C + 1) / 2, gpw = kThreads / lds_group_threads(p->N);
    const long long ntasks = (long long)B * CP * F;
    const size_t lds2 = ((size_t)gpw * lds_fwd_floats_per_group(p->N) + p->N) * sizeof(float);
    const bool alias = lds_fwd_floats_per_group(p->N) == 4 * p->N;
    const int st2 = alias ? allow_lds(k_fwd_lds<T16, true>, lds2) : allow_lds(k_fwd_lds<T16, false>, lds2);
    if (st2) return st2;
    const dim3 grid((unsigned)((ntasks + gpw - 1) / gpw));
    if (alias)
      hipLaunchKernelGGL((k_fwd_lds<T16, true>), grid, dim3(kThreads), lds2, s, x, X, (const T16*)nullptr, p->d_coef,
                         p->d_ctab, Kin, F, C, CP, p->N, ntasks);
    else
      hipLaunchKernelGGL((k_fwd_lds<T16, false>), grid, dim3(kThreads), lds2, s, x, X, (const T16*)nullptr, p->d_coef,
                         p->d_ctab, Kin, F, C, CP, p->N, ntasks);
    AC_HIP_CHECK(hipGetLastError());
    return AC_OK;
  }
  const size_t lds = (size_t)p->N * sizeof(float);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the generic kernel", p->N);
  hipLaunchKernelGGL((k_fwd_generic<T16, float>), dim3((unsigned)nwg), dim3(kThreads), lds, s, x, X,
                     (const T16*)nullptr, p->d_coef, p->d_ctab, Kin, F, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

template <typename T16>
static int launch_inv_16(const ac_mdct_plan* p, const T16* X, T16* x, int B, int Kp, int nblk, int C, hipStream_t s) {
  const long long nwg = (long long)B * C * nblk;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  if (lds_wave_ok(p->N, true, C, false) && !g_force_generic) {
    const int r = launch_inv_wave<T16>(p, X, x, nullptr, nullptr, B, Kp, nblk, C, s);
    if (r != kWaveDeclined) return r;
  }
  if (lds_fft_ok(p->N) && !g_force_generic) {
    const int seg = 8, CP = (C + 1) / 2, gpw = kThreads / lds_group_threads(p->N);
    const int nseg = (nblk + seg - 1) / seg;
    const long long ntasks = (long long)B * CP * nseg;
    const size_t lds2 = ((size_t)gpw * 7 * p->N + p->N) * sizeof(float);
    const int st2 = allow_lds(k_inv_lds<T16>, lds2);
    if (st2) return st2;
    hipLaunchKernelGGL(k_inv_lds<T16>, dim3((unsigned)((ntasks + gpw - 1) / gpw)), dim3(kThreads), lds2, s, X, x,
                       (const float*)nullptr, (float*)nullptr, p->d_coef, p->d_ctab, Kp, nblk, seg, nseg, C, CP, p->N,
                       ntasks);
    AC_HIP_CHECK(hipGetLastError());
    return AC_OK;
  }
  const size_t lds = 2 * (size_t)p->N * sizeof(float);
  AC_REQUIRE(lds <= 64 * 1024, "filters_n = %d too large for the generic kernel", p->N);
  hipLaunchKernelGGL((k_inv_generic<T16, float>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, x,
                     (const float*)nullptr, (float*)nullptr, p->d_coef, p->d_ctab, Kp, nblk, nblk, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

int launch_fwd_bf16(const ac_mdct_plan* p, const bf16_t* x, bf16_t* X, int B, int Kin, int F, int C, hipStream_t s) {
  return launch_fwd_16<bf16_t>(p, x, X, B, Kin, F, C, s);
}
int launch_inv_bf16(const ac_mdct_plan* p, const bf16_t* X, bf16_t* x, int B, int Kp, int nblk, int C, hipStream_t s) {
  return launch_inv_16<bf16_t>(p, X, x, B, Kp, nblk, C, s);
}
int launch_fwd_f16(const ac_mdct_plan* p, const f16_t* x, f16_t* X, int B, int Kin, int F, int C, hipStream_t s) {
  return launch_fwd_16<f16_t>(p, x, X, B, Kin, F, C, s);
}
int launch_inv_f16(const ac_mdct_plan* p, const f16_t* X, f16_t* x, int B, int Kp, int nblk, int C, hipStream_t s) {
  return launch_inv_16<f16_t>(p, X, x, B, Kp, nblk, C, s);
}

template <typename TIO>
static int launch_tonality_T(const ac_psy_plan* p, const TIO* X, TIO* t, int B, int F, int C, hipStream_t s) {
  const long long nwg = (long long)B * F * C;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  hipLaunchKernelGGL((k_tonality_generic<TIO>), dim3((unsigned)nwg), dim3(kThreads), 0, s, X, t, C, p->N);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
int launch_tonality_f64(const ac_psy_plan* p, const double* X, double* t, int B, int F, int C, hipStream_t s) {
  return launch_tonality_T<double>(p, X, t, B, F, C, s);
}
int launch_tonality_bf16(const ac_psy_plan* p, const bf16_t* X, bf16_t* t, int B, int F, int C, hipStream_t s) {
  return launch_tonality_T<bf16_t>(p, X, t, B, F, C, s);
}

int launch_threshold_f64(const ac_psy_plan* p, const double* X, const double* t, double drown, double* thr, int B, int F,
                         int C, hipStream_t s) {
  const long long nwg = (long long)B * F * C;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  const size_t lds = ((size_t)p->N + 2 * (size_t)p->M) * sizeof(double);
  AC_REQUIRE(lds <= 64 * 1024, "filter_bands_n = %d / bark_bands_n = %d too large for the float64 kernel", p->N, p->M);
  hipLaunchKernelGGL((k_threshold_generic<double, double>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, t, thr, drown,
                     p->alpha, p->d_wb_ptr, p->d_wb_idx, p->d_wb_val64, p->d_wi_ptr, p->d_wi_idx, p->d_wi_val64,
                     p->d_S64, p->d_quiet64, p->d_beta64, C, p->N, p->M);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
int launch_threshold_bf16(const ac_psy_plan* p, const bf16_t* X, const bf16_t* t, float drown, bf16_t* thr, int B, int F,
                          int C, hipStream_t s) {
  const long long nwg = (long long)B * F * C;
  const int st = check_grid(nwg);
  if (st) return st < 0 ? st : AC_OK;
  const size_t lds = ((size_t)p->N + 2 * (size_t)p->M) * sizeof(float);
  AC_REQUIRE(lds <= 64 * 1024, "filter_bands_n = %d / bark_bands_n = %d too large for the generic kernel", p->N, p->M);
  hipLaunchKernelGGL((k_threshold_generic<bf16_t, float>), dim3((unsigned)nwg), dim3(kThreads), lds, s, X, t, thr, drown,
                     (float)p->alpha, p->d_wb_ptr, p->d_wb_idx, p->d_wb_val, p->d_wi_ptr, p->d_wi_idx, p->d_wi_val,
                     p->d_S, p->d_quiet, p->d_beta, C, p->N, p->M);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}

template <typename TIO>
static int launch_db_T(const TIO* a, TIO* out, size_t n, int norm, hipStream_t s) {
  if (n == 0) return AC_OK;
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL((k_db_typed<TIO>), dim3(grid), dim3(256), 0, s, a, out, n, norm);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
template <typename TIO>
static int launch_add_noise_T(const TIO* X, const TIO* thr, TIO* out, size_t n, uint64_t seed, hipStream_t s) {
  if (n == 0) return AC_OK;
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL((k_add_noise_typed<TIO>), dim3(grid), dim3(256), 0, s, X, thr, out, n, seed);
  AC_HIP_CHECK(hipGetLastError());
  return AC_OK;
}
int launch_db_typed(const void* a, void* out, size_t n, int norm, int dtype, hipStream_t s) {
  if (dtype == AC_F64) return launch_db_T(static_cast<const double*>(a), static_cast<double*>(out), n, norm, s);
  return launch_db_T(static_cast<const bf16_t*>(a), static_cast<bf16_t*>(out), n, norm, s);
}
int launch_add_noise_typed(const void* X, const void* thr, void* out, size_t n, uint64_t seed, int dtype, hipStream_t s) {
  if (dtype == AC_F64)
    return launch_add_noise_T(static_cast<const double*>(X), static_cast<const double*>(thr), static_cast<double*>(out), n,
                              seed, s);
  return launch_add_noise_T(static_cast<const bf16_t*>(X), static_cast<const bf16_t*>(thr), static_cast<bf16_t*>(out), n, seed,
                            s);
}

#endif   // AC_WAVE_ROWS_TU

}  // namespace ac
