// Host-side constant builders.  Everything here runs once per plan on the CPU, in the plan's precompute dtype (fp64 by
// default, as the reference's precompute_dtype).
#include "ac_tables.h"

#include <algorithm>
#include <cmath>

#include "../../include/audiocodec_amd.h"

namespace ac {

static const double kPi = 3.14159265358979323846;

// The builders are templates over the arithmetic type T of the reference's `precompute_dtype` (mdctransformer.py:13-14,
// 31-35; psychoacoustic.py:14-15, 61-69): double (the default) or float.  TensorFlow converts a Python scalar that meets a
// tensor to the tensor's dtype, so every literal below is rounded to T first and every operation runs in T, in the
// reference's order; results are stored as doubles (exact for both).
template <typename T>
static void window_samples_t(int N, int window, std::vector<double>& w) {
  const int L = N + N / 2;
  w.resize(L);
  for (int n = 0; n < L; ++n) {
    const T p = (T)n + (T)0.5;                                               // tf.range(0.5, ...): k + 1/2, exact
    if (window == AC_WINDOW_SINE) {
      w[n] = (double)std::sin((T)(kPi / (2 * N)) * p);                       // :199-203
    } else if (window == AC_WINDOW_VORBIS) {
      const T s = std::sin((T)(kPi / (2.0 * N)) * p);                        // :204-208
      const T s2 = s * s;
      w[n] = (double)std::sin((T)(kPi / 2.0) * s2);
    } else {
      w[n] = 1.0;                                                            // :209-211
    }
  }
}

void window_samples(int N, int window, std::vector<double>& w, int pre) {
  if (pre == AC_F32) window_samples_t<float>(N, window, w);
  else window_samples_t<double>(N, window, w);
}

template <typename T>
static void fold_coefficients_t(int N, int window, FoldCoef& c) {
  const int h = N / 2;
  std::vector<double> wd;
  window_samples_t<T>(N, window, wd);
  for (auto* v : {&c.a1, &c.a2, &c.a3, &c.a4, &c.s1, &c.s2, &c.s3, &c.s4}) v->resize(h);
  for (int j = 0; j < h; ++j) {
    // 2x2 block of F coupling rows {j, N-1-j} with columns {h-1-j, h+j}  (:214-229)
    const T p = (T)wd[j];                                                    // upper-left
    const T q = (T)wd[N + j];                                                // upper-right
    const T r = (T)wd[N - 1 - j];                                            // lower-left
    const T s = -(((T)1 - q * r) / p);                                       // lower-right (:218-226): cancels for small j
    // (in float this is exactly 0 for j = 0 at N = 64, as in the reference's float32 precompute)
    // F^-1 (tf.linalg.inv, :185: an LU inverse of the full matrix there): the 2x2 block inverted in closed form, in T -- the
    // dense H / H_inv of the reference's source run over numpy in float32 are met to 3e-7 / 5e-7 absolute (tests/test_host.py)
    const T det = p * s - q * r;
    c.a1[j] = (double)q;
    c.a2[j] = (double)s;
    c.a3[j] = wd[h - 1 - j];
    c.a4[j] = wd[h + j];
    c.s1[j] = (double)(s / det);
    c.s2[j] = (double)(-r / det);
    c.s3[j] = (double)(-q / det);
    c.s4[j] = (double)(p / det);
  }
}

void fold_coefficients(int N, int window, FoldCoef& c, int pre) {
  if (pre == AC_F32) fold_coefficients_t<float>(N, window, c);
  else fold_coefficients_t<double>(N, window, c);
}

template <typename T> static inline T bark2freq_t(T z) { return (T)600 * std::sinh(z / (T)6); }   // :337-339
template <typename T> static inline T freq2bark_t(T f) { return (T)6 * std::asinh(f / (T)600); }  // :333-335

template <typename T>
static void psy_tables_t(int N, int M, double sample_rate, double alpha, PsyTables& t) {
  t.N = N;
  t.M = M;
  t.sample_rate = sample_rate;
  t.alpha = alpha;
  // _dB_MIN = amplitude_to_dB(_INTENSITY_EPS) evaluated in the compute dtype (float32)   :56-58, :83
  {
    const float eps = 1e-14f;
    const float a2 = eps * eps;
    t.dB_MIN = 10.f * std::log(std::max(eps, a2)) / std::log(10.f) + 120.f;
  }
  const T dB_MAX = (T)120, dB_MIN = (T)t.dB_MIN;
  const T max_frequency = (T)sample_rate / (T)2;                             // :61
  const T max_bark = freq2bark_t<T>(max_frequency);                          // :62
  const T bw = max_bark / (T)M;                                              // :63
  t.max_frequency = (double)max_frequency;
  t.max_bark = (double)max_bark;
  t.bark_band_width = (double)bw;

  // _bark_freq_mapping  :257-299
  t.W.assign((size_t)N * M, 0.0);
  t.W_inv.assign((size_t)M * N, 0.0);
  const T fbw = max_frequency / (T)N;                                        // :282
  for (int j = 0; j < M; ++j) {
    const T bark_low = bw * (T)j;                                            // :285
    const T lo = bark2freq_t<T>(bark_low);
    const T hi = bark2freq_t<T>(bark_low + bw);
    for (int f = 0; f < N; ++f) {
      const T f_lo = fbw * (T)f;                                             // :289
      const T f_hi = f_lo + fbw;
      const T lo_c = std::min(std::max(lo, f_lo), f_hi);
      const T hi_c = std::min(std::max(hi, f_lo), f_hi);
      const T overlap = hi_c - lo_c;
      t.W[(size_t)f * M + j] = (double)(overlap / fbw);                      // :294
      t.W_inv[(size_t)j * N + f] = (double)(overlap / (hi - lo));
    }
  }

  // _quiet_threshold_intensity_in_bark  :232-255
  t.quiet.resize(M);
  for (int j = 0; j < M; ++j) {
    const T mid = bw * (T)j + bw / (T)2;
    const T kHz = bark2freq_t<T>(mid) / (T)1000;
    T dB = (T)3.64 * std::pow(kHz, (T)-0.8) - (T)6.5 * std::exp((T)-0.6 * std::pow(kHz - (T)3.3, (T)2)) +
           (T)1e-3 * std::pow(kHz, (T)4);
    dB = std::min(std::max(dB, dB_MIN), dB_MAX);
    t.quiet[j] = (double)std::pow((T)10, (dB - dB_MAX) / (T)10);
  }

  // _spreading_matrix_in_bark  :212-230   (z = linspace(-max_bark, max_bark, 2M), in max_bark's dtype)
  std::vector<double> g(2 * (size_t)M);
  for (int i = 0; i < 2 * M; ++i) {
    T z;
    if (2 * M == 1) {
      z = -max_bark;
    } else {
      const T step = (max_bark - (-max_bark)) / (T)(2 * M - 1);
      z = (i == 2 * M - 1) ? max_bark : (-max_bark + step * (T)i);
    }
    const T f = (T)15.81 + (T)7.5 * (z + (T)0.474) - (T)17.5 * std::sqrt((T)1 + std::pow(z + (T)0.474, (T)2));
    g[i] = (double)std::pow((T)10, (T)alpha * f / (T)10);                    // :223
  }
  t.g = g;
  t.S.resize((size_t)M * M);
  for (int row = 0; row < M; ++row)
    for (int col = 0; col < M; ++col) t.S[(size_t)row * M + col] = g[M - row + col];   // :227-228

  // offset grid: tf.linspace(0, float32(max_bark), M) in the compute dtype  :187-189
  t.beta.resize(M);
  {
    const float stop = (float)t.max_bark;
    const float step = (M > 1) ? stop / (float)(M - 1) : 0.f;
    for (int j = 0; j < M; ++j) t.beta[j] = step * (float)j;
    if (M > 1) t.beta[M - 1] = stop;
  }
}

void psy_tables(int N, int M, double sample_rate, double alpha, PsyTables& t, int pre) {
  if (pre == AC_F32) psy_tables_t<float>(N, M, sample_rate, alpha, t);
  else psy_tables_t<double>(N, M, sample_rate, alpha, t);
}

void w_by_band(const PsyTables& t, SparseRows& out) {
  out.ptr.assign(t.M + 1, 0);
  out.idx.clear();
  out.val.clear();
  out.max_row = 0;
  for (int j = 0; j < t.M; ++j) {
    for (int f = 0; f < t.N; ++f) {
      const float v = (float)t.W[(size_t)f * t.M + j];
      if (v != 0.f) {
        out.idx.push_back(f);
        out.val.push_back(v);
      }
    }
    out.ptr[j + 1] = (int32_t)out.idx.size();
    out.max_row = std::max(out.max_row, out.ptr[j + 1] - out.ptr[j]);
  }
}

void winv_by_bin(const PsyTables& t, SparseRows& out) {
  out.ptr.assign(t.N + 1, 0);
  out.idx.clear();
  out.val.clear();
  out.max_row = 0;
  for (int f = 0; f < t.N; ++f) {
    for (int j = 0; j < t.M; ++j) {
      const float v = (float)t.W_inv[(size_t)j * t.N + f];
      if (v != 0.f) {
        out.idx.push_back(j);
        out.val.push_back(v);
      }
    }
    out.ptr[f + 1] = (int32_t)out.idx.size();
    out.max_row = std::max(out.max_row, out.ptr[f + 1] - out.ptr[f]);
  }
}

void w_by_bin(const PsyTables& t, SparseRows& out) {
  out.ptr.assign(t.N + 1, 0);
  out.idx.clear();
  out.val.clear();
  out.max_row = 0;
  for (int f = 0; f < t.N; ++f) {
    for (int j = 0; j < t.M; ++j) {
      const float v = (float)t.W[(size_t)f * t.M + j];
      if (v != 0.f) {
        out.idx.push_back(j);
        out.val.push_back(v);
      }
    }
    out.ptr[f + 1] = (int32_t)out.idx.size();
    out.max_row = std::max(out.max_row, out.ptr[f + 1] - out.ptr[f]);
  }
}

void winv_by_band(const PsyTables& t, SparseRows& out) {
  out.ptr.assign(t.M + 1, 0);
  out.idx.clear();
  out.val.clear();
  out.max_row = 0;
  for (int j = 0; j < t.M; ++j) {
    for (int f = 0; f < t.N; ++f) {
      const float v = (float)t.W_inv[(size_t)j * t.N + f];
      if (v != 0.f) {
        out.idx.push_back(f);
        out.val.push_back(v);
      }
    }
    out.ptr[j + 1] = (int32_t)out.idx.size();
    out.max_row = std::max(out.max_row, out.ptr[j + 1] - out.ptr[j]);
  }
}

}  // namespace ac
