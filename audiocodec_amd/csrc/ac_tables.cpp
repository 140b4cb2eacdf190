// Host-side constant builders.  Everything here runs once per plan, in fp64, on the CPU.
#include "ac_tables.h"

#include <algorithm>
#include <cmath>

#include "../../include/audiocodec_amd.h"

namespace ac {

static const double kPi = 3.14159265358979323846;

void window_samples(int N, int window, std::vector<double>& w) {
  const int L = N + N / 2;
  w.resize(L);
  for (int n = 0; n < L; ++n) {
    const double p = n + 0.5;
    if (window == AC_WINDOW_SINE) {
      w[n] = std::sin(kPi / (2 * N) * p);                                   // :199-203
    } else if (window == AC_WINDOW_VORBIS) {
      const double s = std::sin(kPi / (2.0 * N) * p);                        // :204-208
      w[n] = std::sin(kPi / 2.0 * s * s);
    } else {
      w[n] = 1.0;                                                            // :209-211
    }
  }
}

void fold_coefficients(int N, int window, FoldCoef& c) {
  const int h = N / 2;
  std::vector<double> w;
  window_samples(N, window, w);
  for (auto* v : {&c.a1, &c.a2, &c.a3, &c.a4, &c.s1, &c.s2, &c.s3, &c.s4}) v->resize(h);
  for (int j = 0; j < h; ++j) {
    // 2x2 block of F coupling rows {j, N-1-j} with columns {h-1-j, h+j}  (:214-229)
    const double p = w[j];                                                   // upper-left
    const double q = w[N + j];                                               // upper-right
    const double r = w[N - 1 - j];                                           // lower-left
    const double s = -(1.0 - w[N + j] * w[N - 1 - j]) / w[j];                // lower-right (:218-226)
    const double det = p * s - q * r;
    c.a1[j] = q;
    c.a2[j] = s;
    c.a3[j] = w[h - 1 - j];
    c.a4[j] = w[h + j];
    c.s1[j] = s / det;
    c.s2[j] = -r / det;
    c.s3[j] = -q / det;
    c.s4[j] = p / det;
  }
}

static inline double bark2freq(double z) { return 600.0 * std::sinh(z / 6.0); }   // :337-339
static inline double freq2bark(double f) { return 6.0 * std::asinh(f / 600.0); }  // :333-335

void psy_tables(int N, int M, double sample_rate, double alpha, PsyTables& t) {
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
  const double dB_MAX = 120.0;
  t.max_frequency = sample_rate / 2.0;                                       // :61
  t.max_bark = freq2bark(t.max_frequency);                                   // :62
  t.bark_band_width = t.max_bark / M;                                        // :63
  const double bw = t.bark_band_width;

  // _bark_freq_mapping  :257-299
  t.W.assign((size_t)N * M, 0.0);
  t.W_inv.assign((size_t)M * N, 0.0);
  const double fbw = t.max_frequency / N;                                    // :282
  for (int j = 0; j < M; ++j) {
    const double bark_low = bw * j;                                          // :285
    const double lo = bark2freq(bark_low);
    const double hi = bark2freq(bark_low + bw);
    for (int f = 0; f < N; ++f) {
      const double f_lo = fbw * f;                                           // :289
      const double f_hi = f_lo + fbw;
      const double lo_c = std::min(std::max(lo, f_lo), f_hi);
      const double hi_c = std::min(std::max(hi, f_lo), f_hi);
      const double overlap = hi_c - lo_c;
      t.W[(size_t)f * M + j] = overlap / fbw;                                // :294
      t.W_inv[(size_t)j * N + f] = overlap / (hi - lo);
    }
  }

  // _quiet_threshold_intensity_in_bark  :232-255
  t.quiet.resize(M);
  for (int j = 0; j < M; ++j) {
    const double mid = bw * j + bw / 2.0;
    const double kHz = bark2freq(mid) / 1000.0;
    double dB = 3.64 * std::pow(kHz, -0.8) - 6.5 * std::exp(-0.6 * std::pow(kHz - 3.3, 2.0)) +
                1e-3 * std::pow(kHz, 4.0);
    dB = std::min(std::max(dB, t.dB_MIN), dB_MAX);
    t.quiet[j] = std::pow(10.0, (dB - dB_MAX) / 10.0);
  }

  // _spreading_matrix_in_bark  :212-230   (z = linspace(-max_bark, max_bark, 2M))
  std::vector<double> g(2 * (size_t)M);
  for (int i = 0; i < 2 * M; ++i) {
    double z;
    if (2 * M == 1) {
      z = -t.max_bark;
    } else {
      const double step = (t.max_bark - (-t.max_bark)) / (2 * M - 1);
      z = (i == 2 * M - 1) ? t.max_bark : (-t.max_bark + step * i);
    }
    const double f = 15.81 + 7.5 * (z + 0.474) - 17.5 * std::sqrt(1.0 + std::pow(z + 0.474, 2.0));
    g[i] = std::pow(10.0, alpha * f / 10.0);                                 // :223
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
