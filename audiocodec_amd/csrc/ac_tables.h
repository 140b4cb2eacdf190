// Host-side constant builders (fp64 precompute, then down-cast), shared by the C ABI and the plans.
// Each function cites the reference code it replaces (file:line into korneelvdbroek/audiocodec).
#pragma once
#include <cstdint>
#include <vector>

namespace ac {

// `pre` = the arithmetic type the constants are computed in, the reference's precompute_dtype (mdctransformer.py:13-14,
// 31-35; psychoacoustic.py:14-15): AC_F64 (= 1, the default) or AC_F32 (= 0); values are returned as doubles either way.
constexpr int kPreF64 = 1;

// mdctransformer.py:199-211 -- window samples w[n], n = 0 .. 3N/2-1, evaluated at n + 1/2.
void window_samples(int N, int window, std::vector<double>& w, int pre = kPreF64);

// Non-zeros of F and F^-1 (mdctransformer.py:155-229), N/2 entries each.
struct FoldCoef {
  std::vector<double> a1, a2, a3, a4;  // analysis
  std::vector<double> s1, s2, s3, s4;  // synthesis (2x2 blocks of F inverted in closed form)
};
void fold_coefficients(int N, int window, FoldCoef& c, int pre = kPreF64);

// psychoacoustic.py:52-69, 212-299 -- all constants of PsychoacousticModel.__init__.
struct PsyTables {
  int N = 0, M = 0;
  double sample_rate = 0, alpha = 0;
  double max_frequency = 0, max_bark = 0, bark_band_width = 0, dB_MIN = 0;
  std::vector<double> W;      // [N, M]   psychoacoustic.py:257-299
  std::vector<double> W_inv;  // [M, N]
  std::vector<double> S;      // [M, M]   psychoacoustic.py:212-230
  std::vector<double> quiet;  // [M]      psychoacoustic.py:232-255
  std::vector<float> beta;    // [M]      linspace(0, max_bark, M) in float32, psychoacoustic.py:187-189
  std::vector<double> g;      // [2M]     spreading prototype, S[i][j] = g[M - i + j]  (psychoacoustic.py:223-228)
};
void psy_tables(int N, int M, double sample_rate, double alpha, PsyTables& t, int pre = kPreF64);

// Compressed forms used by the kernels.
struct SparseRows {           // CSR: row r has entries ptr[r] .. ptr[r+1]-1
  std::vector<int32_t> ptr;
  std::vector<int32_t> idx;
  std::vector<float> val;
  int max_row = 0;
};
// W as "by band" lists (rows = Bark band j, entries = (bin f, W[f,j]))
void w_by_band(const PsyTables& t, SparseRows& out);
// W_inv as "by bin" lists (rows = bin f, entries = (band j, W_inv[j,f]))
void winv_by_bin(const PsyTables& t, SparseRows& out);
// the transposed walks used by the backward pass: W "by bin" (rows = bin f, entries = (band j, W[f,j])) and
// W_inv "by band" (rows = band j, entries = (bin f, W_inv[j,f]))
void w_by_bin(const PsyTables& t, SparseRows& out);
void winv_by_band(const PsyTables& t, SparseRows& out);

}  // namespace ac
