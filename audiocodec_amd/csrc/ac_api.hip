// C ABI of libaudiocodec_amd.so (see include/audiocodec_amd.h).  gfx950 only.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <algorithm>

#include "ac_internal.h"

namespace ac {

static thread_local std::string g_err;
int g_force_generic = 0;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

template <typename T>
static int upload(const std::vector<T>& h, T** d) {
  *d = nullptr;
  const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
  AC_HIP_CHECK(hipMalloc((void**)d, bytes));
  if (!h.empty()) AC_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return AC_OK;
}

static int check_device(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_error("no HIP device available");
    return AC_ENODEV;
  }
  if (device < 0 || device >= count) {
    set_error("device %d out of range (have %d)", device, count);
    return AC_EINVAL;
  }
  return AC_OK;
}

static bool valid_window(int w) { return w == AC_WINDOW_VORBIS || w == AC_WINDOW_SINE || w == AC_WINDOW_RECT; }

}  // namespace ac

using namespace ac;

extern "C" {

int ac_version(void) { return AC_VERSION; }
const char* ac_last_error(void) { return g_err.c_str(); }

int ac_set_force_generic(int on) {
  // a test hook, not a product switch: honoured only in processes started with AC_TESTING=1 (read once)
  static const bool testing = [] { const char* e = getenv("AC_TESTING"); return e && atoi(e) != 0; }();
  if (!testing) {
    set_error("ac_set_force_generic is a test hook: start the process with AC_TESTING=1 to use it");
    return AC_EUNSUPPORTED;
  }
  g_force_generic = on ? 1 : 0;
  return AC_OK;
}

int ac_testing_runs_image(int N, int M, double sample_rate, double alpha, int precompute, unsigned* image, int cap, int* layout) {
  AC_REQUIRE(N >= 2 && (N & 1) == 0 && M >= 1 && layout, "ac_testing_runs_image: bad arguments");
  PsyTables t;
  psy_tables(N, M, sample_rate, alpha, t, precompute);
  std::vector<uint32_t> w;
  RunsLayout L;
  if (!build_runs(t, &w, &L)) {
    set_error("tables without the run structure");
    return AC_EUNSUPPORTED;
  }
  const int v[17] = {L.words, L.lw, L.kb, L.n4, L.n16, L.n64, L.o4, L.o16, L.o64, L.oz, L.slot,
                     L.off_S, L.off_bc, L.off_bd, L.off_lst, L.off_bw, L.off_idx};
  for (int i = 0; i < 17; ++i) layout[i] = v[i];
  if (image)
    for (int i = 0; i < L.words && i < cap; ++i) image[i] = w[(size_t)i];
  return AC_OK;
}

// ---- host-only builders ------------------------------------------------------------------------

#define AC_REQUIRE_PRE(d) AC_REQUIRE((d) == AC_F32 || (d) == AC_F64, "precompute = %d is not AC_F32 or AC_F64", (d))

int ac_mdct_fold_coefficients_host(int N, int window, double* coef) {
  return ac_mdct_fold_coefficients_host_pre(N, window, AC_F64, coef);
}

int ac_mdct_fold_coefficients_host_pre(int N, int window, int precompute, double* coef) {
  AC_REQUIRE(N >= 2 && (N % 2) == 0, "number of filters used in mdct transformation needs to be even (got %d)", N);
  AC_REQUIRE(valid_window(window), "unknown window id %d", window);
  AC_REQUIRE_PRE(precompute);
  AC_REQUIRE(coef != nullptr, "coef is NULL");
  FoldCoef c;
  fold_coefficients(N, window, c, precompute);
  const int h = N / 2;
  const std::vector<double>* v[8] = {&c.a1, &c.a2, &c.a3, &c.a4, &c.s1, &c.s2, &c.s3, &c.s4};
  for (int i = 0; i < 8; ++i) std::memcpy(coef + (size_t)i * h, v[i]->data(), h * sizeof(double));
  return AC_OK;
}

int ac_mdct_dense_matrices_host(int N, int window, float* H, float* H_inv) {
  return ac_mdct_dense_matrices_host_pre(N, window, AC_F64, H, H_inv);
}

int ac_mdct_dense_matrices_host_pre(int N, int window, int precompute, float* H, float* H_inv) {
  AC_REQUIRE(N >= 2 && (N % 2) == 0, "number of filters used in mdct transformation needs to be even (got %d)", N);
  AC_REQUIRE(valid_window(window), "unknown window id %d", window);
  AC_REQUIRE_PRE(precompute);
  FoldCoef c;
  fold_coefficients(N, window, c, precompute);
  const int h = N / 2;
  const size_t NN = (size_t)N * N;
  if (H) {
    // H[n, r, k] = F[r, k] d_n[k]; n = 0 uses columns k >= h (current block), n = 1 columns k < h
    std::memset(H, 0, 2 * NN * sizeof(float));
    for (int j = 0; j < h; ++j) {
      H[0 * NN + (size_t)j * N + (h + j)] = (float)c.a1[j];                  // F[j, h+j]
      H[0 * NN + (size_t)(N - 1 - j) * N + (h + j)] = (float)c.a2[j];        // F[N-1-j, h+j]
      H[1 * NN + (size_t)(h - 1 - j) * N + j] = (float)c.a3[j];              // F[h-1-j, j]
      H[1 * NN + (size_t)(h + j) * N + j] = (float)c.a4[j];                  // F[h+j, j]
    }
  }
  if (H_inv) {
    // H_inv[n, r, k] = e_n[r] Finv[r, k]; n = 0 rows r < h, n = 1 rows r >= h
    std::memset(H_inv, 0, 2 * NN * sizeof(float));
    for (int j = 0; j < h; ++j) {
      H_inv[0 * NN + (size_t)(h - 1 - j) * N + j] = (float)c.s1[j];          // Finv[h-1-j, j]
      H_inv[0 * NN + (size_t)(h - 1 - j) * N + (N - 1 - j)] = (float)c.s3[j];
      H_inv[1 * NN + (size_t)(h + j) * N + j] = (float)c.s2[j];              // Finv[h+j, j]
      H_inv[1 * NN + (size_t)(h + j) * N + (N - 1 - j)] = (float)c.s4[j];
    }
  }
  return AC_OK;
}

int ac_psy_tables_host(int N, int M, double sample_rate, double alpha, float* W, float* W_inv, float* S,
                       float* quiet, double* scalars) {
  AC_REQUIRE(N >= 1 && M >= 1, "filter_bands_n (%d) and bark_bands_n (%d) must be positive", N, M);
  AC_REQUIRE(sample_rate > 0 && alpha > 0, "sample_rate and alpha must be positive");
  PsyTables t;
  psy_tables(N, M, sample_rate, alpha, t);
  if (W) for (size_t i = 0; i < t.W.size(); ++i) W[i] = (float)t.W[i];
  if (W_inv) for (size_t i = 0; i < t.W_inv.size(); ++i) W_inv[i] = (float)t.W_inv[i];
  if (S) for (size_t i = 0; i < t.S.size(); ++i) S[i] = (float)t.S[i];
  if (quiet) for (size_t i = 0; i < t.quiet.size(); ++i) quiet[i] = (float)t.quiet[i];
  if (scalars) {
    scalars[0] = t.max_frequency;
    scalars[1] = t.max_bark;
    scalars[2] = t.bark_band_width;
    scalars[3] = t.dB_MIN;
  }
  return AC_OK;
}

int ac_psy_tables_host_f64(int N, int M, double sample_rate, double alpha, double* W, double* W_inv, double* S,
                           double* quiet, double* scalars) {
  return ac_psy_tables_host_pre(N, M, sample_rate, alpha, AC_F64, W, W_inv, S, quiet, scalars);
}

int ac_psy_tables_host_pre(int N, int M, double sample_rate, double alpha, int precompute, double* W, double* W_inv,
                           double* S, double* quiet, double* scalars) {
  AC_REQUIRE(N >= 1 && M >= 1, "filter_bands_n (%d) and bark_bands_n (%d) must be positive", N, M);
  AC_REQUIRE(sample_rate > 0 && alpha > 0, "sample_rate and alpha must be positive");
  AC_REQUIRE_PRE(precompute);
  PsyTables t;
  psy_tables(N, M, sample_rate, alpha, t, precompute);
  if (W) std::copy(t.W.begin(), t.W.end(), W);
  if (W_inv) std::copy(t.W_inv.begin(), t.W_inv.end(), W_inv);
  if (S) std::copy(t.S.begin(), t.S.end(), S);
  if (quiet) std::copy(t.quiet.begin(), t.quiet.end(), quiet);
  if (scalars) {
    scalars[0] = t.max_frequency;
    scalars[1] = t.max_bark;
    scalars[2] = t.bark_band_width;
    scalars[3] = t.dB_MIN;
  }
  return AC_OK;
}

// ---- plans --------------------------------------------------------------------------------------

int ac_mdct_plan_create(int N, int window, int device, ac_mdct_plan** out) {
  return ac_mdct_plan_create_pre(N, window, AC_F64, device, out);
}

static int mdct_plan_build(int N, int window, int precompute, const FoldCoef& c, int adjoint, int device, ac_mdct_plan** out) {
  int st = check_device(device);
  if (st) return st;
  DeviceGuard guard(device);
  ac_mdct_plan* p = new (std::nothrow) ac_mdct_plan();
  if (!p) {
    set_error("out of host memory");
    return AC_ENOMEM;
  }
  p->N = N;
  p->window = window;
  p->device = device;
  p->pre = precompute;
  p->adjoint = adjoint;
  p->coef = c;
  if (hipDeviceGetAttribute(&p->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || p->cus <= 0)
    p->cus = 256;
  const int h = N / 2;
  std::vector<float> coef(8 * (size_t)h);
  const std::vector<double>* v[8] = {&c.a1, &c.a2, &c.a3, &c.a4, &c.s1, &c.s2, &c.s3, &c.s4};
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < h; ++j) coef[(size_t)i * h + j] = (float)(*v[i])[j];
  std::vector<float> ctab(8 * (size_t)N);
  for (size_t i = 0; i < ctab.size(); ++i) ctab[i] = (float)std::cos(3.14159265358979323846 * (double)i / (4.0 * N));
  std::vector<double> coef64(8 * (size_t)h), ctab64(8 * (size_t)N);
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < h; ++j) coef64[(size_t)i * h + j] = (*v[i])[j];
  for (size_t i = 0; i < ctab64.size(); ++i) ctab64[i] = std::cos(3.14159265358979323846 * (double)i / (4.0 * N));
  st = upload(coef, &p->d_coef);
  if (!st && N % 4 == 0) {
    // per pair of samples (2 i, 2 i + 1), i < N / 4: analysis (a1, a2)(2i), (a1, a2)(2i+1) | (a3, a4)(h-1-2i), (a3, a4)(h-2-2i);
    // synthesis (behind the N / 2 analysis entries) (s1, s2)(2i), (s1, s2)(2i+1) | (s3, s4)(2i), (s3, s4)(2i+1)
    std::vector<float> cv(8 * (size_t)h);
    for (int i = 0; i < h / 2; ++i) {
      float* f = &cv[8 * (size_t)i];
      float* g = &cv[4 * (size_t)h + 8 * (size_t)i];
      const int j = 2 * i, m = h - 1 - 2 * i;
      f[0] = (float)c.a1[j], f[1] = (float)c.a2[j], f[2] = (float)c.a1[j + 1], f[3] = (float)c.a2[j + 1];
      f[4] = (float)c.a3[m], f[5] = (float)c.a4[m], f[6] = (float)c.a3[m - 1], f[7] = (float)c.a4[m - 1];
      g[0] = (float)c.s1[j], g[1] = (float)c.s2[j], g[2] = (float)c.s1[j + 1], g[3] = (float)c.s2[j + 1];
      g[4] = (float)c.s3[j], g[5] = (float)c.s4[j], g[6] = (float)c.s3[j + 1], g[7] = (float)c.s4[j + 1];
    }
    st = upload(cv, &p->d_coefv);
  }
  if (!st) st = upload(ctab, &p->d_ctab);
  if (!st) st = upload(coef64, &p->d_coef64);
  if (!st) st = upload(ctab64, &p->d_ctab64);
  if (!st && fast_mdct_supported(N, c)) {
    st = fast_mdct_plan_init(p);
    if (!st) p->fast = 1;
  }
  if (st) {
    ac_mdct_plan_destroy(p);
    return st;
  }
  *out = p;
  return AC_OK;
}

int ac_mdct_plan_create_pre(int N, int window, int precompute, int device, ac_mdct_plan** out) {
  AC_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  AC_REQUIRE_PRE(precompute);
  AC_REQUIRE(N >= 2 && (N % 2) == 0, "number of filters used in mdct transformation needs to be even (got %d)", N);
  AC_REQUIRE(N <= 8192, "filters_n = %d not supported (max 8192)", N);
  AC_REQUIRE(valid_window(window), "unknown window id %d", window);
  FoldCoef c;
  fold_coefficients(N, window, c, precompute);
  return mdct_plan_build(N, window, precompute, c, 0, device, out);
}

// The transposed filter bank of `plan` as a plan of its own: with T = the analysis bank (x -> X) and S = the synthesis bank
// (X -> x) of `plan`,  ac_mdct_inverse(adjoint, g)[:, N:-N] = 4 N T^T g   and   ac_mdct_forward(adjoint, g)[:, 1:-1] = S^T g / (4 N).
// The DCT-IV is symmetric, so only the O(N) fold transposes: block m of T^T g takes the second half of frame m through
// (a1, a2) and the first half of frame m + 1 through (a3, a4) -- the synthesis form with s1' = reverse(a3), s2' = a1,
// s3' = reverse(a4), s4' = a2; likewise a1' = s2, a2' = s4, a3' = reverse(s1), a4' = reverse(s3).  For Princen-Bradley
// windows computed in float64 the adjoint equals the plan itself (F^-1 = F^T); for the rectangular window
// (mdctransformer.py:209-229) and float32-precomputed ones it does not, and the backward passes need it.
int ac_mdct_plan_adjoint(const ac_mdct_plan* plan, ac_mdct_plan** out) {
  AC_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  AC_REQUIRE(plan != nullptr, "plan is NULL");
  const FoldCoef& c = plan->coef;
  FoldCoef t;
  t.a1 = c.s2;
  t.a2 = c.s4;
  t.a3.assign(c.s1.rbegin(), c.s1.rend());
  t.a4.assign(c.s3.rbegin(), c.s3.rend());
  t.s1.assign(c.a3.rbegin(), c.a3.rend());
  t.s2 = c.a1;
  t.s3.assign(c.a4.rbegin(), c.a4.rend());
  t.s4 = c.a2;
  return mdct_plan_build(plan->N, plan->window, plan->pre, t, plan->adjoint ? 0 : 1, plan->device, out);
}

int ac_mdct_plan_destroy(ac_mdct_plan* p) {
  if (!p) return AC_OK;
  DeviceGuard guard(p->device);
  (void)hipFree(p->d_coef);
  (void)hipFree(p->d_coefv);
  (void)hipFree(p->d_ctab);
  (void)hipFree(p->d_coef64);
  (void)hipFree(p->d_ctab64);
  (void)hipFree(p->d_fast);
  delete p;
  return AC_OK;
}

static int default_spreading() {
  // The spreading product of a plan created without an explicit choice: the split-bfloat16 matrix-core form where the
  // wave-level kernels serve the plan (measured 2 % faster on the fused encode, thresholds within 1e-5 of the float32
  // form), the float32 form everywhere else.  AC_SPREAD (0 / 1 / 2) overrides: tuning hook, read once.
  static const int dflt = [] {
    const char* e = getenv("AC_SPREAD");
    const int v = e ? atoi(e) : AC_SPREAD_BF16X2_MFMA;
    return v >= 0 && v <= 2 ? v : AC_SPREAD_BF16X2_MFMA;
  }();
  return dflt;
}

int ac_psy_plan_create(int N, int M, double sample_rate, double alpha, int device, ac_psy_plan** out) {
  return ac_psy_plan_create_pre(N, M, sample_rate, alpha, device, -1, AC_F64, out);
}

int ac_psy_plan_create_ex(int N, int M, double sample_rate, double alpha, int device, int spreading, ac_psy_plan** out) {
  AC_REQUIRE(spreading >= AC_SPREAD_F32 && spreading <= AC_SPREAD_BF16X2_MFMA, "spreading = %d is not one of AC_SPREAD_*", spreading);
  return ac_psy_plan_create_pre(N, M, sample_rate, alpha, device, spreading, AC_F64, out);
}

static int psy_plan_build(int N, int M, double sample_rate, double alpha, int device, int spreading, int precompute,
                          ac_psy_plan** out);

int ac_psy_plan_create_pre(int N, int M, double sample_rate, double alpha, int device, int spreading, int precompute,
                           ac_psy_plan** out) {
  AC_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  AC_REQUIRE_PRE(precompute);
  AC_REQUIRE(spreading >= -1 && spreading <= AC_SPREAD_BF16X2_MFMA, "spreading = %d is not one of AC_SPREAD_* (or -1: default)", spreading);
  if (spreading >= 0) return psy_plan_build(N, M, sample_rate, alpha, device, spreading, precompute, out);
  const int dflt = default_spreading();
  int st = psy_plan_build(N, M, sample_rate, alpha, device, dflt, precompute, out);
  if (st == AC_EUNSUPPORTED && dflt != 0) st = psy_plan_build(N, M, sample_rate, alpha, device, 0, precompute, out);
  return st;
}

static int psy_plan_build(int N, int M, double sample_rate, double alpha, int device, int spreading, int precompute,
                          ac_psy_plan** out) {
  AC_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  AC_REQUIRE(spreading >= AC_SPREAD_F32 && spreading <= AC_SPREAD_BF16X2_MFMA, "spreading = %d is not one of AC_SPREAD_*", spreading);
  AC_REQUIRE(N >= 1 && M >= 1, "filter_bands_n (%d) and bark_bands_n (%d) must be positive", N, M);
  AC_REQUIRE(N <= 8192 && M <= 4096, "filter_bands_n = %d / bark_bands_n = %d not supported", N, M);
  AC_REQUIRE(sample_rate > 0 && alpha > 0, "sample_rate and alpha must be positive");
  int st = check_device(device);
  if (st) return st;
  DeviceGuard guard(device);
  ac_psy_plan* p = new (std::nothrow) ac_psy_plan();
  if (!p) {
    set_error("out of host memory");
    return AC_ENOMEM;
  }
  p->N = N;
  p->M = M;
  p->device = device;
  p->sample_rate = sample_rate;
  p->alpha = alpha;
  p->pre = precompute;
  if (hipDeviceGetAttribute(&p->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || p->cus <= 0) p->cus = 256;
  psy_tables(N, M, sample_rate, alpha, p->host, precompute);
  SparseRows wb, wi, wf, vb;
  w_by_band(p->host, wb);
  winv_by_bin(p->host, wi);
  w_by_bin(p->host, wf);
  winv_by_band(p->host, vb);
  p->wb_max = wb.max_row;
  p->wi_max = wi.max_row;
  std::vector<float> S(p->host.S.size()), quiet(M);
  for (size_t i = 0; i < S.size(); ++i) S[i] = (float)p->host.S[i];
  for (int j = 0; j < M; ++j) quiet[j] = (float)p->host.quiet[j];
  st = upload(wb.ptr, &p->d_wb_ptr);
  if (!st) st = upload(wb.idx, &p->d_wb_idx);
  if (!st) st = upload(wb.val, &p->d_wb_val);
  if (!st) st = upload(wi.ptr, &p->d_wi_ptr);
  if (!st) st = upload(wi.idx, &p->d_wi_idx);
  if (!st) st = upload(wi.val, &p->d_wi_val);
  if (!st) st = upload(wf.ptr, &p->d_wf_ptr);
  if (!st) st = upload(wf.idx, &p->d_wf_idx);
  if (!st) st = upload(wf.val, &p->d_wf_val);
  if (!st) st = upload(vb.ptr, &p->d_vb_ptr);
  if (!st) st = upload(vb.idx, &p->d_vb_idx);
  if (!st) st = upload(vb.val, &p->d_vb_val);
  if (!st) st = upload(S, &p->d_S);
  if (!st) st = upload(quiet, &p->d_quiet);
  if (!st) st = upload(p->host.beta, &p->d_beta);
  {
    // the same constants in float64 (AC_F64 entry points): W / W_inv entries in CSR order, S, quiet, and the offset
    // grid linspace(0, max_bark, M) evaluated in float64 (psychoacoustic.py:187-189 with compute_dtype = float64)
    std::vector<double> wbv(wb.idx.size()), wiv(wi.idx.size()), wfv(wf.idx.size()), vbv(vb.idx.size()), beta64(M);
    for (int j = 0; j < M; ++j)
      for (int e = wb.ptr[j]; e < wb.ptr[j + 1]; ++e) wbv[e] = p->host.W[(size_t)wb.idx[e] * M + j];
    for (int f = 0; f < N; ++f)
      for (int e = wi.ptr[f]; e < wi.ptr[f + 1]; ++e) wiv[e] = p->host.W_inv[(size_t)wi.idx[e] * N + f];
    for (int f = 0; f < N; ++f)   // the transposed walks (backward passes): W by bin, W_inv by band
      for (int e = wf.ptr[f]; e < wf.ptr[f + 1]; ++e) wfv[e] = p->host.W[(size_t)f * M + wf.idx[e]];
    for (int j = 0; j < M; ++j)
      for (int e = vb.ptr[j]; e < vb.ptr[j + 1]; ++e) vbv[e] = p->host.W_inv[(size_t)j * N + vb.idx[e]];
    const double stop = p->host.max_bark, step = (M > 1) ? stop / (double)(M - 1) : 0.0;
    for (int j = 0; j < M; ++j) beta64[j] = step * (double)j;
    if (M > 1) beta64[M - 1] = stop;
    if (!st) st = upload(wbv, &p->d_wb_val64);
    if (!st) st = upload(wiv, &p->d_wi_val64);
    if (!st) st = upload(wfv, &p->d_wf_val64);
    if (!st) st = upload(vbv, &p->d_vb_val64);
    if (!st) st = upload(p->host.S, &p->d_S64);
    if (!st) st = upload(p->host.quiet, &p->d_quiet64);
    if (!st) st = upload(beta64, &p->d_beta64);
  }
  if (!st && fast_psy_supported(p)) {
    st = fast_psy_plan_init(p);
    if (!st) p->fast = 1;
  }
  if (!st && mid_psy_supported(p)) {   // (beside the fused epilogue's tables where both apply: more than two channels, psy_mid_serves)
    st = mid_psy_plan_init(p);
    if (!st) p->mid = 1;
    if (!st && runs_psy_plan_init(p) == AC_EHIP) st = AC_EHIP;   // (tables without the run structure keep the band walk)
  }
  if (!st && spreading != AC_SPREAD_F32) {
    if (p->fast) {
      p->spread = spreading;
    } else {
      set_error("the matrix-core spreading product needs the wave-level masking model (filter_bands_n 1024 or 2048, 64 Bark "
                "bands); got filter_bands_n = %d, bark_bands_n = %d", N, M);
      st = AC_EUNSUPPORTED;
    }
  }
  if (st) {
    ac_psy_plan_destroy(p);
    return st;
  }
  *out = p;
  return AC_OK;
}

int ac_psy_plan_destroy(ac_psy_plan* p) {
  if (!p) return AC_OK;
  DeviceGuard guard(p->device);
  (void)hipFree(p->d_wb_ptr);
  (void)hipFree(p->d_wb_idx);
  (void)hipFree(p->d_wb_val);
  (void)hipFree(p->d_wi_ptr);
  (void)hipFree(p->d_wi_idx);
  (void)hipFree(p->d_wi_val);
  (void)hipFree(p->d_wf_ptr);
  (void)hipFree(p->d_wf_idx);
  (void)hipFree(p->d_wf_val);
  (void)hipFree(p->d_vb_ptr);
  (void)hipFree(p->d_vb_idx);
  (void)hipFree(p->d_vb_val);
  (void)hipFree(p->d_S);
  (void)hipFree(p->d_quiet);
  (void)hipFree(p->d_beta);
  (void)hipFree(p->d_wb_val64);
  (void)hipFree(p->d_wi_val64);
  (void)hipFree(p->d_wf_val64);
  (void)hipFree(p->d_vb_val64);
  (void)hipFree(p->d_S64);
  (void)hipFree(p->d_quiet64);
  (void)hipFree(p->d_beta64);
  (void)hipFree(p->d_fast);
  (void)hipFree(p->d_mid);
  (void)hipFree(p->d_runs);
  delete p;
  return AC_OK;
}

int ac_mdct_plan_is_fast(const ac_mdct_plan* p) { return p ? p->fast : 0; }

int ac_psy_plan_is_fast(const ac_psy_plan* p) { return p ? p->fast : 0; }
int ac_psy_plan_spreading(const ac_psy_plan* p) { return p ? p->spread : 0; }
int ac_psy_plan_tier(const ac_psy_plan* p) { return !p ? 0 : p->fast ? 2 : p->mid ? 1 : 0; }

// ---- hot path -------------------------------------------------------------------------------------

// whether the wave-level kernels take this call (plans at filters_n 512 / 256 serve a subset, see ac_internal.h)
static bool wave_level(const ac_mdct_plan* p, int C, int iof, int blocks) {
  if (!p->fast || g_force_generic) return false;
  // float32 tensors of more than two channels: the channel-pair instances of the LDS-FFT tier (2.2 - 3 TB/s) rather than
  // the wave-level kernels' strided form (0.9 - 1.5); every float32 route -- one-shot, streaming, the encode -- asks here, so
  // they stay bit-equal to each other.  16-bit PCM of more than two channels keeps the strided form.
  if (iof == 0 && C > 2 && lds_fft_tier_of(p, C) == 2) return false;
  return fast_mdct_frames_per_wave(p->N) == 1 || fast_multi_serves(p, C, iof, blocks);
}
// Which masking-model kernels serve C channels of a plan: the general-layout ones (strided channel pairs) take more than
// two channels off the fused epilogue's kernels too when the plan holds both forms and runs the default spreading product
// (theirs: split bfloat16 on the matrix cores)
static bool psy_mid_serves(const ac_psy_plan* p, int C) {
  return p->mid && !g_force_generic && (!p->fast || (C > 2 && p->spread == AC_SPREAD_BF16X2_MFMA));
}
static bool psy_fast_serves(const ac_psy_plan* p, int C) { return p->fast && !g_force_generic && !psy_mid_serves(p, C); }

int ac_mdct_plan_tier(const ac_mdct_plan* p, int C) {
  if (!p || C < 1) return -1;
  if (wave_level(p, C, 0, 1)) return 3;
  return lds_fft_tier_of(p, C);
}

static int check_dims(int B, int K, int C) {
  AC_REQUIRE(B >= 0 && K >= 0 && C >= 0, "negative dimension (B=%d, blocks=%d, C=%d)", B, K, C);
  return AC_OK;
}

static int mdct_forward(const ac_mdct_plan* p, const void* x, bool pcm16, float* X, int B, int K, int C, void* stream) {
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, K, C);
  if (st) return st;
  if (B == 0 || C == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && (x != nullptr || K == 0), "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x, X);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (wave_level(p, C, pcm16 ? 1 : 0, K))
    return launch_fwd_fast(p, nullptr, x, pcm16, X, nullptr, nullptr, 0.f, nullptr, B, K, K + 1, C, s);
  if (pcm16) {
    if (lds_fft_serves_pcm16(p, C)) return launch_fwd_lds_pcm16(p, static_cast<const int16_t*>(x), X, B, K, C, s);
    set_error("16-bit PCM is served at filters_n 64 ... 2048 in powers of two ('vorbis' or 'sine' window; below 1024: mono / "
              "stereo) and at 120 / 240 / 480 / 960 / 1920 / 576 / 1152 (mono / stereo)");
    return AC_EUNSUPPORTED;
  }
  return launch_fwd_generic(p, static_cast<const float*>(x), X, nullptr, B, K, K + 1, C, s);
}

static int mdct_inverse(const ac_mdct_plan* p, const float* X, void* x, bool pcm16, int B, int Kp, int C, void* stream) {
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, Kp, C);
  if (st) return st;
  if (B == 0 || C == 0) return AC_OK;
  AC_REQUIRE(x != nullptr && (X != nullptr || Kp == 0), "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x, X);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (wave_level(p, C, pcm16 ? 1 : 0, Kp)) return launch_inv_fast(p, X, x, pcm16, nullptr, nullptr, B, Kp, Kp + 1, C, s);
  if (pcm16) {
    if (lds_fft_serves_pcm16(p, C)) return launch_inv_lds_pcm16(p, X, static_cast<int16_t*>(x), B, Kp, C, s);
    set_error("16-bit PCM is served at filters_n 64 ... 2048 in powers of two ('vorbis' or 'sine' window; below 1024: mono / "
              "stereo) and at 120 / 240 / 480 / 960 / 1920 / 576 / 1152 (mono / stereo)");
    return AC_EUNSUPPORTED;
  }
  return launch_inv_generic(p, X, static_cast<float*>(x), nullptr, nullptr, B, Kp, Kp + 1, C, s);
}

int ac_mdct_forward(const ac_mdct_plan* p, const float* x, float* X, int B, int K, int C, void* stream) {
  return mdct_forward(p, x, false, X, B, K, C, stream);
}
int ac_mdct_forward_pcm16(const ac_mdct_plan* p, const int16_t* x, float* X, int B, int K, int C, void* stream) {
  return mdct_forward(p, x, true, X, B, K, C, stream);
}
int ac_mdct_inverse(const ac_mdct_plan* p, const float* X, float* x, int B, int Kp, int C, void* stream) {
  return mdct_inverse(p, X, x, false, B, Kp, C, stream);
}
int ac_mdct_inverse_pcm16(const ac_mdct_plan* p, const float* X, int16_t* x, int B, int Kp, int C, void* stream) {
  return mdct_inverse(p, X, x, true, B, Kp, C, stream);
}

int ac_tonality(const ac_psy_plan* p, const float* X, float* t, int B, int F, int C, void* stream) {
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (psy_fast_serves(p, C)) return launch_psy_fast(p, X, nullptr, t, nullptr, 0.f, B, F, C, s);
  if (psy_mid_serves(p, C)) return launch_psy_mid(p, X, nullptr, t, nullptr, 0.f, B, F, C, s);
  return launch_tonality_generic(p, X, t, B, F, C, s);
}

int ac_mask_threshold(const ac_psy_plan* p, const float* X, const float* t, float drown, float* thr, int B, int F,
                      int C, void* stream) {
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr && thr != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X, thr);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (psy_fast_serves(p, C)) return launch_psy_fast(p, X, t, nullptr, thr, drown, B, F, C, s);
  if (psy_mid_serves(p, C)) return launch_psy_mid(p, X, t, nullptr, thr, drown, B, F, C, s);
  return launch_threshold_generic(p, X, t, drown, thr, B, F, C, s);
}

int ac_tonality_backward(const ac_psy_plan* p, const float* X, const float* grad_t, float* grad_X, int accumulate, int B,
                         int F, int C, void* stream) {
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && grad_t != nullptr && grad_X != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X, grad_X);
  DeviceGuard guard(p->device);
  if (p->fast && !g_force_generic)
    return launch_psy_bwd_fast(p, X, nullptr, 0.f, nullptr, grad_t, grad_X, nullptr, accumulate, B, F, C,
                               (hipStream_t)stream);
  return launch_tonality_bwd_generic(p, X, grad_t, grad_X, accumulate, B, F, C, (hipStream_t)stream);
}

int ac_mask_threshold_backward(const ac_psy_plan* p, const float* X, const float* t, float drown, const float* grad_thr,
                               float* grad_X, float* grad_t, int B, int F, int C, void* stream) {
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr && grad_thr != nullptr && grad_X != nullptr && grad_t != nullptr,
             "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X, grad_thr, grad_X);
  DeviceGuard guard(p->device);
  if (p->fast && !g_force_generic)
    return launch_psy_bwd_fast(p, X, t, drown, grad_thr, nullptr, grad_X, grad_t, 0, B, F, C, (hipStream_t)stream);
  return launch_threshold_bwd_generic(p, X, t, drown, grad_thr, grad_X, grad_t, B, F, C, (hipStream_t)stream);
}

static int encode_fused(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const void* x, bool pcm16, float* X, float* t,
                        float* thr, float drown, int B, int K, int C, void* stream) {
  AC_REQUIRE(mdct != nullptr && psy != nullptr, "plan is NULL");
  AC_REQUIRE(mdct->N == psy->N, "mdct filters_n (%d) != psychoacoustic filter_bands_n (%d)", mdct->N, psy->N);
  AC_REQUIRE(mdct->device == psy->device, "plans live on different devices");
  int st = check_dims(B, K, C);
  if (st) return st;
  if (B == 0 || C == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr && thr != nullptr && (x != nullptr || K == 0), "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x, X, thr);
  DeviceGuard guard(mdct->device);
  hipStream_t s = (hipStream_t)stream;
  if (mdct->fast && psy->fast && !g_force_generic && wave_level(mdct, C, pcm16 ? 1 : 0, K) && psy_fast_serves(psy, C)) {
    // one fused launch, except at filters_n = 2048 for mono input and for 16-bit PCM with 3 or more channels, where the
    // fused kernel would spill registers (7 / 4 at the 256 budget): there two wave-level launches, the second computing
    // tonality and threshold in one pass over X
    if (!(mdct->N == 2048 && (C == 1 || (pcm16 && C > 2))))
      return launch_fwd_fast(mdct, psy, x, pcm16, X, t, thr, drown, nullptr, B, K, K + 1, C, s);
    st = launch_fwd_fast(mdct, nullptr, x, pcm16, X, nullptr, nullptr, 0.f, nullptr, B, K, K + 1, C, s);
    if (!st) st = launch_psy_fast(psy, X, nullptr, t, thr, drown, B, K + 1, C, s);
    return st;
  }
  // filters_n 64 ... 512: the several-frames-per-wave kernels with the general-layout masking model in the same launch
  if (!g_force_generic && mdct->fast && fast_multi_fuses(mdct, psy, C, pcm16 ? 1 : 0, K))
    return launch_fwd_fast(mdct, psy, x, 0, X, t, thr, drown, nullptr, B, K, K + 1, C, s);
  // the LDS-FFT tier's instances with the masking model on the frame while it is in LDS
  if (!pcm16 && !wave_level(mdct, C, 0, K) && K >= 1 && wave_encode_fuses(mdct, psy, C, x, X, thr))
    return launch_enc_wave(mdct, psy, static_cast<const float*>(x), X, t, thr, drown, nullptr, B, K, K + 1, C, s);
  // un-fused composition for configurations the fused kernel does not cover: the transform, then tonality + threshold in
  // one wave-level pass over X where the general-layout masking kernels serve the plan, else the two generic kernels
  st = mdct_forward(mdct, x, pcm16, X, B, K, C, stream);
  if (!st && psy_mid_serves(psy, C)) return launch_psy_mid(psy, X, nullptr, t, thr, drown, B, K + 1, C, s);
  if (!st) st = ac_tonality(psy, X, t, B, K + 1, C, stream);
  if (!st) st = ac_mask_threshold(psy, X, t, drown, thr, B, K + 1, C, stream);
  return st;
}

int ac_encode_launches(const ac_mdct_plan* mdct, const ac_psy_plan* psy, int C) {
  if (!mdct || !psy || mdct->N != psy->N || mdct->device != psy->device || C < 1) return 0;
  if (g_force_generic) return 3;
  if (mdct->fast && psy->fast && wave_level(mdct, C, 0, 1) && psy_fast_serves(psy, C)) return (mdct->N == 2048 && C == 1) ? 2 : 1;   // (see encode_fused)
  if (mdct->fast && fast_multi_fuses(mdct, psy, C, 0, 1)) return 1;
  if (!wave_level(mdct, C, 0, 1) && wave_encode_fuses(mdct, psy, C, nullptr, nullptr, nullptr)) return 1;   // (tensors on the 16-byte grid)
  return (psy_mid_serves(psy, C) || psy_fast_serves(psy, C)) ? 2 : 3;
}

int ac_encode_fused(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr,
                    float drown, int B, int K, int C, void* stream) {
  return encode_fused(mdct, psy, x, false, X, t, thr, drown, B, K, C, stream);
}
int ac_encode_fused_ex(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr,
                       float drown, int flags, float* noisy, float* db_norm, uint64_t seed, int B, int K, int C,
                       void* stream) {
  AC_REQUIRE((flags & ~(AC_EMIT_NOISY | AC_EMIT_DB_NORM)) == 0, "unknown flags %#x", flags);
  AC_REQUIRE(!(flags & AC_EMIT_NOISY) || noisy != nullptr || B == 0 || C == 0, "AC_EMIT_NOISY without an output tensor");
  AC_REQUIRE(!(flags & AC_EMIT_DB_NORM) || db_norm != nullptr || B == 0 || C == 0, "AC_EMIT_DB_NORM without an output tensor");
  float* o_noisy = (flags & AC_EMIT_NOISY) ? noisy : nullptr;
  float* o_dbn = (flags & AC_EMIT_DB_NORM) ? db_norm : nullptr;
  if (!o_noisy && !o_dbn) return encode_fused(mdct, psy, x, false, X, t, thr, drown, B, K, C, stream);
  AC_REQUIRE(mdct != nullptr && psy != nullptr, "plan is NULL");
  AC_REQUIRE(mdct->N == psy->N, "mdct filters_n (%d) != psychoacoustic filter_bands_n (%d)", mdct->N, psy->N);
  AC_REQUIRE(mdct->device == psy->device, "plans live on different devices");
  int st = check_dims(B, K, C);
  if (st) return st;
  if (B == 0 || C == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr && thr != nullptr && (x != nullptr || K == 0), "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x, X, thr, o_noisy, o_dbn);
  if (mdct->fast && psy->fast && !g_force_generic && fast_epilogue_supported(mdct, psy, 0, C)) {
    DeviceGuard guard(mdct->device);
    return launch_fwd_fast(mdct, psy, x, 0, X, t, thr, drown, nullptr, B, K, K + 1, C, (hipStream_t)stream, nullptr, o_noisy,
                           o_dbn, seed);
  }
  // elsewhere: the encode, then the two element-wise kernels over X (same values as the fused epilogue)
  st = encode_fused(mdct, psy, x, false, X, t, thr, drown, B, K, C, stream);
  const size_t n = (size_t)B * (size_t)(K + 1) * (size_t)mdct->N * (size_t)C;
  if (!st && o_noisy) st = ac_add_noise(X, thr, o_noisy, n, seed, stream);
  if (!st && o_dbn) st = ac_amplitude_to_db(X, o_dbn, n, 1, stream);
  return st;
}

int ac_encode_fused_pcm16(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const int16_t* x, float* X, float* t,
                          float* thr, float drown, int B, int K, int C, void* stream) {
  return encode_fused(mdct, psy, x, true, X, t, thr, drown, B, K, C, stream);
}

// ---- streaming ---------------------------------------------------------------------------------------

int ac_stream_create(const ac_mdct_plan* plan, int B, int C, ac_stream** out) {
  AC_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  AC_REQUIRE(plan != nullptr, "plan is NULL");
  AC_REQUIRE(B >= 1 && C >= 1, "B (%d) and C (%d) must be positive", B, C);
  DeviceGuard guard(plan->device);
  ac_stream* s = new (std::nothrow) ac_stream();
  if (!s) {
    set_error("out of host memory");
    return AC_ENOMEM;
  }
  s->plan = plan;
  s->B = B;
  s->C = C;
  s->N = plan->N;
  s->device = plan->device;
  const size_t nb = (size_t)B * plan->N * C * sizeof(float);
  const size_t nt = (size_t)B * C * (plan->N / 2) * sizeof(float);
  hipError_t e = hipMalloc((void**)&s->d_prev_block, nb);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_prev_tmp, nb);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_tail, nt);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_tail_tmp, nt);
  if (e == hipSuccess) e = hipMemset(s->d_prev_block, 0, nb);
  if (e == hipSuccess) e = hipMemset(s->d_prev_tmp, 0, nb);
  if (e == hipSuccess) e = hipMemset(s->d_tail, 0, nt);
  if (e == hipSuccess) e = hipMemset(s->d_tail_tmp, 0, nt);
  if (e != hipSuccess) {
    set_error("stream state allocation failed: %s", hipGetErrorString(e));
    ac_stream_destroy(s);
    return e == hipErrorOutOfMemory ? AC_ENOMEM : AC_EHIP;
  }
  s->d_prev_home = s->d_prev_block;
  s->d_tail_home = s->d_tail;
  *out = s;
  return AC_OK;
}

int ac_stream_reset(ac_stream* s, void* stream) {
  AC_REQUIRE(s != nullptr, "stream is NULL");
  DeviceGuard guard(s->device);
  if (s->d_prev_block != s->d_prev_home) std::swap(s->d_prev_block, s->d_prev_tmp);   // the zero state lives at home
  if (s->d_tail != s->d_tail_home) std::swap(s->d_tail, s->d_tail_tmp);
  const size_t nb = (size_t)s->B * s->N * s->C * sizeof(float);
  const size_t nt = (size_t)s->B * s->C * (s->N / 2) * sizeof(float);
  AC_HIP_CHECK(hipMemsetAsync(s->d_prev_block, 0, nb, (hipStream_t)stream));
  AC_HIP_CHECK(hipMemsetAsync(s->d_tail, 0, nt, (hipStream_t)stream));
  if (s->d_prev64) AC_HIP_CHECK(hipMemsetAsync(s->d_prev64, 0, 2 * nb, (hipStream_t)stream));
  if (s->d_tail64) AC_HIP_CHECK(hipMemsetAsync(s->d_tail64, 0, 2 * nt, (hipStream_t)stream));
  return AC_OK;
}

int ac_stream_destroy(ac_stream* s) {
  if (!s) return AC_OK;
  DeviceGuard guard(s->device);   // (not s->plan->device: the plan may already be gone when a process tears down)
  (void)hipFree(s->d_prev_block);
  (void)hipFree(s->d_prev_tmp);
  (void)hipFree(s->d_tail);
  (void)hipFree(s->d_tail_tmp);
  (void)hipFree(s->d_prev64);
  (void)hipFree(s->d_tail64);
  (void)hipFree(s->d_tail64_tmp);
  delete s;
  return AC_OK;
}

// analysis of one chunk, with or without the masking model; the wave-level kernels write the new state (the chunk's last
// block) themselves into the second state buffer, the other tiers take a strided copy
static int stream_analysis(ac_stream* s, const ac_psy_plan* psy, const float* x_chunk, float* X, float* t, float* thr,
                           float drown, int k, void* stream) {
  AC_REQUIRE(s != nullptr, "stream is NULL");
  AC_REQUIRE(k >= 0, "negative chunk length %d", k);
  if (k == 0) return AC_OK;
  AC_REQUIRE(x_chunk != nullptr && X != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x_chunk, X, thr);
  const ac_mdct_plan* p = s->plan;
  if (psy) {
    AC_REQUIRE(t != nullptr && thr != nullptr, "NULL tensor pointer");
    AC_REQUIRE(p->N == psy->N, "mdct filters_n (%d) != psychoacoustic filter_bands_n (%d)", p->N, psy->N);
    AC_REQUIRE(p->device == psy->device, "plans live on different devices");
  }
  DeviceGuard guard(s->device);
  hipStream_t hs = (hipStream_t)stream;
  int st;
  const bool fast = wave_level(p, s->C, 0, k);
  // (filters_n = 2048 mono: the fused kernel is not instantiated, see encode_fused; filters_n 64 ... 512: the masking model
  // for general band layouts rides in the several-frames-per-wave kernels)
  const bool fused = psy && fast && ((psy_fast_serves(psy, s->C) && fast_mdct_frames_per_wave(p->N) == 1 && !(p->N == 2048 && s->C == 1)) ||
                                     fast_multi_fuses(p, psy, s->C, 0, k));
  if (fast) {
    st = launch_fwd_fast(p, fused ? psy : nullptr, x_chunk, false, X, fused ? t : nullptr, fused ? thr : nullptr, drown,
                         s->d_prev_block, s->B, k, k, s->C, hs, s->d_prev_tmp);
    if (st) return st;
    std::swap(s->d_prev_block, s->d_prev_tmp);
  } else {
    st = launch_fwd_generic(p, x_chunk, X, s->d_prev_block, s->B, k, k, s->C, hs);
    if (st) return st;
    // new state = last block of the chunk, per clip: B rows of N*C floats, source pitch k*N*C floats
    const size_t row = (size_t)p->N * s->C * sizeof(float);
    AC_HIP_CHECK(hipMemcpy2DAsync(s->d_prev_block, row, x_chunk + (size_t)(k - 1) * p->N * s->C, row * k, row,
                                  (size_t)s->B, hipMemcpyDeviceToDevice, hs));
  }
  if (psy && !fused) {
    // the same second step encode_fused takes for these configurations (so that chunked and one-shot results agree bit
    // for bit): tonality + threshold in one wave-level pass over X, or the two generic kernels
    if (psy_fast_serves(psy, s->C)) return launch_psy_fast(psy, X, nullptr, t, thr, drown, s->B, k, s->C, hs);
    if (psy_mid_serves(psy, s->C)) return launch_psy_mid(psy, X, nullptr, t, thr, drown, s->B, k, s->C, hs);
    st = ac_tonality(psy, X, t, s->B, k, s->C, stream);
    if (!st) st = ac_mask_threshold(psy, X, t, drown, thr, s->B, k, s->C, stream);
  }
  return st;
}

int ac_stream_forward(ac_stream* s, const float* x_chunk, float* X, int k, void* stream) {
  return stream_analysis(s, nullptr, x_chunk, X, nullptr, nullptr, 0.f, k, stream);
}

int ac_stream_encode(ac_stream* s, const ac_psy_plan* psy, const float* x_chunk, float* X, float* t, float* thr,
                     float drown, int k, void* stream) {
  AC_REQUIRE(psy != nullptr, "plan is NULL");
  return stream_analysis(s, psy, x_chunk, X, t, thr, drown, k, stream);
}

int ac_stream_inverse(ac_stream* s, const float* X_chunk, float* x, int k, void* stream) {
  AC_REQUIRE(s != nullptr, "stream is NULL");
  AC_REQUIRE(k >= 0, "negative chunk length %d", k);
  if (k == 0) return AC_OK;
  AC_REQUIRE(X_chunk != nullptr && x != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X_chunk, x);
  const ac_mdct_plan* p = s->plan;
  DeviceGuard guard(s->device);
  hipStream_t hs = (hipStream_t)stream;
  int st;
  if (wave_level(p, s->C, 0, k))
    st = launch_inv_fast(p, X_chunk, x, false, s->d_tail, s->d_tail_tmp, s->B, k, k, s->C, hs);
  else
    st = launch_inv_generic(p, X_chunk, x, s->d_tail, s->d_tail_tmp, s->B, k, k, s->C, hs);
  if (st) return st;
  std::swap(s->d_tail, s->d_tail_tmp);
  return AC_OK;
}

// The state buffers swap roles with every chunk; after an odd number of chunks the current state sits in the other
// buffer.  ac_stream_settle moves it back to the stream's home buffers (one small device copy per state, none when it is
// already there).  ac_stream_run settles on entry and on exit, so the launches of a call captured into a HIP graph
// address the home buffers, and a replay reads and writes the same addresses as the captured call -- provided the state is
// at home when the replay starts: after chunk calls (ac_stream_forward / _encode / _inverse), call ac_stream_settle first.
static int settle_state(ac_stream* s, hipStream_t hs) {
  DeviceGuard guard(s->device);
  if (s->d_prev_block != s->d_prev_home) {
    AC_HIP_CHECK(hipMemcpyAsync(s->d_prev_home, s->d_prev_block, (size_t)s->B * s->N * s->C * sizeof(float), hipMemcpyDeviceToDevice, hs));
    std::swap(s->d_prev_block, s->d_prev_tmp);
  }
  if (s->d_tail != s->d_tail_home) {
    AC_HIP_CHECK(hipMemcpyAsync(s->d_tail_home, s->d_tail, (size_t)s->B * s->C * (s->N / 2) * sizeof(float), hipMemcpyDeviceToDevice, hs));
    std::swap(s->d_tail, s->d_tail_tmp);
  }
  return AC_OK;
}

int ac_stream_settle(ac_stream* s, void* stream) {
  AC_REQUIRE(s != nullptr, "stream is NULL");
  return settle_state(s, (hipStream_t)stream);
}

int ac_stream_run(ac_stream* s, const ac_psy_plan* psy, int nchunks, int k, const float* const* x_chunks,
                  float* const* X_chunks, float* const* t_chunks, float* const* thr_chunks, float* const* xhat_chunks,
                  float drown, void* stream) {
  AC_REQUIRE(s != nullptr, "stream is NULL");
  AC_REQUIRE(nchunks >= 0 && k >= 0, "negative chunk count / length (%d, %d)", nchunks, k);
  if (nchunks == 0 || k == 0) return AC_OK;
  AC_REQUIRE(x_chunks != nullptr && X_chunks != nullptr, "NULL chunk list");
  AC_REQUIRE(psy == nullptr || (t_chunks != nullptr && thr_chunks != nullptr), "NULL chunk list");
  for (int i = 0; i < nchunks; ++i) {
    AC_REQUIRE(x_chunks[i] != nullptr && X_chunks[i] != nullptr, "chunk %d: NULL tensor pointer", i);
    AC_REQUIRE(psy == nullptr || (t_chunks[i] != nullptr && thr_chunks[i] != nullptr), "chunk %d: NULL tensor pointer", i);
    AC_REQUIRE(xhat_chunks == nullptr || xhat_chunks[i] != nullptr, "chunk %d: NULL tensor pointer", i);
    AC_REQUIRE_ALIGNED(x_chunks[i], X_chunks[i], psy ? thr_chunks[i] : nullptr, xhat_chunks ? xhat_chunks[i] : nullptr);
  }
  // Small chunks with synthesis (one clip, a few hundred frames per chunk): the analysis of chunk i + 1 rides in the same
  // launch as the synthesis of chunk i (k_duplex_fast) -- the two are independent, and a chunk's two dependent launches
  // of ~8 us each are latency, not bandwidth.  (Synthesis on a second HIP stream beside the analysis was measured slower
  // on MI355X: 25-30 us per chunk of 256 stereo frames against 17-20 us -- a cross-stream event hop costs more than
  // the ~7 us kernel it would hide; DESIGN_LOG.md section 7.)  Same kernel bodies: the results equal the chain's.
  const ac_mdct_plan* p = s->plan;
  hipStream_t hs = (hipStream_t)stream;
  if (psy) {
    AC_REQUIRE(p->N == psy->N, "mdct filters_n (%d) != psychoacoustic filter_bands_n (%d)", p->N, psy->N);
    AC_REQUIRE(p->device == psy->device, "plans live on different devices");
  }
  int st = settle_state(s, hs);   // the state at its home buffers on entry and again on exit
  if (st) return st;
  bool duplex = xhat_chunks && nchunks >= 2 && wave_level(p, s->C, 0, k) && fast_duplex_serves(p, psy, s->B, s->C, k, k);
  if (duplex) {
    // the two halves of a launch must not touch each other's tensors: a caller that reuses one X (or PCM) buffer for
    // consecutive chunks gets the dependent chain
    const size_t nX = (size_t)s->B * k * p->N * s->C, nt = (size_t)s->B * k * s->C;
    auto apart = [](const float* a, size_t na, const float* b, size_t nb) { return a + na <= b || b + nb <= a; };
    for (int i = 0; i + 1 < nchunks && duplex; ++i) {
      duplex = apart(X_chunks[i], nX, X_chunks[i + 1], nX) && apart(xhat_chunks[i], nX, x_chunks[i + 1], nX) &&
               apart(xhat_chunks[i], nX, X_chunks[i + 1], nX) && apart(X_chunks[i], nX, x_chunks[i + 1], nX);
      if (duplex && psy)
        duplex = apart(X_chunks[i], nX, thr_chunks[i + 1], nX) && apart(xhat_chunks[i], nX, thr_chunks[i + 1], nX) &&
                 apart(X_chunks[i], nX, t_chunks[i + 1], nt) && apart(xhat_chunks[i], nX, t_chunks[i + 1], nt);
    }
  }
  if (duplex) {
    DeviceGuard guard(s->device);
    st = stream_analysis(s, psy, x_chunks[0], X_chunks[0], psy ? t_chunks[0] : nullptr, psy ? thr_chunks[0] : nullptr, drown,
                         k, stream);
    for (int i = 0; i + 1 < nchunks && !st; ++i) {
      st = launch_duplex_fast(p, psy, x_chunks[i + 1], X_chunks[i + 1], psy ? t_chunks[i + 1] : nullptr,
                              psy ? thr_chunks[i + 1] : nullptr, drown, s->d_prev_block, s->d_prev_tmp, k, X_chunks[i],
                              xhat_chunks[i], s->d_tail, s->d_tail_tmp, k, s->B, s->C, hs);
      if (!st) {
        std::swap(s->d_prev_block, s->d_prev_tmp);
        std::swap(s->d_tail, s->d_tail_tmp);
      }
    }
    if (!st) st = ac_stream_inverse(s, X_chunks[nchunks - 1], xhat_chunks[nchunks - 1], k, stream);
    return st ? st : settle_state(s, hs);
  }
  for (int i = 0; i < nchunks && !st; ++i) {
    st = stream_analysis(s, psy, x_chunks[i], X_chunks[i], psy ? t_chunks[i] : nullptr, psy ? thr_chunks[i] : nullptr,
                         drown, k, stream);
    if (!st && xhat_chunks) st = ac_stream_inverse(s, X_chunks[i], xhat_chunks[i], k, stream);
  }
  return st ? st : settle_state(s, hs);
}

// ---- element-wise utilities ------------------------------------------------------------------------

int ac_amplitude_to_db(const float* a, float* out, size_t n, int norm, void* stream) {
  AC_REQUIRE(n == 0 || (a != nullptr && out != nullptr), "NULL tensor pointer");
  return launch_db(a, out, n, norm, (hipStream_t)stream);
}

int ac_amplitude_to_db_backward(const float* a, const float* grad_out, float* grad_a, size_t n, int norm, void* stream) {
  AC_REQUIRE(n == 0 || (a != nullptr && grad_out != nullptr && grad_a != nullptr), "NULL tensor pointer");
  return launch_db_bwd(a, grad_out, grad_a, n, norm, (hipStream_t)stream);
}

int ac_add_noise(const float* X, const float* thr, float* out, size_t n, uint64_t seed, void* stream) {
  AC_REQUIRE(n == 0 || (thr != nullptr && out != nullptr), "NULL tensor pointer");   // X == NULL: zeros
  return launch_add_noise(X, thr, out, n, seed, (hipStream_t)stream);
}

// ---- buffer placement probe ---------------------------------------------------------------------
int ac_probe_placement(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const float* x, float* X, float* t,
                       float* const* thr_candidates, int n_candidates, int B, int K, int C, void* stream, int* best,
                       float* ms) {
  AC_REQUIRE(mdct != nullptr && psy != nullptr, "plan is NULL");
  AC_REQUIRE(thr_candidates != nullptr && n_candidates >= 1 && best != nullptr, "no candidates");
  for (int j = 0; j < n_candidates; ++j) AC_REQUIRE(thr_candidates[j] != nullptr, "candidate %d is NULL", j);
  DeviceGuard guard(mdct->device);
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  AC_HIP_CHECK(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    set_error("hipEventCreate failed in ac_probe_placement");
    return AC_EHIP;
  }
  int st = AC_OK, arg = 0;
  float best_ms = 0.f;
  for (int j = 0; j < n_candidates && !st; ++j) {
    float med[3];
    st = ac_encode_fused(mdct, psy, x, X, t, thr_candidates[j], 0.f, B, K, C, stream);   // warm-up
    for (int r = 0; r < 3 && !st; ++r) {
      if (hipEventRecord(e0, s) != hipSuccess) st = AC_EHIP;
      if (!st) st = ac_encode_fused(mdct, psy, x, X, t, thr_candidates[j], 0.f, B, K, C, stream);
      if (!st && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                  hipEventElapsedTime(&med[r], e0, e1) != hipSuccess))
        st = AC_EHIP;
    }
    if (st) break;
    std::sort(med, med + 3);
    if (ms) ms[j] = med[1];
    if (j == 0 || med[1] < best_ms) {
      best_ms = med[1];
      arg = j;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (st == AC_EHIP) set_error("HIP event call failed in ac_probe_placement");
  if (!st) *best = arg;
  return st;
}

// ---- compute_dtype variants ---------------------------------------------------------------------
#define AC_REQUIRE_DTYPE(d) AC_REQUIRE((d) == AC_F32 || (d) == AC_F64 || (d) == AC_BF16, "dtype = %d is not one of AC_F32, AC_F64, AC_BF16", (d))
// the filter bank also takes float16 tensors (mdctransformer.py:327-344); the masking model does not (psychoacoustic.py:42-43)
#define AC_REQUIRE_DTYPE_MDCT(d) AC_REQUIRE((d) == AC_F32 || (d) == AC_F64 || (d) == AC_BF16 || (d) == AC_F16, "dtype = %d is not one of AC_F32, AC_F64, AC_BF16, AC_F16", (d))

int ac_mdct_forward_typed(const ac_mdct_plan* p, const void* x, void* X, int dtype, int B, int K, int C, void* stream) {
  AC_REQUIRE_DTYPE_MDCT(dtype);
  if (dtype == AC_F32) return ac_mdct_forward(p, static_cast<const float*>(x), static_cast<float*>(X), B, K, C, stream);
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, K, C);
  if (st) return st;
  if (B == 0 || C == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && (x != nullptr || K == 0), "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x, X);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AC_F64) return launch_fwd_f64(p, static_cast<const double*>(x), static_cast<double*>(X), B, K, K + 1, C, s);
  if (dtype == AC_F16) return launch_fwd_f16(p, static_cast<const f16_t*>(x), static_cast<f16_t*>(X), B, K, K + 1, C, s);
  if (wave_level(p, C, 2, K) && C <= 2)   // bfloat16 on the wave-level kernels (stereo / mono)
    return launch_fwd_fast(p, nullptr, x, 2, static_cast<float*>(X), nullptr, nullptr, 0.f, nullptr, B, K, K + 1, C, s);
  return launch_fwd_bf16(p, static_cast<const bf16_t*>(x), static_cast<bf16_t*>(X), B, K, K + 1, C, s);
}

int ac_mdct_inverse_typed(const ac_mdct_plan* p, const void* X, void* x, int dtype, int B, int Kp, int C, void* stream) {
  AC_REQUIRE_DTYPE_MDCT(dtype);
  if (dtype == AC_F32) return ac_mdct_inverse(p, static_cast<const float*>(X), static_cast<float*>(x), B, Kp, C, stream);
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, Kp, C);
  if (st) return st;
  if (B == 0 || C == 0) return AC_OK;
  AC_REQUIRE(x != nullptr && (X != nullptr || Kp == 0), "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x, X);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AC_F64) return launch_inv_f64(p, static_cast<const double*>(X), static_cast<double*>(x), B, Kp, Kp + 1, C, s);
  if (dtype == AC_F16) return launch_inv_f16(p, static_cast<const f16_t*>(X), static_cast<f16_t*>(x), B, Kp, Kp + 1, C, s);
  if (wave_level(p, C, 2, Kp) && C <= 2)
    return launch_inv_fast(p, static_cast<const float*>(X), x, 2, nullptr, nullptr, B, Kp, Kp + 1, C, s);
  return launch_inv_bf16(p, static_cast<const bf16_t*>(X), static_cast<bf16_t*>(x), B, Kp, Kp + 1, C, s);
}

int ac_tonality_typed(const ac_psy_plan* p, const void* X, void* t, int dtype, int B, int F, int C, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32) return ac_tonality(p, static_cast<const float*>(X), static_cast<float*>(t), B, F, C, stream);
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AC_F64) return launch_tonality_f64(p, static_cast<const double*>(X), static_cast<double*>(t), B, F, C, s);
  if (p->fast && !g_force_generic && C <= 2)
    return launch_psy_fast(p, static_cast<const float*>(X), nullptr, static_cast<float*>(t), nullptr, 0.f, B, F, C, s, 2);
  return launch_tonality_bf16(p, static_cast<const bf16_t*>(X), static_cast<bf16_t*>(t), B, F, C, s);
}

int ac_mask_threshold_typed(const ac_psy_plan* p, const void* X, const void* t, double drown, void* thr, int dtype, int B,
                            int F, int C, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32)
    return ac_mask_threshold(p, static_cast<const float*>(X), static_cast<const float*>(t), (float)drown,
                             static_cast<float*>(thr), B, F, C, stream);
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr && thr != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X, thr);
  DeviceGuard guard(p->device);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AC_F64)
    return launch_threshold_f64(p, static_cast<const double*>(X), static_cast<const double*>(t), drown,
                                static_cast<double*>(thr), B, F, C, s);
  if (p->fast && !g_force_generic && C <= 2)
    return launch_psy_fast(p, static_cast<const float*>(X), static_cast<const float*>(t), nullptr, static_cast<float*>(thr),
                           (float)drown, B, F, C, s, 2);
  return launch_threshold_bf16(p, static_cast<const bf16_t*>(X), static_cast<const bf16_t*>(t), (float)drown,
                               static_cast<bf16_t*>(thr), B, F, C, s);
}

int ac_encode_fused_typed(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const void* x, void* X, void* t, void* thr,
                          double drown, int dtype, int B, int K, int C, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32)
    return ac_encode_fused(mdct, psy, static_cast<const float*>(x), static_cast<float*>(X), static_cast<float*>(t),
                           static_cast<float*>(thr), (float)drown, B, K, C, stream);
  AC_REQUIRE(mdct != nullptr && psy != nullptr, "plan is NULL");
  AC_REQUIRE(mdct->N == psy->N, "mdct filters_n (%d) != psychoacoustic filter_bands_n (%d)", mdct->N, psy->N);
  AC_REQUIRE(mdct->device == psy->device, "plans live on different devices");
  int st = check_dims(B, K, C);
  if (st) return st;
  if (B == 0 || C == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr && thr != nullptr && (x != nullptr || K == 0), "NULL tensor pointer");
  // bfloat16 tensors, stereo / mono, both plans wave-level: one fused launch (mono at filters_n = 2048 excepted, as in
  // encode_fused); otherwise the three typed entry points in sequence
  if (dtype == AC_BF16 && mdct->fast && psy->fast && !g_force_generic && C <= 2 && !(mdct->N == 2048 && C == 1)) {
    DeviceGuard guard(mdct->device);
    return launch_fwd_fast(mdct, psy, x, 2, static_cast<float*>(X), static_cast<float*>(t), static_cast<float*>(thr),
                           (float)drown, nullptr, B, K, K + 1, C, (hipStream_t)stream);
  }
  st = ac_mdct_forward_typed(mdct, x, X, dtype, B, K, C, stream);
  if (!st) st = ac_tonality_typed(psy, X, t, dtype, B, K + 1, C, stream);
  if (!st) st = ac_mask_threshold_typed(psy, X, t, drown, thr, dtype, B, K + 1, C, stream);
  return st;
}

// ---- streaming on bfloat16 tensors: the wave-level kernels (filters_n 1024 / 2048, mono / stereo) with the conversion in
// their loads and stores; the state stays float32 (a bfloat16 block is exact in it, the aliased half is kept unrounded),
// so chunked results equal the one-shot *_typed calls bit for bit
// float64 streams: the state in double (allocated by the first float64 call), the float64 kernels (O(N^2), any even size)
static int stream_state64(ac_stream* s) {
  if (s->d_prev64) return AC_OK;
  const size_t nb = (size_t)s->B * s->N * s->C * sizeof(double), nt = (size_t)s->B * s->C * (s->N / 2) * sizeof(double);
  hipError_t e = hipMalloc((void**)&s->d_prev64, nb);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_tail64, nt);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_tail64_tmp, nt);
  if (e == hipSuccess) e = hipMemset(s->d_prev64, 0, nb);
  if (e == hipSuccess) e = hipMemset(s->d_tail64, 0, nt);
  if (e == hipSuccess) e = hipMemset(s->d_tail64_tmp, 0, nt);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("float64 stream state allocation failed: %s", hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? AC_ENOMEM : AC_EHIP;
  }
  return AC_OK;
}
static int stream_typed_check(const ac_stream* s, int dtype, int k) {
  AC_REQUIRE(s != nullptr, "stream is NULL");
  AC_REQUIRE_DTYPE(dtype);
  AC_REQUIRE(k >= 0, "negative chunk length %d", k);
  if (dtype == AC_F64) return AC_OK;
  if (!(s->plan->fast && fast_mdct_frames_per_wave(s->N) == 1 && s->C <= 2 && !g_force_generic)) {
    set_error("streaming on bfloat16 tensors is served by the wave-level kernels only (filters_n 1024 / 2048, mono / stereo); "
              "float32 and float64 streams take every size");
    return AC_EUNSUPPORTED;
  }
  return AC_OK;
}

int ac_stream_encode_typed(ac_stream* s, const ac_psy_plan* psy, const void* x_chunk, void* X, void* t, void* thr, double drown,
                           int dtype, int k, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32) {
    if (psy) return ac_stream_encode(s, psy, static_cast<const float*>(x_chunk), static_cast<float*>(X), static_cast<float*>(t),
                                     static_cast<float*>(thr), (float)drown, k, stream);
    return ac_stream_forward(s, static_cast<const float*>(x_chunk), static_cast<float*>(X), k, stream);
  }
  int st = stream_typed_check(s, dtype, k);
  if (st) return st;
  if (k == 0) return AC_OK;
  AC_REQUIRE(x_chunk != nullptr && X != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(x_chunk, X, thr);
  const ac_mdct_plan* p = s->plan;
  if (psy) {
    AC_REQUIRE(t != nullptr && thr != nullptr, "NULL tensor pointer");
    AC_REQUIRE(p->N == psy->N && p->device == psy->device, "plans do not belong together");
    AC_REQUIRE(dtype == AC_F64 || psy->fast, "internal: the masking model of this size has no bfloat16 wave-level kernels");
  }
  DeviceGuard guard(s->device);
  hipStream_t hs = (hipStream_t)stream;
  if (dtype == AC_F64) {   // analysis with the stored block -1, the new state = the chunk's last block, the model on the chunk
    st = stream_state64(s);
    if (st) return st;
    const double* xd = static_cast<const double*>(x_chunk);
    st = launch_fwd_f64_stream(p, xd, static_cast<double*>(X), s->d_prev64, s->B, k, k, s->C, hs);
    if (st) return st;
    const size_t row = (size_t)p->N * s->C * sizeof(double);
    AC_HIP_CHECK(hipMemcpy2DAsync(s->d_prev64, row, xd + (size_t)(k - 1) * p->N * s->C, row * k, row, (size_t)s->B, hipMemcpyDeviceToDevice, hs));
    if (psy) {
      st = launch_tonality_f64(psy, static_cast<const double*>(X), static_cast<double*>(t), s->B, k, s->C, hs);
      if (!st) st = launch_threshold_f64(psy, static_cast<const double*>(X), static_cast<const double*>(t), drown, static_cast<double*>(thr),
                                         s->B, k, s->C, hs);
    }
    return st;
  }
  const bool fused = psy && !(p->N == 2048 && s->C == 1);   // (as ac_encode_fused_typed)
  st = launch_fwd_fast(p, fused ? psy : nullptr, x_chunk, 2, static_cast<float*>(X), fused ? static_cast<float*>(t) : nullptr,
                       fused ? static_cast<float*>(thr) : nullptr, (float)drown, s->d_prev_block, s->B, k, k, s->C, hs, s->d_prev_tmp);
  if (st) return st;
  std::swap(s->d_prev_block, s->d_prev_tmp);
  if (psy && !fused)
    st = launch_psy_fast(psy, static_cast<const float*>(X), nullptr, static_cast<float*>(t), static_cast<float*>(thr), (float)drown,
                         s->B, k, s->C, hs, 2);
  return st;
}

int ac_stream_inverse_typed(ac_stream* s, const void* X_chunk, void* x, int dtype, int k, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32) return ac_stream_inverse(s, static_cast<const float*>(X_chunk), static_cast<float*>(x), k, stream);
  int st = stream_typed_check(s, dtype, k);
  if (st) return st;
  if (k == 0) return AC_OK;
  AC_REQUIRE(X_chunk != nullptr && x != nullptr, "NULL tensor pointer");
  AC_REQUIRE_ALIGNED(X_chunk, x);
  DeviceGuard guard(s->device);
  if (dtype == AC_F64) {
    st = stream_state64(s);
    if (!st) st = launch_inv_f64_stream(s->plan, static_cast<const double*>(X_chunk), static_cast<double*>(x), s->d_tail64, s->d_tail64_tmp,
                                        s->B, k, k, s->C, (hipStream_t)stream);
    if (st) return st;
    std::swap(s->d_tail64, s->d_tail64_tmp);
    return AC_OK;
  }
  st = launch_inv_fast(s->plan, static_cast<const float*>(X_chunk), x, 2, s->d_tail, s->d_tail_tmp, s->B, k, k, s->C, (hipStream_t)stream);
  if (st) return st;
  std::swap(s->d_tail, s->d_tail_tmp);
  return AC_OK;
}

int ac_amplitude_to_db_typed(const void* a, void* out, size_t n, int norm, int dtype, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32) return ac_amplitude_to_db(static_cast<const float*>(a), static_cast<float*>(out), n, norm, stream);
  AC_REQUIRE(n == 0 || (a != nullptr && out != nullptr), "NULL tensor pointer");
  return launch_db_typed(a, out, n, norm, dtype, (hipStream_t)stream);
}

int ac_tonality_backward_typed(const ac_psy_plan* p, const void* X, const void* grad_t, void* grad_X, int dtype, int B, int F, int C,
                               void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32)
    return ac_tonality_backward(p, static_cast<const float*>(X), static_cast<const float*>(grad_t), static_cast<float*>(grad_X), 0, B, F, C, stream);
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && grad_t != nullptr && grad_X != nullptr, "NULL tensor pointer");
  DeviceGuard guard(p->device);
  return launch_tonality_bwd_typed(p, X, grad_t, grad_X, dtype, B, F, C, (hipStream_t)stream);
}

int ac_mask_threshold_backward_typed(const ac_psy_plan* p, const void* X, const void* t, double drown, const void* grad_thr, void* grad_X,
                                     void* grad_t, int dtype, int B, int F, int C, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32)
    return ac_mask_threshold_backward(p, static_cast<const float*>(X), static_cast<const float*>(t), (float)drown,
                                      static_cast<const float*>(grad_thr), static_cast<float*>(grad_X), static_cast<float*>(grad_t), B, F, C, stream);
  AC_REQUIRE(p != nullptr, "plan is NULL");
  int st = check_dims(B, F, C);
  if (st) return st;
  if (B == 0 || C == 0 || F == 0) return AC_OK;
  AC_REQUIRE(X != nullptr && t != nullptr && grad_thr != nullptr && grad_X != nullptr && grad_t != nullptr, "NULL tensor pointer");
  DeviceGuard guard(p->device);
  return launch_threshold_bwd_typed(p, X, t, drown, grad_thr, grad_X, grad_t, dtype, B, F, C, (hipStream_t)stream);
}

int ac_add_noise_typed(const void* X, const void* thr, void* out, size_t n, uint64_t seed, int dtype, void* stream) {
  AC_REQUIRE_DTYPE(dtype);
  if (dtype == AC_F32)
    return ac_add_noise(static_cast<const float*>(X), static_cast<const float*>(thr), static_cast<float*>(out), n, seed, stream);
  AC_REQUIRE(n == 0 || (X != nullptr && thr != nullptr && out != nullptr), "NULL tensor pointer");
  return launch_add_noise_typed(X, thr, out, n, seed, dtype, (hipStream_t)stream);
}

}  // extern "C"
