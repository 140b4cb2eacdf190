/*
 * audiocodec_amd.h -- C ABI of the MI355X (gfx950) MDCT + psychoacoustic-masking library.
 *
 * The reference (korneelvdbroek/audiocodec) has no FFI: its boundary is the Python API of
 * audiocodec/mdctransformer.py and audiocodec/psychoacoustic.py, whose arithmetic is delegated to
 * TensorFlow ops.  Each entry point below replaces the TensorFlow op sequence of one reference
 * method (cited as file:line into the reference tree); INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to bind them.
 *
 * Conventions
 *  - plain C types only; device pointers are raw HIP device addresses owned by the caller;
 *  - tensor layouts are exactly the reference's, contiguous, float32, channels innermost:
 *        PCM        x   [B, S, C]        S = K * N samples        (mdctransformer.py:104)
 *        spectrum   X   [B, F, N, C]     F frames of N filters    (mdctransformer.py:105-107)
 *        tonality   t   [B, F, 1, C]                              (psychoacoustic.py:110)
 *        threshold  thr [B, F, N, C]                              (psychoacoustic.py:134-135)
 *    PCM, spectrum and threshold tensors start at 16-byte aligned addresses (any allocation does; a view that starts
 *    inside one may not): the kernels move rows with 16- and 8-byte vector accesses.  AC_EINVAL otherwise;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call only
 *    enqueues work and returns without synchronising; the library never allocates, frees or
 *    synchronises caller memory and never changes the current device outside *_create;
 *  - return value 0 = AC_OK, negative = error; ac_last_error() gives a thread-local message;
 *  - plans are immutable after creation and may be shared between host threads and streams.
 *
 * Performance note (MI355X): the kernels stream their tensors at the device's copy rate, and that rate depends on where
 * the caller's buffers live -- when the two tensors a kernel streams side by side (X and thr for ac_encode_fused, X
 * and x for ac_mdct_inverse) sit in stretches of VRAM of the same class, the kernel runs up to 15 % slower.
 * ac_workspace_create (below) hands out buffers placed by timing the encode kernel on a few candidate allocations; the
 * Python package draws the tensors its encode() / decode() return from such a workspace (audiocodec_amd/placement.py).
 */
#ifndef AUDIOCODEC_AMD_H
#define AUDIOCODEC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points declared in this header (and the test hook of
 * audiocodec_amd_testing.h) are its only dynamic symbols. */
#ifndef AC_API
#define AC_API __attribute__((visibility("default")))
#endif

#define AC_VERSION 171 /* 0.1.8: ac_mdct_plan_tier; 16-bit PCM at the Opus / MP3 frame lengths; the LDS-FFT tier on 16-byte kernels with compile-time instances (filters_n % 4 == 0
                          * with a 5-smooth half up to 8192, float32); masking model for general band layouts up to 4096 bins.
                          * 0.1.7: only the ac_* entry points are exported; ac_stream_settle (home buffers for the streaming state);
                          * float32 precompute (ac_*_create_pre, ac_*_host_pre); fused encode at filters_n 64 ... 512; ac_workspace_*.
                          * (0.1.6: ac_stream_run replayable as a HIP graph, duplex launches; filters_n 64 / 128 and 16-bit PCM on the
                          * several-frames-per-wave kernels; mixed-radix FFT tier; masking model for any even filter_bands_n <= 1024) */

enum {
  AC_OK = 0,
  AC_EINVAL = -1,  /* bad shape / size / argument                                   */
  AC_EHIP = -2,    /* a HIP runtime call failed (message carries hipGetErrorString)  */
  AC_ENOMEM = -3,  /* host or device allocation failed                               */
  AC_ENODEV = -4,  /* no usable gfx950 device                                        */
  AC_EUNSUPPORTED = -5
};

/* window_type of MDCTransformer.__init__ (mdctransformer.py:13,199-211) */
enum { AC_WINDOW_VORBIS = 0, AC_WINDOW_SINE = 1, AC_WINDOW_RECT = 2 };

/* element types: `dtype` of the *_typed entry points (compute_dtype) and `precompute` of the *_pre ones (precompute_dtype) */
enum { AC_F32 = 0, AC_F64 = 1, AC_BF16 = 2, AC_F16 = 3 };   /* AC_F16: ac_mdct_forward_typed / ac_mdct_inverse_typed only */

typedef struct ac_mdct_plan ac_mdct_plan;
typedef struct ac_psy_plan ac_psy_plan;
typedef struct ac_stream ac_stream;

AC_API int ac_version(void);
AC_API const char* ac_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Host-only constant builders (fp64 precompute, no GPU needed).
 * ---------------------------------------------------------------------------------------- */

/* Non-zeros of the folding matrix F and of F^-1: replaces _filter_window_matrix / _polyphase_matrix /
 * _inv_polyphase_matrix incl. tf.linalg.inv (mdctransformer.py:155-229).  Eight vectors of N/2 doubles
 * written to coef[8*N/2] in the order a1 a2 a3 a4 s1 s2 s3 s4:
 *   analysis   v[h+j] = a1[j] xc[j] + a2[j] xc[N-1-j],   v[j]       = a3[j] xp[h-1-j] + a4[j] xp[h+j]
 *   synthesis  out[j] = s1[j] u_n[h-1-j] + s2[j] u_{n-1}[h+j],   out[N-1-j] = s3[j] u_n[h-1-j] + s4[j] u_{n-1}[h+j] */
AC_API int ac_mdct_fold_coefficients_host(int N, int window, double* coef);

/* Dense polyphase matrices exactly as the reference stores them: H, H_inv [2,N,N] float32
 * (mdctransformer.py:58-59).  Never used by the kernels; for the Python attributes only. */
AC_API int ac_mdct_dense_matrices_host(int N, int window, float* H, float* H_inv);

/* Constants of PsychoacousticModel.__init__ (psychoacoustic.py:52-69): W [N,M], W_inv [M,N],
 * spreading_matrix [M,M], quiet_threshold_intensity [M] as float32 (any pointer may be NULL) and
 * scalars[4] = {max_frequency, max_bark, bark_band_width, dB_MIN} as double. */
AC_API int ac_psy_tables_host(int N, int M, double sample_rate, double alpha,
                       float* W, float* W_inv, float* S, float* quiet, double* scalars);
/* the same constants unrounded, as PsychoacousticModel holds them with compute_dtype = float64 */
AC_API int ac_psy_tables_host_f64(int N, int M, double sample_rate, double alpha,
                           double* W, double* W_inv, double* S, double* quiet, double* scalars);

/* The same builders in the arithmetic type of the reference's `precompute_dtype` (mdctransformer.py:13-14,31-35,58-59;
 * psychoacoustic.py:14-15,61-69): precompute = AC_F64 (what the functions above compute: the reference's default) or AC_F32
 * -- every constant and operation rounded to float32, in the reference's order, including the cancellation
 * (1 - w[N+j] w[N-1-j]) / w[j] of mdctransformer.py:218-221 (exactly 0 for j = 0 at filters_n = 64 in float32) and the
 * float32 Bark mapping.  The reference's one TensorFlow-generated known-answer vector (tests/test_mdctransformer.py:51-52)
 * stems from a float32-precompute revision.  Values are returned as doubles (exact) / float32 matrices. */
AC_API int ac_mdct_fold_coefficients_host_pre(int N, int window, int precompute, double* coef);
AC_API int ac_mdct_dense_matrices_host_pre(int N, int window, int precompute, float* H, float* H_inv);
AC_API int ac_psy_tables_host_pre(int N, int M, double sample_rate, double alpha, int precompute,
                                  double* W, double* W_inv, double* S, double* quiet, double* scalars);

/* ------------------------------------------------------------------------------------------
 * Plans (own the device copies of the constant tables).
 * ---------------------------------------------------------------------------------------- */

/* MDCTransformer.__init__ (mdctransformer.py:13-59).  N even, >= 2. */
AC_API int ac_mdct_plan_create(int N, int window, int device, ac_mdct_plan** out);
AC_API int ac_mdct_plan_destroy(ac_mdct_plan* plan);

/* PsychoacousticModel.__init__ (psychoacoustic.py:14-69). */
AC_API int ac_psy_plan_create(int N, int M, double sample_rate, double alpha, int device, ac_psy_plan** out);
AC_API int ac_psy_plan_destroy(ac_psy_plan* plan);

/* Plans whose constants are computed in `precompute` = AC_F64 (as the plain *_create) or AC_F32 (see the *_host_pre
 * builders).  With float32 constants the 2x2 fold blocks of F are no longer exact rotations, so the MDCT runs kernels
 * that carry all four coefficients per block: the several-frames-per-wave kernels at filters_n 64 ... 512 (mono / stereo,
 * float32), the LDS-FFT / O(N^2) tiers elsewhere.  ac_psy_plan_create_pre: spreading = AC_SPREAD_* or -1 for the default
 * of ac_psy_plan_create. */
AC_API int ac_mdct_plan_create_pre(int N, int window, int precompute, int device, ac_mdct_plan** out);
/* The transposed filter bank of `plan` as a plan of its own (same size, device, kernels): with T the analysis bank and S the
 * synthesis bank of `plan`, ac_mdct_inverse(adjoint, g)[:, N:-N] = 4 N T^T g and ac_mdct_forward(adjoint, g)[:, 1:-1] =
 * S^T g / (4 N) -- what the backward passes of transform / inverse_transform need (the reference is differentiated by
 * TensorFlow, mdctransformer.py:62-153).  The DCT-IV is symmetric, so only the O(N) fold transposes.  For Princen-Bradley
 * windows computed in float64 the adjoint equals the plan itself; for the rectangular window (mdctransformer.py:209-229)
 * and float32-precomputed constants it does not. */
AC_API int ac_mdct_plan_adjoint(const ac_mdct_plan* plan, ac_mdct_plan** out);
AC_API int ac_psy_plan_create_pre(int N, int M, double sample_rate, double alpha, int device, int spreading, int precompute,
                                  ac_psy_plan** out);

/* Form of the band x band product with the spreading matrix (psychoacoustic.py:205-207: sum_i max(eps, P_i)^alpha S[i,j])
 * in the wave-level kernels -- BASELINE configs[3] "Bark spreading cast as band x band MFMA contraction, bf16":
 *   AC_SPREAD_F32          float32 multiply-adds on the vector ALU;
 *   AC_SPREAD_BF16_MFMA    operands rounded to bfloat16, v_mfma_f32_4x4x4_16b_bf16, float32 accumulation: thresholds
 *                          within 5e-3 relative of the float32 form (bfloat16 has 8 mantissa bits);
 *   AC_SPREAD_BF16X2_MFMA  both operands split into bfloat16 hi + lo parts, four partial products on the matrix cores:
 *                          thresholds within 1e-5 of the float32 form, i.e. inside the 1e-4 parity bar.  This is what
 *                          ac_psy_plan_create builds where the wave-level kernels serve the plan (2 % faster fused
 *                          encode); everywhere else it builds AC_SPREAD_F32.
 * Served for stereo input (float32 or 16-bit PCM) by ac_encode_fused* and ac_mask_threshold; other channel counts and
 * the bfloat16 tensors of the *_typed entry points keep the float32 product.  AC_EUNSUPPORTED unless the plan runs the wave-level kernels. */
enum { AC_SPREAD_F32 = 0, AC_SPREAD_BF16_MFMA = 1, AC_SPREAD_BF16X2_MFMA = 2 };
AC_API int ac_psy_plan_create_ex(int N, int M, double sample_rate, double alpha, int device, int spreading, ac_psy_plan** out);
AC_API int ac_psy_plan_spreading(const ac_psy_plan* plan);

/* 1 when the plan runs the wave-level kernels (filters_n 1024 / 2048, Princen-Bradley window; 64 Bark bands), 0 when it
 * runs the LDS-FFT middle tier (filters_n from 16 to 4096 with a 5-smooth half: 2^a 3^b 5^c; up to 8192 for float32
 * tensors with filters_n % 4 == 0) or the generic O(N^2) kernels. */
AC_API int ac_mdct_plan_is_fast(const ac_mdct_plan* plan);
/* Which kernels serve float32 tensors of `channels_n` channels: 3 = the wave-level kernels (mono / stereo); 2 = a compile-time instance of the
 * LDS-FFT tier's 16-byte kernels (filters_n % 4 == 0 with a 5-smooth half, up to 8192); 1 = the tier's run-time forms;
 * 0 = the O(N^2) kernels; -1 = bad argument. */
AC_API int ac_mdct_plan_tier(const ac_mdct_plan* plan, int channels_n);
AC_API int ac_psy_plan_is_fast(const ac_psy_plan* plan);
/* Which kernels serve the masking model of a plan: 2 = the wave-level kernels fused into the encode (filter_bands_n 1024 /
 * 2048, 64 Bark bands, every bin in at most two adjacent bands); 1 = the wave-level kernels for general band layouts
 * (any even filter_bands_n up to 4096, up to 64 bands; any channel count);
 * 0 = the generic kernels (one workgroup per channel-frame). */
AC_API int ac_psy_plan_tier(const ac_psy_plan* plan);

/* (The test hook that forces the generic kernels is declared in audiocodec_amd_testing.h, not here.) */

/* ------------------------------------------------------------------------------------------
 * Hot path.
 * ---------------------------------------------------------------------------------------- */

/* MDCTransformer.transform (mdctransformer.py:62-125):  x [B, K*N, C] -> X [B, K+1, N, C]. */
AC_API int ac_mdct_forward(const ac_mdct_plan* plan, const float* x, float* X, int B, int K, int C, void* stream);

/* MDCTransformer.inverse_transform (mdctransformer.py:128-153):  X [B, Kp, N, C] -> x [B, (Kp+1)*N, C]. */
AC_API int ac_mdct_inverse(const ac_mdct_plan* plan, const float* X, float* x, int B, int Kp, int C, void* stream);

/* PsychoacousticModel.tonality (psychoacoustic.py:102-120):  X [B, F, N, C] -> t [B, F, 1, C]. */
AC_API int ac_tonality(const ac_psy_plan* plan, const float* X, float* t, int B, int F, int C, void* stream);

/* PsychoacousticModel.global_masking_threshold (psychoacoustic.py:122-148, with 169-210, 301-331):
 * X [B,F,N,C], t [B,F,1,C], drown in [0,1] -> thr [B,F,N,C]. */
AC_API int ac_mask_threshold(const ac_psy_plan* plan, const float* X, const float* t, float drown, float* thr,
                      int B, int F, int C, void* stream);

/* Backward passes of the masking model (the reference is differentiated by TensorFlow when it is used inside a
 * training graph -- see the gradient remark at psychoacoustic.py:311; here the adjoints are explicit kernels).
 * ac_tonality_backward: grad_t [B,F,1,C] -> grad_X [B,F,N,C] (d tonality / d X, psychoacoustic.py:102-120);
 *   accumulate != 0 adds into grad_X instead of overwriting it.
 * ac_mask_threshold_backward: grad_thr [B,F,N,C] -> grad_X [B,F,N,C] and grad_t [B,F,1,C]
 *   (d threshold / d X and d threshold / d tonality, psychoacoustic.py:122-148 with 169-210, 301-331). */
AC_API int ac_tonality_backward(const ac_psy_plan* plan, const float* X, const float* grad_t, float* grad_X, int accumulate,
                         int B, int F, int C, void* stream);
AC_API int ac_mask_threshold_backward(const ac_psy_plan* plan, const float* X, const float* t, float drown,
                               const float* grad_thr, float* grad_X, float* grad_t, int B, int F, int C, void* stream);

/* Fused encode = transform -> tonality -> global_masking_threshold in one pass over the PCM
 * (the composition of tests/test_psychoacoustic.py:38-41 + psychoacoustic.py:130-131).
 * x [B,K*N,C] -> X [B,K+1,N,C], t [B,K+1,1,C], thr [B,K+1,N,C].  mdct N must equal psy N. */
AC_API int ac_encode_fused(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const float* x, float* X, float* t,
                    float* thr, float drown, int B, int K, int C, void* stream);

/* How ac_encode_fused serves (mdct, psy) on tensors of C channels: 1 = ONE launch (filters_n 1024 / 2048 with the fused
 * wave-level epilogue; filters_n 64 ... 512, mono / stereo, with the masking model for general band layouts in the
 * several-frames-per-wave kernels; the LDS-FFT sizes where one launch measured faster than two on an MI355X -- 33 sizes
 * 540 ... 4096, 960 among them -- mono / stereo float32), 2 = the transform, then tonality + threshold in one pass over X,
 * 3 = three launches (generic kernels).  0 for inconsistent plans.  The results do not depend on it: the fused launches
 * return what the separate calls return, bit for bit (filters_n 1024 / 2048: within 1e-5). */
AC_API int ac_encode_launches(const ac_mdct_plan* mdct, const ac_psy_plan* psy, int C);

/* The fused encode with its element-wise tail (psychoacoustic.py:150-167 and :87-100 on the frames just computed), so
 * that a caller who wants them does not pay another pass over X and thr:
 *   AC_EMIT_NOISY    noisy   [B,K+1,N,C] = X + thr * Normal(0, 1/6): the values ac_add_noise(X, thr, seed) gives, bit for bit;
 *   AC_EMIT_DB_NORM  db_norm [B,K+1,N,C] = amplitude_to_dB_norm(X): the values ac_amplitude_to_db(X, norm = 1) gives.
 * One launch for stereo float32 input on the wave-level kernels at filters_n = 1024; elsewhere the encode followed by
 * the two element-wise kernels.  flags = 0 is ac_encode_fused. */
enum { AC_EMIT_NOISY = 1, AC_EMIT_DB_NORM = 2 };
AC_API int ac_encode_fused_ex(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const float* x, float* X, float* t, float* thr,
                       float drown, int flags, float* noisy, float* db_norm, uint64_t seed, int B, int K, int C,
                       void* stream);

/* 16-bit PCM at the boundary (extension; the reference takes float PCM in [-1, 1] only, mdctransformer.py:104):
 * x = pcm / 32768 on the way in, pcm = clamp(round(32768 x), -32768, 32767) on the way out, fused into the kernels'
 * loads / stores, so a frame moves 2 bytes per sample instead of 4.  Served by the wave-level kernels (filters_n 64 ... 2048
 * in powers of two, 'vorbis' / 'sine' window; below 1024 mono / stereo) and by the LDS-FFT tier at filters_n 120, 240, 480,
 * 960, 1920, 576, 1152 (mono / stereo, any window); AC_EUNSUPPORTED otherwise.  Shapes as the float32 entry points. */
AC_API int ac_mdct_forward_pcm16(const ac_mdct_plan* plan, const int16_t* x, float* X, int B, int K, int C, void* stream);
AC_API int ac_mdct_inverse_pcm16(const ac_mdct_plan* plan, const float* X, int16_t* x, int B, int Kp, int C, void* stream);
AC_API int ac_encode_fused_pcm16(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const int16_t* x, float* X, float* t,
                          float* thr, float drown, int B, int K, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Streaming overlap-add (chunked transform with device-resident state).
 * Analysis state: the last input block per (b,c) [B,N,C]; synthesis state: the aliased second half of
 * the last DCT-IV output per (b,c) [B,C,N/2].  A fresh/reset stream starts from zero state, so the
 * concatenation of chunk outputs equals the one-shot transform frame for frame.
 * ---------------------------------------------------------------------------------------- */
AC_API int ac_stream_create(const ac_mdct_plan* plan, int B, int C, ac_stream** out);
AC_API int ac_stream_reset(ac_stream* s, void* stream);
AC_API int ac_stream_destroy(ac_stream* s);
/* x_chunk [B, k*N, C] -> X [B, k, N, C]  (frame i = blocks i-1, i; block -1 = state) */
AC_API int ac_stream_forward(ac_stream* s, const float* x_chunk, float* X, int k, void* stream);
/* the same with the masking model on the chunk's frames: X, t [B, k, 1, C], thr [B, k, N, C] equal, frame for frame and bit
 * for bit, what ac_encode_fused returns for the whole signal (one fused launch where the wave-level kernels serve both
 * plans; psychoacoustic.py:102-148 on mdctransformer.py:62-125) */
AC_API int ac_stream_encode(ac_stream* s, const ac_psy_plan* psy, const float* x_chunk, float* X, float* t, float* thr,
                     float drown, int k, void* stream);
/* X_chunk [B, k, N, C] -> x [B, k*N, C]  (block i = frames i, i-1; frame -1 = state) */
AC_API int ac_stream_inverse(ac_stream* s, const float* X_chunk, float* x, int k, void* stream);
/* The two state buffers of a stream are double-buffered (a kernel reads one while it writes the other), so the current
 * state changes address with every chunk call.  ac_stream_settle moves it back to the stream's home buffers: at most two
 * small device-to-device copies on `stream` (B*N*C and B*C*N/2 floats), nothing when it is already there.  Needed only by
 * callers that replay a captured ac_stream_run (below) after chunk calls: settle before capturing and before each such replay. */
AC_API int ac_stream_settle(ac_stream* s, void* stream);

/* Feeds nchunks consecutive chunks of k blocks through the stream in one call (a long device-resident signal, a ring
 * buffer that has filled up): chunk i is analysed -- with the masking model when psy is not NULL, else t_chunks /
 * thr_chunks are ignored -- from x_chunks[i] [B, k*N, C] into X_chunks[i] (t_chunks[i], thr_chunks[i]) and, when
 * xhat_chunks is not NULL, synthesised from X_chunks[i] into xhat_chunks[i] [B, k*N, C].  Results are those of nchunks
 * calls of ac_stream_encode (ac_stream_forward) and ac_stream_inverse on `stream`, issued from one host call (no per-call
 * binding overhead).  With synthesis, chunks small enough to be latency-bound (one or a few clips) and distinct buffers
 * per chunk, the analysis of chunk i + 1 and the synthesis of chunk i share ONE launch (float32, mono / stereo,
 * filters_n 1024 / 2048): nchunks + 1 launches instead of 2 nchunks, same results bit for bit.  A caller that reuses
 * one buffer for consecutive chunks gets the dependent chain.  The call settles the stream's state at its home buffers on
 * entry and on exit (ac_stream_settle) and nothing synchronises or allocates: a call is capturable into a HIP graph
 * (hipStreamBeginCapture on `stream`; call ac_stream_settle BEFORE the capture so that no state copy is recorded), and
 * replaying the graph repeats it on new contents of the same buffers -- 7.9 us per chunk of one stereo clip against
 * 11.9 us for the plain launches (the gaps between dependent launches go).  A replay addresses the home buffers: if chunk
 * calls (ac_stream_forward / _encode / _inverse) or ac_stream_reset ran since the last ac_stream_run, call
 * ac_stream_settle before replaying.  The pointer lists are host arrays. */
AC_API int ac_stream_run(ac_stream* s, const ac_psy_plan* psy, int nchunks, int k, const float* const* x_chunks,
                  float* const* X_chunks, float* const* t_chunks, float* const* thr_chunks, float* const* xhat_chunks,
                  float drown, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise utilities of PsychoacousticModel (psychoacoustic.py:71-100, 150-167).
 * ---------------------------------------------------------------------------------------- */
/* amplitude_to_dB (norm = 0) / amplitude_to_dB_norm (norm = 1) on n floats. */
AC_API int ac_amplitude_to_db(const float* a, float* out, size_t n, int norm, void* stream);
/* d amplitude_to_dB(_norm) / d a: grad_a = grad_out * (20 / ln 10) / a where a^2 > 1e-14 (0 inside the clamp), / 140 for
 * the normalised form. */
AC_API int ac_amplitude_to_db_backward(const float* a, const float* grad_out, float* grad_a, size_t n, int norm, void* stream);
/* add_noise: out = X + thr * Normal(0, 1/6), counter-based generator keyed by (seed, element-pair index).  X == NULL
 * stands for zeros: ac_add_noise(NULL, g, out, n, seed) is the gradient of add_noise with respect to the threshold
 * (the gradient with respect to X is g itself). */
AC_API int ac_add_noise(const float* X, const float* thr, float* out, size_t n, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Placed buffers (see the performance note at the top).  ac_workspace_create allocates, straight
 * from the HIP runtime, device buffers for batches of B clips x K blocks x C channels of this (mdct, psy) pair:
 *   region A = [copies x (X [B,K+1,N,C] | t [B,K+1,1,C]) | x [B,K*N,C]],   region B = [copies x (thr like X | xhat [B,(K+2)*N,C])]
 * and places region B for the MI355X's HBM: up to max_tries candidate allocations are timed with the fused encode itself
 * (median of three launches on noise, device warmed first), untouched spacers of 12 GiB (at most span_gib in all) move
 * each next candidate along the VRAM, the search stops at the first candidate that reaches the two-class rate or once two
 * candidates differ by the gap between the classes; the fastest stays, every other allocation and every spacer is back
 * with the driver before the call returns.  The one entry point (with ac_probe_placement) that synchronises.  Memory
 * held afterwards: exactly the two regions (ac_workspace_regions).  copies > 1 gives double-buffered outputs.
 * ac_workspace_buffers returns the tensors of copy `copy` (any pointer argument may be NULL); x is filled with
 * uniform(-1, 1) noise by the probe.  Results of the kernels do not depend on where their buffers live.
 * ---------------------------------------------------------------------------------------- */
typedef struct ac_workspace ac_workspace;
AC_API int ac_workspace_create(const ac_mdct_plan* mdct, const ac_psy_plan* psy, int B, int K, int C, int copies, int max_tries,
                               double span_gib, void* stream, ac_workspace** out);
AC_API int ac_workspace_buffers(const ac_workspace* ws, int copy, float** x, float** X, float** t, float** thr, float** xhat);
AC_API int ac_workspace_regions(const ac_workspace* ws, void** a, size_t* bytes_a, void** b, size_t* bytes_b);
/* tries made, index of the chosen one, encode time of every try in ms (encode_ms: room for 16 floats), spacer memory
 * held for a moment during the search in GiB */
AC_API int ac_workspace_report(const ac_workspace* ws, int* tries, int* chosen, float* encode_ms, double* spacer_gib);
AC_API int ac_workspace_destroy(ac_workspace* ws);
/* The two regions as a pool: a float32 tensor of `shape` (ndim <= 8, compact row-major) carved out of region 0 (A: spectra)
 * or 1 (B: thresholds, PCM), returned as a DLManagedTensor* (dlpack.h) whose deleter gives the memory back -- what
 * torch.from_dlpack / any DLPack consumer turns into a tensor that owns its storage.  A released extent is handed out again
 * only for work on the stream its last tenant was allocated for (same-stream order is execution order).  NULL when the
 * region has no room: the caller then allocates elsewhere.  Once used, the fixed tensors of ac_workspace_buffers overlap
 * the pool's and must not be used; ac_workspace_destroy defers freeing the regions until the last tensor is released.
 * ac_workspace_live: tensors handed out and not yet released.
 * ac_workspace_record_stream: the pool's counterpart of torch's Tensor.record_stream (which a DLPack tensor's storage does not
 * reach): tells the pool that the tensor starting at `data` is also used by work on `stream`.  When the tensor dies, the pool
 * records an event on every such stream and the extent's next tenant -- whatever stream it is allocated for -- waits for them
 * first, so a consumer on a side stream that drops its last reference early cannot have the memory rewritten under it.
 * AC_EINVAL when `data` is not the start of a live tensor of this pool. */
AC_API void* ac_workspace_alloc_dlpack(ac_workspace* ws, int region, int ndim, const int64_t* shape, void* stream);
AC_API long ac_workspace_live(ac_workspace* ws);
AC_API int ac_workspace_record_stream(ac_workspace* ws, const void* data, void* stream);

/* The probe on its own: runs ac_encode_fused with each of the n caller-owned candidate threshold buffers (x, X, t fixed;
 * contents of X, t and the candidates are overwritten), times every candidate with HIP events (median of three launches
 * after one warm-up) and returns the index of the fastest in *best and, when ms is not NULL, the n times in milliseconds. */
AC_API int ac_probe_placement(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const float* x, float* X, float* t,
                       float* const* thr_candidates, int n_candidates, int B, int K, int C, void* stream, int* best,
                       float* ms);

/* ------------------------------------------------------------------------------------------
 * compute_dtype variants (mdctransformer.py:13-14,22-23; psychoacoustic.py:14-15,30,42-43: the reference accepts
 * float64 / float32 / bfloat16 and requires the inputs to be of that type).  Same shapes and layouts as the float32
 * entry points; `dtype` names the element type of every tensor argument:
 *   AC_F32   the float32 entry points above (wave-level / LDS-FFT / generic kernels);
 *   AC_F64   float64 tensors, float64 arithmetic and float64 constants throughout (O(N^2) DCT-IV, any even
 *            filters_n): the on-device oracle the tests hold the float32 kernels against at full size;
 *   AC_BF16  bfloat16 tensors (half the bytes of float32), float32 arithmetic inside: the wave-level kernels for mono
 *            and stereo at filters_n 1024 / 2048 (conversion fused into their loads and stores; the fused encode rounds
 *            X and the tonality to bfloat16 before the masking model uses them, so fused and un-fused calls agree),
 *            else the LDS-FFT kernels for filters_n from 16 to 4096 with a 5-smooth half and the O(N^2) kernels; results carry
 *            bfloat16's output rounding (2^-9 relative) -- more accurate than the reference's all-bfloat16 op sequence.
 *   AC_F16   float16 tensors, float32 arithmetic inside; the filter bank only (ac_mdct_forward_typed / ac_mdct_inverse_typed:
 *            MDCTransformer accepts it and up-casts inside its DCT-IV, mdctransformer.py:327-344; PsychoacousticModel refuses it
 *            by name, psychoacoustic.py:42-43 -- the masking entry points return AC_EINVAL): the 8-byte LDS-FFT kernels for
 *            filters_n from 16 to 4096 with a 5-smooth half, else the O(N^2) kernels; results carry float16's rounding
 *            (2^-11 relative) and range (a coefficient beyond 65504 becomes infinity, as a cast makes it).
 *
 * Non-finite inputs (all dtypes): the filter bank is linear arithmetic -- a NaN or infinite sample makes every coefficient
 * of the two frames that contain it NaN (as the reference's dense products do).  The masking model follows tf.maximum /
 * tf.minimum, which propagate NaN (psychoacoustic.py:113-116, 205-208, 331): a frame and signal with a NaN or infinite
 * intensity has a NaN tonality, and a NaN intensity or a NaN tonality makes the whole threshold row of that frame and signal
 * NaN; every other frame and the other signal are untouched.  One corner differs: intensities that overflow float32
 * (|X| > 1.8e19) with a caller-supplied finite tonality give NaN thresholds where the reference has infinities.  Denormal
 * inputs are ordinary numbers (nothing is flushed on the way in; intensities below 1e-14 meet the model's floor).
 * Streaming (ac_stream_*_typed) serves AC_F32 and AC_F64 at every size (float64: a state of its own in double, the float64
 * kernels) and AC_BF16 where the wave-level kernels do (filters_n 1024 / 2048, mono / stereo; the state stays float32);
 * chunked results equal the one-shot calls bit for bit.  Backward passes: the filter bank through ac_mdct_plan_adjoint and
 * the typed forward entry points in every dtype; the masking model through the *_backward_typed entry points below.
 * ---------------------------------------------------------------------------------------- */
AC_API int ac_mdct_forward_typed(const ac_mdct_plan* plan, const void* x, void* X, int dtype, int B, int K, int C, void* stream);
AC_API int ac_mdct_inverse_typed(const ac_mdct_plan* plan, const void* X, void* x, int dtype, int B, int Kp, int C, void* stream);
AC_API int ac_tonality_typed(const ac_psy_plan* plan, const void* X, void* t, int dtype, int B, int F, int C, void* stream);
AC_API int ac_mask_threshold_typed(const ac_psy_plan* plan, const void* X, const void* t, double drown, void* thr, int dtype,
                            int B, int F, int C, void* stream);
/* the adjoints of the two (ac_tonality_backward / ac_mask_threshold_backward on tensors of `dtype`: the reference's op chain is
 * differentiable in every dtype it accepts, psychoacoustic.py:311): AC_F64 in float64 throughout, AC_BF16 bfloat16 tensors with
 * float32 arithmetic; grad_X is overwritten (no accumulate form) */
AC_API int ac_tonality_backward_typed(const ac_psy_plan* plan, const void* X, const void* grad_t, void* grad_X, int dtype, int B, int F,
                                      int C, void* stream);
AC_API int ac_mask_threshold_backward_typed(const ac_psy_plan* plan, const void* X, const void* t, double drown, const void* grad_thr,
                                            void* grad_X, void* grad_t, int dtype, int B, int F, int C, void* stream);
AC_API int ac_encode_fused_typed(const ac_mdct_plan* mdct, const ac_psy_plan* psy, const void* x, void* X, void* t, void* thr,
                          double drown, int dtype, int B, int K, int C, void* stream);
/* ac_stream_forward / ac_stream_encode (psy may be NULL: then t, thr are ignored) and ac_stream_inverse on tensors of `dtype` */
AC_API int ac_stream_encode_typed(ac_stream* s, const ac_psy_plan* psy, const void* x_chunk, void* X, void* t, void* thr,
                                  double drown, int dtype, int k, void* stream);
AC_API int ac_stream_inverse_typed(ac_stream* s, const void* X_chunk, void* x, int dtype, int k, void* stream);
AC_API int ac_amplitude_to_db_typed(const void* a, void* out, size_t n, int norm, int dtype, void* stream);
AC_API int ac_add_noise_typed(const void* X, const void* thr, void* out, size_t n, uint64_t seed, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOCODEC_AMD_H */
