/*
 * audiocodec_amd_testing.h -- test hooks of libaudiocodec_amd.so.  Not part of the product API: nothing under
 * audiocodec_amd/ calls these outside the test suite's fixtures.
 */
#ifndef AUDIOCODEC_AMD_TESTING_H
#define AUDIOCODEC_AMD_TESTING_H

#include "audiocodec_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Process-global: force the generic O(N^2) kernels (1) or restore automatic selection (0), so that the GPU parity tests
 * can hold every wave-level tier against an independent on-device implementation.  Honoured only in a process started
 * with AC_TESTING=1 in its environment; AC_EUNSUPPORTED otherwise. */
AC_API int ac_set_force_generic(int on);

/* Host only (no device needed): the run-structured image of the general-layout masking model (csrc/ac_psy_runs_dev.h) for a
 * model's tables, so that the CPU tests can replay the kernel's arithmetic on it in numpy and hold the plan-time structure
 * -- band lists over partial sums, threshold entries, per-bin entry offsets -- against the oracle before any GPU runs.
 * layout[17] = {words, lw, kb, n4, n16, n64, o4, o16, o64, oz, slot, off_S, off_bc, off_bd, off_lst, off_bw, off_idx};
 * image (may be NULL) receives min(words, cap) 32-bit words.  AC_EUNSUPPORTED when the tables lack the structure. */
AC_API int ac_testing_runs_image(int N, int M, double sample_rate, double alpha, int precompute, unsigned* image, int cap,
                                 int* layout);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOCODEC_AMD_TESTING_H */
