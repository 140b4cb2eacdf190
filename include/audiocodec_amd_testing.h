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

#ifdef __cplusplus
}
#endif
#endif /* AUDIOCODEC_AMD_TESTING_H */
