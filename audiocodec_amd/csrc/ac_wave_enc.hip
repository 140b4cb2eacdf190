// The fused encode of the LDS-FFT tier (k_enc_wave_v: the 16-byte analysis kernels with the run-structured masking model on
// the frame while it is in LDS): ac_generic.hip compiled again for these instances only (their compile is minutes: a
// translation unit of its own, like ac_wave_rows.hip)
#define AC_WAVE_ROWS_TU 3
#include "ac_generic.hip"
