// The 16-byte kernels of the LDS-FFT tier (ac_generic.hip, k_fwd_wave_v / k_inv_wave_v) instantiated for channel pairs of
// any channel count and rows anywhere on the 4-byte grid, as a translation unit of their own (see ac_wave_rows.hip).  gfx950 only.
#define AC_WAVE_ROWS_TU 2
#include "ac_generic.hip"
