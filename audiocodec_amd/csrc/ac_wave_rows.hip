// The 16-byte kernels of the LDS-FFT tier (ac_generic.hip, k_fwd_wave_v / k_inv_wave_v) instantiated for mono rows -- two mono
// signals per complex pair -- as a translation unit of their own: ac_generic.hip compiled a second time with everything but
// those instances and their two launchers left out (the stereo instances alone take a minute to compile).  gfx950 only.
#define AC_WAVE_ROWS_TU 1
#include "ac_generic.hip"
