// The 16-byte kernels of the LDS-FFT tier for channel counts other than one and two in their "team" form (ac_generic.hip,
// k_fwd_wave_c / k_inv_wave_c: whole rows between HBM and LDS, channel pairs picked out of the row image), as a translation
// unit of their own (see ac_wave_rows.hip).  gfx950 only.
#define AC_WAVE_ROWS_TU 4
#include "ac_generic.hip"
