// placeholder: wave-level FFT kernels land here
#include "ac_internal.h"
namespace ac {
bool fast_mdct_supported(int) { return false; }
bool fast_psy_supported(const ac_psy_plan*) { return false; }
int fast_mdct_plan_init(ac_mdct_plan*) { return AC_OK; }
int fast_psy_plan_init(ac_psy_plan*) { return AC_OK; }
int launch_fwd_fast(const ac_mdct_plan*, const ac_psy_plan*, const float*, float*, float*, float*, float, const float*, int, int, int, int, hipStream_t) { set_error("fast path not built"); return AC_EUNSUPPORTED; }
int launch_inv_fast(const ac_mdct_plan*, const float*, float*, const float*, float*, int, int, int, int, hipStream_t) { set_error("fast path not built"); return AC_EUNSUPPORTED; }
int launch_psy_fast(const ac_psy_plan*, const float*, const float*, float*, float*, float, int, int, int, hipStream_t) { set_error("fast path not built"); return AC_EUNSUPPORTED; }
}
