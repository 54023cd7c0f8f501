// ss_td_f4.hip: the time-domain kernels of 4x true-peak oversampling (rates below 96 kHz — the bench shape, the reference's
// 44.1 / 48 kHz files, the 48 kHz tick).  See ss_td_impl.h.
#define SS_TD_DEBUG_OWNER 1      // (development builds: the phase clocks / the trace buffer live in this translation unit)
#include "ss_td_impl.h"

namespace ssk {
SS_TD_DEFINE_FACTOR(4)
}  // namespace ssk

#ifdef SS_TD_TRACE
extern "C" int ss_debug_td_trace(unsigned long long *out512)
{
    return hipMemcpyFromSymbol(out512, HIP_SYMBOL(ssk::g_td_trace), 32 * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef SS_TD_PROF
// development builds only: [0..7] phase clocks (stage, decimate, pass 1, scan, pass 2, tp convert (+ f32 remainder), tp product,
// tile tail), [15] waves counted; reset != 0 clears the totals after reading
extern "C" int ss_debug_td_prof(unsigned long long *out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(ssk::g_td_prof), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(ssk::g_td_prof), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
