// ss_td_f2.hip: the time-domain kernels of 2x true-peak oversampling (96 kHz up to below 192 kHz: BASELINE config 5).  See ss_td_impl.h.
#include "ss_td_impl.h"

namespace ssk {
SS_TD_DEFINE_FACTOR(2)
}  // namespace ssk
