// ss_td_f0.hip: the time-domain kernels without true-peak oversampling (192 kHz and up, loudness-only batches, the second launch
// of the exact segment hand-over).  See ss_td_impl.h.
#include "ss_td_impl.h"

namespace ssk {
SS_TD_DEFINE_FACTOR(0)
}  // namespace ssk
