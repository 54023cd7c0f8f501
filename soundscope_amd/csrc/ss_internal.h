// ss_internal.h — what the library's own translation units share beyond the public C ABI (never installed).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct ss_batch;

namespace ssi {
#define SS_HIDDEN __attribute__((visibility("hidden")))
// the batch's device-resident corpus histograms (2 x 1000 u64: block ++ short-term), its stream and its device
SS_HIDDEN void *batch_corpus_device(ss_batch *b);
SS_HIDDEN hipStream_t batch_stream(ss_batch *b);
SS_HIDDEN int batch_device(const ss_batch *b);
// true once this pass's corpus histograms have been all-reduced (cleared by ss_batch_run): a second all-reduce of the same
// pass would multiply every bin by the world size
SS_HIDDEN bool &batch_corpus_reduced(ss_batch *b);
// text ss_last_device_error() returns on this thread
SS_HIDDEN void set_last_error(const std::string &text);
}  // namespace ssi
