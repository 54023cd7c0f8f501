// ss_host.h — what the host-side translation units of the library share (never installed): the HIP error channel,
// RAII device buffers, the per-device caches of constant tables, per-thread scratch, and the handle type.
//   ss_host.cpp      caches, device selection, status strings, table inspection
//   ss_analyzer.cpp  the Analyzer mirror (one entry point per Rust method, analyzer.rs:29-183)
//   ss_ingest.cpp    RIFF/WAVE header walk and PCM conversion
//   ss_batch.cpp     the batch extension (BASELINE configs 3-5) and the render-side reductions (N3)
//   ss_session.cpp   the tick drivers (N1)
#pragma once
#include "../../include/soundscope_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "ss_internal.h"
#include "ss_kernels.h"
#include "ss_tables.h"

namespace ssh {

// text of the last HIP error on this thread (ss_last_device_error)
SS_HIDDEN std::string &last_error();
SS_HIDDEN bool hip_ok(hipError_t e, const char *what);
#define HIPCHK(expr)                                        \
    do {                                                    \
        if (!ssh::hip_ok((expr), #expr)) return SS_ERR_DEVICE; \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    hipError_t alloc(size_t count)
    {
        release();
        if (!count) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t ensure(size_t count) { return count <= n ? hipSuccess : alloc(count); }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
    hipError_t upload(const std::vector<T> &h)
    {
        hipError_t e = h.size() == n ? hipSuccess : alloc(h.size());      // (same size: the allocation is kept — a free and a malloc are 30-250 us)
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// ---- per-process caches of device-resident constant tables ------------------
struct FftTables {
    size_t n = 0;
    std::vector<float> window_host;
    DevBuf<float> window, half_window;
    DevBuf<float2> tw_n, tw_256;
    const float2 *core_tw4096 = nullptr, *core_tw256 = nullptr;   // n == 16384: tables of the 4096-point core
};

struct BinTables {
    size_t first = 0, count = 0;
    std::vector<double> freq, pink, chart_x;
    DevBuf<float> pink_dev;
    DevBuf<float> offpink4096_dev;      // db_offset(4096) + pink, for the N = 4096 kernels
    DevBuf<float> off4096_dev;          // db_offset(4096) alone (ss_get_fft adds the pink compensation in f64 on the host, analyzer.rs:82)
};

struct TdTables {
    ssk::TdConst host;
    DevBuf<ssk::TdConst> dev;
};

// One context per HIP device: device pointers are only valid on the device that allocated them, and
// hipSetDevice is per thread, so the caches are looked up by the device current at the call.
struct Ctx {
    std::mutex mu;
    std::map<size_t, std::unique_ptr<FftTables>> fft;
    std::map<std::pair<uint32_t, size_t>, std::unique_ptr<BinTables>> bins;
    std::map<std::pair<uint32_t, uint32_t>, std::unique_ptr<TdTables>> td;   // (rate, factor | channels << 8)
    DevBuf<double> hist_energies, hist_bounds;
    // idle streams of handles and batches that were destroyed (stream_acquire / stream_release)
    std::vector<hipStream_t> idle_streams;
};

// the table cache of the calling thread's current device
SS_HIDDEN Ctx &ctx();
// Streams of handles, sessions and batches come from a per-device pool: hipStreamCreateWithFlags costs 1.7-18 ms and
// hipStreamDestroy 1.5 ms on this stack (rocprofv3 --hip-trace over six session opens: 52 of their 70 ms, tools/probe_open_hipapi.sh)
// — a file open made two of each.  stream_release takes an IDLE stream (the caller has synchronised it).
SS_HIDDEN hipError_t stream_acquire(hipStream_t *out);
SS_HIDDEN void stream_release(hipStream_t s);
SS_HIDDEN int current_device();

// Scratch of the handle-less entry points (ss_get_waveform, ss_mid_side, ss_pcm_decode): one set per calling
// thread and device, so concurrent callers never serialise on a shared buffer.
struct Scratch {
    DevBuf<float> in, out;
    DevBuf<unsigned char> raw;
    hipStream_t stream = nullptr;
    ~Scratch() { if (stream) (void)hipStreamDestroy(stream); }
};
SS_HIDDEN Scratch &scratch();

// makes `device` current for the scope of one entry point (handles are bound to the device they were created on)
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device)
    {
        if (device < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define SS_ON_DEVICE(obj) DeviceScope device_scope_((obj) ? (obj)->device : -1)

SS_HIDDEN int probe_devices();
SS_HIDDEN int require_device();
SS_HIDDEN int get_fft_tables(size_t n, FftTables **out);
SS_HIDDEN int get_bin_tables(uint32_t rate, size_t n, BinTables **out);
SS_HIDDEN int get_td_tables(uint32_t rate, int factor, uint32_t channels, TdTables **out);
SS_HIDDEN int get_hist_tables(const double **energies, const double **bounds);
inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }

// run-in of a time segment that starts from a zero filter state, in 100 ms sub-blocks.  What the missing history would
// have contributed to the OUTPUT is the tail of the K-weighting impulse response: the slowest pole pair (38 Hz high-pass,
// |p| = 0.99502 at 48 kHz, a near-double pole) decays by e^-23.9 per sub-block, times a polynomial factor ~ (1 + 24 n).
// Measured on adversarial material (DC offset plus a strong 7 Hz component, every segment one sub-block long): sub-block
// energies within 2.7e-10 of a sequential f64 filter with a one-sub-block run-in, 1.6e-10 with two — both at the
// arithmetic noise of the recurrence on such material (tools: tests/test_gpu_bench_shapes.py
// ::test_segmented_run_in_on_dc_offset_material pins the histograms).  One sub-block it is: the run-in is redundant work
// (config 5: 4 instead of 5 sub-blocks per 3-sub-block segment).
#ifndef SS_TD_WARM_SUB
#define SS_TD_WARM_SUB 1
#endif
constexpr uint32_t kTdWarmSub = SS_TD_WARM_SUB;
// sub-blocks at the head of a time segment > 0 that the fix-up launch re-runs from the exact incoming state.  Behind them the main
// launch's trajectory (zero state at the segment's start) differs from the true one by A^n s: e^-48 (1 + 48) ~ 7e-20 of the state
// after 0.2 s — on the DC-offset torture material (state 4e4 x the offset) 1e-13 of the filtered signal, under the 2e-11 at which any
// two orders of evaluating this recurrence differ (measured: exact whole-stream path vs one-segment path, profiles/r05_ab_td_handover.txt)
constexpr uint32_t kTdFixSub = 2;

SS_HIDDEN int meter_args_ok(uint32_t channels, uint32_t rate);

}  // namespace ssh

// ============================================================================
//  handle
// ============================================================================
struct ss_analyzer {
    int device = 0;            // the HIP device this handle's buffers live on
    uint32_t channels = 0, rate = 0;
    uint32_t meter_rate = 0;   // the rate the current meter was built for (rate sticks on a failed configure, the meter does not change)
    int tp_cfg = 0;            // 0 = crate rule
    int tp_factor = 0;         // effective
    int tp_arith = SS_TP_ARITH_F32;   // ss_analyzer_set_true_peak_arith
    int tp_cfg_applied = 0;    // the tp_cfg the current meter was built with
    bool meter_ok = false;
    hipStream_t stream = nullptr;
    ssh::TdTables *td = nullptr;
    ssh::DevBuf<ssk::TdState> state;
    ssh::DevBuf<uint64_t> hist;          // 2 x 1000
    ssh::DevBuf<double> sub;             // kSubCap x C
    ssh::DevBuf<double> ring;            // ring_frames x C
    ssh::DevBuf<double> weights;
    ssh::DevBuf<uint32_t> counts;
    ssh::DevBuf<double> out2, ring_scratch;
    ssh::DevBuf<float> in;
    // get_fft(&self): the handle is const at the boundary; the payload of the last error (ss_get_fft_error_values) and the
    // page-locked mailboxes below are the call's own scratch
    mutable float fft_err_a = 0.0f, fft_err_b = 0.0f;
    uint64_t ring_frames = 0;
    uint64_t frames_fed = 0;
    static constexpr uint32_t kSubCap = 96;
    // The reference's render loop asks for the integrated loudness, the loudness range and the true peak on EVERY frame (tui.rs:917,
    // :950, :969; a frame every 8 ms, a tick every 21): a reading is taken from the device once per state of the meter —
    // `change_count` moves with every feed, reset and re-configuration — and handed out from here until the state moves again.
    uint64_t change_count = 1;
    uint64_t eval_stamp = 0, peaks_stamp = 0;               // (both readings are taken together: one wait)
    double eval_cache[2] = {0.0, 0.0};                       // (integrated, range) at eval_stamp
    float peaks_cache[2 * ssk::kMaxChannels] = {};          // sample peaks | true peaks at peaks_stamp
    // Small calls — a tick through the Analyzer API: get_fft x 2, add_samples, get_shortterm_lufs on 16384 samples — move no
    // data with pageable copy commands and read-backs (a pageable 64 KB hipMemcpyAsync blocks the caller and costs more than the
    // kernel behind it): the samples are copied by the host into page-locked memory, from where ONE DMA takes them to HBM (the
    // kernels read their input in small pieces: in place over PCIe that cost them 10-14 us), and results are written by the
    // kernels into page-locked memory.  Two input buffers, so that add_samples returns behind its launches (an event says when
    // a buffer's copy has left it); every call that waits for the stream frees both.
    static constexpr size_t kPinFloats = 32768;
    float *pin_in[2] = {nullptr, nullptr}, *pin_in_dev[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[3] = {nullptr, nullptr, nullptr};      // [0], [1]: an input buffer's kernel has read it; [2]: a ring reading is there
    bool pin_busy[2] = {false, false};
    int pin_next = 0;
    float *pin_out = nullptr, *pin_out_dev = nullptr;       // kPinFloats / 2 + 1 dB values
    double *pin_d = nullptr, *pin_d_dev = nullptr;          // a getter's pair of doubles
    double *pin_d_pending = nullptr;                        // the same while pin_ready has not finished (pin_d set = all of them ready)
    float *pin_peaks = nullptr;                              // 2 * kMaxChannels floats: the state's peaks, copied there
    double *pin_eval = nullptr, *pin_eval_dev = nullptr;    // (integrated, range) of the state the readings were last asked for
    float *pin_peaks_dev = nullptr;
    uint32_t *pin_flag = nullptr, *pin_flag_dev = nullptr;  // the readings launch stores its number there, last
    uint32_t readings_seq = 0;                               // readings launches so far
    uint64_t prefetch_stamp = 0;                             // change_count the launch in flight is of (0: none)
};

namespace ssh {
// ss_analyzer.cpp: pieces of the handle the batch one-shot and the tick drivers reuse
SS_HIDDEN int handle_reset(ss_analyzer *h);
// the handle's page-locked mailboxes (allocated at first use); pin_acquire: an input buffer no kernel is still reading
SS_HIDDEN int pin_ready(ss_analyzer *h);
SS_HIDDEN int pin_acquire(ss_analyzer *h, int *idx);
SS_HIDDEN void pin_all_free(ss_analyzer *h);           // behind a hipStreamSynchronize of h->stream
// enqueue (no wait) the readings the reference's render loop asks for on every frame — integrated loudness and range, every
// channel's peaks — behind whatever has just changed the meter's state: the getters then find them waiting
SS_HIDDEN int prefetch_readings(ss_analyzer *h, bool on_demand = false);      // on_demand: a getter asking (never switched off)
// the same riding a gating launch that is about to be enqueued on h->stream (k_finalize_stream takes the readings behind its
// histogram updates): fills the launch's fields and marks the readings as on their way
SS_HIDDEN int attach_readings(ss_analyzer *h, ssk::FinalizeParams *gating);
// add_frames_f32 on the handle's meter; on_device: `samples` already lives in HBM (nothing is copied or waited for)
// `deferred`: a single-piece device-resident call hands the gating launch (k_finalize_stream) of its new sub-blocks back to the
// caller instead of enqueueing it (n_streams != 0: launch it with ssk::launch_finalize on h->stream before anything else reads the
// histograms) — the tick drivers put the short-term reading in front of it
// `tick`: what a tick wants to ride the launch of a single-piece device-resident call (ssk::launch_time_domain / k_tick): its
// spectrum, and the short-term reading of the window that ends with the call; `fused` tells whether both did — if not, neither
// was launched
struct TickExtras {
    const ssk::FftBatchParams *fft = nullptr;   // nullptr: no spectrum this tick (then nothing is fused)
    double *shortterm_out = nullptr;            // (energy, loudness), device-visible; nullptr: no reading wanted
    bool fused = false;                         // out: the spectrum rode the loudness call's launch (k_tick)
    bool st_fused = false;                      // out: ... and so did the short-term reading (its window parameters were set)
};
SS_HIDDEN int add_samples_impl(ss_analyzer *h, const float *samples, size_t n, bool on_device, ssk::FinalizeParams *deferred = nullptr,
                               TickExtras *tick = nullptr);
SS_HIDDEN int ring_loudness_enqueue(ss_analyzer *h, uint64_t frames, double *out2_dev = nullptr);   // out2_dev: where (energy, loudness) go — default the handle's device pair; the tick drivers pass mapped pinned memory
SS_HIDDEN void waveform_shape(size_t n, double waveform_window, size_t *window_out, size_t *bins_out);
// ss_batch.cpp: Analyzer::calculate_integrated_lufs on a host or device-resident buffer (a one-stream batch pass)
SS_HIDDEN int integrated_oneshot(uint32_t rate, uint32_t channels, const float *samples, size_t n, bool on_device, double *out);
}  // namespace ssh
