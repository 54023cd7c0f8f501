// ss_api.cpp — C ABI of include/soundscope_hip.h on top of the gfx950 kernels.
// Host logic only: argument validation with the reference's error order,
// device-resident state management, table caches, launches.  No CPU compute path.
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

namespace {

thread_local std::string g_last_err;

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    g_last_err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIPCHK(expr)                                   \
    do {                                               \
        if (!hip_ok((expr), #expr)) return SS_ERR_DEVICE; \
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
        hipError_t e = alloc(h.size());
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
};

struct Process {
    std::mutex mu;
    bool probed = false;
    int n_devices = 0;
    std::map<int, std::unique_ptr<Ctx>> per_device;
};

Process &process()
{
    static Process p;
    return p;
}

int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}

// the table cache of the calling thread's current device
Ctx &ctx()
{
    Process &p = process();
    const int d = current_device();
    std::lock_guard<std::mutex> lk(p.mu);
    auto &slot = p.per_device[d];
    if (!slot) slot = std::make_unique<Ctx>();
    return *slot;
}

// Scratch of the handle-less entry points (ss_get_waveform, ss_mid_side, ss_pcm_decode): one set per calling
// thread and device, so concurrent callers never serialise on a shared buffer.
struct Scratch {
    DevBuf<float> in, out;
    DevBuf<unsigned char> raw;
    hipStream_t stream = nullptr;
    ~Scratch() { if (stream) (void)hipStreamDestroy(stream); }
};

Scratch &scratch()
{
    thread_local std::map<int, std::unique_ptr<Scratch>> per_device;
    auto &slot = per_device[current_device()];
    if (!slot) slot = std::make_unique<Scratch>();
    return *slot;
}

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

int probe_devices()
{
    Process &c = process();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.probed) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) { g_last_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); n = 0; }
        c.n_devices = n;
        c.probed = true;
    }
    return c.n_devices;
}

int require_device()
{
    if (probe_devices() <= 0) {
        if (g_last_err.empty()) g_last_err = "no HIP device visible";
        return SS_ERR_DEVICE;
    }
    return SS_OK;
}

int get_fft_tables(size_t n, FftTables **out)
{
    Ctx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.fft.find(n);
    if (it != c.fft.end()) { *out = it->second.get(); return SS_OK; }
    auto t = std::make_unique<FftTables>();
    t->n = n;
    t->window_host = sst::hann_window(n);
    std::vector<float> half(n);
    for (size_t i = 0; i < n; i++) half[i] = 0.5f * t->window_host[i];
    HIPCHK(t->window.upload(t->window_host));
    HIPCHK(t->half_window.upload(half));
    std::vector<float> tw;
    // the 4096 kernel indexes W_N^(t*ka) up to 255*15; the generic kernel k < N/2
    sst::twiddles(n, n == 4096 ? n : (n / 2 ? n / 2 : 1), tw);
    std::vector<float2> tw2(tw.size() / 2);
    for (size_t i = 0; i < tw2.size(); i++) tw2[i] = make_float2(tw[2 * i], tw[2 * i + 1]);
    HIPCHK(t->tw_n.upload(tw2));
    if (n == 4096) {
        sst::twiddles(256, 256, tw);
        std::vector<float2> t256(256);
        for (size_t i = 0; i < 256; i++) t256[i] = make_float2(tw[2 * i], tw[2 * i + 1]);
        HIPCHK(t->tw_256.upload(t256));
    }
    if (n == 16384) {
        // (the mutex is not recursive: build the core tables inline)
        auto it4 = c.fft.find(4096);
        if (it4 == c.fft.end()) {
            auto t4 = std::make_unique<FftTables>();
            t4->n = 4096;
            t4->window_host = sst::hann_window(4096);
            std::vector<float> half4(4096);
            for (size_t i = 0; i < 4096; i++) half4[i] = 0.5f * t4->window_host[i];
            HIPCHK(t4->window.upload(t4->window_host));
            HIPCHK(t4->half_window.upload(half4));
            std::vector<float> tw4;
            sst::twiddles(4096, 4096, tw4);
            std::vector<float2> v4(4096);
            for (size_t i = 0; i < 4096; i++) v4[i] = make_float2(tw4[2 * i], tw4[2 * i + 1]);
            HIPCHK(t4->tw_n.upload(v4));
            sst::twiddles(256, 256, tw4);
            std::vector<float2> v256(256);
            for (size_t i = 0; i < 256; i++) v256[i] = make_float2(tw4[2 * i], tw4[2 * i + 1]);
            HIPCHK(t4->tw_256.upload(v256));
            it4 = c.fft.emplace(4096, std::move(t4)).first;
        }
        t->core_tw4096 = it4->second->tw_n.p;
        t->core_tw256 = it4->second->tw_256.p;
    }
    *out = t.get();
    c.fft[n] = std::move(t);
    return SS_OK;
}

int get_bin_tables(uint32_t rate, size_t n, BinTables **out)
{
    Ctx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    auto key = std::make_pair(rate, n);
    auto it = c.bins.find(key);
    if (it != c.bins.end()) { *out = it->second.get(); return SS_OK; }
    auto t = std::make_unique<BinTables>();
    t->count = sst::fft_bins(rate, n, &t->first);
    sst::bin_tables(rate, n, t->freq, t->pink, t->chart_x);
    std::vector<float> pf((t->count + 3) & ~(size_t)3, 0.0f);    // padded to the output row stride
    for (size_t i = 0; i < t->count; i++) pf[i] = (float)t->pink[i];
    HIPCHK(t->pink_dev.upload(pf));
    if (n == 4096) {
        const float off = (float)(10.0 * std::log10(4.0 / (4096.0 * 4096.0)));
        std::vector<float> op(pf.size());
        for (size_t i = 0; i < pf.size(); i++) op[i] = off + pf[i];
        HIPCHK(t->offpink4096_dev.upload(op));
    }
    *out = t.get();
    c.bins[key] = std::move(t);
    return SS_OK;
}

int get_td_tables(uint32_t rate, int factor, uint32_t channels, TdTables **out)
{
    Ctx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    auto key = std::make_pair(rate, (uint32_t)factor | (channels << 8));
    auto it = c.td.find(key);
    if (it != c.td.end()) { *out = it->second.get(); return SS_OK; }
    auto t = std::make_unique<TdTables>();
    ssk::TdConst &k = t->host;
    std::memset(&k, 0, sizeof k);
    sst::kweight_design((double)rate, k.b, k.a);
    for (int s = 0; s < 8; s++) sst::kweight_transition_pow(k.a, (uint64_t)ssk::td_chunk_frames(channels, (rate + 5) / 10) << s, k.m_pow[s]);
    k.tp_factor = factor;
    k.tp_len = 0;
    if (factor) {
        std::vector<std::vector<sst::PolyTap>> ph; int delay;
        sst::true_peak_design(factor, ph, &delay);
        // branch 0 is the identity tap (x[n - 6] * 1.0 / x[n - 12] * 1.0): it can only
        // reproduce the sample peak, which true_peak() maxes in anyway.
        k.tp_len = factor == 4 ? 12 : 24;
        for (int f = 1; f < factor; f++)
            for (const auto &tap : ph[f]) k.tp[f - 1][tap.delay] = tap.coeff;
    }
    k.s100 = (rate + 5) / 10;
    std::vector<ssk::TdConst> v(1, k);
    HIPCHK(t->dev.upload(v));
    *out = t.get();
    c.td[key] = std::move(t);
    return SS_OK;
}

int get_hist_tables(const double **energies, const double **bounds)
{
    Ctx &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.hist_energies.p) {
        std::vector<double> e(sst::kHistBins), b(sst::kHistBins + 1);
        sst::histogram_tables(e.data(), b.data());
        HIPCHK(c.hist_energies.upload(e));
        HIPCHK(c.hist_bounds.upload(b));
    }
    *energies = c.hist_energies.p;
    *bounds = c.hist_bounds.p;
    return SS_OK;
}

bool is_pow2(size_t n) { return n && !(n & (n - 1)); }

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

int meter_args_ok(uint32_t channels, uint32_t rate)
{
    // EbuR128::new: channels == 0 || > 64, rate < 16 || > 2_822_400 -> Error::NoMem
    if (channels == 0 || channels > 64) return SS_ERR_NOMEM;
    if (rate < 16 || rate > 2822400) return SS_ERR_NOMEM;
    return SS_OK;
}

}  // namespace

// ============================================================================
//  handle
// ============================================================================
struct ss_analyzer {
    int device = 0;            // the HIP device this handle's buffers live on
    uint32_t channels = 0, rate = 0;
    uint32_t meter_rate = 0;   // the rate the current meter was built for (rate sticks on a failed configure, the meter does not change)
    int tp_cfg = 0;            // 0 = crate rule
    int tp_factor = 0;         // effective
    int tp_cfg_applied = 0;    // the tp_cfg the current meter was built with
    bool meter_ok = false;
    hipStream_t stream = nullptr;
    TdTables *td = nullptr;
    DevBuf<ssk::TdState> state;
    DevBuf<uint64_t> hist;          // 2 x 1000
    DevBuf<double> sub;             // kSubCap x C
    DevBuf<double> ring;            // ring_frames x C
    DevBuf<double> weights;
    DevBuf<uint32_t> counts;
    DevBuf<double> out2, ring_scratch;
    DevBuf<float> in, fft_out;
    uint64_t ring_frames = 0;
    uint64_t frames_fed = 0;
    static constexpr uint32_t kSubCap = 96;
};

namespace {

// EbuR128::new + assignment (analyzer.rs:49-53): the new meter replaces the old one only when every fallible step
// has succeeded — on failure the handle keeps its previous meter (only sample_rate has changed by then).
int handle_make_meter(ss_analyzer *h, uint32_t channels, uint32_t rate)
{
    int rc = meter_args_ok(channels, rate);
    if (rc) return rc;
    const int tp_factor = h->tp_cfg ? h->tp_cfg : sst::true_peak_factor_for_rate(rate);
    TdTables *td = nullptr;
    rc = get_td_tables(rate, tp_factor, channels, &td);
    if (rc) return rc;
    const uint64_t s100 = (rate + 5) / 10;
    uint64_t ring_frames = (uint64_t)rate * 3000 / 1000;
    if (ring_frames % s100) ring_frames += s100 - ring_frames % s100;
    DevBuf<ssk::TdState> state;
    DevBuf<uint64_t> hist;
    DevBuf<double> sub, ring, weights, out2, ring_scratch;
    DevBuf<uint32_t> counts;
    HIPCHK(state.alloc(1));
    HIPCHK(hist.alloc(2 * sst::kHistBins));
    HIPCHK(sub.alloc((size_t)ss_analyzer::kSubCap * channels));
    HIPCHK(ring.alloc(ring_frames * channels));
    HIPCHK(counts.alloc(2));
    HIPCHK(out2.alloc(2));
    HIPCHK(ring_scratch.alloc(128));
    std::vector<double> w(channels);
    sst::channel_weights(channels, w.data());
    HIPCHK(weights.upload(w));
    // commit
    h->channels = channels; h->meter_rate = rate; h->tp_factor = tp_factor; h->tp_cfg_applied = h->tp_cfg; h->td = td; h->ring_frames = ring_frames;
    h->state.swap(state); h->hist.swap(hist); h->sub.swap(sub); h->ring.swap(ring); h->counts.swap(counts);
    h->out2.swap(out2); h->ring_scratch.swap(ring_scratch); h->weights.swap(weights);
    h->meter_ok = true;
    return SS_OK;
}

int handle_reset(ss_analyzer *h)
{
    if (!h->meter_ok) return SS_OK;
    if (h->tp_cfg != h->tp_cfg_applied) {        // ss_analyzer_set_true_peak_factor since the meter was built
        HIPCHK(hipStreamSynchronize(h->stream));
        int rc = handle_make_meter(h, h->channels, h->meter_rate);
        if (rc) return rc;
    }
    HIPCHK(hipMemsetAsync(h->state.p, 0, sizeof(ssk::TdState), h->stream));
    HIPCHK(hipMemsetAsync(h->hist.p, 0, h->hist.n * sizeof(uint64_t), h->stream));
    HIPCHK(hipMemsetAsync(h->sub.p, 0, h->sub.n * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->ring.p, 0, h->ring.n * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->counts.p, 0, 2 * sizeof(uint32_t), h->stream));
    h->frames_fed = 0;
    return SS_OK;
}

}  // namespace

extern "C" {

const char *ss_status_string(int s)
{
    switch (s) {
        case SS_OK: return "ok";
        case SS_ERR_NOMEM: return "ebur128: NoMem";
        case SS_ERR_INVALID_MODE: return "ebur128: InvalidMode";
        case SS_ERR_INVALID_CHANNEL: return "ebur128: InvalidChannelIndex";
        case SS_ERR_TOO_FEW_SAMPLES: return "spectrum-analyzer: TooFewSamples";
        case SS_ERR_NAN: return "spectrum-analyzer: NaNValuesNotSupported";
        case SS_ERR_INFINITY: return "spectrum-analyzer: InfinityValuesNotSupported";
        case SS_ERR_NOT_POW2: return "spectrum-analyzer: SamplesLengthNotAPowerOfTwo";
        case SS_ERR_FREQ_LIMIT: return "spectrum-analyzer: InvalidFrequencyLimit";
        case SS_ERR_SCALING: return "spectrum-analyzer: ScalingError";
        case SS_ERR_CAPACITY: return "output buffer too small";
        case SS_ERR_UNSUPPORTED: return "unsupported configuration";
        case SS_ERR_INVALID_ARG: return "invalid argument";
        case SS_ERR_DEVICE: return "HIP device error";
        default: return "unknown status";
    }
}

int ss_abi_version(void) { return SS_ABI_VERSION; }
int ss_device_count(void) { return probe_devices(); }
int ss_set_device(int device)
{
    if (require_device()) return SS_ERR_DEVICE;
    HIPCHK(hipSetDevice(device));
    return SS_OK;
}
int ss_device_synchronize(void)
{
    if (require_device()) return SS_ERR_DEVICE;
    HIPCHK(hipDeviceSynchronize());
    return SS_OK;
}
const char *ss_last_device_error(void) { return g_last_err.c_str(); }

// ---- inspection of the host-designed tables (no device involved) -------------------------------------------------
int ss_inspect_kweight(uint32_t rate, double b5[5], double a5[5])
{
    if (rate < 16 || rate > 2822400) return SS_ERR_INVALID_ARG;
    double b[5], a[5];
    sst::kweight_design((double)rate, b, a);
    if (b5) std::memcpy(b5, b, sizeof b);
    if (a5) std::memcpy(a5, a, sizeof a);
    return SS_OK;
}

int ss_inspect_true_peak(int factor, float *taps, uint32_t cap, uint32_t *len)
{
    if (factor != 2 && factor != 4) return SS_ERR_INVALID_ARG;
    std::vector<std::vector<sst::PolyTap>> ph; int delay = 0;
    sst::true_peak_design(factor, ph, &delay);
    const uint32_t n = factor == 4 ? 12u : 24u;            // taps per branch as the kernels hold them (TdConst::tp)
    if (len) *len = n;
    if (taps) {
        if (cap < (uint32_t)(factor - 1) * n) return SS_ERR_CAPACITY;
        std::memset(taps, 0, sizeof(float) * (size_t)(factor - 1) * n);
        for (int f = 1; f < factor; f++)
            for (const auto &tap : ph[f]) if ((uint32_t)tap.delay < n) taps[(size_t)(f - 1) * n + tap.delay] = tap.coeff;
    }
    return SS_OK;
}

int ss_inspect_hann(uint32_t n, float *w)
{
    if (!w) return SS_ERR_INVALID_ARG;
    const std::vector<float> h = sst::hann_window(n);
    if (n) std::memcpy(w, h.data(), sizeof(float) * n);
    return SS_OK;
}

int ss_inspect_bins(uint32_t rate, uint32_t n, uint32_t *first_bin, uint32_t *n_bins)
{
    size_t first = 0;
    const size_t cnt = sst::fft_bins(rate, n, &first);
    if (first_bin) *first_bin = (uint32_t)first;
    if (n_bins) *n_bins = (uint32_t)cnt;
    return SS_OK;
}

int ss_inspect_histogram(double energies1000[1000], double bounds1001[1001])
{
    double e[sst::kHistBins], b[sst::kHistBins + 1];
    sst::histogram_tables(e, b);
    if (energies1000) std::memcpy(energies1000, e, sizeof e);
    if (bounds1001) std::memcpy(bounds1001, b, sizeof b);
    return SS_OK;
}

int ss_analyzer_create(uint32_t channels, uint32_t rate, ss_analyzer **out)
{
    if (!out) return SS_ERR_INVALID_ARG;
    *out = nullptr;
    if (require_device()) return SS_ERR_DEVICE;
    // destroyed (stream included) on every early return
    std::unique_ptr<ss_analyzer, decltype(&ss_analyzer_destroy)> h(new ss_analyzer(), &ss_analyzer_destroy);
    h->device = current_device();
    h->rate = rate;
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(h->in.alloc(32768));
    HIPCHK(h->fft_out.alloc(16385));
    int rc = handle_make_meter(h.get(), channels, rate);
    if (rc) return rc;
    rc = handle_reset(h.get());
    if (rc) return rc;
    *out = h.release();
    return SS_OK;
}

void ss_analyzer_destroy(ss_analyzer *h)
{
    SS_ON_DEVICE(h);
    if (!h) return;
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}

int ss_analyzer_configure(ss_analyzer *h, uint32_t channels, uint32_t rate)
{
    SS_ON_DEVICE(h);
    if (!h) return SS_ERR_INVALID_ARG;
    h->rate = rate;                         // analyzer.rs:50: before the fallible call
    HIPCHK(hipStreamSynchronize(h->stream));
    int rc = handle_make_meter(h, channels, rate);
    if (rc) return rc;
    return handle_reset(h);
}

int ss_analyzer_set_true_peak_factor(ss_analyzer *h, int factor)
{
    SS_ON_DEVICE(h);
    if (!h || (factor != 0 && factor != 2 && factor != 4)) return SS_ERR_INVALID_ARG;
    h->tp_cfg = factor;
    return SS_OK;
}

int ss_get_fft(const ss_analyzer *hc, const float *samples, size_t n,
               double *out_xy, size_t cap_pairs, size_t *out_n)
{
    SS_ON_DEVICE(hc);
    ss_analyzer *h = const_cast<ss_analyzer *>(hc);
    if (out_n) *out_n = 0;
    if (!h || (!samples && n) || !out_xy) return SS_ERR_INVALID_ARG;
    // samples_fft_to_spectrum checks, in the crate's order, applied to the
    // windowed samples (hann_window runs first, analyzer.rs:57)
    if (n < 2) return SS_ERR_TOO_FEW_SAMPLES;
    const bool pow2 = is_pow2(n);
    if (n > 32768 && pow2) return SS_ERR_UNSUPPORTED;
    {
        // w[i] == 0 turns an infinite sample into NaN (0 * inf); only the first few
        // window entries can be exactly zero
        bool any_nan = false, any_inf = false;
        const std::vector<float> *win = nullptr;
        std::vector<float> win_local;
        for (size_t i = 0; i < n; i++) {
            const float x = samples[i];
            if (std::isnan(x)) { any_nan = true; continue; }
            if (!std::isinf(x)) continue;
            if (!win) {
                if (pow2) {
                    FftTables *wt = nullptr;
                    int rc = get_fft_tables(n, &wt);
                    if (rc) return rc;
                    win = &wt->window_host;
                } else {
                    win_local = sst::hann_window(n);
                    win = &win_local;
                }
            }
            if ((*win)[i] == 0.0f) any_nan = true; else any_inf = true;
        }
        if (any_nan) return SS_ERR_NAN;
        if (any_inf) return SS_ERR_INFINITY;
    }
    if (!pow2) return SS_ERR_NOT_POW2;
    if (20000.0f > (float)h->rate / 2.0f) return SS_ERR_FREQ_LIMIT;

    FftTables *ft; BinTables *bt;
    int rc = get_fft_tables(n, &ft);
    if (rc) return rc;
    rc = get_bin_tables(h->rate, n, &bt);
    if (rc) return rc;
    if (bt->count > cap_pairs) return SS_ERR_CAPACITY;
    if (bt->count == 0) return SS_OK;

    HIPCHK(hipMemcpyAsync(h->in.p, samples, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    ssk::FftBatchParams p{};
    p.pcm = h->in.p; p.out = h->fft_out.p;
    p.window = ft->window.p; p.half_window = ft->half_window.p;
    p.tw_n = ft->tw_n.p; p.tw_256 = ft->tw_256.p; p.pink = nullptr;
    p.frames_per_stream = n; p.first_start = 0; p.n_streams = 1; p.channels = 1;
    p.n_windows = 1; p.hop = 0; p.n = (uint32_t)n;
    p.first_bin = (uint32_t)bt->first; p.n_bins = (uint32_t)bt->count; p.bin_stride = p.n_bins; p.windows_per_block = 1;
    p.db_offset = (float)(20.0 * std::log10(4.0 / (double)n));
    if (n == 16384) {
        p.tw_core = ft->core_tw4096; p.tw_256 = ft->core_tw256;
        HIPCHK(ssk::launch_fft16k(p, 0, h->stream));
    } else {
        HIPCHK(ssk::launch_fft_generic(p, 0, h->stream));
    }
    std::vector<float> db(bt->count);
    HIPCHK(hipMemcpyAsync(db.data(), h->fft_out.p, bt->count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < bt->count; i++)
        if (std::isnan(db[i]) || std::isinf(db[i])) return SS_ERR_SCALING;
    // analyzer.rs:75-102 in f64: + pink compensation, log-x chart position
    for (size_t i = 0; i < bt->count; i++) {
        out_xy[2 * i] = bt->chart_x[i];
        out_xy[2 * i + 1] = (double)db[i] + bt->pink[i];
    }
    if (out_n) *out_n = bt->count;
    return SS_OK;
}

// get_waveform's shape (analyzer.rs:108-118): W = (window_s * 1000.) as usize, and the number of
// bins whose start floor(i * spp) is still inside the buffer
static void waveform_shape(size_t n, double waveform_window, size_t *window_out, size_t *bins_out)
{
    const double wd = waveform_window * 1000.0;
    // Rust `as usize`: saturating, NaN -> 0
    size_t window = (wd != wd || wd <= 0.0) ? 0 : (wd >= 1.8446744073709552e19 ? SIZE_MAX : (size_t)wd);
    size_t bins = window;
    if (window != 0 && n != 0 && window > n) {
        // first i with floor(i*spp) >= n; start = floor(i * spp) is monotone
        const double spp = (double)n / (double)window;
        size_t lo = 0, hi = window;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if ((size_t)((double)mid * spp) >= n) hi = mid; else lo = mid + 1;
        }
        bins = lo;
    }
    if (window == 0 || n == 0) bins = 0;
    *window_out = window; *bins_out = bins;
}

int ss_get_waveform(const float *samples, size_t n, double waveform_window,
                    double *out_xy, size_t cap_pairs, size_t *out_n)
{
    if (out_n) *out_n = 0;
    if ((!samples && n) || (!out_xy && cap_pairs)) return SS_ERR_INVALID_ARG;
    if (require_device()) return SS_ERR_DEVICE;
    size_t window, bins;
    waveform_shape(n, waveform_window, &window, &bins);
    if (window == 0 || n == 0) return SS_OK;        // loop body never pushes a point
    if (bins > 0xFFFFFFFFull) return SS_ERR_UNSUPPORTED;
    if (2 * bins > cap_pairs) return SS_ERR_CAPACITY;
    Scratch &c = scratch();
    if (!c.stream) HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIPCHK(c.in.ensure(n));
    HIPCHK(c.out.ensure(2 * bins));
    HIPCHK(hipMemcpyAsync(c.in.p, samples, n * sizeof(float), hipMemcpyHostToDevice, c.stream));
    ssk::WaveParams p{};
    p.pcm = c.in.p; p.stream_stride = n; p.n_samples = n; p.n_streams = 1;
    p.window = (uint32_t)window; p.out = c.out.p; p.out_stride = 2 * bins;
    if (window > 0xFFFFFFFFull) return SS_ERR_UNSUPPORTED;
    HIPCHK(ssk::launch_waveform(p, c.stream));
    std::vector<float> mm(2 * bins);
    HIPCHK(hipMemcpyAsync(mm.data(), c.out.p, 2 * bins * sizeof(float), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    for (size_t i = 0; i < bins; i++) {
        out_xy[4 * i + 0] = (double)i; out_xy[4 * i + 1] = (double)mm[2 * i];
        out_xy[4 * i + 2] = (double)i; out_xy[4 * i + 3] = (double)mm[2 * i + 1];
    }
    if (out_n) *out_n = 2 * bins;
    return SS_OK;
}

int ss_mid_side(const float *interleaved, size_t n, float *mid, float *side, size_t *out_frames)
{
    if (out_frames) *out_frames = 0;
    const size_t frames = n / 2;
    if (!frames) return SS_OK;
    if (!interleaved || !mid || !side) return SS_ERR_INVALID_ARG;
    if (require_device()) return SS_ERR_DEVICE;
    Scratch &c = scratch();
    if (!c.stream) HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIPCHK(c.in.ensure(2 * frames));
    HIPCHK(c.out.ensure(2 * frames));
    HIPCHK(hipMemcpyAsync(c.in.p, interleaved, 2 * frames * sizeof(float), hipMemcpyHostToDevice, c.stream));
    HIPCHK(ssk::launch_mid_side(c.in.p, frames, c.out.p, c.out.p + frames, c.stream));
    HIPCHK(hipMemcpyAsync(mid, c.out.p, frames * sizeof(float), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(side, c.out.p + frames, frames * sizeof(float), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (out_frames) *out_frames = frames;
    return SS_OK;
}

size_t ss_pcm_sample_bytes(int format)
{
    switch (format) {
        case SS_PCM_U8: return 1; case SS_PCM_S16: return 2; case SS_PCM_S24: return 3;
        case SS_PCM_S32: return 4; case SS_PCM_F32: return 4; case SS_PCM_F64: return 8;
        default: return 0;
    }
}

// RIFF/WAVE header walk: "RIFF" size "WAVE", then chunks (id, size, payload padded to even)
int ss_wav_parse(const void *file_bytes, size_t len, ss_wav_info *out)
{
    if (!file_bytes || !out) return SS_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const unsigned char *p = static_cast<const unsigned char *>(file_bytes);
    auto u16 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (len < 12 || std::memcmp(p, "RIFF", 4) != 0 || std::memcmp(p + 8, "WAVE", 4) != 0) return SS_ERR_INVALID_ARG;
    size_t pos = 12;
    bool have_fmt = false;
    uint32_t tag = 0, block_align = 0;
    while (pos + 8 <= len) {
        const uint32_t size = u32(pos + 4);
        const size_t body = pos + 8;
        if (std::memcmp(p + pos, "fmt ", 4) == 0) {
            if (size < 16 || body + 16 > len) return SS_ERR_INVALID_ARG;
            tag = u16(body);
            out->channels = u16(body + 2);
            out->sample_rate = u32(body + 4);
            block_align = u16(body + 12);
            out->bits_per_sample = u16(body + 14);
            if (tag == 0xFFFE) {                        // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                if (size < 40 || body + 40 > len) return SS_ERR_INVALID_ARG;
                tag = u16(body + 24);
            }
            have_fmt = true;
        } else if (std::memcmp(p + pos, "data", 4) == 0) {
            if (!have_fmt) return SS_ERR_INVALID_ARG;
            out->data_offset = body;
            uint64_t avail = len - body;
            out->data_bytes = size < avail ? size : avail;   // tolerate a truncated / streaming length
            break;
        }
        pos = body + (size_t)size + (size & 1u);
    }
    if (!have_fmt || !out->data_offset) return SS_ERR_INVALID_ARG;
    if (out->channels == 0) return SS_ERR_INVALID_ARG;
    const uint32_t bits = out->bits_per_sample;
    if (tag == 1) {
        out->format = bits == 8 ? SS_PCM_U8 : bits == 16 ? SS_PCM_S16 : bits == 24 ? SS_PCM_S24 : bits == 32 ? SS_PCM_S32 : 0;
    } else if (tag == 3) {
        out->format = bits == 32 ? SS_PCM_F32 : bits == 64 ? SS_PCM_F64 : 0;
    } else {
        return SS_ERR_UNSUPPORTED;
    }
    if (!out->format) return SS_ERR_UNSUPPORTED;
    const size_t fb = ss_pcm_sample_bytes((int)out->format) * out->channels;
    if (block_align && block_align != fb) return SS_ERR_UNSUPPORTED;
    out->frames = out->data_bytes / fb;
    return SS_OK;
}

int ss_pcm_decode(const void *pcm, size_t n_samples, int format, float *out)
{
    const size_t sb = ss_pcm_sample_bytes(format);
    if (!sb) return SS_ERR_INVALID_ARG;
    if (!n_samples) return SS_OK;
    if (!pcm || !out) return SS_ERR_INVALID_ARG;
    if (require_device()) return SS_ERR_DEVICE;
    Scratch &c = scratch();
    if (!c.stream) HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIPCHK(c.raw.ensure(n_samples * sb + 8));
    HIPCHK(c.out.ensure(n_samples));
    HIPCHK(hipMemcpyAsync(c.raw.p, pcm, n_samples * sb, hipMemcpyHostToDevice, c.stream));
    HIPCHK(ssk::launch_pcm_to_f32(c.raw.p, n_samples, format, c.out.p, c.stream));
    HIPCHK(hipMemcpyAsync(out, c.out.p, n_samples * sizeof(float), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return SS_OK;
}

// add_frames_f32 on the handle's meter.  on_device: `samples` already lives in HBM (tick drivers):
// no staging copy and no synchronisation — everything is only enqueued on the handle's stream.
static int add_samples_impl(ss_analyzer *h, const float *samples, size_t n, bool on_device)
{
    SS_ON_DEVICE(h);
    if (!h) return SS_ERR_INVALID_ARG;
    if (!h->meter_ok) return SS_ERR_INVALID_MODE;
    if (n == 0) return SS_OK;
    if (!samples) return SS_ERR_INVALID_ARG;
    const uint32_t C = h->channels;
    if (n % C) return SS_ERR_NOMEM;             // add_frames_f32: partial frame
    const uint64_t S = h->td->host.s100;
    const double *he, *hb;
    int rc = get_hist_tables(&he, &hb);
    if (rc) return rc;
    // pieces of at most 32 sub-blocks so the sub-block ring (96) always holds the
    // 30-block history a short-term block needs
    const uint64_t piece_frames = 32 * S;
    uint64_t frames = n / C, done = 0;
    while (done < frames) {
        const uint64_t take = frames - done < piece_frames ? frames - done : piece_frames;
        if (!on_device) {
            HIPCHK(h->in.ensure(take * C));
            HIPCHK(hipMemcpyAsync(h->in.p, samples + done * C, take * C * sizeof(float), hipMemcpyHostToDevice, h->stream));
        }
        ssk::TdParams p{};
        p.pcm = on_device ? samples + done * C : h->in.p; p.stream_stride = 0; p.n_frames = take; p.n_streams = 1; p.channels = C;
        p.k = h->td->dev.p; p.state = h->state.p;
        p.subblocks = h->sub.p; p.sub_stride = 0; p.sub_cap = ss_analyzer::kSubCap;
        p.ring = h->ring.p; p.ring_frames = h->ring_frames; p.tp_factor = h->tp_factor;
        p.s100 = (uint32_t)S; p.nseg = 1; p.seg_sub = 0; p.warm_sub = 0;
        HIPCHK(ssk::launch_time_domain(p, h->stream));
        const uint64_t sb0 = h->frames_fed / S, sb1 = (h->frames_fed + take) / S;
        if (sb1 > sb0) {
            ssk::FinalizeParams f{};
            f.k = h->td->dev.p; f.subblocks = h->sub.p; f.sub_stride = 0; f.sub_cap = ss_analyzer::kSubCap;
            f.hist_energies = he; f.hist_bounds = hb; f.weights = h->weights.p;
            f.hist = h->hist.p; f.corpus_hist = nullptr; f.n_streams = 1; f.channels = C;
            f.sub_begin = sb0; f.sub_end = sb1;
            f.out_integrated = nullptr; f.out_lra = nullptr; f.out_counts = h->counts.p;
            HIPCHK(ssk::launch_finalize(f, h->stream));
        }
        // the staging buffer is reused by the next piece
        if (!on_device) HIPCHK(hipStreamSynchronize(h->stream));
        h->frames_fed += take;
        done += take;
    }
    return SS_OK;
}

int ss_add_samples(ss_analyzer *h, const float *samples, size_t n)
{
    SS_ON_DEVICE(h);
    return add_samples_impl(h, samples, n, false);
}

void ss_reset(ss_analyzer *h)
{
    SS_ON_DEVICE(h);
    if (h) (void)handle_reset(h);
}

// energy of the last `frames` frames of the filtered ring -> out2[1] = loudness (enqueue only)
static int ring_loudness_enqueue(ss_analyzer *h, uint64_t frames)
{
    SS_ON_DEVICE(h);
    HIPCHK(ssk::launch_ring_energy(h->ring.p, h->ring_frames, h->channels, h->frames_fed, frames,
                                   h->weights.p, h->out2.p, h->ring_scratch.p, h->stream));
    return SS_OK;
}

static int ring_loudness(ss_analyzer *h, uint64_t frames, double *out)
{
    SS_ON_DEVICE(h);
    if (!h || !out) return SS_ERR_INVALID_ARG;
    if (!h->meter_ok) return SS_ERR_INVALID_MODE;
    if (frames > h->ring_frames) return SS_ERR_INVALID_MODE;
    int rc = ring_loudness_enqueue(h, frames);
    if (rc) return rc;
    double r[2];
    HIPCHK(hipMemcpyAsync(r, h->out2.p, sizeof r, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = r[1];
    return SS_OK;
}

int ss_get_shortterm_lufs(ss_analyzer *h, double *out)
{
    SS_ON_DEVICE(h);
    if (!h || !h->meter_ok) return h ? SS_ERR_INVALID_MODE : SS_ERR_INVALID_ARG;
    return ring_loudness(h, (uint64_t)h->td->host.s100 * 30, out);
}

int ss_get_momentary_lufs(ss_analyzer *h, double *out)
{
    SS_ON_DEVICE(h);
    if (!h || !h->meter_ok) return h ? SS_ERR_INVALID_MODE : SS_ERR_INVALID_ARG;
    return ring_loudness(h, (uint64_t)h->td->host.s100 * 4, out);
}

static int hist_eval(ss_analyzer *h, double r[2])
{
    SS_ON_DEVICE(h);
    if (!h) return SS_ERR_INVALID_ARG;
    if (!h->meter_ok) return SS_ERR_INVALID_MODE;
    const double *he, *hb;
    int rc = get_hist_tables(&he, &hb);
    if (rc) return rc;
    HIPCHK(ssk::launch_hist_eval(h->hist.p, he, hb, h->out2.p, h->stream));
    HIPCHK(hipMemcpyAsync(r, h->out2.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SS_OK;
}

int ss_get_integrated_lufs(ss_analyzer *h, double *out)
{
    SS_ON_DEVICE(h);
    if (!out) return SS_ERR_INVALID_ARG;
    double r[2];
    int rc = hist_eval(h, r);
    if (rc) return rc;
    *out = r[0];
    return SS_OK;
}

int ss_get_loudness_range(ss_analyzer *h, double *out)
{
    SS_ON_DEVICE(h);
    if (!out) return SS_ERR_INVALID_ARG;
    double r[2];
    int rc = hist_eval(h, r);
    if (rc) return rc;
    *out = r[1];
    return SS_OK;
}

static int read_peaks(ss_analyzer *h, uint32_t ch, double *sample_pk, double *true_pk)
{
    SS_ON_DEVICE(h);
    if (!h) return SS_ERR_INVALID_ARG;
    if (!h->meter_ok) return SS_ERR_INVALID_MODE;
    if (ch >= h->channels) return SS_ERR_INVALID_CHANNEL;
    float sp, tp;
    HIPCHK(hipMemcpyAsync(&sp, &h->state.p->sample_peak[ch], sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&tp, &h->state.p->true_peak[ch], sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (sample_pk) *sample_pk = (double)sp;
    if (true_pk) *true_pk = (double)(tp > sp ? tp : sp);    // true_peak(): max(true, sample)
    return SS_OK;
}

int ss_get_true_peak(ss_analyzer *h, double *left, double *right)
{
    SS_ON_DEVICE(h);
    if (!left || !right) return SS_ERR_INVALID_ARG;
    double l, r;
    int rc = read_peaks(h, 0, nullptr, &l);     // analyzer.rs:160
    if (rc) return rc;
    rc = read_peaks(h, 1, nullptr, &r);         // analyzer.rs:161
    if (rc) return rc;
    *left = l; *right = r;
    return SS_OK;
}

int ss_get_true_peak_channel(ss_analyzer *h, uint32_t channel, double *out)
{
    SS_ON_DEVICE(h);
    if (!out) return SS_ERR_INVALID_ARG;
    return read_peaks(h, channel, nullptr, out);
}

int ss_get_sample_peak_channel(ss_analyzer *h, uint32_t channel, double *out)
{
    SS_ON_DEVICE(h);
    if (!out) return SS_ERR_INVALID_ARG;
    return read_peaks(h, channel, out, nullptr);
}

uint32_t ss_sample_rate(const ss_analyzer *h) { return h ? h->rate : 0; }

}  // extern "C"

// ============================================================================
//  batch
// ============================================================================
struct ss_batch {
    int device = 0;             // the HIP device this batch lives on
    ss_batch_config cfg{};
    ss_batch_layout lay{};
    hipStream_t stream = nullptr;
    int tp_factor = 0;
    int fft_mode = 1;           // generic-kernel mode (1 mid/side, 2 per channel)
    bool fft_fast = false;      // N=4096 stereo kernel
    bool fft_pairw = false;     // N=4096, hop 1024, mono / per-channel: two windows per transform
    uint64_t first_start = 0;
    uint32_t wave_window = 0;
    uint32_t windows_per_block = 16;
    uint32_t td_nseg = 1, td_seg_sub = 0;
    bool wave_fused = false;     // decimation runs inside the time-domain kernel
    uint32_t wave_halo = 0;
    FftTables *ft = nullptr;
    BinTables *bt = nullptr;
    TdTables *td = nullptr;
    DevBuf<float> pcm, fft, wave;
    DevBuf<ssk::TdState> state;
    DevBuf<double> sub, weights, integrated, lra, out2;
    DevBuf<uint64_t> hist, corpus;
    DevBuf<uint32_t> counts;
    DevBuf<unsigned char> raw;      // device staging of raw PCM for the asynchronous ingest
    // ragged batches (ss_batch_set_lengths): per-stream frames / windows / sub-blocks / decimation bins
    bool ragged = false;
    std::vector<uint64_t> frames_h, wave_samples_h;
    std::vector<uint32_t> windows_h, sub_h, wave_window_h, wave_bins_h;
    DevBuf<uint64_t> frames_d, wave_samples_d;
    DevBuf<uint32_t> windows_d, sub_d, wave_window_d;
    // render-side reductions (N3)
    DevBuf<float> render_spec, render_wave;
    DevBuf<uint32_t> col_start;
    uint32_t render_cols = 0, render_wave_cols = 0;
    // opt-in (SS_BATCH_OVERLAP=1): the spectrum kernel on a second stream beside the time-domain chain
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int overlap = 0;                  // 0 sequential, 1 the spectrum kernel beside the time-domain chain, 2 beside its tail only
    bool timing = false;
    hipEvent_t ev[2 * SS_KERNEL_COUNT] = {};
    bool ev_ready = false;
    double t_ms[SS_KERNEL_COUNT] = {0, 0, 0, 0};
    uint64_t t_n[SS_KERNEL_COUNT] = {0, 0, 0, 0};
    bool pending_events = false;
};

namespace ssi {
void *batch_corpus_device(ss_batch *b) { return b ? b->corpus.p : nullptr; }
hipStream_t batch_stream(ss_batch *b) { return b ? b->stream : nullptr; }
int batch_device(const ss_batch *b) { return b ? b->device : 0; }
void set_last_error(const std::string &text) { g_last_err = text; }
}  // namespace ssi

namespace {

int batch_collect_timing(ss_batch *b)
{
    if (!b->pending_events) return SS_OK;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int k = 0; k < SS_KERNEL_COUNT; k++) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, b->ev[2 * k], b->ev[2 * k + 1]);
        if (e == hipSuccess) { b->t_ms[k] += ms; b->t_n[k]++; }
    }
    b->pending_events = false;
    return SS_OK;
}

}  // namespace

extern "C" {

int ss_batch_create(const ss_batch_config *cfg, ss_batch **out)
{
    if (!cfg || !out) return SS_ERR_INVALID_ARG;
    *out = nullptr;
    if (require_device()) return SS_ERR_DEVICE;
    if (cfg->n_streams == 0 || cfg->frames_per_stream == 0) return SS_ERR_INVALID_ARG;
    if ((cfg->flags & SS_BATCH_ALL) == 0) return SS_ERR_INVALID_ARG;
    // destroyed (streams and events included) on every early return
    std::unique_ptr<ss_batch, decltype(&ss_batch_destroy)> b(new ss_batch(), &ss_batch_destroy);
    b->device = current_device();
    b->cfg = *cfg;
    const uint32_t C = cfg->channels;
    const uint64_t F = cfg->frames_per_stream;
    if (C == 0 || C > 64) return SS_ERR_NOMEM;
    if (cfg->flags & (SS_BATCH_LUFS | SS_BATCH_TRUE_PEAK)) {
        int rc = meter_args_ok(C, cfg->sample_rate);
        if (rc) return rc;
    }
    if (cfg->true_peak_factor != 0 && cfg->true_peak_factor != 2 && cfg->true_peak_factor != 4) return SS_ERR_INVALID_ARG;
    HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    ss_batch_layout &L = b->lay;
    L.input_bytes = (uint64_t)cfg->n_streams * F * C * sizeof(float);
    HIPCHK(b->pcm.alloc((size_t)cfg->n_streams * F * C));

    if (cfg->flags & SS_BATCH_FFT) {
        const size_t n = cfg->fft_n;
        if (n < 2) return SS_ERR_TOO_FEW_SAMPLES;
        if (!is_pow2(n)) return SS_ERR_NOT_POW2;
        if (n > 32768) return SS_ERR_UNSUPPORTED;
        if (20000.0f > (float)cfg->sample_rate / 2.0f) return SS_ERR_FREQ_LIMIT;
        if (cfg->hop_frames == 0) return SS_ERR_INVALID_ARG;
        int rc = get_fft_tables(n, &b->ft);
        if (rc) return rc;
        rc = get_bin_tables(cfg->sample_rate, n, &b->bt);
        if (rc) return rc;
        // cadence of analyze_audio_file_samples (tui.rs:1482-1526): window [p-N, p) at
        // p = k*hop, skipped when p - N == 0 (saturating_sub) => k from N/hop + 1
        const uint64_t hop = cfg->hop_frames;
        const uint64_t k_min = n / hop + 1, k_max = F / hop;
        L.n_windows = k_max >= k_min ? (uint32_t)(k_max - k_min + 1) : 0;
        b->first_start = k_min * hop - n;
        L.fft_channels = (C == 2) ? 2 : C;
        b->fft_mode = (C == 2) ? 1 : (C == 1 ? 0 : 2);
        L.n_bins = (uint32_t)b->bt->count;
        L.first_bin = (uint32_t)b->bt->first;
        b->fft_fast = (C == 2 && n == 4096 && hop % 256 == 0);
        b->fft_pairw = (C != 2 && n == 4096 && hop == 1024);
#ifdef SS_TUNING        // development builds only: the shipped library takes no kernel selection from the environment
        if (std::getenv("SS_FFT_NO_PAIRW")) b->fft_pairw = false;
#endif
        {
            // windows per workgroup: long runs amortise the per-workgroup constants and the 3-hop halo,
            // but keep >= ~4096 workgroups (8 rounds of the 512 resident ones) for load balance
            uint32_t tgt = (4096u + cfg->n_streams - 1) / cfg->n_streams;
            if (tgt > L.n_windows / 16) tgt = L.n_windows / 16;
            if (tgt < 1) tgt = 1;
            uint32_t wpb = (L.n_windows + tgt - 1) / tgt;
            wpb = (wpb + 1) & ~1u;
            b->windows_per_block = wpb < 2 ? 2 : wpb;
        }
#ifdef SS_TUNING
        if (const char *e = std::getenv("SS_FFT_WPB")) { int v = std::atoi(e); if (v >= 2 && v <= 4096) b->windows_per_block = (uint32_t)(v & ~1); }
#endif
        // Rows start 16-byte aligned (16-byte stores).  Padding them to whole 128-byte lines lifts a pure streaming-store
        // kernel with this row pattern from 3.7 to 4.4 TB/s (tools/ubench_fftio.hip) but does nothing for the real kernel
        // (A/B in one process: 3.14 vs 3.12 ms), so the rows stay compact.  -DSS_FFT_ROW_ALIGN=32u rebuilds the padded form.
#ifndef SS_FFT_ROW_ALIGN
#define SS_FFT_ROW_ALIGN 4u
#endif
        L.fft_bin_stride = (L.n_bins + (SS_FFT_ROW_ALIGN - 1u)) & ~(SS_FFT_ROW_ALIGN - 1u);
        L.fft_bytes = (uint64_t)cfg->n_streams * L.n_windows * L.fft_channels * L.fft_bin_stride * sizeof(float);
        HIPCHK(b->fft.alloc((size_t)(L.fft_bytes / sizeof(float))));
    }
    if (cfg->flags & (SS_BATCH_LUFS | SS_BATCH_TRUE_PEAK)) {
        b->tp_factor = (cfg->flags & SS_BATCH_TRUE_PEAK)
                           ? (cfg->true_peak_factor ? cfg->true_peak_factor : sst::true_peak_factor_for_rate(cfg->sample_rate))
                           : 0;
        int rc = get_td_tables(cfg->sample_rate, b->tp_factor, C, &b->td);
        if (rc) return rc;
        const uint64_t S = b->td->host.s100;
        L.n_subblocks = (uint32_t)(F / S);
        HIPCHK(b->state.alloc(cfg->n_streams));
        HIPCHK(b->sub.alloc((size_t)cfg->n_streams * (L.n_subblocks ? L.n_subblocks : 1) * C));
        HIPCHK(b->hist.alloc((size_t)cfg->n_streams * 2 * sst::kHistBins));
        HIPCHK(b->corpus.alloc(2 * sst::kHistBins));
        HIPCHK(b->integrated.alloc(cfg->n_streams));
        HIPCHK(b->lra.alloc(cfg->n_streams));
        HIPCHK(b->counts.alloc((size_t)cfg->n_streams * 2));
        HIPCHK(b->out2.alloc(2));
        std::vector<double> w(C);
        sst::channel_weights(C, w.data());
        HIPCHK(b->weights.upload(w));
    }
    if (cfg->flags & SS_BATCH_WAVEFORM) {
        const double win = cfg->waveform_window > 0.0 ? cfg->waveform_window : (double)F / (double)cfg->sample_rate;
        const double wd = win * 1000.0;
        const uint64_t W = (wd != wd || wd <= 0.0) ? 0 : (uint64_t)wd;
        if (W > 0xFFFFFFFFull) return SS_ERR_UNSUPPORTED;
        b->wave_window = (uint32_t)W;
        // points: bins whose start floor(i*spp) < len
        const uint64_t len = F * C;
        const double spp = (double)len / (double)W;
        uint64_t bins = W;
        if (W > len) {
            uint64_t lo = 0, hi = W;
            while (lo < hi) { uint64_t mid = lo + (hi - lo) / 2; if ((uint64_t)((double)mid * spp) >= len) hi = mid; else lo = mid + 1; }
            bins = lo;
        }
        L.n_wave_points = (uint32_t)(2 * bins);
        HIPCHK(b->wave.alloc((size_t)cfg->n_streams * (W ? 2 * W : 2)));
        // fuse into the time-domain pass when that pass runs and a bin (plus its shared edge sample)
        // fits the per-wave halo; otherwise the standalone kernel handles it
        if (b->td && W > 0 && spp >= 16.0 && spp <= 1000.0 && len < (1ull << 31)) {
            const uint32_t need = ((uint32_t)std::ceil(spp) + 2 + C - 1) / C;
            uint32_t halo = need < 24 ? 24 : need;
            halo = (halo + 3u) & ~3u;
            if (halo <= 512) { b->wave_fused = true; b->wave_halo = halo; }
        }
    }
    // time segments per stream.  A segment costs a kTdWarmSub-sub-block filter run-in and the chip holds W0
    // waves at once (LDS per wave grows with channels and halo).  Pick the segment length that maximises
    //   useful fraction  seg / (seg + warm)  x  fill of the last round  waves / (ceil(waves / W0) W0)
    // (ranks the measured config-5 sweep seg = 2..13 in the right order; measured within noise for config 3)
    if (b->td) {
        const uint32_t nsub = L.n_subblocks;
        const double W0 = 256.0 * ssk::td_resident_waves_per_cu(C, b->td->host.s100, b->wave_fused ? b->wave_halo : 0);
        auto score_of = [&](uint32_t seg, uint32_t nseg) {
            const double waves = (double)cfg->n_streams * nseg;
            const double useful = nseg > 1 ? (double)seg / (double)(seg + kTdWarmSub) : 1.0;
            return useful * waves / (std::ceil(waves / W0) * W0);
        };
        uint32_t best_seg = 0;
        double best = nsub ? score_of(nsub, 1) : 0.0;                         // one segment: no run-in
        for (uint32_t want = 2; want <= nsub; want++) {                     // balanced segments: seg = ceil(nsub / want)
            const uint32_t seg = (nsub + want - 1) / want;
            if (seg < kTdWarmSub) break;
            const double sc = score_of(seg, (nsub + seg - 1) / seg);
            if (sc > best * 1.0000001) { best = sc; best_seg = seg; }
        }
#ifdef SS_TUNING
        if (const char *e = std::getenv("SS_TD_SEG_SUB")) best_seg = (uint32_t)std::atoi(e);
#endif
        if (best_seg >= kTdWarmSub && best_seg < nsub) {
            b->td_seg_sub = best_seg;
            b->td_nseg = (nsub + best_seg - 1) / best_seg;
        } else {
            b->td_nseg = 1; b->td_seg_sub = 0;
        }
    }
    for (auto &e : b->ev) HIPCHK(hipEventCreate(&e));
    b->ev_ready = true;
    *out = b.release();
    return SS_OK;
}

void ss_batch_destroy(ss_batch *b)
{
    SS_ON_DEVICE(b);
    if (!b) return;
    if (b->stream) { (void)hipStreamSynchronize(b->stream); }
    if (b->stream2) { (void)hipStreamSynchronize(b->stream2); (void)hipStreamDestroy(b->stream2); }
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int ss_batch_layout_get(const ss_batch *b, ss_batch_layout *out)
{
    SS_ON_DEVICE(b);
    if (!b || !out) return SS_ERR_INVALID_ARG;
    *out = b->lay;
    return SS_OK;
}

int ss_batch_upload(ss_batch *b, uint32_t first, uint32_t count, const float *pcm)
{
    SS_ON_DEVICE(b);
    if (!b || !pcm) return SS_ERR_INVALID_ARG;
    if ((uint64_t)first + count > b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)b->cfg.frames_per_stream * b->cfg.channels;
    HIPCHK(hipMemcpyAsync(b->pcm.p + (size_t)first * per, pcm, (size_t)count * per * sizeof(float),
                          hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_upload_pcm(ss_batch *b, uint32_t first, uint32_t count, const void *pcm, int format)
{
    SS_ON_DEVICE(b);
    const size_t sb = ss_pcm_sample_bytes(format);
    if (!b || !pcm || !sb) return SS_ERR_INVALID_ARG;
    if ((uint64_t)first + count > b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)b->cfg.frames_per_stream * b->cfg.channels;
    const size_t n = per * count;
    DevBuf<unsigned char> raw;
    HIPCHK(raw.alloc(n * sb + 8));
    HIPCHK(hipMemcpyAsync(raw.p, pcm, n * sb, hipMemcpyHostToDevice, b->stream));
    HIPCHK(ssk::launch_pcm_to_f32(raw.p, n, format, b->pcm.p + (size_t)first * per, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

// ---- ragged batches: streams of different lengths in one batch ------------------------------------------------
// The batch is created for the longest stream (frames_per_stream = the slot size); every stream then gets its own
// window count, sub-block count and decimation geometry by the very rules ss_batch_create applies to the whole
// batch.  Slots are uploaded as before (the tail of a short stream's slot is never read).
int ss_batch_set_lengths(ss_batch *b, const uint64_t *frames, uint32_t count)
{
    SS_ON_DEVICE(b);
    if (!b || !frames || count != b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const ss_batch_config &c = b->cfg;
    const uint32_t C = c.channels;
    for (uint32_t i = 0; i < count; i++) if (frames[i] > c.frames_per_stream) return SS_ERR_INVALID_ARG;
    b->frames_h.assign(frames, frames + count);
    b->windows_h.assign(count, 0); b->sub_h.assign(count, 0);
    b->wave_window_h.assign(count, 0); b->wave_bins_h.assign(count, 0); b->wave_samples_h.assign(count, 0);
    for (uint32_t i = 0; i < count; i++) {
        const uint64_t F = frames[i];
        if (c.flags & SS_BATCH_FFT) {
            const uint64_t hop = c.hop_frames, k_min = c.fft_n / hop + 1, k_max = F / hop;
            b->windows_h[i] = k_max >= k_min ? (uint32_t)(k_max - k_min + 1) : 0;
        }
        if (b->td) b->sub_h[i] = (uint32_t)(F / b->td->host.s100);
        if (c.flags & SS_BATCH_WAVEFORM) {
            const double win = c.waveform_window > 0.0 ? c.waveform_window : (double)F / (double)c.sample_rate;
            size_t window, bins;
            waveform_shape((size_t)(F * C), win, &window, &bins);
            if (window > b->wave_window) return SS_ERR_INVALID_ARG;      // cannot happen for F <= frames_per_stream
            b->wave_window_h[i] = (uint32_t)window; b->wave_bins_h[i] = (uint32_t)bins; b->wave_samples_h[i] = F * C;
        }
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(b->frames_d.upload(b->frames_h));
    HIPCHK(b->windows_d.upload(b->windows_h));
    HIPCHK(b->sub_d.upload(b->sub_h));
    HIPCHK(b->wave_window_d.upload(b->wave_window_h));
    HIPCHK(b->wave_samples_d.upload(b->wave_samples_h));
    b->ragged = true;
    return SS_OK;
}

int ss_batch_stream_shape(const ss_batch *b, uint32_t stream, ss_stream_shape *out)
{
    SS_ON_DEVICE(b);
    if (!b || !out || stream >= b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    if (b->ragged) {
        out->frames = b->frames_h[stream]; out->n_windows = b->windows_h[stream];
        out->n_subblocks = b->sub_h[stream]; out->n_wave_points = 2 * b->wave_bins_h[stream];
    } else {
        out->frames = b->cfg.frames_per_stream; out->n_windows = b->lay.n_windows;
        out->n_subblocks = b->lay.n_subblocks; out->n_wave_points = b->lay.n_wave_points;
    }
    out->reserved = 0;
    return SS_OK;
}

// ---- pipelined ingest: page-locked host memory + uploads that do not wait -------------------------------------
// A batch owns its stream, so two batches are a double buffer: while one runs, the other's upload is in flight
// on the copy engine.  That only holds for page-locked host memory (pageable copies are staged synchronously).
int ss_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return SS_ERR_INVALID_ARG;
    if (require_device()) return SS_ERR_DEVICE;
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return SS_OK;
}

int ss_host_unregister(void *ptr)
{
    if (!ptr) return SS_ERR_INVALID_ARG;
    HIPCHK(hipHostUnregister(ptr));
    return SS_OK;
}

// like ss_batch_upload_pcm, but returns as soon as the copy and the conversion are queued on the batch's stream:
// `pcm` must stay valid (and should be page-locked) until the next ss_batch_sync / ss_batch_results on this batch
int ss_batch_upload_pcm_async(ss_batch *b, uint32_t first, uint32_t count, const void *pcm, int format)
{
    SS_ON_DEVICE(b);
    const size_t sb = ss_pcm_sample_bytes(format);
    if (!b || !pcm || !sb) return SS_ERR_INVALID_ARG;
    if ((uint64_t)first + count > b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)b->cfg.frames_per_stream * b->cfg.channels;
    const size_t n = per * count;
    if (format == SS_PCM_F32) {
        HIPCHK(hipMemcpyAsync(b->pcm.p + (size_t)first * per, pcm, n * sizeof(float), hipMemcpyHostToDevice, b->stream));
        return SS_OK;
    }
    // one raw staging area per batch, sized for the whole batch; ranges of different `first` do not overlap
    const size_t total = per * b->cfg.n_streams;
    if (b->raw.n < total * sb + 8) {
        HIPCHK(hipStreamSynchronize(b->stream));
        HIPCHK(b->raw.alloc(total * sb + 8));
    }
    unsigned char *dst = b->raw.p + (size_t)first * per * sb;
    HIPCHK(hipMemcpyAsync(dst, pcm, n * sb, hipMemcpyHostToDevice, b->stream));
    HIPCHK(ssk::launch_pcm_to_f32(dst, n, format, b->pcm.p + (size_t)first * per, b->stream));
    return SS_OK;
}

// the first n_samples interleaved samples of one stream's slot, raw PCM of any supported format (ragged batches:
// a stream shorter than the slot).  Queued on the batch's stream like ss_batch_upload_pcm_async.
int ss_batch_upload_samples(ss_batch *b, uint32_t stream, const void *pcm, size_t n_samples, int format)
{
    SS_ON_DEVICE(b);
    const size_t sb = ss_pcm_sample_bytes(format);
    if (!b || (!pcm && n_samples) || !sb || stream >= b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)b->cfg.frames_per_stream * b->cfg.channels;
    if (n_samples > per) return SS_ERR_INVALID_ARG;
    if (!n_samples) return SS_OK;
    float *dst = b->pcm.p + (size_t)stream * per;
    if (format == SS_PCM_F32) {
        HIPCHK(hipMemcpyAsync(dst, pcm, n_samples * sizeof(float), hipMemcpyHostToDevice, b->stream));
        return SS_OK;
    }
    const size_t total = per * b->cfg.n_streams;
    if (b->raw.n < total * sb + 8) {
        HIPCHK(hipStreamSynchronize(b->stream));
        HIPCHK(b->raw.alloc(total * sb + 8));
    }
    unsigned char *raw = b->raw.p + (size_t)stream * per * sb;
    HIPCHK(hipMemcpyAsync(raw, pcm, n_samples * sb, hipMemcpyHostToDevice, b->stream));
    HIPCHK(ssk::launch_pcm_to_f32(raw, n_samples, format, dst, b->stream));
    return SS_OK;
}

int ss_batch_download_input(ss_batch *b, uint32_t stream, float *pcm, size_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !pcm || stream >= b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)b->cfg.frames_per_stream * b->cfg.channels;
    if (cap < per) return SS_ERR_CAPACITY;
    HIPCHK(hipMemcpyAsync(pcm, b->pcm.p + (size_t)stream * per, per * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

void *ss_batch_input_device_ptr(ss_batch *b) { return b ? b->pcm.p : nullptr; }

int ss_batch_synthesize(ss_batch *b, uint64_t seed, uint32_t first_stream_id)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    HIPCHK(ssk::launch_synth(b->pcm.p, b->cfg.n_streams, b->cfg.frames_per_stream, b->cfg.channels,
                             b->cfg.sample_rate, seed, first_stream_id, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_run(ss_batch *b)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    int rc = batch_collect_timing(b);
    if (rc) return rc;
    const ss_batch_config &c = b->cfg;
    const ss_batch_layout &L = b->lay;
    const uint32_t C = c.channels;
    const bool tm = b->timing;
    auto rec = [&](int idx) -> hipError_t { return tm ? hipEventRecord(b->ev[idx], b->stream) : hipSuccess; };

    // overlap mode: fork — the spectrum kernel goes to stream2 after everything already queued on the main stream
    // (uploads, the previous pass), the time-domain chain stays on the main stream, join at the end.  Per-kernel
    // event timing is meaningless while two kernels share the chip, so timing passes stay sequential.
    // Mode 2 (tail overlap): the time-domain kernel runs first and alone; the spectrum kernel starts behind it on stream2
    // while the short latency-bound tail of the chain (gating / histograms per stream, a standalone decimation) runs on
    // the main stream beside it.
    const int mode = tm ? 0 : b->overlap;
    const bool ov = mode != 0;
    hipStream_t fft_stream = ov ? b->stream2 : b->stream;
    if (mode == 1) {
        HIPCHK(hipEventRecord(b->ev_fork, b->stream));
        HIPCHK(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
    }
    auto launch_spectrum = [&]() -> int {
    HIPCHK(rec(2 * SS_KERNEL_FFT));
    if ((c.flags & SS_BATCH_FFT) && L.n_windows) {
        ssk::FftBatchParams p{};
        p.pcm = b->pcm.p; p.out = b->fft.p;
        p.window = b->ft->window.p; p.half_window = b->ft->half_window.p;
        p.tw_n = b->ft->tw_n.p; p.tw_256 = b->ft->tw_256.p; p.pink = b->bt->pink_dev.p;
        p.frames_per_stream = c.frames_per_stream; p.first_start = b->first_start;
        p.n_streams = c.n_streams; p.channels = C; p.n_windows = L.n_windows; p.hop = c.hop_frames;
        p.n = c.fft_n; p.first_bin = L.first_bin; p.n_bins = L.n_bins; p.bin_stride = L.fft_bin_stride;
        p.windows_per_block = b->windows_per_block;
        p.windows_of = b->ragged ? b->windows_d.p : nullptr;
        if (b->fft_fast || b->fft_pairw) {
            p.db_offset = (float)(10.0 * std::log10(4.0 / ((double)c.fft_n * (double)c.fft_n)));
            p.offpink = b->bt->offpink4096_dev.p;
            {
                const uint32_t lo = L.first_bin, hi = L.first_bin + L.n_bins - 1;      // retained bins and their mirrors
                uint32_t mask = 0;
                for (uint32_t kc = 0; kc < 16; kc++) {
                    const uint32_t a0 = 256 * kc, a1 = a0 + 255;
                    const bool direct = a0 <= hi + 3 && a1 >= lo;                       // +3: the last group of four may run past
                    const bool mirror = a0 <= 4096 - lo && a1 + 3 >= 4096 - hi - 3;
                    if (direct || mirror) mask |= 1u << kc;
                }
                p.publish_mask = mask;
            }
            if (b->fft_fast) HIPCHK(ssk::launch_fft4096_ms(p, fft_stream));
            else HIPCHK(ssk::launch_fft4096_pairw(p, b->fft_mode, fft_stream));
        } else {
            p.db_offset = (float)(20.0 * std::log10(4.0 / (double)c.fft_n));
            if (c.fft_n == 16384) {
                p.tw_core = b->ft->core_tw4096; p.tw_256 = b->ft->core_tw256;
                bool run_kernel = (c.hop_frames == 1024 && L.n_windows >= 8);
#ifdef SS_TUNING
                if (std::getenv("SS_FFT16K_SINGLE")) run_kernel = false;
#endif
                if (run_kernel)
                    HIPCHK(ssk::launch_fft16k_run(p, b->fft_mode, fft_stream));
                else
                    HIPCHK(ssk::launch_fft16k(p, b->fft_mode, fft_stream));
            } else {
                HIPCHK(ssk::launch_fft_generic(p, b->fft_mode, fft_stream));
            }
        }
    }
    HIPCHK(rec(2 * SS_KERNEL_FFT + 1));
    return SS_OK;
    };
    if (mode != 2) { rc = launch_spectrum(); if (rc) return rc; }

    const bool td = (c.flags & (SS_BATCH_LUFS | SS_BATCH_TRUE_PEAK)) != 0;
    HIPCHK(rec(2 * SS_KERNEL_TIME_DOMAIN));
    if (td) {
        HIPCHK(hipMemsetAsync(b->state.p, 0, b->state.n * sizeof(ssk::TdState), b->stream));
        HIPCHK(hipMemsetAsync(b->hist.p, 0, b->hist.n * sizeof(uint64_t), b->stream));
        HIPCHK(hipMemsetAsync(b->corpus.p, 0, b->corpus.n * sizeof(uint64_t), b->stream));
        HIPCHK(hipMemsetAsync(b->counts.p, 0, b->counts.n * sizeof(uint32_t), b->stream));
        HIPCHK(rec(2 * SS_KERNEL_TIME_DOMAIN));   // time the kernel, not the memsets
        ssk::TdParams p{};
        p.pcm = b->pcm.p; p.stream_stride = c.frames_per_stream * C; p.n_frames = c.frames_per_stream;
        p.n_streams = c.n_streams; p.channels = C; p.k = b->td->dev.p; p.state = b->state.p;
        p.subblocks = b->sub.p; p.sub_cap = L.n_subblocks ? L.n_subblocks : 1;
        p.sub_stride = (uint64_t)p.sub_cap * C; p.ring = nullptr; p.ring_frames = 0; p.tp_factor = b->tp_factor;
        p.s100 = b->td->host.s100; p.nseg = b->td_nseg; p.seg_sub = b->td_seg_sub; p.warm_sub = kTdWarmSub;
        p.frames_of = b->ragged ? b->frames_d.p : nullptr;
        if (b->wave_fused && !b->ragged) { p.wave_out = b->wave.p; p.wave_stride = (uint64_t)2 * b->wave_window; p.wave_window = b->wave_window; p.halo_frames = b->wave_halo; }
        HIPCHK(ssk::launch_time_domain(p, b->stream));
    }
    HIPCHK(rec(2 * SS_KERNEL_TIME_DOMAIN + 1));
    if (mode == 2) {
        HIPCHK(hipEventRecord(b->ev_fork, b->stream));
        HIPCHK(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
        rc = launch_spectrum(); if (rc) return rc;
    }

    HIPCHK(rec(2 * SS_KERNEL_FINALIZE));
    if (td) {
        const double *he, *hb;
        rc = get_hist_tables(&he, &hb);
        if (rc) return rc;
        ssk::FinalizeParams f{};
        f.k = b->td->dev.p; f.subblocks = b->sub.p; f.sub_cap = L.n_subblocks ? L.n_subblocks : 1;
        f.sub_stride = (uint64_t)f.sub_cap * C;
        f.hist_energies = he; f.hist_bounds = hb; f.weights = b->weights.p; f.hist = b->hist.p;
        f.corpus_hist = b->corpus.p; f.n_streams = c.n_streams; f.channels = C;
        f.sub_begin = 0; f.sub_end = L.n_subblocks;
        f.sub_end_of = b->ragged ? b->sub_d.p : nullptr;
        f.out_integrated = b->integrated.p; f.out_lra = b->lra.p; f.out_counts = b->counts.p;
        HIPCHK(ssk::launch_finalize(f, b->stream));
    }
    HIPCHK(rec(2 * SS_KERNEL_FINALIZE + 1));

    HIPCHK(rec(2 * SS_KERNEL_WAVEFORM));
    if ((c.flags & SS_BATCH_WAVEFORM) && b->wave_window && (!b->wave_fused || b->ragged)) {
        ssk::WaveParams p{};
        p.pcm = b->pcm.p; p.stream_stride = c.frames_per_stream * C; p.n_samples = c.frames_per_stream * C;
        p.n_streams = c.n_streams; p.window = b->wave_window; p.out = b->wave.p; p.out_stride = (uint64_t)2 * b->wave_window;
        if (b->ragged) { p.samples_of = b->wave_samples_d.p; p.window_of = b->wave_window_d.p; }
        HIPCHK(ssk::launch_waveform(p, b->stream));
    }
    HIPCHK(rec(2 * SS_KERNEL_WAVEFORM + 1));
    if (ov) {                                  // join: later work on the main stream (downloads, the next pass) sees the spectrum
        HIPCHK(hipEventRecord(b->ev_join, b->stream2));
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_join, 0));
    }
    b->pending_events = tm;
    return SS_OK;
}

// measurement utility: the spectrum kernel's loads and stores alone (see the header)
int ss_batch_traffic_floor(ss_batch *b, uint32_t reps, double *ms_per_launch)
{
    SS_ON_DEVICE(b);
    if (!b || !ms_per_launch || reps == 0) return SS_ERR_INVALID_ARG;
    const ss_batch_config &c = b->cfg;
    const ss_batch_layout &L = b->lay;
    if (!(c.flags & SS_BATCH_FFT) || !b->fft_fast || c.hop_frames != 1024 || !L.n_windows || b->ragged) return SS_ERR_UNSUPPORTED;
    ssk::FftBatchParams p{};
    p.pcm = b->pcm.p; p.out = b->fft.p;
    p.frames_per_stream = c.frames_per_stream; p.first_start = b->first_start;
    p.n_streams = c.n_streams; p.channels = c.channels; p.n_windows = L.n_windows; p.hop = c.hop_frames;
    p.n = c.fft_n; p.first_bin = L.first_bin; p.n_bins = L.n_bins; p.bin_stride = L.fft_bin_stride;
    p.windows_per_block = b->windows_per_block;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0));
    hipError_t err = hipEventCreate(&e1);
    if (err == hipSuccess) err = ssk::launch_fft4096_traffic(p, b->stream);            // warm
    if (err == hipSuccess) err = hipEventRecord(e0, b->stream);
    for (uint32_t r = 0; r < reps && err == hipSuccess; r++) err = ssk::launch_fft4096_traffic(p, b->stream);
    if (err == hipSuccess) err = hipEventRecord(e1, b->stream);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    HIPCHK(err);
    *ms_per_launch = (double)ms / reps;
    return SS_OK;
}

int ss_batch_sync(ss_batch *b)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    HIPCHK(hipStreamSynchronize(b->stream));
    return batch_collect_timing(b);
}

int ss_batch_results(ss_batch *b, ss_stream_result *out, uint32_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !out) return SS_ERR_INVALID_ARG;
    const uint32_t n = b->cfg.n_streams;
    if (cap < n) return SS_ERR_CAPACITY;
    std::memset(out, 0, sizeof(ss_stream_result) * n);
    if (!b->state.p) return SS_OK;
    std::vector<double> integ(n), lra(n);
    std::vector<uint32_t> cnt(2 * (size_t)n);
    std::vector<ssk::TdState> st(n);
    HIPCHK(hipMemcpyAsync(integ.data(), b->integrated.p, n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(lra.data(), b->lra.p, n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(cnt.data(), b->counts.p, 2 * (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(st.data(), b->state.p, n * sizeof(ssk::TdState), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint32_t i = 0; i < n; i++) {
        out[i].integrated_lufs = integ[i];
        out[i].loudness_range = lra[i];
        for (uint32_t c = 0; c < 2 && c < b->cfg.channels; c++) {
            const float sp = st[i].sample_peak[c], tp = st[i].true_peak[c];
            out[i].sample_peak[c] = sp;
            out[i].true_peak[c] = tp > sp ? tp : sp;
        }
        out[i].n_gating_blocks = cnt[2 * i];
        out[i].n_st_blocks = cnt[2 * i + 1];
    }
    return SS_OK;
}

// every channel's peaks of one stream: EbuR128::true_peak(c) = max(true, sample) and EbuR128::sample_peak(c)
int ss_batch_peaks(ss_batch *b, uint32_t stream, double *true_pk, double *sample_pk, uint32_t cap_channels)
{
    SS_ON_DEVICE(b);
    if (!b || stream >= b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    if (!b->state.p) return SS_ERR_INVALID_MODE;                 // the batch runs no meter pass
    const uint32_t C = b->cfg.channels;
    if (cap_channels < C) return SS_ERR_CAPACITY;
    float sp[ssk::kMaxChannels], tp[ssk::kMaxChannels];
    const ssk::TdState *st = b->state.p + stream;
    HIPCHK(hipMemcpyAsync(sp, st->sample_peak, C * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(tp, st->true_peak, C * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint32_t c = 0; c < C; c++) {
        if (sample_pk) sample_pk[c] = (double)sp[c];
        if (true_pk) true_pk[c] = (double)(tp[c] > sp[c] ? tp[c] : sp[c]);
    }
    return SS_OK;
}

int ss_batch_geometry_get(const ss_batch *b, ss_batch_geometry *out)
{
    if (!b || !out) return SS_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const ss_batch_layout &L = b->lay;
    if ((b->cfg.flags & SS_BATCH_FFT) && L.n_windows) {
        out->fft_windows_per_block = b->windows_per_block;
        const bool run16k = b->cfg.fft_n == 16384 && b->cfg.hop_frames == 1024 && L.n_windows >= 8 && !b->fft_fast && !b->fft_pairw;
        if (b->fft_fast) {
            out->fft_blocks = b->cfg.n_streams * ((L.n_windows + b->windows_per_block - 1) / b->windows_per_block);
        } else if (b->fft_pairw) {
            const uint32_t ppb = b->windows_per_block >> 1, np = (L.n_windows + 1) >> 1;
            out->fft_blocks = b->cfg.n_streams * L.fft_channels * ((np + ppb - 1) / ppb);
        } else if (run16k) {
            uint32_t wpb = 0, groups = 0;
            ssk::fft16k_run_geometry(b->cfg.n_streams, L.fft_channels, L.n_windows, &wpb, &groups);
            out->fft_windows_per_block = wpb;
            out->fft_blocks = b->cfg.n_streams * L.fft_channels * groups;
        } else {
            out->fft_windows_per_block = 1;
            out->fft_blocks = b->cfg.n_streams * L.n_windows * L.fft_channels;
        }
    }
    if (b->td) {
        out->td_segments = b->td_nseg;
        out->td_segment_subblocks = b->td_seg_sub;
        out->td_warm_subblocks = b->td_nseg > 1 ? kTdWarmSub : 0;
        out->td_true_peak_factor = (uint32_t)b->tp_factor;
    }
    out->waveform_fused = (b->wave_fused && !b->ragged) ? 1u : 0u;
    out->overlap = (uint32_t)b->overlap;
    return SS_OK;
}

int ss_batch_set_overlap(ss_batch *b, int enable)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    if (enable && !b->stream2) {
        HIPCHK(hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    b->overlap = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return SS_OK;
}

int ss_batch_download_fft(ss_batch *b, uint32_t stream, float *out, size_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !out || stream >= b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const size_t rows = (size_t)b->lay.n_windows * b->lay.fft_channels;
    const size_t per = rows * b->lay.n_bins;
    if (cap < per) return SS_ERR_CAPACITY;
    if (!per) return SS_OK;
    // device rows are padded to fft_bin_stride floats; hand back the compact [window][channel][bin] array
    HIPCHK(hipMemcpy2DAsync(out, (size_t)b->lay.n_bins * sizeof(float),
                            b->fft.p + (size_t)stream * rows * b->lay.fft_bin_stride,
                            (size_t)b->lay.fft_bin_stride * sizeof(float), (size_t)b->lay.n_bins * sizeof(float), rows,
                            hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_bin_tables(const ss_batch *b, double *chart_x, double *freq, double *pink_db)
{
    SS_ON_DEVICE(b);
    if (!b || !b->bt) return SS_ERR_INVALID_ARG;
    const size_t n = b->bt->count;
    if (chart_x) std::memcpy(chart_x, b->bt->chart_x.data(), n * sizeof(double));
    if (freq) std::memcpy(freq, b->bt->freq.data(), n * sizeof(double));
    if (pink_db) std::memcpy(pink_db, b->bt->pink.data(), n * sizeof(double));
    return SS_OK;
}

int ss_batch_download_waveform(ss_batch *b, uint32_t stream, float *out, size_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !out || stream >= b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    const size_t pts = b->lay.n_wave_points;
    if (cap < pts) return SS_ERR_CAPACITY;
    if (!pts) return SS_OK;
    HIPCHK(hipMemcpyAsync(out, b->wave.p + (size_t)stream * 2 * b->wave_window, pts * sizeof(float),
                          hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_download_subblocks(ss_batch *b, uint32_t stream, double *out, size_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !out || stream >= b->cfg.n_streams || !b->sub.p) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)b->lay.n_subblocks * b->cfg.channels;
    if (cap < per) return SS_ERR_CAPACITY;
    if (!per) return SS_OK;
    HIPCHK(hipMemcpyAsync(out, b->sub.p + (size_t)stream * per, per * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_histograms(ss_batch *b, uint64_t *out2000)
{
    SS_ON_DEVICE(b);
    if (!b || !out2000 || !b->corpus.p) return SS_ERR_INVALID_ARG;
    HIPCHK(hipMemcpyAsync(out2000, b->corpus.p, 2 * sst::kHistBins * sizeof(uint64_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_histograms_device(ss_batch *b, void *dst)
{
    SS_ON_DEVICE(b);
    if (!b || !dst || !b->corpus.p) return SS_ERR_INVALID_ARG;
    HIPCHK(hipMemcpyAsync(dst, b->corpus.p, 2 * sst::kHistBins * sizeof(uint64_t), hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

// The corpus gate without leaving the device: [sum over the ranks] + loudness_global / loudness_range of the corpus
// histograms, queued on the batch's stream behind ss_batch_run.  Nothing is copied or waited for, so a loop of passes
// needs no host synchronisation per pass; ss_batch_corpus_gate_read fetches the pair.
int ss_batch_corpus_gate_enqueue(ss_batch *b, ss_comm *comm)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    if (!b->corpus.p) return SS_ERR_INVALID_MODE;
    if (comm) {
        int rc = ss_batch_allreduce_histograms(b, comm, nullptr);
        if (rc) return rc;
    }
    const double *he, *hb;
    int rc = get_hist_tables(&he, &hb);
    if (rc) return rc;
    HIPCHK(ssk::launch_hist_eval(b->corpus.p, he, hb, b->out2.p, b->stream));
    return SS_OK;
}

int ss_batch_corpus_gate_read(ss_batch *b, double *integrated, double *lra)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    if (!b->corpus.p) return SS_ERR_INVALID_MODE;
    double r[2];
    HIPCHK(hipMemcpyAsync(r, b->out2.p, sizeof r, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (integrated) *integrated = r[0];
    if (lra) *lra = r[1];
    return SS_OK;
}

double ss_corpus_integrated_lufs(const uint64_t *h) { return h ? sst::gated_loudness(h) : NAN; }
double ss_corpus_loudness_range(const uint64_t *h) { return h ? sst::loudness_range(h) : NAN; }

// ---- render-side reductions ---------------------------------
int ss_batch_render_spectrum(ss_batch *b, uint32_t cols, int gain_mode, float gain_db)
{
    SS_ON_DEVICE(b);
    if (!b || cols == 0 || cols > 65536 || (gain_mode != SS_GAIN_FIXED && gain_mode != SS_GAIN_REFERENCE))
        return SS_ERR_INVALID_ARG;
    const ss_batch_layout &L = b->lay;
    if (!(b->cfg.flags & SS_BATCH_FFT) || !L.n_windows || !L.n_bins) return SS_ERR_INVALID_MODE;
    if (gain_mode == SS_GAIN_REFERENCE && !(b->cfg.flags & SS_BATCH_LUFS)) return SS_ERR_INVALID_MODE;
    // column of a bin: floor(chart_x / 100 * cols), the last column closed on the right; chart_x ascends
    std::vector<uint32_t> start(cols + 1, L.n_bins);
    {
        uint32_t c = 0;
        start[0] = 0;
        for (uint32_t i = 0; i < L.n_bins; i++) {
            double f = std::floor(b->bt->chart_x[i] / 100.0 * (double)cols);
            if (f < 0) f = 0;
            uint32_t ci = f >= (double)cols ? cols - 1 : (uint32_t)f;
            while (c < ci) start[++c] = i;
        }
        while (c < cols) start[++c] = L.n_bins;
    }
    const uint64_t rows = (uint64_t)b->cfg.n_streams * L.n_windows * L.fft_channels;
    HIPCHK(b->col_start.ensure(cols + 1));
    HIPCHK(hipMemcpyAsync(b->col_start.p, start.data(), (cols + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));           // `start` is a local
    HIPCHK(b->render_spec.ensure(rows * cols));
    b->render_cols = cols;
    HIPCHK(ssk::launch_render_spectrum(b->fft.p, L.fft_bin_stride, L.n_bins, rows, L.n_windows * L.fft_channels,
                                       b->col_start.p, cols,
                                       gain_mode == SS_GAIN_REFERENCE ? b->integrated.p : nullptr, gain_db,
                                       b->render_spec.p, b->stream));
    return SS_OK;
}

int ss_batch_download_spectrum_columns(ss_batch *b, uint32_t stream, float *out, size_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !out || stream >= b->cfg.n_streams || !b->render_cols) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)b->lay.n_windows * b->lay.fft_channels * b->render_cols;
    if (cap < per) return SS_ERR_CAPACITY;
    HIPCHK(hipMemcpyAsync(out, b->render_spec.p + (size_t)stream * per, per * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_render_waveform(ss_batch *b, uint32_t cols, uint32_t x_min, uint32_t x_max)
{
    SS_ON_DEVICE(b);
    if (!b || cols == 0 || cols > 65536 || x_max <= x_min) return SS_ERR_INVALID_ARG;
    if (!(b->cfg.flags & SS_BATCH_WAVEFORM) || !b->wave_window) return SS_ERR_INVALID_MODE;
    HIPCHK(b->render_wave.ensure((size_t)b->cfg.n_streams * cols * 2));
    b->render_wave_cols = cols;
    HIPCHK(ssk::launch_render_waveform(b->wave.p, (uint64_t)2 * b->wave_window, b->lay.n_wave_points / 2,
                                       b->cfg.n_streams, x_min, x_max, cols, b->render_wave.p, b->stream));
    return SS_OK;
}

int ss_batch_download_waveform_columns(ss_batch *b, uint32_t stream, float *out, size_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !out || stream >= b->cfg.n_streams || !b->render_wave_cols) return SS_ERR_INVALID_ARG;
    const size_t per = (size_t)2 * b->render_wave_cols;
    if (cap < per) return SS_ERR_CAPACITY;
    HIPCHK(hipMemcpyAsync(out, b->render_wave.p + (size_t)stream * per, per * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

// the waveform chart's x bounds in Player mode (tui.rs:664-681), f64 like the reference
void ss_waveform_view(double playhead_ms, double waveform_window_s, size_t chart_points, double *x_min, double *x_max)
{
    const double half_window = waveform_window_s * 500.0;
    const double max_x = (double)chart_points / 2.0;
    double lo = playhead_ms - half_window;
    lo = std::fmin(lo, max_x - waveform_window_s * 1000.0);
    lo = std::fmax(lo, 0.0);
    double hi = playhead_ms + half_window;
    hi = std::fmin(hi, max_x);
    hi = std::fmax(hi, waveform_window_s * 1000.0);
    if (x_min) *x_min = lo;
    if (x_max) *x_max = hi;
}

int ss_batch_timing_enable(ss_batch *b, int enable)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    int rc = batch_collect_timing(b);
    if (rc) return rc;
    b->timing = enable != 0;
    for (int k = 0; k < SS_KERNEL_COUNT; k++) { b->t_ms[k] = 0; b->t_n[k] = 0; }
    return SS_OK;
}

int ss_batch_timing_read(ss_batch *b, int kernel, double *total_ms, uint64_t *launches)
{
    SS_ON_DEVICE(b);
    if (!b || kernel < 0 || kernel >= SS_KERNEL_COUNT) return SS_ERR_INVALID_ARG;
    int rc = batch_collect_timing(b);
    if (rc) return rc;
    if (total_ms) *total_ms = b->t_ms[kernel];
    if (launches) *launches = b->t_n[kernel];
    return SS_OK;
}

// the spectrum kernel a batch of this shape launches (names as rocprofv3 prints them, without template arguments)
const char *ss_batch_kernel_name(const ss_batch *b, int kernel)
{
    if (!b || kernel != SS_KERNEL_FFT) return ss_kernel_name(kernel);
    if (b->fft_pairw) return "k_fft4096_pairw";
    if (b->fft_fast) {
        const uint32_t hop = b->cfg.hop_frames;
        return hop == 1024 ? "k_fft4096_ms1" : ((hop == 512 || hop == 2048) ? "k_fft4096_ms" : "k_fft4096_ms_anyhop");
    }
    if (b->cfg.fft_n == 16384)
        return (b->cfg.hop_frames == 1024 && b->lay.n_windows >= 8) ? "k_fft16k_run" : "k_fft16k";
    return "k_fft_generic";
}

const char *ss_kernel_name(int kernel)
{
    switch (kernel) {
        case SS_KERNEL_FFT: return "k_fft4096_ms1";
        case SS_KERNEL_TIME_DOMAIN: return "k_time_domain";
        case SS_KERNEL_FINALIZE: return "k_finalize";
        case SS_KERNEL_WAVEFORM: return "k_waveform";
        default: return "?";
    }
}

// Analyzer::calculate_integrated_lufs (analyzer.rs:170-182): fresh meter at the
// handle's sample rate, whole buffer fed in 2*sr-sample chunks, loudness_global.
static int integrated_oneshot(uint32_t rate, uint32_t channels, const float *samples, size_t n,
                              bool on_device, double *out)
{
    int rc = meter_args_ok(channels, rate);           // EbuR128::new(...) else return None
    if (rc) return rc;
    // every chunk of samples.chunks(2*sr) must hold whole frames, else add_frames fails -> None
    const size_t chunk = (size_t)rate * 2;
    if (n > 0) {
        if (chunk % channels) { if (n >= chunk || n % channels) return SS_ERR_NOMEM; }
        else if (n % channels) return SS_ERR_NOMEM;
    }
    if (n == 0) { *out = -INFINITY; return SS_OK; }    // no blocks: loudness_global() = -inf
    if (!samples) return SS_ERR_INVALID_ARG;
    ss_batch_config cfg{};
    cfg.sample_rate = rate; cfg.channels = channels; cfg.n_streams = 1; cfg.flags = SS_BATCH_LUFS;
    cfg.frames_per_stream = n / channels; cfg.fft_n = 0; cfg.hop_frames = 0;
    ss_batch *b = nullptr;
    rc = ss_batch_create(&cfg, &b);
    if (rc) return rc;
    if (on_device) {
        if (!hip_ok(hipMemcpyAsync(b->pcm.p, samples, n * sizeof(float), hipMemcpyDeviceToDevice, b->stream),
                    "hipMemcpyAsync(D2D)")) rc = SS_ERR_DEVICE;
    } else {
        rc = ss_batch_upload(b, 0, 1, samples);
    }
    if (!rc) rc = ss_batch_run(b);
    if (!rc) rc = ss_batch_sync(b);
    ss_stream_result r{};
    if (!rc) rc = ss_batch_results(b, &r, 1);
    ss_batch_destroy(b);
    if (rc) return rc;
    *out = r.integrated_lufs;
    return SS_OK;
}

int ss_calculate_integrated_lufs(ss_analyzer *h, uint32_t channels, const float *samples, size_t n, double *out)
{
    SS_ON_DEVICE(h);
    if (!h || !out) return SS_ERR_INVALID_ARG;
    return integrated_oneshot(h->rate, channels, samples, n, false, out);
}

}  // extern "C"

// ============================================================================
//  Tick drivers: App's per-file / per-device analysis state
// ============================================================================
struct ss_session {
    int device = 0;                     // the HIP device this session lives on
    ss_analyzer *an = nullptr;          // file_analyzer / device_analyzer
    bool is_file = false;
    uint32_t file_channels = 2, rate = 0;
    size_t n_samples = 0;               // file: interleaved samples; capture: 30 * rate
    DevBuf<float> pcm;                  // the file / the capture ring, resident
    DevBuf<float> ms;                   // capture: mid | side (n/2 each)
    DevBuf<float> spec;                 // [2][bin_stride] dB rows of a tick
    DevBuf<float> wave;                 // capture: [bins][2]
    hipStream_t fft_stream = nullptr;   // the spectrum of a tick runs beside the loudness chain
    float *stage = nullptr;             // pinned: 2 * bin_stride floats | wave floats
    double *stage_d = nullptr;          // pinned: short-term loudness (2 doubles)
    size_t stage_floats = 0;
    FftTables *ft = nullptr;
    BinTables *bt = nullptr;
    uint32_t bin_stride = 0;
    // pairs whose mid or side value is NaN / infinite (normally none): index -> class bits
    // (1 mid NaN, 2 mid inf, 4 side NaN, 8 side inf)
    std::vector<std::pair<size_t, uint8_t>> bad;
    std::vector<double> waveform_xy;    // audio_file_chart
    float gain_db = 0.0f;
    uint64_t duration_ms = 0;
    double lufs[SS_LUFS_HISTORY];
};

namespace {

uint8_t pair_class(float l, float r)
{
    const float m = (l + r) * 0.5f, sd = (l - r) * 0.5f;
    uint8_t c = 0;
    if (std::isnan(m)) c |= 1; else if (std::isinf(m)) c |= 2;
    if (std::isnan(sd)) c |= 4; else if (std::isinf(sd)) c |= 8;
    return c;
}

// the crate's NaN / infinity rejection on the windowed slice [lb, lb + n) of mid (shift 0) or side (shift 2)
int window_value_status(const ss_session *s, size_t lb, size_t n, int shift,
                        const std::vector<std::pair<size_t, uint8_t>> &bad)
{
    auto it = std::lower_bound(bad.begin(), bad.end(), std::make_pair(lb, (uint8_t)0));
    bool any_nan = false, any_inf = false;
    for (; it != bad.end() && it->first < lb + n; ++it) {
        const uint8_t c = (uint8_t)((it->second >> shift) & 3u);
        if (c & 1u) any_nan = true;
        else if (c & 2u) { if (s->ft->window_host[it->first - lb] == 0.0f) any_nan = true; else any_inf = true; }
    }
    return any_nan ? SS_ERR_NAN : (any_inf ? SS_ERR_INFINITY : SS_OK);
}

int session_common_init(ss_session *s, uint32_t meter_channels, uint32_t rate)
{
    s->rate = rate;
    for (double &v : s->lufs) v = -100.0;
    int rc = ss_analyzer_create(2, 44100, &s->an);                 // Analyzer::default()
    if (rc) return rc;
    // create_loudness_meter: the rate sticks even when the meter cannot be made (analyzer.rs:50);
    // the reference only reports the error and carries on
    (void)ss_analyzer_configure(s->an, meter_channels, rate);
    rc = get_fft_tables(SS_TICK_WINDOW, &s->ft);
    if (rc) return rc;
    rc = get_bin_tables(rate, SS_TICK_WINDOW, &s->bt);
    if (rc) return rc;
    s->bin_stride = (uint32_t)((s->bt->count + 3) & ~(size_t)3);
    if (s->bin_stride == 0) s->bin_stride = 4;
    HIPCHK(s->spec.alloc((size_t)2 * s->bin_stride));
    HIPCHK(hipStreamCreateWithFlags(&s->fft_stream, hipStreamNonBlocking));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->stage_d), 2 * sizeof(double), hipHostMallocDefault));
    return SS_OK;
}

int session_stage(ss_session *s, size_t floats)
{
    if (floats <= s->stage_floats) return SS_OK;
    if (s->stage) (void)hipHostFree(s->stage);
    s->stage = nullptr; s->stage_floats = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->stage), floats * sizeof(float), hipHostMallocDefault));
    s->stage_floats = floats;
    return SS_OK;
}

// enqueue the mid/side spectrum of pairs [lb, lb + 16384) of an interleaved pair buffer
int session_enqueue_fft(ss_session *s, const float *pairs, size_t lb, hipStream_t stream)
{
    ssk::FftBatchParams p{};
    p.pcm = pairs; p.out = s->spec.p;
    p.window = s->ft->window.p; p.half_window = s->ft->half_window.p;
    p.tw_n = s->ft->tw_n.p; p.tw_core = s->ft->core_tw4096; p.tw_256 = s->ft->core_tw256; p.pink = nullptr;
    p.frames_per_stream = 0; p.first_start = lb; p.n_streams = 1; p.channels = 2;
    p.n_windows = 1; p.hop = 0; p.n = SS_TICK_WINDOW;
    p.first_bin = (uint32_t)s->bt->first; p.n_bins = (uint32_t)s->bt->count; p.bin_stride = s->bin_stride;
    p.windows_per_block = 1;
    p.db_offset = (float)(20.0 * std::log10(4.0 / (double)SS_TICK_WINDOW));
    HIPCHK(ssk::launch_fft16k(p, 1, stream));
    return SS_OK;
}

// after the synchronisation: dB rows in the pinned stage -> (chart_x, dB + pink) pairs, or the (0,0) fallback
void session_emit_spectrum(const ss_session *s, const float *row, int status_in, double *xy,
                           int32_t *status_out, uint32_t *n_out)
{
    int st = status_in;
    const size_t nb = s->bt->count;
    if (!st)
        for (size_t i = 0; i < nb; i++)
            if (std::isnan(row[i]) || std::isinf(row[i])) { st = SS_ERR_SCALING; break; }
    if (st) {
        xy[0] = 0.0; xy[1] = 0.0;                    // vec![(0., 0.)] (tui.rs:1437-1452, :1505-1524)
        *n_out = 1;
    } else {
        for (size_t i = 0; i < nb; i++) {
            xy[2 * i] = s->bt->chart_x[i];
            xy[2 * i + 1] = (double)row[i] + s->bt->pink[i];
        }
        *n_out = (uint32_t)nb;
    }
    *status_out = st;
}

}  // namespace

extern "C" {

void ss_session_close(ss_session *s)
{
    SS_ON_DEVICE(s);
    if (!s) return;
    if (s->fft_stream) { (void)hipStreamSynchronize(s->fft_stream); (void)hipStreamDestroy(s->fft_stream); }
    if (s->an) ss_analyzer_destroy(s->an);
    if (s->stage) (void)hipHostFree(s->stage);
    if (s->stage_d) (void)hipHostFree(s->stage_d);
    delete s;
}

ss_analyzer *ss_session_analyzer(ss_session *s) { return s ? s->an : nullptr; }

int ss_session_open_file(const float *interleaved, size_t n_samples, uint32_t channels,
                         uint32_t sample_rate, ss_session **out)
{
    if (!out) return SS_ERR_INVALID_ARG;
    *out = nullptr;
    if ((!interleaved && n_samples) || channels == 0 || sample_rate == 0) return SS_ERR_INVALID_ARG;
    if (require_device()) return SS_ERR_DEVICE;
    std::unique_ptr<ss_session, void (*)(ss_session *)> s(new ss_session(), ss_session_close);
    s->device = current_device();
    s->is_file = true; s->file_channels = channels; s->n_samples = n_samples;
    int rc = session_common_init(s.get(), 2, sample_rate);          // meter: 2 channels (tui.rs:1217-1221)
    if (rc) return rc;
    ss_analyzer *h = s->an;
    HIPCHK(s->pcm.alloc(n_samples ? n_samples : 1));
    if (n_samples)
        HIPCHK(hipMemcpyAsync(s->pcm.p, interleaved, n_samples * sizeof(float), hipMemcpyHostToDevice, h->stream));
    const size_t pairs = n_samples / 2;
    for (size_t i = 0; i < pairs; i++) {
        const uint8_t c = pair_class(interleaved[2 * i], interleaved[2 * i + 1]);
        if (c) s->bad.emplace_back(i, c);
    }
    // AudioFile::from_file: duration = mid.len() / rate * 1000. ms, truncated (audio_player.rs:153-161)
    const double dur_ms = (double)pairs / (double)sample_rate * 1000.0;
    s->duration_ms = dur_ms >= 1.8446744073709552e19 ? UINT64_MAX : (uint64_t)dur_ms;
    // Duration::as_secs_f64
    const double dur_s = (double)(s->duration_ms / 1000) + (double)((s->duration_ms % 1000) * 1000000ull) / 1e9;
    // audio_file_chart = get_waveform(samples, duration_s) on the resident buffer
    size_t window, bins;
    waveform_shape(n_samples, dur_s, &window, &bins);
    if (bins > 0xFFFFFFFFull || window > 0xFFFFFFFFull) return SS_ERR_UNSUPPORTED;
    if (bins) {
        HIPCHK(s->wave.alloc(2 * bins));
        ssk::WaveParams p{};
        p.pcm = s->pcm.p; p.stream_stride = n_samples; p.n_samples = n_samples; p.n_streams = 1;
        p.window = (uint32_t)window; p.out = s->wave.p; p.out_stride = 2 * bins;
        HIPCHK(ssk::launch_waveform(p, h->stream));
        std::vector<float> mm(2 * bins);
        HIPCHK(hipMemcpyAsync(mm.data(), s->wave.p, 2 * bins * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        s->waveform_xy.resize(4 * bins);
        for (size_t i = 0; i < bins; i++) {
            s->waveform_xy[4 * i + 0] = (double)i; s->waveform_xy[4 * i + 1] = (double)mm[2 * i];
            s->waveform_xy[4 * i + 2] = (double)i; s->waveform_xy[4 * i + 3] = (double)mm[2 * i + 1];
        }
        s->wave.release();
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    // fft_gain_compensation_db (tui.rs:1229-1238), f32 arithmetic
    double integrated = 0.0;
    rc = integrated_oneshot(sample_rate, 2, s->pcm.p, n_samples, true, &integrated);
    s->gain_db = rc ? 0.0f : (-13.0f - (float)integrated);
    rc = session_stage(s.get(), (size_t)2 * s->bin_stride);
    if (rc) return rc;
    *out = s.release();
    return SS_OK;
}

int ss_session_open_capture(uint32_t channels, uint32_t sample_rate, ss_session **out)
{
    if (!out) return SS_ERR_INVALID_ARG;
    *out = nullptr;
    if (sample_rate == 0) return SS_ERR_INVALID_ARG;
    if ((uint64_t)15 * sample_rate < SS_TICK_WINDOW) return SS_ERR_INVALID_ARG;   // 15*sr - 2^14 underflows (tui.rs:1431)
    if (require_device()) return SS_ERR_DEVICE;
    std::unique_ptr<ss_session, void (*)(ss_session *)> s(new ss_session(), ss_session_close);
    s->device = current_device();
    s->is_file = false; s->file_channels = channels; s->n_samples = (size_t)30 * sample_rate;
    int rc = session_common_init(s.get(), channels, sample_rate);
    if (rc) return rc;
    HIPCHK(s->pcm.alloc(s->n_samples));
    HIPCHK(s->ms.alloc(s->n_samples));
    size_t window, bins;
    waveform_shape(s->n_samples / 2, 15.0, &window, &bins);
    HIPCHK(s->wave.alloc(2 * bins ? 2 * bins : 2));
    rc = session_stage(s.get(), (size_t)2 * s->bin_stride + 2 * bins);
    if (rc) return rc;
    *out = s.release();
    return SS_OK;
}

int ss_session_waveform(ss_session *s, double *out_xy, size_t cap_pairs, size_t *out_n)
{
    SS_ON_DEVICE(s);
    if (out_n) *out_n = 0;
    if (!s || !s->is_file || (!out_xy && cap_pairs)) return SS_ERR_INVALID_ARG;
    const size_t pairs = s->waveform_xy.size() / 2;
    if (pairs > cap_pairs) return SS_ERR_CAPACITY;
    if (pairs) std::memcpy(out_xy, s->waveform_xy.data(), s->waveform_xy.size() * sizeof(double));
    if (out_n) *out_n = pairs;
    return SS_OK;
}

int ss_session_gain_db(ss_session *s, float *out)
{
    SS_ON_DEVICE(s);
    if (!s || !out) return SS_ERR_INVALID_ARG;
    *out = s->gain_db;
    return SS_OK;
}

int ss_session_duration_ms(ss_session *s, uint64_t *out)
{
    SS_ON_DEVICE(s);
    if (!s || !out || !s->is_file) return SS_ERR_INVALID_ARG;
    *out = s->duration_ms;
    return SS_OK;
}

int ss_session_restart(ss_session *s)
{
    SS_ON_DEVICE(s);
    if (!s) return SS_ERR_INVALID_ARG;
    for (double &v : s->lufs) v = -100.0;
    ss_reset(s->an);
    return SS_OK;
}

int ss_session_lufs_history(ss_session *s, double *out300)
{
    SS_ON_DEVICE(s);
    if (!s || !out300) return SS_ERR_INVALID_ARG;
    std::memcpy(out300, s->lufs, sizeof s->lufs);
    return SS_OK;
}

// analyze_audio_file_samples(pos) (tui.rs:1482-1552) on the resident file
int ss_session_tick_file(ss_session *s, size_t pos, double *mid_xy, double *side_xy,
                         size_t cap_pairs, ss_tick_result *res)
{
    SS_ON_DEVICE(s);
    if (!s || !s->is_file || !res || !mid_xy || !side_xy) return SS_ERR_INVALID_ARG;
    if (cap_pairs < s->bt->count || cap_pairs < 1) return SS_ERR_CAPACITY;
    ss_analyzer *h = s->an;
    std::memset(res, 0, sizeof *res);
    const size_t pos_f = pos / s->file_channels;
    res->playhead = pos_f;
    bool fft_launched = false, st_launched = false;
    int mid_st = SS_OK, side_st = SS_OK;

    // ---- spectrum: the last 16384 mid / side samples before the playhead
    const size_t fft_lb = pos_f > SS_TICK_WINDOW ? pos_f - SS_TICK_WINDOW : 0;     // saturating_sub
    if (fft_lb != 0) {
        res->fft_ran = 1;
        const size_t ms_len = s->n_samples / 2;
        if (pos_f <= ms_len && fft_lb < ms_len) {
            // get_fft's own checks on a 16384-sample slice
            mid_st = window_value_status(s, fft_lb, SS_TICK_WINDOW, 0, s->bad);
            side_st = window_value_status(s, fft_lb, SS_TICK_WINDOW, 2, s->bad);
            const int lim = (20000.0f > (float)h->rate / 2.0f) ? SS_ERR_FREQ_LIMIT : SS_OK;
            if (!mid_st) mid_st = lim;
            if (!side_st) side_st = lim;
            if ((!mid_st || !side_st) && s->bt->count) {
                // the file is resident and read-only: the spectrum runs on its own stream beside the loudness chain
                int rc = session_enqueue_fft(s, s->pcm.p, fft_lb, s->fft_stream);
                if (rc) return rc;
                HIPCHK(hipMemcpyAsync(s->stage, s->spec.p, (size_t)2 * s->bin_stride * sizeof(float),
                                      hipMemcpyDeviceToHost, s->fft_stream));
                fft_launched = true;
            }
        } else {
            mid_st = side_st = SS_ERR_TOO_FEW_SAMPLES;          // get_fft(&[])
        }
    }

    // ---- loudness: the last 16384 interleaved samples, every tick (8x overlap at hop 1024 frames)
    const size_t pos_i = pos_f * s->file_channels;
    const size_t lufs_lb = pos_i > SS_TICK_WINDOW ? pos_i - SS_TICK_WINDOW : 0;
    if (lufs_lb != 0) {
        res->lufs_ran = 1;
        std::memmove(&s->lufs[0], &s->lufs[1], (SS_LUFS_HISTORY - 1) * sizeof(double));
        if (pos_i <= s->n_samples && lufs_lb < s->n_samples) {
            res->fed = 1;
            res->add_status = add_samples_impl(h, s->pcm.p + lufs_lb, SS_TICK_WINDOW, true);
            if (res->add_status == SS_ERR_DEVICE) return SS_ERR_DEVICE;
            if (!h->meter_ok) {
                res->shortterm_status = SS_ERR_INVALID_MODE;
            } else {
                int rc = ring_loudness_enqueue(h, (uint64_t)h->td->host.s100 * 30);
                if (rc) return rc;
                HIPCHK(hipMemcpyAsync(s->stage_d, h->out2.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                st_launched = true;
            }
        }
    }
    if (st_launched || res->fed) HIPCHK(hipStreamSynchronize(h->stream));
    if (fft_launched) HIPCHK(hipStreamSynchronize(s->fft_stream));

    if (res->fft_ran) {
        session_emit_spectrum(s, s->stage, fft_launched ? mid_st : (mid_st ? mid_st : SS_OK), mid_xy,
                              &res->mid_status, &res->n_mid);
        session_emit_spectrum(s, s->stage + s->bin_stride, side_st, side_xy, &res->side_status, &res->n_side);
    }
    if (res->fed) s->lufs[SS_LUFS_HISTORY - 1] = st_launched ? s->stage_d[1] : 0.0;
    res->shortterm = s->lufs[SS_LUFS_HISTORY - 1];
    return SS_OK;
}

// analyze_microphone_input (tui.rs:1427-1480) on one snapshot of the capture ring
int ss_session_tick_capture(ss_session *s, const float *latest, size_t n, double *mid_xy,
                            double *side_xy, size_t cap_pairs, double *wave_xy,
                            size_t wave_cap_pairs, size_t *wave_n, ss_tick_result *res)
{
    SS_ON_DEVICE(s);
    if (wave_n) *wave_n = 0;
    if (!s || s->is_file || !res || !mid_xy || !side_xy || !latest) return SS_ERR_INVALID_ARG;
    if (n != s->n_samples) return SS_ERR_INVALID_ARG;
    if (cap_pairs < s->bt->count || cap_pairs < 1) return SS_ERR_CAPACITY;
    ss_analyzer *h = s->an;
    const size_t pairs = n / 2;                                 // 15 * rate
    size_t window, bins;
    waveform_shape(pairs, 15.0, &window, &bins);
    if (wave_xy && 2 * bins > wave_cap_pairs) return SS_ERR_CAPACITY;
    std::memset(res, 0, sizeof *res);
    res->fft_ran = 1; res->lufs_ran = 1; res->fed = 1;
    HIPCHK(hipMemcpyAsync(s->pcm.p, latest, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    const size_t lb = pairs - SS_TICK_WINDOW;
    // get_fft's value checks on the two 16384-sample slices
    std::vector<std::pair<size_t, uint8_t>> bad;
    for (size_t i = lb; i < pairs; i++) {
        const uint8_t c = pair_class(latest[2 * i], latest[2 * i + 1]);
        if (c) bad.emplace_back(i, c);
    }
    int mid_st = window_value_status(s, lb, SS_TICK_WINDOW, 0, bad);
    int side_st = window_value_status(s, lb, SS_TICK_WINDOW, 2, bad);
    const int lim = (20000.0f > (float)h->rate / 2.0f) ? SS_ERR_FREQ_LIMIT : SS_OK;
    if (!mid_st) mid_st = lim;
    if (!side_st) side_st = lim;
    bool fft_launched = false;
    if ((!mid_st || !side_st) && s->bt->count) {
        int rc = session_enqueue_fft(s, s->pcm.p, lb, h->stream);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(s->stage, s->spec.p, (size_t)2 * s->bin_stride * sizeof(float),
                              hipMemcpyDeviceToHost, h->stream));
        fft_launched = true;
    }
    // microphone_input_chart = get_waveform(&mid_samples, 15.)
    if (wave_xy && bins) {
        HIPCHK(ssk::launch_mid_side(s->pcm.p, pairs, s->ms.p, s->ms.p + pairs, h->stream));
        ssk::WaveParams p{};
        p.pcm = s->ms.p; p.stream_stride = pairs; p.n_samples = pairs; p.n_streams = 1;
        p.window = (uint32_t)window; p.out = s->wave.p; p.out_stride = 2 * bins;
        HIPCHK(ssk::launch_waveform(p, h->stream));
        HIPCHK(hipMemcpyAsync(s->stage + (size_t)2 * s->bin_stride, s->wave.p, 2 * bins * sizeof(float),
                              hipMemcpyDeviceToHost, h->stream));
    }
    // loudness: shift, feed the newest 16384 samples, read short-term
    std::memmove(&s->lufs[0], &s->lufs[1], (SS_LUFS_HISTORY - 1) * sizeof(double));
    res->add_status = add_samples_impl(h, s->pcm.p + (n - SS_TICK_WINDOW), SS_TICK_WINDOW, true);
    if (res->add_status == SS_ERR_DEVICE) return SS_ERR_DEVICE;
    bool st_launched = false;
    if (!h->meter_ok) {
        res->shortterm_status = SS_ERR_INVALID_MODE;
    } else {
        int rc = ring_loudness_enqueue(h, (uint64_t)h->td->host.s100 * 30);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(s->stage_d, h->out2.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        st_launched = true;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    (void)fft_launched;
    session_emit_spectrum(s, s->stage, mid_st, mid_xy, &res->mid_status, &res->n_mid);
    session_emit_spectrum(s, s->stage + s->bin_stride, side_st, side_xy, &res->side_status, &res->n_side);
    if (wave_xy && bins) {
        const float *mm = s->stage + (size_t)2 * s->bin_stride;
        for (size_t i = 0; i < bins; i++) {
            wave_xy[4 * i + 0] = (double)i; wave_xy[4 * i + 1] = (double)mm[2 * i];
            wave_xy[4 * i + 2] = (double)i; wave_xy[4 * i + 3] = (double)mm[2 * i + 1];
        }
        if (wave_n) *wave_n = 2 * bins;
    }
    s->lufs[SS_LUFS_HISTORY - 1] = st_launched ? s->stage_d[1] : 0.0;
    res->shortterm = s->lufs[SS_LUFS_HISTORY - 1];
    return SS_OK;
}

}  // extern "C"
