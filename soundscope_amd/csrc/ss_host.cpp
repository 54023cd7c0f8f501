// ss_host.cpp — shared host-side infrastructure of the C ABI: HIP error channel, per-device table caches,
// device selection, status strings and the inspection of the host-designed tables.  No CPU compute path.
#include "ss_host.h"

namespace ssh {

std::string &last_error()
{
    thread_local std::string text;
    return text;
}

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    last_error() = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

struct Process {
    std::mutex mu;
    bool probed = false;
    int n_devices = 0;
    std::map<int, std::unique_ptr<Ctx>> per_device;
};

SS_HIDDEN Process &process()
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

hipError_t stream_acquire(hipStream_t *out)
{
    Ctx &c = ctx();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (!c.idle_streams.empty()) { *out = c.idle_streams.back(); c.idle_streams.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void stream_release(hipStream_t s)
{
    if (!s) return;
    Ctx &c = ctx();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.idle_streams.size() < 8) { c.idle_streams.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}

Scratch &scratch()
{
    thread_local std::map<int, std::unique_ptr<Scratch>> per_device;
    auto &slot = per_device[current_device()];
    if (!slot) slot = std::make_unique<Scratch>();
    return *slot;
}

int probe_devices()
{
    Process &c = process();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.probed) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) { last_error() = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); n = 0; }
        c.n_devices = n;
        c.probed = true;
    }
    return c.n_devices;
}

int require_device()
{
    if (probe_devices() <= 0) {
        if (last_error().empty()) last_error() = "no HIP device visible";
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
        HIPCHK(t->off4096_dev.upload(std::vector<float>(pf.size(), off)));
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
    for (int s = 0; s < 8; s++) sst::kweight_transition_pow(k.a, (uint64_t)ssk::td_split_chunk_frames(channels, (rate + 5) / 10) << s, k.m_pow_split[s]);
    for (int nstep = 0; nstep < 68; nstep++) sst::kweight_transition_pow(k.a, (uint64_t)nstep, k.m_step_split[nstep]);
    for (int cidx = 0; cidx < 64; cidx++)
        sst::kweight_transition_pow(k.a, (uint64_t)ssk::td_split_chunk_frames(channels, (rate + 5) / 10) * (uint64_t)(cidx + 1), k.m_chunk_split[cidx]);
    for (int cidx = 0; cidx < 64; cidx++)
        sst::kweight_transition_pow(k.a, (uint64_t)ssk::td_chunk_frames(channels, (rate + 5) / 10) * (uint64_t)(cidx + 1), k.m_chunk[cidx]);
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
    {
        uint64_t ring = (uint64_t)rate * 3000 / 1000;                    // the meter's ring (ss_analyzer.cpp: the same rule)
        if (ring % k.s100) ring += k.s100 - ring % k.s100;
        k.st_off = (uint64_t)k.s100 * 30 > ring ? 1u : 0u;
    }
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

int meter_args_ok(uint32_t channels, uint32_t rate)
{
    // EbuR128::new: channels == 0 || > 64, rate < 16 || > 2_822_400 -> Error::NoMem
    if (channels == 0 || channels > 64) return SS_ERR_NOMEM;
    if (rate < 16 || rate > 2822400) return SS_ERR_NOMEM;
    return SS_OK;
}

}  // namespace ssh

namespace ssi {
void set_last_error(const std::string &text) { ssh::last_error() = text; }
}  // namespace ssi

using namespace ssh;

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
const char *ss_last_device_error(void) { return last_error().c_str(); }

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
}  // extern "C"
