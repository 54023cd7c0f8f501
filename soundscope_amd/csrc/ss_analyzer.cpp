// ss_analyzer.cpp — the Analyzer mirror of include/soundscope_hip.h: one entry point per Rust method of
// `pub struct Analyzer` (/root/reference/src/analyzer.rs:29-183) plus get_mid_and_side_samples
// (/root/reference/src/audio_player.rs:400-419).  Host logic only: argument validation with the reference's error
// order, device-resident meter state, launches.  No CPU compute path.
#include "ss_host.h"

using namespace ssh;

namespace ssh {

// EbuR128::new + assignment (analyzer.rs:49-53): the new meter replaces the old one only when every fallible step
// has succeeded — on failure the handle keeps its previous meter (only sample_rate has changed by then).
SS_HIDDEN int handle_make_meter(ss_analyzer *h, uint32_t channels, uint32_t rate)
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
    HIPCHK(ring_scratch.alloc(ssk::kRingScratchDoubles));
    HIPCHK(hipMemset(ring_scratch.p, 0, ssk::kRingScratchDoubles * sizeof(double)));      // (k_ring_energy's completion counter starts at zero)
    std::vector<double> w(channels);
    sst::channel_weights(channels, w.data());
    HIPCHK(weights.upload(w));
    // commit
    h->channels = channels; h->meter_rate = rate; h->tp_factor = tp_factor; h->tp_cfg_applied = h->tp_cfg; h->td = td; h->ring_frames = ring_frames;
    h->state.swap(state); h->hist.swap(hist); h->sub.swap(sub); h->ring.swap(ring); h->counts.swap(counts);
    h->out2.swap(out2); h->ring_scratch.swap(ring_scratch); h->weights.swap(weights);
    h->meter_ok = true;
    h->change_count++;
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
    h->change_count++;
    return SS_OK;
}

int pin_ready(ss_analyzer *h)
{
    if (h->pin_d) return SS_OK;
    // Idempotent per resource: a call that failed part-way leaves what it got in the handle, and the next one allocates only what
    // is still missing (nothing is leaked; ss_analyzer_destroy frees whatever is there).
    auto host_alloc = [](auto **slot, size_t bytes) -> hipError_t {
        return *slot ? hipSuccess : hipHostMalloc(reinterpret_cast<void **>(slot), bytes, hipHostMallocDefault);
    };
    auto event = [](hipEvent_t *e) -> hipError_t { return *e ? hipSuccess : hipEventCreateWithFlags(e, hipEventDisableTiming); };
    for (int i = 0; i < 2; i++) {
        HIPCHK(host_alloc(&h->pin_in[i], ss_analyzer::kPinFloats * sizeof(float)));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->pin_in_dev[i]), h->pin_in[i], 0));
        HIPCHK(event(&h->pin_ev[i]));
    }
    HIPCHK(event(&h->pin_ev[2]));                                                // behind a short-term / momentary reading
    HIPCHK(host_alloc(&h->pin_out, (ss_analyzer::kPinFloats / 2 + 4) * sizeof(float)));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->pin_out_dev), h->pin_out, 0));
    HIPCHK(host_alloc(&h->pin_peaks, 2 * ssk::kMaxChannels * sizeof(float)));
    HIPCHK(host_alloc(&h->pin_eval, 2 * sizeof(double)));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->pin_eval_dev), h->pin_eval, 0));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->pin_peaks_dev), h->pin_peaks, 0));
    if (!h->pin_flag) {
        HIPCHK(host_alloc(&h->pin_flag, sizeof(uint32_t)));
        *h->pin_flag = 0u;
    }
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->pin_flag_dev), h->pin_flag, 0));
    HIPCHK(host_alloc(&h->pin_d_pending, 2 * sizeof(double)));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->pin_d_dev), h->pin_d_pending, 0));
    h->pin_d = h->pin_d_pending;                // (last: "ready" means all of them)
    h->pin_d_pending = nullptr;
    return SS_OK;
}

int pin_acquire(ss_analyzer *h, int *idx)
{
    int rc = pin_ready(h);
    if (rc) return rc;
    const int i = h->pin_next;
    if (h->pin_busy[i]) { HIPCHK(hipEventSynchronize(h->pin_ev[i])); h->pin_busy[i] = false; }
    h->pin_next = i ^ 1;
    *idx = i;
    return SS_OK;
}

void pin_all_free(ss_analyzer *h) { h->pin_busy[0] = h->pin_busy[1] = false; }

int attach_readings(ss_analyzer *h, ssk::FinalizeParams *gating)
{
    if (!h->meter_ok || !gating || gating->n_streams != 1) return SS_OK;
#ifdef SS_TUNING
    { static const bool off = std::getenv("SS_NO_PREFETCH") != nullptr; if (off) return SS_OK; }
#endif
    int rc = pin_ready(h);
    if (rc) return rc;
    h->readings_seq = h->readings_seq + 1u ? h->readings_seq + 1u : 1u;
    gating->readings_out = h->pin_eval_dev;
    gating->readings_peaks_src = &h->state.p->sample_peak[0]; gating->readings_peaks_dst = h->pin_peaks_dev;
    gating->readings_flag = h->pin_flag_dev; gating->readings_seq = h->readings_seq;
    h->prefetch_stamp = h->change_count;
    return SS_OK;
}

int prefetch_readings(ss_analyzer *h, bool on_demand)
{
    if (!h->meter_ok) return SS_OK;
#ifdef SS_TUNING        // development builds only: SS_NO_PREFETCH=1 leaves the readings to the getters (A/B of the tick)
    { static const bool off = std::getenv("SS_NO_PREFETCH") != nullptr; if (off && !on_demand) return SS_OK; }
#endif
    (void)on_demand;
    const double *he, *hb;
    int rc = get_hist_tables(&he, &hb);
    if (rc) return rc;
    rc = pin_ready(h);
    if (rc) return rc;
    static_assert(offsetof(ssk::TdState, true_peak) == offsetof(ssk::TdState, sample_peak) + sizeof(float) * ssk::kMaxChannels,
                  "sample_peak and true_peak are read as one block");
    // ONE launch, nothing else: the kernel copies the peaks beside its evaluation and stores the launch's number into a flag in
    // pinned memory behind everything — what the getter waits for (no event, no copy command: the launch has to fit the few
    // microseconds a tick has left between its charts and the end of the loudness call)
    h->readings_seq = h->readings_seq + 1u ? h->readings_seq + 1u : 1u;
    const ssk::ReadingsExtra x{&h->state.p->sample_peak[0], h->pin_peaks_dev, h->pin_flag_dev, h->readings_seq};
    HIPCHK(ssk::launch_hist_eval(h->hist.p, he, hb, h->pin_eval_dev, h->stream, &x));
    h->prefetch_stamp = h->change_count;
    return SS_OK;
}

}  // namespace ssh

extern "C" {

int ss_analyzer_create(uint32_t channels, uint32_t rate, ss_analyzer **out)
{
    if (!out) return SS_ERR_INVALID_ARG;
    *out = nullptr;
    if (require_device()) return SS_ERR_DEVICE;
    // destroyed (stream included) on every early return
    std::unique_ptr<ss_analyzer, decltype(&ss_analyzer_destroy)> h(new ss_analyzer(), &ss_analyzer_destroy);
    h->device = current_device();
    h->rate = rate;
    HIPCHK(stream_acquire(&h->stream));
    HIPCHK(h->in.alloc(32768));
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
    if (h->stream) { (void)hipStreamSynchronize(h->stream); stream_release(h->stream); }
    for (int i = 0; i < 2; i++) if (h->pin_in[i]) (void)hipHostFree(h->pin_in[i]);
    for (int i = 0; i < 3; i++) if (h->pin_ev[i]) (void)hipEventDestroy(h->pin_ev[i]);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_peaks) (void)hipHostFree(h->pin_peaks);
    if (h->pin_eval) (void)hipHostFree(h->pin_eval);
    if (h->pin_flag) (void)hipHostFree(h->pin_flag);
    if (h->pin_d) (void)hipHostFree(h->pin_d);
    if (h->pin_d_pending) (void)hipHostFree(h->pin_d_pending);
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

int ss_analyzer_set_true_peak_arith(ss_analyzer *h, int arith)
{
    if (!h || (arith != SS_TP_ARITH_F16X3 && arith != SS_TP_ARITH_F32)) return SS_ERR_INVALID_ARG;
    h->tp_arith = arith;                          // read by the next ss_add_samples / tick (a launch parameter, no state behind it)
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
    {
        // w[i] == 0 turns an infinite sample into NaN (0 * inf); only the first few
        // window entries can be exactly zero
        bool any_nan = false, any_inf = false;
        const std::vector<float> *win = nullptr;
        std::vector<float> win_local;
        // (first a branch-free pass: is there a NaN or an infinity at all?  An all-ones exponent; normally there is none and
        // the classifying loop below — a branch per sample — is skipped)
        uint32_t non_finite = 0;
        for (size_t i = 0; i < n; i++) {
            uint32_t u;
            std::memcpy(&u, samples + i, sizeof u);
            non_finite |= (uint32_t)((u & 0x7F800000u) == 0x7F800000u);
        }
        for (size_t i = 0; non_finite && i < n; i++) {
            const float x = samples[i];
            if (std::isnan(x)) { any_nan = true; continue; }
            if (!std::isinf(x)) continue;
            if (!win) {
                if (pow2 && n <= 32768) {
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
    if (20000.0f > (float)h->rate / 2.0f) {
        // FrequencyLimit::Range(20., 20000.).verify: InvalidFrequencyLimit(ValueAboveNyquist(max)); the payload is the limit
        // (the Nyquist frequency rides along as the second value)
        h->fft_err_a = 20000.0f; h->fft_err_b = (float)h->rate / 2.0f;
        return SS_ERR_FREQ_LIMIT;
    }
    // the crate's transform itself: microfft's real FFTs end at 32768 points, spectrum-analyzer panics beyond — behind every
    // input check above (a NaN in 65536 samples is still reported as a NaN)
    if (n > 32768) return SS_ERR_UNSUPPORTED;

    FftTables *ft; BinTables *bt;
    int rc = get_fft_tables(n, &ft);
    if (rc) return rc;
    rc = get_bin_tables(h->rate, n, &bt);
    if (rc) return rc;
    if (bt->count > cap_pairs) return SS_ERR_CAPACITY;
    if (bt->count == 0) return SS_OK;

    // the window goes into page-locked memory the kernel reads in place; the dB row comes back the same way
    int pin = 0;
    rc = pin_acquire(h, &pin);
    if (rc) return rc;
    std::memcpy(h->pin_in[pin], samples, n * sizeof(float));
    // (the transform reads its window many times in small pieces: from page-locked host memory in place that costs the kernel
    // 20 us more than it takes from HBM — one DMA of the page-locked copy first)
    HIPCHK(hipMemcpyAsync(h->in.p, h->pin_in[pin], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    ssk::FftBatchParams p{};
    p.pcm = h->in.p; p.out = h->pin_out_dev;
    p.window = ft->window.p; p.half_window = ft->half_window.p;
    p.tw_n = ft->tw_n.p; p.tw_256 = ft->tw_256.p; p.pink = nullptr;
    p.frames_per_stream = n; p.first_start = 0; p.n_streams = 1; p.channels = 1;
    p.n_windows = 1; p.hop = 0; p.n = (uint32_t)n;
    p.first_bin = (uint32_t)bt->first; p.n_bins = (uint32_t)bt->count; p.bin_stride = p.n_bins; p.windows_per_block = 1;
    p.db_offset = (float)(20.0 * std::log10(4.0 / (double)n));
    if (n == 16384) {
        p.tw_core = ft->core_tw4096; p.tw_256 = ft->core_tw256;
        HIPCHK(ssk::launch_fft16k(p, 0, h->stream));
    } else if (n == 4096) {
        // the radix-16 machine of the batch path on one window: k_fft4096_pairw with its second window absent
        // (through round 3 this size took k_fft_generic's radix-2 passes)
        p.hop = 1024; p.windows_per_block = 2;
        p.bin_stride = (uint32_t)((bt->count + 3) & ~(size_t)3);
        p.db_offset = (float)(10.0 * std::log10(4.0 / (4096.0 * 4096.0)));
        p.offpink = bt->off4096_dev.p;
        p.publish_mask = ssk::fft4096_publish_mask(p.first_bin, p.n_bins);
        HIPCHK(ssk::launch_fft4096_pairw(p, 0, h->stream));
    } else {
        HIPCHK(ssk::launch_fft_generic(p, 0, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    pin_all_free(h);
    const float *db = h->pin_out;
    for (size_t i = 0; i < bt->count; i++)
        if (std::isnan(db[i]) || std::isinf(db[i])) {
            // ScalingError(original, scaled) of the first bin the scaling function spoiled.  scale_to_dbfs maps a finite
            // magnitude to a finite value (0 -> -150), so the magnitude itself was +inf (its square overflowed; scaled = +inf)
            // or NaN (inf - inf inside the transform; scaled = NaN): the dB value read back IS both payload values.
            h->fft_err_a = db[i]; h->fft_err_b = db[i];
            return SS_ERR_SCALING;
        }
    // analyzer.rs:75-102 in f64: + pink compensation, log-x chart position
    for (size_t i = 0; i < bt->count; i++) {
        out_xy[2 * i] = bt->chart_x[i];
        out_xy[2 * i + 1] = (double)db[i] + bt->pink[i];
    }
    if (out_n) *out_n = bt->count;
    return SS_OK;
}

int ss_get_fft_error_values(const ss_analyzer *h, float *a, float *b)
{
    if (!h || !a || !b) return SS_ERR_INVALID_ARG;
    *a = h->fft_err_a; *b = h->fft_err_b;
    return SS_OK;
}

// get_waveform's shape (analyzer.rs:108-118): W = (window_s * 1000.) as usize, and the number of
// bins whose start floor(i * spp) is still inside the buffer
}  // extern "C"
void ssh::waveform_shape(size_t n, double waveform_window, size_t *window_out, size_t *bins_out)
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
extern "C" {

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

// add_frames_f32 on the handle's meter.  on_device: `samples` already lives in HBM (tick drivers):
// no staging copy and no synchronisation — everything is only enqueued on the handle's stream.
}  // extern "C"
int ssh::add_samples_impl(ss_analyzer *h, const float *samples, size_t n, bool on_device, ssk::FinalizeParams *deferred,
                          TickExtras *tick)
{
    if (deferred) deferred->n_streams = 0;
    if (tick) { tick->fused = false; tick->st_fused = false; }
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
    h->change_count++;                                  // (whatever happens below: cached readings are of the past)
    // a tick-sized host buffer: copied into page-locked memory the kernel reads in place, and the call returns behind its
    // launches (the event tells the next user of that buffer when the kernel is through with it)
    int pin = -1;
    if (!on_device && n <= ss_analyzer::kPinFloats && frames <= piece_frames) {
        rc = pin_acquire(h, &pin);
        if (rc) return rc;
        std::memcpy(h->pin_in[pin], samples, n * sizeof(float));
        // ... and on to HBM by one DMA (reading the page-locked copy in place over PCIe cost the kernel 33 instead of 22 us)
        HIPCHK(h->in.ensure(n));
        HIPCHK(hipMemcpyAsync(h->in.p, h->pin_in[pin], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
        samples = h->in.p;
        on_device = true;
    }
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
        p.tp_f32 = h->tp_arith == SS_TP_ARITH_F32 ? 1u : 0u;           // default: ebur128's f32 interpolator width (the handle IS the reference's Analyzer)
        const bool with_tick = tick && tick->fft && on_device && take == frames;
        const uint64_t st_frames = S * 30;                              // the short-term window (3 s)
        if (with_tick && tick->shortterm_out && take <= st_frames && st_frames <= h->ring_frames &&
            h->ring_frames * C < (1ull << 31)) {
            // the window ends with this call: frames [fed + take - st_frames, fed + take); the part in front of the call is the
            // ring workgroups' (frames before 0 are the zeroed ring)
            const uint64_t end_new = h->frames_fed + take;
            const uint64_t begin = (end_new % h->ring_frames + h->ring_frames - st_frames) % h->ring_frames;
            p.st_out = tick->shortterm_out; p.st_scratch = h->ring_scratch.p + ssk::kRingTickScratch; p.st_weights = h->weights.p;
            p.st_frames = (double)st_frames;
            p.st_begin_elem = (uint32_t)(begin * C); p.st_old_total = (uint32_t)((st_frames - take) * C);
            p.st_blocks = ssk::kRingTickBlocks;
        }
        HIPCHK(ssk::launch_time_domain(p, h->stream, with_tick ? tick->fft : nullptr, with_tick ? &tick->fused : nullptr));
        if (with_tick) tick->st_fused = tick->fused && p.st_out != nullptr;
        const uint64_t sb0 = h->frames_fed / S, sb1 = (h->frames_fed + take) / S;
        if (sb1 > sb0) {
            ssk::FinalizeParams f{};
            f.k = h->td->dev.p; f.subblocks = h->sub.p; f.sub_stride = 0; f.sub_cap = ss_analyzer::kSubCap;
            f.hist_energies = he; f.hist_bounds = hb; f.weights = h->weights.p;
            f.hist = h->hist.p; f.corpus_hist = nullptr; f.n_streams = 1; f.channels = C;
            f.sub_begin = sb0; f.sub_end = sb1;
            f.out_integrated = nullptr; f.out_lra = nullptr; f.out_counts = h->counts.p;
            f.state = h->state.p;
            // a caller that has something shorter to put in front (a tick's short-term reading) launches the gating of a
            // single-piece call itself, on the same stream
            if (deferred && on_device && take == frames) *deferred = f;
            else HIPCHK(ssk::launch_finalize(f, h->stream));
        }
        // the staging buffer is reused by the next piece
        if (!on_device) HIPCHK(hipStreamSynchronize(h->stream));
        h->frames_fed += take;
        done += take;
    }
    if (pin >= 0) {
        HIPCHK(hipEventRecord(h->pin_ev[pin], h->stream));
        h->pin_busy[pin] = true;
    }
    return SS_OK;
}
extern "C" {

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
}  // extern "C"
int ssh::ring_loudness_enqueue(ss_analyzer *h, uint64_t frames, double *out2_dev)
{
    SS_ON_DEVICE(h);
    HIPCHK(ssk::launch_ring_energy(h->ring.p, h->ring_frames, h->channels, h->frames_fed, frames,
                                   h->weights.p, out2_dev ? out2_dev : h->out2.p, h->ring_scratch.p, h->stream));
    return SS_OK;
}
extern "C" {

static int ring_loudness(ss_analyzer *h, uint64_t frames, double *out)
{
    SS_ON_DEVICE(h);
    if (!h || !out) return SS_ERR_INVALID_ARG;
    if (!h->meter_ok) return SS_ERR_INVALID_MODE;
    if (frames > h->ring_frames) return SS_ERR_INVALID_MODE;
    int rc = pin_ready(h);
    if (rc) return rc;
    rc = ring_loudness_enqueue(h, frames, h->pin_d_dev);        // (energy, loudness) straight into page-locked memory
    if (rc) return rc;
    // the call waits for its own reading only; behind it the readings the reference's render loop asks for on its next frame
    // (integrated loudness, range, peaks: tui.rs:917, :950, :969) are put on their way
    HIPCHK(hipEventRecord(h->pin_ev[2], h->stream));
    if (h->prefetch_stamp != h->change_count) { rc = prefetch_readings(h); if (rc) return rc; }
    HIPCHK(hipEventSynchronize(h->pin_ev[2]));
    *out = h->pin_d[1];
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

// integrated loudness + loudness range (one histogram evaluation) and every channel's peaks (one copy), behind ONE wait: the
// readings of the current meter state, kept in the handle until the state moves
static int refresh_readings(ss_analyzer *h)
{
    if (h->eval_stamp == h->change_count && h->peaks_stamp == h->change_count) return SS_OK;
    if (h->prefetch_stamp != h->change_count) {         // nobody has asked for this state's readings yet
        int rc = prefetch_readings(h, true);
        if (rc) return rc;
    }
    // the flag (bounded polling: the launch may still be running, or the memory may deliver the flag with the kernel's end),
    // then — for errors, and as the fallback — the stream
    bool seen = false;
    for (int spin = 0; spin < 4096 && !seen; spin++) {
        seen = __atomic_load_n(h->pin_flag, __ATOMIC_ACQUIRE) == h->readings_seq;
#if defined(__x86_64__) || defined(__i386__)
        if (!seen) __builtin_ia32_pause();
#endif
    }
    if (!seen) { HIPCHK(hipStreamSynchronize(h->stream)); pin_all_free(h); }
    h->eval_cache[0] = h->pin_eval[0]; h->eval_cache[1] = h->pin_eval[1];
    std::memcpy(h->peaks_cache, h->pin_peaks, sizeof h->peaks_cache);
    h->eval_stamp = h->peaks_stamp = h->change_count;
    return SS_OK;
}

static int hist_eval(ss_analyzer *h, double r[2])
{
    SS_ON_DEVICE(h);
    if (!h) return SS_ERR_INVALID_ARG;
    if (!h->meter_ok) return SS_ERR_INVALID_MODE;
    int rc = refresh_readings(h);
    if (rc) return rc;
    r[0] = h->eval_cache[0]; r[1] = h->eval_cache[1];
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
    int rc = refresh_readings(h);
    if (rc) return rc;
    const float sp = h->peaks_cache[ch], tp = h->peaks_cache[ssk::kMaxChannels + ch];
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

int ss_inspect_filter_state(ss_analyzer *h, uint32_t channel, double v4[4])
{
    SS_ON_DEVICE(h);
    if (!h || !v4) return SS_ERR_INVALID_ARG;
    if (!h->meter_ok) return SS_ERR_INVALID_MODE;
    if (channel >= h->channels) return SS_ERR_INVALID_CHANNEL;
    {   // a channel the crate maps to Channel::Unused is not filtered there at all: its state stays what reset left.  (The device
        // runs the recurrence on every channel — the lanes are there anyway — and no reading ever looks at such a channel.)
        std::vector<double> w(h->channels);
        sst::channel_weights(h->channels, w.data());
        if (w[channel] == 0.0) { v4[0] = v4[1] = v4[2] = v4[3] = 0.0; return SS_OK; }
    }
    HIPCHK(hipMemcpyAsync(v4, &h->state.p->v[channel][0], 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SS_OK;
}

uint32_t ss_sample_rate(const ss_analyzer *h) { return h ? h->rate : 0; }

}  // extern "C"
