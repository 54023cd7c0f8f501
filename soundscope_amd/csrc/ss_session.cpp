// ss_session.cpp — the tick drivers: the per-file / per-device state of the reference's App and its
// per-tick analysis with the audio resident in HBM (/root/reference/src/tui.rs:1207-1241, :1427-1552, :1586-1614).
#include "ss_host.h"

using namespace ssh;

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
    // capture, resident ring (ss_session_capture_push / ss_session_tick_capture_resident): what the capture callback has pushed
    // since the last tick waits in page-locked memory; a tick shifts the ring by that much on the device (into pcm_alt, then the two
    // swap) and uploads only the new samples
    DevBuf<float> pcm_alt;
    float *pending = nullptr;           // pinned, n_samples floats: the newest pushed samples, oldest first
    size_t pending_n = 0;
    std::vector<float> tail;            // host copy of the ring's newest 2 * SS_TICK_WINDOW samples (the crate's value checks)
    DevBuf<float> wave;                 // file open: [bins][2] of the whole-file chart (released behind it)
    hipEvent_t ev_tick = nullptr;       // behind a file tick's last result (the gating of the new sub-blocks runs after it)
    float *stage = nullptr;             // pinned: 2 * bin_stride floats (the two dB rows of a tick) | capture: [bins][2] chart floats
    double *stage_d = nullptr;          // pinned: short-term loudness (2 doubles)
    float *stage_dev = nullptr;         // the same two, as the device sees them: the file tick's kernels write their results
    double *stage_d_dev = nullptr;      // straight into the pinned memory (no copy launch behind them)
    uint32_t *row_flag = nullptr, *row_flag_dev = nullptr;      // pinned: [mid, side] = the tick whose row stands in `stage`
    uint32_t tick_seq = 0;              // file ticks so far (the value the spectrum's workgroups store into row_flag)
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
    // Analyzer::default() followed by create_loudness_meter(channels, rate) (tui.rs:1217-1221).  Where the second call will succeed
    // the default's 2-channel 44.1 kHz meter is never observable, so the handle is built at the file's shape straight away (the
    // detour cost a 2.3 MB ring allocated, freed — 0.2 ms — and allocated again); where it will fail the reference's order is kept:
    // the rate sticks, the default meter stays (analyzer.rs:50), the reference only reports the error and carries on
    int rc;
    if (meter_args_ok(meter_channels, rate) == SS_OK) {
        rc = ss_analyzer_create(meter_channels, rate, &s->an);
        if (rc) return rc;
    } else {
        rc = ss_analyzer_create(2, 44100, &s->an);                 // Analyzer::default()
        if (rc) return rc;
        (void)ss_analyzer_configure(s->an, meter_channels, rate);
    }
    rc = get_fft_tables(SS_TICK_WINDOW, &s->ft);
    if (rc) return rc;
    rc = get_bin_tables(rate, SS_TICK_WINDOW, &s->bt);
    if (rc) return rc;
    s->bin_stride = (uint32_t)((s->bt->count + 3) & ~(size_t)3);
    if (s->bin_stride == 0) s->bin_stride = 4;
    HIPCHK(hipEventCreateWithFlags(&s->ev_tick, hipEventDisableTiming));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->stage_d), 2 * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->stage_d_dev), s->stage_d, 0));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->row_flag), 2 * sizeof(uint32_t), hipHostMallocDefault));
    s->row_flag[0] = s->row_flag[1] = 0u;
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->row_flag_dev), s->row_flag, 0));
    return SS_OK;
}

int session_stage(ss_session *s, size_t floats)
{
    if (floats <= s->stage_floats) return SS_OK;
    if (s->stage) (void)hipHostFree(s->stage);
    s->stage = nullptr; s->stage_dev = nullptr; s->stage_floats = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->stage), floats * sizeof(float), hipHostMallocDefault));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->stage_dev), s->stage, 0));
    s->stage_floats = floats;
    return SS_OK;
}

// the mid/side spectrum of pairs [lb, lb + 16384) of an interleaved pair buffer; the two dB rows go to `out`
// (the session's device rows, or the pinned stage as the device sees it)
ssk::FftBatchParams session_fft_params(const ss_session *s, const float *pairs, size_t lb, float *out)
{
    ssk::FftBatchParams p{};
    p.pcm = pairs; p.out = out;
    p.window = s->ft->window.p; p.half_window = s->ft->half_window.p;
    p.tw_n = s->ft->tw_n.p; p.tw_core = s->ft->core_tw4096; p.tw_256 = s->ft->core_tw256; p.pink = nullptr;
    p.frames_per_stream = 0; p.first_start = lb; p.n_streams = 1; p.channels = 2;
    p.n_windows = 1; p.hop = 0; p.n = SS_TICK_WINDOW;
    p.first_bin = (uint32_t)s->bt->first; p.n_bins = (uint32_t)s->bt->count; p.bin_stride = s->bin_stride;
    p.windows_per_block = 1;
    p.db_offset = (float)(20.0 * std::log10(4.0 / (double)SS_TICK_WINDOW));
    return p;
}

// wait (bounded: about 200 us) until a spectrum workgroup has stored `seq` behind its row
bool wait_row_flag(const uint32_t *flag, uint32_t seq)
{
    for (int spin = 0; spin < 64; spin++) {
        for (int k = 0; k < 256; k++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
    }
    return false;
}

// before the synchronisation: the x halves of a chart whose row is on its way (they are the session's constants)
void session_emit_x(const ss_session *s, int status_in, double *xy)
{
    if (status_in) return;
    const size_t nb = s->bt->count;
    const double *cx = s->bt->chart_x.data();
    for (size_t i = 0; i < nb; i++) xy[2 * i] = cx[i];
}

// after the synchronisation: dB rows in the pinned stage -> (chart_x, dB + pink) pairs, or the (0,0) fallback
// (x_done: session_emit_x has written the x halves already)
void session_emit_spectrum(const ss_session *s, const float *row, int status_in, double *xy,
                           int32_t *status_out, uint32_t *n_out, bool x_done = false)
{
    int st = status_in;
    const size_t nb = s->bt->count;
    if (!st) {
        // NaN or infinite anywhere in the row?  (an all-ones exponent; one pass without branches)
        uint32_t bad = 0;
        for (size_t i = 0; i < nb; i++) {
            uint32_t u;
            std::memcpy(&u, row + i, sizeof u);
            bad |= (uint32_t)((u & 0x7F800000u) == 0x7F800000u);
        }
        if (bad) st = SS_ERR_SCALING;
    }
    if (st) {
        xy[0] = 0.0; xy[1] = 0.0;                    // vec![(0., 0.)] (tui.rs:1437-1452, :1505-1524)
        *n_out = 1;
    } else {
        const double *pk = s->bt->pink.data();
        if (x_done) {
            for (size_t i = 0; i < nb; i++) xy[2 * i + 1] = (double)row[i] + pk[i];
        } else {
            const double *cx = s->bt->chart_x.data();
            for (size_t i = 0; i < nb; i++) {
                xy[2 * i] = cx[i];
                xy[2 * i + 1] = (double)row[i] + pk[i];
            }
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
    if (s->an) ss_analyzer_destroy(s->an);
    if (s->ev_tick) (void)hipEventDestroy(s->ev_tick);
    if (s->stage) (void)hipHostFree(s->stage);
    if (s->stage_d) (void)hipHostFree(s->stage_d);
    if (s->row_flag) (void)hipHostFree(s->row_flag);
    if (s->pending) (void)hipHostFree(s->pending);
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
    // the pairs whose mid or side value is NaN or infinite (normally none), looked for on the device — the file is there anyway,
    // and the host's loop over every pair was most of what an open cost (0.9 of 1.45 ms for 12 s, half of a ten-minute file's)
    constexpr uint32_t kBadCap = 4096;
    DevBuf<uint32_t> bad_count;
    DevBuf<ssk::NonFinitePair> bad_list;
    HIPCHK(bad_count.alloc(1));
    HIPCHK(bad_list.alloc(kBadCap));
    HIPCHK(ssk::launch_nonfinite_pairs(s->pcm.p, pairs, bad_count.p, bad_list.p, kBadCap, h->stream));
    uint32_t n_bad = 0;
    // (the copy below targets this stack variable: whatever way this function is left, the stream is drained first)
    struct DrainOnExit { hipStream_t st; ~DrainOnExit() { (void)hipStreamSynchronize(st); } } drain_guard{h->stream};
    HIPCHK(hipMemcpyAsync(&n_bad, bad_count.p, sizeof n_bad, hipMemcpyDeviceToHost, h->stream));
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
    if (n_bad > kBadCap) {                               // (a file full of them: the list did not hold all, the host looks itself)
        for (size_t i = 0; i < pairs; i++) {
            const uint8_t c = pair_class(interleaved[2 * i], interleaved[2 * i + 1]);
            if (c) s->bad.emplace_back(i, c);
        }
    } else if (n_bad) {
        std::vector<ssk::NonFinitePair> got(n_bad);
        HIPCHK(hipMemcpy(got.data(), bad_list.p, n_bad * sizeof(ssk::NonFinitePair), hipMemcpyDeviceToHost));
        for (const auto &e : got) s->bad.emplace_back((size_t)e.index, (uint8_t)e.cls);
        std::sort(s->bad.begin(), s->bad.end());
    }
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
    HIPCHK(hipMemset(s->pcm.p, 0, s->n_samples * sizeof(float)));      // the reference's ring starts full of zeros (tui.rs:1783-1784)
    s->tail.assign((size_t)2 * SS_TICK_WINDOW, 0.0f);
    // what the capture callback pushes between two ticks: page-locked, allocated HERE — ss_session_capture_push stands in for the
    // audio callback's audio_buf.extend (a real-time thread) and must not call into the driver
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->pending), s->n_samples * sizeof(float), hipHostMallocDefault));
    size_t window, bins;
    waveform_shape(s->n_samples / 2, 15.0, &window, &bins);
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
#ifdef SS_TUNING        // development builds only: where a tick's wall time goes on the host side (tools/probe_tick_host.py)
#include <chrono>
static double g_tick_prof[8];
static inline double tick_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define SS_TICK_T(i) do { const double n_ = tick_now(); g_tick_prof[i] += n_ - t_; t_ = n_; } while (0)
extern "C" void ss_debug_tick_prof(double *out8, int reset) { for (int i = 0; i < 8; i++) { out8[i] = g_tick_prof[i]; if (reset) g_tick_prof[i] = 0.0; } }
#else
#define SS_TICK_T(i)
#endif

int ss_session_tick_file(ss_session *s, size_t pos, double *mid_xy, double *side_xy,
                         size_t cap_pairs, ss_tick_result *res)
{
    SS_ON_DEVICE(s);
#ifdef SS_TUNING
    double t_ = tick_now();
#endif
    if (!s || !s->is_file || !res || !mid_xy || !side_xy) return SS_ERR_INVALID_ARG;
    if (cap_pairs < s->bt->count || cap_pairs < 1) return SS_ERR_CAPACITY;
    ss_analyzer *h = s->an;
    std::memset(res, 0, sizeof *res);
    const size_t pos_f = pos / s->file_channels;
    res->playhead = pos_f;
    bool fft_launched = false, st_launched = false;
    int mid_st = SS_OK, side_st = SS_OK;

    // ---- what the tick will run (host-side checks only)
    // spectrum: the last 16384 mid / side samples before the playhead
    const size_t fft_lb = pos_f > SS_TICK_WINDOW ? pos_f - SS_TICK_WINDOW : 0;     // saturating_sub
    bool fft_wanted = false;
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
            fft_wanted = (!mid_st || !side_st) && s->bt->count;
        } else {
            mid_st = side_st = SS_ERR_TOO_FEW_SAMPLES;          // get_fft(&[])
        }
    }

    SS_TICK_T(0);
    // ---- one stream, one wait.  The spectrum's two workgroups ride the loudness call's launch (k_tick), the short-term reading
    // waits for it, an event behind that is what the tick waits for, and the gating of the new sub-blocks (histograms, block
    // counts: nothing a tick reads) runs after the event — long done when the next tick arrives.  (Two streams did the same
    // until round 4 — when their hardware queues happened to sit on one pipe of the command processor the spectrum did not
    // start before the time-domain kernel had finished: 98 instead of 65 us for the whole life of such a session,
    // tools/probe_tick_queues.sh.)  Results land in pinned memory straight from the kernels.
    bool any_launch = false;
    ssk::FftBatchParams fft_p = fft_wanted ? session_fft_params(s, s->pcm.p, fft_lb, s->stage_dev) : ssk::FftBatchParams{};
    const uint32_t seq = ++s->tick_seq ? s->tick_seq : ++s->tick_seq;          // (never 0: the flags' initial value)
    fft_p.done_flag = s->row_flag_dev; fft_p.done_value = seq;
    // loudness: the last 16384 interleaved samples, every tick (8x overlap at hop 1024 frames)
    const size_t pos_i = pos_f * s->file_channels;
    const size_t lufs_lb = pos_i > SS_TICK_WINDOW ? pos_i - SS_TICK_WINDOW : 0;
    ssk::FinalizeParams gating{};
    // The gating of this tick's new sub-blocks is handed back by add_samples_impl AFTER frames_fed has advanced: if the tick is
    // left early (a failed launch, an event error) it is still launched — otherwise those sub-blocks would be missing from the
    // histograms for the rest of the session.
    struct DeferredGating {
        ss_analyzer *h; ssk::FinalizeParams *g; bool launched = false;
        ~DeferredGating() { if (!launched && g->n_streams) (void)ssk::launch_finalize(*g, h->stream); }
    } gating_guard{h, &gating};
    if (lufs_lb != 0) {
        res->lufs_ran = 1;
        std::memmove(&s->lufs[0], &s->lufs[1], (SS_LUFS_HISTORY - 1) * sizeof(double));
        if (pos_i <= s->n_samples && lufs_lb < s->n_samples) {
            res->fed = 1;
            TickExtras extras;
            extras.fft = fft_wanted ? &fft_p : nullptr;
            extras.shortterm_out = s->stage_d_dev;
            res->add_status = add_samples_impl(h, s->pcm.p + lufs_lb, SS_TICK_WINDOW, true, &gating, &extras);
            if (res->add_status == SS_ERR_DEVICE) return SS_ERR_DEVICE;
            if (res->add_status == SS_OK) any_launch = true;
            if (extras.fused) fft_launched = true;
            if (extras.st_fused) st_launched = true;
        }
    }
    SS_TICK_T(1);
    if (fft_wanted && !fft_launched) {                  // (no loudness call this tick, or one that could not take the spectrum along)
        HIPCHK(ssk::launch_fft16k(fft_p, 1, h->stream));
        fft_launched = true; any_launch = true;
    }
    SS_TICK_T(2);
    if (res->fed && !st_launched) {                     // (the reading did not ride the tick launch)
        if (!h->meter_ok) {
            res->shortterm_status = SS_ERR_INVALID_MODE;
        } else {
            // (energy, loudness) written by the kernel into the pinned pair itself
            int rc = ring_loudness_enqueue(h, (uint64_t)h->td->host.s100 * 30, s->stage_d_dev);
            if (rc) return rc;
            st_launched = true; any_launch = true;
        }
    }
    if (any_launch) HIPCHK(hipEventRecord(s->ev_tick, h->stream));
    // behind the event: the gating — and riding it, what the render loop asks the file analyzer for on its next frame
    // (tui.rs:917, :950, :969: integrated loudness, range, peaks)
    if (gating.n_streams) {
        int rc = attach_readings(h, &gating);
        if (rc) return rc;
        gating_guard.launched = true;
        HIPCHK(ssk::launch_finalize(gating, h->stream));
    }
    SS_TICK_T(3);
    // while the device works: the x halves of the two charts (they do not depend on it)
    if (res->fft_ran) {
        session_emit_x(s, fft_launched ? mid_st : (mid_st ? mid_st : SS_OK), mid_xy);
        session_emit_x(s, side_st, side_xy);
    }
    SS_TICK_T(4);
    // The spectrum's two workgroups are done long before the loudness call's: each tells so through a flag in pinned memory
    // behind its row, and the y halves of its chart are written while the rest of the launch is still running.  (A bounded
    // wait: if the flags do not show up — memory that is not host-coherent would deliver them with the end of the kernel —
    // the rows are taken behind the event as before.)
    bool mid_done = false, side_done = false;
    if (res->fft_ran && fft_launched) {
        if (wait_row_flag(&s->row_flag[0], seq)) {
            session_emit_spectrum(s, s->stage, mid_st, mid_xy, &res->mid_status, &res->n_mid, true);
            mid_done = true;
            if (wait_row_flag(&s->row_flag[1], seq)) {
                session_emit_spectrum(s, s->stage + s->bin_stride, side_st, side_xy, &res->side_status, &res->n_side, true);
                side_done = true;
            }
        }
    }
    SS_TICK_T(5);
    // (a tick that completed no sub-block has no gating launch to ride: the readings get their own)
    if (res->fed && res->add_status == SS_OK && h->prefetch_stamp != h->change_count) { int rc = prefetch_readings(h); if (rc) return rc; }
    if (any_launch) HIPCHK(hipEventSynchronize(s->ev_tick));
    SS_TICK_T(6);
    if (res->fft_ran) {
        if (!mid_done)
            session_emit_spectrum(s, s->stage, fft_launched ? mid_st : (mid_st ? mid_st : SS_OK), mid_xy,
                                  &res->mid_status, &res->n_mid, true);
        if (!side_done)
            session_emit_spectrum(s, s->stage + s->bin_stride, side_st, side_xy, &res->side_status, &res->n_side, true);
    }
    if (res->fed) s->lufs[SS_LUFS_HISTORY - 1] = st_launched ? s->stage_d[1] : 0.0;
    res->shortterm = s->lufs[SS_LUFS_HISTORY - 1];
    return SS_OK;
}

// analyze_microphone_input (tui.rs:1427-1480) on the ring resident in s->pcm; `newest`: the ring's newest 2 * SS_TICK_WINDOW
// samples on the host (the crate's value checks run there while the device works)
static int capture_tick_body(ss_session *s, const float *newest, double *mid_xy, double *side_xy, double *wave_xy,
                             size_t *wave_n, size_t window, size_t bins, ss_tick_result *res)
{
    ss_analyzer *h = s->an;
    const size_t n = s->n_samples;
    const size_t pairs = n / 2;                                 // 15 * rate
    const size_t lb = pairs - SS_TICK_WINDOW;
    const int lim = (20000.0f > (float)h->rate / 2.0f) ? SS_ERR_FREQ_LIMIT : SS_OK;
    // One launch for the spectrum of the newest 16384 pairs, the loudness call on the newest 16384 samples and the short-term
    // reading (k_tick, like the file tick), one for the chart of the 15 s mid signal (formed on the fly from the pairs);
    // everything is written into pinned memory by the kernels.  The crate's value checks on the two slices run on the host
    // while the device works.
    const bool fft_wanted = !lim && s->bt->count;
    ssk::FftBatchParams fft_p = fft_wanted ? session_fft_params(s, s->pcm.p, lb, s->stage_dev) : ssk::FftBatchParams{};
    bool fft_launched = false, st_launched = false;
    std::memmove(&s->lufs[0], &s->lufs[1], (SS_LUFS_HISTORY - 1) * sizeof(double));
    ssk::FinalizeParams gating{};
    // The gating of this tick's new sub-blocks is handed back by add_samples_impl AFTER frames_fed has advanced: if the tick is
    // left early (a failed launch, an event error) it is still launched — otherwise those sub-blocks would be missing from the
    // histograms for the rest of the session.
    struct DeferredGating {
        ss_analyzer *h; ssk::FinalizeParams *g; bool launched = false;
        ~DeferredGating() { if (!launched && g->n_streams) (void)ssk::launch_finalize(*g, h->stream); }
    } gating_guard{h, &gating};
    TickExtras extras;
    extras.fft = fft_wanted ? &fft_p : nullptr;
    extras.shortterm_out = s->stage_d_dev;
    res->add_status = add_samples_impl(h, s->pcm.p + (n - SS_TICK_WINDOW), SS_TICK_WINDOW, true, &gating, &extras);
    if (res->add_status == SS_ERR_DEVICE) return SS_ERR_DEVICE;
    if (extras.fused) fft_launched = true;
    if (extras.st_fused) st_launched = true;
    if (fft_wanted && !fft_launched) {
        HIPCHK(ssk::launch_fft16k(fft_p, 1, h->stream));
        fft_launched = true;
    }
    if (!st_launched) {
        if (!h->meter_ok) {
            res->shortterm_status = SS_ERR_INVALID_MODE;
        } else {
            int rc = ring_loudness_enqueue(h, (uint64_t)h->td->host.s100 * 30, s->stage_d_dev);
            if (rc) return rc;
            st_launched = true;
        }
    }
    // microphone_input_chart = get_waveform(&mid_samples, 15.)
    if (wave_xy && bins) {
        ssk::WaveParams p{};
        p.pcm = s->pcm.p; p.stream_stride = 0; p.n_samples = pairs; p.n_streams = 1; p.mid_of_pairs = 1;
        p.window = (uint32_t)window; p.out = s->stage_dev + (size_t)2 * s->bin_stride; p.out_stride = 2 * bins;
        HIPCHK(ssk::launch_waveform(p, h->stream));
    }
    HIPCHK(hipEventRecord(s->ev_tick, h->stream));
    if (gating.n_streams) {
        int rc = attach_readings(h, &gating);
        if (rc) return rc;
        gating_guard.launched = true;
        HIPCHK(ssk::launch_finalize(gating, h->stream));
    }
    if (res->add_status == SS_OK && h->prefetch_stamp != h->change_count) { int rc = prefetch_readings(h); if (rc) return rc; }
    // get_fft's value checks on the two 16384-sample slices (the device is working)
    std::vector<std::pair<size_t, uint8_t>> bad;
    for (size_t i = 0; i < (size_t)SS_TICK_WINDOW; i++) {
        const uint8_t c = pair_class(newest[2 * i], newest[2 * i + 1]);
        if (c) bad.emplace_back(lb + i, c);
    }
    int mid_st = window_value_status(s, lb, SS_TICK_WINDOW, 0, bad);
    int side_st = window_value_status(s, lb, SS_TICK_WINDOW, 2, bad);
    if (!mid_st) mid_st = lim;
    if (!side_st) side_st = lim;
    session_emit_x(s, mid_st, mid_xy);
    session_emit_x(s, side_st, side_xy);
    if (wave_xy && bins)
        for (size_t i = 0; i < bins; i++) { wave_xy[4 * i + 0] = (double)i; wave_xy[4 * i + 2] = (double)i; }
    HIPCHK(hipEventSynchronize(s->ev_tick));
    session_emit_spectrum(s, s->stage, mid_st, mid_xy, &res->mid_status, &res->n_mid, true);
    session_emit_spectrum(s, s->stage + s->bin_stride, side_st, side_xy, &res->side_status, &res->n_side, true);
    if (wave_xy && bins) {
        const float *mm = s->stage + (size_t)2 * s->bin_stride;
        for (size_t i = 0; i < bins; i++) { wave_xy[4 * i + 1] = (double)mm[2 * i]; wave_xy[4 * i + 3] = (double)mm[2 * i + 1]; }
        if (wave_n) *wave_n = 2 * bins;
    }
    s->lufs[SS_LUFS_HISTORY - 1] = st_launched ? s->stage_d[1] : 0.0;
    res->shortterm = s->lufs[SS_LUFS_HISTORY - 1];
    return SS_OK;
}


static int capture_args(ss_session *s, double *mid_xy, double *side_xy, size_t cap_pairs, double *wave_xy, size_t wave_cap_pairs,
                        size_t *wave_n, ss_tick_result *res, size_t *window, size_t *bins)
{
    if (wave_n) *wave_n = 0;
    if (!s || s->is_file || !res || !mid_xy || !side_xy) return SS_ERR_INVALID_ARG;
    if (cap_pairs < s->bt->count || cap_pairs < 1) return SS_ERR_CAPACITY;
    waveform_shape(s->n_samples / 2, 15.0, window, bins);
    if (wave_xy && 2 * *bins > wave_cap_pairs) return SS_ERR_CAPACITY;
    std::memset(res, 0, sizeof *res);
    res->fft_ran = 1; res->lufs_ran = 1; res->fed = 1;
    return SS_OK;
}

// ... on one snapshot of the capture ring (latest_captured_samples.to_vec(), tui.rs:1428): the whole ring is uploaded
int ss_session_tick_capture(ss_session *s, const float *latest, size_t n, double *mid_xy,
                            double *side_xy, size_t cap_pairs, double *wave_xy,
                            size_t wave_cap_pairs, size_t *wave_n, ss_tick_result *res)
{
    SS_ON_DEVICE(s);
    size_t window, bins;
    int rc = capture_args(s, mid_xy, side_xy, cap_pairs, wave_xy, wave_cap_pairs, wave_n, res, &window, &bins);
    if (rc) return rc;
    if (!latest || n != s->n_samples) return SS_ERR_INVALID_ARG;
    // (one copy: the newest 16384 pairs first and the rest behind the tick launch — so that spectrum and loudness run while
    // the host stages the chart's 5.6 MB — measured 219 against 199 us: a second pageable copy costs more than it hides)
    HIPCHK(hipMemcpyAsync(s->pcm.p, latest, n * sizeof(float), hipMemcpyHostToDevice, s->an->stream));
    s->pending_n = 0;                                            // (a snapshot supersedes whatever was pushed before it)
    const float *newest = latest + (n - (size_t)2 * SS_TICK_WINDOW);
    rc = capture_tick_body(s, newest, mid_xy, side_xy, wave_xy, wave_n, window, bins, res);
    std::memcpy(s->tail.data(), newest, s->tail.size() * sizeof(float));
    return rc;
}

// The capture callback's `audio_buf.extend(data)` (audio_capture.rs:41-52) for a ring that lives on the device: the samples are
// kept (page-locked) until the next tick moves the ring.  Host work only; as with every call on a session, one caller at a time.
int ss_session_capture_push(ss_session *s, const float *samples, size_t n)
{
    // host work only (two memcpy): no HIP call, nothing that can block or fail on the capture callback's thread
    if (!s || s->is_file || !s->pending || (!samples && n)) return SS_ERR_INVALID_ARG;
    if (n == 0) return SS_OK;
    const size_t N = s->n_samples;
    if (n >= N) {                                                // more than a whole ring at once: its newest N samples are the ring
        std::memcpy(s->pending, samples + (n - N), N * sizeof(float));
        s->pending_n = N;
    } else {
        if (s->pending_n + n > N) {                              // (no tick for 15 s: the oldest pushed samples have left the ring)
            const size_t drop = s->pending_n + n - N;
            std::memmove(s->pending, s->pending + drop, (s->pending_n - drop) * sizeof(float));
            s->pending_n -= drop;
        }
        std::memcpy(s->pending + s->pending_n, samples, n * sizeof(float));
        s->pending_n += n;
    }
    const size_t T = s->tail.size();
    if (n >= T) std::memcpy(s->tail.data(), samples + (n - T), T * sizeof(float));
    else {
        std::memmove(s->tail.data(), s->tail.data() + n, (T - n) * sizeof(float));
        std::memcpy(s->tail.data() + (T - n), samples, n * sizeof(float));
    }
    return SS_OK;
}

// analyze_microphone_input on the resident ring: what was pushed since the last tick moves the ring on the device (one device copy of
// the part that stays, one DMA of the new samples) — no snapshot crosses PCIe
int ss_session_tick_capture_resident(ss_session *s, double *mid_xy, double *side_xy, size_t cap_pairs, double *wave_xy,
                                     size_t wave_cap_pairs, size_t *wave_n, ss_tick_result *res)
{
    SS_ON_DEVICE(s);
    size_t window, bins;
    int rc = capture_args(s, mid_xy, side_xy, cap_pairs, wave_xy, wave_cap_pairs, wave_n, res, &window, &bins);
    if (rc) return rc;
    const size_t N = s->n_samples, k = s->pending_n;
    hipStream_t st = s->an->stream;
    if (k >= N) {
        HIPCHK(hipMemcpyAsync(s->pcm.p, s->pending, N * sizeof(float), hipMemcpyHostToDevice, st));
    } else if (k) {
        if (!s->pcm_alt.p) HIPCHK(s->pcm_alt.alloc(N));
        HIPCHK(hipMemcpyAsync(s->pcm_alt.p, s->pcm.p + k, (N - k) * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(s->pcm_alt.p + (N - k), s->pending, k * sizeof(float), hipMemcpyHostToDevice, st));
        s->pcm.swap(s->pcm_alt);
    }
    s->pending_n = 0;                                            // (the tick below waits for the stream: `pending` is free again)
    return capture_tick_body(s, s->tail.data(), mid_xy, side_xy, wave_xy, wave_n, window, bins, res);
}

}  // extern "C"
