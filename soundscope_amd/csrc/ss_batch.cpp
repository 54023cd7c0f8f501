// ss_batch.cpp — the batch extension of the C ABI (NOT in the reference): many streams resident in HBM analysed in one
// pass — the data-parallel form of receive_audio_file + analyze_audio_file_samples
// (/root/reference/src/tui.rs:1207-1241, :1482-1552) over a corpus — plus the render-side reductions (N3) and the
// one-shot calculate_integrated_lufs (/root/reference/src/analyzer.rs:170-182), which runs the batch path on one stream.
#include "ss_host.h"

using namespace ssh;

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
    uint32_t td_nsub_hint = 0;      // ragged batches: sub-blocks of the LONGEST stream (the geometry follows the lengths, not the slot size)
    bool td_split = false;          // whole-stream workgroups (choose_td_geometry)
    bool td_split_segments = false; // the same per time segment, eight waves each (a handful of streams)
    int td_mode = 0;                // ss_batch_set_time_domain_mode
    bool wave_fused = false;     // decimation runs inside the time-domain kernel
    uint32_t wave_halo = 0;
    FftTables *ft = nullptr;
    BinTables *bt = nullptr;
    TdTables *td = nullptr;
    DevBuf<float> pcm, fft, wave;
    DevBuf<ssk::TdState> state;
    DevBuf<double> sub, weights, integrated, lra, out2;
    DevBuf<double> seg_state;       // [stream][segment][channel][4]: the filter state behind every time segment (exact hand-over)
    DevBuf<uint64_t> hist, corpus;
    DevBuf<uint32_t> counts;
    DevBuf<unsigned char> raw;      // device staging of raw PCM for the asynchronous ingest
    DevBuf<uint64_t> checks;        // ss_batch_checksums: [stream][3]
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
    // columns-only spectrum (SS_BATCH_FFT_COLUMNS): the reduction fused into the spectrum kernel's epilogue
    bool columns_only = false;
    int columns_gain_mode = SS_GAIN_FIXED;
    float columns_gain_db = 0.0f;
    DevBuf<uint16_t> bin_col;       // chart column of every retained bin (0xFFFF for the row padding)
    DevBuf<uint2> col_groups;       // the same per group of four bins (FftBatchParams::col_groups)
    DevBuf<float> col_init;         // FftBatchParams::col_init
    DevBuf<uint2> col_bins;         // FftBatchParams::col_bins
    // opt-in (SS_BATCH_OVERLAP=1): the spectrum kernel on a second stream beside the time-domain chain
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool corpus_reduced = false;      // this pass's corpus histograms already hold the all-reduced sums
    int tp_arith = SS_TP_ARITH_F32;   // SS_TP_ARITH_*: the reference's width unless the caller opts into the f16 split
    int overlap = 0;                  // 0 sequential, 1 the spectrum kernel beside the time-domain chain, 2 beside its tail only
    bool timing = false;
    // per-kernel event timing: a ring of kTimingDepth passes' event sets, so that timed passes queue back to back without a host
    // synchronisation between them (bench.py times its kernels INSIDE the timed region); collected when read, or when the ring is full
    static constexpr int kTimingDepth = 32;
    hipEvent_t ev[kTimingDepth * 2 * SS_KERNEL_COUNT] = {};
    uint32_t ev_mask[kTimingDepth] = {};       // which of a pass's events were recorded
    uint32_t ev_head = 0, ev_count = 0;        // next slot to record into; passes recorded and not yet collected
    bool ev_ready = false;
    double t_ms[SS_KERNEL_COUNT] = {0, 0, 0, 0};
    uint64_t t_n[SS_KERNEL_COUNT] = {0, 0, 0, 0};
};

namespace ssi {
void *batch_corpus_device(ss_batch *b) { return b ? b->corpus.p : nullptr; }
hipStream_t batch_stream(ss_batch *b) { return b ? b->stream : nullptr; }
int batch_device(const ss_batch *b) { return b ? b->device : 0; }
bool &batch_corpus_reduced(ss_batch *b) { return b->corpus_reduced; }
}  // namespace ssi

namespace {

// chart column of a bin: floor(chart_x / 100 * cols), the last column closed on the right (include/soundscope_hip.h)
uint32_t spectrum_column_of(double chart_x, uint32_t cols)
{
    double f = std::floor(chart_x / 100.0 * (double)cols);
    if (f < 0) f = 0;
    return f >= (double)cols ? cols - 1 : (uint32_t)f;
}

int batch_collect_timing(ss_batch *b)
{
    if (!b->ev_count) return SS_OK;
    HIPCHK(hipStreamSynchronize(b->stream));
    constexpr uint32_t D = ss_batch::kTimingDepth;
    for (uint32_t i = 0; i < b->ev_count; i++) {
        const uint32_t slot = (b->ev_head + D - b->ev_count + i) % D;
        hipEvent_t *ev = b->ev + (size_t)slot * 2 * SS_KERNEL_COUNT;
        for (int k = 0; k < SS_KERNEL_COUNT; k++) {
            if ((b->ev_mask[slot] >> (2 * k) & 3u) != 3u) continue;       // this pass did not run kernel k
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) == hipSuccess) { b->t_ms[k] += ms; b->t_n[k]++; }
        }
        b->ev_mask[slot] = 0;
    }
    b->ev_count = 0;
    return SS_OK;
}

}  // namespace

extern "C" {

// How the time-domain kernel walks a stream.
//  * whole-stream workgroups (td_split): a stream is ONE segment, its tiles dealt to the four waves of a workgroup, the filter state
//    and the lanes' energy shares handed from tile to tile through LDS — the whole recurrence, nothing truncated.  Needs enough
//    streams to fill the chip with one workgroup each (n_streams x 4 waves against the W0 the chip holds); stereo and eight
//    channels, equal lengths.
//  * segments (the rest): a stream is cut into nseg runs of whole sub-blocks, one wave each; a segment > 0 starts its filter
//    kTdWarmSub sub-blocks (0.1 s) early from a zero state and drops that run-in (what the missing history would add to the
//    output has decayed to 1e-9 of a DC step by then, ss_time_domain.hip).  A segment costs its run-in; pick the segment
//    length that maximises   useful fraction  seg / (seg + warm)  x  fill of the last round  waves / (ceil(waves / W0) W0)
//    (ranks the measured config-5 sweep seg = 2..13 in the right order; measured within noise for config 3).
// mode (ss_batch_set_time_domain_mode): 0 the better score of the two, 1 segments, 2 whole-stream workgroups where the shape allows.
#ifndef SS_TD_SPLIT_LONG_RUN_IN
#define SS_TD_SPLIT_LONG_RUN_IN 1      // 0: split segments hand over through the second launch like every other segmented batch (A/B builds)
#endif
static void choose_td_geometry(ss_batch *b)
{
    if (!b->td) return;
    const ss_batch_config *cfg = &b->cfg;
    const ss_batch_layout &L = b->lay;
    const uint32_t C = cfg->channels;
    const uint32_t nsub = b->ragged ? b->td_nsub_hint : L.n_subblocks;
    const double W0 = 256.0 * ssk::td_resident_waves_per_cu(C, b->td->host.s100, b->wave_fused ? b->wave_halo : 0);
    // what a segment boundary costs, in sub-blocks of full work: the run-in (mode 1), or the fix-up's re-run of kTdFixSub sub-blocks
    // at about 0.8 of a full tile each (filter and energies are most of a tile) — config 5's sweep seg = 2 ... 10 with the fix-up:
    // 2.39 / 2.21 / 2.13 / 2.53 / 2.65 / 3.60 ms, best at 4
    const double boundary_cost = b->td_mode == 1 ? (double)kTdWarmSub : 0.8 * (double)kTdFixSub;
    auto score_of = [&](uint32_t seg, uint32_t nseg) {
        const double waves = (double)cfg->n_streams * nseg;
        const double useful = nseg > 1 ? (double)seg / ((double)seg + boundary_cost) : 1.0;
        return useful * waves / (std::ceil(waves / W0) * W0);
    };
    // shortest segment: with the exact hand-over the state a segment leaves is only as good as the segment is long (it started from
    // zero): kTdFixSub sub-blocks at least, so that what it hands on has converged like the fix-up's own re-run
    const uint32_t min_seg = b->td_mode == 1 ? kTdWarmSub : kTdFixSub;
    // Time segments stand on the filter FORGETTING: a segment starts from zero, and what it hands on (fix-up) or what it ran in
    // from (mode 1) is right once the zero-input response of the true state has died — radius^n over min_seg sub-blocks,
    // exp(-48) = 1.6e-21 at every ordinary rate (the 38 Hz high-pass pair: n and 1 - radius scale with the rate alike).  The
    // crate accepts rates from 16 Hz (a sub-block is 2 frames there; between ~100 Hz and ~3.4 kHz the design is not even
    // stable): where radius^n has not fallen below 1e-18 a stream is ONE segment (tools/fuzz_handle.py: 0.1 - 1 LU off at 16 Hz).
    bool forgets = false;
    {
        const double r = sst::kweight_pole_radius((double)cfg->sample_rate);
        const double n = (double)min_seg * (double)b->td->host.s100;
        forgets = r < 1.0 && n * std::log(r) < std::log(b->td_mode == 1 ? 1e-9 : 1e-18);       // (mode 1 is the approximate one: 4e-11 after its 0.1 s)
    }
    uint32_t best_seg = 0;
    double best = nsub ? score_of(nsub, 1) : 0.0;                         // one segment: no run-in
    for (uint32_t want = 2; forgets && want <= nsub; want++) {          // balanced segments: seg = ceil(nsub / want)
        const uint32_t seg = (nsub + want - 1) / want;
        if (seg < min_seg) break;
        const double sc = score_of(seg, (nsub + seg - 1) / seg);
        if (sc > best * 1.0000001) { best = sc; best_seg = seg; }
    }
#ifdef SS_TUNING
    if (const char *e = std::getenv("SS_TD_SEG_SUB")) best_seg = (uint32_t)std::atoi(e);
#endif
    const bool split_ok = (C == 2 || C == 8) && nsub > 0;
    const double wsplit = 4.0 * cfg->n_streams;
    const double split_score = split_ok ? wsplit / (std::ceil(wsplit / W0) * W0) : 0.0;
    // (measured at the bench shape: the coupled waves of a workgroup run 16 % behind independent segment waves — the chain makes a
    // workgroup as slow as its slowest wave, tile by tile — so the automatic choice wants a clear win in fill)
    b->td_split = split_ok && (b->td_mode == 2 || (b->td_mode == 0 && 0.8 * split_score > best));
    b->td_split_segments = false;
    if (b->td_split) {
        b->td_nseg = 1; b->td_seg_sub = 0;
    } else if (best_seg >= min_seg && best_seg < nsub) {
        b->td_seg_sub = best_seg;
        b->td_nseg = (nsub + best_seg - 1) / best_seg;
    } else {
        b->td_nseg = 1; b->td_seg_sub = 0;
    }
    // A handful of streams (one file): even the shortest segments leave most of the chip idle, and a pass takes as long as ONE
    // wave needs for its segment's tiles, one after the other.  There a segment's tiles are dealt to the eight waves of a
    // workgroup instead (the whole-stream form, per segment): if eight waves per shortest segment still fit the chip at once.
    const uint32_t nseg_min = (nsub + min_seg - 1) / min_seg;
    if (forgets && split_ok && b->td_mode == 0 && !b->td_split && nsub > min_seg && 8.0 * cfg->n_streams * nseg_min <= W0) {
        b->td_split_segments = true;
        b->td_seg_sub = min_seg;
        b->td_nseg = nseg_min;
    }
}

int ss_batch_create(const ss_batch_config *cfg, ss_batch **out)
{
    if (!cfg || !out) return SS_ERR_INVALID_ARG;
    *out = nullptr;
    if (require_device()) return SS_ERR_DEVICE;
    if (cfg->n_streams == 0 || cfg->frames_per_stream == 0) return SS_ERR_INVALID_ARG;
    if ((cfg->flags & SS_BATCH_ALL) == 0) return SS_ERR_INVALID_ARG;
    const bool columns_only = (cfg->flags & SS_BATCH_FFT_COLUMNS) != 0;
    if (columns_only && (!(cfg->flags & SS_BATCH_FFT) || cfg->spectrum_columns == 0 || cfg->spectrum_columns > 512)) return SS_ERR_INVALID_ARG;
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
    {   // sizes that cannot be a buffer: refused before any product of them is formed (a wrapped product would allocate a small
        // buffer and index far beyond it).  2^40 samples = 4 TB of f32, fourteen times the HBM of the card.
        unsigned long long samples = 0;
        if (F > (1ull << 40) || __builtin_mul_overflow((unsigned long long)cfg->n_streams, (unsigned long long)F, &samples) ||
            __builtin_mul_overflow(samples, (unsigned long long)C, &samples) || samples > (1ull << 40))
            return SS_ERR_NOMEM;
    }
    HIPCHK(stream_acquire(&b->stream));
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
            // ... unless runs of sixteen leave most of the chip idle (one file, a handful of streams): then the pass is bound by
            // the length of a run, not by its constants — as many workgroups as the chip holds at once (three per CU), runs of two
            // windows at least (config 2, one 10 s stream: 29 workgroups x 16 windows 50 us -> 232 x 2)
            const uint64_t total = (uint64_t)cfg->n_streams * L.n_windows;
            if ((uint64_t)cfg->n_streams * ((L.n_windows + wpb - 1) / (wpb ? wpb : 1)) < 512u) {
                const uint64_t w = (total + 767u) / 768u;
                wpb = (uint32_t)(w < 2 ? 2 : w);
            }
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
        if (columns_only) {
            // the fused reduction lives in the epilogue of k_fft4096_ms1
            if (!(b->fft_fast && hop == 1024)) return SS_ERR_UNSUPPORTED;
            const uint32_t cols = cfg->spectrum_columns;
            std::vector<uint16_t> bc(L.fft_bin_stride, (uint16_t)0xFFFF);
            for (uint32_t i = 0; i < L.n_bins; i++) bc[i] = (uint16_t)spectrum_column_of(b->bt->chart_x[i], cols);
            HIPCHK(b->bin_col.upload(bc));
            std::vector<uint2> cg(L.fft_bin_stride / 4), cbins(L.fft_bin_stride / 4);
            std::vector<float> cinit(cols, std::numeric_limits<float>::quiet_NaN());
            for (uint32_t i = 0; i < L.n_bins; i++) cinit[bc[i]] = -std::numeric_limits<float>::infinity();
            for (uint32_t g = 0; g < cg.size(); g++) {
                uint32_t o[4];
                bool general = false;
                for (uint32_t e = 0; e < 4; e++) {
                    const bool pad = bc[4 * g + e] == 0xFFFF;
                    o[e] = pad ? 2048u : 4u * bc[4 * g + e];
                    general = general || pad;
                }
                uint32_t n = 1;
                while (n < 4 && o[n] == o[0]) n++;
                for (uint32_t e = n; e < 4; e++) general = general || o[e] != o[3];
                cg[g] = make_uint2(o[0] | (o[3] << 16), general ? 0u : n);
                cbins[g] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
            }
            HIPCHK(b->col_bins.upload(cbins));
            HIPCHK(b->col_init.upload(cinit));
            HIPCHK(b->col_groups.upload(cg));
            const uint64_t rows = (uint64_t)cfg->n_streams * L.n_windows * L.fft_channels;
            HIPCHK(b->render_spec.alloc(rows * cols));
            b->render_cols = cols;
            b->columns_only = true;
            b->columns_gain_mode = (cfg->flags & SS_BATCH_LUFS) ? SS_GAIN_REFERENCE : SS_GAIN_FIXED;
            L.fft_bytes = rows * cols * sizeof(float);
        } else {
            L.fft_bytes = (uint64_t)cfg->n_streams * L.n_windows * L.fft_channels * L.fft_bin_stride * sizeof(float);
            HIPCHK(b->fft.alloc((size_t)(L.fft_bytes / sizeof(float))));
        }
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
            // (+ 3: the general decimation path reads the aligned 16-byte piece a bin starts in, up to three samples in front of it)
            const uint32_t need = ((uint32_t)std::ceil(spp) + 5 + C - 1) / C;
            uint32_t halo = need < 24 ? 24 : need;
            halo = (halo + 3u) & ~3u;
            if (halo <= 512) { b->wave_fused = true; b->wave_halo = halo; }
        }
    }
    choose_td_geometry(b.get());
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
    if (b->stream2) { (void)hipStreamSynchronize(b->stream2); stream_release(b->stream2); }
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    if (b->stream) stream_release(b->stream);
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
    // the time-domain geometry follows the lengths actually set (the longest stream), not the slot size the batch was created with:
    // a batch that is kept and re-used — the one-shot loudness call keeps one — then cuts the same input the same way whatever
    // was analysed before it (the low bits of a reading do not depend on the process' history)
    b->td_nsub_hint = 0;
    for (uint32_t i = 0; i < count; i++) if (b->sub_h[i] > b->td_nsub_hint) b->td_nsub_hint = b->sub_h[i];
    choose_td_geometry(b);
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
    int rc = (b->ev_count >= (uint32_t)ss_batch::kTimingDepth || !b->timing) ? batch_collect_timing(b) : SS_OK;     // (a full ring: collect first)
    if (rc) return rc;
    b->corpus_reduced = false;
    const ss_batch_config &c = b->cfg;
    const ss_batch_layout &L = b->lay;
    const uint32_t C = c.channels;
    const bool tm = b->timing;
    const uint32_t ev_slot = b->ev_head;
    auto rec = [&](int idx) -> hipError_t {
        if (!tm) return hipSuccess;
        b->ev_mask[ev_slot] |= 1u << idx;
        return hipEventRecord(b->ev[(size_t)ev_slot * 2 * SS_KERNEL_COUNT + idx], b->stream);
    };

    // overlap mode: fork — the spectrum kernel goes to stream2 after everything already queued on the main stream
    // (uploads, the previous pass), the time-domain chain stays on the main stream, join at the end.  Per-kernel
    // event timing is meaningless while two kernels share the chip, so timing passes stay sequential.
    // Mode 2 (tail overlap): the time-domain kernel runs first and alone; the spectrum kernel starts behind it on stream2
    // while the short latency-bound tail of the chain (gating / histograms per stream, a standalone decimation) runs on
    // the main stream beside it.
    // columns-only spectrum with the reference's per-file gain: the gain is -13 - integrated of each stream, so the whole
    // time-domain chain (kernel + gating / histograms) runs first and the spectrum kernel last, on the one stream
    const bool spectrum_last = b->columns_only && b->columns_gain_mode == SS_GAIN_REFERENCE;
    const int mode = (tm || spectrum_last) ? 0 : b->overlap;
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
        if (b->columns_only) {
            p.out_cols = b->render_spec.p; p.bin_col = b->bin_col.p; p.col_groups = b->col_groups.p; p.col_init = b->col_init.p; p.col_bins = b->col_bins.p; p.cols = b->render_cols;
            p.integrated = b->columns_gain_mode == SS_GAIN_REFERENCE ? b->integrated.p : nullptr;
            p.gain_db = b->columns_gain_db;
        }
        if (b->fft_fast || b->fft_pairw) {
            p.db_offset = (float)(10.0 * std::log10(4.0 / ((double)c.fft_n * (double)c.fft_n)));
            p.offpink = b->bt->offpink4096_dev.p;
            p.publish_mask = ssk::fft4096_publish_mask(L.first_bin, L.n_bins);
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
    if (mode != 2 && !spectrum_last) { rc = launch_spectrum(); if (rc) return rc; }

    const bool td = (c.flags & (SS_BATCH_LUFS | SS_BATCH_TRUE_PEAK)) != 0;
    HIPCHK(rec(2 * SS_KERNEL_TIME_DOMAIN));
    if (td) {
        {   // meter state, histograms, corpus histograms and block counts start from zero: one launch (they were four fills)
            void *const ptrs[4] = {b->state.p, b->hist.p, b->corpus.p, b->counts.p};
            const size_t bytes[4] = {b->state.n * sizeof(ssk::TdState), b->hist.n * sizeof(uint64_t), b->corpus.n * sizeof(uint64_t),
                                     b->counts.n * sizeof(uint32_t)};
            HIPCHK(ssk::launch_zero4(ptrs, bytes, b->stream));
        }
        HIPCHK(rec(2 * SS_KERNEL_TIME_DOMAIN));   // time the kernel, not the memsets
        ssk::TdParams p{};
        p.pcm = b->pcm.p; p.stream_stride = c.frames_per_stream * C; p.n_frames = c.frames_per_stream;
        p.n_streams = c.n_streams; p.channels = C; p.k = b->td->dev.p; p.state = b->state.p;
        p.subblocks = b->sub.p; p.sub_cap = L.n_subblocks ? L.n_subblocks : 1;
        p.sub_stride = (uint64_t)p.sub_cap * C; p.ring = nullptr; p.ring_frames = 0; p.tp_factor = b->tp_factor;
        p.s100 = b->td->host.s100; p.nseg = b->td_nseg; p.seg_sub = b->td_seg_sub;
        // segments > 0: exact hand-over (no run-in; their first kTdFixSub sub-blocks re-run from the true state by the second launch
        // below), or — mode 1 — the 0.1 s run-in from a zero state of rounds 1-4
        // ... or, a handful of streams cut into the shortest segments (td_split_segments: the pass is a latency chain, and a second
        // launch is a fifth of it): every segment runs the FILTER over the kTdFixSub sub-blocks in front of it, from zero, inside the
        // one launch — the state it starts its own frames with is what the second launch would have started from (a zero-state run
        // over 0.2 s: 1.6e-21 of the true state's response left), the run-in tiles cost the filter passes only, no second launch
        const bool long_run_in = b->td_nseg > 1 && b->td_mode == 0 && b->td_split_segments && SS_TD_SPLIT_LONG_RUN_IN;
        const bool exact_segments = b->td_nseg > 1 && b->td_mode != 1 && !long_run_in;
        p.warm_sub = long_run_in ? kTdFixSub : (exact_segments ? 0u : kTdWarmSub);
        if (exact_segments) {
            const size_t need = (size_t)c.n_streams * b->td_nseg * C * 4;
            if (b->seg_state.n < need) HIPCHK(b->seg_state.alloc(need));
            p.seg_state = b->seg_state.p;
            p.fix_sub = b->td_seg_sub < kTdFixSub ? b->td_seg_sub : kTdFixSub;
        }
        p.frames_of = b->ragged ? b->frames_d.p : nullptr;
        p.tp_f32 = b->tp_arith == SS_TP_ARITH_F32 ? 1u : 0u;
        p.split_batch = b->ragged ? 0u : (b->td_split ? 1u : (b->td_split_segments ? 2u : 0u));      // (ragged lengths: one wave per stream / segment)
        if (b->wave_fused && !b->ragged) { p.wave_out = b->wave.p; p.wave_stride = (uint64_t)2 * b->wave_window; p.wave_window = b->wave_window; p.halo_frames = b->wave_halo; }
        HIPCHK(ssk::launch_time_domain(p, b->stream));
        if (mode == 2) {
            // the tail starts HERE: the hand-over's second launch (filter and energies of every segment's first 0.2 s: one wave per
            // segment, a chain of ten short tiles each — latency, not throughput, and no matrix-core work) runs beside the spectrum
            // kernel like the gating behind it
            HIPCHK(hipEventRecord(b->ev_fork, b->stream));
            HIPCHK(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
            rc = launch_spectrum(); if (rc) return rc;
        }
        if (exact_segments) HIPCHK(ssk::launch_time_domain_fixup(p, b->stream));
    } else if (mode == 2) {                                  // (no time-domain work at all: the spectrum kernel is the pass)
        HIPCHK(hipEventRecord(b->ev_fork, b->stream));
        HIPCHK(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
        rc = launch_spectrum(); if (rc) return rc;
    }
    HIPCHK(rec(2 * SS_KERNEL_TIME_DOMAIN + 1));

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
        f.state = b->state.p;
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
    if (spectrum_last) { rc = launch_spectrum(); if (rc) return rc; }
    if (ov) {                                  // join: later work on the main stream (downloads, the next pass) sees the spectrum
        HIPCHK(hipEventRecord(b->ev_join, b->stream2));
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_join, 0));
    }
    if (tm) { b->ev_head = (b->ev_head + 1u) % (uint32_t)ss_batch::kTimingDepth; b->ev_count++; }
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
    if (!b->fft.p) return SS_ERR_INVALID_MODE;          // columns-only batches have no spectrum rows to store into
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
        out->td_warm_subblocks = (b->td_nseg > 1 && b->td_mode == 1) ? kTdWarmSub : 0;
        // (ragged lengths: one wave per stream / segment whatever the shape would have allowed — ss_batch_run)
        out->td_split = b->ragged ? 0u : (b->td_split ? 1u : (b->td_split_segments ? 2u : 0u));
        const bool long_run_in = b->td_nseg > 1 && b->td_mode == 0 && b->td_split_segments && SS_TD_SPLIT_LONG_RUN_IN;
        out->td_fixup_subblocks = (b->td_nseg > 1 && b->td_mode != 1 && !long_run_in) ? (b->td_seg_sub < kTdFixSub ? b->td_seg_sub : kTdFixSub) : 0;
        if (long_run_in) out->td_warm_subblocks = kTdFixSub;
        out->td_true_peak_factor = (uint32_t)b->tp_factor;
    }
    out->waveform_fused = (b->wave_fused && !b->ragged) ? 1u : 0u;
    out->overlap = (uint32_t)b->overlap;
    return SS_OK;
}

int ss_batch_geometry_get_sized(const ss_batch *b, void *out, size_t out_bytes)
{
    if (!b || !out) return SS_ERR_INVALID_ARG;
    ss_batch_geometry g;
    const int rc = ss_batch_geometry_get(b, &g);
    if (rc) return rc;
    std::memcpy(out, &g, out_bytes < sizeof g ? out_bytes : sizeof g);
    return SS_OK;
}

int ss_batch_set_overlap(ss_batch *b, int enable)
{
    SS_ON_DEVICE(b);
    if (!b) return SS_ERR_INVALID_ARG;
    if (enable && !b->stream2) {
        HIPCHK(stream_acquire(&b->stream2));
        HIPCHK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    b->overlap = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return SS_OK;
}

int ss_batch_checksums(ss_batch *b, uint64_t *out, uint32_t cap_streams)
{
    SS_ON_DEVICE(b);
    if (!b || !out) return SS_ERR_INVALID_ARG;
    const uint32_t ns = b->cfg.n_streams;
    if (cap_streams < ns) return SS_ERR_CAPACITY;
    HIPCHK(b->checks.ensure((size_t)3 * ns));
    HIPCHK(hipMemsetAsync(b->checks.p, 0, (size_t)3 * ns * sizeof(uint64_t), b->stream));
    const ss_batch_layout &L = b->lay;
    if (b->fft.p && L.n_windows) {
        const uint64_t words = (uint64_t)L.n_windows * L.fft_channels * L.fft_bin_stride;
        HIPCHK(ssk::launch_checksum(b->fft.p, words, words, ns, b->checks.p, 3, b->stream));
    }
    if (b->wave.p && b->wave_window) {
        const uint64_t words = (uint64_t)2 * b->wave_window;
        HIPCHK(ssk::launch_checksum(b->wave.p, words, words, ns, b->checks.p + 1, 3, b->stream));
    }
    if (b->sub.p && L.n_subblocks) {
        const uint64_t words = (uint64_t)2 * L.n_subblocks * b->cfg.channels;
        HIPCHK(ssk::launch_checksum(b->sub.p, words, words, ns, b->checks.p + 2, 3, b->stream));
    }
    HIPCHK(hipMemcpyAsync(out, b->checks.p, (size_t)3 * ns * sizeof(uint64_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SS_OK;
}

int ss_batch_set_true_peak_arith(ss_batch *b, int arith)
{
    if (!b || (arith != SS_TP_ARITH_F16X3 && arith != SS_TP_ARITH_F32)) return SS_ERR_INVALID_ARG;
    b->tp_arith = arith;
    return SS_OK;
}

int ss_batch_get_true_peak_arith(const ss_batch *b) { return b ? b->tp_arith : SS_ERR_INVALID_ARG; }

int ss_batch_set_time_domain_mode(ss_batch *b, int mode)
{
    SS_ON_DEVICE(b);
    if (!b || mode < 0 || mode > 2) return SS_ERR_INVALID_ARG;      // SS_TD_AUTO / SS_TD_RUN_IN / SS_TD_WHOLE_STREAMS
    HIPCHK(hipStreamSynchronize(b->stream));
    b->td_mode = mode;
    choose_td_geometry(b);
    return SS_OK;
}

int ss_batch_download_fft(ss_batch *b, uint32_t stream, float *out, size_t cap)
{
    SS_ON_DEVICE(b);
    if (!b || !out || stream >= b->cfg.n_streams) return SS_ERR_INVALID_ARG;
    if (b->columns_only) return SS_ERR_INVALID_MODE;          // the rows were never stored: ss_batch_download_spectrum_columns
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
    if (!(b->cfg.flags & SS_BATCH_FFT) || !L.n_windows || !L.n_bins || b->columns_only) return SS_ERR_INVALID_MODE;   // (columns-only: no rows to reduce)
    if (gain_mode == SS_GAIN_REFERENCE && !(b->cfg.flags & SS_BATCH_LUFS)) return SS_ERR_INVALID_MODE;
    // column of a bin: floor(chart_x / 100 * cols), the last column closed on the right; chart_x ascends
    std::vector<uint32_t> start(cols + 1, L.n_bins);
    {
        uint32_t c = 0;
        start[0] = 0;
        for (uint32_t i = 0; i < L.n_bins; i++) {
            const uint32_t ci = spectrum_column_of(b->bt->chart_x[i], cols);
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

int ss_batch_set_columns_gain(ss_batch *b, int gain_mode, float gain_db)
{
    if (!b || !b->columns_only || (gain_mode != SS_GAIN_FIXED && gain_mode != SS_GAIN_REFERENCE)) return SS_ERR_INVALID_ARG;
    if (gain_mode == SS_GAIN_REFERENCE && !(b->cfg.flags & SS_BATCH_LUFS)) return SS_ERR_INVALID_MODE;
    b->columns_gain_mode = gain_mode;
    b->columns_gain_db = gain_db;
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
}  // extern "C"
namespace oneshot {
// the one loudness-only batch the process keeps for calculate_integrated_lufs / receive_audio_file (see integrated_oneshot)
struct Slot { ss_batch *b = nullptr; uint32_t rate = 0, channels = 0; uint64_t cap = 0; int device = -1; };
static std::mutex mu;
static Slot slot;
}  // namespace oneshot
extern "C" int ss_release_caches(void)
{
    std::lock_guard<std::mutex> lk(oneshot::mu);
    if (oneshot::slot.b) { ss_batch_destroy(oneshot::slot.b); oneshot::slot = oneshot::Slot{}; }
    return SS_OK;
}
int ssh::integrated_oneshot(uint32_t rate, uint32_t channels, const float *samples, size_t n,
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
    // A one-stream, loudness-only batch pass.  Building and tearing down a batch is fourteen device allocations and as many
    // hipFree calls (20-50 us each: most of what opening a file cost), so ONE batch is kept per process for inputs of up to
    // 64 MB — created with a quarter of headroom and re-used through the ragged-length path (ss_batch_set_lengths) for every
    // later call of the same (device, rate, channel count) that fits; the pass runs under a lock.  Longer inputs take a
    // batch of their own as before.
    const uint64_t frames = n / channels;
    constexpr size_t kCacheMaxFloats = (size_t)16 << 20;
    using oneshot::Slot; using oneshot::slot;
    std::unique_lock<std::mutex> lk(oneshot::mu, std::defer_lock);
    ss_batch *b = nullptr;
    bool cached = false;
    if (n <= kCacheMaxFloats) {
        lk.lock();
        const int dev = current_device();
        if (!(slot.b && slot.device == dev && slot.rate == rate && slot.channels == channels && slot.cap >= frames)) {
            if (slot.b) { ss_batch_destroy(slot.b); slot = Slot{}; }
            ss_batch_config cfg{};
            cfg.sample_rate = rate; cfg.channels = channels; cfg.n_streams = 1; cfg.flags = SS_BATCH_LUFS;
            cfg.frames_per_stream = frames + frames / 4 + rate; cfg.fft_n = 0; cfg.hop_frames = 0;
            rc = ss_batch_create(&cfg, &slot.b);
            if (rc) { slot = Slot{}; return rc; }
            slot.rate = rate; slot.channels = channels; slot.cap = cfg.frames_per_stream; slot.device = dev;
        }
        b = slot.b;
        cached = true;
        rc = ss_batch_set_lengths(b, &frames, 1);
        if (rc) return rc;
    } else {
        ss_batch_config cfg{};
        cfg.sample_rate = rate; cfg.channels = channels; cfg.n_streams = 1; cfg.flags = SS_BATCH_LUFS;
        cfg.frames_per_stream = frames; cfg.fft_n = 0; cfg.hop_frames = 0;
        rc = ss_batch_create(&cfg, &b);
        if (rc) return rc;
    }
    if (!hip_ok(hipMemcpyAsync(b->pcm.p, samples, n * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, b->stream),
                "hipMemcpyAsync(one-shot input)")) rc = SS_ERR_DEVICE;
    if (!rc) rc = ss_batch_run(b);
    if (!rc) rc = ss_batch_sync(b);
    ss_stream_result r{};
    if (!rc) rc = ss_batch_results(b, &r, 1);
    if (!cached) ss_batch_destroy(b);
    if (rc) return rc;
    *out = r.integrated_lufs;
    return SS_OK;
}
extern "C" {

int ss_calculate_integrated_lufs(ss_analyzer *h, uint32_t channels, const float *samples, size_t n, double *out)
{
    SS_ON_DEVICE(h);
    if (!h || !out) return SS_ERR_INVALID_ARG;
    return integrated_oneshot(h->rate, channels, samples, n, false, out);
}

}  // extern "C"