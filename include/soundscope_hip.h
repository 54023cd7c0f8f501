/*
 * soundscope_hip.h — C ABI of the MI355X-native soundscope analyzer hot path.
 *
 * This is the drop-in boundary for `pub struct Analyzer`
 * (/root/reference/src/analyzer.rs:29-183).  Every Rust method of that type has
 * exactly one entry point here with the same argument meaning and the same
 * error behaviour; a Rust `analyzer.rs` that keeps the reference's signatures
 * and forwards to these symbols is shown in INTEGRATION.md.  The arithmetic
 * underneath runs in hand-written HIP kernels for gfx950 (soundscope_amd/csrc).
 *
 * Conventions
 *   - plain C types only; host pointers unless a name says `_device`.
 *   - every fallible call returns an `ss_status`; SS_OK == 0.  Nothing aborts.
 *   - `(f64,f64)` vectors of the reference are written as interleaved doubles
 *     `out_xy[2*i] = x, out_xy[2*i+1] = y` into caller-allocated storage.
 *   - a handle is used by one thread at a time (the reference only ever calls
 *     from the single TUI thread, src/main.rs:67-77); different handles are
 *     independent (two coexist in the reference, src/tui.rs:459-460).
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     returns SS_ERR_DEVICE.
 */
#ifndef SOUNDSCOPE_HIP_H
#define SOUNDSCOPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ss_add_samples returns behind its launch (a device fault surfaces at the next getter), ss_get_fft_error_values,
 *    ss_batch_get_true_peak_arith; the batch true peak defaults to SS_TP_ARITH_F32 */
#define SS_ABI_VERSION 2

typedef enum ss_status {
    SS_OK = 0,
    /* ebur128::Error (returned by add_samples / getters / create_loudness_meter) */
    SS_ERR_NOMEM = 1,
    SS_ERR_INVALID_MODE = 2,
    SS_ERR_INVALID_CHANNEL = 3,
    /* spectrum_analyzer::SpectrumAnalyzerError (returned by get_fft) */
    SS_ERR_TOO_FEW_SAMPLES = 10,
    SS_ERR_NAN = 11,
    SS_ERR_INFINITY = 12,
    SS_ERR_NOT_POW2 = 13,
    SS_ERR_FREQ_LIMIT = 14,
    SS_ERR_SCALING = 15,
    /* this library */
    SS_ERR_CAPACITY = 20,      /* caller buffer too small */
    SS_ERR_UNSUPPORTED = 21,   /* e.g. FFT length > 32768 (the reference panics there) */
    SS_ERR_INVALID_ARG = 22,
    SS_ERR_DEVICE = 30         /* no HIP device / HIP runtime error */
} ss_status;

const char *ss_status_string(int status);
int ss_abi_version(void);
/* number of visible HIP devices (0 if none); ss_set_device binds the calling
 * process's subsequent handle/batch creations to a device. */
int ss_device_count(void);
/* Tables and scratch are cached PER DEVICE; every handle / batch / session / communicator remembers the device
 * that was current when it was created and makes it current again at each of its entry points, so one process
 * may drive several GPUs from several threads.  (The one-process-per-GPU flow calls ss_set_device once.) */
int ss_set_device(int device);
/* hipDeviceSynchronize on the calling thread's current device (bench fences) */
int ss_device_synchronize(void);
/* last HIP runtime error text seen by this library on this thread ("" if none) */
const char *ss_last_device_error(void);

/* ------------------------------------------------------------------------ *
 *  Inspection of the constant tables the kernels consume, exactly as the library designs them on the host (no device
 *  needed).  They exist so that the PRODUCT's tables can be checked against published numbers (BS.1770-4 coefficient
 *  table, EBU histogram definition, SURVEY's bin counts) independently of the test oracle.  Any pointer may be NULL.
 * ------------------------------------------------------------------------ */
/* K-weighting of ebur128 0.1.10 at `rate` (one 4th-order section: b[0..4], a[0..4], a[0] = 1) */
int ss_inspect_kweight(uint32_t rate, double b5[5], double a5[5]);
/* true-peak interpolator (49-tap Hann-windowed sinc, `factor` 2 or 4): taps[(f - 1) * len + t] = coefficient of x[n - t] in
 * polyphase branch f = 1 .. factor - 1 (branch 0 is the identity tap); *len = taps per branch (12 or 24) */
int ss_inspect_true_peak(int factor, float *taps, uint32_t cap, uint32_t *len);
/* spectrum-analyzer's periodic Hann window in f32 (n values) */
int ss_inspect_hann(uint32_t n, float *w);
/* retained bins of FrequencyLimit::Range(20, 20000) for (rate, n): FFT index of the first one and their number */
int ss_inspect_bins(uint32_t rate, uint32_t n, uint32_t *first_bin, uint32_t *n_bins);
/* ebur128 histogram: 1000 representative energies and 1001 bin boundaries */
int ss_inspect_histogram(double energies1000[1000], double bounds1001[1001]);

/* ------------------------------------------------------------------------ *
 *  Analyzer mirror (one entry point per Rust method)
 * ------------------------------------------------------------------------ */
typedef struct ss_analyzer ss_analyzer;

/* Analyzer::default() is ss_analyzer_create(2, 44100, &h)   analyzer.rs:34-45 */
int ss_analyzer_create(uint32_t channels, uint32_t rate, ss_analyzer **out);
/* Drop                                                                      */
void ss_analyzer_destroy(ss_analyzer *h);
/* create_loudness_meter(&mut self, channels, rate)           analyzer.rs:49-53
 * sample_rate is updated BEFORE the fallible meter creation (rate sticks on error). */
int ss_analyzer_configure(ss_analyzer *h, uint32_t channels, uint32_t rate);
/* get_fft(&self, samples) -> Vec<(chart_x, dB)>              analyzer.rs:55-105
 * n must be a power of two in [2, 32768]; cap_pairs >= n/2+1 is always enough. */
int ss_get_fft(const ss_analyzer *h, const float *samples, size_t n,
               double *out_xy, size_t cap_pairs, size_t *out_n);
/* payload of the LAST failed ss_get_fft on this handle, so that the binding can rebuild the reference's error value and the
 * text the TUI prints from it (analyzer.rs:60-65 `?` -> tui.rs:1439-1442):
 *   SS_ERR_SCALING    -> SpectrumAnalyzerError::ScalingError(a = original magnitude, b = scaled value) of the first bin
 *   SS_ERR_FREQ_LIMIT -> InvalidFrequencyLimit(ValueAboveNyquist(a = 20000.0)); b = the Nyquist frequency it exceeds */
int ss_get_fft_error_values(const ss_analyzer *h, float *a, float *b);
/* get_waveform(samples, waveform_window) — associated fn, no handle
 *                                                            analyzer.rs:107-137
 * out_xy receives (i,min),(i,max) pairs; cap_pairs >= 2*floor(window*1000). */
int ss_get_waveform(const float *samples, size_t n, double waveform_window,
                    double *out_xy, size_t cap_pairs, size_t *out_n);
/* add_samples(&mut self, samples)                            analyzer.rs:139-141
 * `samples` is copied before the call returns (the caller may reuse it at once).  A call of tick size (n <= 32768) returns
 * behind its launches, without waiting for the device: the status covers the arguments and the meter's state (the reference's
 * own error cases); every getter waits for the samples fed before it, and a device fault surfaces there as SS_ERR_DEVICE. */
int ss_add_samples(ss_analyzer *h, const float *samples, size_t n);
/* reset(&mut self)                                           analyzer.rs:143-145 */
void ss_reset(ss_analyzer *h);
/* get_shortterm_lufs / get_integrated_lufs / get_loudness_range
 *                                                            analyzer.rs:147-157
 * -inf is a valid SS_OK result (no blocks above the gate / silence). */
int ss_get_shortterm_lufs(ss_analyzer *h, double *out);
int ss_get_integrated_lufs(ss_analyzer *h, double *out);
int ss_get_loudness_range(ss_analyzer *h, double *out);
/* get_true_peak(&mut self) -> (left, right), linear amplitude analyzer.rs:159-164
 * SS_ERR_INVALID_CHANNEL when the meter has fewer than 2 channels. */
int ss_get_true_peak(ss_analyzer *h, double *left, double *right);
/* sample_rate(&self)                                         analyzer.rs:166-168 */
uint32_t ss_sample_rate(const ss_analyzer *h);
/* calculate_integrated_lufs(&mut self, channels, samples) -> Option<f64>
 *                                                            analyzer.rs:170-182
 * SS_OK => Some(*out); any other status => None.  Uses the handle's
 * sample_rate only; does not touch the handle's meter. */
int ss_calculate_integrated_lufs(ss_analyzer *h, uint32_t channels,
                                 const float *samples, size_t n, double *out);
/* Process-wide caches.  calculate_integrated_lufs (and ss_session_open_file, which calls it) keeps ONE loudness-only batch per
 * process for inputs of up to 64 MB — sized by its first caller with a quarter of headroom, up to ~80 MB of HBM plus the batch's
 * side buffers — and runs under a lock: concurrent calls from several threads are serialised.  The way an input is cut into
 * time segments follows the input's own length, not the kept batch's size, so a reading does not depend on what the process
 * analysed before.  ss_release_caches frees the kept batch (the next call builds a new one); always SS_OK. */
int ss_release_caches(void);
/* get_mid_and_side_samples(samples)                   audio_player.rs:400-419
 * mid/side need n/2 floats each; *out_frames = n/2. */
int ss_mid_side(const float *interleaved, size_t n, float *mid, float *side,
                size_t *out_frames);

/* extensions that the reference's ebur128 meter has under Mode::all() but the
 * app does not surface */
int ss_get_momentary_lufs(ss_analyzer *h, double *out);
int ss_get_true_peak_channel(ss_analyzer *h, uint32_t channel, double *out);
int ss_get_sample_peak_channel(ss_analyzer *h, uint32_t channel, double *out);
/* true-peak oversampling: 0 = ebur128's rule (<96 kHz: 4x, <192 kHz: 2x, else
 * off); 2 or 4 = forced (BASELINE config 5 asks for 4x at 96 kHz).  Takes
 * effect at the next ss_analyzer_configure or ss_reset (both start a fresh meter). */
int ss_analyzer_set_true_peak_factor(ss_analyzer *h, int factor);
/* arithmetic of the handle's 4x true-peak interpolator: SS_TP_ARITH_F32 (default, ebur128's width) or SS_TP_ARITH_F16X3
 * (see ss_batch_set_true_peak_arith below); takes effect with the next ss_add_samples / session tick. */
int ss_analyzer_set_true_peak_arith(ss_analyzer *h, int arith);
/* inspection (tests): the K-weighting filter's carried DF-II state v1..v4 of one channel, as ebur128's Filter keeps it
 * between add_frames calls — sub-normal values flushed to zero at the end of every internal filter call like the crate
 * does.  Waits for the handle's stream. */
int ss_inspect_filter_state(ss_analyzer *h, uint32_t channel, double v4[4]);

/* ------------------------------------------------------------------------ *
 *  PCM ingest: the step before the path.  The reference decodes a
 *  file with symphonia into one interleaved Vec<f32> (audio_player.rs:169-267,
 *  SampleBuffer::<f32>::copy_interleaved_ref); for RIFF/WAVE PCM that is a
 *  sample-format conversion, done here on the device:
 *    u8: s/128 - 1   s16: s/32768   s24: s/8388608   s32: (f32)(s/2147483648.0)
 *    f32: as is      f64: (f32)s          (all divisions are exact powers of two)
 * ------------------------------------------------------------------------ */
typedef enum ss_pcm_format {
    SS_PCM_U8 = 1, SS_PCM_S16 = 2, SS_PCM_S24 = 3, SS_PCM_S32 = 4, SS_PCM_F32 = 5, SS_PCM_F64 = 6
} ss_pcm_format;

typedef struct ss_wav_info {
    uint32_t format;        /* ss_pcm_format */
    uint32_t channels;
    uint32_t sample_rate;
    uint32_t bits_per_sample;
    uint64_t data_offset;   /* byte offset of the first sample in the file image */
    uint64_t data_bytes;    /* bytes of sample data actually present            */
    uint64_t frames;        /* whole frames in the data chunk                   */
} ss_wav_info;

/* Parse a RIFF/WAVE image held in memory (little-endian PCM / IEEE float, incl.
 * WAVE_FORMAT_EXTENSIBLE).  Host logic only; no device needed.
 * SS_ERR_INVALID_ARG: not a WAVE file / truncated header; SS_ERR_UNSUPPORTED: other codecs. */
int ss_wav_parse(const void *file_bytes, size_t len, ss_wav_info *out);
/* bytes per sample of a format (0 if unknown) */
size_t ss_pcm_sample_bytes(int format);
/* interleaved little-endian PCM -> interleaved f32 (host in, host out, converted on the device) */
int ss_pcm_decode(const void *pcm, size_t n_samples, int format, float *out);

/* ------------------------------------------------------------------------ *
 *  Batch extension (NOT in the reference): many independent streams of equal
 *  length analysed in one pass — the data-parallel form of
 *  receive_audio_file + analyze_audio_file_samples (tui.rs:1207-1241,
 *  :1482-1552) over a corpus.  Streams stay resident in HBM.
 * ------------------------------------------------------------------------ */
typedef struct ss_batch ss_batch;

enum {
    SS_BATCH_FFT = 1u,       /* per-window mid/side (2 ch) or per-channel spectrum */
    SS_BATCH_LUFS = 2u,      /* K-weighting, gating blocks, histograms, I, LRA    */
    SS_BATCH_TRUE_PEAK = 4u, /* sample peak + oversampled true peak               */
    SS_BATCH_WAVEFORM = 8u,  /* min-max decimation of the interleaved buffer      */
    SS_BATCH_ALL = 15u,
    /* with SS_BATCH_FFT: columns-only spectrum.  The render-side reduction (below: gain, chart bounds, chart columns) runs
     * INSIDE the spectrum kernel's epilogue, on the window's spectrum while it is still in LDS: the full rows are never
     * stored (nor allocated: 6.5 GB at the bench shape) and only `spectrum_columns` values per row leave the chip — what a
     * terminal can show.  Results equal ss_batch_render_spectrum's bit for bit.  Stereo, fft_n = 4096, hop 1024 only
     * (SS_ERR_UNSUPPORTED otherwise); ss_batch_download_fft is not available (SS_ERR_INVALID_MODE).
     * What the mode buys is BYTES, not kernel time: 5.9 GB less HBM footprint at the bench shape (1024 x 10 s: 0.61 GB of columns
     * instead of 6.49 GB of rows), the second pass over the rows gone, and 10.6x less to download over PCIe.  The kernel itself is
     * no faster than the full-row one (3.0-3.2 ms against 2.9): the spectrum kernel is bound by VALU issue, not by its stores, and
     * folding 1705 bins into columns costs about as many instructions as storing them saves waiting. */
    SS_BATCH_FFT_COLUMNS = 16u
};

typedef struct ss_batch_config {
    uint32_t sample_rate;
    uint32_t channels;          /* 2 => spectrum of mid/side (audio_player.rs:400-419);
                                   otherwise per-channel spectrum */
    uint32_t n_streams;
    uint32_t fft_n;             /* window length (power of two, <= 32768) */
    uint32_t hop_frames;        /* 1024 in the reference (audio_player.rs:65) */
    uint32_t flags;             /* SS_BATCH_* */
    int32_t true_peak_factor;   /* 0 = ebur128 rule, 2 / 4 = forced */
    uint32_t spectrum_columns;  /* SS_BATCH_FFT_COLUMNS: chart columns per spectrum row, 1 .. 512 (otherwise 0) */
    uint64_t frames_per_stream;
    double waveform_window;     /* seconds; <= 0 => frames_per_stream / sample_rate */
} ss_batch_config;

typedef struct ss_stream_result {
    double integrated_lufs;     /* loudness_global(), may be -inf            */
    double loudness_range;      /* loudness_range()                          */
    double true_peak[2];        /* channels 0,1: max(true, sample), linear   */
    double sample_peak[2];
    uint32_t n_gating_blocks;   /* 400 ms blocks evaluated                   */
    uint32_t n_st_blocks;       /* 3 s blocks evaluated                      */
} ss_stream_result;

typedef struct ss_batch_layout {
    uint32_t n_windows;         /* per stream */
    uint32_t fft_channels;      /* 2 (mid, side) or channels */
    uint32_t n_bins;            /* retained bins, 20 Hz..20 kHz */
    uint32_t first_bin;         /* FFT bin index of retained bin 0 */
    uint32_t n_wave_points;     /* per stream: 2 per decimation bin */
    uint32_t n_subblocks;       /* complete 100 ms sub-blocks per stream */
    uint32_t fft_bin_stride;    /* floats between spectrum rows on the device (n_bins rounded up to 4) */
    uint32_t reserved;
    uint64_t input_bytes;       /* device bytes of the input corpus */
    uint64_t fft_bytes;         /* device bytes of the spectrum output */
} ss_batch_layout;

int ss_batch_create(const ss_batch_config *cfg, ss_batch **out);
void ss_batch_destroy(ss_batch *b);
int ss_batch_layout_get(const ss_batch *b, ss_batch_layout *out);
/* copy host PCM into streams [first, first+count): interleaved f32,
 * count*frames_per_stream*channels floats */
int ss_batch_upload(ss_batch *b, uint32_t first, uint32_t count, const float *pcm);
int ss_batch_download_input(ss_batch *b, uint32_t stream, float *pcm, size_t cap_floats);
/* same as ss_batch_upload for raw PCM of `format`: the bytes are copied to the GPU and converted
 * there straight into the resident f32 corpus (no host-side f32 copy) */
int ss_batch_upload_pcm(ss_batch *b, uint32_t first, uint32_t count, const void *pcm, int format);
/* Ragged batches: streams of different lengths in one batch.  Create the batch for the longest stream
 * (frames_per_stream is the slot size), then give every stream its own length; each gets its own window count,
 * sub-block count and decimation geometry by the rules ss_batch_create applies to a uniform batch.  Rows beyond a
 * stream's own counts in the downloads are unspecified; ss_batch_stream_shape says how many are valid. */
typedef struct ss_stream_shape {
    uint64_t frames;
    uint32_t n_windows, n_subblocks, n_wave_points, reserved;
} ss_stream_shape;              /* 24 bytes */
int ss_batch_set_lengths(ss_batch *b, const uint64_t *frames, uint32_t n_streams);
int ss_batch_stream_shape(const ss_batch *b, uint32_t stream, ss_stream_shape *out);
/* the first n_samples interleaved samples of one stream's slot (raw PCM of any ss_pcm_format), queued on the batch's
 * stream; `pcm` must stay valid until the next ss_batch_sync */
int ss_batch_upload_samples(ss_batch *b, uint32_t stream, const void *pcm, size_t n_samples, int format);
/* Pipelined ingest.  A batch owns its HIP stream, so two batches are a double buffer: while one runs, the other's
 * upload is in flight — provided the host memory is page-locked (ss_host_register pins caller memory in place) and
 * the upload does not wait: ss_batch_upload_pcm_async only queues the copy and the conversion; `pcm` must stay
 * valid until the next ss_batch_sync / ss_batch_results on that batch.  soundscope_amd/pipeline.py is the loop. */
int ss_host_register(void *ptr, size_t bytes);
int ss_host_unregister(void *ptr);
int ss_batch_upload_pcm_async(ss_batch *b, uint32_t first, uint32_t count, const void *pcm, int format);
/* device pointer of the resident corpus ([stream][frame][channel] f32) for
 * producers that already hold data on the GPU */
void *ss_batch_input_device_ptr(ss_batch *b);
/* fill the corpus with the documented synthetic signal: utility
 * for benchmarks, not part of the measured path */
int ss_batch_synthesize(ss_batch *b, uint64_t seed, uint32_t first_stream_id);
/* measurement utility (like ss_batch_synthesize, not part of the measured path): run ONLY the HBM traffic of this batch's
 * spectrum kernel — same grid, workgroup size, LDS footprint (occupancy) and addresses, loads and stores, no arithmetic —
 * `reps` times and return the average milliseconds per launch: the floor the memory system sets for that access pattern
 * on this device, next to which the kernel's time can be read.  Overwrites the batch's spectrum buffer.
 * SS_ERR_UNSUPPORTED unless the batch takes the N = 4096 stereo kernel at hop 1024. */
int ss_batch_traffic_floor(ss_batch *b, uint32_t reps, double *ms_per_launch);
/* enqueue one pass of the hot path over the whole batch; asynchronous */
int ss_batch_run(ss_batch *b);
int ss_batch_sync(ss_batch *b);
/* results (after ss_batch_sync) */
int ss_batch_results(ss_batch *b, ss_stream_result *out, uint32_t cap);
/* every channel's peaks of one stream (ss_stream_result only carries channels 0 and 1, like
 * Analyzer::get_true_peak, analyzer.rs:159-164; ebur128 keeps them per channel and the app leaves the rest as
 * "TODO: channels", tui.rs:1217-1221): true_pk[c] = max(true, sample) like EbuR128::true_peak(c), sample_pk[c]
 * = EbuR128::sample_peak(c), linear.  Either array may be NULL; cap_channels >= the batch's channel count. */
int ss_batch_peaks(ss_batch *b, uint32_t stream, double *true_pk, double *sample_pk, uint32_t cap_channels);
/* the launch geometry this batch's shape selected (tests assert that the benchmark geometry is the one they check) */
typedef struct ss_batch_geometry {
    uint32_t fft_windows_per_block;  /* windows one spectrum workgroup walks                          */
    uint32_t fft_blocks;             /* spectrum workgroups per pass                                  */
    uint32_t td_segments;            /* time segments per stream in the time-domain kernel            */
    uint32_t td_segment_subblocks;   /* 100 ms sub-blocks per segment (0: one segment)                */
    uint32_t td_warm_subblocks;      /* filter run-in of segments > 0 (SS_TD_RUN_IN: 1; td_split == 2: 2) */
    uint32_t td_true_peak_factor;    /* 0, 2, 4                                                       */
    uint32_t waveform_fused;         /* 1: decimation runs inside the time-domain kernel              */
    uint32_t overlap;                /* ss_batch_set_overlap mode: 0, 1 or 2                          */
    uint32_t td_split;               /* 1: a stream is one segment walked by the four waves of a workgroup, the filter state handed
                                      *    from tile to tile (the whole recurrence: no run-in); td_segments is 1 then.  2: a handful of
                                      *    streams: every SEGMENT's tiles dealt to the eight waves of a workgroup (latency); a segment
                                      *    then runs the filter over the td_warm_subblocks (2) in front of it inside the one launch —
                                      *    the state the second launch would start from — and td_fixup_subblocks is 0              */
    uint32_t td_fixup_subblocks;     /* sub-blocks at the head of every segment > 0 re-run from the exact state by the second launch */
} ss_batch_geometry;                 /* 40 bytes (32 up to ABI version 1) */
int ss_batch_geometry_get(const ss_batch *b, ss_batch_geometry *out);
/* the same for a caller built against an older (shorter) or newer (longer) struct: writes min(out_bytes, sizeof(ss_batch_geometry))
 * bytes, never more than the caller has — a client compiled against the 32-byte struct of ABI version 1 passes 32 and gets the
 * fields it knows (ss_batch_geometry_get itself writes the whole 40-byte struct and is for callers that check ss_abi_version()) */
int ss_batch_geometry_get_sized(const ss_batch *b, void *out, size_t out_bytes);
/* how the time-domain kernel walks a stream.  The K-weighting recurrence has a long memory (poles at |z| = 0.995): a stream cut
 * into time segments for parallelism must hand the filter state from one segment to the next.
 *   SS_TD_AUTO (default)   time segments, one wave each, EXACT hand-over: every segment starts at its boundary from a zero state,
 *                          leaves its end state behind, and a second light launch re-runs the first 0.2 s of every segment > 0 from
 *                          the state the segment in front of it left, overwriting those sub-blocks' energies (after 0.2 s the
 *                          zero start differs from the true trajectory by e^-48 of the state: 1e-13 of the filtered signal on
 *                          DC-offset material, under the 2e-11 at which any two evaluation orders of this recurrence differ).
 *                          Segment 0 and the re-run head of segment 1 equal the one-segment path bit for bit; elsewhere the
 *                          difference is the rounding noise of the recurrence (3e-10 at worst on the bench corpus, the same as
 *                          for the exact-by-construction SS_TD_WHOLE_STREAMS), not a dropped term (tests pin both).
 *                          A handful of streams (one file; td_split == 2 in the geometry): the same segments on eight waves each,
 *                          and every segment runs the filter over the 0.2 s in front of it, from zero, inside the ONE launch —
 *                          the state the second launch would have started from, without a second launch.
 *   SS_TD_RUN_IN           the form of rounds 1-4, kept for comparison: a segment > 0 starts its filter 0.1 s early from a zero
 *                          state and drops that run-in (3e-10 on sub-block energies of DC-offset material: a truncation).
 *   SS_TD_WHOLE_STREAMS    stereo / eight channels, equal lengths: a stream is ONE segment walked by the four waves of a
 *                          workgroup, the state handed from tile to tile through LDS — exact by construction, but the waves of a
 *                          workgroup wait for each other: 16 % slower at the bench shape than independent segment waves. */
enum { SS_TD_AUTO = 0, SS_TD_RUN_IN = 1, SS_TD_WHOLE_STREAMS = 2 };
int ss_batch_set_time_domain_mode(ss_batch *b, int mode);
/* how a pass is laid over the batch's two HIP streams; results are identical in every mode.
 *   0  sequential: spectrum kernel, then the time-domain chain (default);
 *   1  the spectrum kernel on a second stream beside the whole time-domain chain (they use different pipes: packed f32
 *      VALU + LDS vs f64 VALU + matrix cores);
 *   2  the time-domain kernel alone first, then the spectrum kernel on the second stream beside the chain's short
 *      latency-bound tail (per-stream gating / histograms, a standalone decimation).
 * Per-kernel event timing (ss_batch_timing_enable) always runs sequentially. */
int ss_batch_set_overlap(ss_batch *b, int mode);
/* arithmetic of the 4x true-peak interpolator in batches (the Analyzer handle and the sessions: ss_analyzer_set_true_peak_arith, same default):
 *   SS_TP_ARITH_F32 (default)    an f32 fmaf chain per output, the width of ebur128's interpolator (its 12 products per phase
 *                                summed in f32; analyzer.rs:139-141,159-164).  2, 6 or 8 channels: on the packed-f32 VALU
 *                                (v_pk_fma_f32; a frame's pair of adjacent channels is one packed operand, round 6; at factor 2
 *                                the 24-tap branch's halves on neighbouring lanes — in the four-waves register builds with plain
 *                                v_fma_f32 instead); channel counts that do not divide 16 (3, 5, 7 ...): plain v_fma_f32, a lane
 *                                per channel; mono, 4 and 16 channels: v_mfma_f32_16x16x4_f32 as a banded-Toeplitz product
 *                                (measured the faster form there);
 *   SS_TP_ARITH_F16X3 (opt-in)   three-term f16 split on the matrix cores with f32 accumulation, scaled per tile by a
 *                                power of two from the tile's own peak: within 2^-21 of the tile peak of the f32 result
 *                                (measured 2.0e-7 relative on the bench corpus against 1.1e-7 for the f32 product; north_star's
 *                                bar is 1e-4), 3-5 % less time-domain kernel time since the f32 form left the matrix pipe
 *                                (it was 25 %).  2 / 8 channels only: other shapes, and tiles with non-finite samples, take
 *                                the f32 form whatever the mode. */
enum { SS_TP_ARITH_F16X3 = 0, SS_TP_ARITH_F32 = 1 };
int ss_batch_set_true_peak_arith(ss_batch *b, int arith);
int ss_batch_get_true_peak_arith(const ss_batch *b);      /* SS_TP_ARITH_* */
/* verification utility: order-independent 64-bit checksums, computed on the device, of everything a pass left in HBM for
 * each stream: out[3 * s + 0] the stream's whole spectrum block ([n_windows][fft_channels][fft_bin_stride] f32 bit patterns),
 * out[3 * s + 1] its decimation bins, out[3 * s + 2] its sub-block energies (f64 bit patterns).  Two passes agree on a
 * checksum iff (up to 2^-64) they agree bit for bit on the data: a stress test can compare EVERY window of a multi-gigabyte
 * batch between launch modes (ss_batch_set_overlap) without downloading it.  Waits for the batch's stream. */
int ss_batch_checksums(ss_batch *b, uint64_t *out, uint32_t cap_streams);
/* spectrum of one stream: compact [n_windows][fft_channels][n_bins] f32 dB (pink-compensated);
 * on the device the rows are fft_bin_stride floats apart */
int ss_batch_download_fft(ss_batch *b, uint32_t stream, float *out, size_t cap_floats);
/* chart_x / frequency / pink compensation per retained bin (f64, n_bins each; any may be NULL) */
int ss_batch_bin_tables(const ss_batch *b, double *chart_x, double *freq, double *pink_db);
/* waveform of one stream: n_wave_points pairs (i, value) like get_waveform, as f32 values only:
 * out[2*i] = min, out[2*i+1] = max of decimation bin i */
int ss_batch_download_waveform(ss_batch *b, uint32_t stream, float *out, size_t cap_floats);
/* K-weighted energy of the 100 ms sub-blocks of one stream: [n_subblocks][channels] f64 */
int ss_batch_download_subblocks(ss_batch *b, uint32_t stream, double *out, size_t cap_doubles);
/* corpus histograms summed over this batch's streams: 1000 block-energy bins
 * followed by 1000 short-term bins (u64 each).  `_device` copies them into a
 * caller-owned device buffer (e.g. the send buffer of an RCCL all-reduce). */
int ss_batch_histograms(ss_batch *b, uint64_t *out2000);
int ss_batch_histograms_device(ss_batch *b, void *dst_device_2000_u64);
/* A7/A8 on a (summed / all-reduced) histogram: ebur128 loudness_global_multiple
 * and loudness_range_multiple semantics.  Host-side, O(1000). */
double ss_corpus_integrated_lufs(const uint64_t *block_hist1000);
double ss_corpus_loudness_range(const uint64_t *st_hist1000);

/* ------------------------------------------------------------------------- *
 *  Multi-GPU: one process per GPU, streams sharded over the ranks, and exactly ONE exchange —
 *  the SUM all-reduce of the two 1000-bin u64 histograms for the corpus-level integrated-LUFS gate
 *  (ebur128 loudness_global_multiple semantics; the reference app has no collective).  The library talks to
 *  RCCL itself (librccl is opened at ss_comm_init; ncclAllReduce(buf, buf, 2000, ncclUint64, ncclSum) on the
 *  batch's stream, over xGMI) — no PyTorch, no MPI.
 *    SS_COMM_RCCL      device buffers, RCCL
 *    SS_COMM_HOST_TCP  the same calls staged through host memory over loopback TCP: CPU tests of the rank
 *                      logic, and ranks that share one GPU
 *  Rendezvous (one node): rank 0 publishes the RCCL unique id (and its TCP port) in `rendezvous_file`, the other
 *  ranks poll for it.  ss_comm_init_from_env derives rank / world from RANK / WORLD_SIZE and the file name from
 *  the launcher's process id and MASTER_PORT (torchrun as a launcher only), or takes SS_COMM_FILE.
 *  Environment: RCCL across processes needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose driver has only dmabuf IPC, and
 *  the HSA runtime reads it at the process' first HIP call.  Loading the library changes nothing in the environment; the
 *  ss_comm_init* calls (RCCL, world > 1) set the variable if it is unset, which is in time only when they make the process'
 *  FIRST HIP call.  A rank therefore does NOT call ss_set_device first: ss_comm_init_on_device(…, device, …) and
 *  ss_comm_init_from_env (device = SS_COMM_DEVICE — explicit: it must parse as a number and name a visible device —, else
 *  LOCAL_RANK; where the launcher masks ONE GPU per rank, ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES per task as under
 *  SLURM, LOCAL_RANK still counts 0 .. n-1 against a single visible device: a LOCAL_RANK beyond the visible devices is
 *  taken modulo their count, i.e. device 0 there) make the rank's GPU current themselves, behind the
 *  setenv; plain ss_comm_init keeps whatever device is current (for callers that export the variable in the launcher, as
 *  bench.py does).  When the variable is wrong the failure surfaces as "hipIpcGetMemHandle: invalid argument" inside
 *  ncclCommInitRank; the error text of a failed init names the variable.
 *  Failure is all-or-nothing and prompt: what goes wrong on one rank before the ranks meet (no such device, no librccl, no
 *  memory) and the outcome of ncclCommInitRank are exchanged over the join sockets, so every rank returns SS_ERR_DEVICE within
 *  seconds and ss_last_device_error names the reason (the failing rank's own, "another rank ..." elsewhere).  Only a rank that
 *  never arrives (its process died) is waited for: SS_COMM_TIMEOUT_S seconds (default 180) at the rendezvous and again
 *  inside ncclCommInitRank.
 * ------------------------------------------------------------------------- */
typedef struct ss_comm ss_comm;
enum { SS_COMM_RCCL = 0, SS_COMM_HOST_TCP = 1 };
int ss_comm_init(int transport, int rank, int world, const char *rendezvous_file, ss_comm **out);
int ss_comm_init_on_device(int transport, int rank, int world, int device, const char *rendezvous_file, ss_comm **out);
int ss_comm_init_from_env(int transport, ss_comm **out);
void ss_comm_destroy(ss_comm *c);
int ss_comm_rank(const ss_comm *c);
/* number of ranks as the transport itself reports it (ncclCommCount for RCCL) */
int ss_comm_size(const ss_comm *c);
const char *ss_comm_transport_name(const ss_comm *c);
/* ncclGetVersion's code of the librccl this communicator runs on (e.g. 22707 = 2.27.7); 0 for the host transport */
int ss_comm_library_version(const ss_comm *c);
/* small host-buffer collectives (fences and the max-over-ranks clock of the benchmark) */
int ss_comm_barrier(ss_comm *c);
int ss_comm_allreduce_u64_sum(ss_comm *c, uint64_t *inout, size_t n);
int ss_comm_allreduce_f64_max(ss_comm *c, double *inout, size_t n);
/* the corpus gate's exchange: in-place SUM all-reduce of this batch's corpus histograms (block ++ short-term,
 * 2000 u64) across the ranks, queued on the batch's stream behind ss_batch_run; out2000 (nullable) receives
 * the reduced histograms after the stream has drained.  Afterwards ss_batch_histograms returns the corpus-wide
 * histograms on every rank, and ss_corpus_integrated_lufs / ss_corpus_loudness_range evaluate the gate. */
int ss_batch_allreduce_histograms(ss_batch *b, ss_comm *c, uint64_t *out2000);
/* the whole corpus gate on the device, asynchronously: [ss_batch_allreduce_histograms over `c` when c != NULL] + A7 / A8
 * (loudness_global_multiple, loudness_range_multiple) of the summed histograms into a device pair, queued on the batch's
 * stream behind ss_batch_run — no copy, no wait, so a loop of passes needs no host synchronisation per pass.
 * ss_batch_corpus_gate_read waits for the stream and returns the pair of the last pass. */
int ss_batch_corpus_gate_enqueue(ss_batch *b, ss_comm *c);
int ss_batch_corpus_gate_read(ss_batch *b, double *integrated_lufs, double *loudness_range);

/* ------------------------------------------------------------------------- *
 *  Render-side reductions: the step after the path.
 *  The reference adds fft_gain_compensation_db to every bin (tui.rs:801-821) and draws inside the chart
 *  bounds [FFT_LOWER_BOUND, FFT_UPPER_BOUND] = [-100, 0] dB (tui.rs:49-51, :890); the waveform chart shows
 *  the view [x_min, x_max] of the decimation bins (tui.rs:664-681).  Here both are reduced on the device to
 *  `cols` chart columns so that only what a terminal can show leaves HBM.  Gain and bounds are the
 *  reference's; the column rule is this library's (the reference hands every point to ratatui):
 *    spectrum column c = bins with floor(chart_x / 100 * cols) == c (last column closed), value =
 *                        max(clamp(dB + gain, -100, 0)); a column without a bin is NaN
 *    waveform column c = decimation bins i with floor((i - x_min) * cols / (x_max - x_min)) == c,
 *                        value = (min of mins, max of maxes); a column without a bin is (NaN, NaN)
 * ------------------------------------------------------------------------- */
enum { SS_GAIN_FIXED = 0,       /* gain_db as given                                                   */
       SS_GAIN_REFERENCE = 1 }; /* per stream FFT_TARGET_LUFS(-13) - integrated as f32 (tui.rs:1234) */
/* after ss_batch_run: [stream][window][fft_channel][cols] f32 into a device buffer of the batch */
int ss_batch_render_spectrum(ss_batch *b, uint32_t cols, int gain_mode, float gain_db);
/* SS_BATCH_FFT_COLUMNS batches: the gain of the fused reduction, for the passes that follow.  Default: SS_GAIN_REFERENCE
 * when the batch also runs the meter (SS_BATCH_LUFS: the pass then runs the time-domain chain first — the gain needs
 * every stream's integrated loudness), otherwise SS_GAIN_FIXED with 0 dB.  ss_batch_download_spectrum_columns reads
 * the result. */
int ss_batch_set_columns_gain(ss_batch *b, int gain_mode, float gain_db);
int ss_batch_download_spectrum_columns(ss_batch *b, uint32_t stream, float *out, size_t cap_floats);
/* [stream][cols][2] f32 (min, max) of the decimation bins x_min <= i < x_max */
int ss_batch_render_waveform(ss_batch *b, uint32_t cols, uint32_t x_min, uint32_t x_max);
int ss_batch_download_waveform_columns(ss_batch *b, uint32_t stream, float *out, size_t cap_floats);
/* the Player-mode view bounds of the waveform chart (tui.rs:664-681); host arithmetic, f64 */
void ss_waveform_view(double playhead_ms, double waveform_window_s, size_t chart_points,
                      double *x_min, double *x_max);

/* kernel timing with HIP events on the batch's own stream (for roofline
 * reporting): enable, run N passes, then read accumulated per-kernel time.
 * Timed passes queue back to back: the event pairs go into a ring of 32 passes'
 * sets and are read (one stream synchronisation) by ss_batch_timing_read /
 * ss_batch_sync / ss_batch_timing_enable — or by ss_batch_run itself when 32 passes
 * are pending — so a timed region of K passes measures its own kernels. */
enum { SS_KERNEL_FFT = 0, SS_KERNEL_TIME_DOMAIN = 1, SS_KERNEL_FINALIZE = 2, SS_KERNEL_WAVEFORM = 3, SS_KERNEL_COUNT = 4 };
int ss_batch_timing_enable(ss_batch *b, int enable);
int ss_batch_timing_read(ss_batch *b, int kernel, double *total_ms, uint64_t *launches);
const char *ss_kernel_name(int kernel);            /* the bench configuration's kernels */
const char *ss_batch_kernel_name(const ss_batch *b, int kernel);   /* the kernel this batch's shape selects */

/* ------------------------------------------------------------------------- *
 *  Tick drivers: the per-file / per-device state of the
 *  reference's App and its per-tick analysis, with the audio resident in HBM.
 *    ss_session_open_file     receive_audio_file            tui.rs:1207-1241
 *                             (+ AudioFile::from_file's mid/side and duration,
 *                              audio_player.rs:150-166)
 *    ss_session_tick_file     analyze_audio_file_samples    tui.rs:1482-1552
 *    ss_session_open_capture  device selection              tui.rs:1780-1808
 *    ss_session_tick_capture  analyze_microphone_input      tui.rs:1427-1480
 *    ss_session_restart       play / seek handlers          tui.rs:1586-1614
 *    ss_session_capture_push  the capture callback's extend audio_capture.rs:41-52
 *  A tick is ONE call: nothing is uploaded for a file tick (the file was
 *  uploaded at open; its spectrum, its loudness call and its short-term
 *  reading are one launch); the capture tick either uploads the 30*rate-sample snapshot it is given, or — the ring kept on the
 *  device, ss_session_capture_push + ss_session_tick_capture_resident — only what was captured since the last tick.  The session
 *  owns its analyzer (file_analyzer / device_analyzer) and the 300-entry short-term history (tui.rs:420,463).
 * ------------------------------------------------------------------------- */
typedef struct ss_session ss_session;

#define SS_LUFS_HISTORY 300
#define SS_TICK_WINDOW 16384      /* tui.rs:1488 / :1431: FFT window; :1529 / :1466: LUFS slice (samples) */

typedef struct ss_tick_result {
    uint64_t playhead;         /* pos / channels (tui.rs:1484-1485); capture: 0                     */
    int32_t fft_ran;           /* 0: the reference skips the FFT block (left bound == 0)            */
    int32_t mid_status;        /* status of get_fft(mid); != 0 => the reference stores [(0., 0.)]   */
    int32_t side_status;
    uint32_t n_mid, n_side;    /* pairs written to mid_xy / side_xy (1 = the (0,0) fallback)        */
    int32_t lufs_ran;          /* 0: LUFS block skipped (left bound == 0): history not shifted      */
    int32_t fed;               /* 1: add_samples + get_shortterm_lufs ran (bounds check passed)     */
    int32_t add_status;        /* status of add_samples                                             */
    int32_t shortterm_status;  /* status of get_shortterm_lufs; != 0 => lufs[299] = 0.0             */
    uint32_t reserved;
    double shortterm;          /* lufs[299] after this tick                                         */
} ss_tick_result;              /* 56 bytes */

/* interleaved: the decoded file (AudioFile::samples), n_samples floats; channels: the file's channel
 * count (AudioFile::channels — used only for pos / channels, mid/side always pair samples 2i, 2i+1);
 * the meter is created with 2 channels like the reference (tui.rs:1217-1221). */
int ss_session_open_file(const float *interleaved, size_t n_samples, uint32_t channels,
                         uint32_t sample_rate, ss_session **out);
/* device_analyzer.create_loudness_meter(channels, rate) + a 30*rate capture ring */
int ss_session_open_capture(uint32_t channels, uint32_t sample_rate, ss_session **out);
void ss_session_close(ss_session *s);
/* the session's Analyzer (for get_integrated_lufs / get_true_peak / get_loudness_range reads) */
ss_analyzer *ss_session_analyzer(ss_session *s);
/* waveform.audio_file_chart = get_waveform(samples, duration_s) (tui.rs:1213-1216) */
int ss_session_waveform(ss_session *s, double *out_xy, size_t cap_pairs, size_t *out_n);
/* fft_gain_compensation_db = FFT_TARGET_LUFS(-13) - integrated as f32, or 0 (tui.rs:1229-1238) */
int ss_session_gain_db(ss_session *s, float *out);
/* AudioFile::duration in ms (audio_player.rs:154) */
int ss_session_duration_ms(ss_session *s, uint64_t *out);
/* pos: the playback position in interleaved samples as passed to analyze_audio_file_samples */
int ss_session_tick_file(ss_session *s, size_t pos, double *mid_xy, double *side_xy,
                         size_t cap_pairs, ss_tick_result *res);
/* latest: the capture ring oldest-first (latest_captured_samples.to_vec()), n must be 30 * rate.
 * wave_xy receives waveform.microphone_input_chart = get_waveform(mid, 15.) */
int ss_session_tick_capture(ss_session *s, const float *latest, size_t n, double *mid_xy,
                            double *side_xy, size_t cap_pairs, double *wave_xy,
                            size_t wave_cap_pairs, size_t *wave_n, ss_tick_result *res);
/* The capture ring resident on the device (it starts full of zeros like the reference's, tui.rs:1783-1784):
 *   ss_session_capture_push           the capture callback's `audio_buf.extend(data)` (audio_capture.rs:41-52; mono devices push
 *                                     the zero-interleaved data the callback builds): host work only
 *   ss_session_tick_capture_resident  analyze_microphone_input on the ring as it stands after every push so far: the same results
 *                                     as ss_session_tick_capture on `latest_captured_samples.to_vec()`, without the snapshot
 * The two forms can be mixed (a snapshot tick replaces the ring).  One caller at a time per session: where the reference
 * locks its ring's mutex (audio_capture.rs:41, tui.rs:1428), lock around these calls. */
int ss_session_capture_push(ss_session *s, const float *samples, size_t n);
int ss_session_tick_capture_resident(ss_session *s, double *mid_xy, double *side_xy, size_t cap_pairs, double *wave_xy,
                                     size_t wave_cap_pairs, size_t *wave_n, ss_tick_result *res);
/* lufs = [-100.; 300]; analyzer.reset() */
int ss_session_restart(ss_session *s);
int ss_session_lufs_history(ss_session *s, double *out300);

#ifdef __cplusplus
}
#endif
#endif /* SOUNDSCOPE_HIP_H */
