/*
 * ss_oracle.h — CPU restatement of soundscope's analyzer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or
 * executed by the product path (soundscope_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED at the crate boundary: the arithmetic of the reference path
 * lives in three un-vendored crates (ebur128 0.1.10, spectrum-analyzer 1.7.0,
 * microfft 0.6.0 — /root/reference/Cargo.lock:566-569, :1941-1944, :1074-1077)
 * whose sources are not under /root/reference, and there is no Rust toolchain
 * in the build image.  The functions below restate the published algorithms
 * (ITU-R BS.1770-4, EBU Tech 3341/3342, libebur128's design, the crates'
 * documented behaviour) and follow /root/reference/src/analyzer.rs and
 * src/audio_player.rs:400-419 line by line where those are readable.  They are
 * pinned by the EBU/ITU known-answer cases, the BS.1770 coefficient table and
 * the reference's own unit tests (tests/test_oracle_*.py), not by crate output.
 */
#ifndef SS_ORACLE_H
#define SS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (shared numbering with include/soundscope_hip.h) */
enum {
    SO_OK = 0,
    SO_ERR_NOMEM = 1,            /* ebur128::Error::NoMem              */
    SO_ERR_INVALID_MODE = 2,     /* ebur128::Error::InvalidMode        */
    SO_ERR_INVALID_CHANNEL = 3,  /* ebur128::Error::InvalidChannelIndex*/
    SO_ERR_TOO_FEW_SAMPLES = 10, /* SpectrumAnalyzerError::TooFewSamples */
    SO_ERR_NAN = 11,
    SO_ERR_INFINITY = 12,
    SO_ERR_NOT_POW2 = 13,
    SO_ERR_FREQ_LIMIT = 14,
    SO_ERR_SCALING = 15,
    SO_ERR_CAPACITY = 20,
    SO_ERR_UNSUPPORTED = 21      /* where the crates panic: a transform longer than 32768 points */
};

/* ---- spectrum (analyzer.rs:11-27, :55-105; spectrum-analyzer 1.7.0) ---- */
void so_hann_window(const float *x, size_t n, float *out);
/* real FFT (microfft-style radix-2 DIT on n/2 complex + recombination), f32.
 * out_re/out_im get n/2+1 bins (DC..Nyquist). */
void so_rfft(const float *x, size_t n, float *out_re, float *out_im);
/* number of bins with 20 <= k*(sr/n) <= 20000 and the first such k */
size_t so_fft_bins(uint32_t sample_rate, size_t n, size_t *first_k);
/* Analyzer::get_fft: out_xy = pairs (chart_x, dB+pink) as f64. */
int so_get_fft(uint32_t sample_rate, const float *x, size_t n,
               double *out_xy, size_t cap_pairs, size_t *out_n);
/* same, but also returns the f32 dBFS values before pink compensation
 * (what spectrum-analyzer hands back to analyzer.rs:75); either may be NULL. */
int so_get_fft_ex(uint32_t sample_rate, const float *x, size_t n,
                  double *out_xy, float *out_dbfs, size_t cap, size_t *out_n,
                  int with_stats);

/* ---- waveform (analyzer.rs:107-137) ---- */
size_t so_get_waveform(const float *x, size_t n, double window_s,
                       double *out_xy, size_t cap_pairs);

/* ---- PCM ingest (audio_player.rs:169-267 via symphonia) ---- */
void so_pcm_to_f32(const unsigned char *src, size_t n, int fmt, float *dst);

/* ---- mid/side (audio_player.rs:400-419) ---- */
size_t so_mid_side(const float *interleaved, size_t n, float *mid, float *side);

/* ---- loudness meter (ebur128 0.1.10, Mode::all()) ---- */
typedef struct so_meter so_meter;
int so_meter_new(uint32_t channels, uint32_t rate, so_meter **out);
/* force_factor: 0 = reference rule (<96k:4, <192k:2, else none); 2 or 4 = forced */
int so_meter_new_ex(uint32_t channels, uint32_t rate, int force_tp_factor, so_meter **out);
void so_meter_free(so_meter *m);
void so_meter_reset(so_meter *m);
int so_meter_add_frames_f32(so_meter *m, const float *src, size_t n_samples);
int so_meter_loudness_momentary(so_meter *m, double *out);
int so_meter_loudness_shortterm(so_meter *m, double *out);
int so_meter_loudness_global(so_meter *m, double *out);
int so_meter_loudness_range(so_meter *m, double *out);
int so_meter_sample_peak(so_meter *m, uint32_t ch, double *out);
int so_meter_true_peak(so_meter *m, uint32_t ch, double *out);
const uint64_t *so_meter_block_hist(so_meter *m);   /* 1000 bins */
const uint64_t *so_meter_st_hist(so_meter *m);      /* 1000 bins */
/* sub-normal handling of the K-weighting filter: SO_FTZ_END_OF_CALL (default: the state flushed at the end of every internal
 * filter call, the crate's build without SSE2 — and what the device path restates) or SO_FTZ_PER_OP (MXCSR flush-to-zero for the
 * duration of the call, the crate's x86 build; see filter_process) */
enum { SO_FTZ_END_OF_CALL = 0, SO_FTZ_PER_OP = 1 };
int so_meter_set_ftz(so_meter *m, int mode);
void so_meter_filter_coeffs(so_meter *m, double b[5], double a[5]);
/* carried DF-II state v1..v4 of one channel after the last add_frames (sub-normals flushed like ebur128's Filter::process) */
int so_meter_filter_state(so_meter *m, uint32_t ch, double v4[4]);
/* loudness_global_multiple / loudness_range_multiple on summed histograms */
double so_gated_loudness_hist(const uint64_t *hist);
double so_loudness_range_hist(const uint64_t *st_hist);
/* polyphase layout probe: taps per phase, delay */
int so_interp_layout(int taps, int factor, int *counts /*factor*/, int *delay);
size_t so_interp_coeffs(int taps, int factor, int phase, float *coeff, int *index, size_t cap);

/* ---- Analyzer::calculate_integrated_lufs (analyzer.rs:170-182) ---- */
/* returns SO_OK and *out (may be -inf), or an error (=> None) */
int so_calculate_integrated_lufs(uint32_t sample_rate, uint32_t channels,
                                 const float *x, size_t n, double *out);

/* ---- whole-stream "one pass of the hot path" used as the CPU baseline ----
 * Runs, on one interleaved stereo stream: get_waveform (W = duration*1000),
 * mid/side split, get_fft on mid and side for every window [p-N,p), p=k*hop,
 * N<p<=F (tui.rs:1482-1526 cadence), and one meter pass (add_frames in
 * 2*sr-sample chunks, analyzer.rs:175) with integrated/LRA/true-peak read out.
 * Outputs may be NULL; fft_out receives nwin*2*nbins f32 dB(+pink) values. */
typedef struct {
    double integrated, lra, true_peak[2], sample_peak[2];
    size_t n_windows, n_bins, n_wave_points;
} so_stream_result;
int so_analyze_stream(uint32_t sample_rate, const float *interleaved, size_t n_samples,
                      size_t fft_n, size_t hop_frames, int force_tp_factor,
                      float *fft_out, double *wave_out, so_stream_result *res);

/* bench.py's all-cores CPU leg: n_streams equal-length stereo streams (stream s = buffer s % n_distinct of `interleaved`)
 * dealt round-robin to n_threads POSIX threads, each running the whole so_analyze_stream pass `reps` times;
 * *elapsed_s = wall clock of the threaded region (no Python inside it). */
int so_analyze_streams_mt(uint32_t sample_rate, const float *interleaved, size_t n_samples_per_stream, size_t n_distinct,
                          size_t n_streams, size_t fft_n, size_t hop_frames, int n_threads, int reps, double *elapsed_s);

#ifdef __cplusplus
}
#endif
#endif
