/*
 * ss_oracle.c — CPU restatement of soundscope's analyzer hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY — see ss_oracle.h.  PARITY UNPINNED at the crate
 * boundary (ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 are not
 * vendored under /root/reference and cannot be built here).
 *
 * Every function cites the reference file:line it follows, or — for arithmetic
 * that lives in an un-vendored crate — the crate (pinned version from
 * /root/reference/Cargo.lock) and the published algorithm it restates.
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off: the Rust reference
 * never contracts a*b+c into an FMA, so neither may this file).
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime, pthread barriers (all-cores timing leg) */
#include "ss_oracle.h"

#include <float.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ======================================================================= *
 *  Spectrum path
 * ======================================================================= */

/* spectrum-analyzer 1.7.0 `windows::hann_window` (called at analyzer.rs:57):
 * periodic Hann, every step in f32:
 *   two_pi_i = 2.0 * PI * i as f32;  c = cosf(two_pi_i / n as f32);
 *   out[i] = (0.5 * (1.0 - c)) * x[i]
 * libm::cosf is within 1 ulp of the correctly rounded value; we take the
 * correctly rounded one. */
void so_hann_window(const float *x, size_t n, float *out)
{
    const float n_f = (float)n;
    const float two_pi = 2.0f * 3.14159265358979323846f;
    for (size_t i = 0; i < n; i++) {
        float two_pi_i = two_pi * (float)i;
        float arg = two_pi_i / n_f;
        float c = (float)cos((double)arg);
        float mult = 0.5f * (1.0f - c);
        out[i] = mult * x[i];
    }
}

/* twiddle cache: W_n^k = exp(-2*pi*i*k/n), k < n/2, rounded to f32 once
 * (microfft 0.6.0 keeps an f32 sine table of exactly these values). */
typedef struct { size_t n; float *re, *im; } tw_table;
static tw_table g_tw[40];

static const tw_table *twiddles(size_t n)
{
    int slot = 0;
    while (((size_t)1 << slot) < n) slot++;
    tw_table *t = &g_tw[slot];
    if (t->n == n) return t;
    size_t h = n / 2 ? n / 2 : 1;
    float *re = (float *)malloc(h * sizeof(float));
    float *im = (float *)malloc(h * sizeof(float));
    for (size_t k = 0; k < h; k++) {
        double ang = -2.0 * M_PI * (double)k / (double)n;
        re[k] = (float)cos(ang);
        im[k] = (float)sin(ang);
    }
    /* benign race: tests call single-threaded first */
    t->re = re; t->im = im; t->n = n;
    return t;
}

/* microfft 0.6.0 `cfft`: in-place radix-2 decimation-in-time, bit-reversal
 * first, butterflies  y = w * x[k+half];  x[k] = x_k + y;  x[k+half] = x_k - y
 * with num-complex multiplication (re = a.re*b.re - a.im*b.im,
 * im = a.re*b.im + a.im*b.re), all f32. */
static void cfft_radix2(float *re, float *im, size_t m)
{
    if (m < 2) return;
    /* bit reversal */
    for (size_t i = 1, j = 0; i < m; i++) {
        size_t bit = m >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    const tw_table *tw = twiddles(m);
    for (size_t len = 2; len <= m; len <<= 1) {
        size_t half = len >> 1, step = m / len;
        for (size_t s = 0; s < m; s += len) {
            for (size_t k = 0; k < half; k++) {
                float wr = tw->re[k * step], wi = tw->im[k * step];
                float xr = re[s + k + half], xi = im[s + k + half];
                float yr = wr * xr - wi * xi;
                float yi = wr * xi + wi * xr;
                float ar = re[s + k], ai = im[s + k];
                re[s + k] = ar + yr;        im[s + k] = ai + yi;
                re[s + k + half] = ar - yr; im[s + k + half] = ai - yi;
            }
        }
    }
}

/* microfft 0.6.0 `rfft`: pack n reals as n/2 complex, cfft, recombine
 *   X[k] = (Z[k]+conj(Z[m-k]))/2 - i*W_n^k*(Z[k]-conj(Z[m-k]))/2.
 * spectrum-analyzer then unpacks the Nyquist bin that microfft returns in
 * Im(X[0]) into its own bin n/2. */
void so_rfft(const float *x, size_t n, float *out_re, float *out_im)
{
    size_t m = n / 2;
    float *zr = (float *)malloc((m ? m : 1) * sizeof(float));
    float *zi = (float *)malloc((m ? m : 1) * sizeof(float));
    for (size_t i = 0; i < m; i++) { zr[i] = x[2 * i]; zi[i] = x[2 * i + 1]; }
    cfft_radix2(zr, zi, m);
    const tw_table *tw = twiddles(n);
    out_re[0] = zr[0] + zi[0]; out_im[0] = 0.0f;
    out_re[m] = zr[0] - zi[0]; out_im[m] = 0.0f;
    for (size_t k = 1; k < m; k++) {
        float ar = zr[k], ai = zi[k];
        float br = zr[m - k], bi = -zi[m - k];          /* conj */
        float sr = (ar + br) * 0.5f, si = (ai + bi) * 0.5f;
        float dr = (ar - br) * 0.5f, di = (ai - bi) * 0.5f;
        float wr = tw->re[k], wi = tw->im[k];
        float tr = wr * dr - wi * di;
        float ti = wr * di + wi * dr;
        out_re[k] = sr + ti;                           /* sum + (-i)*t */
        out_im[k] = si - tr;
    }
    free(zr); free(zi);
}

/* spectrum-analyzer 1.7.0 `fft_result_to_spectrum`: bin k has frequency
 * k as f32 * (sr as f32 / n as f32); FrequencyLimit::Range(20,20000)
 * (analyzer.rs:63) keeps 20 <= f <= 20000. */
size_t so_fft_bins(uint32_t sample_rate, size_t n, size_t *first_k)
{
    float res = (float)sample_rate / (float)n;
    size_t cnt = 0, first = 0;
    for (size_t k = 0; k <= n / 2; k++) {
        float f = (float)k * res;
        if (f >= 20.0f && f <= 20000.0f) { if (!cnt) first = k; cnt++; }
    }
    if (first_k) *first_k = first;
    return cnt;
}

static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* spectrum-analyzer 1.7.0 `FrequencySpectrum::calc_statistics`: min, max,
 * average and median (by sorting a copy).  The caller (analyzer.rs:11-27)
 * only reads stats.n, but the crate computes all of it — twice per spectrum
 * (before and after the scaling fn) — so the CPU baseline does too. */
static float spectrum_stats(const float *v, size_t cnt, float *scratch)
{
    if (!cnt) return 0.0f;
    memcpy(scratch, v, cnt * sizeof(float));
    qsort(scratch, cnt, sizeof(float), cmp_float);
    float sum = 0.0f;
    for (size_t i = 0; i < cnt; i++) sum += scratch[i];
    float avg = sum / (float)cnt;
    float med = (cnt & 1) ? scratch[cnt / 2] : 0.5f * (scratch[cnt / 2 - 1] + scratch[cnt / 2]);
    return avg + med + scratch[0] + scratch[cnt - 1];
}

volatile float so_stats_sink;

int so_get_fft_ex(uint32_t sample_rate, const float *x, size_t n,
                  double *out_xy, float *out_dbfs, size_t cap, size_t *out_n,
                  int with_stats)
{
    if (out_n) *out_n = 0;
    /* samples_fft_to_spectrum input checks, in the crate's order */
    if (n < 2) return SO_ERR_TOO_FEW_SAMPLES;
    float *win = (float *)malloc(n * sizeof(float));
    so_hann_window(x, n, win);                              /* analyzer.rs:57 */
    for (size_t i = 0; i < n; i++) if (isnan(win[i])) { free(win); return SO_ERR_NAN; }
    for (size_t i = 0; i < n; i++) if (isinf(win[i])) { free(win); return SO_ERR_INFINITY; }
    if (n & (n - 1)) { free(win); return SO_ERR_NOT_POW2; }
    float nyq = (float)sample_rate / 2.0f;
    if (20000.0f > nyq) { free(win); return SO_ERR_FREQ_LIMIT; } /* Range(20,20000).verify */
    /* the transform: microfft 0.6.0's real FFTs end at 32768 points and spectrum-analyzer 1.7.0 panics for longer inputs
     * (behind the checks above) — restated as a status, the library's SS_ERR_UNSUPPORTED */
    if (n > 32768) { free(win); return SO_ERR_UNSUPPORTED; }

    float *re = (float *)malloc((n / 2 + 1) * sizeof(float));
    float *im = (float *)malloc((n / 2 + 1) * sizeof(float));
    so_rfft(win, n, re, im);

    float res = (float)sample_rate / (float)n;
    float n_f = (float)n;                                    /* stats.n */
    size_t cnt = 0;
    int rc = SO_OK;
    float *vals = (float *)malloc((n / 2 + 1) * sizeof(float));
    float *freqs = (float *)malloc((n / 2 + 1) * sizeof(float));
    for (size_t k = 0; k <= n / 2; k++) {
        float f = (float)k * res;
        if (!(f >= 20.0f)) continue;
        if (!(f <= 20000.0f)) continue;
        /* complex_to_magnitude: sqrtf(re*re + im*im) */
        float sum = re[k] * re[k] + im[k] * im[k];
        vals[cnt] = sqrtf(sum);
        freqs[cnt] = f;
        cnt++;
    }
    if (with_stats) {
        float *scratch = (float *)malloc((cnt ? cnt : 1) * sizeof(float));
        so_stats_sink = spectrum_stats(vals, cnt, scratch);
        free(scratch);
    }
    /* scale_to_dbfs, analyzer.rs:11-27 */
    for (size_t i = 0; i < cnt; i++) {
        float v = vals[i];
        if (v == 0.0f) vals[i] = -150.0f;
        else {
            float scaled = v * 4.0f / n_f;
            vals[i] = 20.0f * log10f(scaled / 1.0f);
        }
        if (isnan(vals[i]) || isinf(vals[i])) rc = SO_ERR_SCALING;
    }
    if (with_stats) {
        float *scratch = (float *)malloc((cnt ? cnt : 1) * sizeof(float));
        so_stats_sink = spectrum_stats(vals, cnt, scratch);
        free(scratch);
    }
    if (rc == SO_OK && cnt > cap) rc = SO_ERR_CAPACITY;
    if (rc == SO_OK) {
        /* analyzer.rs:67-102: pink compensation and log-x mapping in f64 */
        const double min_freq_log = log10(20.0);
        const double max_freq_log = log10(20000.0);
        const double log_range = max_freq_log - min_freq_log;
        for (size_t i = 0; i < cnt; i++) {
            double freq = (double)freqs[i];
            double val = (double)vals[i];
            double comp = 10.0 * log10(freq / 1000.0);
            if (out_dbfs) out_dbfs[i] = vals[i];
            if (out_xy) {
                double log_freq = log10(freq);
                double norm = (log_freq - min_freq_log) / log_range;
                out_xy[2 * i] = norm * 100.0;
                out_xy[2 * i + 1] = val + comp;
            }
        }
        if (out_n) *out_n = cnt;
    }
    free(win); free(re); free(im); free(vals); free(freqs);
    return rc;
}

int so_get_fft(uint32_t sample_rate, const float *x, size_t n,
               double *out_xy, size_t cap_pairs, size_t *out_n)
{
    return so_get_fft_ex(sample_rate, x, n, out_xy, NULL, cap_pairs, out_n, 1);
}

/* ======================================================================= *
 *  Waveform: Analyzer::get_waveform, analyzer.rs:107-137
 * ======================================================================= */

/* f32::min / f32::max (IEEE minNum/maxNum): a NaN operand is ignored. */
static inline float f32_min(float a, float b) { return isnan(a) ? b : (isnan(b) ? a : (b < a ? b : a)); }
static inline float f32_max(float a, float b) { return isnan(a) ? b : (isnan(b) ? a : (b > a ? b : a)); }

size_t so_get_waveform(const float *x, size_t n, double window_s,
                       double *out_xy, size_t cap_pairs)
{
    double wd = window_s * 1000.0;
    /* Rust `as usize`: saturating, NaN -> 0 */
    size_t window = (wd != wd || wd <= 0.0) ? 0 : (wd >= 1.8446744073709552e19 ? SIZE_MAX : (size_t)wd);
    double spp = (double)n / (double)window;                /* analyzer.rs:109 */
    size_t np = 0;
    for (size_t i = 0; i < window; i++) {
        double sd = (double)i * spp;
        double ed = ceil((double)(i + 1) * spp);
        size_t start = (sd != sd || sd <= 0.0) ? 0 : (sd >= 1.8446744073709552e19 ? SIZE_MAX : (size_t)sd);
        size_t end = (ed != ed || ed <= 0.0) ? 0 : (ed >= 1.8446744073709552e19 ? SIZE_MAX : (size_t)ed);
        if (end > n) end = n;
        if (start >= n) break;                               /* analyzer.rs:122 */
        float mn = 0.0f, mx = 0.0f;
        if (end > start) {
            mn = x[start]; mx = x[start];
            for (size_t j = start + 1; j < end; j++) { mn = f32_min(mn, x[j]); mx = f32_max(mx, x[j]); }
        }
        if (np + 2 > cap_pairs) return np;
        out_xy[2 * np] = (double)i;     out_xy[2 * np + 1] = (double)mn; np++;
        out_xy[2 * np] = (double)i;     out_xy[2 * np + 1] = (double)mx; np++;
    }
    return np;
}

/* PCM sample conversion to f32 as symphonia 0.5 performs it for SampleBuffer::<f32>
 * (audio_player.rs:236-247 copy_interleaved_ref) [crate not vendored: conversions restated from its
 * documented FromSample impls; every scale is an exact power of two]:
 * fmt 1 u8, 2 s16, 3 s24 (packed LE), 4 s32, 5 f32, 6 f64 */
void so_pcm_to_f32(const unsigned char *src, size_t n, int fmt, float *dst)
{
    for (size_t i = 0; i < n; i++) {
        float v;
        switch (fmt) {
            case 1: v = (float)src[i] / 128.0f - 1.0f; break;
            case 2: { short s; memcpy(&s, src + 2 * i, 2); v = (float)s / 32768.0f; break; }
            case 3: { const unsigned char *q = src + 3 * i;
                      int s = (int)q[0] | ((int)q[1] << 8) | ((int)(signed char)q[2] << 16);
                      v = (float)s / 8388608.0f; break; }
            case 4: { int s; memcpy(&s, src + 4 * i, 4); v = (float)((double)s / 2147483648.0); break; }
            case 5: memcpy(&v, src + 4 * i, 4); break;
            default: { double d; memcpy(&d, src + 8 * i, 8); v = (float)d; break; }
        }
        dst[i] = v;
    }
}

/* get_mid_and_side_samples, audio_player.rs:400-419 */
size_t so_mid_side(const float *s, size_t n, float *mid, float *side)
{
    size_t f = n / 2;           /* zip drops an odd trailing sample */
    for (size_t i = 0; i < f; i++) {
        float l = s[2 * i], r = s[2 * i + 1];
        mid[i] = (l + r) / 2.0f;
        side[i] = (l - r) / 2.0f;
    }
    return f;
}

/* ======================================================================= *
 *  Loudness meter: ebur128 0.1.10 with Mode::all() (analyzer.rs:36,51,171)
 *  — a Rust port of libebur128; algorithm per ITU-R BS.1770-4 / EBU R128.
 * ======================================================================= */

enum { CH_UNUSED = 0, CH_LEFT, CH_RIGHT, CH_CENTER, CH_LS, CH_RS, CH_DUAL_MONO };

#define HIST_BINS 1000
static double g_hist_energies[HIST_BINS];
static double g_hist_bounds[HIST_BINS + 1];
static int g_hist_ready;

static void hist_init(void)
{
    if (g_hist_ready) return;
    g_hist_bounds[0] = pow(10.0, (-70.0 + 0.691) / 10.0);
    for (int i = 0; i < HIST_BINS; i++)
        g_hist_energies[i] = pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
    for (int i = 1; i <= HIST_BINS; i++)
        g_hist_bounds[i] = pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
    g_hist_ready = 1;
}

static size_t find_histogram_index(double energy)
{
    size_t lo = 0, hi = HIST_BINS;
    do {
        size_t mid = (lo + hi) / 2;
        if (energy >= g_hist_bounds[mid]) lo = mid; else hi = mid;
    } while (hi - lo != 1);
    return lo;
}

static double energy_to_loudness(double e) { return 10.0 * log10(e) - 0.691; }

typedef struct { int count; int index[64]; float coeff[64]; } interp_phase;

struct so_meter {
    uint32_t channels, rate;
    int channel_map[64];
    size_t s100, audio_data_frames, audio_data_index, needed_frames, st_counter;
    double *audio_data;
    double b[5], a[5];
    double v[64][5];
    double sample_peak[64], true_peak[64];
    uint64_t block_hist[HIST_BINS], st_hist[HIST_BINS];
    int ftz_mode;                /* SO_FTZ_*: how sub-normal filter state is flushed (so_meter_set_ftz) */
    /* true-peak interpolator */
    int tp_factor, tp_delay, tp_zi;
    interp_phase tp_phase[4];
    float *tp_z;                 /* [channels][delay] */
};

/* libebur128 / ebur128 crate interpolator design: 49-tap Hann-windowed sinc,
 * split into `factor` polyphase sub-filters, coefficients with |c| <= 1e-6
 * dropped.  factor 4 -> taps/phase [1,12,12,12], delay 13. */
static void interp_design(int taps, int factor, interp_phase *ph, int *delay)
{
    for (int f = 0; f < factor; f++) ph[f].count = 0;
    *delay = (taps + factor - 1) / factor;
    for (int j = 0; j < taps; j++) {
        double m = (double)j - (double)(taps - 1) / 2.0;
        double c = 1.0;
        if (fabs(m) > 0.000001) c = sin(m * M_PI / factor) / (m * M_PI / factor);
        c *= 0.5 * (1.0 - cos(2.0 * M_PI * j / (taps - 1)));
        if (fabs(c) > 0.000001) {
            int f = j % factor;
            int t = ph[f].count++;
            ph[f].coeff[t] = (float)c;      /* the Rust port keeps f32 taps */
            ph[f].index[t] = j / factor;
        }
    }
}

int so_interp_layout(int taps, int factor, int *counts, int *delay)
{
    interp_phase ph[8];
    if (factor < 1 || factor > 8 || taps > 64 * factor) return -1;
    interp_design(taps, factor, ph, delay);
    for (int f = 0; f < factor; f++) counts[f] = ph[f].count;
    return 0;
}

size_t so_interp_coeffs(int taps, int factor, int phase, float *coeff, int *index, size_t cap)
{
    interp_phase ph[8]; int delay;
    if (factor < 1 || factor > 8 || phase >= factor) return 0;
    interp_design(taps, factor, ph, &delay);
    size_t c = (size_t)ph[phase].count;
    for (size_t i = 0; i < c && i < cap; i++) { coeff[i] = ph[phase].coeff[i]; index[i] = ph[phase].index[i]; }
    return c;
}

/* BS.1770 K-weighting as one 4th-order section (libebur128 ebur128_init_filter):
 * bilinear high-shelf x high-pass, numerators/denominators convolved. */
static void kweight_design(double rate, double b[5], double a[5])
{
    double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
    double K = tan(M_PI * f0 / rate);
    double Vh = pow(10.0, G / 20.0);
    double Vb = pow(Vh, 0.4996667741545416);
    double pb[3], pa[3] = {1.0, 0.0, 0.0}, rb[3] = {1.0, -2.0, 1.0}, ra[3] = {1.0, 0.0, 0.0};
    double a0 = 1.0 + K / Q + K * K;
    pb[0] = (Vh + Vb * K / Q + K * K) / a0;
    pb[1] = 2.0 * (K * K - Vh) / a0;
    pb[2] = (Vh - Vb * K / Q + K * K) / a0;
    pa[1] = 2.0 * (K * K - 1.0) / a0;
    pa[2] = (1.0 - K / Q + K * K) / a0;
    f0 = 38.13547087602444; Q = 0.5003270373238773;
    K = tan(M_PI * f0 / rate);
    ra[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
    ra[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
    b[0] = pb[0] * rb[0];
    b[1] = pb[0] * rb[1] + pb[1] * rb[0];
    b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
    b[3] = pb[1] * rb[2] + pb[2] * rb[1];
    b[4] = pb[2] * rb[2];
    a[0] = pa[0] * ra[0];
    a[1] = pa[0] * ra[1] + pa[1] * ra[0];
    a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
    a[3] = pa[1] * ra[2] + pa[2] * ra[1];
    a[4] = pa[2] * ra[2];
}

int so_meter_new_ex(uint32_t channels, uint32_t rate, int force_tp_factor, so_meter **out)
{
    *out = NULL;
    /* EbuR128::new: channels 1..=64, rate 16..=2_822_400, else Error::NoMem */
    if (channels == 0 || channels > 64) return SO_ERR_NOMEM;
    if (rate < 16 || rate > 2822400) return SO_ERR_NOMEM;
    hist_init();
    so_meter *m = (so_meter *)calloc(1, sizeof(so_meter));
    if (!m) return SO_ERR_NOMEM;
    m->channels = channels; m->rate = rate;
    for (uint32_t i = 0; i < channels; i++) {
        int c = CH_UNUSED;
        if (channels == 4) { static const int q[4] = {CH_LEFT, CH_RIGHT, CH_LS, CH_RS}; c = q[i]; }
        else if (channels == 5) { static const int q[5] = {CH_LEFT, CH_RIGHT, CH_CENTER, CH_LS, CH_RS}; c = q[i]; }
        else switch (i) { case 0: c = CH_LEFT; break; case 1: c = CH_RIGHT; break; case 2: c = CH_CENTER; break;
                          case 3: c = CH_UNUSED; break; case 4: c = CH_LS; break; case 5: c = CH_RS; break; default: c = CH_UNUSED; }
        m->channel_map[i] = c;
    }
    m->s100 = ((size_t)rate + 5) / 10;
    /* Mode::all() contains S => 3000 ms window, rounded up to a 100 ms multiple */
    m->audio_data_frames = (size_t)rate * 3000 / 1000;
    if (m->audio_data_frames % m->s100) m->audio_data_frames += m->s100 - (m->audio_data_frames % m->s100);
    m->audio_data = (double *)calloc(m->audio_data_frames * channels, sizeof(double));
    if (!m->audio_data) { free(m); return SO_ERR_NOMEM; }
    m->needed_frames = m->s100 * 4;
    kweight_design((double)rate, m->b, m->a);
    /* true-peak oversampling rule of the crate: <96k: 4x, <192k: 2x, else none */
    int factor = force_tp_factor ? force_tp_factor : (rate < 96000 ? 4 : (rate < 192000 ? 2 : 0));
    m->tp_factor = factor;
    if (factor) {
        interp_design(49, factor, m->tp_phase, &m->tp_delay);
        m->tp_z = (float *)calloc((size_t)channels * m->tp_delay, sizeof(float));
    }
    *out = m;
    return SO_OK;
}

int so_meter_new(uint32_t channels, uint32_t rate, so_meter **out)
{
    return so_meter_new_ex(channels, rate, 0, out);
}

void so_meter_free(so_meter *m)
{
    if (!m) return;
    free(m->audio_data); free(m->tp_z); free(m);
}

void so_meter_reset(so_meter *m)
{
    memset(m->audio_data, 0, m->audio_data_frames * m->channels * sizeof(double));
    m->needed_frames = m->s100 * 4;
    m->audio_data_index = 0;
    m->st_counter = 0;
    memset(m->block_hist, 0, sizeof m->block_hist);
    memset(m->st_hist, 0, sizeof m->st_hist);
    memset(m->sample_peak, 0, sizeof m->sample_peak);
    memset(m->true_peak, 0, sizeof m->true_peak);
    memset(m->v, 0, sizeof m->v);
    if (m->tp_z) memset(m->tp_z, 0, (size_t)m->channels * m->tp_delay * sizeof(float));
    m->tp_zi = 0;
}

int so_meter_set_ftz(so_meter *m, int mode)
{
    if (!m || (mode != SO_FTZ_END_OF_CALL && mode != SO_FTZ_PER_OP)) return SO_ERR_INVALID_MODE;
    m->ftz_mode = mode;
    return SO_OK;
}

void so_meter_filter_coeffs(so_meter *m, double b[5], double a[5])
{
    memcpy(b, m->b, sizeof m->b); memcpy(a, m->a, sizeof m->a);
}

/* carried DF-II state v1..v4 of one channel (what Filter::process leaves behind, sub-normals flushed) */
int so_meter_filter_state(so_meter *m, uint32_t ch, double v4[4])
{
    if (ch >= m->channels) return SO_ERR_INVALID_CHANNEL;
    for (int k = 0; k < 4; k++) v4[k] = m->v[ch][k + 1];
    return SO_OK;
}

/* Filter::process: sample peak, true peak (polyphase FIR, f32), K-weighting
 * (DF-II, f64) into the ring buffer at audio_data_index. */
static void filter_process(so_meter *m, const float *src, size_t frames)
{
#if defined(__x86_64__) && defined(__SSE2_MATH__)
    /* SO_FTZ_PER_OP: flush-to-zero for the whole of Filter::process (peaks, interpolator, filter), like the crate's x86 build */
    unsigned int mxcsr_saved = 0;
    if (m->ftz_mode == SO_FTZ_PER_OP) { mxcsr_saved = __builtin_ia32_stmxcsr(); __builtin_ia32_ldmxcsr(mxcsr_saved | 0x8000u); }
#endif
    const uint32_t C = m->channels;
    /* sample peak */
    for (uint32_t c = 0; c < C; c++) {
        double mx = m->sample_peak[c];
        for (size_t i = 0; i < frames; i++) {
            double v = fabs((double)src[i * C + c]);
            if (v > mx) mx = v;
        }
        m->sample_peak[c] = mx;
    }
    /* true peak */
    if (m->tp_factor) {
        const int D = m->tp_delay;
        int zi = m->tp_zi;
        for (size_t i = 0; i < frames; i++) {
            for (uint32_t c = 0; c < C; c++) {
                float *z = m->tp_z + (size_t)c * D;
                z[zi] = src[i * C + c];
                double pk = m->true_peak[c];
                for (int f = 0; f < m->tp_factor; f++) {
                    const interp_phase *ph = &m->tp_phase[f];
                    float acc = 0.0f;
                    for (int t = 0; t < ph->count; t++) {
                        int k = zi - ph->index[t];
                        if (k < 0) k += D;
                        acc += z[k] * ph->coeff[t];
                    }
                    double av = fabs((double)acc);
                    if (av > pk) pk = av;
                }
                m->true_peak[c] = pk;
            }
            if (++zi == D) zi = 0;
        }
        m->tp_zi = zi;
    }
    /* K-weighting.  Sub-normals, [RECALLED] from ebur128 0.1.10's filter.rs (after libebur128's TURN_ON_FTZ / FLUSH_MANUALLY):
     *   SO_FTZ_END_OF_CALL  builds without SSE2 (and the crate's own fallback): gradual underflow inside the call, the four
     *                       state variables flushed to zero at its end;
     *   SO_FTZ_PER_OP       x86 / x86-64 builds (what a user of the reference runs): MXCSR.FTZ is set for the duration of
     *                       Filter::process, every sub-normal RESULT of an SSE operation becomes a signed zero, and there is
     *                       no flush at the end.  Restated with the hardware bit itself where this file is built for x86-64
     *                       (the loop below is SSE2 scalar arithmetic, -ffp-contract=off), by a per-operation software flush
     *                       elsewhere.
     * The two differ only while the state decays through 2.2e-308 .. 4.9e-324 (about 2.9 s of digital silence behind a
     * programme at 48 kHz): the filtered samples there square to zero in either, so no reading of the meter can tell them
     * apart — tests/test_oracle_known_answers.py::test_filter_ftz_models_agree_on_every_reading pins that. */
    const double *a = m->a, *b = m->b;
    double *dst = m->audio_data + m->audio_data_index;
#if defined(__x86_64__) && defined(__SSE2_MATH__)
#define SO_FZ(x) (x)
#else
#define SO_FZ(x) ((m->ftz_mode == SO_FTZ_PER_OP && fabs(x) < DBL_MIN) ? copysign(0.0, (x)) : (x))
#endif
    for (uint32_t c = 0; c < C; c++) {
        if (m->channel_map[c] == CH_UNUSED) continue;
        double *v = m->v[c];
        for (size_t i = 0; i < frames; i++) {
            double t = (double)src[i * C + c];
            t = SO_FZ(t - SO_FZ(a[1] * v[1])); t = SO_FZ(t - SO_FZ(a[2] * v[2]));
            t = SO_FZ(t - SO_FZ(a[3] * v[3])); t = SO_FZ(t - SO_FZ(a[4] * v[4]));
            v[0] = t;
            double y = SO_FZ(b[0] * v[0]);
            y = SO_FZ(y + SO_FZ(b[1] * v[1])); y = SO_FZ(y + SO_FZ(b[2] * v[2]));
            y = SO_FZ(y + SO_FZ(b[3] * v[3])); y = SO_FZ(y + SO_FZ(b[4] * v[4]));
            dst[i * C + c] = y;
            v[4] = v[3]; v[3] = v[2]; v[2] = v[1]; v[1] = v[0];
        }
        /* denormal flush after every call (the build without hardware flush-to-zero) */
        if (m->ftz_mode == SO_FTZ_END_OF_CALL)
            for (int k = 1; k <= 4; k++) if (fabs(v[k]) < DBL_MIN) v[k] = 0.0;
    }
#undef SO_FZ
#if defined(__x86_64__) && defined(__SSE2_MATH__)
    if (m->ftz_mode == SO_FTZ_PER_OP) __builtin_ia32_ldmxcsr(mxcsr_saved);
#endif
}

/* calc_gating_block: mean square over the last frames_per_block frames of the
 * ring, channel-weighted (1.0 L/R/C, 1.41 surrounds, 2.0 dual mono). */
static double calc_gating_block(so_meter *m, size_t fpb, int add)
{
    const uint32_t C = m->channels;
    const size_t idx_frames = m->audio_data_index / C;
    double sum = 0.0;
    for (uint32_t c = 0; c < C; c++) {
        if (m->channel_map[c] == CH_UNUSED) continue;
        double cs = 0.0;
        if (m->audio_data_index < fpb * C) {
            for (size_t i = 0; i < idx_frames; i++) { double y = m->audio_data[i * C + c]; cs += y * y; }
            for (size_t i = m->audio_data_frames - (fpb - idx_frames); i < m->audio_data_frames; i++) {
                double y = m->audio_data[i * C + c]; cs += y * y;
            }
        } else {
            for (size_t i = idx_frames - fpb; i < idx_frames; i++) { double y = m->audio_data[i * C + c]; cs += y * y; }
        }
        if (m->channel_map[c] == CH_LS || m->channel_map[c] == CH_RS) cs *= 1.41;
        else if (m->channel_map[c] == CH_DUAL_MONO) cs *= 2.0;
        sum += cs;
    }
    sum /= (double)fpb;
    if (add && sum >= g_hist_bounds[0]) m->block_hist[find_histogram_index(sum)]++;
    return sum;
}

/* EbuR128::add_frames_f32 (called at analyzer.rs:140 and :176) */
int so_meter_add_frames_f32(so_meter *m, const float *src, size_t n_samples)
{
    const uint32_t C = m->channels;
    if (n_samples == 0) return SO_OK;
    if (n_samples % C) return SO_ERR_NOMEM;
    size_t frames = n_samples / C, src_index = 0;
    while (frames > 0) {
        if (frames >= m->needed_frames) {
            filter_process(m, src + src_index, m->needed_frames);
            src_index += m->needed_frames * C;
            frames -= m->needed_frames;
            m->audio_data_index += m->needed_frames * C;
            calc_gating_block(m, m->s100 * 4, 1);                       /* Mode::I */
            m->st_counter += m->needed_frames;                          /* Mode::LRA */
            if (m->st_counter == m->s100 * 30) {
                /* `if let Ok(st_energy) = self.energy_shortterm()`: energy_in_interval refuses an interval longer than the ring
                 * (InvalidMode) — thirty sub-blocks are at rates under 145 Hz that round up to their sub-block (16 Hz: 60 > 48 frames) —
                 * and the block is skipped */
                if (m->s100 * 30 <= m->audio_data_frames) {
                    double e = calc_gating_block(m, m->s100 * 30, 0);
                    if (e >= g_hist_bounds[0]) m->st_hist[find_histogram_index(e)]++;
                }
                m->st_counter = m->s100 * 20;
            }
            m->needed_frames = m->s100;
            if (m->audio_data_index == m->audio_data_frames * C) m->audio_data_index = 0;
        } else {
            filter_process(m, src + src_index, frames);
            m->audio_data_index += frames * C;
            m->st_counter += frames;
            m->needed_frames -= frames;
            frames = 0;
        }
    }
    return SO_OK;
}

int so_meter_loudness_momentary(so_meter *m, double *out)
{
    double e = calc_gating_block(m, m->s100 * 4, 0);
    *out = e <= 0.0 ? -INFINITY : energy_to_loudness(e);
    return SO_OK;
}

/* loudness_shortterm (analyzer.rs:148): last 3 s of the ring "as is" */
int so_meter_loudness_shortterm(so_meter *m, double *out)
{
    size_t fr = m->s100 * 30;
    if (fr > m->audio_data_frames) return SO_ERR_INVALID_MODE;
    double e = calc_gating_block(m, fr, 0);
    *out = e <= 0.0 ? -INFINITY : energy_to_loudness(e);
    return SO_OK;
}

/* gated loudness on a (possibly summed) histogram: -10 LU relative gate */
double so_gated_loudness_hist(const uint64_t *hist)
{
    hist_init();
    double rel = 0.0; uint64_t cnt = 0;
    for (int i = 0; i < HIST_BINS; i++) { rel += (double)hist[i] * g_hist_energies[i]; cnt += hist[i]; }
    if (!cnt) return -INFINITY;
    rel /= (double)cnt;
    rel *= pow(10.0, -10.0 / 10.0);
    size_t start;
    if (rel < g_hist_bounds[0]) start = 0;
    else { start = find_histogram_index(rel); if (rel > g_hist_energies[start]) start++; }
    double g = 0.0; cnt = 0;
    for (size_t i = start; i < HIST_BINS; i++) { g += (double)hist[i] * g_hist_energies[i]; cnt += hist[i]; }
    if (!cnt) return -INFINITY;
    return energy_to_loudness(g / (double)cnt);
}

int so_meter_loudness_global(so_meter *m, double *out)       /* analyzer.rs:152,181 */
{
    *out = so_gated_loudness_hist(m->block_hist);
    return SO_OK;
}

/* EBU Tech 3342 loudness range on the short-term histogram */
double so_loudness_range_hist(const uint64_t *h)
{
    hist_init();
    uint64_t size = 0; double power = 0.0;
    for (int j = 0; j < HIST_BINS; j++) { size += h[j]; power += (double)h[j] * g_hist_energies[j]; }
    if (!size) return 0.0;
    power /= (double)size;
    double integ = pow(10.0, -20.0 / 10.0) * power;
    size_t index;
    if (integ < g_hist_bounds[0]) index = 0;
    else { index = find_histogram_index(integ); if (integ > g_hist_energies[index]) index++; }
    size = 0;
    for (size_t j = index; j < HIST_BINS; j++) size += h[j];
    if (!size) return 0.0;
    uint64_t plow = (uint64_t)((double)(size - 1) * 0.1 + 0.5);
    uint64_t phigh = (uint64_t)((double)(size - 1) * 0.95 + 0.5);
    size = 0; size_t j = index;
    while (size <= plow) size += h[j++];
    double l_en = g_hist_energies[j - 1];
    while (size <= phigh) size += h[j++];
    double h_en = g_hist_energies[j - 1];
    return energy_to_loudness(h_en) - energy_to_loudness(l_en);
}

int so_meter_loudness_range(so_meter *m, double *out)        /* analyzer.rs:156 */
{
    *out = so_loudness_range_hist(m->st_hist);
    return SO_OK;
}

int so_meter_sample_peak(so_meter *m, uint32_t ch, double *out)
{
    if (ch >= m->channels) return SO_ERR_INVALID_CHANNEL;
    *out = m->sample_peak[ch];
    return SO_OK;
}

/* true_peak(ch) (analyzer.rs:160-161): max(true peak, sample peak), linear */
int so_meter_true_peak(so_meter *m, uint32_t ch, double *out)
{
    if (ch >= m->channels) return SO_ERR_INVALID_CHANNEL;
    *out = m->true_peak[ch] > m->sample_peak[ch] ? m->true_peak[ch] : m->sample_peak[ch];
    return SO_OK;
}

const uint64_t *so_meter_block_hist(so_meter *m) { return m->block_hist; }
const uint64_t *so_meter_st_hist(so_meter *m) { return m->st_hist; }

/* Analyzer::calculate_integrated_lufs, analyzer.rs:170-182 */
int so_calculate_integrated_lufs(uint32_t sample_rate, uint32_t channels,
                                 const float *x, size_t n, double *out)
{
    so_meter *m;
    int rc = so_meter_new(channels, sample_rate, &m);
    if (rc) return rc;
    size_t chunk = (size_t)sample_rate * 2;                 /* analyzer.rs:175 */
    for (size_t off = 0; off < n; off += chunk) {
        size_t len = n - off < chunk ? n - off : chunk;
        rc = so_meter_add_frames_f32(m, x + off, len);
        if (rc) { so_meter_free(m); return rc; }
    }
    rc = so_meter_loudness_global(m, out);
    so_meter_free(m);
    return rc;
}

/* One pass of the whole hot path over one stereo stream (CPU baseline and the
 * batch parity oracle).  Cadence per tui.rs:1482-1526: window [p-N,p) for
 * p = k*hop, skipped when p-N == 0; waveform per tui.rs:1213-1216; meter per
 * analyzer.rs:170-182 (one-shot feed in 2*sr chunks). */
int so_analyze_stream(uint32_t sample_rate, const float *x, size_t n_samples,
                      size_t fft_n, size_t hop, int force_tp_factor,
                      float *fft_out, double *wave_out, so_stream_result *res)
{
    size_t F = n_samples / 2;
    memset(res, 0, sizeof *res);
    /* waveform over the interleaved buffer, window = duration */
    double dur = (double)F / (double)sample_rate;
    size_t W = (size_t)(dur * 1000.0);
    double *wtmp = wave_out ? wave_out : (double *)malloc((W * 4 + 4) * sizeof(double));
    res->n_wave_points = so_get_waveform(x, n_samples, dur, wtmp, W * 2 + 2);
    if (!wave_out) free(wtmp);
    /* spectrum */
    float *mid = (float *)malloc((F ? F : 1) * sizeof(float));
    float *side = (float *)malloc((F ? F : 1) * sizeof(float));
    so_mid_side(x, n_samples, mid, side);
    size_t first_k; size_t nb = so_fft_bins(sample_rate, fft_n, &first_k);
    res->n_bins = nb;
    double *xy = (double *)malloc((fft_n / 2 + 1) * 2 * sizeof(double));
    size_t nwin = 0;
    for (size_t p = hop; p <= F; p += hop) {
        if (p <= fft_n) continue;                          /* saturating_sub == 0 -> skip */
        for (int ch = 0; ch < 2; ch++) {
            size_t cnt;
            int rc = so_get_fft(sample_rate, (ch ? side : mid) + (p - fft_n), fft_n, xy, fft_n / 2 + 1, &cnt);
            if (rc) { free(mid); free(side); free(xy); return rc; }
            if (fft_out) for (size_t i = 0; i < cnt; i++) fft_out[(nwin * 2 + ch) * nb + i] = (float)xy[2 * i + 1];
        }
        nwin++;
    }
    res->n_windows = nwin;
    free(mid); free(side); free(xy);
    /* loudness + peaks */
    so_meter *m;
    int rc = so_meter_new_ex(2, sample_rate, force_tp_factor, &m);
    if (rc) return rc;
    size_t chunk = (size_t)sample_rate * 2;
    for (size_t off = 0; off < n_samples; off += chunk) {
        size_t len = n_samples - off < chunk ? n_samples - off : chunk;
        rc = so_meter_add_frames_f32(m, x + off, len);
        if (rc) { so_meter_free(m); return rc; }
    }
    so_meter_loudness_global(m, &res->integrated);
    so_meter_loudness_range(m, &res->lra);
    for (uint32_t c = 0; c < 2; c++) {
        so_meter_true_peak(m, c, &res->true_peak[c]);
        so_meter_sample_peak(m, c, &res->sample_peak[c]);
    }
    so_meter_free(m);
    return SO_OK;
}

/* ---- all-cores timing leg of bench.py's cpu_baseline (no Python inside the timed region) ---------------------------
 * n_streams equal-length stereo streams (stream s reads buffer s % n_distinct), dealt round-robin to n_threads POSIX threads; every thread runs the whole
 * so_analyze_stream pass (waveform + mid/side + two spectra per window incl. the crate's stats sorts + meter with true
 * peak) on its streams, `reps` times.  Returns the wall-clock seconds between the start barrier and the last join. */
#include <malloc.h>
#include <pthread.h>
#include <time.h>
typedef struct {
    uint32_t rate; const float *x; size_t n_samples, n_streams, n_distinct, fft_n, hop; int tid, n_threads, reps, rc;
    pthread_barrier_t *start;
} mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    size_t first_k, nb = so_fft_bins(j->rate, j->fft_n, &first_k);
    size_t F = j->n_samples / 2, nwin = F / j->hop > j->fft_n / j->hop ? F / j->hop - j->fft_n / j->hop : 0;
    float *fft_out = (float *)malloc((nwin * 2 * nb + 1) * sizeof(float));
    pthread_barrier_wait(j->start);
    for (int r = 0; r < j->reps; r++)
        for (size_t s = (size_t)j->tid; s < j->n_streams; s += (size_t)j->n_threads) {
            so_stream_result res;
            int rc = so_analyze_stream(j->rate, j->x + (s % j->n_distinct) * j->n_samples, j->n_samples, j->fft_n, j->hop, 0, fft_out, NULL, &res);
            if (rc) j->rc = rc;
        }
    free(fft_out);
    return NULL;
}

int so_analyze_streams_mt(uint32_t rate, const float *x, size_t n_samples, size_t n_distinct, size_t n_streams, size_t fft_n,
                          size_t hop, int n_threads, int reps, double *elapsed_s)
{
    if (n_threads < 1 || !n_streams || !n_distinct || !elapsed_s) return SO_ERR_NOMEM;
    /* The pass allocates like the reference does (per-stream mid / side / chart vectors of megabytes, per-window vectors):
     * with glibc's defaults every such block is its own mmap / munmap, and hundreds of threads of ONE process then queue on
     * the address-space lock for the page faults (measured on the 256-core GPU host: 41 Msamples/s, 9.5x one core).  Keep
     * freed blocks in the per-thread arenas instead, as any allocator tuned for a threaded server would. */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_ARENA_MAX, n_threads > 8 ? n_threads : 8);
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof *th);
    mt_job *jobs = (mt_job *)calloc((size_t)n_threads, sizeof *jobs);
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)n_threads + 1);
    hist_init();                                  /* shared tables are built before the threads start */
    int started = 0;
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (mt_job){rate, x, n_samples, n_streams, n_distinct, fft_n, hop, t, n_threads, reps, 0, &start};
        if (pthread_create(&th[t], NULL, mt_worker, &jobs[t]) != 0) break;
        started++;
    }
    int rc = SO_OK;
    if (started != n_threads) {                   /* cannot release the barrier with fewer threads: fail loudly */
        fprintf(stderr, "so_analyze_streams_mt: only %d of %d threads started\n", started, n_threads);
        abort();
    }
    struct timespec t0, t1;
    pthread_barrier_wait(&start);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; t++) { pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *elapsed_s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    pthread_barrier_destroy(&start);
    free(th); free(jobs);
    return rc;
}
