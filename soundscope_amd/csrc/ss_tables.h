// ss_tables.h — host-side design of every constant table the kernels consume.
// Pure C++ (no HIP).  Each function cites the reference behaviour it encodes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace sst {

constexpr int kHistBins = 1000;

// spectrum-analyzer 1.7.0 windows::hann_window (used at analyzer.rs:57):
// w[i] = 0.5 * (1 - cosf(2*pi*i / n)), every step in f32, periodic.
std::vector<float> hann_window(size_t n);

// W_n^k = exp(-2*pi*i*k/n) rounded to f32, k in [0, count)
void twiddles(size_t n, size_t count, std::vector<float> &re_im_interleaved);

// retained bins for FrequencyLimit::Range(20, 20000) (analyzer.rs:63): bin k has
// frequency k as f32 * (sr as f32 / n as f32); returns count, sets first k.
size_t fft_bins(uint32_t sample_rate, size_t n, size_t *first_k);
// per retained bin: frequency (f32 as the crate computes it, widened), pink
// compensation 10*log10(f/1000) (analyzer.rs:82) and chart x (analyzer.rs:88-98), f64
void bin_tables(uint32_t sample_rate, size_t n, std::vector<double> &freq,
                std::vector<double> &pink_db, std::vector<double> &chart_x);

// BS.1770 K-weighting as ebur128 0.1.10 designs it (one 4th-order DF-II section)
void kweight_design(double rate, double b[5], double a[5]);
// zero-input state transition of the DF-II state (v1..v4) over `steps` samples,
// row-major 4x4
double kweight_pole_radius(double rate);      // largest pole radius of the K-weighting filter (>= 1: not stable at this rate)
void kweight_transition_pow(const double a[5], uint64_t steps, double out[16]);

// ebur128 true-peak interpolator: 49-tap Hann-windowed sinc split into
// `factor` polyphase branches, |c| <= 1e-6 dropped.  taps[f] lists
// (delay index, coefficient) in ascending delay; coefficient kept in f32.
struct PolyTap { int delay; float coeff; };
void true_peak_design(int factor, std::vector<std::vector<PolyTap>> &phases, int *delay_len);
// oversampling rule of the crate: <96 kHz: 4, <192 kHz: 2, else 0 (off)
int true_peak_factor_for_rate(uint32_t rate);

// ebur128 histogram tables: 1000 representative energies, 1001 boundaries
void histogram_tables(double energies[kHistBins], double bounds[kHistBins + 1]);
// loudness_global / loudness_range on a histogram (ebur128 histogram mode);
// host copies used for the corpus gate after an all-reduce
double gated_loudness(const uint64_t *hist);
double loudness_range(const uint64_t *st_hist);

// default channel map weights of ebur128 (1.0 L/R/C, 1.41 surrounds, 0 unused)
void channel_weights(uint32_t channels, double *w);

}  // namespace sst
