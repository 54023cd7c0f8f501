// ss_tables.cpp — host-side constant design (see ss_tables.h).
#include "ss_tables.h"

#include <cmath>
#include <cstring>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace sst {

std::vector<float> hann_window(size_t n)
{
    std::vector<float> w(n);
    const float n_f = (float)n;
    const float two_pi = 2.0f * 3.14159265358979323846f;
    for (size_t i = 0; i < n; i++) {
        float two_pi_i = two_pi * (float)i;          // 2.0 * PI * i as f32
        float arg = two_pi_i / n_f;
        float c = (float)std::cos((double)arg);      // libm::cosf, correctly rounded
        w[i] = 0.5f * (1.0f - c);
    }
    return w;
}

void twiddles(size_t n, size_t count, std::vector<float> &out)
{
    out.resize(2 * count);
    for (size_t k = 0; k < count; k++) {
        double ang = -2.0 * M_PI * (double)k / (double)n;
        out[2 * k] = (float)std::cos(ang);
        out[2 * k + 1] = (float)std::sin(ang);
    }
}

size_t fft_bins(uint32_t sample_rate, size_t n, size_t *first_k)
{
    float res = (float)sample_rate / (float)n;
    size_t cnt = 0, first = 0;
    for (size_t k = 0; k <= n / 2; k++) {
        float f = (float)k * res;
        if (f >= 20.0f && f <= 20000.0f) {
            if (!cnt) first = k;
            cnt++;
        }
    }
    if (first_k) *first_k = first;
    return cnt;
}

void bin_tables(uint32_t sample_rate, size_t n, std::vector<double> &freq,
                std::vector<double> &pink_db, std::vector<double> &chart_x)
{
    size_t first;
    size_t cnt = fft_bins(sample_rate, n, &first);
    freq.resize(cnt); pink_db.resize(cnt); chart_x.resize(cnt);
    float res = (float)sample_rate / (float)n;
    const double min_log = std::log10(20.0), max_log = std::log10(20000.0);
    const double range = max_log - min_log;
    for (size_t i = 0; i < cnt; i++) {
        double f = (double)((float)(first + i) * res);
        freq[i] = f;
        pink_db[i] = 10.0 * std::log10(f / 1000.0);
        chart_x[i] = (std::log10(f) - min_log) / range * 100.0;
    }
}

void kweight_design(double rate, double b[5], double a[5])
{
    double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
    double K = std::tan(M_PI * f0 / rate);
    double Vh = std::pow(10.0, G / 20.0);
    double Vb = std::pow(Vh, 0.4996667741545416);
    double pb[3], pa[3] = {1.0, 0.0, 0.0}, rb[3] = {1.0, -2.0, 1.0}, ra[3] = {1.0, 0.0, 0.0};
    double a0 = 1.0 + K / Q + K * K;
    pb[0] = (Vh + Vb * K / Q + K * K) / a0;
    pb[1] = 2.0 * (K * K - Vh) / a0;
    pb[2] = (Vh - Vb * K / Q + K * K) / a0;
    pa[1] = 2.0 * (K * K - 1.0) / a0;
    pa[2] = (1.0 - K / Q + K * K) / a0;
    f0 = 38.13547087602444; Q = 0.5003270373238773;
    K = std::tan(M_PI * f0 / rate);
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

// largest pole radius of the K-weighting filter at `rate` (the two biquads of kweight_design, each on its own: the roots of
// z^2 + a1 z + a2).  Below 1 the filter forgets its state like radius^n; the crate accepts rates (16 Hz .. ~3.4 kHz) at which the
// 1682 Hz shelf lies beyond Nyquist and the design is not stable at all.
double kweight_pole_radius(double rate)
{
    auto radius = [](double a1, double a2) {
        const double disc = a1 * a1 - 4.0 * a2;
        if (disc < 0.0) return std::sqrt(a2);                                   // complex pair: |z|^2 = a2
        const double r = std::sqrt(disc);
        return std::fmax(std::fabs((-a1 + r) * 0.5), std::fabs((-a1 - r) * 0.5));
    };
    double f0 = 1681.974450955533, Q = 0.7071752369554196;
    double K = std::tan(M_PI * f0 / rate);
    double a0 = 1.0 + K / Q + K * K;
    const double r_shelf = radius(2.0 * (K * K - 1.0) / a0, (1.0 - K / Q + K * K) / a0);
    f0 = 38.13547087602444; Q = 0.5003270373238773;
    K = std::tan(M_PI * f0 / rate);
    a0 = 1.0 + K / Q + K * K;
    const double r_hp = radius(2.0 * (K * K - 1.0) / a0, (1.0 - K / Q + K * K) / a0);
    const double r = std::fmax(r_shelf, r_hp);
    return std::isfinite(r) ? r : 2.0;
}

static void mat4_mul(const long double *x, const long double *y, long double *o)
{
    long double t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            long double s = 0;
            for (int k = 0; k < 4; k++) s += x[i * 4 + k] * y[k * 4 + j];
            t[i * 4 + j] = s;
        }
    std::memcpy(o, t, sizeof t);
}

void kweight_transition_pow(const double a[5], uint64_t steps, double out[16])
{
    // s = (v1,v2,v3,v4); zero input: v0 = -a1 v1 - a2 v2 - a3 v3 - a4 v4; s' = (v0,v1,v2,v3): the companion matrix A.
    //
    // The table holds powers of  D A D  instead of A, D = [(-1)^j C(i,j)] the backward-difference transform of the state
    // (w = D s = (v1, v1 - v2, v1 - 2 v2 + v3, v1 - 3 v2 + 3 v3 - v4);  D is its own inverse).  The K-weighting poles sit
    // close to z = 1 (the high-pass pair at |z| = 0.995 at 48 kHz, 0.9988 at 192 kHz), so the DF-II state is a large,
    // slowly varying sequence (1e5 ... 1e8 times the input) and the entries of A^n are large with alternating signs: a
    // product A^n s in f64 cancels eight to sixteen digits (the kernel's chunk scan lost 3e-8 of the sub-block energies at
    // 48 kHz, 5e-6 at 96 kHz, 2e-3 at 192 kHz against the sequential recurrence).  In difference coordinates the same map
    // is a slowly growing integrator chain without cancellation, and the differences themselves are exact in floating
    // point (neighbouring states agree to within a factor of two).  D A D is formed and raised to the power in long double.
    static const long double D[16] = {1, 0, 0, 0, 1, -1, 0, 0, 1, -2, 1, 0, 1, -3, 3, -1};
    long double A[16] = {-(long double)a[1], -(long double)a[2], -(long double)a[3], -(long double)a[4],
                         1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    mat4_mul(A, D, A);
    mat4_mul(D, A, A);
    long double R[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    while (steps) {
        if (steps & 1) mat4_mul(A, R, R);
        mat4_mul(A, A, A);
        steps >>= 1;
    }
    for (int i = 0; i < 16; i++) out[i] = (double)R[i];
}

void true_peak_design(int factor, std::vector<std::vector<PolyTap>> &phases, int *delay_len)
{
    const int taps = 49;
    phases.assign(factor, {});
    *delay_len = (taps + factor - 1) / factor;
    for (int j = 0; j < taps; j++) {
        double m = (double)j - (double)(taps - 1) / 2.0;
        double c = 1.0;
        if (std::fabs(m) > 0.000001) c = std::sin(m * M_PI / factor) / (m * M_PI / factor);
        c *= 0.5 * (1.0 - std::cos(2.0 * M_PI * j / (taps - 1)));
        if (std::fabs(c) > 0.000001) phases[j % factor].push_back({j / factor, (float)c});
    }
}

int true_peak_factor_for_rate(uint32_t rate)
{
    return rate < 96000 ? 4 : (rate < 192000 ? 2 : 0);
}

void histogram_tables(double energies[kHistBins], double bounds[kHistBins + 1])
{
    bounds[0] = std::pow(10.0, (-70.0 + 0.691) / 10.0);
    for (int i = 0; i < kHistBins; i++)
        energies[i] = std::pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
    for (int i = 1; i <= kHistBins; i++)
        bounds[i] = std::pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
}

namespace {
struct HistTables {
    double e[kHistBins], b[kHistBins + 1];
    HistTables() { histogram_tables(e, b); }
};
const HistTables &ht() { static HistTables t; return t; }

size_t find_index(double energy)
{
    const double *b = ht().b;
    size_t lo = 0, hi = kHistBins;
    do {
        size_t mid = (lo + hi) / 2;
        if (energy >= b[mid]) lo = mid; else hi = mid;
    } while (hi - lo != 1);
    return lo;
}
}  // namespace

double gated_loudness(const uint64_t *hist)
{
    const double *e = ht().e, *b = ht().b;
    double rel = 0.0; uint64_t cnt = 0;
    for (int i = 0; i < kHistBins; i++) { rel += (double)hist[i] * e[i]; cnt += hist[i]; }
    if (!cnt) return -INFINITY;
    rel /= (double)cnt;
    rel *= std::pow(10.0, -10.0 / 10.0);
    size_t start;
    if (rel < b[0]) start = 0;
    else { start = find_index(rel); if (rel > e[start]) start++; }
    double g = 0.0; cnt = 0;
    for (size_t i = start; i < (size_t)kHistBins; i++) { g += (double)hist[i] * e[i]; cnt += hist[i]; }
    if (!cnt) return -INFINITY;
    return 10.0 * std::log10(g / (double)cnt) - 0.691;
}

double loudness_range(const uint64_t *h)
{
    const double *e = ht().e, *b = ht().b;
    uint64_t size = 0; double power = 0.0;
    for (int j = 0; j < kHistBins; j++) { size += h[j]; power += (double)h[j] * e[j]; }
    if (!size) return 0.0;
    power /= (double)size;
    double integ = std::pow(10.0, -20.0 / 10.0) * power;
    size_t index;
    if (integ < b[0]) index = 0;
    else { index = find_index(integ); if (integ > e[index]) index++; }
    size = 0;
    for (size_t j = index; j < (size_t)kHistBins; j++) size += h[j];
    if (!size) return 0.0;
    uint64_t plow = (uint64_t)((double)(size - 1) * 0.1 + 0.5);
    uint64_t phigh = (uint64_t)((double)(size - 1) * 0.95 + 0.5);
    size = 0; size_t j = index;
    while (size <= plow) size += h[j++];
    double l_en = e[j - 1];
    while (size <= phigh) size += h[j++];
    double h_en = e[j - 1];
    return (10.0 * std::log10(h_en) - 0.691) - (10.0 * std::log10(l_en) - 0.691);
}

void channel_weights(uint32_t channels, double *w)
{
    for (uint32_t i = 0; i < channels; i++) {
        double v = 0.0;
        if (channels == 4) { static const double q[4] = {1.0, 1.0, 1.41, 1.41}; v = q[i]; }
        else if (channels == 5) { static const double q[5] = {1.0, 1.0, 1.0, 1.41, 1.41}; v = q[i]; }
        else switch (i) { case 0: case 1: case 2: v = 1.0; break; case 3: v = 0.0; break;
                          case 4: case 5: v = 1.41; break; default: v = 0.0; }
        w[i] = v;
    }
}

}  // namespace sst
