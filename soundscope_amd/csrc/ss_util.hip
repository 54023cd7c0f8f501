// ss_util.hip: standalone decimation, mid/side, render reductions, PCM ingest, synthetic corpus — hand-written gfx950 (CDNA4, wave64) kernels of the soundscope analyzer hot path.
// Reference semantics: /root/reference/src/analyzer.rs (get_fft :55-105, get_waveform :107-137,
// add_samples/getters :139-164, calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as restated in DESIGN.md.
// Nothing here is translated from the reference: the reference has no GPU code.
#include "ss_kernels.h"

namespace ssk {
// ============================================================================
//  Waveform: min-max decimation, Analyzer::get_waveform (analyzer.rs:107-137).
//  Bin i covers [floor(i*spp), min(ceil((i+1)*spp), len)), spp = len / W in
//  f64 — the same f64 expressions as the reference, evaluated per bin.
//  16 lanes per bin; min/max are IEEE minNum/maxNum (NaN-ignoring, like
//  f32::min/max), seeded with NaN so an all-NaN bin stays NaN.
// ============================================================================
__global__ __launch_bounds__(256) void k_waveform(WaveParams p)
{
    const uint32_t lane16 = threadIdx.x & 15;
    const uint64_t gbin = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const uint64_t total_bins = (uint64_t)p.n_streams * p.window;
    if (gbin >= total_bins) return;
    const uint32_t stream = (uint32_t)(gbin / p.window);
    const uint32_t i = (uint32_t)(gbin - (uint64_t)stream * p.window);
    if (p.window_of) {                          // ragged batches: this stream's own length and bin count
        if (i >= p.window_of[stream]) return;
        p.n_samples = p.samples_of[stream];
        p.window = p.window_of[stream];
    }
    const double spp = (double)p.n_samples / (double)p.window;
    const double sd = (double)i * spp;
    const double ed = ceil((double)(i + 1) * spp);
    uint64_t start = (uint64_t)sd;
    uint64_t end = (ed >= 1.8446744073709552e19) ? ~0ull : (uint64_t)ed;
    if (end > p.n_samples) end = p.n_samples;
    float *o = p.out + (size_t)stream * p.out_stride + (size_t)i * 2;
    if (start >= p.n_samples) return;            // `break`: this and all later bins produce no point
    const float *x = p.pcm + (size_t)stream * p.stream_stride;
    float mn = __builtin_nanf(""), mx = __builtin_nanf("");
    if (p.mid_of_pairs) {
        const float2 *xp = reinterpret_cast<const float2 *>(x);
        for (uint64_t j = start + lane16; j < end; j += 16) {
            const float2 f = xp[j];
            const float v = (f.x + f.y) / 2.0f;
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    } else {
        for (uint64_t j = start + lane16; j < end; j += 16) {
            const float v = x[j];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
#pragma unroll
    for (int ofs = 8; ofs >= 1; ofs >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, ofs, 16));
        mx = fmaxf(mx, __shfl_xor(mx, ofs, 16));
    }
    if (lane16 == 0) { o[0] = mn; o[1] = mx; }
}

hipError_t launch_waveform(const WaveParams &p, hipStream_t s)
{
    const uint64_t total_bins = (uint64_t)p.n_streams * p.window;
    if (total_bins == 0) return hipSuccess;
    const uint64_t blocks = (total_bins * 16 + 255) / 256;
    hipLaunchKernelGGL(k_waveform, dim3((uint32_t)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ============================================================================
//  Utilities
// ============================================================================
// get_mid_and_side_samples, audio_player.rs:400-419
__global__ void k_mid_side(const float2 *in, size_t frames, float *mid, float *side)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < frames) {
        const float2 v = in[i];
        mid[i] = (v.x + v.y) / 2.0f;
        side[i] = (v.x - v.y) / 2.0f;
    }
}

hipError_t launch_mid_side(const float *interleaved, size_t frames, float *mid, float *side, hipStream_t s)
{
    if (!frames) return hipSuccess;
    hipLaunchKernelGGL(k_mid_side, dim3((uint32_t)((frames + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(interleaved), frames, mid, side);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_nonfinite_pairs(const float2 *in, size_t pairs, uint32_t *count, NonFinitePair *list, uint32_t cap)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < pairs; i += stride) {
        const float2 v = in[i];
        const float m = (v.x + v.y) * 0.5f, sd = (v.x - v.y) * 0.5f;
        uint32_t c = 0;
        if (m != m) c |= 1u; else if (fabsf(m) == INFINITY) c |= 2u;
        if (sd != sd) c |= 4u; else if (fabsf(sd) == INFINITY) c |= 8u;
        if (c) {
            const uint32_t slot = atomicAdd(count, 1u);
            if (slot < cap) list[slot] = NonFinitePair{(unsigned long long)i, c, 0u};
        }
    }
}

hipError_t launch_nonfinite_pairs(const float *interleaved, size_t pairs, uint32_t *count, NonFinitePair *list, uint32_t cap, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), s);
    if (e != hipSuccess || !pairs) return e;
    const size_t want = (pairs + 1023) / 1024;                      // four pairs per thread or more
    hipLaunchKernelGGL(k_nonfinite_pairs, dim3((uint32_t)(want < 4096 ? want : 4096)), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(interleaved), pairs, count, list, cap);
    return hipGetLastError();
}

// ============================================================================
//  Render-side reductions (tui.rs:49-51, :801-821, :664-681)
//  Spectrum: y + gain, clamped to the chart's [-100, 0] dB, reduced to chart columns on the log-x axis
//  (column c owns the contiguous bin range [col_start[c], col_start[c+1]); value = maximum; no bin -> NaN).
//  One wave per spectrum row: the row is staged in wave-private LDS with coalesced loads, then lane c
//  walks its bins.  gain: fixed, or the reference's per-file rule FFT_TARGET_LUFS - integrated (f32).
// ============================================================================
__global__ __launch_bounds__(256) void k_render_spectrum(const float *rows, uint32_t bin_stride, uint32_t n_bins,
                                                         uint64_t n_rows, uint32_t rows_per_stream,
                                                         const uint32_t *col_start, uint32_t cols,
                                                         const double *integrated, float gain_db, float *out)
{
    extern __shared__ float rs_lds[];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + wv;
    if (row >= n_rows) return;
    float *mine = rs_lds + (size_t)wv * bin_stride;
    const float4 *src = reinterpret_cast<const float4 *>(rows + row * bin_stride);
    for (uint32_t i = lane; i < bin_stride / 4; i += 64u) reinterpret_cast<float4 *>(mine)[i] = src[i];
    __builtin_amdgcn_wave_barrier();
    float gain = gain_db;
    if (integrated) gain = -13.0f - (float)integrated[row / rows_per_stream];      // tui.rs:1234
    float *o = out + row * cols;
    for (uint32_t c = lane; c < cols; c += 64u) {
        const uint32_t b0 = col_start[c], b1 = col_start[c + 1];
        float m = __builtin_nanf("");
        for (uint32_t b = b0; b < b1 && b < n_bins; b++) {
            float v = mine[b] + gain;
            v = fminf(fmaxf(v, -100.0f), 0.0f);
            m = fmaxf(m, v);                 // maxNum: the NaN seed disappears with the first bin
        }
        o[c] = m;
    }
}

hipError_t launch_render_spectrum(const float *rows, uint32_t bin_stride, uint32_t n_bins, uint64_t n_rows,
                                  uint32_t rows_per_stream, const uint32_t *col_start, uint32_t cols,
                                  const double *integrated, float gain_db, float *out, hipStream_t s)
{
    if (!n_rows || !cols) return hipSuccess;
    const size_t lds = (size_t)4 * bin_stride * sizeof(float);
    hipLaunchKernelGGL(k_render_spectrum, dim3((uint32_t)((n_rows + 3) / 4)), dim3(256), lds, s, rows, bin_stride,
                       n_bins, n_rows, rows_per_stream, col_start, cols, integrated, gain_db, out);
    return hipGetLastError();
}

// Waveform: the (min, max) decimation bins inside the view [x_min, x_max) reduced to `cols` columns:
// column c owns bins i with floor((i - x_min) * cols / (x_max - x_min)) == c; min of mins, max of maxes
// (f32::min / f32::max semantics like get_waveform itself).
__global__ __launch_bounds__(256) void k_render_waveform(const float *wave, uint64_t wave_stride, uint32_t n_points,
                                                         uint32_t n_streams, uint32_t x_min, uint32_t x_max,
                                                         uint32_t cols, float *out)
{
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (uint64_t)n_streams * cols) return;
    const uint32_t stream = (uint32_t)(idx / cols), c = (uint32_t)(idx % cols);
    const uint64_t span = (uint64_t)x_max - x_min;
    // first bin of column c: smallest i with (i - x_min) * cols >= c * span
    const uint32_t i0 = x_min + (uint32_t)(((uint64_t)c * span + cols - 1) / cols);
    const uint32_t i1 = x_min + (uint32_t)(((uint64_t)(c + 1) * span + cols - 1) / cols);
    const float2 *w = reinterpret_cast<const float2 *>(wave + (uint64_t)stream * wave_stride);
    float lo = __builtin_nanf(""), hi = __builtin_nanf("");
    for (uint32_t i = i0; i < i1 && i < n_points; i++) {
        const float2 v = w[i];
        lo = fminf(lo, v.x);
        hi = fmaxf(hi, v.y);
    }
    reinterpret_cast<float2 *>(out)[idx] = make_float2(lo, hi);
}

hipError_t launch_render_waveform(const float *wave, uint64_t wave_stride, uint32_t n_points, uint32_t n_streams,
                                  uint32_t x_min, uint32_t x_max, uint32_t cols, float *out, hipStream_t s)
{
    if (!n_streams || !cols || x_max <= x_min) return hipSuccess;
    const uint64_t n = (uint64_t)n_streams * cols;
    hipLaunchKernelGGL(k_render_waveform, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, wave, wave_stride,
                       n_points, n_streams, x_min, x_max, cols, out);
    return hipGetLastError();
}

// PCM ingest: symphonia's sample conversions to f32 (audio_player.rs:169-267 decodes through
// SampleBuffer::<f32>::copy_interleaved_ref).  Every scale is an exact power of two.
__global__ __launch_bounds__(256) void k_pcm_to_f32(const unsigned char *src, size_t n, int format, float *dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v;
        switch (format) {
            case 1: v = (float)src[i] / 128.0f - 1.0f; break;
            case 2: v = (float)reinterpret_cast<const short *>(src)[i] / 32768.0f; break;
            case 3: {
                const unsigned char *q = src + 3 * i;
                int s = (int)q[0] | ((int)q[1] << 8) | ((int)(signed char)q[2] << 16);
                v = (float)s / 8388608.0f;
                break;
            }
            case 4: v = (float)((double)reinterpret_cast<const int *>(src)[i] / 2147483648.0); break;
            case 5: v = reinterpret_cast<const float *>(src)[i]; break;
            default: v = (float)reinterpret_cast<const double *>(src)[i]; break;
        }
        dst[i] = v;
    }
}

hipError_t launch_pcm_to_f32(const void *src, size_t n_samples, int format, float *dst, hipStream_t s)
{
    if (!n_samples) return hipSuccess;
    size_t blocks = (n_samples + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pcm_to_f32, dim3((uint32_t)blocks), dim3(256), 0, s,
                       static_cast<const unsigned char *>(src), n_samples, format, dst);
    return hipGetLastError();
}

// Synthetic corpus: per stream two sines + uniform noise, level
// spread over ~20 dB, 5 % of streams carry a 3 s near-silent segment.
__device__ __forceinline__ uint32_t mix32(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}
__device__ __forceinline__ float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void k_synth(float *pcm, uint32_t n_streams, uint64_t frames, uint32_t C,
                                               uint32_t rate, uint64_t seed, uint32_t first_id)
{
    const uint64_t per_stream = frames * C;
    const uint64_t total = per_stream * n_streams;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (uint64_t)gridDim.x * 256) {
        const uint32_t s = (uint32_t)(g / per_stream);
        const uint64_t r = g - (uint64_t)s * per_stream;
        const uint64_t f = r / C; const uint32_t c = (uint32_t)(r - f * C);
        const uint64_t sid = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(first_id + s) * 0xD1B54A32D192ED03ull;
        const float uf = u01(mix32(sid + 11 + c * 7919ull));
        const float freq = 50.0f * __expf(uf * 5.480639f);                 // log-uniform 50..12000 Hz
        const float phase = u01(mix32(sid + 23 + c));
        const float level = __expf(-2.3025851f * u01(mix32(sid + 5)));     // amplitude 1 .. 0.1
        const bool has_gap = (mix32(sid + 99) % 20u) == 0u;
        const uint64_t gap0 = (uint64_t)(u01(mix32(sid + 101)) * 0.6f * (float)frames);
        float gain = level;
        if (has_gap && f >= gap0 && f < gap0 + 3ull * rate) gain *= 1e-4f;
        const double cyc = (double)freq * (double)f / (double)rate + (double)phase;
        const float ph = (float)(cyc - floor(cyc));
        const float noise = 2.0f * u01(mix32(sid ^ (r * 0x2545F4914F6CDD1Dull + 77))) - 1.0f;
        pcm[g] = gain * (0.25f * __sinf(6.2831853f * ph) + 0.05f * noise);
    }
}

hipError_t launch_synth(float *pcm, uint32_t n_streams, uint64_t frames, uint32_t channels,
                        uint32_t rate, uint64_t seed, uint32_t first_id, hipStream_t s)
{
    if (!n_streams || !frames) return hipSuccess;
    hipLaunchKernelGGL(k_synth, dim3(4096), dim3(256), 0, s, pcm, n_streams, frames, channels, rate, seed, first_id);
    return hipGetLastError();
}

// ---- verification utility: order-independent 64-bit checksum of `n_items` equally sized word ranges ---------------------
// out[item * out_stride] += sum over the item's 32-bit words of mix(word, index).  A sum, so any partition of the range
// over threads gives the same value: two passes over the same data agree bit for bit iff (up to 2^-64) the data do.
// Lets a stress test compare EVERY spectrum row of a 6.5 GB batch between launch modes without downloading it.
__global__ __launch_bounds__(256) void k_checksum(const uint32_t *base, uint64_t words, uint64_t stride_words, uint64_t *out,
                                                  uint32_t out_stride)
{
    const uint32_t item = blockIdx.y;
    const uint32_t *x = base + (size_t)item * stride_words;
    uint64_t acc = 0;
    auto mix = [](uint32_t w, uint64_t idx) -> uint64_t {
        uint32_t h = w + 0x9E3779B9u * (uint32_t)idx;
        h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        return (uint64_t)h * (uint64_t)(((uint32_t)(idx >> 3) & 0xFFFFu) | 1u);
    };
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (uint64_t)gridDim.x * 256;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        const uint64_t n4 = words >> 2;
        for (uint64_t i = tid; i < n4; i += nthreads) {
            const uint4 v = reinterpret_cast<const uint4 *>(x)[i];
            acc += mix(v.x, 4 * i) + mix(v.y, 4 * i + 1) + mix(v.z, 4 * i + 2) + mix(v.w, 4 * i + 3);
        }
        for (uint64_t i = (n4 << 2) + tid; i < words; i += nthreads) acc += mix(x[i], i);
    } else {
        for (uint64_t i = tid; i < words; i += nthreads) acc += mix(x[i], i);
    }
#pragma unroll
    for (int ofs = 32; ofs >= 1; ofs >>= 1) acc += __shfl_xor(acc, ofs, 64);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(reinterpret_cast<unsigned long long *>(out + (size_t)item * out_stride), (unsigned long long)acc);
}

hipError_t launch_checksum(const void *base, uint64_t words, uint64_t stride_words, uint32_t n_items, uint64_t *out,
                           uint32_t out_stride, hipStream_t s)
{
    if (!n_items || !words) return hipSuccess;
    uint64_t per_item = (words / 4 + 255) / 256;                  // one uint4 per thread and trip at most ...
    const uint64_t want = (8192 + n_items - 1) / n_items;          // ... but no more workgroups than fill the chip a few times
    if (per_item > want) per_item = want;
    if (per_item < 1) per_item = 1;
    // the item index rides in gridDim.y (at most 65535): larger batches go in slices
    for (uint32_t i0 = 0; i0 < n_items; i0 += 65535u) {
        const uint32_t cnt = n_items - i0 < 65535u ? n_items - i0 : 65535u;
        hipLaunchKernelGGL(k_checksum, dim3((uint32_t)per_item, cnt), dim3(256), 0, s,
                           static_cast<const uint32_t *>(base) + (size_t)i0 * stride_words, words, stride_words,
                           out + (size_t)i0 * out_stride, out_stride);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- one launch that clears up to four buffers (a pass's per-stream meter state, histograms, corpus histograms and block
// counts: four hipMemsetAsync = four fill kernels before; a one-stream pass is launch-bound)
struct Zero4 { uint32_t *p[4]; uint64_t words[4]; };
__global__ __launch_bounds__(256) void k_zero4(Zero4 z)
{
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (uint64_t)gridDim.x * 256;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        uint32_t *p = z.p[b];
        const uint64_t n = z.words[b];
        if (!p || !n) continue;
        if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
            const uint64_t n4 = n >> 2;
            for (uint64_t i = tid; i < n4; i += nthreads) reinterpret_cast<uint4 *>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
            for (uint64_t i = (n4 << 2) + tid; i < n; i += nthreads) p[i] = 0u;
        } else {
            for (uint64_t i = tid; i < n; i += nthreads) p[i] = 0u;
        }
    }
}

hipError_t launch_zero4(void *const ptrs[4], const size_t bytes[4], hipStream_t s)
{
    Zero4 z{};
    uint64_t most = 0;
    for (int b = 0; b < 4; b++) {
        if (bytes[b] & 3u) return hipErrorInvalidValue;
        z.p[b] = static_cast<uint32_t *>(ptrs[b]); z.words[b] = bytes[b] >> 2;
        if (z.words[b] > most) most = z.words[b];
    }
    if (!most) return hipSuccess;
    uint64_t blocks = (most / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_zero4, dim3((uint32_t)blocks), dim3(256), 0, s, z);
    return hipGetLastError();
}

// ---- measurement utility: the spectrum kernel's HBM traffic with no arithmetic -----------------------------------------
// Same grid, same workgroup size, same LDS footprint (so the same three workgroups per CU), same addresses: every
// workgroup walks its run of windows, loads the four new 256-frame slots of each window (8-byte loads) and stores the two
// output rows as 16-byte stores.  What this takes is the floor the memory system sets for k_fft4096_ms1's access pattern
// on this chip — a copy-shaped ceiling next to which the kernel's own time can be read (bench.py: roofline.io_floor).
__global__ __launch_bounds__(256, 3) void k_fft4096_traffic(FftBatchParams p)
{
    __shared__ float pad[9728];                                           // 38 912 B: the spectrum kernel's footprint
    const int t = threadIdx.x;
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    const uint32_t stream = blockIdx.x / groups;
    const uint32_t grp = blockIdx.x - stream * groups;
    const uint32_t w_begin = grp * p.windows_per_block;
    uint32_t w_end = w_begin + p.windows_per_block;
    if (w_begin >= p.n_windows) return;
    if (w_end > p.n_windows) w_end = p.n_windows;
    const float2 *src = reinterpret_cast<const float2 *>(p.pcm) + (size_t)stream * p.frames_per_stream + p.first_start + (size_t)w_begin * p.hop;
    float *outp = p.out + ((size_t)stream * p.n_windows + w_begin) * 2 * p.bin_stride;
    const uint32_t ngroups = (p.n_bins + 3) >> 2;
    float acc = 0.0f;
    pad[t] = 0.0f;
#pragma unroll
    for (int j = 0; j < 12; j++) { const float2 v = src[t + 256 * j]; acc += v.x + v.y; }       // the run's first window: all 16 slots
    for (uint32_t w = w_begin; w < w_end; ++w) {
#pragma unroll
        for (int q = 0; q < 4; q++) { const float2 v = src[(size_t)(w - w_begin) * p.hop + t + 256 * (12 + q)]; acc += v.x + v.y; }
        float *o = outp + (size_t)(w - w_begin) * 2 * p.bin_stride;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t g = (uint32_t)t + 256u * i;
            if (g < ngroups) {
                const float4 v = make_float4(acc, acc, acc, acc);
                reinterpret_cast<float4 *>(o)[g] = v;
                reinterpret_cast<float4 *>(o + p.bin_stride)[g] = v;
            }
        }
    }
    if (acc == 1.2345e33f) pad[t] = acc;
}

hipError_t launch_fft4096_traffic(const FftBatchParams &p, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0) return hipSuccess;
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    hipLaunchKernelGGL(k_fft4096_traffic, dim3(groups * p.n_streams), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace ssk
