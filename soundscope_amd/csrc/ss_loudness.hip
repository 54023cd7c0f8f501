// ss_loudness.hip: gating blocks, histograms, integrated loudness, LRA, ring energies — hand-written gfx950 (CDNA4, wave64) kernels of the soundscope analyzer hot path.
// Reference semantics: /root/reference/src/analyzer.rs (get_fft :55-105, get_waveform :107-137,
// add_samples/getters :139-164, calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as restated in DESIGN.md.
// Nothing here is translated from the reference: the reference has no GPU code.
#include "ss_kernels.h"

namespace ssk {
// ============================================================================
//  Gating blocks, histograms, integrated loudness and LRA
//  (ebur128 calc_gating_block / loudness_global / loudness_range, histogram mode)
// ============================================================================
// largest i with bounds[i] <= energy (the caller has checked energy >= bounds[0]) — what ebur128's binary search
// over the bin boundaries returns.  bounds[i] is the energy of -70 + i/10 LUFS, so the index is guessed in closed
// form and then corrected against the table itself (at most a step or two): two dependent loads instead of ten.
__device__ __forceinline__ uint32_t hist_index(const double *__restrict__ bounds, double energy)
{
    const double g = (10.0 * log10(energy) - 0.691 + 70.0) * 10.0;
    int i = g > 0.0 ? (g < (double)(kHistBins - 1) ? (int)g : kHistBins - 1) : 0;
    while (i > 0 && energy < bounds[i]) i--;
    while (i < kHistBins - 1 && energy >= bounds[i + 1]) i++;
    return (uint32_t)i;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// First sub-block of a stream in which a WEIGHTED channel met a non-finite sample (TdState::bad_key), 0xFFFFFFFF if none: one whole
// wave asks, every lane gets the answer.  The crate's filter state is NaN from that sample on, for good: every window that ends
// BEHIND this sub-block has a NaN energy (`sum >= boundary` fails: no histogram entry), whichever segment of a launch computed
// its sub-blocks from whichever starting state; the window that ends WITH it holds what the recurrence itself produced (NaN, or
// +Inf if the sample was an infinity in the sub-block's very last frame).  Channels the crate does not filter (weight 0) never count.
__device__ __forceinline__ uint32_t first_bad_subblock(const TdState *st, uint32_t C, const double *__restrict__ weights, uint32_t lane)
{
    uint32_t key = 0u;
    if (st) for (uint32_t c = lane; c < C; c += 64u) { const uint32_t k = st->bad_key[c]; if (weights[c] != 0.0 && k > key) key = k; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)key, o, 64); key = t > key ? t : key; }
    return ~key;
}

// one wave evaluates gate and LRA on an LDS histogram pair (block, short-term); inlined: `en` / `bd` are LDS copies of the tables
// in the latency-bound callers (every dependent table read is then an LDS access, not a round trip to L2)
__device__ __forceinline__ void eval_hist(const unsigned long long *hb, const unsigned long long *hs,
                          const double *__restrict__ en, const double *__restrict__ bd,
                          double *out_i, double *out_lra)
{
    const int lane = threadIdx.x & 63;
    // ---- integrated: relative gate at -10 LU of the mean of all blocks
    double sum = 0.0; unsigned long long cnt = 0;
    for (int i = lane; i < kHistBins; i += 64) { sum += (double)hb[i] * en[i]; cnt += hb[i]; }
    sum = wave_sum(sum); cnt = wave_sum_u64(cnt);
    double integrated;
    if (cnt == 0) integrated = -INFINITY;
    else {
        const double rel = (sum / (double)cnt) * 0.1;
        uint32_t start;
        if (rel < bd[0]) start = 0;
        else { start = hist_index(bd, rel); if (rel > en[start]) start++; }
        double g = 0.0; unsigned long long c2 = 0;
        for (int i = lane; i < kHistBins; i += 64) if ((uint32_t)i >= start) { g += (double)hb[i] * en[i]; c2 += hb[i]; }
        g = wave_sum(g); c2 = wave_sum_u64(c2);
        integrated = c2 ? 10.0 * log10(g / (double)c2) - 0.691 : -INFINITY;
    }
    // ---- LRA (EBU Tech 3342) on the short-term histogram
    double power = 0.0; unsigned long long size = 0;
    for (int i = lane; i < kHistBins; i += 64) { power += (double)hs[i] * en[i]; size += hs[i]; }
    power = wave_sum(power); size = wave_sum_u64(size);
    double lra = 0.0;
    if (size != 0) {
        const double integ = 0.01 * (power / (double)size);
        uint32_t index;
        if (integ < bd[0]) index = 0;
        else { index = hist_index(bd, integ); if (integ > en[index]) index++; }
        // Percentile bins without a serial walk over the histogram: lane l owns bins [16 l, 16 l + 16), an inclusive scan
        // over the lanes' gated counts tells which lane holds an entry of a given rank, and that lane walks its own 16
        // bins.  (ebur128 walks all bins from the gate: `while acc <= rank { acc += hist[j]; j += 1 }`, entry = bin j - 1,
        // i.e. the first bin at which the cumulative gated count exceeds the rank.)
        const int b0 = lane * 16;
        unsigned long long own = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int i = b0 + ((q + lane) & 15);                    // rotated start: spreads the lanes over the LDS banks
            if (i < kHistBins && (uint32_t)i >= index) own += hs[i];
        }
        unsigned long long incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const unsigned long long above = __shfl(incl, 63, 64);
        if (above != 0) {
            const unsigned long long plow = (unsigned long long)((double)(above - 1) * 0.1 + 0.5);
            const unsigned long long phigh = (unsigned long long)((double)(above - 1) * 0.95 + 0.5);
            const unsigned long long excl = incl - own;
            auto energy_of_rank = [&](unsigned long long r) -> double {
                const bool mine = excl <= r && r < incl;             // exactly one lane (ranks are < above)
                double e = 0.0;
                if (mine) {
                    unsigned long long acc = excl;
                    int j = b0 > (int)index ? b0 : (int)index;
                    for (;;) { acc += hs[j]; if (acc > r) break; j++; }
                    e = en[j];
                }
                const int src = __ffsll((long long)__ballot(mine)) - 1;
                return __shfl(e, src, 64);
            };
            const double l_en = energy_of_rank(plow), h_en = energy_of_rank(phigh);
            lra = (10.0 * log10(h_en) - 0.691) - (10.0 * log10(l_en) - 0.691);
        }
    }
    if (lane == 0) { if (out_i) *out_i = integrated; if (out_lra) *out_lra = lra; }
}

// slot of sub-block (j - q) in a ring of `cap` slots, given jm = j % cap and q <= 29 (no division per term)
__device__ __forceinline__ uint32_t ring_back(uint32_t jm, uint32_t q, uint32_t cap)
{
    uint32_t s = jm + cap * 30u - q;          // cap >= 1: the bias keeps it positive for q <= 29 < 30 cap
    return s % cap;
}

// Weighted energy of the N sub-blocks ending with sub-block j (slot jm = j % cap):  sum_c w_c (P[j-N+1][c] + ... + P[j][c]),
// each channel added oldest first.  All loads of a batch are issued before the first addition — written as a loop of
// `cs += P[...]` the thirty terms of a short-term block were thirty dependent round trips to memory (12 us of a tick, and
// the whole of this kernel's time on a long stream).
template <int N, bool DIRECT, int KB = (N < 10 ? N : 10)>
__device__ __forceinline__ double window_energy(const double *__restrict__ P, uint32_t jm, uint32_t cap, uint32_t C,
                                                const double *__restrict__ weights)
{
    constexpr int kBatch = KB;
    static_assert(N % kBatch == 0, "whole batches");
    uint32_t s0 = DIRECT ? jm - (uint32_t)(N - 1) : ring_back(jm, (uint32_t)(N - 1), cap);      // slot of the oldest term
    double sum = 0.0;
    for (uint32_t c = 0; c < C; c++) {
        const double w = weights[c];
        if (w == 0.0) continue;
        double cs = 0.0;
        uint32_t sl = s0;
#pragma unroll
        for (int b = 0; b < N; b += kBatch) {
            double v[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; q++) {
                v[q] = P[(size_t)sl * C + c];
                sl = DIRECT ? sl + 1u : (sl + 1u == cap ? 0u : sl + 1u);
            }
#pragma unroll
            for (int q = 0; q < kBatch; q++) cs += v[q];
        }
        sum += w * cs;
    }
    return sum;
}

// The same sums for the latency-bound launches (a handful of streams: the kernel's time is its chain of dependent round trips to
// memory, not its work): channels two at a time, every term of both and their weights requested before anything is used — ONE
// round trip per pair of channels where the form above takes one for the weight and N / 10 per channel behind it.  Same additions
// in the same order (a channel whose weight is zero is loaded and dropped, as the `continue` above drops it unread).
template <int N, bool DIRECT>
__device__ __forceinline__ double window_energy_eager(const double *__restrict__ P, uint32_t jm, uint32_t cap, uint32_t C,
                                                      const double *__restrict__ weights)
{
    const uint32_t s0 = DIRECT ? jm - (uint32_t)(N - 1) : ring_back(jm, (uint32_t)(N - 1), cap);
    double sum = 0.0;
    for (uint32_t c = 0; c < C; c += 2) {
        const bool two = c + 1u < C;
        const uint32_t c1 = two ? c + 1u : c;
        const double w0 = weights[c], w1 = weights[c1];
        double v0[N], v1[N];
        uint32_t sl = s0;
#pragma unroll
        for (int q = 0; q < N; q++) {
            v0[q] = P[(size_t)sl * C + c];
            v1[q] = P[(size_t)sl * C + c1];
            sl = DIRECT ? sl + 1u : (sl + 1u == cap ? 0u : sl + 1u);
        }
        double cs0 = 0.0, cs1 = 0.0;
#pragma unroll
        for (int q = 0; q < N; q++) { cs0 += v0[q]; cs1 += v1[q]; }
        if (w0 != 0.0) sum += w0 * cs0;
        if (two && w1 != 0.0) sum += w1 * cs1;
    }
    return sum;
}

// One workgroup per stream; the gating blocks of a stream are independent (histogram increments are LDS atomics), so a
// long stream (config 2's 600 s: 6000 sub-blocks) is spread over up to 1024 threads — a thread's iteration is a chain of
// dependent loads, so the kernel's time is its iteration count; wave 0 then evaluates gate and LRA.
// SMALL (a handful of streams: config 2, a file open, calculate_integrated_lufs): the two tables the gate reads — bin energies and
// bin boundaries, 16 KB — come into LDS with the histograms, and the sub-block sums are requested eagerly: the kernel's chain of
// dependent trips to L2 / HBM is four long instead of eighteen (config 2: 22.5 -> 20 us, config 5: 31 -> 25 us, profiles/r05_small_batch_finalize.txt).  The big grids keep
// the lean form: 16 KB more LDS traffic per workgroup buys nothing where a thousand workgroups hide each other's latency.
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? 256 : 1024) void k_finalize(FinalizeParams p)
{
    __shared__ unsigned long long hb[kHistBins];
    __shared__ unsigned long long hs[kHistBins];
    __shared__ unsigned int counts[2];
    __shared__ uint32_t bad_from_s;
    __shared__ double tab[SMALL ? 2 * kHistBins + 1 : 1];          // SMALL: [energies 1000][bounds 1001]
    const uint32_t stream = blockIdx.x;
    const int lane = threadIdx.x, nthr = (int)blockDim.x;
    unsigned long long *gh = reinterpret_cast<unsigned long long *>(p.hist) + (size_t)stream * 2 * kHistBins;
    unsigned long long *corpus = reinterpret_cast<unsigned long long *>(p.corpus_hist);
    for (int i = lane; i < kHistBins; i += nthr) {
        hb[i] = gh[i]; hs[i] = gh[kHistBins + i];
        if (SMALL) { tab[i] = p.hist_energies[i]; tab[kHistBins + i] = p.hist_bounds[i]; }
    }
    if (SMALL && lane == 0) tab[2 * kHistBins] = p.hist_bounds[kHistBins];
    if (lane < 2) counts[lane] = 0;
    if (lane < 64) {
        const uint32_t bf = first_bad_subblock(p.state ? p.state + stream : nullptr, p.channels, p.weights, (uint32_t)lane);
        if (lane == 0) bad_from_s = bf;
    }
    const double *en = SMALL ? tab : p.hist_energies;
    const double *bd = SMALL ? tab + kHistBins : p.hist_bounds;
    const double bd0 = p.hist_bounds[0];                                                  // (read before the barrier: in flight with the rest)
    // gating block ending with sub-block j: j-3..j ; short-term block: j-29..j when (j-29) % 10 == 0
    const uint64_t sub_end = p.sub_end_of ? p.sub_end_of[stream] : p.sub_end;          // ragged batches
    __syncthreads();

    const uint32_t C = p.channels;
    const double S = (double)p.k->s100;
    const double *P = p.subblocks + (size_t)stream * p.sub_stride;
    const uint32_t cap = p.sub_cap;
    const bool direct = p.sub_end <= (uint64_t)cap && p.sub_begin == 0;               // batches: slot == sub-block index
    const uint64_t bad_from = bad_from_s;
    uint32_t nb = 0, ns = 0;
    // gating blocks: one per sub-block j >= 3
    for (uint64_t j = (p.sub_begin > 3 ? p.sub_begin : 3) + lane; j < sub_end; j += nthr) {
        const uint32_t jm = direct ? (uint32_t)j : (uint32_t)(j % cap);
        double sum;
        if (SMALL) sum = direct ? window_energy_eager<4, true>(P, jm, cap, C, p.weights) : window_energy_eager<4, false>(P, jm, cap, C, p.weights);
        else sum = direct ? window_energy<4, true>(P, jm, cap, C, p.weights) : window_energy<4, false>(P, jm, cap, C, p.weights);
        sum /= 4.0 * S;
        nb++;
        if (j > bad_from) sum = __builtin_nan("");
        if (sum >= bd0) atomicAdd(&hb[hist_index(bd, sum)], 1ull);
    }
    // short-term blocks: j = 29 + 10 m.  Dealt densely (thread = m), not as every tenth lane of the loop above
    if (!p.k->st_off) {
        const uint64_t m_begin = p.sub_begin > 29 ? (p.sub_begin - 29 + 9) / 10 : 0;      // first m with 29 + 10 m >= sub_begin
        for (uint64_t m = m_begin + lane;; m += nthr) {
            const uint64_t j = 29 + 10 * m;
            if (j >= sub_end) break;
            const uint32_t jm = direct ? (uint32_t)j : (uint32_t)(j % cap);
            double sum;
            if (SMALL) sum = direct ? window_energy_eager<30, true>(P, jm, cap, C, p.weights) : window_energy_eager<30, false>(P, jm, cap, C, p.weights);
            else sum = direct ? window_energy<30, true>(P, jm, cap, C, p.weights) : window_energy<30, false>(P, jm, cap, C, p.weights);
            sum /= 30.0 * S;
            ns++;
            if (j > bad_from) sum = __builtin_nan("");
            if (sum >= bd0) atomicAdd(&hs[hist_index(bd, sum)], 1ull);
        }
    }
    if (nb) atomicAdd(&counts[0], nb);
    if (ns) atomicAdd(&counts[1], ns);
    __syncthreads();
    // corpus contribution = what this call added
    for (int i = lane; i < kHistBins; i += nthr) {
        const unsigned long long db = hb[i] - gh[i], ds = hs[i] - gh[kHistBins + i];
        if (corpus) {
            if (db) atomicAdd(&corpus[i], db);
            if (ds) atomicAdd(&corpus[kHistBins + i], ds);
        }
        gh[i] = hb[i];
        gh[kHistBins + i] = hs[i];
    }
    if (p.out_counts && lane < 2) p.out_counts[stream * 2 + lane] += counts[lane];
    if (lane < 64)          // (wave 0; the histograms in LDS are complete: the barrier above)
        eval_hist(hb, hs, en, bd,
                  p.out_integrated ? &p.out_integrated[stream] : nullptr,
                  p.out_lra ? &p.out_lra[stream] : nullptr);
}

// Streaming form (one handle, a few new sub-blocks per call, no per-call read-out): the same gating rules
// with the histogram updated in place by global atomics instead of a 16 KB round trip through LDS.
__global__ __launch_bounds__(64) void k_finalize_stream(FinalizeParams p)
{
    // one wave, and every step waits for the one before it: the tables the gate reads (bin energies, bin boundaries) are staged in
    // LDS first — ONE round trip to L2 instead of one per dependent table read (two per histogram index, eight in eval_hist)
    __shared__ double tab[2 * kHistBins + 1];
    const int lane = threadIdx.x;
    for (int i = lane; i < kHistBins; i += 64) { tab[i] = p.hist_energies[i]; tab[kHistBins + i] = p.hist_bounds[i]; }
    if (lane == 0) tab[2 * kHistBins] = p.hist_bounds[kHistBins];
    const double *en = tab, *bd = tab + kHistBins;
    const double bd0 = p.hist_bounds[0];
    const uint64_t bad_from = first_bad_subblock(p.state, p.channels, p.weights, (uint32_t)lane);
    __syncthreads();
    unsigned long long *gh = reinterpret_cast<unsigned long long *>(p.hist);
    const uint32_t C = p.channels;
    const double S = (double)p.k->s100;
    const double *P = p.subblocks;
    uint32_t nb = 0, ns = 0;
    const uint32_t cap = p.sub_cap;
    for (uint64_t j = (p.sub_begin > 3 ? p.sub_begin : 3) + lane; j < p.sub_end; j += 64) {
        double sum = window_energy_eager<4, false>(P, (uint32_t)(j % cap), cap, C, p.weights) / (4.0 * S);
        if (j > bad_from) sum = __builtin_nan("");
        nb++;
        if (sum >= bd0) atomicAdd(&gh[hist_index(bd, sum)], 1ull);
    }
    if (!p.k->st_off) {
        const uint64_t m_begin = p.sub_begin > 29 ? (p.sub_begin - 29 + 9) / 10 : 0;
        for (uint64_t m = m_begin + lane;; m += 64) {
            const uint64_t j = 29 + 10 * m;
            if (j >= p.sub_end) break;
            double sum = window_energy_eager<30, false>(P, (uint32_t)(j % cap), cap, C, p.weights) / (30.0 * S);
            if (j > bad_from) sum = __builtin_nan("");
            ns++;
            if (sum >= bd0) atomicAdd(&gh[kHistBins + hist_index(bd, sum)], 1ull);
        }
    }
    if (p.out_counts) {
        if (nb) atomicAdd(&p.out_counts[0], nb);
        if (ns) atomicAdd(&p.out_counts[1], ns);
    }
    // the handle's readings behind the update (what the reference's render loop asks for on its next frame): the same wave
    // evaluates the histograms it has just touched — no launch of its own inside a tick
    if (p.readings_out) {
        __shared__ unsigned long long hb[kHistBins];
        __shared__ unsigned long long hs[kHistBins];
        __threadfence();                                    // this wave's atomics have landed
        for (int i = lane; i < kHistBins; i += 64) {
            hb[i] = __hip_atomic_load(&gh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hs[i] = __hip_atomic_load(&gh[kHistBins + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p.readings_peaks_dst) {
            p.readings_peaks_dst[lane] = p.readings_peaks_src[lane];
            p.readings_peaks_dst[64 + lane] = p.readings_peaks_src[64 + lane];
        }
        __syncthreads();
        eval_hist(hb, hs, en, bd, &p.readings_out[0], &p.readings_out[1]);
        if (p.readings_flag) {
            __threadfence_system();
            __syncthreads();
            if (lane == 0) __hip_atomic_store(p.readings_flag, p.readings_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

#ifndef SS_FINALIZE_SMALL_MAX
#define SS_FINALIZE_SMALL_MAX 64u      // streams up to which a launch counts as latency-bound (the form with the tables in LDS)
#endif
hipError_t launch_finalize(const FinalizeParams &p, hipStream_t s)
{
    if (p.n_streams == 0) return hipSuccess;
    const bool streaming = p.n_streams == 1 && !p.corpus_hist && !p.out_integrated && !p.out_lra && p.sub_stride == 0;
    if (streaming) hipLaunchKernelGGL(k_finalize_stream, dim3(1), dim3(64), 0, s, p);
    else {
        // four waves per stream (even a 100-sub-block stream moves two 8 KB histograms in and out of LDS: 64 / 128 / 256 / 512
        // threads at the bench shape: 0.043 / 0.036 / 0.033 / 0.045 ms), sixteen for a long stream
        const uint64_t nsub = p.sub_end - p.sub_begin;
        const uint32_t threads = nsub > 2048 ? 1024u : 256u;
        // a handful of short streams: the launch is a chain of memory round trips, not work -> the form that shortens the chain
        if (p.n_streams <= SS_FINALIZE_SMALL_MAX && threads == 256u) hipLaunchKernelGGL(k_finalize<true>, dim3(p.n_streams), dim3(threads), 0, s, p);
        else hipLaunchKernelGGL(k_finalize<false>, dim3(p.n_streams), dim3(threads), 0, s, p);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void k_hist_eval(const unsigned long long *hist2000, const double *en,
                                                  const double *bd, double *out2, ReadingsExtra x)
{
    __shared__ unsigned long long hb[kHistBins];
    __shared__ unsigned long long hs[kHistBins];
    __shared__ double tab[2 * kHistBins + 1];               // the two tables beside the histograms: one round trip for all four
    for (int i = threadIdx.x; i < kHistBins; i += 64) {
        hb[i] = hist2000[i]; hs[i] = hist2000[kHistBins + i];
        tab[i] = en[i]; tab[kHistBins + i] = bd[i];
    }
    if (threadIdx.x == 0) tab[2 * kHistBins] = bd[kHistBins];
    if (x.peaks_dst) {
        x.peaks_dst[threadIdx.x] = x.peaks_src[threadIdx.x];
        x.peaks_dst[64 + threadIdx.x] = x.peaks_src[64 + threadIdx.x];
    }
    __syncthreads();
    eval_hist(hb, hs, tab, tab + kHistBins, &out2[0], &out2[1]);
    if (x.flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(x.flag, x.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_hist_eval(const uint64_t *hist2000, const double *energies, const double *bounds,
                            double *out2, hipStream_t s, const ReadingsExtra *peaks)
{
    static_assert(2 * kMaxChannels == 128, "k_hist_eval copies two floats per lane");
    hipLaunchKernelGGL(k_hist_eval, dim3(1), dim3(64), 0, s,
                       reinterpret_cast<const unsigned long long *>(hist2000), energies, bounds, out2,
                       peaks ? *peaks : ReadingsExtra{nullptr, nullptr, nullptr, 0u});
    return hipGetLastError();
}

// mean square over the last `frames` frames of the filtered ring, channel-weighted
// (calc_gating_block on the ring "as is": loudness_shortterm / loudness_momentary).
// Two stages with a fixed reduction shape (bit-reproducible): kRingBlocks partial sums, then one block — in ONE launch: the
// workgroup that finishes last (a counter behind the partial sums, wrapped back to zero by atomicInc for the next launch)
// reduces the partial sums.  (Through round 3 the second stage was a launch of its own: one more of a tick's launches.)
// The window is ONE run of ring elements with at most one wrap (frames <= ring_frames), so an element's position is an add and
// a compare in 32 bits and its channel advances by a constant step: no division in the loop (the first version divided two
// 64-bit numbers per element — most of its 9.8 us inside a tick).  256 workgroups: a thread takes four or five elements of the
// three-second window at 48 kHz stereo, all requested before the first is used.
constexpr int kRingBlocks = 256;
__global__ __launch_bounds__(256) void k_ring_energy(const double *ring, uint32_t ring_elems, uint32_t C, uint32_t begin_elem,
                                                     uint32_t total, double frames,
                                                     const double *weights, double *partial, double *out)
{
    __shared__ double red[256];
    __shared__ uint32_t is_last;
    constexpr uint32_t kStride = (uint32_t)kRingBlocks * 256u;
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t cstep = kStride % C;
    uint32_t c = tid % C;
    double acc = 0.0;
    uint32_t i = tid;
    for (; i + 3u * kStride < total; i += 4u * kStride) {
        double y[4], w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t e = begin_elem + i + (uint32_t)q * kStride;
            if (e >= ring_elems) e -= ring_elems;
            y[q] = ring[e];
            w[q] = weights[c];
            c += cstep; if (c >= C) c -= C;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc = w[q] != 0.0 ? fma(w[q] * y[q], y[q], acc) : acc;      // (weight 0: a channel the crate does not
    }                                                                                              //  filter — its ring stays zero there, whatever the input)
    for (; i < total; i += kStride) {
        uint32_t e = begin_elem + i;
        if (e >= ring_elems) e -= ring_elems;
        const double y = ring[e];
        const double wc = weights[c];
        acc = wc != 0.0 ? fma(wc * y, y, acc) : acc;
        c += cstep; if (c >= C) c -= C;
    }
    // workgroup sum: shuffle tree inside each wave, then the four wave sums in a fixed order
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();                                                   // the partial sum is visible before the count
        is_last = atomicInc(reinterpret_cast<unsigned int *>(partial + kRingBlocks), (unsigned int)kRingBlocks - 1u) == (unsigned int)kRingBlocks - 1u;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    red[threadIdx.x] = __builtin_nontemporal_load(partial + threadIdx.x);
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double e = red[0] / frames;
        out[0] = e;
        out[1] = e <= 0.0 ? -INFINITY : 10.0 * log10(e) - 0.691;   // energy_to_loudness
    }
}

// scratch: kRingBlocks partial sums + the completion counter (zero before the first launch, see ss_analyzer.cpp).
// `out` may be device memory or pinned host memory mapped into the device (the tick drivers: no copy behind the kernel).
hipError_t launch_ring_energy(const double *ring, uint64_t ring_frames, uint32_t channels,
                              uint64_t end_frame, uint64_t frames, const double *weights,
                              double *out, double *scratch, hipStream_t s)
{
    if (frames == 0 || frames > ring_frames || ring_frames * channels >= (1ull << 31)) return hipErrorInvalidValue;
    // ring position of absolute frame f is f % ring_frames; frames before 0 are the zeroed ring
    const uint64_t begin = (end_frame % ring_frames + ring_frames - frames) % ring_frames;
    hipLaunchKernelGGL(k_ring_energy, dim3(kRingBlocks), dim3(256), 0, s, ring, (uint32_t)(ring_frames * channels), channels,
                       (uint32_t)(begin * channels), (uint32_t)(frames * channels), (double)frames, weights, scratch, out);
    return hipGetLastError();
}

}  // namespace ssk
