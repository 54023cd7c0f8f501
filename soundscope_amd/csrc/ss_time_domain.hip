// ss_time_domain.hip: K-weighting, sub-block energies, peaks, fused decimation — hand-written gfx950 (CDNA4, wave64) kernels of the soundscope analyzer hot path.
// Reference semantics: /root/reference/src/analyzer.rs (get_fft :55-105, get_waveform :107-137,
// add_samples/getters :139-164, calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as restated in DESIGN.md.
// Nothing here is translated from the reference: the reference has no GPU code.
#include "ss_kernels.h"
#include <cstdlib>

#ifndef SS_TD_WAVES
#define SS_TD_WAVES 4    // min waves per SIMD the time-domain kernel is register-allocated for
#endif

namespace ssk {
// ============================================================================
//  Time domain: K-weighting IIR (f64), 100 ms sub-block energies, sample peak
//  and polyphase true peak (f32) — EbuR128::add_frames_f32 of ebur128 0.1.10
//  (called at analyzer.rs:140 and :176), re-cut for CDNA4:
//
//  Unit of work = one WAVE (64 lanes) walking one time segment of one stream,
//  tile by tile; a tile is (a piece of) one 100 ms sub-block staged into the
//  wave's private LDS slice in its natural interleaved layout behind a
//  24-frame halo.  Waves never synchronise with each other: no s_barrier in
//  the kernel, 16 waves per CU hide each other's LDS / HBM latency.
//
//  * Segments.  A stream is cut into `nseg` runs of whole sub-blocks so that a
//    batch of a few hundred streams still fills 4096 wave slots.  Segment k > 0
//    starts its filter `warm` sub-blocks (0.3 s) early from a zero state and
//    discards that run-in: the K-weighting poles (|z| <= 0.99502 at 48 kHz,
//    i.e. e^-240 per second at any rate) shrink the influence of the unknown
//    initial state by e^-72 ~ 5e-32 — sixteen orders below f64 rounding — so the
//    result equals the sequential recurrence to the last bit that f64 carries.
//    Segment 0 (and every streaming call, nseg = 1) starts from the true state.
//  * K-weighting on the f64 VALU.  Each lane owns one (chunk of L frames,
//    channel); the recurrence is cut by  state_out = A^L state_in + zero_state:
//      pass 1: per chunk, run the state recurrence from zero              (4 FMA)
//      scan  : in-wave Hillis-Steele over chunks (ds_bpermute shuffles) with the
//              constant matrices (A^L)^(2^k)
//      pass 2: rerun each chunk from its true initial state, accumulate y^2.
//    L is chosen with (L-1)*C = 0 (mod 32) so the per-lane walk through the
//    interleaved tile is bank-conflict free without padding.
//  * True peak on the f32 MATRIX pipe, concurrently with other waves' f64 VALU
//    work: the polyphase FIR  y_f[n] = sum_t c_f[t] x[n-t]  over a block of BLK
//    consecutive outputs is a banded-Toeplitz product
//      D[(f,r), col] = sum_k A[(f,r), k] * B[k, col],
//      A[(f,r), k] = c_f[HIST-1 + r - k],  B[k, col] = x[start_col - (HIST-1) + k],
//    issued as v_mfma_f32_16x16x4_f32 (exact f32 fma chain).  Factor 4: 3 phases x 5
//    outputs = 15 rows over a 16-sample window (4 MFMAs per 16 columns, 70 % of
//    the MACs useful); factor 2: 16 outputs over a 39-sample window (10 MFMAs).
//    Phase 0 of the interpolator is the identity tap: it equals the sample peak,
//    which true_peak() maxes in anyway (analyzer.rs:159-164 -> ebur128 true_peak).
// ============================================================================
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));


template <int FACTOR>
struct TpCfg {
    static constexpr int HIST = (FACTOR == 2) ? 24 : 12;     // taps per polyphase branch
    static constexpr int NPH = (FACTOR == 4) ? 3 : (FACTOR == 2 ? 1 : 0);
    static constexpr int BLK = (FACTOR == 4) ? 5 : 16;       // outputs per column
    static constexpr int ROWS = NPH * BLK;                   // 15 or 16
    static constexpr int KSTEPS = (BLK + HIST - 1 + 3) / 4;  // 4 or 10
};

__device__ __forceinline__ void mat4_apply_add(const double *__restrict__ M, const double (&x)[4], double (&z)[4])
{
#pragma unroll
    for (int r = 0; r < 4; r++)
        z[r] = fma(M[r * 4 + 0], x[0], fma(M[r * 4 + 1], x[1], fma(M[r * 4 + 2], x[2], fma(M[r * 4 + 3], x[3], z[r]))));
}

#ifndef SS_TP_F16
#define SS_TP_F16 1
#endif
constexpr int kTdHaloFrames = 24;     // minimum halo: >= HIST-1 of the longest branch (multiple of 4: the tile stays 16-B aligned)
constexpr int kTdTailFrames = 16;     // slack past the tile end for the last MFMA window
constexpr int kTdWavesPerBlock = 4;
#ifndef SS_TD_PREFETCH
#define SS_TD_PREFETCH 8
#endif
constexpr int kTdPrefetch = SS_TD_PREFETCH;        // float4 per lane held in flight for the next tile
constexpr int kTdBatch = 11;          // LDS reads issued together in the sequential passes

// one K-weighting state step (DF-II, zero-based state v1..v4); the critical path is one FMA
#define SS_KW_STATE(xd)                         \
    double t_ = fma(-a2, v2, (xd));             \
    t_ = fma(-a3, v3, t_);                      \
    t_ = fma(-a4, v4, t_);                      \
    const double v0_ = fma(-a1, v1, t_);
#define SS_KW_SHIFT() v4 = v3; v3 = v2; v2 = v1; v1 = v0_;
#define SS_KW_OUT()                              \
    double u_ = b1 * v1;                         \
    u_ = fma(b2, v2, u_);                        \
    u_ = fma(b3, v3, u_);                        \
    u_ = fma(b4, v4, u_);                        \
    const double y_ = fma(b0, v0_, u_);
// Look-ahead form of the same recurrence for full chunks: the terms that do not involve the newest
// state are folded into partial sums one, two and three samples ahead, so every step issues four
// independent FMAs and the loop-carried dependency is a single FMA (v_i = r1 - a1 v_{i-1}).
//   r1 = x_i     - a2 v_{i-2} - a3 v_{i-3} - a4 v_{i-4}
//   r2 = x_{i+1} - a3 v_{i-2} - a4 v_{i-3}
//   r3 = x_{i+2} - a4 v_{i-2}
#define SS_KW_LA_INIT(x0, x1, x2)                                   \
    double r1 = fma(-a4, v4, fma(-a3, v3, fma(-a2, v2, (x0))));      \
    double r2 = fma(-a4, v3, fma(-a3, v2, (x1)));                    \
    double r3 = fma(-a4, v2, (x2));
#define SS_KW_LA_STEP(xn)                        \
    const double v0_ = fma(-a1, v1, r1);         \
    r1 = fma(-a2, v1, r2);                       \
    r2 = fma(-a3, v1, r3);                       \
    r3 = fma(-a4, v1, (xn));
// output taps as partial sums too: y_i = b0 v_i + u1, every update depends on v_i only
#define SS_KW_LA_OUT_INIT()                                          \
    double u1 = fma(b4, v4, fma(b3, v3, fma(b2, v2, b1 * v1)));      \
    double u2 = fma(b4, v3, fma(b3, v2, b2 * v1));                   \
    double u3 = fma(b4, v2, b3 * v1);                                \
    double u4 = b4 * v1;
#define SS_KW_LA_OUT()                           \
    const double y_ = fma(b0, v0_, u1);          \
    u1 = fma(b1, v0_, u2);                       \
    u2 = fma(b2, v0_, u3);                       \
    u3 = fma(b3, v0_, u4);                       \
    u4 = b4 * v0_;

// CT: compile-time channel count (0 = runtime)
// WAVE: 0 no decimation, 1 fused get_waveform (any bin geometry), 2 the same for an exact-integer samples-per-bin that is
// a multiple of four (<= 128) with 16-byte aligned tiles (the host checks), 3 the same for 128 < spp <= 1000
template <int FACTOR, bool RING, int CT, int WAVE>
__global__ __launch_bounds__(64 * kTdWavesPerBlock, SS_TD_WAVES) void k_time_domain(TdParams p, uint32_t L, uint32_t tile_len,
                                                                                      uint32_t wave_lds_floats, uint32_t halo_frames)
{
    using Cfg = TpCfg<FACTOR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * kTdWavesPerBlock + wave_in_block;   // global wave = (stream, segment)
    if (gw >= p.n_streams * p.nseg) return;                                // whole wave leaves (no barriers used)
    const uint32_t stream = gw / p.nseg, sg = gw - stream * p.nseg;

    float *tilebuf = reinterpret_cast<float *>(smem) + (size_t)wave_in_block * wave_lds_floats;
    const TdConst &K = *p.k;
    const uint32_t C = CT ? (uint32_t)CT : p.channels;
    const uint32_t S = p.s100;
    const uint32_t nch = 64u / C;                       // chunks per tile (C <= 64)
    const uint32_t chunk = lane / C, ch = lane - chunk * C;
    const bool lane_ok = chunk < nch;
    float *tile = tilebuf + halo_frames * C;            // tile[f*C + c]; tile[-q*C + c] = x[-q]
    unsigned *tpk = reinterpret_cast<unsigned *>(tilebuf + wave_lds_floats - kMaxChannels);   // per-channel peak slots
    TdState &st = p.state[stream];
    const float *src = p.pcm + (size_t)stream * p.stream_stride;

    // ---- this wave's frame range (relative to the call) and its run-in.
    // Segment boundaries sit on the absolute sub-block grid so every sub-block has one owner.
    // Multi-segment (batch) launches start from a reset meter by contract: segments must not read
    // state another segment of the same launch writes at its end.
    const bool carry_in = (p.nseg == 1);
    const uint64_t fed0 = carry_in ? st.frames_fed : 0;
    uint64_t seg_begin, seg_end;                        // frames of this call, [begin, end)
    const uint64_t n_frames = p.frames_of ? p.frames_of[stream] : p.n_frames;         // ragged batches: this stream's own length
    if (p.nseg == 1) { seg_begin = 0; seg_end = n_frames; }
    else {
        seg_begin = (uint64_t)sg * p.seg_sub * S;
        seg_end = (sg + 1 == p.nseg) ? n_frames : (uint64_t)(sg + 1) * p.seg_sub * S;
        if (seg_begin > n_frames) seg_begin = n_frames;
        if (seg_end > n_frames) seg_end = n_frames;
    }
    const uint64_t warm_frames = (sg == 0) ? 0 : (uint64_t)p.warm_sub * S;   // sg > 0 implies seg_begin >= warm
    uint64_t pos = seg_begin - warm_frames;             // first frame this wave reads
    uint32_t off = (uint32_t)((fed0 + pos) % S);        // position inside the current sub-block
    uint64_t sb = (fed0 + pos) / S;                     // absolute sub-block index

    // ---- initial state: the stream's carried state (streaming call) or zeros
    double cv[4] = {0.0, 0.0, 0.0, 0.0};               // carry, held by every lane of channel `ch`
    double e_run = 0.0;                                 // this lane's share of the current sub-block's energy
    float sp_run = 0.0f, tp_run = 0.0f;
    if (carry_in && lane_ok) {
#pragma unroll
        for (int q = 0; q < 4; q++) cv[q] = st.v[ch][q];
        if (lane < C) e_run = st.acc[lane];
    }
    if (lane < C) {
        for (uint32_t q = 1; q <= halo_frames; q++)
            tile[-(int)(q * C) + (int)lane] = (carry_in && q <= (uint32_t)kTpHistMax) ? st.tp_hist[lane][q - 1] : 0.0f;
    }
    tpk[lane] = 0u;

    // ---- min-max decimation cursor (Analyzer::get_waveform fused into this pass): a bin is produced by
    // the wave whose tile holds the bin's LAST sample; its first samples may sit in the halo.
    const uint64_t wv_len = n_frames * C;
    const double wv_spp = WAVE ? (double)wv_len / (double)p.wave_window : 0.0;
    const uint32_t wv_spp_i = WAVE ? (uint32_t)wv_spp : 0u;
    uint32_t wv_cur = 0;
    if (WAVE && sg != 0) {
        const uint64_t b0 = seg_begin * C;               // first interleaved index this wave owns
        double gq = floor((double)b0 / wv_spp) - 2.0;
        uint32_t g = gq > 0.0 ? (uint32_t)gq : 0u;
        for (;;) {                                        // first bin whose end lies beyond b0
            const double ed = ceil((double)(g + 1) * wv_spp);
            uint64_t e = (uint64_t)ed;
            if (e > wv_len) e = wv_len;
            if (e > b0 || g >= p.wave_window) break;
            g++;
        }
        wv_cur = g;
    }

    // ---- constant A fragments of the banded-Toeplitz true-peak product, and this lane's column role
    const int mrow = lane & 15, kq = lane >> 4;
    float afrag[Cfg::KSTEPS > 0 ? Cfg::KSTEPS : 1];
    const bool tp_fixed = (16u % C) == 0u;              // each lane's column always belongs to one channel
    const uint32_t tp_bpg = 16u / (tp_fixed ? C : 1u);  // blocks per 16-column group
    const uint32_t tp_c = (uint32_t)mrow % C;
    int tp_lane_off = 0;                                // float offset of this lane's window inside group 0
    if (FACTOR != 0) {
        const int fph = mrow / Cfg::BLK, r = mrow - fph * Cfg::BLK;
#pragma unroll
        for (int s = 0; s < Cfg::KSTEPS; s++) {
            const int k = 4 * s + kq;
            const int t = Cfg::HIST - 1 + r - k;
            afrag[s] = (mrow < Cfg::ROWS && t >= 0 && t < Cfg::HIST) ? K.tp[fph][t] : 0.0f;   // row 15 (factor 4) is all zero
        }
        tp_lane_off = ((int)((uint32_t)mrow / C) * Cfg::BLK - (Cfg::HIST - 1) + kq) * (int)C + (int)tp_c;
    }
    // f16 form of the same product (factor 4, K = 16 in ONE v_mfma_f32_16x16x16_f16): taps and samples are split
    // into f16 pairs, c = c_hi + c_lo, 256 x = x_hi + x_lo, and hi*hi + hi*lo + lo*hi accumulate in f32 (the dropped
    // lo*lo and the split remainders are < 1e-6 relative).  18 cycles per MFMA instead of 32 and four times the
    // depth, and unlike the f32 MFMA it runs beside other waves' f64 VALU work (tools/ubench4.hip).
    // Lane (mrow, kq) holds A[mrow][4 kq + j] and B[4 kq + j][mrow], j = 0..3.
    constexpr bool kTpF16 = (FACTOR == 4) && (SS_TP_F16 != 0);
    halfx4 a16_hi = {0, 0, 0, 0}, a16_lo = {0, 0, 0, 0};
    int tp_lane_off16 = 0;
    if (kTpF16) {
        const int fph = mrow / Cfg::BLK, r = mrow - fph * Cfg::BLK;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 4 * kq + j;
            const int t = Cfg::HIST - 1 + r - k;
            const float c = (mrow < Cfg::ROWS && t >= 0 && t < Cfg::HIST) ? K.tp[fph][t] : 0.0f;
            const _Float16 ch_ = (_Float16)c;
            a16_hi[j] = ch_;
            a16_lo[j] = (_Float16)(c - (float)ch_);
        }
        tp_lane_off16 = ((int)((uint32_t)mrow / C) * Cfg::BLK - (Cfg::HIST - 1) + 4 * kq) * (int)C + (int)tp_c;
    }
    float tp_run16 = 0.0f;                              // running max of the f16 path, in units of 256
    uint32_t tp_clean = carry_in ? 0u : 0x40000000u;    // frames before the current tile known to be within +-128 (a carried halo may hold anything)
    const double a1 = K.a[1], a2 = K.a[2], a3 = K.a[3], a4 = K.a[4];
    const double b0 = K.b[0], b1 = K.b[1], b2 = K.b[2], b3 = K.b[3], b4 = K.b[4];

    // tile geometry: a tile never crosses a sub-block boundary of the absolute grid; a sub-block is
    // cut into equal pieces of at most tile_len frames
#define SS_TILE_FRAMES(at, off_in, out)                                     \
    do {                                                                    \
        uint64_t n_ = 0;                                                    \
        if ((at) < seg_end) {                                               \
            n_ = tile_len - ((off_in) % tile_len);                          \
            if (n_ > S - (off_in)) n_ = S - (off_in);                       \
            if (n_ > seg_end - (at)) n_ = seg_end - (at);                   \
        }                                                                   \
        (out) = (uint32_t)n_;                                               \
    } while (0)
    // register prefetch of a tile: kTdPrefetch float4 per lane (clamped index keeps it branch-free)
#define SS_PREFETCH(at, frames)                                             \
    do {                                                                    \
        const float *g_ = src + (at) * C;                                   \
        const uint32_t nv_ = ((frames) * C) >> 2;                           \
        if (nv_ != 0 && (reinterpret_cast<uintptr_t>(g_) & 15u) == 0) {     \
            const float4 *g4_ = reinterpret_cast<const float4 *>(g_);       \
            _Pragma("unroll") for (int q_ = 0; q_ < kTdPrefetch; q_++) {    \
                uint32_t i_ = lane + 64u * q_;                              \
                i_ = i_ < nv_ ? i_ : nv_ - 1;                               \
                pf[q_] = g4_[i_];                                           \
            }                                                               \
        }                                                                   \
    } while (0)

    float4 pf[kTdPrefetch];
#pragma unroll
    for (int q = 0; q < kTdPrefetch; q++) pf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t seg;
    SS_TILE_FRAMES(pos, off, seg);
    SS_PREFETCH(pos, seg);

    while (seg != 0) {
        // keep the scan matrices in memory (scalar loads at the point of use): hoisting all of them
        // out of the tile loop would cost 224 SGPRs
        const double *mpow = &K.m_pow[0][0];
        asm volatile("" : "+s"(mpow));
        const bool warm = pos < seg_begin;              // run-in tile: filter only
        const uint32_t nchunks = (seg + L - 1) / L;

        // ---- stage the tile from the prefetch registers (remainder / unaligned: direct)
        {
            const float *g = src + pos * C;
            const uint32_t total = seg * C;
            uint32_t done = 0;
            if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
                const uint32_t nv = total >> 2;
                float4 *t4 = reinterpret_cast<float4 *>(tile);
#pragma unroll
                for (int q = 0; q < kTdPrefetch; q++) {
                    const uint32_t i = lane + 64u * q;
                    if (i < nv) t4[i] = pf[q];
                }
                const float4 *g4 = reinterpret_cast<const float4 *>(g);
                for (uint32_t i = lane + 64u * kTdPrefetch; i < nv; i += 64u) t4[i] = g4[i];
                done = nv << 2;
            }
            for (uint32_t i = done + lane; i < total; i += 64u) tile[i] = g[i];
            for (uint32_t i = total + lane; i < total + kTdTailFrames * C; i += 64u) tile[i] = 0.0f;
        }
        // next tile's loads fly while this one is processed
        const uint64_t npos = pos + seg;
        uint32_t noff = off + seg;
        const bool sub_done = (noff == S);
        if (sub_done) noff = 0;
        uint32_t nseg_frames;
        SS_TILE_FRAMES(npos, noff, nseg_frames);
        SS_PREFETCH(npos, nseg_frames);
        __builtin_amdgcn_wave_barrier();               // LDS is in-order per wave: only ordering is needed

        // ---- min-max decimation of the bins that END inside this tile (analyzer.rs:107-137): bin i =
        // [floor(i*spp), min(ceil((i+1)*spp), len)), the same f64 expressions as the reference; 16 lanes
        // per bin, IEEE minNum/maxNum seeded with NaN (f32::min/max ignore NaN; an all-NaN bin stays NaN)
        if (WAVE && !warm) {
            // (the host only fuses when the stream length fits 31 bits, so 32-bit indices are exact)
            const uint32_t t0 = (uint32_t)(pos * C), t1 = (uint32_t)((pos + seg) * C);   // tile's interleaved index range
            const uint32_t wlen = (uint32_t)wv_len;
            const uint32_t lane16 = lane & 15u;
            // Exact-integer samples-per-bin that is a multiple of four (96 at 48 kHz stereo, W = duration in ms):
            // floor(i spp) and ceil((i+1) spp) are the integer products themselves, bins are 16-byte aligned in the
            // tile, so eight lanes cover a bin with 16-byte LDS reads: eight bins per iteration.
            if (WAVE >= 2) {
                const uint32_t lane8 = lane & 7u, n4 = wv_spp_i >> 2;
                for (;;) {
                    const uint32_t i = wv_cur + (lane >> 3);
                    const uint32_t bs = i * wv_spp_i, be = bs + wv_spp_i;           // be <= len: W spp == len exactly
                    const bool valid = i < p.wave_window && be <= t1;
                    float mn = __builtin_nanf(""), mx = __builtin_nanf("");
                    if (valid) {
                        const float4 *bp4 = reinterpret_cast<const float4 *>(tile + ((int)bs - (int)t0));   // may reach into the halo
#pragma unroll
                        for (int it = 0; it < 4; it++) {                              // spp <= 128: four clamped reads cover a bin
                            uint32_t j = lane8 + 8u * it;
                            j = j < n4 ? j : n4 - 1;
                            const float4 v = bp4[j];
                            mn = fminf(fminf(mn, v.x), fminf(fminf(v.y, v.z), v.w));
                            mx = fmaxf(fmaxf(mx, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
                        }
                        if (WAVE == 3)                                                 // longer bins (192 at 96 kHz stereo)
                            for (uint32_t j = lane8 + 32u; j < n4; j += 8u) {
                                const float4 v = bp4[j];
                                mn = fminf(fminf(mn, v.x), fminf(fminf(v.y, v.z), v.w));
                                mx = fmaxf(fmaxf(mx, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
                            }
                    }
                    // 8-lane all-reduce: xor 1, xor 2 (quad_perm), then the mirror inside each half row
#define SS_DPP(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (ctrl), 0xF, 0xF, false))
                    mn = fminf(mn, SS_DPP(mn, 0xB1)); mx = fmaxf(mx, SS_DPP(mx, 0xB1));
                    mn = fminf(mn, SS_DPP(mn, 0x4E)); mx = fmaxf(mx, SS_DPP(mx, 0x4E));
                    mn = fminf(mn, SS_DPP(mn, 0x141)); mx = fmaxf(mx, SS_DPP(mx, 0x141));
#undef SS_DPP
                    if (valid && lane8 == 0) {
                        float2 *o = reinterpret_cast<float2 *>(p.wave_out + (size_t)stream * p.wave_stride) + i;
                        *o = make_float2(mn, mx);
                    }
                    const uint32_t nvalid = (uint32_t)__popcll(__ballot(valid && lane8 == 0));
                    wv_cur += nvalid;
                    if (nvalid < 8u) break;                   // the next bin ends beyond this tile
                }
            } else
            for (;;) {
                const uint32_t i = wv_cur + (lane >> 4);
                const uint32_t bs = (uint32_t)((double)i * wv_spp);
                uint32_t be = (uint32_t)ceil((double)(i + 1) * wv_spp);
                if (be > wlen) be = wlen;
                const bool valid = i < p.wave_window && be <= t1 && bs < wlen;
                float mn = __builtin_nanf(""), mx = __builtin_nanf("");
                if (valid) {
                    const float *bp = tile + ((int)bs - (int)t0);      // may reach into the halo
                    const uint32_t n = be - bs;                       // >= 1
                    // seven clamped reads cover n <= 112 without predicates (a repeated element cannot
                    // change a min/max); longer bins finish in the loop
#pragma unroll
                    for (int it = 0; it < 7; it++) {
                        uint32_t j = lane16 + 16u * it;
                        j = j < n ? j : n - 1;
                        const float v = bp[j];
                        mn = fminf(mn, v);
                        mx = fmaxf(mx, v);
                    }
                    for (uint32_t j = lane16 + 112u; j < n; j += 16u) {
                        const float v = bp[j];
                        mn = fminf(mn, v);
                        mx = fmaxf(mx, v);
                    }
                }
                // 16-lane all-reduce with DPP row rotations (VALU rate; ds_bpermute costs ~8x more)
#define SS_ROW_ROR(x, n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 | (n_), 0xF, 0xF, false))
                mn = fminf(mn, SS_ROW_ROR(mn, 8)); mx = fmaxf(mx, SS_ROW_ROR(mx, 8));
                mn = fminf(mn, SS_ROW_ROR(mn, 4)); mx = fmaxf(mx, SS_ROW_ROR(mx, 4));
                mn = fminf(mn, SS_ROW_ROR(mn, 2)); mx = fmaxf(mx, SS_ROW_ROR(mx, 2));
                mn = fminf(mn, SS_ROW_ROR(mn, 1)); mx = fmaxf(mx, SS_ROW_ROR(mx, 1));
#undef SS_ROW_ROR
                if (valid && lane16 == 0) {
                    float2 *o = reinterpret_cast<float2 *>(p.wave_out + (size_t)stream * p.wave_stride) + i;
                    *o = make_float2(mn, mx);
                }
                const uint32_t nvalid = (uint32_t)__popcll(__ballot(valid && lane16 == 0));
                wv_cur += nvalid;
                if (nvalid < 4u) break;                   // the next bin ends beyond this tile
            }
        }

        const bool active = lane_ok && chunk < nchunks;
        const uint32_t len = active ? ((seg - chunk * L) < L ? (seg - chunk * L) : L) : 0u;
        const float *xs = tile + (size_t)chunk * L * C + ch;
        const uint32_t nb_full = L / kTdBatch;          // whole batches in a full chunk

        // ---- pass 1: zero-state response of the state recurrence
        double z[4] = {0.0, 0.0, 0.0, 0.0};
        {
            double v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
            uint32_t i = 0;
            if (len == L) {                             // full chunk: batched, predicate-free, look-ahead form
                const float *xp = xs + 3 * C;           // the batch loop consumes x[i + 3]
                SS_KW_LA_INIT((double)xs[0], (double)xs[C], (double)xs[2 * C])
                for (uint32_t bq = 0; bq < nb_full; bq++, xp += kTdBatch * C) {
                    float xb[kTdBatch];
#pragma unroll
                    for (int u = 0; u < kTdBatch; u++) xb[u] = xp[u * (int)C];   // reaches <= 3 frames past the chunk (slack)
#pragma unroll
                    for (int u = 0; u < kTdBatch; u++) { SS_KW_LA_STEP((double)xb[u]) SS_KW_SHIFT() }
                }
                i = nb_full * kTdBatch;
                for (; i < len; i++) { SS_KW_LA_STEP((double)xs[(i + 3) * C]) SS_KW_SHIFT() }
            }
            for (; i < len; i++) { SS_KW_STATE((double)xs[i * C]) SS_KW_SHIFT() }
            z[0] = v1; z[1] = v2; z[2] = v3; z[3] = v4;
            if (active && chunk == 0) mat4_apply_add(mpow, cv, z);
        }
        // ---- in-wave scan over chunks: z_i += (A^L)^(2^k) z_{i - 2^k}
        for (int kstep = 0; (1u << kstep) < nchunks; kstep++) {
            const uint32_t d = (1u << kstep) * C;
            const double xin[4] = {__shfl_up(z[0], d, 64), __shfl_up(z[1], d, 64), __shfl_up(z[2], d, 64), __shfl_up(z[3], d, 64)};
            if (active && lane >= d) mat4_apply_add(mpow + 16 * kstep, xin, z);
        }
        // z = state after this lane's chunk (valid for full chunks); initial state = previous chunk's
        double v1, v2, v3, v4;
        {
            const double p0 = __shfl_up(z[0], C, 64), p1 = __shfl_up(z[1], C, 64), p2 = __shfl_up(z[2], C, 64), p3 = __shfl_up(z[3], C, 64);
            const bool first = chunk == 0;
            v1 = first ? cv[0] : p0; v2 = first ? cv[1] : p1; v3 = first ? cv[2] : p2; v4 = first ? cv[3] : p3;
        }

        // ---- pass 2: true-state rerun + energy + sample peak
        float sp = 0.0f;                                // this lane's max |x| over its chunk (also steers the true-peak path)
        {
            double e = 0.0;
            uint32_t i = 0;
            const uint64_t ring_base = fed0 + pos + (uint64_t)chunk * L;
            if (len == L) {
                // sample peak over x[0 .. L+2]: the three look-ahead samples are the next chunk's (or the
                // zeroed slack behind the tile), so including them cannot change the channel's maximum
                const float *xp = xs + 3 * C;
                const float xa = xs[0], xb1 = xs[C], xc = xs[2 * C];
                sp = fmaxf(fmaxf(fabsf(xa), fabsf(xb1)), fabsf(xc));
                SS_KW_LA_INIT((double)xa, (double)xb1, (double)xc)
                SS_KW_LA_OUT_INIT()
                for (uint32_t bq = 0; bq < nb_full; bq++, xp += kTdBatch * C) {
                    float xb[kTdBatch];
#pragma unroll
                    for (int u = 0; u < kTdBatch; u++) xb[u] = xp[u * (int)C];
#pragma unroll
                    for (int u = 0; u < kTdBatch; u++) {
                        sp = fmaxf(sp, fabsf(xb[u]));
                        SS_KW_LA_STEP((double)xb[u]) SS_KW_LA_OUT() SS_KW_SHIFT()
                        e = fma(y_, y_, e);
                        if (RING) p.ring[((ring_base + bq * kTdBatch + u) % p.ring_frames) * C + ch] = y_;
                    }
                }
                i = nb_full * kTdBatch;
                for (; i < len; i++) {
                    const float xn = xs[(i + 3) * C];
                    sp = fmaxf(sp, fabsf(xn));
                    SS_KW_LA_STEP((double)xn) SS_KW_LA_OUT() SS_KW_SHIFT()
                    e = fma(y_, y_, e);
                    if (RING) p.ring[((ring_base + i) % p.ring_frames) * C + ch] = y_;
                }
            }
            for (; i < len; i++) {
                const float xf = xs[i * C];
                sp = fmaxf(sp, fabsf(xf));
                SS_KW_STATE((double)xf) SS_KW_OUT() SS_KW_SHIFT()
                e = fma(y_, y_, e);
                if (RING) p.ring[((ring_base + i) % p.ring_frames) * C + ch] = y_;
            }
            if (!warm) { e_run += e; sp_run = fmaxf(sp_run, sp); }
        }
        // the f16 true-peak product needs 256 |x| inside the f16 range: a wave-uniform test on the sample peaks of this
        // tile and of the frames before it that the FIR window can reach
        const bool tp_big_now = kTpF16 && (__ballot(sp > 128.0f) != 0ull);
        const bool tp_big = tp_big_now || tp_clean < (uint32_t)(Cfg::HIST - 1);
        tp_clean = tp_big_now ? 0u : (tp_clean + seg < 0x40000000u ? tp_clean + seg : 0x40000000u);
        // ---- true peak on the matrix pipe (not during the run-in)
        if (FACTOR != 0 && !warm) {
            const uint32_t nblk = (seg + Cfg::BLK - 1) / Cfg::BLK;     // blocks per channel
            const uint32_t ncol = nblk * C;
            const uint32_t ngroups = (ncol + 15) >> 4;
            if (tp_fixed) {
                constexpr int GS = 16 * Cfg::BLK;                      // floats per group (16 columns x BLK outputs)
                const uint32_t nfull = seg / (tp_bpg * Cfg::BLK);      // groups whose every output lies inside the tile
                uint32_t gi = 0;
                if (kTpF16 && !tp_big) {                                // anything beyond +-128 full scale takes the f32 product below
                    const float *bq = tile + tp_lane_off16;
                    for (; gi + 2 <= nfull; gi += 2, bq += 2 * GS) {
                        halfx4 h0, l0, h1, l1;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float x0 = bq[j * (int)C] * 256.0f, x1 = bq[GS + j * (int)C] * 256.0f;
                            const _Float16 xh0 = (_Float16)x0, xh1 = (_Float16)x1;
                            h0[j] = xh0; l0[j] = (_Float16)(x0 - (float)xh0);      // exact remainder, then rounded to f16
                            h1[j] = xh1; l1[j] = (_Float16)(x1 - (float)xh1);
                        }
                        floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, h0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, h1, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, l0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, l1, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_lo, h0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_lo, h1, acc1, 0, 0, 0);
                        tp_run16 = fmaxf(fmaxf(tp_run16, fmaxf(fabsf(acc0[0]), fabsf(acc0[1]))), fmaxf(fabsf(acc0[2]), fabsf(acc0[3])));
                        tp_run16 = fmaxf(fmaxf(tp_run16, fmaxf(fabsf(acc1[0]), fabsf(acc1[1]))), fmaxf(fabsf(acc1[2]), fabsf(acc1[3])));
                    }
                }
                const float *bp = tile + tp_lane_off + (size_t)gi * GS;
                for (; gi + 2 <= nfull; gi += 2, bp += 2 * GS) {
                    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    float bv0[Cfg::KSTEPS], bv1[Cfg::KSTEPS];
#pragma unroll
                    for (int s = 0; s < Cfg::KSTEPS; s++) { bv0[s] = bp[4 * s * (int)C]; bv1[s] = bp[GS + 4 * s * (int)C]; }
#pragma unroll
                    for (int s = 0; s < Cfg::KSTEPS; s++) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], bv0[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], bv1[s], acc1, 0, 0, 0);
                    }
                    tp_run = fmaxf(fmaxf(tp_run, fmaxf(fabsf(acc0[0]), fabsf(acc0[1]))), fmaxf(fabsf(acc0[2]), fabsf(acc0[3])));
                    tp_run = fmaxf(fmaxf(tp_run, fmaxf(fabsf(acc1[0]), fabsf(acc1[1]))), fmaxf(fabsf(acc1[2]), fabsf(acc1[3])));
                }
                for (; gi < ngroups; gi++, bp += GS) {                 // odd full group and the masked tail
                    const uint32_t bi = gi * tp_bpg + (uint32_t)mrow / C;
                    const bool col_ok = (gi * 16 + (uint32_t)mrow) < ncol;
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < Cfg::KSTEPS; s++)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], col_ok ? bp[4 * s * (int)C] : 0.0f, acc, 0, 0, 0);
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int row = 4 * kq + reg;
                        const bool ok = col_ok && (bi * Cfg::BLK + (uint32_t)(row % Cfg::BLK)) < seg;
                        tp_run = fmaxf(tp_run, ok ? fabsf(acc[reg]) : 0.0f);
                    }
                }
            } else {
                // channel counts that do not divide 16 (5.1 = 6 channels, 3, 5, 7 ...): channel-major groups — the 16
                // columns of a group are 16 consecutive blocks of ONE channel, so a lane's running maximum belongs to
                // that channel and one LDS atomic per channel and tile closes it (it was one per group)
                const uint32_t gpc = (nblk + 15) >> 4;                  // groups per channel
                for (uint32_t c = 0; c < C; c++) {
                    float m = 0.0f;
                    for (uint32_t gi = 0; gi < gpc; gi++) {
                        const uint32_t bi = gi * 16 + (uint32_t)mrow;
                        const bool col_ok = bi < nblk;
                        const float *bp = tile + ((int)(bi * Cfg::BLK) - (Cfg::HIST - 1) + kq) * (int)C + (int)c;
                        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < Cfg::KSTEPS; s++)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], col_ok ? bp[4 * s * (int)C] : 0.0f, acc, 0, 0, 0);
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int row = 4 * kq + reg;
                            const bool ok = col_ok && (bi * Cfg::BLK + (uint32_t)(row % Cfg::BLK)) < seg;
                            m = fmaxf(m, ok ? fabsf(acc[reg]) : 0.0f);
                        }
                    }
                    atomicMax(&tpk[c], __float_as_uint(m));
                }
            }
        }

        // carry-out: exact state after the last valid sample, broadcast to every lane of the channel
        {
            const uint32_t src_lane = (nchunks - 1) * C + ch;
            cv[0] = __shfl(v1, src_lane, 64); cv[1] = __shfl(v2, src_lane, 64);
            cv[2] = __shfl(v3, src_lane, 64); cv[3] = __shfl(v4, src_lane, 64);
        }
        // ---- sub-block complete: deterministic tree over the lanes' energy shares (fixed shape)
        if (sub_done) {
            if (!warm) {
                double e = e_run;
                for (uint32_t d = 32; d >= 1; d >>= 1) {
                    const double o = __shfl_down(e, d * C, 64);
                    if (lane + d * C < 64u) e += o;
                }
                if (lane < C) p.subblocks[(size_t)stream * p.sub_stride + (size_t)(sb % p.sub_cap) * C + lane] = e;
            }
            e_run = 0.0;
            sb++;
        }
        // ---- new halo: the halo_frames frames before the tile end (a contiguous copy; when the tile
        // is shorter than the halo the source reaches into the old halo).  Ascending order is safe:
        // the source of element j sits seg*C floats above its destination, beyond anything written so far.
        {
            const uint32_t hn = halo_frames * C;
            float *dst = tile - hn;
            const float *srcp = dst + (size_t)seg * C;
            for (uint32_t j = lane; j < hn; j += 64u) {
                const float v = srcp[j];
                __builtin_amdgcn_wave_barrier();
                dst[j] = v;
            }
        }
        pos = npos;
        off = noff;
        seg = nseg_frames;
    }

    // ---- fold this wave's results into the stream state
    // energy of the trailing incomplete sub-block: reduce the lanes' shares (streaming calls carry it over)
    {
        double e = e_run;
        for (uint32_t d = 32; d >= 1; d >>= 1) {
            const double o = __shfl_down(e, d * C, 64);
            if (lane + d * C < 64u) e += o;
        }
        e_run = e;
    }
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_down(sp_run, d * C, 64);
        if (lane + d * C < 64u) sp_run = fmaxf(sp_run, o);
    }
    if (kTpF16) tp_run = fmaxf(tp_run, tp_run16 * (1.0f / 256.0f));
    if (FACTOR != 0 && tp_fixed) atomicMax(&tpk[tp_c], __float_as_uint(tp_run));
    if (lane < C) {
        if (FACTOR != 0) atomicMax(reinterpret_cast<unsigned *>(&st.true_peak[lane]), tpk[lane]);
        atomicMax(reinterpret_cast<unsigned *>(&st.sample_peak[lane]), __float_as_uint(sp_run));
    }
    if (sg + 1 == p.nseg) {                              // the last segment owns the carried filter state
        if (lane_ok && chunk == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) st.v[ch][q] = cv[q];
        }
        if (lane < C) {
            st.acc[lane] = e_run;
            for (int q = 1; q <= kTpHistMax; q++) st.tp_hist[lane][q - 1] = tile[-(int)(q * C) + (int)lane];
        }
        if (lane == 0) st.frames_fed = fed0 + n_frames;
    }
#undef SS_TILE_FRAMES
#undef SS_PREFETCH
}

// Chunk length L: (L-1)*C = 0 (mod 32) makes the per-lane walk through the interleaved tile touch
// lane-linear banks; among the candidates pick the one with the fewest sequential steps per
// sub-block (pieces * L, pieces = tiles a sub-block is cut into).
uint32_t td_chunk_frames(uint32_t C, uint32_t s100)
{
    const uint32_t nch = 64u / C;
    uint32_t best = 33, best_cost = 0xFFFFFFFFu;
    for (uint32_t L : {33u, 49u, 65u}) {
        if (((L - 1) * C) % 32u) continue;
        const uint32_t cap = nch * L;
        const uint32_t pieces = (s100 + cap - 1) / cap;
        const uint32_t cost = pieces * L + 8 * pieces;       // per-tile fixed work ~ 8 steps
        if (cost < best_cost) { best_cost = cost; best = L; }
    }
    return best;
}

// waves of k_time_domain one CU holds at once (LDS per wave grows with the channel count and the decimation halo)
uint32_t td_resident_waves_per_cu(uint32_t C, uint32_t s100, uint32_t halo_frames)
{
    const uint32_t L = td_chunk_frames(C, s100);
    const uint32_t cap = (64u / C) * L;
    const uint32_t pieces = (s100 + cap - 1) / cap;
    uint32_t tile_len = (s100 + pieces - 1) / pieces;
    if (tile_len > cap) tile_len = cap;
    const uint32_t halo = halo_frames ? halo_frames : (uint32_t)kTdHaloFrames;
    uint32_t wave_floats = (halo + tile_len + kTdTailFrames) * C + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    const size_t lds = (size_t)wave_floats * 4 * kTdWavesPerBlock;
    uint32_t blocks = lds ? (uint32_t)((160u * 1024u) / lds) : 4u;
    const uint32_t max_blocks = (4u * SS_TD_WAVES) / kTdWavesPerBlock;      // launch bound: SS_TD_WAVES waves per SIMD
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    return blocks * kTdWavesPerBlock;
}

template <int FACTOR, bool RING, int CT, int WAVE>
static hipError_t td_launch(const TdParams &p, hipStream_t s)
{
    const uint32_t C = p.channels;
    const uint32_t S = p.s100;
    const uint32_t L = td_chunk_frames(C, S);
    const uint32_t nch = 64u / C;
    const uint32_t cap = nch * L;                                   // frames one wave can scan at once
    const uint32_t pieces = (S + cap - 1) / cap;                    // equal tiles per sub-block
    uint32_t tile_len = (S + pieces - 1) / pieces;
    if (tile_len > cap) tile_len = cap;
    // per-wave LDS: halo + tile + slack + 64 peak slots
    const uint32_t halo = WAVE ? p.halo_frames : (uint32_t)kTdHaloFrames;
    uint32_t wave_floats = (halo + tile_len + kTdTailFrames) * C + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    const size_t lds = (size_t)wave_floats * 4 * kTdWavesPerBlock;
    auto fn = k_time_domain<FACTOR, RING, CT, WAVE>;
    static std::atomic<uint64_t> prepared{0};
    if (first_use_on_device(prepared)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { prepared = 0; return e; }
    }
    const uint32_t waves = p.n_streams * p.nseg;
    const uint32_t blocks = (waves + kTdWavesPerBlock - 1) / kTdWavesPerBlock;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * kTdWavesPerBlock), lds, s, p, L, tile_len, wave_floats, halo);
    return hipGetLastError();
}

// Decimation fast path (WAVE = 2): samples per bin spp = len / W is an exact integer multiple of four (<= 128), so
// floor(i spp) / ceil((i+1) spp) are the integer products, and every tile starts on a multiple of four floats.
static int td_wave_int4(const TdParams &p)
{
    const uint64_t len = p.n_frames * p.channels;
    if (!p.wave_window || len % p.wave_window) return 0;
    const uint64_t spp = len / p.wave_window;
    if (spp < 4 || spp > 1000 || (spp & 3u)) return 0;         // the fused path itself stops at 1000 samples per bin
    const uint32_t C = p.channels, S = p.s100;
    const uint32_t L = td_chunk_frames(C, S);
    const uint32_t cap = (64u / C) * L;
    const uint32_t pieces = (S + cap - 1) / cap;
    uint32_t tile_len = (S + pieces - 1) / pieces;
    if (tile_len > cap) tile_len = cap;
    if (!(((uint64_t)S * C) % 4u == 0 && ((uint64_t)tile_len * C) % 4u == 0 && (p.halo_frames * C) % 4u == 0)) return 0;
    return spp <= 128 ? 2 : 3;
}

template <int FACTOR, bool RING>
static hipError_t td_launch_c(const TdParams &p, hipStream_t s)
{
    if (!RING && p.wave_out) {    // fused decimation is a batch feature (never together with the ring)
        if (p.channels == 8) return td_launch<FACTOR, false, 8, 1>(p, s);      // BASELINE config 5
        if (p.channels == 6) return td_launch<FACTOR, false, 6, 1>(p, s);      // 5.1
        if (p.channels == 2) {
            const int fast = td_wave_int4(p);
            if (fast == 2) return td_launch<FACTOR, false, 2, 2>(p, s);
            if (fast == 3) return td_launch<FACTOR, false, 2, 3>(p, s);
            return td_launch<FACTOR, false, 2, 1>(p, s);
        }
        if (p.channels == 1) {                                                   // mono corpora
            const int fast = td_wave_int4(p);
            if (fast == 2) return td_launch<FACTOR, false, 1, 2>(p, s);
            return td_launch<FACTOR, false, 1, 1>(p, s);
        }
        return td_launch<FACTOR, false, 0, 1>(p, s);
    }
    return p.channels == 2 ? td_launch<FACTOR, RING, 2, 0>(p, s) : td_launch<FACTOR, RING, 0, 0>(p, s);
}

hipError_t launch_time_domain(const TdParams &p, hipStream_t s)
{
    if (p.n_streams == 0 || p.n_frames == 0) return hipSuccess;
    const int factor = p.tp_factor;
    const bool ring = p.ring != nullptr;
    switch (factor) {
        case 4: return ring ? td_launch_c<4, true>(p, s) : td_launch_c<4, false>(p, s);
        case 2: return ring ? td_launch_c<2, true>(p, s) : td_launch_c<2, false>(p, s);
        default: return ring ? td_launch_c<0, true>(p, s) : td_launch_c<0, false>(p, s);
    }
}

}  // namespace ssk
