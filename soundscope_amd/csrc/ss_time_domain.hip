// ss_time_domain.hip: K-weighting, sub-block energies, peaks, fused decimation — hand-written gfx950 (CDNA4, wave64) kernels of the soundscope analyzer hot path.
// Reference semantics: /root/reference/src/analyzer.rs (get_fft :55-105, get_waveform :107-137,
// add_samples/getters :139-164, calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as restated in DESIGN.md.
// Nothing here is translated from the reference: the reference has no GPU code.
#include "ss_kernels.h"
#include <atomic>
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
//    starts its filter `warm` sub-blocks (1 = 0.1 s) early from a zero state and
//    discards that run-in: what the missing history would have added to the output is
//    the tail of the K-weighting impulse response — the slowest pole pair (|z| = 0.99502
//    at 48 kHz, e^-240 per second at any rate, a near-double pole) leaves e^-24 (1 + 24)
//    ~ 1e-9 of a DC step after 0.1 s; measured on DC-offset + infrasonic material the
//    sub-block energies stay within 3e-10 of a sequential f64 filter, the arithmetic noise
//    of the recurrence on such material.  Far below the 0.01 dB bar or a 0.1 LU histogram
//    bin (tests pin the latter), but not "to the last bit".  Segment 0 (and every
//    streaming call, nseg = 1) starts from the true carried state.
//  * K-weighting on the f64 VALU.  Each lane owns one (chunk of L frames,
//    channel); the recurrence is cut by  state_out = A^L state_in + zero_state:
//      pass 1: per chunk, run the state recurrence from zero              (4 FMA)
//      scan  : in-wave Hillis-Steele over chunks with the constant matrices
//              (A^L)^(2^k); chunks are dealt round-robin to the four DPP rows so that
//              the long distances are in-row v_mov_dpp shifts, the short ones ds_bpermute
//      pass 2: rerun each chunk from its true initial state, accumulate y^2.
//    L is chosen by td_chunk_frames (below): whole tiles of whole chunks first (48 kHz stereo: L = 30, a
//    sub-block = 5 tiles x 32 chunks), then occupancy and the bank conflicts of the per-lane walk.
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


// Wave priority per phase of the tile loop (s_setprio at the phase marks): the two K-weighting passes 3, the scan 2, the
// true-peak conversion and MFMA loop 1, staging / decimation / tile tail 0.  The sixteen waves of a CU never synchronise,
// so left alone (all at priority 0, oldest first) the four waves of a SIMD drift into the same phase and queue for the
// same unit; graded priorities let a wave inside a dependent f64 chain (one FMA of latency per step) run through while
// the waves in the throughput phases (LDS staging, conversion, MFMA operands) fill the issue slots it leaves: measured
// 1.98-2.01 -> 1.82-1.84 ms at the bench shape (-8 %), profiles/r03_ab_td_wave_priorities.txt; passes alone -3.5 %,
// passes + true peak at one level -5.6 %, the true-peak phases or staging at the top level, or unprioritised: worse.
// phase BEHIND mark: 0 decimation, 1 pass 1, 2 scan, 3 pass 2, 4 true-peak conversion, 5 MFMA loop, 6 tile tail, 7 staging
// (the builtin wants a literal: two priority bits per phase, packed)
#define SS_TD_PHASE_PRIORITY(mark) __builtin_amdgcn_s_setprio((0x05ECu >> (2 * (mark))) & 3u)      // {0, 3, 2, 3, 1, 1, 0, 0}

// Development build (-DSS_TD_PROF): per-phase shader-clock totals of k_time_domain, summed over all waves
// (s_memtime at the phase boundaries of the tile loop; read back with ss_debug_td_prof).  Not in release builds.
#ifdef SS_TD_PROF
__device__ unsigned long long g_td_prof[16];
#define SS_PROF_DECL uint64_t pt_ = __builtin_amdgcn_s_memtime(); uint64_t pacc_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SS_PROF_MARK(i) do { const uint64_t n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; SS_TD_PHASE_PRIORITY(i); } while (0)
#define SS_PROF_END do { if (lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 9; i_++) atomicAdd(&g_td_prof[i_], (unsigned long long)pacc_[i_]); atomicAdd(&g_td_prof[15], 1ull); } } while (0)
#else
#define SS_PROF_DECL
#define SS_PROF_MARK(i) SS_TD_PHASE_PRIORITY(i)
#define SS_PROF_END
#endif

template <int FACTOR>
struct TpCfg {
    static constexpr int HIST = (FACTOR == 2) ? 24 : 12;     // taps per polyphase branch
    static constexpr int NPH = (FACTOR == 4) ? 3 : (FACTOR == 2 ? 1 : 0);
    static constexpr int BLK = (FACTOR == 4) ? 5 : 16;       // outputs per column
    static constexpr int ROWS = NPH * BLK;                   // 15 or 16
    static constexpr int KSTEPS = (BLK + HIST - 1 + 3) / 4;  // 4 or 10
};

// (the matrix pointer is in the CONSTANT address space: the tables are written once by the host before any launch, and a
// uniform constant-space address makes the sixteen loads scalar (s_load into SGPRs, K$) instead of per-lane flat loads)
typedef const __attribute__((address_space(4))) double *const_f64_ptr;
__device__ __forceinline__ void mat4_apply_add(const_f64_ptr M, const double (&x)[4], double (&z)[4])
{
#pragma unroll
    for (int r = 0; r < 4; r++)
        z[r] = fma(M[r * 4 + 0], x[0], fma(M[r * 4 + 1], x[1], fma(M[r * 4 + 2], x[2], fma(M[r * 4 + 3], x[3], z[r]))));
}

// w = D s, D = [(-1)^j C(i,j)]: the backward differences of the DF-II state (D is its own inverse).  The chunk scan
// works on w (ss_tables.cpp, kweight_transition_pow): the subtractions are exact for the slowly varying states that make
// the plain product A^n s cancel, and six of them replace nothing else.
__device__ __forceinline__ void state_diff(double (&s)[4])
{
    const double d12 = s[0] - s[1], d23 = s[1] - s[2], d34 = s[2] - s[3];
    const double e1 = d12 - d23, e2 = d23 - d34;
    s[1] = d12; s[2] = e1; s[3] = e1 - e2;
}
__device__ __forceinline__ void state_undiff(double &v1, double &v2, double &v3, double &v4)
{
    // (v1, d12, e1, f) -> (v1, v2, v3, v4)
    const double d12 = v2, e1 = v3, f = v4;
    const double d23 = d12 - e1, e2 = e1 - f, d34 = d23 - e2;
    v2 = v1 - d12; v3 = v2 - d23; v4 = v3 - d34;
}

constexpr int kTdHaloFrames = 24;     // minimum halo: >= HIST-1 of the longest branch (multiple of 4: the tile stays 16-B aligned)
constexpr int kTdTailFrames = 16;     // zeroed slack past the tile end: K-weighting look-ahead and the last f32 MFMA window
// floats of slack behind a wave's tile: the zeroed frames above, or — larger — room for the planar f16 true-peak layout
// (12 frames of history + the last 256-byte block, which may run past the tile); kept tight: at 8 channels 64 more
// floats per wave would cost a quarter of the resident waves
__host__ __device__ constexpr uint32_t td_slack_floats(uint32_t C) { return (12u * C + 64u) > (16u * C) ? (12u * C + 64u) : (16u * C); }
constexpr int kTdWavesPerBlock = 4;
#ifndef SS_TD_PREFETCH
#define SS_TD_PREFETCH 8
#endif
constexpr int kTdPrefetch = SS_TD_PREFETCH;        // float4 per lane held in flight for the next tile
#ifndef SS_TD_BATCH
#define SS_TD_BATCH 10
#endif
constexpr int kTdBatch = SS_TD_BATCH;          // LDS reads issued together in the sequential passes

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

// maximum over the wave of a non-negative float, as its bit pattern in an SGPR (non-negative floats order like
// unsigned integers): four DPP row rotations and three scalar maxima — no LDS crossbar traffic
__device__ __forceinline__ uint32_t wave_max_nonneg_bits(float v)
{
#define SS_ROW_ROR_(x, n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 | (n_), 0xF, 0xF, false))
    v = fmaxf(v, SS_ROW_ROR_(v, 8));
    v = fmaxf(v, SS_ROW_ROR_(v, 4));
    v = fmaxf(v, SS_ROW_ROR_(v, 2));
    v = fmaxf(v, SS_ROW_ROR_(v, 1));
#undef SS_ROW_ROR_
    const int b = __float_as_int(v);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane(b, 0), r1 = (uint32_t)__builtin_amdgcn_readlane(b, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane(b, 32), r3 = (uint32_t)__builtin_amdgcn_readlane(b, 48);
    const uint32_t m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
    return m01 > m23 ? m01 : m23;
}

// CT: compile-time channel count (0 = runtime)
// WAVE: 0 no decimation, 1 fused get_waveform (any bin geometry), 2 the same for an exact-integer samples-per-bin that is
// a multiple of four (<= 128) with 16-byte aligned tiles (the host checks), 3 the same for 128 < spp <= 1000
// WPS: waves per SIMD the instantiation is register-allocated for.  4 (128 VGPRs, 30-120 bytes of scratch per lane) is what a
// grid of >= 4096 waves needs; a grid that fits the chip at three waves per SIMD (BASELINE config 5: 64 streams x 34
// segments = 2176 waves) runs the 3-wave build instead — up to 168 VGPRs, nothing spilled: 1.63-1.72 -> 1.58-1.60 ms there
// (the same build on the 4096-wave bench grid: 1.85 -> 2.24 ms, it needs a second round of waves).
// SPLIT (streaming calls, nseg == 1: the handle's add_samples and the session ticks): the four waves of a workgroup share
// ONE stream's call, wave w taking tiles w, w + 4, w + 8, ...  A tile's staging, its zero-state pass and — behind the
// hand-over — its true-peak product run beside the other waves' tiles; what stays in sequence is what the recurrence makes
// sequential: the carried filter state (and the lanes' running energy shares) pass from tile to tile through TdShare, the
// wave of tile i waiting for tile i - 1 in front of its scan and publishing behind its second pass.  Same arithmetic per
// tile, same order of the energy sums; the tick's 8192-frame refeed (nine tiles) no longer walks them one after another.
struct TdShare {
    uint32_t tiles_done;                 // tiles whose energy shares stand in `e_lane` (release / acquire, workgroup scope)
    uint32_t state_ready;                // tiles whose state stands in `carry`: behind the scan for a tile of whole chunks (the scan
                                         // leaves the tile's end state in its last chunk's lanes), behind the second pass otherwise
    double carry[kMaxChannels][4];       // DF-II state behind the last published tile
    double e_lane[64];                   // lane (chunk, channel)'s share of the current sub-block's energy
};

template <int FACTOR, bool RING, int CT, int WAVE, int WPS, bool SPLIT = false>
__global__ __launch_bounds__(64 * kTdWavesPerBlock, WPS) void k_time_domain(TdParams p, uint32_t L, uint32_t tile_len,
                                                                                      uint32_t wave_lds_floats, uint32_t halo_frames)
{
    using Cfg = TpCfg<FACTOR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    // wave-uniform by construction: tell the compiler, so that everything derived from it (stream, segment, tile
    // geometry, loop bounds) lives in SGPRs and branches on the scalar unit instead of through exec masks
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t gw = SPLIT ? blockIdx.x : blockIdx.x * kTdWavesPerBlock + wave_in_block;   // global wave = (stream, segment); SPLIT: stream
    if (gw >= p.n_streams * p.nseg) return;                                // whole wave leaves (no barriers used) — SPLIT: the whole workgroup
    const uint32_t stream = gw / p.nseg, sg = gw - stream * p.nseg;
    TdShare *const sh = reinterpret_cast<TdShare *>(smem + (size_t)kTdWavesPerBlock * wave_lds_floats * sizeof(float));

    float *tilebuf = reinterpret_cast<float *>(smem) + (size_t)wave_in_block * wave_lds_floats;
    const TdConst &K = *p.k;
    const uint32_t C = CT ? (uint32_t)CT : p.channels;
    const uint32_t S = p.s100;
    const uint32_t nch = 64u / C;                       // chunks per tile (C <= 64)
    // Lane -> (chunk, channel).  Natural: chunk = lane / C.  Channel counts that divide 16 (compile-time) deal the chunks
    // round-robin to the four DPP rows of the wave instead: row r = lane / 16 holds chunks r, r + 4, r + 8, ..., so the scan
    // steps of distance 4, 8, 16 ... are shifts INSIDE a row (v_mov_dpp row_shr, VALU rate) and only the distances 1 and 2
    // cross rows (ds_bpermute: 22 cycles of the LDS crossbar each, and a dependent latency per step).  The walk through the
    // interleaved tile is bank-conflict free for odd L ((r + 4 q) L C + c covers 64 distinct banks) and 2-way conflicted for
    // L = 30, which td_chunk_frames still prefers where it cuts the sub-block into whole tiles of whole chunks.
    constexpr bool kRowScan = (CT != 0) && (16 % (CT ? CT : 1) == 0);
    const uint32_t lane_q = (lane & 15u) / C;           // position of this lane's chunk inside its row
    const uint32_t chunk = kRowScan ? (lane >> 4) + 4u * lane_q : lane / C;
    const uint32_t ch = kRowScan ? (lane & 15u) - lane_q * C : lane - chunk * C;
    const bool lane_ok = chunk < nch;
    // lane holding chunk c of this lane's channel
    auto lane_of_chunk = [&](uint32_t c) -> uint32_t { return kRowScan ? ((c & 3u) << 4) + (c >> 2) * C + ch : c * C + ch; };
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
    if (SPLIT) {
        if (wave_in_block == 0) {
            if (lane == 0) { sh->tiles_done = 0u; sh->state_ready = 0u; }
            sh->e_lane[lane] = e_run;                    // (st.acc in the lanes of chunk 0 ... as e_run carries it)
            if (lane_ok && chunk == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) sh->carry[ch][q] = cv[q];
            }
        }
        __syncthreads();
    }

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
    // Planar f16-split form of the same product (factor 4, channel counts that divide 16): after the K-weighting
    // passes of a tile every sample is converted ONCE to an f16 pair, s x = x_hi + x_lo with s a power of two taken
    // from the tile's own sample peak (so accuracy does not depend on level and nothing over- or underflows), and
    // stored IN PLACE over the f32 tile (dead by then; the next tile's halo is copied out first) in 256-byte blocks
    // of FB = 64 / C frames: [hi c = 0 .. C-1][lo c = 0 .. C-1], FB halves each, block k holding converted frames
    // [k FB - 12, (k + 1) FB - 12).  A block occupies exactly the bytes of FB source frames, so the conversion walks
    // the blocks from the top down and never overwrites a sample it still has to read.
    // Column (channel, block of 4 outputs b) then reads its 16-sample window [4 b - 12, 4 b + 4) as aligned 8-byte
    // pieces of the hi and the lo plane, and D = A_hi B_hi + A_hi B_lo + A_lo B_hi runs as three
    // v_mfma_f32_16x16x16_f16 with rows (phase f, output r), A[(f,r)][k] = c_f[12 + r - k]; the dropped lo*lo term and
    // the split remainders are < 2^-21 of the tile's peak.  One group of 16 columns advances every window by exactly one
    // block, so the read addresses of a lane just step by 256 bytes per group.  Unlike the f32 MFMA this runs beside
    // other waves' f64 VALU work (tools/ubench4.hip), and the per-window split that used to cost 18 VALU
    // lane-instructions per sample is about 4.
    // (gfx950's K = 32 form, v_mfma_f32_16x16x32_f16 — two products instead of three — is NOT used: a wave issuing it
    // corrupts v_pk_add_f32 results of OTHER waves on the same SIMD, i.e. of the spectrum kernel when the two kernels
    // share the chip; isolated by tools/ubench_k32_interference.hip, profiles/r03_ubench_k32_interference.txt)
    constexpr bool kTpPlanar = (FACTOR == 4);
    halfx4 a16_lo = {0, 0, 0, 0};                       // lane (mrow, kq): a_lo[4 kq + j], j = 0..3
    halfx4 a16_hi = {0, 0, 0, 0};                       //                  a_hi[4 kq + j]
    if (kTpPlanar) {
        const int fph = mrow >> 2, r = mrow & 3;       // rows 0..11 = (phase, output); rows 12..15 are zero
        auto tap = [&](int k) -> float {               // A[(f, r)][k] = c_f[12 + r - k]
            const int t = 12 + r - k;
            return (mrow < 12 && t >= 0 && t < 12) ? K.tp[fph][t] : 0.0f;
        };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float c = tap(4 * kq + j);
            a16_lo[j] = (_Float16)(c - (float)(_Float16)c);
            a16_hi[j] = (_Float16)c;
        }
    }
    // sample peak (as bits of a non-negative float) of the 12 frames in front of the current tile: the FIR window
    // reaches them, so they take part in the choice of the tile's scale.  A streaming call starts from carried history.
    uint32_t tp_prev_bits = 0;
    if (kTpPlanar && carry_in) {
        float h = 0.0f;
        if (lane < C)
            for (int q = 1; q <= 12; q++) h = fmaxf(h, fabsf(tile[-(int)(q * C) + (int)lane]));
        tp_prev_bits = wave_max_nonneg_bits(h);
    }
    const double a1 = K.a[1], a2 = K.a[2], a3 = K.a[3], a4 = K.a[4];
    const double b0 = K.b[0], b1 = K.b[1], b2 = K.b[2], b3 = K.b[3], b4 = K.b[4];

    // tile geometry: a tile never crosses a sub-block boundary of the absolute grid; a sub-block is
    // cut into equal pieces of at most tile_len frames
    // (toff = off % tile_len is carried along instead of being divided out for every tile)
#define SS_TILE_FRAMES(at, off_in, toff_in, out)                            \
    do {                                                                    \
        uint64_t n_ = 0;                                                    \
        if ((at) < seg_end) {                                               \
            n_ = tile_len - (toff_in);                                      \
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
    uint32_t toff = off % tile_len;                     // position inside the current piece of the sub-block
    uint32_t slot = (uint32_t)(sb % p.sub_cap);         // where the current sub-block's energies go (ring of sub_cap)
    SS_TILE_FRAMES(pos, off, toff, seg);
    SS_PREFETCH(pos, seg);
    SS_PROF_DECL

    // ---- fold this wave's results into the stream state (a lambda: a wave without a single tile — an empty segment of a
    // ragged batch — leaves through it in front of the tile loop, so that the loop itself runs at least once and the compiler
    // keeps no spare copy of the initial state for a zero-trip path)
    auto fold_results = [&](const uint32_t ti) {
    // SPLIT: the carried state (filter, energy shares, history) is the last tile's; its wave writes it, the others only their peaks
    const bool state_owner = !SPLIT || (ti != 0u && ((ti - 1u) & (uint32_t)(kTdWavesPerBlock - 1)) == wave_in_block) || (ti == 0u && wave_in_block == 0u);
    if (SPLIT && state_owner && ti != 0u) e_run = sh->e_lane[lane];
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
    if (FACTOR != 0 && tp_fixed) {
        uint32_t lane_f = lane;                          // (rebuilt here: see the rare paths of the true-peak product)
        asm volatile("" : "+v"(lane_f));
        atomicMax(&tpk[(lane_f & 15u) % C], __float_as_uint(tp_run));
    }
    {
        uint32_t lane_p = lane;
        asm volatile("" : "+v"(lane_p));
        if (lane_p < C) {
            if (FACTOR != 0) atomicMax(reinterpret_cast<unsigned *>(&st.true_peak[lane_p]), tpk[lane_p]);
            atomicMax(reinterpret_cast<unsigned *>(&st.sample_peak[lane_p]), __float_as_uint(sp_run));
        }
    }
    if (sg + 1 == p.nseg && state_owner) {               // the last segment owns the carried filter state
        if (lane_ok && chunk == 0) {             // ... flushed like at the end of every add_frames call (see the sub-block boundary above)
#pragma unroll
            for (int q = 0; q < 4; q++) st.v[ch][q] = fabs(cv[q]) < 2.2250738585072014e-308 ? 0.0 : cv[q];
        }
        if (lane < C) {
            st.acc[lane] = e_run;
            for (int q = 1; q <= kTpHistMax; q++) st.tp_hist[lane][q - 1] = tile[-(int)(q * C) + (int)lane];
        }
        if (lane == 0) st.frames_fed = fed0 + n_frames;
    }
    };
    if (seg == 0) { fold_results(0u); return; }
    uint32_t ti = 0;                                    // index of the tile within the call (SPLIT: whose turn it is)
    do {
        // the tile behind this one
        const uint64_t npos = pos + seg;
        uint32_t noff = off + seg, ntoff = toff + seg;
        const bool sub_done = (noff == S);
        if (sub_done) noff = 0;
        if (sub_done || ntoff >= tile_len) ntoff = 0;
        uint32_t nseg_frames;
        SS_TILE_FRAMES(npos, noff, ntoff, nseg_frames);
        if (!SPLIT || (ti & (uint32_t)(kTdWavesPerBlock - 1)) == wave_in_block) {
        // keep the scan matrices in memory (scalar loads at the point of use): hoisting all of them
        // out of the tile loop would cost 224 SGPRs
        const_f64_ptr mpow = (const_f64_ptr)(uintptr_t)&K.m_pow[0][0];
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
                // no predicate: a lane past the tile's end holds a copy of the last float4 (clamped prefetch index) and writes
                // it where it belongs — the same value to the same place, without exec-mask branches
                const uint32_t last = nv ? nv - 1u : 0u;
                if (!SPLIT) {
#pragma unroll
                    for (int q = 0; q < kTdPrefetch; q++) {
                        const uint32_t i = lane + 64u * q;
                        if (nv) t4[i < last ? i : last] = pf[q];           // (nv is wave-uniform)
                    }
                }
                const float4 *g4 = reinterpret_cast<const float4 *>(g);
                uint32_t lane_s = lane;
                asm volatile("" : "+v"(lane_s));
                for (uint32_t i = lane_s + 64u * (SPLIT ? 0 : kTdPrefetch); i < nv; i += 64u) t4[i] = g4[i];   // SPLIT: no register prefetch (the other waves' tiles lie between)
                done = nv << 2;
            }
            for (uint32_t i = done + lane; i < total; i += 64u) tile[i] = g[i];
            for (uint32_t i = total + lane; i < total + kTdTailFrames * C; i += 64u) tile[i] = 0.0f;
        }
        // next tile's loads fly while this one is processed
        if (!SPLIT) SS_PREFETCH(npos, nseg_frames);
        if (SPLIT) {
            // the halo of a tile another wave left: the frames in front of it come from the call's input, or — in front of
            // the call — from the carried history (tp_hist[c][k] = frame -1 - k of the call)
            for (uint32_t j = lane; j < halo_frames * C; j += 64u) {
                const uint32_t q = j / C + 1u, c = j - (q - 1u) * C;
                float v = 0.0f;
                if (pos >= q) v = src[(pos - q) * C + c];
                else if (q - (uint32_t)pos <= (uint32_t)kTpHistMax) v = st.tp_hist[c][q - (uint32_t)pos - 1u];
                tile[-(int)(q * C) + (int)c] = v;
            }
            if (kTpPlanar) {                            // sample peak of the twelve frames the first FIR windows reach back to
                __builtin_amdgcn_wave_barrier();
                float h = 0.0f;
                if (lane < C)
                    for (int q = 1; q <= 12; q++) h = fmaxf(h, fabsf(tile[-(int)(q * C) + (int)lane]));
                tp_prev_bits = wave_max_nonneg_bits(h);
            }
        }
        __builtin_amdgcn_wave_barrier();               // LDS is in-order per wave: only ordering is needed
        SS_PROF_MARK(0);

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
                        // three reads cover spp <= 96 (48 kHz stereo), a fourth spp <= 128; clamped indices repeat an element,
                        // which cannot change a min / max.  All reads are issued before the arithmetic.
                        float4 v[3];
#pragma unroll
                        for (int it = 0; it < 3; it++) {
                            uint32_t j = lane8 + 8u * it;
                            j = j < n4 ? j : n4 - 1;
                            v[it] = bp4[j];
                        }
#if defined(__HIP_DEVICE_COMPILE__)
                        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                        for (int it = 0; it < 3; it++) {
                            mn = fminf(fminf(mn, v[it].x), fminf(fminf(v[it].y, v[it].z), v[it].w));
                            mx = fmaxf(fmaxf(mx, v[it].x), fmaxf(fmaxf(v[it].y, v[it].z), v[it].w));
                        }
                        if (n4 > 24u) {                                                // wave-uniform
                            const float4 w = bp4[lane8 + 24u < n4 ? lane8 + 24u : n4 - 1];
                            mn = fminf(fminf(mn, w.x), fminf(fminf(w.y, w.z), w.w));
                            mx = fmaxf(fmaxf(mx, w.x), fmaxf(fmaxf(w.y, w.z), w.w));
                        }
                        if (WAVE == 3)                                                 // longer bins (192 at 96 kHz stereo)
                            for (uint32_t j = lane8 + 32u; j < n4; j += 8u) {
                                const float4 w = bp4[j];
                                mn = fminf(fminf(mn, w.x), fminf(fminf(w.y, w.z), w.w));
                                mx = fmaxf(fmaxf(mx, w.x), fmaxf(fmaxf(w.y, w.z), w.w));
                            }
                    }
                    // 8-lane all-reduce: xor 1, xor 2 (quad_perm), then the mirror inside each half row.  DPP on the
                    // operand of v_min / v_max itself (IEEE minNum / maxNum: a NaN partner is ignored, an all-NaN bin stays NaN);
                    // written out because the builtin route costs four instructions per step and value.  The s_nop keep the
                    // two wait states a DPP read needs after a VALU write of the same register.
                    asm volatile("s_nop 1\n\t"
                                 "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                 "s_nop 0\n\t"
                                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "s_nop 0\n\t"
                                 "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                                 "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf"
                                 : "+v"(mn), "+v"(mx));
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

        SS_PROF_MARK(1);
        const bool active = lane_ok && chunk < nchunks;
        const uint32_t len = active ? ((seg - chunk * L) < L ? (seg - chunk * L) : L) : 0u;
        const float *xs = tile + (size_t)chunk * L * C + ch;
        const uint32_t nb_full = L / kTdBatch;          // whole batches in a full chunk
        // (Stereo reading whole frames as ds_read_b64 + select in the two passes below — mod-64 banks, two deep instead of the
        // four of ds_read_b32's mod-32 banks at this chunk stride — was measured: 1.99 -> 2.13 ms,
        // profiles/r03_ab_ms1_td_variants.txt; the select and the unmergeable reads cost more than the conflicts.  And the
        // conflicts cost little: with both passes walking the tile lane-linear (conflict-free, wrong results, timing only)
        // the kernel went from 1.946 to 1.918 ms, profiles/r03_ab_td_conflict_free_passes.txt.)

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
                if (i < L) {                            // chunk lengths that are no multiple of the batch (L = 49 at 44.1 kHz):
                    const uint32_t rem = L - i;         // one more batch of reads (into the next chunk or the zeroed slack), of
                    float xb[kTdBatch];                 // which the first `rem` (wave-uniform) are consumed
#pragma unroll
                    for (int u = 0; u < kTdBatch; u++) xb[u] = xp[u * (int)C];
#pragma unroll
                    for (int u = 0; u < kTdBatch - 1; u++)
                        if ((uint32_t)u < rem) { SS_KW_LA_STEP((double)xb[u]) SS_KW_SHIFT() }
                    i = L;
                }
            }
            for (; i < len; i++) { SS_KW_STATE((double)xs[i * C]) SS_KW_SHIFT() }
            z[0] = v1; z[1] = v2; z[2] = v3; z[3] = v4;
            state_diff(z);                              // the scan runs in difference coordinates
            if (SPLIT) {
                // the state in front of this tile: published by the wave of the tile before it
                while (__hip_atomic_load(&sh->state_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < ti) __builtin_amdgcn_s_sleep(1);
                if (lane_ok) {
#pragma unroll
                    for (int q = 0; q < 4; q++) cv[q] = sh->carry[ch][q];
                }
            }
            if (active && chunk == 0) {
                double cw[4] = {cv[0], cv[1], cv[2], cv[3]};
                state_diff(cw);
                mat4_apply_add(mpow, cw, z);
            }
        }
        SS_PROF_MARK(2);
        // ---- in-wave scan over chunks: z_i += (A^L)^(2^k) z_{i - 2^k}  (the steps commute: powers of one matrix)
        if (kRowScan) {
            // distances 1 and 2: the source chunk sits in another row
#pragma unroll
            for (int kstep = 0; kstep < 2; kstep++) {
                const uint32_t d = 1u << kstep;
                if (d >= nchunks) break;                // wave-uniform
                const int src = (int)lane_of_chunk(chunk >= d ? chunk - d : 0u);
                const double xin[4] = {__shfl(z[0], src, 64), __shfl(z[1], src, 64), __shfl(z[2], src, 64), __shfl(z[3], src, 64)};
                if (active && chunk >= d) mat4_apply_add(mpow + 16 * kstep, xin, z);
            }
            // distances 4, 8, 16, 32: q - 1, q - 2, q - 4, q - 8 inside the row; lanes without a source read zeros (bound_ctrl)
#define SS_DPP_SHR64(v, N)                                                                                              \
    __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + (N), 0xF, 0xF, true),                      \
                     __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + (N), 0xF, 0xF, true))
#define SS_ROW_STEP(N, KSTEP)                                                                                           \
    if ((4u << ((KSTEP) - 2)) < nchunks) {                                                                                 \
        const double xin[4] = {SS_DPP_SHR64(z[0], N), SS_DPP_SHR64(z[1], N), SS_DPP_SHR64(z[2], N), SS_DPP_SHR64(z[3], N)}; \
        mat4_apply_add(mpow + 16 * (KSTEP), xin, z);                                                                    \
    }
            constexpr int CC = CT ? CT : 1;
            if (CC * 1 <= 8) { SS_ROW_STEP(CC * 1 <= 8 ? CC * 1 : 1, 2) }
            if (CC * 2 <= 8) { SS_ROW_STEP(CC * 2 <= 8 ? CC * 2 : 1, 3) }
            if (CC * 4 <= 8) { SS_ROW_STEP(CC * 4 <= 8 ? CC * 4 : 1, 4) }
            if (CC * 8 <= 8) { SS_ROW_STEP(CC * 8 <= 8 ? CC * 8 : 1, 5) }
#undef SS_ROW_STEP
#undef SS_DPP_SHR64
        } else {
            for (int kstep = 0; (1u << kstep) < nchunks; kstep++) {
                const uint32_t d = (1u << kstep) * C;
                const double xin[4] = {__shfl_up(z[0], d, 64), __shfl_up(z[1], d, 64), __shfl_up(z[2], d, 64), __shfl_up(z[3], d, 64)};
                if (active && lane >= d) mat4_apply_add(mpow + 16 * kstep, xin, z);
            }
        }
        // z = state after this lane's chunk (valid for full chunks); initial state = previous chunk's
        double v1, v2, v3, v4;
        {
            const int src = (int)lane_of_chunk(chunk ? chunk - 1u : 0u);
            const double p0 = __shfl(z[0], src, 64), p1 = __shfl(z[1], src, 64), p2 = __shfl(z[2], src, 64), p3 = __shfl(z[3], src, 64);
            const bool first = chunk == 0;
            double q1 = p0, q2 = p1, q3 = p2, q4 = p3;
            state_undiff(q1, q2, q3, q4);               // back to (v1 .. v4)
            v1 = first ? cv[0] : q1; v2 = first ? cv[1] : q2; v3 = first ? cv[2] : q3; v4 = first ? cv[3] : q4;
        }
        // SPLIT, a tile of whole chunks: its end state stands in the last chunk's lanes right here — the next tile's wave gets it
        // a whole second pass earlier (flushed like every carry across a gating-block boundary, see tile_carry_out).  Not in the
        // tail of a decay: near the sub-normal range the scan's difference coordinates lose digits the plain recurrence keeps
        // (the crate's flush decides on the state's last bits there) — such a tile hands over behind its second pass.
        bool early_state = false;
        if (SPLIT && seg == nchunks * L) {
            double s1 = z[0], s2 = z[1], s3 = z[2], s4 = z[3];
            state_undiff(s1, s2, s3, s4);
            const double sv[4] = {s1, s2, s3, s4};
            const bool last = lane_ok && chunk == nchunks - 1u;
            bool tiny = false;
#pragma unroll
            for (int q = 0; q < 4; q++) tiny = tiny || (sv[q] != 0.0 && fabs(sv[q]) < 1e-200);
            early_state = __ballot(last && tiny) == 0ull;
            if (early_state) {
                if (last) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        sh->carry[ch][q] = (sub_done && sb + 1 >= 4 && fabs(sv[q]) < 2.2250738585072014e-308) ? 0.0 : sv[q];
                }
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) __hip_atomic_store(&sh->state_ready, ti + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        SS_PROF_MARK(3);
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
                if (i < L) {
                    const uint32_t rem = L - i;
                    float xb[kTdBatch];
#pragma unroll
                    for (int u = 0; u < kTdBatch; u++) xb[u] = xp[u * (int)C];
#pragma unroll
                    for (int u = 0; u < kTdBatch - 1; u++)
                        if ((uint32_t)u < rem) {
                            // (x[i + 3] of the last three steps belongs to the next chunk: harmless in a maximum, see above)
                            sp = fmaxf(sp, fabsf(xb[u]));
                            SS_KW_LA_STEP((double)xb[u]) SS_KW_LA_OUT() SS_KW_SHIFT()
                            e = fma(y_, y_, e);
                            if (RING) p.ring[((ring_base + i + u) % p.ring_frames) * C + ch] = y_;
                        }
                    i = L;
                }
            }
            for (; i < len; i++) {
                const float xf = xs[i * C];
                sp = fmaxf(sp, fabsf(xf));
                SS_KW_STATE((double)xf) SS_KW_OUT() SS_KW_SHIFT()
                e = fma(y_, y_, e);
                if (RING) p.ring[((ring_base + i) % p.ring_frames) * C + ch] = y_;
            }
            if (SPLIT) {                                // the lanes' shares travel from tile to tile (same sums, same order)
                while (__hip_atomic_load(&sh->tiles_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < ti) __builtin_amdgcn_s_sleep(1);
                e_run = sh->e_lane[lane];
            }
            if (!warm) { e_run += e; sp_run = fmaxf(sp_run, sp); }
        }
        // carry-out: exact state after the last valid sample, to the lanes of the channel (chunk 0's lane consumes it), and the
        // close of a sub-block.  Runs at the end of the tile — or, SPLIT, right behind the second pass, where the next tile's
        // wave is waiting for it.
        auto tile_carry_out = [&]() {
        if (kRowScan && CT <= 2) {
            // the source lanes are wave-uniform: v_readlane per channel instead of eight trips through the LDS crossbar
            const uint32_t lastc = nchunks - 1u;
            const uint32_t l0 = ((lastc & 3u) << 4) + (lastc >> 2) * C;     // lane of (last chunk, channel 0)
            auto rl64 = [](double v, uint32_t src) -> double {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (int)src), __builtin_amdgcn_readlane(__double2loint(v), (int)src));
            };
            const double vv[4] = {v1, v2, v3, v4};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double a = rl64(vv[q], l0);
                cv[q] = (CT == 2 && ch == 1u) ? rl64(vv[q], l0 + 1u) : a;
            }
        } else {
            const uint32_t src_lane = lane_of_chunk(nchunks - 1);
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
                uint32_t lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                if (lane_e < C) p.subblocks[(size_t)stream * p.sub_stride + (size_t)slot * C + lane_e] = e;
            }
            e_run = 0.0;
            // ebur128 flushes sub-normal filter state to zero at the end of every internal filter call (restated at
            // oracle/ss_oracle.c:570-571).  add_frames cuts its input where a gating block completes: at the fourth
            // 100 ms boundary after a reset and at every boundary after it (needed_frames = 4 s100, then s100).
            if (sb + 1 >= 4) {
#pragma unroll
                for (int q = 0; q < 4; q++) cv[q] = fabs(cv[q]) < 2.2250738585072014e-308 ? 0.0 : cv[q];
            }
        }
        if (SPLIT) {                                    // hand over: energy shares — and the state, unless the scan has published it
            sh->e_lane[lane] = e_run;
            if (!early_state && lane_ok && chunk == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) sh->carry[ch][q] = cv[q];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                if (!early_state) __hip_atomic_store(&sh->state_ready, ti + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&sh->tiles_done, ti + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        };
        if (SPLIT) tile_carry_out();
        SS_PROF_MARK(4);
        // ---- true peak on the matrix pipe (not during the run-in)
        bool halo_done = false;
        const uint32_t tp_now_bits = kTpPlanar ? wave_max_nonneg_bits(sp) : 0u;      // this tile's sample peak, wave-uniform
        if (FACTOR != 0 && !warm) {
            // f32 product (bit-exact fmaf chain) over `nfr` frames starting at float offset `base_f` of the tile
            auto tp_f32_range = [&](uint32_t first_frame, uint32_t nfr) {
                const float *t0 = tile + (size_t)first_frame * C;
                const uint32_t nblk = (nfr + Cfg::BLK - 1) / Cfg::BLK;     // blocks per channel
                const uint32_t ncol = nblk * C;
                const uint32_t ngroups = (ncol + 15) >> 4;
                constexpr int GS = 16 * Cfg::BLK;                          // floats per group (16 columns x BLK outputs)
                const uint32_t nfull = nfr / (tp_bpg * Cfg::BLK);          // groups whose every output lies inside the range
                uint32_t gi = 0;
                const float *bp = t0 + tp_lane_off;
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
                // (rare paths below rebuild what they need from an opaque copy of the lane id: hoisted out of the tile loop their
                // lane-dependent constants were what the four-waves-per-SIMD build kept in scratch)
                for (; gi < ngroups; gi++, bp += GS) {                     // odd full group and the masked tail
                    uint32_t lane_t = lane;
                    asm volatile("" : "+v"(lane_t));
                    const int kq = (int)(lane_t >> 4), mrow = (int)(lane_t & 15u);
                    const uint32_t bi = gi * tp_bpg + (uint32_t)mrow / C;
                    const bool col_ok = (gi * 16 + (uint32_t)mrow) < ncol;
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < Cfg::KSTEPS; s++)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], col_ok ? bp[4 * s * (int)C] : 0.0f, acc, 0, 0, 0);
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int row = 4 * kq + reg;
                        const bool ok = col_ok && (bi * Cfg::BLK + (uint32_t)(row % Cfg::BLK)) < nfr;
                        tp_run = fmaxf(tp_run, ok ? fabsf(acc[reg]) : 0.0f);
                    }
                }
            };
            if (tp_fixed) {
                // planar f16 product for the whole groups of the tile, f32 product for what is left (and for tiles
                // whose peak, or whose predecessor's, is not finite)
                uint32_t nplanar = 0;                                      // groups (16 columns x 4 outputs) on the f16 path
                uint32_t scale_bits = 0, inv_bits = 0;
                if (kTpPlanar) {
                    const uint32_t pkb = tp_now_bits > tp_prev_bits ? tp_now_bits : tp_prev_bits;
                    const uint32_t e = pkb >> 23;                          // biased exponent of the peak (sign is 0)
                    nplanar = (seg >> 2) / tp_bpg;
                    if (e == 255u || C > 8u || p.tp_f32) nplanar = 0;      // (16 channels: round 0 would not hold the 12 history frames)
                    // scale = 2^(14 - floor(log2 peak)): the scaled peak lies in [2^14, 2^15), inside the f16 range
                    uint32_t sf = 268u - e;                                // biased exponent of the scale
                    sf = sf > 254u ? 254u : sf;
                    scale_bits = sf << 23;
                    inv_bits = (254u - sf) << 23;                          // exact reciprocal (0 when the peak is below 2^-113)
                }
                const uint32_t f0 = nplanar * tp_bpg * 4;                  // frames covered by the planar path
                if (f0 < seg) tp_f32_range(f0, seg - f0);
                if (nplanar) {
                    const float scale = __uint_as_float(scale_bits);
                    const uint32_t ps = f0 + 12;                           // converted frames: [-12, f0), even
                    const uint32_t FB = 64u / C, lb = 31u - (uint32_t)__clz((int)FB);    // frames per block (a power of two >= 4)
                    // a conversion round is two blocks: lane -> (block of the round, pair of the block) = frames 2 fpl, 2 fpl + 1 of channel cc
                    const uint32_t half = lane >> 5, pr = lane & 31u, fpl = pr / C, cc = pr - fpl * C;
                    // (1) the 12 frames in front of the tile live in the halo, which (2) is about to replace
                    float s0 = 0.0f, s1 = 0.0f;
                    {
                        const uint32_t q = lane < 6u * C ? lane : 0u, qf = q / C, qc = q - qf * C;      // pair q: frames -12 + 2 qf, channel qc
                        const float *xp = tile + ((int)(2 * qf) - 12) * (int)C + (int)qc;
                        s0 = xp[0]; s1 = xp[C];
                    }
                    __builtin_amdgcn_wave_barrier();
                    // (2) the next tile's halo leaves the f32 tile before it is overwritten
                    {
                        const uint32_t hn = halo_frames * C;
                        float *dst = tile - hn;
                        const float *srcp = dst + (size_t)seg * C;
                        for (uint32_t j = lane; j < hn; j += 64u) {
                            const float v = srcp[j];
                            __builtin_amdgcn_wave_barrier();
                            dst[j] = v;
                        }
                        halo_done = true;
                    }
                    __builtin_amdgcn_wave_barrier();
                    // (3) convert in place, rounds (two blocks each: 128 floats in, 128 dwords out) from the top down, four
                    // rounds per batch: all reads of a batch, then its writes.  Lane addresses step by 512 bytes per round.
                    // The top block may run past the converted range into the slack (sized for it); nothing reads that.
                    const uint32_t nblocks = (ps + FB - 1) >> lb;
                    const int nround = (int)((nblocks + 1u) >> 1);
                    const bool top_ok = 2u * (uint32_t)(nround - 1) + half < nblocks;     // odd block count: the top round's upper block does not exist
                    const float *rbase = tile + ((int)(half * FB + 2u * fpl) - 12) * (int)C + (int)cc;    // round 0; + 128 floats per round
                    uint32_t *wbase = reinterpret_cast<uint32_t *>(tile) + half * 64u + cc * (FB >> 1) + fpl;   // hi pair; lo: + 32
                    const bool saver = half * FB + 2u * fpl < 12u;         // this lane saved its round-0 pair from the old halo (lane < 6 C)
                    // (hi, hi) = f16(s x0), f16(s x1);  (lo, lo) = f16(s x - hi): four v_fma_mix
                    auto split2 = [&](float xa, float xb, uint32_t &hh, uint32_t &ll) {
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(xa), "s"(scale));
                        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(xb), "s"(scale));
                        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ll) : "v"(xa), "s"(scale), "v"(hh));
                        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ll) : "v"(xb), "s"(scale), "v"(hh));
                    };
                    // The top round (its upper block may not exist) and round 0 (whose first pairs come from the saved halo)
                    // are peeled off, so the rounds in between run without predicates or selects.
                    int rd = nround - 1;
                    const float *rp = rbase + rd * 128;
                    uint32_t *wp = wbase + rd * 128;
                    if (rd > 0) {
                        uint32_t hh, ll;
                        split2(rp[0], rp[(int)C], hh, ll);
                        if (top_ok) { wp[0] = hh; wp[32] = ll; }
                        rd--; rp -= 128; wp -= 128;
                        __builtin_amdgcn_wave_barrier();
                    }
                    for (; rd >= 4; rd -= 4, rp -= 512, wp -= 512) {        // rounds rd .. rd - 3, all >= 1
                        float x0[4], x1[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { x0[u] = rp[-128 * u]; x1[u] = rp[-128 * u + (int)C]; }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            uint32_t hh, ll;
                            split2(x0[u], x1[u], hh, ll);
                            wp[-128 * u] = hh;
                            wp[-128 * u + 32] = ll;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    for (; rd >= 1; rd--, rp -= 128, wp -= 128) {           // up to three rounds left above round 0
                        uint32_t hh, ll;
                        split2(rp[0], rp[(int)C], hh, ll);
                        wp[0] = hh; wp[32] = ll;
                        __builtin_amdgcn_wave_barrier();
                    }
                    {                                                       // round 0
                        uint32_t hh, ll;
                        split2(saver ? s0 : rbase[0], saver ? s1 : rbase[(int)C], hh, ll);
                        if (nround > 1 || top_ok) { wbase[0] = hh; wbase[32] = ll; }
                        __builtin_amdgcn_wave_barrier();
                    }
                    // (4) column (channel tp_c, block of outputs b = 16/C g + mrow / C) reads halves 4 kq .. + 4 of its window
                    // [4 b, 4 b + 16) of converted frames from the hi plane and from the lo plane (aligned 8-byte pieces)
                    const char *tb = reinterpret_cast<const char *>(tile);
                    auto plane_addr = [&](uint32_t j, uint32_t lo) -> const char * {
                        return tb + ((j >> lb) << 8) + (lo << 7) + ((tp_c * FB + (j & (FB - 1u))) << 1);
                    };
                    const uint32_t jb = 4u * ((uint32_t)mrow / C);
                    auto ld4 = [](const char *q, int off) -> halfx4 { return __builtin_bit_cast(halfx4, *reinterpret_cast<const uint2 *>(q + off)); };
                    SS_PROF_MARK(5);
                    float m16 = 0.0f;
                    {
                        // three K = 16 products per group (hi and lo planes at halves 4 kq .. + 4), four groups per iteration:
                        // four independent accumulator chains, and the next iteration's operands are read before this one's MFMAs
                        const char *ph = plane_addr(jb + 4u * (uint32_t)kq, 0);          // lo plane: + 128 bytes; next group: + 256
                        auto absmax4 = [](float m, const floatx4 &a) {
                            return fmaxf(fmaxf(fmaxf(m, fabsf(a[0])), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
                        };
#ifndef SS_TD_MFMA_GROUPS4
#define SS_TD_MFMA_GROUPS4 2      // groups per iteration of this loop in the four-waves-per-SIMD build (the three-waves build: 4)
#endif
                        constexpr int NG = (WPS >= 4) ? SS_TD_MFMA_GROUPS4 : 4;              // independent accumulator chains (2 or 4)
                        constexpr int NGS = NG == 4 ? 2 : 1;
                        const uint32_t nquad = nplanar >> NGS;
                        if (nquad) {
                            halfx4 h[NG], l[NG];
#pragma unroll
                            for (int g = 0; g < NG; g++) { h[g] = ld4(ph, 256 * g); l[g] = ld4(ph, 256 * g + 128); }
                            for (uint32_t it = 0; it < nquad; it++) {
                                ph += (it + 1 < nquad) ? 256 * NG : 0;     // the last iteration re-reads its own operands: nothing past the planes
                                halfx4 hn[NG], ln[NG];
#pragma unroll
                                for (int g = 0; g < NG; g++) { hn[g] = ld4(ph, 256 * g); ln[g] = ld4(ph, 256 * g + 128); }
                                floatx4 acc[NG];
#pragma unroll
                                for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, h[g], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                                for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, l[g], acc[g], 0, 0, 0);
#pragma unroll
                                for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_lo, h[g], acc[g], 0, 0, 0);
                                // max |.| of the results: v_max3 with |abs| source modifiers, two instructions per group (through the
                                // fmaxf / fabsf builtins the compiler quiets every operand first: more than twice as many).  ONE asm
                                // statement naming every accumulator (so that it cannot be scheduled in front of any of the MFMAs), with
                                // the wait states a VALU read needs behind an MFMA write inside the string — inline asm hides the hazard
                                // from the compiler: twelve states between the LAST MFMA and the v_max3 that reads its result (the last
                                // accumulator is read by the last two v_max3).
                                if (NG == 4)
                                    asm volatile("s_nop 5\n\t"
                                                 "v_max3_f32 %0, %0, |%1|, |%2|\n\t"
                                                 "v_max3_f32 %0, %0, |%3|, |%4|\n\t"
                                                 "v_max3_f32 %0, %0, |%5|, |%6|\n\t"
                                                 "v_max3_f32 %0, %0, |%7|, |%8|\n\t"
                                                 "v_max3_f32 %0, %0, |%9|, |%10|\n\t"
                                                 "v_max3_f32 %0, %0, |%11|, |%12|\n\t"
                                                 "v_max3_f32 %0, %0, |%13|, |%14|\n\t"
                                                 "v_max3_f32 %0, %0, |%15|, |%16|"
                                                 : "+v"(m16)
                                                 : "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]),
                                                   "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[1][2]), "v"(acc[1][3]),
                                                   "v"(acc[NG - 2][0]), "v"(acc[NG - 2][1]), "v"(acc[NG - 2][2]), "v"(acc[NG - 2][3]),
                                                   "v"(acc[NG - 1][0]), "v"(acc[NG - 1][1]), "v"(acc[NG - 1][2]), "v"(acc[NG - 1][3]));
                                else
                                    asm volatile("s_nop 10\n\t"
                                                 "v_max3_f32 %0, %0, |%1|, |%2|\n\t"
                                                 "v_max3_f32 %0, %0, |%3|, |%4|\n\t"
                                                 "v_max3_f32 %0, %0, |%5|, |%6|\n\t"
                                                 "v_max3_f32 %0, %0, |%7|, |%8|"
                                                 : "+v"(m16)
                                                 : "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]),
                                                   "v"(acc[NG - 1][0]), "v"(acc[NG - 1][1]), "v"(acc[NG - 1][2]), "v"(acc[NG - 1][3]));
#pragma unroll
                                for (int g = 0; g < NG; g++) { h[g] = hn[g]; l[g] = ln[g]; }
                            }
                            ph += 256 * NG;
                        }
                        for (uint32_t g = nquad << NGS; g < nplanar; g++, ph += 256) {    // up to NG - 1 groups left
                            const halfx4 h0 = ld4(ph, 0), l0 = ld4(ph, 128);
                            floatx4 acc0 = {0.f, 0.f, 0.f, 0.f};
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, h0, acc0, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_hi, l0, acc0, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16_lo, h0, acc0, 0, 0, 0);
                            m16 = absmax4(m16, acc0);
                        }
                    }
                    tp_run = fmaxf(tp_run, m16 * __uint_as_float(inv_bits));
                }
            } else {
                // channel counts that do not divide 16 (5.1 = 6 channels, 3, 5, 7 ...): channel-major groups — the 16
                // columns of a group are 16 consecutive blocks of ONE channel, so a lane's running maximum belongs to
                // that channel and one LDS atomic per channel and tile closes it (it was one per group)
                const uint32_t nblk = (seg + Cfg::BLK - 1) / Cfg::BLK;
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
        SS_PROF_MARK(6);
        if (kTpPlanar) tp_prev_bits = seg >= 12u ? tp_now_bits : (tp_now_bits > tp_prev_bits ? tp_now_bits : tp_prev_bits);

        if (!SPLIT) tile_carry_out();
        // ---- new halo: the halo_frames frames before the tile end (a contiguous copy; when the tile
        // is shorter than the halo the source reaches into the old halo).  Ascending order is safe:
        // the source of element j sits seg*C floats above its destination, beyond anything written so far.
        if (!halo_done) {
            const uint32_t hn = halo_frames * C;
            float *dst = tile - hn;
            const float *srcp = dst + (size_t)seg * C;
            for (uint32_t j = lane; j < hn; j += 64u) {
                const float v = srcp[j];
                __builtin_amdgcn_wave_barrier();
                dst[j] = v;
            }
        }
        }                                               // (this wave's tile)
        if (sub_done) {
            sb++;
            slot = slot + 1u == p.sub_cap ? 0u : slot + 1u;
        }
        pos = npos;
        off = noff;
        toff = ntoff;
        seg = nseg_frames;
        ti++;
        SS_PROF_MARK(7);
    } while (seg != 0);
    SS_PROF_END;

    fold_results(ti);
#undef SS_TILE_FRAMES
#undef SS_PREFETCH
}

// Chunk length L (frames one lane filters per tile; a tile is 64 / C chunks).  Candidates are rated by a model of the
// sequential work per 100 ms sub-block, in units of one batched K-weighting step of a lane, calibrated on chunk-length
// sweeps of four shapes (tools/sweep_td_chunk.py: 48 / 96 / 44.1 kHz stereo, 96 kHz 8 channels):
//   pieces x (F + steps(L) + 1.2 x frames of the incomplete last chunk)  x  occupancy(workgroups per CU)  x  exposure
//   F        ~ 80   what a tile costs whatever its length (staging, scan, conversion set-up, decimation, tile tail),
//   steps(L) = L, a step outside whole batches of kTdBatch (a look-ahead single) counting 1.5, plus 3 % per extra way of
//              LDS bank conflict of the per-lane walk (lane (chunk, channel) reads tile[chunk L C + channel]),
//   an incomplete last chunk runs the plain recurrence on one lane per channel, both passes: ~1.2 steps per frame,
//   occupancy: three workgroups per CU cost 1.25x, two 1.6x their step count (longer tiles need more LDS),
//   exposure : a tile beyond the 2048 floats the prefetch registers hold loads the rest at the point of use; tiles that
//              do not start on 16-byte boundaries are staged with scalar loads (x 1.6).
// A length that cuts the sub-block into whole tiles of whole chunks keeps every lane busy and has no incomplete chunk:
// 48 kHz stereo 4800 = 5 x 32 x 30 — k_time_domain 2.14 -> 1.98 ms against L = 33 (5 tiles of 29 chunks + 3 frames)
// although the walk of an even L is 2-way bank conflicted; 96 kHz stereo 2.91 -> 2.17 ms (was L = 65); BASELINE
// config 5 (96 kHz, 8 channels) 3.02 -> 2.08 ms (was L = 65: 19 tiles of 7 chunks + 51 frames; now 40 tiles of 8 x 30).
static uint32_t td_lds_blocks(uint32_t C, uint32_t tile_len)
{
    uint32_t wave_floats = ((uint32_t)kTdHaloFrames + tile_len) * C + td_slack_floats(C) + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    const size_t lds = (size_t)wave_floats * 4 * kTdWavesPerBlock;
    uint32_t blocks = (uint32_t)((160u * 1024u) / lds);
    const uint32_t max_blocks = (4u * SS_TD_WAVES) / kTdWavesPerBlock;
    return blocks > max_blocks ? max_blocks : blocks;
}

uint32_t td_chunk_frames(uint32_t C, uint32_t s100)
{
#ifdef SS_TUNING        // development builds only: force a chunk length (tools/sweep_td_chunk.py)
    if (const char *e = std::getenv("SS_TD_L")) { const int v = std::atoi(e); if (v >= 8 && v <= 128) return (uint32_t)v; }
#endif
    const uint32_t nch = 64u / C;
    const bool rowscan = C == 1 || C == 2 || C == 8;          // the compile-time channel counts whose chunks are dealt to the DPP rows
    static const double occupancy[5] = {8.0, 3.0, 1.6, 1.25, 1.0};
    uint32_t best = 33; double best_cost = 1e300;
    for (uint32_t L : {20u, 25u, 30u, 33u, 35u, 40u, 45u, 49u, 50u, 55u, 60u, 65u}) {
        const uint32_t cap = nch * L;
        const uint32_t pieces = (s100 + cap - 1) / cap;
        uint32_t tile_len = (s100 + pieces - 1) / pieces;
        if (tile_len > cap) tile_len = cap;
        const uint32_t blocks = td_lds_blocks(C, tile_len);
        if (blocks == 0) continue;
        const uint32_t rem = tile_len % L;
        uint32_t ways = 1;                                    // bank conflicts of one pass read, per half wave
        for (uint32_t half = 0; half < 2; half++) {
            uint32_t hits[64] = {0};
            for (uint32_t l = 32 * half; l < 32 * half + 32; l++) {
                uint32_t chunk, ch;
                if (rowscan) { const uint32_t q = (l & 15u) / C; chunk = (l >> 4) + 4u * q; ch = (l & 15u) - q * C; }
                else { chunk = l / C; ch = l - chunk * C; }
                if (chunk >= nch) continue;
                const uint32_t n = ++hits[(chunk * L * C + ch) & 63u];
                if (n > ways) ways = n;
            }
        }
        const double steps = ((double)(kTdBatch * (L / kTdBatch)) + 1.5 * (double)(L % kTdBatch)) * (1.0 + 0.03 * (double)(ways - 1));
        const uint32_t tile_floats = tile_len * C;
        const double exposure = 1.0 + (tile_floats > 2048u ? 0.35 * (double)(tile_floats - 2048u) / 2048.0 : 0.0);
        double cost = (double)pieces * (80.0 + steps + 1.2 * (double)rem) * occupancy[blocks > 4 ? 4 : blocks] * exposure;
        if ((tile_floats & 3u) != 0u) cost *= 1.6;           // tiles that do not start on 16 bytes are staged with scalar loads
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
    uint32_t wave_floats = (halo + tile_len) * C + td_slack_floats(C) + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    const size_t lds = (size_t)wave_floats * 4 * kTdWavesPerBlock;
    uint32_t blocks = lds ? (uint32_t)((160u * 1024u) / lds) : 4u;
    const uint32_t max_blocks = (4u * SS_TD_WAVES) / kTdWavesPerBlock;      // launch bound: SS_TD_WAVES waves per SIMD
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    return blocks * kTdWavesPerBlock;
}

template <int FACTOR, bool RING, int CT, int WAVE, int WPS, bool SPLIT = false>
static hipError_t td_launch_w(const TdParams &p, hipStream_t s)
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
    uint32_t wave_floats = (halo + tile_len) * C + td_slack_floats(C) + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    const size_t lds = (size_t)wave_floats * 4 * kTdWavesPerBlock + (SPLIT ? sizeof(TdShare) : 0);
    auto fn = k_time_domain<FACTOR, RING, CT, WAVE, WPS, SPLIT>;
    static DevicePrep prepared;                     // one per kernel instantiation
    const hipError_t pe = prepare_on_device(prepared, [fn] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (pe != hipSuccess) return pe;
    const uint32_t waves = p.n_streams * p.nseg;
    const uint32_t blocks = SPLIT ? p.n_streams : (waves + kTdWavesPerBlock - 1) / kTdWavesPerBlock;      // SPLIT: a workgroup per stream
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * kTdWavesPerBlock), lds, s, p, L, tile_len, wave_floats, halo);
    return hipGetLastError();
}

// compute units of the current device (256 on MI355X), asked once
static uint32_t td_device_cus()
{
    static std::atomic<uint32_t> cached{0};
    uint32_t n = cached.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = (uint32_t)v;
        else n = 256;
        cached.store(n, std::memory_order_relaxed);
    }
    return n;
}

template <int FACTOR, bool RING, int CT, int WAVE>
static hipError_t td_launch(const TdParams &p, hipStream_t s)
{
    // a workgroup is four waves, one per SIMD: three workgroups per CU hold the whole grid -> the spill-free build
    const uint64_t waves = (uint64_t)p.n_streams * p.nseg;
    const uint64_t blocks = (waves + kTdWavesPerBlock - 1) / kTdWavesPerBlock;
    if (SS_TD_WAVES == 4 && blocks <= 3ull * td_device_cus()) return td_launch_w<FACTOR, RING, CT, WAVE, 3>(p, s);
    return td_launch_w<FACTOR, RING, CT, WAVE, SS_TD_WAVES>(p, s);
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
    // streaming calls (the handle's add_samples, the session ticks) longer than one tile: the four waves of a workgroup share
    // the call's tiles (SPLIT, see TdShare); the spill-free three-waves-per-SIMD build — a handful of workgroups at most
    bool split = RING && p.nseg == 1 && !p.frames_of;
#ifdef SS_TUNING        // development builds only: SS_TD_SPLIT=0 keeps streaming calls on one wave (A/B, drift measurements)
    if (const char *e = std::getenv("SS_TD_SPLIT")) split = split && std::atoi(e) != 0;
#endif
    if (split) {
        const uint32_t C = p.channels, S = p.s100;
        const uint32_t cap = (64u / C) * td_chunk_frames(C, S);
        const uint32_t pieces = (S + cap - 1) / cap;
        uint32_t tile_len = (S + pieces - 1) / pieces;
        if (tile_len > cap) tile_len = cap;
        if (p.n_frames > tile_len)
            return p.channels == 2 ? td_launch_w<FACTOR, RING, 2, 0, 3, true>(p, s) : td_launch_w<FACTOR, RING, 0, 0, 3, true>(p, s);
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

#ifdef SS_TD_PROF
// development builds only: [0..7] phase clocks (stage, decimate, pass 1, scan, pass 2, tp convert (+ f32 remainder), tp product,
// tile tail), [15] waves counted; reset != 0 clears the totals after reading
extern "C" int ss_debug_td_prof(unsigned long long *out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(ssk::g_td_prof), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(ssk::g_td_prof), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
