// ss_fft.hip: spectrum kernels (N = 4096 mid/side, N = 16384, generic power of two) — hand-written gfx950 (CDNA4, wave64) kernels of the soundscope analyzer hot path.
// Reference semantics: /root/reference/src/analyzer.rs (get_fft :55-105, get_waveform :107-137,
// add_samples/getters :139-164, calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as restated in DESIGN.md.
// Nothing here is translated from the reference: the reference has no GPU code.
#include "ss_kernels.h"
#include "ss_fft_dev.h"
#include <cstdlib>
#include <type_traits>

#ifndef SS_COLS_FOLD
#define SS_COLS_FOLD 1    // columns-only epilogue: 1 folds a lane's four bins in registers first (one or two atomics per row and group); 0 one atomic per bin
#endif
#ifndef SS_FFT_PAIRW_WAVES
#define SS_FFT_PAIRW_WAVES 3   // min waves per SIMD k_fft4096_pairw is register-allocated for (4: 128 VGPRs with 13 spilled, measured 10 % slower)
#endif
#ifndef SS_FFT_WAVES
#define SS_FFT_WAVES 2   // min waves per SIMD the N=4096 pair kernel is register-allocated for
#endif

namespace ssk {

// ============================================================================
//  Spectrum, N = 4096, stereo -> mid/side packed as one complex FFT.
//
//  z[n] = (mid[n] + i side[n]) * hann[n];  Z = FFT_4096(z);
//  M[k] = (Z[k] + conj Z[N-k]) / 2,  S[k] = (Z[k] - conj Z[N-k]) / (2i).
//  4096 = 16 x 16 x 16: three register-resident radix-16 passes, two full LDS
//  exchanges plus a half-size mirror exchange.  256 threads = one window at a
//  time; a workgroup walks `windows_per_block` consecutive windows of one
//  stream and keeps the raw samples in registers, so with hop = 256*HS each
//  sample is fetched from HBM once per workgroup (HS new slots per window).
//
//  Index algebra (n = t + 256 j, t = tb + 16 ta, k = ka + 16 kb + 256 kc):
//   P1: A[ka]  = sum_j  z[t+256j] W16^(j ka)            ; *= W4096^(t ka)
//   P2: B[kb]  = sum_ta A'[ka; tb+16ta] W16^(ta kb)     ; *= W256^(tb kb)
//   P3: Z[ka+16kb+256kc] = sum_tb B'[ka,kb; tb] W16^(tb kc)
//  Thread roles: P1 thread = t; P2 thread = tb + 16 ka; P3 thread = ka + 16 kb,
//  which then owns bins v + 256 kc — stride-256, so output stores coalesce and
//  the mirror bin N-k lives at thread 256-v, slot 15-kc.
// ============================================================================
// kX1Stride = 272 (ss_fft_dev.h): anyhop kernel: 256 + 16 de-phases the 4 ka-groups of a wave across banks
constexpr int kX2Stride = 17;    // anyhop kernel: row of 16 padded to 17, conflict-free b64 row reads
// pair kernel: both exchanges store rows of 16 complex padded to 18 (144 B): the reader's row is
// 16-B aligned and contiguous (8 x ds_read_b128), 16-lane write groups and 16-lane read groups both
// land on 16 distinct 4-bank slots (36*i mod 64 is a permutation of the multiples of 4).
// (kRow = 18, kPlane = 16 * kRow = 288 complex per outer index; 16 planes = 4608 complex = 36864 B: ss_fft_dev.h)
// batch kernels (k_fft4096_ms1, k_fft4096_pairw, k_fft16k_run): rows of 16 complex padded to 17, planes of 16 rows (272, = 16
// mod 32).  By the LDS rules of MI355X_MICROARCH.md (ds_write_b64: contiguous 16-lane groups, banks (a/4) mod 32;
// ds_read_b64: 32-lane groups, (a/4) mod 64) both exchanges are then free of bank conflicts in BOTH directions:
//   write (ka; tb, hi) at ka*272 + tb*17 + hi : a group is tb = 0..15, dword 34 tb + c = 2 tb + c mod 32 — 32 distinct banks
//     (with rows of 18 it was 4 tb mod 32: two deep, the largest single share of the kernels' conflict cycles);
//   write (kb; hi, tb) at kb*272 + hi*17 + tb : 32 consecutive dwords;
//   read  (hi; tb, 0..15) at hi*272 + tb*17 + j as sixteen ds_read_b64: a group is two planes x 16 rows, and
//     {17 tb} U {17 tb + 16} mod 32 covers every residue once.
// An odd row stride rules out ds_read_b128 (rows are 8-byte aligned), which costs nothing in the LDS array (b64 and b128
// both move 256 B per cycle) — but left to itself the compiler pairs such reads into ds_read2_b64, which moves 128 B per
// cycle: the reads go through lds_ld64 (volatile, LDS address space: one ds_read_b64 each, never merged).
constexpr int kRowB = 17;
constexpr int kPlaneB = 16 * kRowB;

// Published spectrum layout of the N = 4096 kernels: bin k lives at k with bits 1:0 XORed with bits 6:5.  A lane owns four
// consecutive retained bins k0 + e (k0 = first_bin + 4 g) and reads bin e of all lanes with one ds_read_b64 (32-lane groups,
// banks (a/4) mod 64): the lanes' addresses are 32 bytes apart, so unswizzled only 8 of the 32 bank pairs would be used,
// four deep; the XOR sends lanes g, g + 8, g + 16, g + 24 to the four different pairs of their 32-byte slot — conflict-free
// for the bins and for their mirrors 4096 - k (modelled with the rules of MI355X_MICROARCH.md, then measured; flipping bit 1
// by bit 6 alone, as before, left them two deep).  The publishing writes (16 consecutive k per 16-lane group, permuted
// inside aligned quads) stay conflict-free.
#define SPEC_POS(k) ((k) ^ (((k) >> 5) & 3))

// dB epilogue of one window.  xb holds the full spectrum Z[0..4095] in that order; a thread
// owns groups of FOUR consecutive retained bins (g = t, t + 256), so both output rows are written
// with 16-byte stores (rows are padded to a multiple of 4 floats): the 4-byte-per-lane stores of
// a stride-256 ownership were store-issue bound (1.6 ms of 4.7 ms at the config-3 size).
// LDS_TABLE: `offpink` is the workgroup's LDS copy of the table (k_fft4096_ms1 / k_fft4096_pairw stage it once per run of
// windows): the rows are read right where they are used instead of being requested from global memory a whole epilogue
// ahead (eight registers held across it, and as many vector-memory requests per window as the samples themselves).
// COLS: nothing is stored — every dB value is folded into its chart column in LDS (tui.rs:49-51, :801-821; the column rule is
// the library's, include/soundscope_hip.h).  The accumulators hold plain dB values: the gain and the chart's [-100, 0] clamp
// are monotone, so they commute with the maximum and are applied ONCE per column when a window's columns are flushed —
// max(clamp(v + gain)) == clamp(max(v) + gain) to the bit.  Folding is an LDS float atomic (ds_max_f32, which ignores a NaN
// operand like IEEE maxNum) at a byte offset from a host-built table — no per-bin branch, key conversion or clamp in the vector
// ALU, which is what bounds this kernel (the divergent per-bin ds_min_u32 form of rounds 3-4: 3.30 ms; this one 3.00).
//   * Chart columns are monotone in the bin index and, above the lowest few dozen bins, at least four bins wide: a lane's four
//     consecutive bins lie in ONE column or straddle one boundary.  coltab[g] = (o0 | o3 << 16, n): the offsets
//     of the first and the last bin's column and the number n of bins in the first (0: a general group, below).  The two run maxima are taken in registers
//     (five selects, three maxima per row) and one atomic per row goes out — two where the group straddles.  One atomic per BIN
//     and no arithmetic at all (SS_COLS_FOLD 0) was measured: 3.86 ms — the sixteen-odd lanes that share a wide column queue on
//     one LDS address (SQ_LDS_BANK_CONFLICT 40 % of the LDS cycles), profiles/r05_ab_columns.txt.
//   * Groups the two-run form cannot express (three or four columns inside the group: the lowest bins; the row padding in the
//     last group, whose offsets point at a spare slot) carry n = 0 and fold bin by bin from colbins — a wave-uniform branch.
//   * An accumulator starts at -inf where the column owns a bin and at NaN where it owns none (p.col_init); the flush turns NaN
//     into "no bin" and -inf (every bin of the column was NaN) into the -100 the two-pass kernel gives.
// `side_off` (wave-uniform): the second row rode the transform multiplied by 2^E (its block exponent, see "Two rows, one
// transform" below); side_off = -E * 20 log10(2) takes the factor out again in dB.
constexpr uint32_t kColStride = 516;            // floats between the two rows' accumulators (512 columns + the spare slot)
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ void lds_fmax(lds_f32 *p, float v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // ds_max_f32
template <bool LDS_TABLE = false, bool COLS = false>
__device__ __forceinline__ void fft4096_epilogue(const v2f *xb, int t, uint32_t first_bin, uint32_t n_bins,
                                                 float db_offset, const float *__restrict__ offpink,
                                                 float *o_mid, float *o_side, bool store_side = true,
                                                 float side_off = 0.0f, float *colbuf = nullptr, const uint2 *coltab = nullptr,
                                                 const uint2 *colbins = nullptr)
{
    const uint32_t ngroups = (n_bins + 3) >> 2;
    // dB = 10 log10(2) * log2(q) + (db_offset + pink[bin]); an exact zero reads -150 (+ pink): the log operand is
    // replaced by the value that lands on -150.  Exact zeros are rare (digital silence), so the wave first asks whether
    // any of its squared magnitudes is zero (a min tree and one compare) and only then pays the per-value selects.
    //
    // Memory order: every load of the window (both groups' table rows here; the caller's prefetched samples before the
    // call) is consumed before the first store is issued.  Loads and stores share vmcnt and may complete out of order
    // with each other, so a load waited for behind a store costs s_waitcnt vmcnt(0), i.e. the stores' completion.  In the
    // per-phase clock profile (-DSS_FFT_PROF) this order takes a fifth off the epilogue's wave time; the kernel time does
    // not move (the other two workgroups of the CU cover the wait), it is kept because it also frees two spilled registers.
    constexpr float kDb = 3.01029995663981195f;
    // (wave-uniform, but a VALU division: pinned to a scalar register instead of a vector register held across the window loop)
    const float lg0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint((-150.0f - db_offset) / kDb)));
    float4 op[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t g = (uint32_t)t + 256u * i;
        op[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        // byte offset recomputed per window (opaque to the optimiser): hoisted out of the window loop the two 64-bit row
        // addresses cost four vector registers for its whole length — this kernel sits at the three-waves-per-SIMD edge
        uint32_t boff = 16u * g;
        asm volatile("" : "+v"(boff));
        if (!LDS_TABLE && g < ngroups) op[i] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(offpink) + boff);   // table padded to the row stride
    }
    float rm[2][4], rs[2][4];
    // COLS: a group's eight values are folded as soon as they exist (nothing is stored, so nothing has to wait for the loads of
    // the other group; sixteen values would otherwise live across both groups' arithmetic)
    lds_char *const cb = (lds_char *)colbuf;
    auto fold_group = [&](const int i, const uint32_t g) {
                const uint2 ct = coltab[g];
#if SS_COLS_FOLD
                const uint32_t o0 = ct.x & 0xFFFFu, o3 = ct.x >> 16, nf = ct.y;     // (a whole dword: compares against inline constants)
                const bool general = nf == 0u;
                if (__builtin_expect(__ballot(general) != 0ull, 0)) {          // (wave-uniform: the wave that owns the lowest bins / the padding)
                    if (general) {
                        uint32_t gg = g;                       // (opaque: hoisted out of the window loop this address was spilled)
                        asm volatile("" : "+v"(gg));
                        const uint2 cw = colbins[gg];
                        const uint32_t o[4] = {cw.x & 0xFFFFu, cw.x >> 16, cw.y & 0xFFFFu, cw.y >> 16};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            lds_f32 *a = (lds_f32 *)(cb + o[e]);
                            lds_fmax(a, rm[i][e]);
                            lds_fmax(a + kColStride, rs[i][e]);
                        }
                    }
                }
                if (!general) {
                    const bool p1 = nf > 1u, p2 = nf > 2u, p3 = nf > 3u;
                    auto max3 = [](float a, float b, float c) -> float { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
                    auto max2 = [](float a, float b) -> float { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
#pragma unroll
                    for (int row = 0; row < 2; row++) {
                        const float *r = row ? rs[i] : rm[i];
                        // (a bin outside a run is replaced by a bin inside it: a duplicate cannot change a maximum)
                        float f = max3(r[0], p1 ? r[1] : r[0], p2 ? r[2] : r[0]);
                        f = max2(f, p3 ? r[3] : r[0]);
                        const float l = max3(r[3], p2 ? r[3] : r[2], p1 ? r[3] : r[1]);
                        lds_fmax((lds_f32 *)(cb + o0) + row * kColStride, f);
                        if (!p3) lds_fmax((lds_f32 *)(cb + o3) + row * kColStride, l);
                    }
                }
#else
                const uint32_t o[4] = {ct.x & 0xFFFFu, ct.x >> 16, ct.y & 0xFFFFu, ct.y >> 16};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    lds_f32 *a = (lds_f32 *)(cb + o[e]);
                    lds_fmax(a, rm[i][e]);
                    lds_fmax(a + kColStride, rs[i][e]);
                }
#endif
    };
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t g = (uint32_t)t + 256u * i;
        if (g < ngroups) {
            uint32_t k0 = first_bin + 4 * g;
            // The sixteen LDS addresses of a thread's two groups are window-loop invariants the compiler keeps in registers;
            // the columns-only instantiation sits at the three-waves-per-SIMD limit, where two of them end up in scratch
            // (reloaded at the top of every epilogue).  There the second group's index is opaque per window instead: its
            // eight addresses are recomputed (a few integer instructions each) and nothing is spilled.
            if (COLS && i == 1) asm volatile("" : "+v"(k0));
            // every LDS read of the group is requested before the first value is used (the transform's 32 registers are dead
            // here): left to itself the compiler reads a pair of values, waits, computes, reads the next — one exposed LDS round
            // trip per bin
            v2f zk[4], zm[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t k = k0 + e;                       // k <= 2051 < 4096: the mirror index stays positive
                zk[e] = xb[SPEC_POS(k)];
                zm[e] = xb[SPEC_POS(4096 - k)];                  // Z[N - k]
            }
            if (LDS_TABLE) op[i] = reinterpret_cast<const float4 *>(offpink)[g];
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            float qm[4], qs[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                // 2 M = (zk.x + zm.x, zk.y - zm.y), 2 S ~ (zk.y + zm.y, zk.x - zm.x).  Held across the two rows — P = (2M.x, 2S.x)
                // = zk + zm, Q = (2M.y, 2S.y) — both squared magnitudes are ONE packed multiply and one packed fma,
                // (|2M|^2, |2S|^2) = P P + Q Q, with the roundings of fmaf(x, x, y * y) per row.
                v2f P = zk[e] + zm[e], Q, t, qq;
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Q) : "v"(zk[e]), "v"(zm[e]));
                asm("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(Q));
                asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(qq) : "v"(P), "v"(t));
                qm[e] = qq.x;
                qs[e] = qq.y;
            }
            const float opv[4] = {op[i].x, op[i].y, op[i].z, op[i].w};
            // the second row's offsets carry its block exponent (two packed adds per group)
            const v2f so2 = {side_off, side_off};
            const v2f osa = v2f{op[i].x, op[i].y} + so2, osb = v2f{op[i].z, op[i].w} + so2;
            const float ops[4] = {osa.x, osa.y, osb.x, osb.y};
            const float qmin = fminf(fminf(fminf(qm[0], qm[1]), fminf(qm[2], qm[3])), fminf(fminf(qs[0], qs[1]), fminf(qs[2], qs[3])));
            if (__builtin_expect(__ballot(qmin == 0.0f) == 0ull, 1)) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    rm[i][e] = fmaf(__log2f(qm[e]), kDb, opv[e]);
                    rs[i][e] = fmaf(__log2f(qs[e]), kDb, ops[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    rm[i][e] = fmaf(qm[e] == 0.0f ? lg0 : __log2f(qm[e]), kDb, opv[e]);
                    rs[i][e] = qs[e] == 0.0f ? fmaf(lg0, kDb, opv[e]) : fmaf(__log2f(qs[e]), kDb, ops[e]);      // a zero is -150 whatever the exponent
                }
            }
            if (COLS) fold_group(i, g);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) { rm[i][e] = 0.0f; rs[i][e] = 0.0f; }
        }
    }
    if (COLS) return;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);          // the stores stay behind everything above
#endif
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t g = (uint32_t)t + 256u * i;
        if (g < ngroups) {
            reinterpret_cast<float4 *>(o_mid)[g] = make_float4(rm[i][0], rm[i][1], rm[i][2], rm[i][3]);
            if (store_side) reinterpret_cast<float4 *>(o_side)[g] = make_float4(rs[i][0], rs[i][1], rs[i][2], rs[i][3]);
        }
    }
}

// The two rows of a window ride ONE complex transform, whose rounding noise (about -134 dB under the louder row) stands in
// a row whose signal was EXACTLY zero over the whole window — where the reference, which transforms each signal on its own,
// reports its floor: a buffer of zeros has magnitude 0 in every bin, hence -150 dB (analyzer.rs:20-22).  Dual-mono files
// (L == R: the side signal is zero) and digital silence in front of a programme (the window pairs of k_fft4096_pairw) are
// the everyday cases.  The kernels know the level of either row of a window (see "Two rows, one transform" below: level 0 =
// every sample is +-0; a NaN does not count, its row is garbage either way); a window whose row is empty takes this rare
// path BEHIND the ordinary epilogue and overwrites that row with the floor, -150 dB + pink (exactly what the epilogue
// writes for a zero magnitude).  Kept out of the epilogue itself: the hot path's registers.
// Rows of windows the reference REFUSES (a NaN or an infinite sample inside: SpectrumAnalyzerError::NaNValuesNotSupported /
// InfinityValuesNotSupported, analyzer.rs:60-65): NaN in every bin.  (A transform that carries such a sample comes out non-finite
// in every bin by itself; this is for the kernels that pack TWO windows into one transform and keep the refused one out of it.)
__device__ __forceinline__ void fft4096_nan_rows(int t, uint32_t n_bins, float *o_first, float *o_second, bool first_nan, bool second_nan)
{
    const uint32_t ngroups = (n_bins + 3) >> 2;
    __builtin_amdgcn_s_waitcnt(0);                  // the ordinary stores of these rows have left the wave: these come after them
    const float qn = __builtin_nanf("");
    for (uint32_t g = (uint32_t)t; g < ngroups; g += 256u) {
        if (first_nan) reinterpret_cast<float4 *>(o_first)[g] = make_float4(qn, qn, qn, qn);
        if (second_nan) reinterpret_cast<float4 *>(o_second)[g] = make_float4(qn, qn, qn, qn);
    }
}

__device__ __forceinline__ void fft4096_floor_rows(int t, uint32_t n_bins, float db_offset, const float *__restrict__ offpink,
                                                float *o_first, float *o_second, bool first_zero, bool second_zero)
{
    const uint32_t ngroups = (n_bins + 3) >> 2;
    __builtin_amdgcn_s_waitcnt(0);                  // the ordinary stores of these rows have left the wave: these come after them
    for (uint32_t g = (uint32_t)t; g < ngroups; g += 256u) {
        const float4 op = *reinterpret_cast<const float4 *>(offpink + 4 * g);
        // -150 + pink, with pink = table - offset: EXACTLY -150 where the table carries no compensation (ss_get_fft's, which
        // adds it in f64 on the host like analyzer.rs:82 — a buffer of zeros must read -150 to the bit there)
        const float4 v = make_float4(-150.0f + (op.x - db_offset), -150.0f + (op.y - db_offset), -150.0f + (op.z - db_offset), -150.0f + (op.w - db_offset));
        if (first_zero) reinterpret_cast<float4 *>(o_first)[g] = v;
        if (second_zero) reinterpret_cast<float4 *>(o_second)[g] = v;
    }
}

// the same for the columns-only mode: the empty row's columns are REPLACED by the floor's (a barrier separates this from
// the epilogue's atomics; every column of the row is rewritten from scratch by a plain store, then the floor values — a
// monotone function of the bin's pink compensation — are folded in like any other row)
__device__ __forceinline__ void fft4096_floor_columns(int t, uint32_t n_bins, float db_offset, const float *__restrict__ offpink,
                                                      float *colbuf, const uint16_t *bincol, const float *col_init, uint32_t cols,
                                                      bool first_zero, bool second_zero)
{
    __syncthreads();
    for (uint32_t c = (uint32_t)t; c < cols; c += 256u) {
        if (first_zero) colbuf[c] = col_init[c];
        if (second_zero) colbuf[kColStride + c] = col_init[c];
    }
    __syncthreads();
    for (uint32_t k = (uint32_t)t; k < n_bins; k += 256u) {
        const uint32_t c = bincol[k];                                   // (global memory: this path is rare)
        const float v = -150.0f + (offpink[k] - db_offset);           // the SAME expression as fft4096_floor_rows (fmaf(lg0, kDb, offpink) rounds
                                                                       // differently: 2 ulp at 88.2 kHz, tools/fuzz_columns.py seed 87)
        if (first_zero) lds_fmax((lds_f32 *)colbuf + c, v);
        if (second_zero) lds_fmax((lds_f32 *)colbuf + kColStride + c, v);
    }
}

// ============================================================================
//  Two rows, one transform: the block exponent of the second row
//
//  The packed kernels carry two real signals a, b on one complex transform, z = (a + i b) hann.  Every twiddle multiply
//  and every (a - i b)-type butterfly rounds relative to |z|, so the rounding noise of BOTH extracted spectra sits about
//  134 dB under the LOUDER row: a side signal 40 dB under mid came out 0.04 dB off inside its own 70 dB range — where the
//  reference, which transforms each signal on its own (tui.rs:1505,1515 -> analyzer.rs:55-65), has its noise 134 dB
//  under the side row's own peak.  The kernels therefore transform  z = (a + i 2^E b) hann  with E chosen per window so
//  that the two windowed signals have the same level, and take 2^E out again in the epilogue as a dB offset on the second
//  row (fft4096_epilogue's side_off).  Powers of two are exact: row b comes out as the transform of b itself would, with
//  rounding noise relative to max(|a|, |2^E b|) = its own level.
//
//  Level of a row = its largest windowed sample magnitude, as a bit pattern (non-negative floats order like unsigned
//  integers, and the difference of two patterns is 2^23 log2 of the ratio to within 0.09: block_exp).
//    * k_fft4096_ms1 / k_fft4096_pairw (hop 1024) keep, per hop of 1024 frames, the largest RAW magnitude of either signal
//      (one wave reduction per entering hop and signal, combined across the four waves through LDS behind a barrier the
//      pass structure already has).  A window is four hops; the two inner ones carry Hann weights in [1/2, 1], the two outer
//      ones in [0, 1/2].  While the outer hops are at most twice as loud as the inner ones, the windowed level of a row is
//      within a factor two of its inner hops' raw level and E follows from those (the ordinary case: no extra work).
//      Otherwise (an onset or a decay inside the window) the workgroup takes the exact path: the windowed energy
//      sum (sample x weight)^2 of either row (block_exp_energy), one more barrier — rare, and uniform across the workgroup.
//    * The second signal's registers are kept multiplied by 2^E (ms1) or its window weights are (pairw), and rewritten only
//      when E moves by two or more: the ordinary window pays for the level of the entering hop, four scalings and two
//      packed adds per group of bins.
//    * k_fft4096_ms<HS> / k_fft4096_ms_anyhop (other hops: not the reference's cadence) take the exact path every window.
//  An all-zero row has no level: E stays, and the row reads the reference's floor (fft4096_floor_rows).
// ============================================================================
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// maximum over the wave of a non-negative float's bit pattern; the result stands in lane 63
__device__ __forceinline__ uint32_t wave_umax_lane63(uint32_t v)
{
#define SS_UMAX_DPP(ctrl, rows) v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, true))
    SS_UMAX_DPP(0x111, 0xF);    // row_shr:1
    SS_UMAX_DPP(0x112, 0xF);    // row_shr:2
    SS_UMAX_DPP(0x114, 0xF);    // row_shr:4
    SS_UMAX_DPP(0x118, 0xF);    // row_shr:8     lane 15 of every row holds its row's maximum
    SS_UMAX_DPP(0x142, 0xA);    // row_bcast:15  rows 1 and 3 take in rows 0 and 2
    SS_UMAX_DPP(0x143, 0xC);    // row_bcast:31  rows 2 and 3 take in lane 31: lane 63 holds the wave's
#undef SS_UMAX_DPP
    return v;
}
// sum over the wave of a float; the result stands in lane 63 (fixed order: the same bits on every launch)
__device__ __forceinline__ float wave_fsum_lane63(float v)
{
#define SS_FSUM_DPP(ctrl, rows) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xF, true))
    SS_FSUM_DPP(0x111, 0xF);    // row_shr:1
    SS_FSUM_DPP(0x112, 0xF);    // row_shr:2
    SS_FSUM_DPP(0x114, 0xF);    // row_shr:4
    SS_FSUM_DPP(0x118, 0xF);    // row_shr:8     lane 15 of every row holds its row's sum
    SS_FSUM_DPP(0x142, 0xA);    // row_bcast:15  rows 1 and 3 take in rows 0 and 2
    SS_FSUM_DPP(0x143, 0xC);    // row_bcast:31  rows 2 and 3 take in lane 31: lane 63 holds the wave's
#undef SS_FSUM_DPP
    return v;
}
// the same from two windowed ENERGIES (sums of squares): half the exponent of their ratio.  The exact path matches energies,
// not peaks: a transform's rounding noise in every bin goes with the total energy of what it carries, so a row keeps its own
// noise floor when the other row, scaled, carries about the same energy — whatever the two crest factors are (a decaying
// burst under the window's edge beside a steady row: peak-matched, the burst's row sat 18 dB under its partner's spectrum).
__device__ __forceinline__ int block_exp_energy(uint32_t a, uint32_t b)
{
    const int e = ((int)a - (int)b + (1 << 23)) >> 24;
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}
// the four waves' energy pairs (lv[wave][first, second], floats) -> the workgroup's, in scalar registers
__device__ __forceinline__ void read_energies2(const uint32_t (*lv)[2], uint32_t &a, uint32_t &b)
{
    const uint4 q0 = *reinterpret_cast<const uint4 *>(&lv[0][0]), q1 = *reinterpret_cast<const uint4 *>(&lv[2][0]);
    const float sa = (__uint_as_float(q0.x) + __uint_as_float(q0.z)) + (__uint_as_float(q1.x) + __uint_as_float(q1.z));
    const float sb = (__uint_as_float(q0.y) + __uint_as_float(q0.w)) + (__uint_as_float(q1.y) + __uint_as_float(q1.w));
    a = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sa));
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sb));
}
// exponent that lifts a row at level b to a row at level a (bit patterns of positive floats)
__device__ __forceinline__ int block_exp(uint32_t a, uint32_t b)
{
    const int e = ((int)a - (int)b + (1 << 22)) >> 23;
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}
__device__ __forceinline__ float exp2i(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }      // |e| <= 126
__device__ __forceinline__ float uniform_f(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
// side_off of a block exponent
__device__ __forceinline__ float block_off(int e) { return uniform_f((float)e * -6.02059991327962390f); }
// the four waves' level pairs (lv[wave][first, second]) -> the workgroup's, in scalar registers
__device__ __forceinline__ void read_levels2(const uint32_t (*lv)[2], uint32_t &a, uint32_t &b)
{
    const uint4 q0 = *reinterpret_cast<const uint4 *>(&lv[0][0]), q1 = *reinterpret_cast<const uint4 *>(&lv[2][0]);
    a = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(q0.x, q0.z), umax(q1.x, q1.z)));
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(q0.y, q0.w), umax(q1.y, q1.w)));
}

// HS = hop / 256.  A workgroup iteration transforms TWO consecutive windows: they share the
// sliding sample registers (16 + HS slots) and every per-thread constant, and every barrier
// phase carries two independent radix-16 problems (half the barriers per window, twice the
// instruction-level parallelism to cover LDS latency).
template <int HS>
__global__ __launch_bounds__(256, SS_FFT_WAVES) void k_fft4096_ms(FftBatchParams p)
{
    constexpr int NS = 16 + HS;                                              // sample slots held
    __shared__ __attribute__((aligned(16))) v2f xbuf[2][16 * kPlane];     // 2 x 36864 B
    // exchange 1: element (ka; tb, ta) at ka*272 + (tb + 16 ta): lane-linear b64 writes; the reader's 4
    // ka-groups per wave are de-phased by the +16 pad (conflict-free b64 reads)
#define X1W(ka, tb_, ta_) ((ka) * kX1Stride + (tb_) + 16 * (ta_))
    // exchange 2: element (kb, ka; tb) in rows of 16 padded to 18 (144 B): contiguous b64 writes, and the
    // reader's row is 16-B aligned and contiguous (8 x ds_read_b128; 36*i mod 64 is a permutation of the
    // multiples of 4, so every 16-lane read group hits 16 distinct 4-bank slots)
#define X2W(kb, ka_, tb_) ((kb) * kPlane + (ka_) * kRow + (tb_))
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];                //  2048 B
    // exact block exponents ("Two rows, one transform"): [wave][window 0 mid, side, window 1 mid, side] = largest windowed
    // magnitude of that wave's lanes (bit patterns; 0 = the row is empty and reads the floor)
    __shared__ __attribute__((aligned(16))) uint32_t xlev4[4][8];          // [wave][4 peaks, 4 energies]

    const int t = threadIdx.x;
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    const uint32_t stream = blockIdx.x / groups;
    const uint32_t grp = blockIdx.x - stream * groups;
    const uint32_t w_begin = grp * p.windows_per_block;
    uint32_t w_end = w_begin + p.windows_per_block;
    const uint32_t n_win = p.windows_of ? p.windows_of[stream] : p.n_windows;     // ragged batches: this stream's own count
    if (w_begin >= n_win) return;
    if (w_end > n_win) w_end = n_win;

    const float2 *src = reinterpret_cast<const float2 *>(p.pcm) + (size_t)stream * p.frames_per_stream +
                        p.first_start + (size_t)w_begin * p.hop;

    // per-thread constants, resident across the window loop
    float hw[16];
#pragma unroll
    for (int j = 0; j < 16; j++) hw[j] = p.half_window[t + 256 * j];
    v2f tw1[16];
#pragma unroll
    for (int ka = 1; ka < 16; ka++) tw1[ka] = reinterpret_cast<const v2f *>(p.tw_n)[t * ka];
    tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[t];

    const int tb = t & 15, hi = t >> 4;
    const int tsw = SPEC_POS(t);
    const size_t out_win_stride = (size_t)2 * p.bin_stride;
    float *outp = p.out + ((size_t)stream * p.n_windows + w_begin) * out_win_stride;

    // raw sums / differences of frame t + 256 j of the first window: (l + r, l - r)
    float sm[NS], df[NS];
    const bool two0 = (w_begin + 1 < w_end);
#pragma unroll
    for (int j = 0; j < NS; j++) {
        float2 v = make_float2(0.f, 0.f);
        if (j < 16 || two0) v = src[t + 256 * j];
        sm[j] = v.x + v.y;
        df[j] = v.x - v.y;
    }

    const uint32_t wvid = (uint32_t)__builtin_amdgcn_readfirstlane(t >> 6);
    __syncthreads();
    for (uint32_t w = w_begin; w < w_end; w += 2) {
        const bool two = (w + 1 < w_end);
        // levels of the four rows of this pair of windows, exact: largest |sample x weight| over the workgroup (the loop-end
        // barrier has retired the previous iteration's reads of xlev4)
        uint32_t X[4], G[4];          // peaks (0 = the row is empty) and energies (what the exponent matches, see block_exp_energy)
        {
            float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f, x3 = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const float a0 = sm[j] * hw[j], a1 = df[j] * hw[j], a2 = sm[j + HS] * hw[j], a3 = df[j + HS] * hw[j];
                x0 = fmaxf(x0, fabsf(a0)); x1 = fmaxf(x1, fabsf(a1)); x2 = fmaxf(x2, fabsf(a2)); x3 = fmaxf(x3, fabsf(a3));
                g0 = fmaf(a0, a0, g0); g1 = fmaf(a1, a1, g1); g2 = fmaf(a2, a2, g2); g3 = fmaf(a3, a3, g3);
            }
            const uint32_t l0 = wave_umax_lane63(__float_as_uint(x0)), l1 = wave_umax_lane63(__float_as_uint(x1));
            const uint32_t l2 = wave_umax_lane63(__float_as_uint(x2)), l3 = wave_umax_lane63(__float_as_uint(x3));
            const float s0 = wave_fsum_lane63(g0), s1 = wave_fsum_lane63(g1), s2 = wave_fsum_lane63(g2), s3 = wave_fsum_lane63(g3);
            if ((t & 63) == 63) {
                *reinterpret_cast<uint4 *>(&xlev4[wvid][0]) = make_uint4(l0, l1, l2, l3);
                *reinterpret_cast<uint4 *>(&xlev4[wvid][4]) = make_uint4(__float_as_uint(s0), __float_as_uint(s1), __float_as_uint(s2), __float_as_uint(s3));
            }
            __syncthreads();
            const uint4 q0 = *reinterpret_cast<const uint4 *>(&xlev4[0][0]), q1 = *reinterpret_cast<const uint4 *>(&xlev4[1][0]);
            const uint4 q2 = *reinterpret_cast<const uint4 *>(&xlev4[2][0]), q3 = *reinterpret_cast<const uint4 *>(&xlev4[3][0]);
            X[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(q0.x, q1.x), umax(q2.x, q3.x)));
            X[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(q0.y, q1.y), umax(q2.y, q3.y)));
            X[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(q0.z, q1.z), umax(q2.z, q3.z)));
            X[3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(q0.w, q1.w), umax(q2.w, q3.w)));
            const uint4 e0 = *reinterpret_cast<const uint4 *>(&xlev4[0][4]), e1 = *reinterpret_cast<const uint4 *>(&xlev4[1][4]);
            const uint4 e2 = *reinterpret_cast<const uint4 *>(&xlev4[2][4]), e3 = *reinterpret_cast<const uint4 *>(&xlev4[3][4]);
            auto fs = [](uint32_t a, uint32_t b, uint32_t c, uint32_t d) -> uint32_t {
                return (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint((__uint_as_float(a) + __uint_as_float(b)) + (__uint_as_float(c) + __uint_as_float(d))));
            };
            G[0] = fs(e0.x, e1.x, e2.x, e3.x); G[1] = fs(e0.y, e1.y, e2.y, e3.y);
            G[2] = fs(e0.z, e1.z, e2.z, e3.z); G[3] = fs(e0.w, e1.w, e2.w, e3.w);
        }
        // (energies when both are normal numbers, peaks when a square under- or overflowed)
        auto row_exp = [](uint32_t xa, uint32_t xb, uint32_t ga, uint32_t gb) -> int {
            if (xa == 0u || xb == 0u) return 0;
            const bool gok = ga != 0u && gb != 0u && ga < 0x7F800000u && gb < 0x7F800000u;
            return gok ? block_exp_energy(ga, gb) : block_exp(xa, xb);
        };
        const int E0 = row_exp(X[0], X[1], G[0], G[1]);
        const int E1 = row_exp(X[2], X[3], G[2], G[3]);
        const float sc0 = exp2i(E0), sc1 = exp2i(E1);
        // prefetch the 2*HS new slots of the next pair (consumed after the epilogue)
        float2 nx[2 * HS];
        const bool more = (w + 2 < w_end), more2 = (w + 3 < w_end);
        const float2 *nsrc = src + (size_t)(w - w_begin) * p.hop + t;
#pragma unroll
        for (int q = 0; q < 2 * HS; q++) {
            nx[q] = make_float2(0.f, 0.f);
            if (q < HS ? more : more2) nx[q] = nsrc[256 * (NS + q)];
        }

        // Every exchange is ordered  [reads] barrier [butterflies of window 0] [writes 0]
        // [butterflies of window 1] [writes 1] barrier [reads]:  the write-after-read barrier sits
        // right behind the reads, so one window's LDS writes drain while the other's butterflies issue.
        v2f z0[16], z1[16];
#pragma unroll
        for (int j = 0; j < 16; j++) z0[j] = v2f{sm[j] * hw[j], df[j] * hw[j] * sc0};
        // ---- pass 1 (the loop-end barrier has retired the previous pair's epilogue reads)
        fft16(z0);
        xbuf[0][X1W(0, tb, hi)] = z0[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) xbuf[0][X1W(ka, tb, hi)] = pk_cmul(z0[R16(ka)], tw1[ka]);
#pragma unroll
        for (int j = 0; j < 16; j++) z1[j] = v2f{sm[j + HS] * hw[j], df[j + HS] * hw[j] * sc1};
        fft16(z1);
        xbuf[1][X1W(0, tb, hi)] = z1[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) xbuf[1][X1W(ka, tb, hi)] = pk_cmul(z1[R16(ka)], tw1[ka]);
        __syncthreads();
        // ---- pass 2 (thread = tb + 16 ka)
#pragma unroll
        for (int ta = 0; ta < 16; ta++) {
            z0[ta] = xbuf[0][X1W(hi, tb, ta)];
            z1[ta] = xbuf[1][X1W(hi, tb, ta)];
        }
        __syncthreads();
        fft16(z0);
        xbuf[0][X2W(0, hi, tb)] = z0[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf[0][X2W(kb, hi, tb)] = pk_cmul(z0[R16(kb)], tw2s[tb * kb]);
        fft16(z1);
        xbuf[1][X2W(0, hi, tb)] = z1[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf[1][X2W(kb, hi, tb)] = pk_cmul(z1[R16(kb)], tw2s[tb * kb]);
        __syncthreads();
        // ---- pass 3 (thread = ka + 16 kb): ka = tb, kb = hi
#pragma unroll
        for (int q = 0; q < 16; q++) {
            z0[q] = xbuf[0][X2W(hi, tb, q)];
            z1[q] = xbuf[1][X2W(hi, tb, q)];
        }
        __syncthreads();
        fft16(z0);
        // ---- publish the whole spectrum in (swizzled) natural order: Z[t + 256 kc] at SPEC_POS(k)
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf[0][kc * 256 + tsw] = z0[R16(kc)];
        fft16(z1);
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf[1][kc * 256 + tsw] = z1[R16(kc)];
        __syncthreads();
        // ---- epilogue: groups of four consecutive bins per thread, 16-byte stores
        float *o_mid = outp + (size_t)(w - w_begin) * out_win_stride;
        fft4096_epilogue(xbuf[0], t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride, true, block_off(E0));
        if (two) fft4096_epilogue(xbuf[1], t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid + out_win_stride,
                                  o_mid + out_win_stride + p.bin_stride, true, block_off(E1));
        // a row whose WINDOWED samples are all zero reads the reference's floor (the transform of zeros, analyzer.rs:20-22)
        if (__builtin_expect(X[0] == 0u || X[1] == 0u, 0))
            fft4096_floor_rows(t, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride, X[0] == 0u, X[1] == 0u);
        if (__builtin_expect(two && (X[2] == 0u || X[3] == 0u), 0))
            fft4096_floor_rows(t, p.n_bins, p.db_offset, p.offpink, o_mid + out_win_stride, o_mid + out_win_stride + p.bin_stride, X[2] == 0u, X[3] == 0u);
        // ---- slide the sample registers by two hops
        if (more) {
#pragma unroll
            for (int j = 0; j < NS - 2 * HS; j++) { sm[j] = sm[j + 2 * HS]; df[j] = df[j + 2 * HS]; }
#pragma unroll
            for (int q = 0; q < 2 * HS; q++) {
                if (NS - 2 * HS + q >= 0) { sm[NS - 2 * HS + q] = nx[q].x + nx[q].y; df[NS - 2 * HS + q] = nx[q].x - nx[q].y; }
            }
        }
        __syncthreads();                             // epilogue reads are done before the next pair's pass-1 writes
    }
}
#undef X1W
#undef X2W

// Single-window variant (one window per iteration): built for occupancy — three (TW6: four) workgroups
// per CU instead of two.  With TW6 the 15 pass-1 twiddles W^(t ka) are rebuilt from six resident ones,
// W^(t ka) = W^(t (ka & 3)) * W^(t (ka & 12)), at the price of 9 extra complex multiplies per window.
#ifndef SS_FFT1_WAVES
#define SS_FFT1_WAVES 3
#endif
#ifndef SS_MS1_TW
#define SS_MS1_TW 12     // resident pass-1 twiddles of k_fft4096_ms1 (6, 9, 12; 0 = all fifteen): 168 VGPRs at 12, the three-waves limit
#endif
#ifndef SS_COLS_TW
#define SS_COLS_TW SS_MS1_TW      // the columns-only instantiation: the SAME count, or its rows would round differently from the stored ones
#endif
// Wave priority by phase: a wave that is exchanging through LDS (writes, barrier, reads) runs at raised priority so
// its few LDS instructions issue ahead of the other workgroups' butterflies; measured 3.15 -> 3.05 ms (A/B in one process)
#ifndef SS_FFT_PRIO
#define SS_FFT_PRIO 3
#endif
#if SS_FFT_PRIO > 0
#define SS_PRIO_HI() __builtin_amdgcn_s_setprio(SS_FFT_PRIO)
#define SS_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define SS_PRIO_HI()
#define SS_PRIO_LO()
#endif
// Development build (-DSS_FFT_PROF): per-phase shader-clock totals of k_fft4096_ms1 over all waves (tools/probe_fft_phases.py)
#ifdef SS_FFT_PROF
__device__ unsigned long long g_fft_prof[16];
#define SS_FPROF_DECL uint64_t pt_ = __builtin_amdgcn_s_memtime(); uint64_t pacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SS_FPROF_MARK(i) do { const uint64_t n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; } while (0)
#define SS_FPROF_END do { if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_fft_prof[i_], (unsigned long long)pacc_[i_]); atomicAdd(&g_fft_prof[15], 1ull); } } while (0)
#else
#define SS_FPROF_DECL
#define SS_FPROF_MARK(i)
#define SS_FPROF_END
#endif
template <int HS, int TWN, bool COLS>
__global__ __launch_bounds__(256, SS_FFT1_WAVES) void k_fft4096_ms1(FftBatchParams p)
{
    constexpr int NH = 16 / HS;                                           // hops per window
    __shared__ __attribute__((aligned(16))) v2f xbuf[16 * kPlaneB];       // 34816 B (the published spectrum uses 32768)
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];                //  2048 B
    // block exponent of the side row ("Two rows, one transform"): hoplev[hop][wave] = (largest |l + r|, largest |l - r|) of that
    // wave's lanes over one hop, as bit patterns — row 0 is rewritten for every entering hop, rows 1.. serve the run's first
    // window once; xlev is the exact path's exchange (largest windowed magnitudes of the window)
    __shared__ __attribute__((aligned(16))) uint32_t hoplev[NH][4][2];
    __shared__ __attribute__((aligned(16))) uint32_t xlev[4][2];
    // 8192 B: db_offset + pink per retained bin (n_bins <= 2047 at N = 4096).  Columns-only mode (COLS) uses the room for the
    // bins' chart columns (u16 each) and the two rows' column accumulators instead, and reads the table from global memory.
#ifdef SS_MS1_NO_LDS_TABLE      // experiment (review item 6): no table in LDS, so that FOUR workgroups fit a CU (with -DSS_FFT1_WAVES=4)
    __shared__ __attribute__((aligned(16))) float offp[4];
#else
    __shared__ __attribute__((aligned(16))) float offp[2048];
#endif
    // COLS: + 4096 B offset table (one uint2 per group of four bins) + 4128 B column accumulators [mid, side][kColStride]
    // (three workgroups per CU leave 53 KB each: 53.4 KB with these)
    __shared__ __attribute__((aligned(16))) uint2 coltab[COLS ? 512 : 1];
    __shared__ __attribute__((aligned(16))) float colbuf[COLS ? 2 * kColStride : 1];
#define X1W(ka, tb_, ta_) ((ka) * kPlaneB + (tb_) * kRowB + (ta_))
#define X2W(kb, ka_, tb_) ((kb) * kPlaneB + (ka_) * kRowB + (tb_))
    const int t = threadIdx.x;
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    const uint32_t stream = blockIdx.x / groups;
    const uint32_t grp = blockIdx.x - stream * groups;
    const uint32_t w_begin = grp * p.windows_per_block;
    uint32_t w_end = w_begin + p.windows_per_block;
    const uint32_t n_win = p.windows_of ? p.windows_of[stream] : p.n_windows;     // ragged batches: this stream's own count
    if (w_begin >= n_win) return;
    if (w_end > n_win) w_end = n_win;
    const float2 *src = reinterpret_cast<const float2 *>(p.pcm) + (size_t)stream * p.frames_per_stream +
                        p.first_start + (size_t)w_begin * p.hop;
    const v2f *twn = reinterpret_cast<const v2f *>(p.tw_n);
    // the sixteen half-window weights, two to a register pair (one packed multiply windows a frame's sum and difference)
    v2f hwp[8];
#pragma unroll
    for (int j = 0; j < 8; j++) hwp[j] = v2f{p.half_window[t + 256 * (2 * j)], p.half_window[t + 256 * (2 * j + 1)]};
    v2f tw1[16];
    constexpr bool TW6 = TWN != 0;                                        // some of the fifteen are rebuilt
    // TWN resident pass-1 twiddles: 6 = {1, 2, 3, 4, 8, 12}; 9 adds {5, 6, 7}; 12 adds {9, 10, 11} (one multiply instead of two each)
    auto tw_resident = [](int ka) -> bool { return (ka & 3) == 0 || (ka & 12) == 0 || (TWN >= 9 && (ka >> 2) == 1) || (TWN >= 12 && (ka >> 2) == 2); };
    if (TW6) {
#pragma unroll
        for (int ka = 1; ka < 16; ka++) if (tw_resident(ka)) tw1[ka] = twn[ka * t];
    } else {
#pragma unroll
        for (int ka = 1; ka < 16; ka++) tw1[ka] = twn[t * ka];
    }
    tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[(t & 15) * (t >> 4)];      // [kb][tb]: W_256^(tb kb) at kb * 16 + tb
    if (COLS) {
        for (uint32_t g = (uint32_t)t; 4u * g < p.bin_stride; g += 256u) coltab[g] = SS_COLS_FOLD ? p.col_groups[g] : p.col_bins[g];
        for (uint32_t c = (uint32_t)t; c < p.cols; c += 256u) { const float v = p.col_init[c]; colbuf[c] = v; colbuf[kColStride + c] = v; }
        if (t < 4) { colbuf[512 + t] = 0.0f; colbuf[kColStride + 512 + t] = 0.0f; }      // the spare slots the row padding folds into
    }
#ifndef SS_MS1_NO_LDS_TABLE
    for (uint32_t g = (uint32_t)t; 4u * g < p.bin_stride; g += 256u)
        reinterpret_cast<float4 *>(offp)[g] = reinterpret_cast<const float4 *>(p.offpink)[g];
#endif
    // columns-only mode: gain of this stream, and where a finished window's columns go (flushed one window late, behind the
    // loop-end barrier that closes its epilogue's atomics, by the threads that idle least: all of them, one column pair each)
    const float cgain = COLS ? (p.integrated ? -13.0f - (float)p.integrated[stream] : p.gain_db) : 0.0f;
    auto flush_columns = [&](uint32_t wdone) {
        float *oc = p.out_cols + ((size_t)stream * p.n_windows + wdone) * 2u * p.cols;
        for (uint32_t c = (uint32_t)t; c < 2u * p.cols; c += 256u) {
            const uint32_t i = c < p.cols ? c : c - p.cols + kColStride;
            const float k = colbuf[i];
            const bool none = k != k;                                 // NaN: the column owns no bin (never written, never reset)
            if (!none) colbuf[i] = -INFINITY;
            oc[c] = none ? k : fminf(fmaxf(k + cgain, -100.0f), 0.0f);
        }
    };
    const int tb = t & 15, hi = t >> 4;
    const int tsw = SPEC_POS(t);
    const size_t out_win_stride = (size_t)2 * p.bin_stride;
    float *outp = p.out + ((size_t)stream * p.n_windows + w_begin) * out_win_stride;
    // sd[j] = (l + r, (l - r) * 2^E) of frame t + 256 j  (E: the side row's block exponent; the halving of mid/side is in
    // half_window) — a register pair per frame: windowing is one packed multiply, the slide one 64-bit move
    v2f sd[16];
#pragma unroll
    for (int j = 0; j < 16; j++) sd[j] = pk_sum_diff(reinterpret_cast<const v2f *>(src)[t + 256 * j]);
    auto windowed = [&](int j) -> v2f { return (j & 1) ? pk_mul_bcast<1>(sd[j], hwp[j >> 1]) : pk_mul_bcast<0>(sd[j], hwp[j >> 1]); };
    // Levels of the first window's hops.  Hop g's wave levels go to row (g + 1) mod NH: rows 1.. are read here, row 0 — the
    // NEWEST hop — is read at the top of the window loop, where every later window finds the hop that entered at its slide.
    const uint32_t wvid = (uint32_t)__builtin_amdgcn_readfirstlane(t >> 6);
    const bool lane63 = (t & 63) == 63;
#pragma unroll
    for (int g = 0; g < NH; g++) {
        float m = 0.0f, d = 0.0f;
#pragma unroll
        for (int q = 0; q < HS; q++) { m = fmaxf(m, fabsf(sd[g * HS + q].x)); d = fmaxf(d, fabsf(sd[g * HS + q].y)); }
        const uint32_t wm = wave_umax_lane63(__float_as_uint(m)), wd = wave_umax_lane63(__float_as_uint(d));
        if (lane63) *reinterpret_cast<uint2 *>(hoplev[(g + 1) % NH][wvid]) = make_uint2(wm, wd);
    }
    __syncthreads();
    // Pm / Pd: levels of the hops of the window in the registers, oldest first; between a slide and the next loop top they
    // still stand one hop back (entry 0 is the hop that left, the newest is in LDS)
    uint32_t Pm[NH], Pd[NH];
    Pm[0] = 0u; Pd[0] = 0u;
#pragma unroll
    for (int g = 1; g < NH; g++) read_levels2(hoplev[g], Pm[g], Pd[g]);
    int E = 0;
    float scE = 1.0f, soffE = 0.0f;     // 2^E, and the dB offset that takes it out of the side row again (rewritten with E)
    // move the side halves of the registers to exponent T when that is two or more away; returns the factor applied
    auto rescale = [&](int T) -> float {
        const int dE = T - E;
        if (dE < 2 && dE > -2) return 1.0f;
        const float sc = exp2i(dE);
#pragma unroll
        for (int j = 0; j < 16; j++) sd[j].y *= sc;
        E = T; scE = exp2i(E); soffE = block_off(E);
        return sc;
    };
    // at a slide (and here, for the first window): the inner hops of the NEXT window are entries 2.. — its exponent in the
    // ordinary case, settled a whole iteration before it is needed.  What the loop top still has to ask of the entering hop is
    // left as two thresholds: `inner_ok` (both rows have inner hops, and the oldest hop is at most twice as loud) and
    // thr_m / thr_d (twice the inner level).
    bool inner_ok = false;
    uint32_t thr_m = 0u, thr_d = 0u;
    auto presettle = [&]() {
        uint32_t mm = 0u, md = 0u;
#pragma unroll
        for (int g = 2; g < NH; g++) { mm = umax(mm, Pm[g]); md = umax(md, Pd[g]); }
        thr_m = mm + 0x800000u; thr_d = md + 0x800000u;
        inner_ok = mm != 0u && md != 0u && Pm[1] <= thr_m && Pd[1] <= thr_d;
        if (mm != 0u && md != 0u) rescale(block_exp(mm, md));
    };
    presettle();
    SS_FPROF_DECL
    for (uint32_t w = w_begin; w < w_end; ++w) {
        float2 nx[HS];
        const bool more = (w + 1 < w_end);
        if (COLS && w != w_begin) flush_columns(w - 1);      // (its next atomics are four barriers away)
        // the newest hop's levels (written at the slide, behind the loop-end barrier) are requested in front of the windowing
        // multiplies, which do not wait for them: the window is formed under the exponent the slide settled on
        const uint4 lq0 = *reinterpret_cast<const uint4 *>(&hoplev[0][0][0]), lq1 = *reinterpret_cast<const uint4 *>(&hoplev[0][2][0]);
        __builtin_amdgcn_sched_barrier(0);
        v2f z[16];
#pragma unroll
        for (int j = 0; j < 16; j++) z[j] = windowed(j);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t nPm = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(lq0.x, lq0.z), umax(lq1.x, lq1.z)));
        const uint32_t nPd = (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(lq0.y, lq0.w), umax(lq1.y, lq1.w)));
#pragma unroll
        for (int g = 0; g < NH - 1; g++) { Pm[g] = Pm[g + 1]; Pd[g] = Pd[g + 1]; }
        Pm[NH - 1] = nPm; Pd[NH - 1] = nPd;
        bool zrow_m = false, zrow_d = false;                 // this window has an all-zero mid / side row
        // do the inner hops carry the level of both rows (the ordinary case: nothing to do)?
        if (__builtin_expect(!inner_ok || nPm > thr_m || nPd > thr_d, 0)) {
            uint32_t am = 0u, ad = 0u;
#pragma unroll
            for (int g = 0; g < NH; g++) { am |= Pm[g]; ad |= Pd[g]; }
            zrow_m = am == 0u; zrow_d = ad == 0u;
            if (!(zrow_m || zrow_d)) {                       // (an empty row has no level: it reads the floor, E stays)
                // an onset or a decay inside the window: exact, the windowed ENERGY of either row (see block_exp_energy)
                float xm = 0.0f, xd = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; j++) { xm = fmaf(z[j].x, z[j].x, xm); xd = fmaf(z[j].y, z[j].y, xd); }
                const float wm = wave_fsum_lane63(xm), wd = wave_fsum_lane63(xd);
                if (lane63) *reinterpret_cast<uint2 *>(xlev[wvid]) = make_uint2(__float_as_uint(wm), __float_as_uint(wd));
                __syncthreads();
                uint32_t Xm, Xd;
                read_energies2(xlev, Xm, Xd);
                if (Xm != 0u && Xd != 0u && Xm < 0x7F800000u && Xd < 0x7F800000u) {      // (a square that under- or overflowed: E stays)
                    int T = E + block_exp_energy(Xm, Xd);    // (the side halves carry 2^E already)
                    T = T < -60 ? -60 : (T > 60 ? 60 : T);
                    const float sc = rescale(T);
#pragma unroll
                    for (int j = 0; j < 16; j++) z[j].y *= sc;
                }
            }
        }
        const float soff = soffE;                            // this window's side row rides the transform as side * 2^E
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
        xbuf[X1W(0, tb, hi)] = z[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) {
            v2f v = z[R16(ka)];
            if (TW6 && !tw_resident(ka)) {
                // (the product tw1[ka & 3] tw1[ka & 12] formed per window off the data's dependent chain, then ONE multiply of the data:
                // measured in round 6, six interleaved repetitions: 2.996 vs 2.995 ms — nothing)
                v = pk_cmul(v, tw1[ka & 3]);
                v = pk_cmul(v, tw1[ka & 12]);
            } else {
                v = pk_cmul(v, tw1[ka]);
            }
            xbuf[X1W(ka, tb, hi)] = v;
        }
#pragma unroll
        for (int q = 0; q < HS; q++) {
            nx[q] = make_float2(0.f, 0.f);
            if (more) nx[q] = src[(size_t)(w + 1 - w_begin) * p.hop + t + 256 * (16 - HS + q)];
        }
        SS_FPROF_MARK(0);
        __syncthreads();
        SS_FPROF_MARK(1);
#pragma unroll
        for (int ta = 0; ta < 16; ta++) z[ta] = lds_ld64(&xbuf[X1W(hi, tb, ta)]);
        SS_FPROF_MARK(2);
        __syncthreads();
        SS_FPROF_MARK(3);
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
        xbuf[X2W(0, hi, tb)] = z[R16(0)];
        {   // second-pass twiddles from the [kb][tb] table (address = tb * 8 + an immediate), four at a time, the next four
            // requested before the current four are used
            const v2f *twp = tw2s + tb;
            v2f twa[4], twb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) twa[j] = lds_ld64(&twp[16 * (1 + j)]);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int kb0 = 1 + 4 * g, nk = g == 3 ? 3 : 4;
                if (g < 3) {
#pragma unroll
                    for (int j = 0; j < 4; j++) if (kb0 + 4 + j < 16) twb[j] = lds_ld64(&twp[16 * (kb0 + 4 + j)]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < nk; j++) xbuf[X2W(kb0 + j, hi, tb)] = pk_cmul(z[R16(kb0 + j)], twa[j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; j++) twa[j] = twb[j];
            }
        }
        SS_FPROF_MARK(4);
        __syncthreads();
        SS_FPROF_MARK(5);
#pragma unroll
        for (int q = 0; q < 16; q++) z[q] = lds_ld64(&xbuf[X2W(hi, tb, q)]);
        SS_FPROF_MARK(6);
        __syncthreads();
        SS_FPROF_MARK(7);
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
#pragma unroll
        for (int kc = 0; kc < 16; kc++)
            if ((p.publish_mask >> kc) & 1u) xbuf[kc * 256 + tsw] = z[R16(kc)];   // blocks with no retained bin or mirror are skipped
        SS_FPROF_MARK(8);
        __syncthreads();
        SS_FPROF_MARK(9);
        SS_PRIO_LO();
        // the sliding registers take the prefetched hop before the epilogue's stores (see fft4096_epilogue); its levels go to
        // the workgroup (read at the next loop top), and the next window's ordinary exponent is settled
        if (more) {
#pragma unroll
            for (int j = 0; j < 16 - HS; j++) sd[j] = sd[j + HS];
            float m = 0.0f, d = 0.0f;
#pragma unroll
            for (int q = 0; q < HS; q++) {
                const v2f v = pk_sum_diff(v2f{nx[q].x, nx[q].y});
                m = fmaxf(m, fabsf(v.x)); d = fmaxf(d, fabsf(v.y));
                sd[16 - HS + q] = v2f{v.x, v.y * scE};
            }
            const uint32_t wm = wave_umax_lane63(__float_as_uint(m)), wd = wave_umax_lane63(__float_as_uint(d));
            if (lane63) *reinterpret_cast<uint2 *>(hoplev[0][wvid]) = make_uint2(wm, wd);
            presettle();
        }
        float *o_mid = outp + (size_t)(w - w_begin) * out_win_stride;
#ifdef SS_MS1_NO_LDS_TABLE
        fft4096_epilogue<false, COLS>(xbuf, t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride, true, soff, colbuf, coltab, p.col_bins);
#else
        fft4096_epilogue<true, COLS>(xbuf, t, p.first_bin, p.n_bins, p.db_offset, offp, o_mid, o_mid + p.bin_stride, true, soff, colbuf, coltab, p.col_bins);
#endif
        if (__builtin_expect(zrow_m || zrow_d, 0)) {
            if (!COLS) fft4096_floor_rows(t, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride, zrow_m, zrow_d);
            else fft4096_floor_columns(t, p.n_bins, p.db_offset, p.offpink, colbuf, p.bin_col, p.col_init, p.cols, zrow_m, zrow_d);
        }
        SS_FPROF_MARK(10);
        __syncthreads();
        SS_FPROF_MARK(11);
    }
    SS_FPROF_END;
    if (COLS) flush_columns(w_end - 1);                     // (the loop's last barrier closed the last epilogue)
#undef X1W
#undef X2W
}

// One REAL channel (mono buffers, or channel `ch` of an interleaved one) at hop 1024: two consecutive windows w, w + 1
// ride one complex transform the way mid and side do — z[n] = (x[n] + i 2^E x[n + 1024]) hann[n] — so the "mid" row of
// the epilogue is window w and the "side" row window w + 1, and the whole of k_fft4096_ms1 carries over.  A workgroup
// walks window PAIRS; the sliding registers hold 20 slots and advance by 8 (2048 frames) per iteration.  E is the second
// window's block exponent ("Two rows, one transform": a fade-in, or a programme behind a quiet passage, puts the two
// windows at different levels); it lives in the second row's window weights hw2 = 2^E hann.
__global__ __launch_bounds__(256, SS_FFT_PAIRW_WAVES) void k_fft4096_pairw(FftBatchParams p, uint32_t fft_ch)
{
    __shared__ __attribute__((aligned(16))) v2f xbuf[16 * kPlaneB];       // 34816 B
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];                //  2048 B
    // hoplev[hop][wave] = largest |sample| of that wave's lanes over one hop (bit pattern): rows 0, 1 are rewritten for the two
    // hops that enter per pair, rows 0..4 serve the run's first pair once; xlev[wave] = (first, second) window, exact path
    __shared__ __attribute__((aligned(16))) uint32_t hoplev[5][4];
    __shared__ __attribute__((aligned(16))) uint32_t xlev[4][2];
#define X1W(ka, tb_, ta_) ((ka) * kPlaneB + (tb_) * kRowB + (ta_))
#define X2W(kb, ka_, tb_) ((kb) * kPlaneB + (ka_) * kRowB + (tb_))
    const int t = threadIdx.x;
    const uint32_t pairs_per_block = p.windows_per_block >> 1;            // the host keeps windows_per_block even
    const uint32_t n_pairs_max = (p.n_windows + 1) >> 1;
    const uint32_t groups = (n_pairs_max + pairs_per_block - 1) / pairs_per_block;
    // channels of one run read the same interleaved lines: keep them on one XCD (the mapping is spelled out in k_fft16k_run)
    const uint32_t per_xcd = gridDim.x >> 3;
    uint32_t bid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (bid >= p.n_streams * groups * fft_ch) return;
    const uint32_t ch = bid % fft_ch; bid /= fft_ch;
    const uint32_t grp = bid % groups;
    const uint32_t stream = bid / groups;
    const uint32_t n_win = p.windows_of ? p.windows_of[stream] : p.n_windows;
    const uint32_t n_pairs = (n_win + 1) >> 1;
    const uint32_t pp_begin = grp * pairs_per_block;
    uint32_t pp_end = pp_begin + pairs_per_block;
    if (pp_begin >= n_pairs) return;
    if (pp_end > n_pairs) pp_end = n_pairs;
    const uint32_t C = p.channels;
    const float *src = p.pcm + ((size_t)stream * p.frames_per_stream + p.first_start + (size_t)pp_begin * 2048u) * C + ch;
    const v2f *twn = reinterpret_cast<const v2f *>(p.tw_n);
    float hw[16], hw2[16];
#pragma unroll
    for (int j = 0; j < 16; j++) hw2[j] = hw[j] = p.window[t + 256 * j];  // the full Hann window: no mid/side halving here
    v2f tw1[16];
    tw1[1] = twn[t]; tw1[2] = twn[2 * t]; tw1[3] = twn[3 * t];
    tw1[4] = twn[4 * t]; tw1[8] = twn[8 * t]; tw1[12] = twn[12 * t];
    tw1[5] = twn[5 * t]; tw1[6] = twn[6 * t]; tw1[7] = twn[7 * t];               // three more resident (k_fft4096_ms1's TWN = 9; round 6: -3 %,
                                                                                 // 166 VGPRs; twelve would cost the third wave per SIMD)
    tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[(t & 15) * (t >> 4)];      // [kb][tb]: W_256^(tb kb) at kb * 16 + tb (see k_fft4096_ms1)
    const int tb = t & 15, hi = t >> 4;
    const int tsw = SPEC_POS(t);
    const size_t row_stride = (size_t)fft_ch * p.bin_stride;              // window w -> window w + 1 of the same channel
    float *outp = p.out + (((size_t)stream * p.n_windows + 2u * pp_begin) * fft_ch + ch) * p.bin_stride;
    // slot s holds frame t + 256 s of the current pair's first window; the second window is slots 4..19
    float raw[20];
    const bool two0 = (2u * pp_begin + 1u < n_win);
#pragma unroll
    for (int j = 0; j < 20; j++) raw[j] = (j < 16 || two0) ? src[(size_t)(t + 256 * j) * C] : 0.0f;
    // levels of the five hops of the first pair (the first window is hops 0..3, the second hops 1..4); an all-zero window —
    // every level 0 — is the zero-row case of fft4096_floor_rows
    const uint32_t wvid = (uint32_t)__builtin_amdgcn_readfirstlane(t >> 6);
    const bool lane63 = (t & 63) == 63;
    // (largest |.| as a bit pattern, taken with INTEGER maxima: a NaN — which fmaxf would drop — and an infinity both read
    // >= 0x7F800000, and that is how a window the reference refuses is recognised below)
    auto hop_level = [&](float a, float b, float c, float d) -> uint32_t {
        const uint32_t ua = __float_as_uint(a) & 0x7FFFFFFFu, ub = __float_as_uint(b) & 0x7FFFFFFFu;
        const uint32_t uc = __float_as_uint(c) & 0x7FFFFFFFu, ud = __float_as_uint(d) & 0x7FFFFFFFu;
        return wave_umax_lane63(umax(umax(ua, ub), umax(uc, ud)));
    };
    auto read_level = [&](const uint32_t *lv) -> uint32_t {
        const uint4 q = *reinterpret_cast<const uint4 *>(lv);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)umax(umax(q.x, q.y), umax(q.z, q.w)));
    };
#pragma unroll
    for (int g = 0; g < 5; g++) {
        const uint32_t wl = hop_level(raw[4 * g], raw[4 * g + 1], raw[4 * g + 2], raw[4 * g + 3]);
        if (lane63) hoplev[g][wvid] = wl;
    }
    __syncthreads();
    uint32_t P[5];
#pragma unroll
    for (int g = 0; g < 5; g++) P[g] = read_level(hoplev[g]);
    int E = 0;
    // block exponent of the pair now in the registers; hw2 is rewritten when it moves by two or more
    auto settle = [&]() {
        if ((P[0] | P[1] | P[2] | P[3]) == 0u || (P[1] | P[2] | P[3] | P[4]) == 0u) return;      // an empty window has no level
        if (umax(umax(umax(P[0], P[1]), umax(P[2], P[3])), P[4]) >= 0x7F800000u) return;          // ... nor has a refused one (E stays)
        const uint32_t ma = umax(P[1], P[2]), ea = umax(P[0], P[3]), mb = umax(P[2], P[3]), eb = umax(P[1], P[4]);
        int T;
        if (ma != 0u && mb != 0u && ea <= ma + 0x800000u && eb <= mb + 0x800000u) {
            T = block_exp(ma, mb);
        } else {
            float xa = 0.0f, xb = 0.0f;                     // exact: the windowed energy of either window (see block_exp_energy)
#pragma unroll
            for (int j = 0; j < 16; j++) { const float pa = raw[j] * hw[j], pb = raw[j + 4] * hw[j]; xa = fmaf(pa, pa, xa); xb = fmaf(pb, pb, xb); }
            const float wa = wave_fsum_lane63(xa), wb = wave_fsum_lane63(xb);
            if (lane63) *reinterpret_cast<uint2 *>(xlev[wvid]) = make_uint2(__float_as_uint(wa), __float_as_uint(wb));
            __syncthreads();
            uint32_t Xa, Xb;
            read_energies2(xlev, Xa, Xb);
            if (Xa == 0u || Xb == 0u || Xa >= 0x7F800000u || Xb >= 0x7F800000u) return;      // (a square that under- or overflowed: E stays)
            T = block_exp_energy(Xa, Xb);
        }
        const int dE = T - E;
        if (dE >= 2 || dE <= -2) {
            const float sc = exp2i(T);
#pragma unroll
            for (int j = 0; j < 16; j++) hw2[j] = hw[j] * sc;
            E = T;
        }
    };
    settle();
    for (uint32_t pp = pp_begin; pp < pp_end; ++pp) {
        const bool two = (2u * pp + 1u < n_win);
        const bool more = (pp + 1 < pp_end);
        const bool more2 = more && (2u * pp + 3u < n_win);
        const bool zrow_a = (P[0] | P[1] | P[2] | P[3]) == 0u, zrow_b = (P[1] | P[2] | P[3] | P[4]) == 0u;     // THIS pair's windows
        float nx[8];
        const float *nsrc = src + ((size_t)(pp - pp_begin) * 2048u + t) * C;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            nx[q] = 0.0f;
            if (q < 4 ? more : more2) nx[q] = nsrc[(size_t)(256 * (20 + q)) * C];
        }
        // A window with a NaN or an infinite sample is refused by the reference; its partner in the transform — three quarters the
        // same samples, but possibly not the one that matters — is not, and must not inherit the poison: the refused window stays
        // OUT of the transform (zeros) and its row is written as NaN behind the epilogue.  Wave-uniform and rare.
        const bool bad_a = umax(umax(P[0], P[1]), umax(P[2], P[3])) >= 0x7F800000u;
        const bool bad_b = umax(umax(P[1], P[2]), umax(P[3], P[4])) >= 0x7F800000u;
        v2f z[16];
#pragma unroll
        for (int j = 0; j < 16; j++) z[j] = v2f{raw[j] * hw[j], raw[j + 4] * hw2[j]};
        if (__builtin_expect(bad_a != bad_b, 0)) {
#pragma unroll
            for (int j = 0; j < 16; j++) { if (bad_a) z[j].x = 0.0f; else z[j].y = 0.0f; }
        }
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
        xbuf[X1W(0, tb, hi)] = z[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) {
            v2f v = z[R16(ka)];
            if ((ka >> 2) == 1) v = pk_cmul(v, tw1[ka]);
            else {
                if (ka & 3) v = pk_cmul(v, tw1[ka & 3]);
                if (ka & 12) v = pk_cmul(v, tw1[ka & 12]);
            }
            xbuf[X1W(ka, tb, hi)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ta = 0; ta < 16; ta++) z[ta] = lds_ld64(&xbuf[X1W(hi, tb, ta)]);
        __syncthreads();
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
        xbuf[X2W(0, hi, tb)] = z[R16(0)];
        {
            const v2f *twp = tw2s + tb;
            v2f twa[4], twb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) twa[j] = lds_ld64(&twp[16 * (1 + j)]);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int kb0 = 1 + 4 * g, nk = g == 3 ? 3 : 4;
                if (g < 3) {
#pragma unroll
                    for (int j = 0; j < 4; j++) if (kb0 + 4 + j < 16) twb[j] = lds_ld64(&twp[16 * (kb0 + 4 + j)]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < nk; j++) xbuf[X2W(kb0 + j, hi, tb)] = pk_cmul(z[R16(kb0 + j)], twa[j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; j++) twa[j] = twb[j];
            }
        }
        {   // the two entering hops' levels to the workgroup (read back behind the next barrier, used at the slide)
            const uint32_t wl0 = hop_level(nx[0], nx[1], nx[2], nx[3]), wl1 = hop_level(nx[4], nx[5], nx[6], nx[7]);
            if (lane63) { hoplev[0][wvid] = wl0; hoplev[1][wvid] = wl1; }
        }
        __syncthreads();
        const uint32_t nP0 = read_level(hoplev[0]), nP1 = read_level(hoplev[1]);
#pragma unroll
        for (int q = 0; q < 16; q++) z[q] = lds_ld64(&xbuf[X2W(hi, tb, q)]);
        __syncthreads();
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
#pragma unroll
        for (int kc = 0; kc < 16; kc++)
            if ((p.publish_mask >> kc) & 1u) xbuf[kc * 256 + tsw] = z[R16(kc)];
        __syncthreads();
        SS_PRIO_LO();
        float *o_first = outp + (size_t)(pp - pp_begin) * 2u * row_stride;
        fft4096_epilogue(xbuf, t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_first, o_first + row_stride, two, block_off(E));
        if (__builtin_expect(zrow_a || (two && zrow_b), 0))
            fft4096_floor_rows(t, p.n_bins, p.db_offset, p.offpink, o_first, o_first + row_stride, zrow_a, two && zrow_b);
        if (__builtin_expect(bad_a || bad_b, 0))
            fft4096_nan_rows(t, p.n_bins, o_first, o_first + row_stride, bad_a, two && bad_b);
        if (more) {
#pragma unroll
            for (int j = 0; j < 12; j++) raw[j] = raw[j + 8];
#pragma unroll
            for (int q = 0; q < 8; q++) raw[12 + q] = nx[q];
            P[0] = P[2]; P[1] = P[3]; P[2] = P[4]; P[3] = nP0; P[4] = nP1;
            settle();
        }
        __syncthreads();
    }
#undef X1W
#undef X2W
}

hipError_t launch_fft4096_pairw(const FftBatchParams &p, int mode, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0) return hipSuccess;
    const uint32_t fft_ch = (mode == 0) ? 1u : p.channels;
    const uint32_t pairs_per_block = p.windows_per_block >> 1;
    const uint32_t n_pairs = (p.n_windows + 1) >> 1;
    const uint32_t groups = (n_pairs + pairs_per_block - 1) / pairs_per_block;
    const uint64_t total = (uint64_t)p.n_streams * groups * fft_ch;
    hipLaunchKernelGGL(k_fft4096_pairw, dim3((uint32_t)((total + 7) & ~(uint64_t)7)), dim3(256), 0, s, p, fft_ch);
    return hipGetLastError();
}

// generic hop (not a multiple of 256 or >= N/2 slots): one window per iteration, full reload
__global__ __launch_bounds__(256, SS_FFT_WAVES) void k_fft4096_ms_anyhop(FftBatchParams p)
{
    __shared__ __attribute__((aligned(16))) v2f xbuf[16 * kX1Stride];
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];
    __shared__ __attribute__((aligned(16))) uint32_t xlev[4][2];      // exact block exponent: [wave][mid, side] largest windowed magnitude
    __shared__ __attribute__((aligned(16))) uint32_t glev[4][2];      //                        ... and windowed energy
    const int t = threadIdx.x;
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    const uint32_t stream = blockIdx.x / groups;
    const uint32_t grp = blockIdx.x - stream * groups;
    const uint32_t w_begin = grp * p.windows_per_block;
    uint32_t w_end = w_begin + p.windows_per_block;
    const uint32_t n_win = p.windows_of ? p.windows_of[stream] : p.n_windows;     // ragged batches: this stream's own count
    if (w_begin >= n_win) return;
    if (w_end > n_win) w_end = n_win;
    const float2 *src = reinterpret_cast<const float2 *>(p.pcm) + (size_t)stream * p.frames_per_stream +
                        p.first_start + (size_t)w_begin * p.hop;
    tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[t];
    const int tb = t & 15, hi = t >> 4;
    const size_t out_win_stride = (size_t)2 * p.bin_stride;
    float *outp = p.out + ((size_t)stream * p.n_windows + w_begin) * out_win_stride;
    const uint32_t wvid = (uint32_t)__builtin_amdgcn_readfirstlane(t >> 6);
    __syncthreads();
    for (uint32_t w = w_begin; w < w_end; ++w) {
        v2f z[16];
        float xm = 0.0f, xd = 0.0f, gm = 0.0f, gd = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float2 v = src[(size_t)(w - w_begin) * p.hop + t + 256 * j];
            const float hwj = p.half_window[t + 256 * j];
            z[j] = v2f{(v.x + v.y) * hwj, (v.x - v.y) * hwj};
            xm = fmaxf(xm, fabsf(z[j].x)); xd = fmaxf(xd, fabsf(z[j].y));
            gm = fmaf(z[j].x, z[j].x, gm); gd = fmaf(z[j].y, z[j].y, gd);
        }
        // exact block exponent of the side row ("Two rows, one transform"); the previous window's reads of xlev lie behind
        // the barriers of its transform
        uint32_t Xm, Xd, Gm, Gd;
        {
            const uint32_t wm = wave_umax_lane63(__float_as_uint(xm)), wd = wave_umax_lane63(__float_as_uint(xd));
            const float sm_ = wave_fsum_lane63(gm), sd_ = wave_fsum_lane63(gd);
            if ((t & 63) == 63) {
                *reinterpret_cast<uint2 *>(xlev[wvid]) = make_uint2(wm, wd);
                *reinterpret_cast<uint2 *>(glev[wvid]) = make_uint2(__float_as_uint(sm_), __float_as_uint(sd_));
            }
            __syncthreads();
            read_levels2(xlev, Xm, Xd);
            read_energies2(glev, Gm, Gd);
        }
        // energies (see block_exp_energy) when both are normal numbers, peaks when a square under- or overflowed
        const bool gok = Gm != 0u && Gd != 0u && Gm < 0x7F800000u && Gd < 0x7F800000u;
        const int E = (Xm != 0u && Xd != 0u) ? (gok ? block_exp_energy(Gm, Gd) : block_exp(Xm, Xd)) : 0;
        {
            const float sc = exp2i(E);
#pragma unroll
            for (int j = 0; j < 16; j++) z[j].y *= sc;
        }
        fft16(z);
        __syncthreads();
        xbuf[t] = z[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) xbuf[ka * kX1Stride + t] = pk_cmul(z[R16(ka)], reinterpret_cast<const v2f *>(p.tw_n)[t * ka]);
        __syncthreads();
#pragma unroll
        for (int ta = 0; ta < 16; ta++) z[ta] = xbuf[hi * kX1Stride + tb + 16 * ta];
        fft16(z);
        __syncthreads();
        xbuf[hi * kX2Stride + tb] = z[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf[kb * kX1Stride + hi * kX2Stride + tb] = pk_cmul(z[R16(kb)], tw2s[tb * kb]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) z[q] = xbuf[hi * kX1Stride + tb * kX2Stride + q];
        fft16(z);
        __syncthreads();
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf[kc * 256 + SPEC_POS(t)] = z[R16(kc)];
        __syncthreads();
        float *o_mid = outp + (size_t)(w - w_begin) * out_win_stride;
        fft4096_epilogue(xbuf, t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride, true, block_off(E));
        if (__builtin_expect(Xm == 0u || Xd == 0u, 0))
            fft4096_floor_rows(t, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride, Xm == 0u, Xd == 0u);
    }
}

// ============================================================================
//  Spectrum, N = 16384 (the reference's native window, tui.rs:1488), one REAL channel per
//  512-thread workgroup: real FFT through an 8192-point complex FFT, split by one radix-2
//  decimation-in-frequency step into two 4096-point problems that reuse the radix-16 machinery:
//    z[i] = (xw[2i], xw[2i+1]),  y_q[i] = (z[i] + (-1)^q z[i+4096]) W_8192^(i q),  Z[2k+q] = FFT_4096(y_q)[k]
//    X[b] = (Z[b] + conj Z[8192-b])/2 - (i/2) W_16384^b (Z[b] - conj Z[8192-b])
//  Threads 0-255 run q = 0, threads 256-511 run q = 1; the mirror 8192-b has the parity of b, so each
//  half only mirrors inside its own published spectrum.  mode 0: mono buffer, 1: stereo -> mid/side
//  (audio_player.rs:400-419), 2: channel `ch` of an interleaved buffer.
// ============================================================================
__global__ __launch_bounds__(512, 2) void k_fft16k(FftBatchParams p, int midside, uint32_t fft_ch)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kFft16kLdsBytes];
    fft16k_window(p, midside, fft_ch, blockIdx.x, lds);     // (ss_fft_dev.h: the tick kernel of ss_time_domain.hip runs the same body)
}

// ============================================================================
//  Spectrum, N = 16384 at hop 1024 for batches: one REAL channel per 512-thread workgroup that walks
//  `windows_per_block` consecutive windows with sliding sample registers (hop 1024 samples = one slot,
//  so a sample is fetched once per run instead of 16 times) and window weights rebuilt from two
//  resident twiddles, w[n0 + 1024 j] = 1/2 - 1/2 cos(a0 + j pi/8).
//  Decimation in time by four:  X[b] = A0[b] + W^b A1[b] + W^2b A2[b] + W^3b A3[b],  W = W_16384,
//  A_r = FFT_4096 of the real sequence xw[4i + r].  Half q transforms the complex sequence
//  z_q[i] = (xw[4i + 2q], xw[4i + 2q + 1]) on the radix-16 passes of k_fft4096_ms1; A_{2q}, A_{2q+1} are its even / odd
//  parts (the mid/side split of the N = 4096 kernel), combined per bin by Horner.  The two halves never exchange data
//  before the epilogue.  Both halves are the SAME signal (its even / odd sample pairs), so the two rows of a transform are
//  at one level by construction: no block exponent here.  MODE 0: mono buffer or channel `ch` of an interleaved buffer,
//  1: stereo -> mid/side (audio_player.rs:400-419), one workgroup per signal.
//
//  ONE half per thread: threads 0-255 carry half q = 0 (samples 4i, 4i + 1), threads 256-511 half q = 1 (samples 4i + 2,
//  4i + 3).  Through round 3 every one of 256 threads carried both halves (250 VGPRs, two workgroups = two waves per SIMD);
//  spread over eight waves a window's 64 KB of sliding samples cost 32 registers per thread, the kernel fits 128 VGPRs and
//  two workgroups are FOUR waves per SIMD: config 5 7.11 -> 6.89 ms, native stereo 4.66 -> 4.53 ms in one call
//  (profiles/r04_ab_fft16k_eight_waves.txt).  q is wave-uniform (a wave belongs to one half), so nothing diverges.
// ============================================================================
#ifndef SS_RUN8_WAVES
#define SS_RUN8_WAVES 4
#endif
template <bool MIDSIDE, int NE_LAST>
__global__ __launch_bounds__(512, SS_RUN8_WAVES) void k_fft16k_run(FftBatchParams p, uint32_t fft_ch)
{
    __shared__ __attribute__((aligned(16))) v2f xbuf2[2][16 * kPlaneB];      // 2 x 34816 B
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];                    //  2048 B
    __shared__ __attribute__((aligned(16))) float stage[8][256];              //  8192 B: 79872 B per workgroup, two per CU
#define X1W(ka, tb_, ta_) ((ka) * kPlaneB + (tb_) * kRowB + (ta_))
#define X2W(kb, ka_, tb_) ((kb) * kPlaneB + (ka_) * kRowB + (tb_))
    const int q = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // which half this wave carries
    const int t = threadIdx.x & 255;
    v2f *xbuf = xbuf2[q];
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    // Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  The channels of one run read the
    // same interleaved lines, so they must sit on ONE XCD: logical id = (id mod 8) * ceil(total / 8) + id / 8 makes
    // the ids of an XCD consecutive (the launcher pads the grid to a multiple of 8; surplus ids leave).
    const uint32_t per_xcd = gridDim.x >> 3;
    uint32_t bid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (bid >= p.n_streams * groups * fft_ch) return;
    const uint32_t ch = bid % fft_ch; bid /= fft_ch;            // the channels of one run are neighbours in the logical order
    const uint32_t grp = bid % groups;
    const uint32_t stream = bid / groups;
    const uint32_t w_begin = grp * p.windows_per_block;
    uint32_t w_end = w_begin + p.windows_per_block;
    const uint32_t n_win = p.windows_of ? p.windows_of[stream] : p.n_windows;     // ragged batches: this stream's own count
    if (w_begin >= n_win) return;
    if (w_end > n_win) w_end = n_win;
    const uint32_t C = p.channels;
    const float *base = p.pcm + ((size_t)stream * p.frames_per_stream + p.first_start + (size_t)w_begin * 1024u) * C;
    const v2f *tw16k = reinterpret_cast<const v2f *>(p.tw_n);       // W_16384^k, k < 8192
    const v2f *tw4k = reinterpret_cast<const v2f *>(p.tw_core);     // W_4096^k
    if (threadIdx.x < 256) tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[(t & 15) * (t >> 4)];      // [kb][tb]: W_256^(tb kb) at kb * 16 + tb

    // samples n + 2q, n + 2q + 1 of this workgroup's channel (n relative to the run's first window): one complex input of half q
    auto ld2 = [&](size_t n) -> v2f {
        if (MIDSIDE) {
            const float2 *f = reinterpret_cast<const float2 *>(base) + n + 2 * q;
            const float2 a = f[0], b = f[1];
            return ch == 0 ? v2f{(a.x + a.y) * 0.5f, (b.x + b.y) * 0.5f} : v2f{(a.x - a.y) * 0.5f, (b.x - b.y) * 0.5f};
        }
        const float *f = base + (n + 2 * q) * C + ch;
        return v2f{f[0], f[C]};
    };
    const uint32_t n0 = 4u * (uint32_t)t;           // slot j holds samples n0 + 1024 j + 2q, + 1 of the current window
    v2f raw[16];
#pragma unroll
    for (int j = 0; j < 16; j++) raw[j] = ld2((size_t)n0 + 1024u * j);
    // Hann weights of slot j: angle a_e + j pi/8, a_e = 2 pi (n0 + 2q + e) / 16384, e = 0, 1 (table holds (cos, -sin)) ...
    // ... except for the two EDGE slots (0 and 15), whose weights are small: rebuilt in f32 they are within 2.4e-7 of the
    // crate's table, which is nothing for a weight of 0.5 but 1e-4 of a weight of 2e-3 — and a window whose only loud
    // samples sit in its last hop (the first window after a near-silent passage) has nothing but such weights under its
    // energy: 0.012 dB against the oracle there.  Those weights come from the table itself (bit-equal to the crate's).
    const uint32_t ne = n0 + 2u * (uint32_t)q;
    const v2f wa = tw16k[ne], wb = tw16k[ne + 1];
    const v2f hc = {-0.5f * wa.x, -0.5f * wb.x}, hs = {-0.5f * wa.y, -0.5f * wb.y};   // -1/2 cos a_e, +1/2 sin a_e
    const v2f half = {0.5f, 0.5f};
    const v2f we0 = {p.window[ne], p.window[ne + 1]}, we15 = {p.window[15360u + ne], p.window[15360u + ne + 1]};
    v2f twg[16];
    twg[1] = tw4k[t]; twg[2] = tw4k[2 * t]; twg[3] = tw4k[3 * t];
    twg[4] = tw4k[4 * t]; twg[8] = tw4k[8 * t]; twg[12] = tw4k[12 * t];
    const int tb = t & 15, hi = t >> 4;
    const uint32_t ngroups = (p.n_bins + 3) >> 2;
    constexpr float kDb = 3.01029995663981195f;
    const float off2 = p.db_offset - 6.02059991327962390f;      // the epilogue carries 2 X
    // Epilogue geometry.  An iteration covers 2048 retained bins: wave wv the 256 starting at 256 wv, as four slices of 64 (one bin
    // per lane and slice).  The LAST iteration covers what is left, R = 4 ngroups - 2048 (n_iter - 1) bins, with ne_last =
    // ceil(R / 512) slices per wave instead of four — 3410 bins at 96 kHz: 2048 + 3 x 512, 6820 at 48 kHz: 3 x 2048 + 2 x 512 — so
    // an eighth of the four-term recombinations is no longer computed for bins nobody keeps, and every wave (hence every SIMD)
    // saves the same share.  Its bins sit 2048 - 64 (4 - ne_last) wv behind the previous iteration's, not 2048: the twiddles of
    // that iteration take one more turn by the wave-uniform constant wlast = W_16384^(-64 (4 - ne_last) wv).
    const uint32_t n_iter = (4u * ngroups + 2047u) >> 11;                  // (the launcher instantiates NE_LAST = ne_last)
    v2f wlast;
    {
        const uint32_t wv0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const v2f wl = tw16k[(64u * (4u - (uint32_t)NE_LAST) * wv0) & 8191u];
        wlast = v2f{uniform_f(wl.x), uniform_f(-wl.y)};                // conjugate: a turn backwards
    }
    // Everything loaded above is "used" once here, in front of the window loop.  A register whose load is still pending at
    // the loop's entry gets its s_waitcnt at its first use INSIDE the loop, and vmcnt counts stores too: in every later
    // iteration that wait — vmcnt(0) in the middle of the first radix pass — stood there for the PREVIOUS window's output
    // stores to reach memory.
    asm volatile("" : "+v"(twg[1]), "+v"(twg[2]), "+v"(twg[3]), "+v"(twg[4]), "+v"(twg[8]), "+v"(twg[12]));
    asm volatile("" ::"v"(we0), "v"(we15));
#pragma unroll
    for (int j = 0; j < 16; j++) asm volatile("" : "+v"(raw[j]));
    __syncthreads();
    SS_FPROF_DECL      // (-DSS_FFT_PROF: the window loop's phases, tools/probe_fft16k_phases.py)

    for (uint32_t w = w_begin; w < w_end; ++w) {
        const bool more = (w + 1 < w_end);
        v2f nx = {0.0f, 0.0f};
        v2f z[16];
        constexpr float cj[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                                  -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                                  -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f,
                                  0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
        constexpr float sj[16] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f,
                                  0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                                  -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                                  -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
        // (the fourteen rebuilt weights are window-loop invariants: left alone the compiler computes them once, keeps 28 registers
        // for them across the loop and — at the 128 registers of four waves per SIMD — spills and reloads them every window;
        // opaque copies of the two seeds keep the rebuild, 28 packed fma, inside the loop)
        v2f hcw = hc, hsw = hs;
        asm volatile("" : "+v"(hcw), "+v"(hsw));
        // Slots j and j + 8 sit half a period apart: w[j + 8] = 1 - w[j], so x w[j + 8] = x - x w[j] is one fma on the weight
        // its partner has built.  The SMALLER weight of a pair is the one that is built (slots 1..3 and 12..14; the edge slots
        // 0 and 15 come from the table), the larger one — 0.5 or more, where an ulp of 1 does not matter — is derived:
        // 28 packed instructions for the sixteen products instead of 44.
        z[0] = raw[0] * we0;   z[8] = raw[8] - raw[8] * we0;
        z[15] = raw[15] * we15; z[7] = raw[7] - raw[7] * we15;
#pragma unroll
        for (int j = 1; j <= 3; j++) {
            const v2f wlo = half + hcw * cj[j] + hsw * sj[j], whi = half + hcw * cj[j + 11] + hsw * sj[j + 11];      // slots j and j + 11 = 12..14
            z[j] = raw[j] * wlo;           z[j + 8] = raw[j + 8] - raw[j + 8] * wlo;
            z[j + 11] = raw[j + 11] * whi; z[j + 3] = raw[j + 3] - raw[j + 3] * whi;
        }
        fft16(z);
        xbuf[X1W(0, tb, hi)] = z[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) {
            v2f v = z[R16(ka)];
            if (ka & 3) v = pk_cmul(v, twg[ka & 3]);
            if (ka & 12) v = pk_cmul(v, twg[ka & 12]);
            xbuf[X1W(ka, tb, hi)] = v;
        }
        if (more) nx = ld2((size_t)(w + 1 - w_begin) * 1024u + n0 + 1024u * 15u);
        SS_FPROF_MARK(0);
        __syncthreads();
        SS_FPROF_MARK(1);
#pragma unroll
        for (int ta = 0; ta < 16; ta++) z[ta] = lds_ld64(&xbuf[X1W(hi, tb, ta)]);
        SS_FPROF_MARK(2);
        __syncthreads();
        SS_FPROF_MARK(3);
        fft16(z);
        xbuf[X2W(0, hi, tb)] = z[R16(0)];
        {   // second-pass twiddles from the [kb][tb] table, four at a time, the next four requested before the current four are used
            const v2f *twp = tw2s + tb;
            v2f twa[4], twb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) twa[j] = lds_ld64(&twp[16 * (1 + j)]);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int kb0 = 1 + 4 * g, nk = g == 3 ? 3 : 4;
                if (g < 3) {
#pragma unroll
                    for (int j = 0; j < 4; j++) if (kb0 + 4 + j < 16) twb[j] = lds_ld64(&twp[16 * (kb0 + 4 + j)]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < nk; j++) xbuf[X2W(kb0 + j, hi, tb)] = pk_cmul(z[R16(kb0 + j)], twa[j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; j++) twa[j] = twb[j];
            }
        }
        SS_FPROF_MARK(4);
        __syncthreads();
        SS_FPROF_MARK(5);
#pragma unroll
        for (int qq = 0; qq < 16; qq++) z[qq] = lds_ld64(&xbuf[X2W(hi, tb, qq)]);
        SS_FPROF_MARK(6);
        __syncthreads();
        SS_FPROF_MARK(7);
        fft16(z);
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf[kc * 256 + t] = z[R16(kc)];          // Z_q[k] at k (natural order)
        // the epilogue's twiddles are requested in front of the barrier that closes the publish
        const uint32_t lane = (uint32_t)threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        v2f wt[4];
        {
            uint32_t fbq = p.first_bin;
            asm volatile("" : "+s"(fbq));
#pragma unroll
            for (int e = 0; e < 4; e++) wt[e] = tw16k[(fbq + 256u * wv + 64u * e + lane) & 8191u];
        }
        SS_FPROF_MARK(8);
        __syncthreads();
        SS_FPROF_MARK(9);
        asm volatile("" : "+v"(nx));        // the prefetched hop is claimed in front of the epilogue's stores

        // ---- epilogue, all eight waves (both spectra are published).  Iteration `it` covers 2048 retained bins, wave wv the 256
        // of them starting at 2048 it + 256 wv.  Reading: lane l takes bins +l, +64+l, +128+l, +192+l, so every LDS read of the
        // two published spectra (and of their mirrors, descending) is stride-1 across the wave: conflict-free whatever
        // first_bin is.  Writing: the four dB values go through a wave-private 1 KB staging row so that lane l stores bins
        // +4l..+4l+3 with one 16-byte store.  From one iteration to the next the bin index grows by 2048: positions move by
        // +-2048, twiddles turn by W_16384^2048 = W_8^1.
        float *o = p.out + (((size_t)stream * p.n_windows + w) * fft_ch + ch) * p.bin_stride;
        {
            float *stg = stage[wv];
            uint32_t fb = p.first_bin;
            asm volatile("" : "+s"(fb));
            const v2f rho = {0.70710678118654752f, -0.70710678118654752f};
            asm volatile("" : "+v"(wt[0]), "+v"(wt[1]), "+v"(wt[2]), "+v"(wt[3]));
            // one iteration over NE slices of 64 bins per wave (NE = 4 but for the last one, where it is the kernel's NE_LAST)
            auto iteration = [&](auto ne_c, const uint32_t it, const bool last_it) {
                constexpr int NE = decltype(ne_c)::value;
                const uint32_t g = 512u * it + 16u * NE * wv + lane;   // group of four bins this lane stores
                const bool g_ok = (NE == 4 || lane < 16u * NE) && g < ngroups;
                float4 pk = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.pink) pk = *reinterpret_cast<const float4 *>(p.pink + 4u * (g < ngroups ? g : ngroups - 1u));
                const uint32_t b0 = fb + 2048u * it + 64u * NE * wv + lane;
                v2f e0v[NE], emv[NE], o0v[NE], omv[NE];
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const uint32_t pb = (b0 + 64u * e) & 4095u, pm = (4096u - pb) & 4095u;
                    e0v[e] = xbuf2[0][pb]; emv[e] = xbuf2[0][pm];
                    o0v[e] = xbuf2[1][pb]; omv[e] = xbuf2[1][pm];
                }
                __builtin_amdgcn_sched_barrier(0);
                float r[4] = {0.f, 0.f, 0.f, 0.f}, qv[NE];
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const v2f e0 = e0v[e], em = emv[e], o0 = o0v[e], om = omv[e];
                    v2f a0, a1, a2, a3;        // 2 A_r[b]
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(a0) : "v"(e0), "v"(em));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(a1) : "v"(e0), "v"(em));
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(a2) : "v"(o0), "v"(om));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(a3) : "v"(o0), "v"(om));
                    // narrower slices: the bins moved by 2048 - 64 (4 - NE) wv since the previous iteration, not by 2048
                    // (twiddles straight from the table for every iteration instead of the turn by W_8 per iteration: measured in round 6
                    // on a row at the edge of the spectrum metric — its worst bin 0.025 -> 0.016 dB from the f64 result, the other
                    // windows of that stream unchanged, +1.5 % kernel time: the turn stays)
                    const v2f wte = NE != 4 ? pk_cmul(wt[e], wlast) : wt[e];
                    v2f x = pk_cmul(a3, wte) + a2;
                    x = pk_cmul(x, wte) + a1;
                    x = pk_cmul(x, wte) + a0;
                    qv[e] = fmaf(x.x, x.x, x.y * x.y);
                }
                // an exact zero reads -150 (analyzer.rs:20-22): rare, so the wave first asks whether any of its magnitudes is zero
                float qmin = qv[0];
#pragma unroll
                for (int e = 1; e < NE; e++) qmin = fminf(qmin, qv[e]);
                if (__builtin_expect(__ballot(qmin == 0.0f) == 0ull, 1)) {
#pragma unroll
                    for (int e = 0; e < NE; e++) r[e] = fmaf(__log2f(qv[e]), kDb, off2);
                } else {
#pragma unroll
                    for (int e = 0; e < NE; e++) r[e] = qv[e] == 0.0f ? -150.0f : fmaf(__log2f(qv[e]), kDb, off2);
                }
                if (!last_it) {                                        // (the last iteration's turn would not be used)
#pragma unroll
                    for (int e = 0; e < 4; e++) wt[e] = pk_cmul(wt[e], rho);
                }
#pragma unroll
                for (int e = 0; e < NE; e++) stg[64 * e + lane] = r[e];
                __builtin_amdgcn_wave_barrier();                      // LDS is in order per wave: ordering is all that is needed
                const float4 v = reinterpret_cast<const float4 *>(stg)[lane];
                __builtin_amdgcn_wave_barrier();
                if (g_ok) reinterpret_cast<float4 *>(o)[g] = make_float4(v.x + pk.x, v.y + pk.y, v.z + pk.z, v.w + pk.w);
            };
            for (uint32_t it = 0; it + 1 < n_iter; it++) iteration(std::integral_constant<int, 4>{}, it, false);
            iteration(std::integral_constant<int, NE_LAST>{}, n_iter - 1u, true);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 15; j++) raw[j] = raw[j + 1];
            raw[15] = nx;
        }
        SS_FPROF_MARK(10);
        __syncthreads();                    // epilogue reads are done before the next window's pass-1 writes
        SS_FPROF_MARK(11);
    }
    SS_FPROF_END;
#undef X1W
#undef X2W
}

// run geometry of k_fft16k_run: enough workgroups to fill 256 CUs x 2, runs of at least 16 windows (the run's first
// window costs a full load)
void fft16k_run_geometry(uint32_t n_streams, uint32_t fft_ch, uint32_t n_windows, uint32_t *windows_per_block, uint32_t *groups_out)
{
    const uint64_t pairs = (uint64_t)n_streams * fft_ch;
    uint32_t groups = (uint32_t)((4096 + pairs - 1) / pairs);
    const uint32_t max_groups = n_windows / 16u ? n_windows / 16u : 1u;
    if (groups > max_groups) groups = max_groups;
    if (groups < 1) groups = 1;
#ifdef SS_TUNING        // development builds only (tools/ab_cfg5.sh with SS_FFT16K_GROUPS)
    if (const char *e = std::getenv("SS_FFT16K_GROUPS")) { const int v = std::atoi(e); if (v >= 1 && (uint32_t)v <= n_windows) groups = (uint32_t)v; }
#endif
    const uint32_t wpb = (n_windows + groups - 1) / groups;
    *windows_per_block = wpb;
    *groups_out = (n_windows + wpb - 1) / wpb;
}

// runs of windows at hop 1024 (batches); `mode` as launch_fft16k
hipError_t launch_fft16k_run(FftBatchParams p, int mode, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0 || p.n_bins == 0) return hipSuccess;
    const uint32_t fft_ch = (mode == 0) ? 1u : (mode == 1 ? 2u : p.channels);
    const uint64_t pairs = (uint64_t)p.n_streams * fft_ch;
    uint32_t groups = 1;
    fft16k_run_geometry(p.n_streams, fft_ch, p.n_windows, &p.windows_per_block, &groups);
    const dim3 grid((uint32_t)((pairs * groups + 7) & ~(uint64_t)7)), block(512);      // multiple of 8: see the XCD mapping in the kernel
    const uint32_t ngroups = (p.n_bins + 3) >> 2, n_iter = (4u * ngroups + 2047u) >> 11;
    const uint32_t ne_last = (4u * ngroups - 2048u * (n_iter - 1u) + 511u) >> 9;       // slices of the epilogue's last iteration, 1 .. 4
#define SS_RUN16K(MS, NE) hipLaunchKernelGGL((k_fft16k_run<MS, NE>), grid, block, 0, s, p, fft_ch)
    if (mode == 1) { if (ne_last == 1) SS_RUN16K(true, 1); else if (ne_last == 2) SS_RUN16K(true, 2); else if (ne_last == 3) SS_RUN16K(true, 3); else SS_RUN16K(true, 4); }
    else { if (ne_last == 1) SS_RUN16K(false, 1); else if (ne_last == 2) SS_RUN16K(false, 2); else if (ne_last == 3) SS_RUN16K(false, 3); else SS_RUN16K(false, 4); }
#undef SS_RUN16K
    return hipGetLastError();
}

hipError_t launch_fft16k(const FftBatchParams &p, int mode, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0 || p.n_bins == 0) return hipSuccess;
    const uint32_t fft_ch = (mode == 0) ? 1u : (mode == 1 ? 2u : p.channels);
    hipLaunchKernelGGL(k_fft16k, dim3(p.n_streams * p.n_windows * fft_ch), dim3(512), 0, s, p, mode == 1 ? 1 : 0, fft_ch);
    return hipGetLastError();
}

hipError_t launch_fft4096_ms(const FftBatchParams &p, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0) return hipSuccess;
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    dim3 grid(groups * p.n_streams), block(256);
    // hop 1024 (the reference's cadence): the single-window kernel at 3 workgroups per CU measured 2.5 %
    // faster than the window-pair kernel k_fft4096_ms<4> at 2 (A/B in one process, 3.48 vs 3.57 ms)
    if (p.hop == 1024 && p.out_cols) hipLaunchKernelGGL((k_fft4096_ms1<4, SS_COLS_TW, true>), grid, block, 0, s, p);
    else if (p.hop == 1024) hipLaunchKernelGGL((k_fft4096_ms1<4, SS_MS1_TW, false>), grid, block, 0, s, p);
    else if (p.hop == 512) hipLaunchKernelGGL(k_fft4096_ms<2>, grid, block, 0, s, p);
    else if (p.hop == 2048) hipLaunchKernelGGL(k_fft4096_ms<8>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_fft4096_ms_anyhop, grid, block, 0, s, p);
    return hipGetLastError();
}

// ============================================================================
//  Spectrum, generic power-of-two N (2..32768), one real channel per workgroup:
//  real FFT through an N/2-point complex FFT held in LDS (in-place radix-2
//  decimation in frequency, bit-reversed read-out), then the same epilogue.
//  Used by the single-window API (analyzer.rs:55-105 takes any power of two)
//  and by batch shapes the specialised kernel does not cover.
// ============================================================================
__device__ __forceinline__ uint32_t bitrev(uint32_t v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

__global__ __launch_bounds__(256) void k_fft_generic(FftBatchParams p, int mode, int log2m)
{
    extern __shared__ __attribute__((aligned(16))) float2 zs[];
    const uint32_t n = p.n, m = n >> 1;
    const uint32_t fft_ch = (mode == 0) ? 1u : (mode == 1 ? 2u : p.channels);
    // block -> (stream, window, channel)
    uint32_t bid = blockIdx.x;
    const uint32_t ch = bid % fft_ch; bid /= fft_ch;
    const uint32_t w = bid % p.n_windows;
    const uint32_t stream = bid / p.n_windows;
    if (p.windows_of && w >= p.windows_of[stream]) return;                          // ragged batches
    const size_t start = p.first_start + (size_t)w * p.hop;
    const float *base = p.pcm + ((size_t)stream * p.frames_per_stream + start) * p.channels;

    auto sample = [&](uint32_t i) -> float {
        if (mode == 1) {
            float2 v = reinterpret_cast<const float2 *>(base)[i];
            // get_mid_and_side_samples: (l + r) / 2., (l - r) / 2.
            return ch == 0 ? (v.x + v.y) * 0.5f : (v.x - v.y) * 0.5f;
        }
        return base[(size_t)i * p.channels + (mode == 0 ? 0 : ch)];
    };
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        float x0 = sample(2 * i) * p.window[2 * i];
        float x1 = sample(2 * i + 1) * p.window[2 * i + 1];
        zs[i] = make_float2(x0, x1);
    }
    // DIF radix-2 stages: span s = m/2 .. 1
    for (uint32_t s = m >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        const uint32_t tw_step = n / (2 * s) ;   // W_{2s}^r = W_n^(r * n/(2s))
        for (uint32_t b = threadIdx.x; b < (m >> 1); b += blockDim.x) {
            const uint32_t r = b & (s - 1);
            const uint32_t i = ((b & ~(s - 1)) << 1) | r;
            const uint32_t j = i + s;
            const float2 a = zs[i], c = zs[j];
            zs[i] = cadd(a, c);
            zs[j] = cmul(csub(a, c), p.tw_n[r * tw_step]);
        }
    }
    __syncthreads();
    float *o = p.out + (((size_t)stream * p.n_windows + w) * fft_ch + ch) * p.bin_stride;
    for (uint32_t idx = threadIdx.x; idx < p.n_bins; idx += blockDim.x) {
        const uint32_t k = p.first_bin + idx;
        float xr, xi;
        const float2 z0 = zs[0];
        if (k == m) { xr = z0.x - z0.y; xi = 0.0f; }                 // Nyquist
        else if (k == 0) { xr = z0.x + z0.y; xi = 0.0f; }
        else {
            const float2 zk = zs[bitrev(k, log2m)];
            const float2 zc = zs[bitrev(m - k, log2m)];
            const float sr = (zk.x + zc.x) * 0.5f, si = (zk.y - zc.y) * 0.5f;
            const float dr = (zk.x - zc.x) * 0.5f, di = (zk.y + zc.y) * 0.5f;
            const float2 wv = p.tw_n[k];
            const float tr = wv.x * dr - wv.y * di;
            const float ti = wv.x * di + wv.y * dr;
            xr = sr + ti;
            xi = si - tr;
        }
        const float q = fmaf(xr, xr, xi * xi);
        float r = fmaf(log2f(q), 3.01029995663981195f, p.db_offset);
        r = (q == 0.0f) ? -150.0f : r;
        o[idx] = r + (p.pink ? p.pink[idx] : 0.0f);
    }
}

hipError_t launch_fft_generic(const FftBatchParams &p, int mode, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0 || p.n_bins == 0) return hipSuccess;
    const uint32_t fft_ch = (mode == 0) ? 1u : (mode == 1 ? 2u : p.channels);
    const uint32_t m = p.n >> 1;
    int log2m = 0;
    while ((1u << log2m) < m) log2m++;
    const size_t lds = (size_t)(m ? m : 1) * sizeof(float2);
    static DevicePrep prepared;
    const hipError_t pe = prepare_on_device(prepared, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_generic), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    });
    if (pe != hipSuccess) return pe;
    dim3 grid(p.n_streams * p.n_windows * fft_ch), block(256);
    hipLaunchKernelGGL(k_fft_generic, grid, block, lds, s, p, mode, log2m);
    return hipGetLastError();
}

}  // namespace ssk

#ifdef SS_FFT_PROF
// development builds only: [0..11] phase clocks of k_fft4096_ms1 (compute / barrier alternating), [15] waves counted
extern "C" int ss_debug_fft_prof(unsigned long long *out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(ssk::g_fft_prof), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(ssk::g_fft_prof), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
