// ss_td_impl.h: the time-domain kernels (k_time_domain, k_tick) and their launch templates.  Included by ss_time_domain.hip
// (host entry points, chunk-length model: nothing instantiated there) and by ss_td_f4.hip / ss_td_f2.hip / ss_td_f0.hip, each of
// which instantiates the kernels of ONE true-peak oversampling factor — a third of the ~140 instantiations per translation unit, so
// that the three compile side by side (one file took 100 s of a 105 s build).
#pragma once
// Reference semantics: /root/reference/src/analyzer.rs (get_fft :55-105, get_waveform :107-137,
// add_samples/getters :139-164, calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as restated in DESIGN.md.
// Nothing here is translated from the reference: the reference has no GPU code.
#include "ss_kernels.h"
#include "ss_fft_dev.h"
#include <atomic>
#include <cstdlib>
#include <cstdio>

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
//  * Segments.  A stream is cut into `nseg` runs of whole sub-blocks so that a batch of a few hundred streams still fills
//    4096 wave slots.  The recurrence has a long memory (the slowest pole pair: |z| = 0.99502 at 48 kHz, e^-240 per second at any
//    rate, a near-double pole), so the state has to be handed from one segment to the next — EXACTLY, in two launches: every
//    segment starts from a zero state at its boundary and leaves the state behind its last frame in seg_state; the fix-up launch
//    (launch_time_domain_fixup: filter and energies only) then re-runs the first fix_sub = 2 sub-blocks of every segment > 0 from
//    the state the segment in front of it left and overwrites their energies.  Behind those 0.2 s the zero start differs from the
//    true trajectory by e^-48 (1 + 48) = 7e-20 of the state — under the 1e-11 at which any two evaluation orders of this
//    recurrence differ (profiles/r05_ab_td_handover.txt).  Segment 0 (and every streaming call, nseg = 1) starts from the true
//    carried state.  (warm_sub > 0, SS_TD_RUN_IN: the form of rounds 1-4 — a segment starts `warm` sub-blocks early from a zero
//    state and drops that run-in; a truncation, 9.4e-8 at a segment's first sub-block on the bench corpus — kept for comparison.)
//    (split_batch, SS_TD_WHOLE_STREAMS: no segments at all — see SPLIT below.)
//  * K-weighting on the f64 VALU.  Each lane owns one (chunk of L frames,
//    channel); the recurrence is cut by  state_out = A^L state_in + zero_state:
//      pass 1: per chunk, run the state recurrence from zero              (4 FMA)
//      scan  : in-wave Hillis-Steele over chunks with the constant matrices
//              (A^L)^(2^k); chunks are dealt round-robin to the four DPP rows so that
//              the long distances are in-row v_mov_dpp shifts, the short ones ds_bpermute
//      pass 2: rerun each chunk from its true initial state, accumulate y^2.
//    L is chosen by td_chunk_frames (below): whole tiles of whole chunks first (48 kHz stereo: L = 30, a
//    sub-block = 5 tiles x 32 chunks), then occupancy and the bank conflicts of the per-lane walk.
//  * True peak at the crate's f32 width, the polyphase FIR  y_f[n] = sum_t c_f[t] x[n-t]  (an f32 fma chain per output), in the
//    form that measured fastest for the shape (round 6):
//      - 2 / 6 / 8 channels: on the packed-f32 VALU — a frame's pair of adjacent channels is one packed operand, a lane takes
//        fifteen frames of one pair, taps broadcast from SGPR pairs through op_sel (SS_TP_VALU_PHASE3); factor 2 in the
//        three-waves builds: the 24-tap branch's halves on neighbouring lanes, taps as per-lane VGPR pairs, one DPP add per result;
//      - channel counts that do not divide sixteen, and factor 2 in the four-waves builds: plain v_fma_f32, a lane takes fifteen
//        frames of ONE channel (SS_TP_VALU_PLAIN3);
//      - mono, 4 and 16 channels (and the four-waves builds of whole-stream workgroups): a banded-Toeplitz product on the f32
//        matrix pipe over a block of BLK consecutive outputs,
//          D[(f,r), col] = sum_k A[(f,r), k] * B[k, col],  A[(f,r), k] = c_f[HIST-1 + r - k],  B[k, col] = x[start_col - (HIST-1) + k],
//        issued as v_mfma_f32_16x16x4_f32.  Factor 4: 3 phases x 5 outputs = 15 rows over a 16-sample window (4 MFMAs per 16
//        columns, 70 % of the MACs useful); factor 2: 16 outputs over a 39-sample window (10 MFMAs).
//      - opt-in SS_TP_ARITH_F16X3 (factor 4; 1, 2, 4, 8 channels): three-term f16 split on v_mfma_f32_16x16x16_f16.
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
// (round 6, the true peak now a VALU phase — packed FMAs instead of MFMAs: its level re-measured, three interleaved repetitions at
// the bench shape: 0 -> 2.14 ms, 1 -> 2.02 (kept), 2 -> 2.05, 3 -> 2.04)
#ifndef SS_TD_PRIO_MASK
#define SS_TD_PRIO_MASK 0x05ECu     // {0, 3, 2, 3, 1, 1, 0, 0}
#endif
#define SS_TD_PHASE_PRIORITY(mark) __builtin_amdgcn_s_setprio((SS_TD_PRIO_MASK >> (2 * (mark))) & 3u)

// Development build (-DSS_TD_PROF): per-phase shader-clock totals of k_time_domain, summed over all waves
// (s_memtime at the phase boundaries of the tile loop; read back with ss_debug_td_prof).  Not in release builds.
#ifdef SS_TD_PROF
#ifdef SS_TD_DEBUG_OWNER
__device__ unsigned long long g_td_prof[16];
#else
static __device__ unsigned long long g_td_prof[16];      // (the other factors' clocks are not read)
#endif
#define SS_PROF_DECL uint64_t pt_ = __builtin_amdgcn_s_memtime(); uint64_t pacc_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SS_PROF_MARK(i) do { const uint64_t n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; SS_TD_PHASE_PRIORITY(i); } while (0)
#define SS_PROF_END do { if (lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 9; i_++) atomicAdd(&g_td_prof[i_], (unsigned long long)pacc_[i_]); atomicAdd(&g_td_prof[15], 1ull); } } while (0)
#elif defined(SS_TD_TRACE)
// Development build (-DSS_TD_TRACE): the timeline of ONE streaming call shared by a workgroup's waves (SPLIT) — for every tile the
// 100 MHz clock at its phase boundaries (slots 0-7: the marks of the tile loop; 8 tile taken up, 9 state of the tile in front
// received, 10 energy shares received, 11 the wave's start, 12 the wave's end; read back with ss_debug_td_trace)
#ifdef SS_TD_DEBUG_OWNER
__device__ unsigned long long g_td_trace[32][16];
#else
static __device__ unsigned long long g_td_trace[32][16];
#endif
#define SS_TRACE(slot) do { if (SPLIT && lane == 0 && ti < 32u) g_td_trace[ti][slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SS_PROF_DECL const unsigned long long tr_start_ = __builtin_amdgcn_s_memrealtime();
#define SS_PROF_MARK(i) do { SS_TRACE(i); SS_TD_PHASE_PRIORITY(i); } while (0)
#define SS_PROF_END do { if (SPLIT && lane == 0 && wave_in_block < 8u) { g_td_trace[wave_in_block][11] = tr_start_; g_td_trace[wave_in_block][12] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define SS_PROF_DECL
#define SS_PROF_MARK(i) SS_TD_PHASE_PRIORITY(i)
#define SS_PROF_END
#endif
#ifndef SS_TRACE
#define SS_TRACE(slot)
#endif

// packed-f32 forms of the true-peak interpolator (stereo: a frame (L, R) is one packed operand): acc += c x with the tap c broadcast
// from the low / high half of an SGPR pair
typedef float v2f_td __attribute__((ext_vector_type(2)));
// One polyphase branch over THREE consecutive frames, both channels: out0..2 = sum over k of c[k] (L, R)[n - k], k ascending, as
// three interleaved chains of twelve packed operations (a multiply, eleven FMAs) in ONE asm statement — between separate statements
// the compiler keeps a wait state for hazards it cannot rule out for inline asm (an s_nop per three instructions).  Operands:
// %0..%2 the three results (frames n, n + 1, n + 2), %3..%16 the window's fourteen frames (n - 11 .. n + 2), %17..%22 the taps as
// six register pairs (c[2 m], c[2 m + 1]), broadcast through op_sel — SGPR pairs where the taps are the wave's, VGPR pairs where
// neighbouring lanes run different halves of a longer branch (factor 2).
#define SS_TP_VALU_PHASE3_(CK, o0, o1, o2, W, g, tc)                                                                        \
    asm("v_pk_mul_f32 %0, %17, %14 op_sel:[0,0] op_sel_hi:[0,1]\n\t" \
        "v_pk_mul_f32 %1, %17, %15 op_sel:[0,0] op_sel_hi:[0,1]\n\t" \
        "v_pk_mul_f32 %2, %17, %16 op_sel:[0,0] op_sel_hi:[0,1]\n\t" \
        "v_pk_fma_f32 %0, %17, %13, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %1, %17, %14, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %2, %17, %15, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %0, %18, %12, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %1, %18, %13, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %2, %18, %14, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %0, %18, %11, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %1, %18, %12, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %2, %18, %13, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %0, %19, %10, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %1, %19, %11, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %2, %19, %12, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %0, %19, %9, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %1, %19, %10, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %2, %19, %11, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %0, %20, %8, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %1, %20, %9, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %2, %20, %10, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %0, %20, %7, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %1, %20, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %2, %20, %9, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %0, %21, %6, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %1, %21, %7, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %2, %21, %8, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %0, %21, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %1, %21, %6, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %2, %21, %7, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %0, %22, %4, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %1, %22, %5, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %2, %22, %6, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t" \
        "v_pk_fma_f32 %0, %22, %3, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %1, %22, %4, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 %2, %22, %5, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]"                                                                                                           \
        : "=&v"(o0), "=&v"(o1), "=&v"(o2)                                                                                \
        : "v"(W[3 * (g) + 0]), "v"(W[3 * (g) + 1]), "v"(W[3 * (g) + 2]), "v"(W[3 * (g) + 3]), "v"(W[3 * (g) + 4]),       \
          "v"(W[3 * (g) + 5]), "v"(W[3 * (g) + 6]), "v"(W[3 * (g) + 7]), "v"(W[3 * (g) + 8]), "v"(W[3 * (g) + 9]),       \
          "v"(W[3 * (g) + 10]), "v"(W[3 * (g) + 11]), "v"(W[3 * (g) + 12]), "v"(W[3 * (g) + 13]),                        \
          CK(tc[0]), CK(tc[1]), CK(tc[2]), CK(tc[3]), CK(tc[4]), CK(tc[5]))
#define SS_TP_VALU_PHASE3(o0, o1, o2, W, g, tc) SS_TP_VALU_PHASE3_("s", o0, o1, o2, W, g, tc)       /* taps wave-uniform: SGPR pairs */
#define SS_TP_VALU_PHASE3V(o0, o1, o2, W, g, tc) SS_TP_VALU_PHASE3_("v", o0, o1, o2, W, g, tc)      /* taps per lane: VGPR pairs */

// The same over ONE channel with plain f32 FMAs (any channel count: a lane takes fifteen frames of one channel; on gfx950 a wave's
// v_fma_f32 issues in 2.8 cycles against 5.2 for v_pk_fma_f32, so per MAC this is within a tenth of the packed form): twelve taps
// (SGPRs, %17..%28) over three frames, window %3..%16 as above.  _ACC adds twelve further taps onto the results (factor 2: the
// branch's second half over the window twelve frames further back).
#define SS_TP_VALU_PLAIN3(o0, o1, o2, W, g, tc)                                                                          \
    asm("v_mul_f32 %0, %17, %14\n\t" \
        "v_mul_f32 %1, %17, %15\n\t" \
        "v_mul_f32 %2, %17, %16\n\t" \
        "v_fma_f32 %0, %18, %13, %0\n\t" \
        "v_fma_f32 %1, %18, %14, %1\n\t" \
        "v_fma_f32 %2, %18, %15, %2\n\t" \
        "v_fma_f32 %0, %19, %12, %0\n\t" \
        "v_fma_f32 %1, %19, %13, %1\n\t" \
        "v_fma_f32 %2, %19, %14, %2\n\t" \
        "v_fma_f32 %0, %20, %11, %0\n\t" \
        "v_fma_f32 %1, %20, %12, %1\n\t" \
        "v_fma_f32 %2, %20, %13, %2\n\t" \
        "v_fma_f32 %0, %21, %10, %0\n\t" \
        "v_fma_f32 %1, %21, %11, %1\n\t" \
        "v_fma_f32 %2, %21, %12, %2\n\t" \
        "v_fma_f32 %0, %22, %9, %0\n\t" \
        "v_fma_f32 %1, %22, %10, %1\n\t" \
        "v_fma_f32 %2, %22, %11, %2\n\t" \
        "v_fma_f32 %0, %23, %8, %0\n\t" \
        "v_fma_f32 %1, %23, %9, %1\n\t" \
        "v_fma_f32 %2, %23, %10, %2\n\t" \
        "v_fma_f32 %0, %24, %7, %0\n\t" \
        "v_fma_f32 %1, %24, %8, %1\n\t" \
        "v_fma_f32 %2, %24, %9, %2\n\t" \
        "v_fma_f32 %0, %25, %6, %0\n\t" \
        "v_fma_f32 %1, %25, %7, %1\n\t" \
        "v_fma_f32 %2, %25, %8, %2\n\t" \
        "v_fma_f32 %0, %26, %5, %0\n\t" \
        "v_fma_f32 %1, %26, %6, %1\n\t" \
        "v_fma_f32 %2, %26, %7, %2\n\t" \
        "v_fma_f32 %0, %27, %4, %0\n\t" \
        "v_fma_f32 %1, %27, %5, %1\n\t" \
        "v_fma_f32 %2, %27, %6, %2\n\t" \
        "v_fma_f32 %0, %28, %3, %0\n\t" \
        "v_fma_f32 %1, %28, %4, %1\n\t" \
        "v_fma_f32 %2, %28, %5, %2"                                                                                                           \
        : "=&v"(o0), "=&v"(o1), "=&v"(o2)                                                                                \
        : "v"((W)[3 * (g) + 0]), "v"((W)[3 * (g) + 1]), "v"((W)[3 * (g) + 2]), "v"((W)[3 * (g) + 3]), "v"((W)[3 * (g) + 4]),       \
          "v"((W)[3 * (g) + 5]), "v"((W)[3 * (g) + 6]), "v"((W)[3 * (g) + 7]), "v"((W)[3 * (g) + 8]), "v"((W)[3 * (g) + 9]),       \
          "v"((W)[3 * (g) + 10]), "v"((W)[3 * (g) + 11]), "v"((W)[3 * (g) + 12]), "v"((W)[3 * (g) + 13]),                        \
          "s"((tc)[0]), "s"((tc)[1]), "s"((tc)[2]), "s"((tc)[3]), "s"((tc)[4]), "s"((tc)[5]),                                    \
          "s"((tc)[6]), "s"((tc)[7]), "s"((tc)[8]), "s"((tc)[9]), "s"((tc)[10]), "s"((tc)[11]))
#define SS_TP_VALU_PLAIN3_ACC(o0, o1, o2, W, g, tc)                                                                          \
    asm("v_fma_f32 %0, %17, %14, %0\n\t" \
        "v_fma_f32 %1, %17, %15, %1\n\t" \
        "v_fma_f32 %2, %17, %16, %2\n\t" \
        "v_fma_f32 %0, %18, %13, %0\n\t" \
        "v_fma_f32 %1, %18, %14, %1\n\t" \
        "v_fma_f32 %2, %18, %15, %2\n\t" \
        "v_fma_f32 %0, %19, %12, %0\n\t" \
        "v_fma_f32 %1, %19, %13, %1\n\t" \
        "v_fma_f32 %2, %19, %14, %2\n\t" \
        "v_fma_f32 %0, %20, %11, %0\n\t" \
        "v_fma_f32 %1, %20, %12, %1\n\t" \
        "v_fma_f32 %2, %20, %13, %2\n\t" \
        "v_fma_f32 %0, %21, %10, %0\n\t" \
        "v_fma_f32 %1, %21, %11, %1\n\t" \
        "v_fma_f32 %2, %21, %12, %2\n\t" \
        "v_fma_f32 %0, %22, %9, %0\n\t" \
        "v_fma_f32 %1, %22, %10, %1\n\t" \
        "v_fma_f32 %2, %22, %11, %2\n\t" \
        "v_fma_f32 %0, %23, %8, %0\n\t" \
        "v_fma_f32 %1, %23, %9, %1\n\t" \
        "v_fma_f32 %2, %23, %10, %2\n\t" \
        "v_fma_f32 %0, %24, %7, %0\n\t" \
        "v_fma_f32 %1, %24, %8, %1\n\t" \
        "v_fma_f32 %2, %24, %9, %2\n\t" \
        "v_fma_f32 %0, %25, %6, %0\n\t" \
        "v_fma_f32 %1, %25, %7, %1\n\t" \
        "v_fma_f32 %2, %25, %8, %2\n\t" \
        "v_fma_f32 %0, %26, %5, %0\n\t" \
        "v_fma_f32 %1, %26, %6, %1\n\t" \
        "v_fma_f32 %2, %26, %7, %2\n\t" \
        "v_fma_f32 %0, %27, %4, %0\n\t" \
        "v_fma_f32 %1, %27, %5, %1\n\t" \
        "v_fma_f32 %2, %27, %6, %2\n\t" \
        "v_fma_f32 %0, %28, %3, %0\n\t" \
        "v_fma_f32 %1, %28, %4, %1\n\t" \
        "v_fma_f32 %2, %28, %5, %2"                                                                                                           \
        : "+v"(o0), "+v"(o1), "+v"(o2)                                                                                \
        : "v"((W)[3 * (g) + 0]), "v"((W)[3 * (g) + 1]), "v"((W)[3 * (g) + 2]), "v"((W)[3 * (g) + 3]), "v"((W)[3 * (g) + 4]),       \
          "v"((W)[3 * (g) + 5]), "v"((W)[3 * (g) + 6]), "v"((W)[3 * (g) + 7]), "v"((W)[3 * (g) + 8]), "v"((W)[3 * (g) + 9]),       \
          "v"((W)[3 * (g) + 10]), "v"((W)[3 * (g) + 11]), "v"((W)[3 * (g) + 12]), "v"((W)[3 * (g) + 13]),                        \
          "s"((tc)[0]), "s"((tc)[1]), "s"((tc)[2]), "s"((tc)[3]), "s"((tc)[4]), "s"((tc)[5]),                                    \
          "s"((tc)[6]), "s"((tc)[7]), "s"((tc)[8]), "s"((tc)[9]), "s"((tc)[10]), "s"((tc)[11]))

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
typedef const __attribute__((address_space(4))) float *const_f32_ptr;
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
constexpr int kTdSplitWaves = 8;        // SPLIT: waves that share one streaming call (four when eight slices do not fit the LDS)
#ifndef SS_TD_PREFETCH
#define SS_TD_PREFETCH 8
#endif
constexpr int kTdPrefetch = SS_TD_PREFETCH;        // float4 per lane held in flight for the next tile
// (round 6, with the true peak on the VALU: 3 / 5 / 6 / 10 reads per batch -> 2.16 / 2.03 / 2.05 / 2.08 ms at the bench shape on one box,
// 15 -> +1 % on another; five is also 1-4 % better at 44.1 kHz, 5.1 and mono, 1.7 % worse at one odd length,
// profiles/r06_ab_td_read_batch.txt)
#ifndef SS_TD_BATCH
#define SS_TD_BATCH 5
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

// one tap of the crate's interpolator loop: the product and the sum rounded separately (Rust does not contract a * b + c)
__device__ __forceinline__ float tp_mul_then_add(float acc, float x, float c)
{
#pragma clang fp contract(off)
    const float pr = x * c;
    return acc + pr;
}

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
// SPLIT (streaming calls, nseg == 1: the handle's add_samples and the session ticks): the eight waves of a workgroup (four
// where eight LDS slices do not fit) share ONE stream's call, wave w taking tiles w, w + 8, w + 16, ...  A tile's staging, its zero-state pass and — behind the
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
    double call_e[kTdSplitWaves];        // a tick's short-term reading: the waves' weighted energy of the whole call
    uint32_t waves_done;                 // waves whose share stands in call_e
};

// A tick's short-term reading is summed by whichever of its st_blocks + 1 contributors (the ring workgroups below, the loudness
// call's workgroup) ARRIVES LAST — nobody waits for anybody, so the launch cannot hang where the workgroups are not all resident
// at once (CU masking, a tiny partition, a debugger).  A contributor leaves its share in st_scratch (ring workgroup r: slot r,
// the loudness call: slot kRingTickBlocks + 1), makes it visible to the device and counts itself in; the one that reads
// st_blocks from the counter adds everything in a fixed order (the same tree whoever runs it) and writes (energy, loudness).
__device__ __forceinline__ bool tick_reading_arrive(const TdParams &p)
{
    __threadfence();                                                       // the share is visible before the count
    uint32_t *const count = reinterpret_cast<uint32_t *>(p.st_scratch + kRingTickBlocks);
    return __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == p.st_blocks;
}
// one whole wave of the last contributor
__device__ __forceinline__ void tick_reading_finish(const TdParams &p, uint32_t lane)
{
    double pr = lane < p.st_blocks ? __builtin_nontemporal_load(p.st_scratch + lane) : 0.0;
    for (int d = 32; d >= 1; d >>= 1) pr += __shfl_down(pr, d, 64);
    if (lane == 0) {
        const double tot = __builtin_nontemporal_load(p.st_scratch + kRingTickBlocks + 1);
        const double en = (pr + tot) / p.st_frames;
        p.st_out[0] = en;
        p.st_out[1] = en <= 0.0 ? -INFINITY : 10.0 * log10(en) - 0.691;      // energy_to_loudness
        __hip_atomic_store(reinterpret_cast<uint32_t *>(p.st_scratch + kRingTickBlocks), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// first part (k_tick's workgroups behind the loudness call's): the weighted energy of the ring over
// the frames of the window that lie IN FRONT of this call — which the call does not touch, so these workgroups run beside it.
// One run of ring elements with at most one wrap, like k_ring_energy (ss_loudness.hip); block r of `blocks` leaves its partial
// sum in scratch[r] and counts itself in behind it.
__device__ __forceinline__ void ring_window_partial(const TdParams &p, uint32_t r)
{
    __shared__ double red[kTdSplitWaves];
    const uint32_t stride = p.st_blocks * 64u * kTdSplitWaves;
    const uint32_t tid = r * 64u * kTdSplitWaves + threadIdx.x;
    const uint32_t C = p.channels;
    const uint32_t ring_elems = (uint32_t)(p.ring_frames * C);
    const uint32_t cstep = stride % C;
    uint32_t c = tid % C;
    double acc = 0.0;
    uint32_t i = tid;
    for (; i + 3u * stride < p.st_old_total; i += 4u * stride) {
        double y[4], w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t e = p.st_begin_elem + i + (uint32_t)q * stride;
            if (e >= ring_elems) e -= ring_elems;
            y[q] = p.ring[e];
            w[q] = p.st_weights[c];
            c += cstep; if (c >= C) c -= C;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc = w[q] != 0.0 ? fma(w[q] * y[q], y[q], acc) : acc;      // (weight 0: the crate does not filter that channel)
    }
    for (; i < p.st_old_total; i += stride) {
        uint32_t e = p.st_begin_elem + i;
        if (e >= ring_elems) e -= ring_elems;
        const double y = p.ring[e];
        const double wc = p.st_weights[c];
        acc = wc != 0.0 ? fma(wc * y, y, acc) : acc;
        c += cstep; if (c >= C) c -= C;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    __shared__ uint32_t last_here;
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kTdSplitWaves; w++) t += red[w];
        p.st_scratch[r] = t;
        last_here = tick_reading_arrive(p) ? 1u : 0u;
    }
    __syncthreads();
    if (last_here && threadIdx.x < 64u) tick_reading_finish(p, threadIdx.x);
}

template <int FACTOR, bool RING, int CT, int WAVE, int WPS, bool SPLIT = false, bool LATE = false>
__global__ __launch_bounds__(64 * (SPLIT ? kTdSplitWaves : kTdWavesPerBlock), WPS) void k_time_domain(TdParams p, uint32_t L, uint32_t tile_len,
                                                                                      uint32_t wave_lds_floats, uint32_t halo_frames)
{
    const uint32_t block_id = blockIdx.x;
#include "ss_td_body.inc"
}

// One launch for a tick of the reference (tui.rs:1482-1552): the first `fft_blocks` workgroups transform the newest 16384
// mid / side samples (fft16k_window: one signal each), the workgroup behind them is the loudness call's eight waves (SPLIT).
// Two kernels on two streams did the same through most of round 4 — side by side only when the streams' hardware queues sat
// on different pipes of the command processor, otherwise one after the other (98 instead of 65 us per tick for the whole life of
// such a session, a lottery at stream creation: tools/probe_tick_queues.sh); gfx9 has no launch flag that lets a kernel start
// beside its predecessor on ONE stream (hipExtAnyOrderLaunch is not supported there).  One dispatch has no such dependence,
// and is one launch less.
template <int FACTOR, int CT>
__global__ __launch_bounds__(64 * kTdSplitWaves, 3) void k_tick(TdParams p, uint32_t L, uint32_t tile_len, uint32_t wave_lds_floats,
                                                                 uint32_t halo_frames, FftBatchParams fp, uint32_t fft_blocks)
{
    if (blockIdx.x < fft_blocks) {
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef SS_TD_TRACE
        const unsigned long long f0_ = __builtin_amdgcn_s_memrealtime();
#endif
        fft16k_window(fp, 1, fft_blocks, blockIdx.x, smem);
#ifdef SS_TD_TRACE
        if (threadIdx.x == 0) { g_td_trace[30 + (blockIdx.x & 1u)][0] = f0_; g_td_trace[30 + (blockIdx.x & 1u)][1] = __builtin_amdgcn_s_memrealtime(); }
#endif
        return;
    }
    if (blockIdx.x > fft_blocks) {
#ifdef SS_TD_TRACE
        const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime();
#endif
        ring_window_partial(p, blockIdx.x - fft_blocks - 1u);
#ifdef SS_TD_TRACE
        if (threadIdx.x == 0 && blockIdx.x == fft_blocks + 1u) { g_td_trace[29][0] = r0_; g_td_trace[29][1] = __builtin_amdgcn_s_memrealtime(); }
#endif
        return;
    }
    constexpr bool RING = true, SPLIT = true, LATE = true;
    constexpr int WAVE = 0, WPS = 3;
    const uint32_t block_id = 0u;
#include "ss_td_body.inc"
}

uint32_t td_lds_blocks(uint32_t C, uint32_t tile_len);       // (ss_time_domain.hip: the chunk-length model)

template <int FACTOR, bool RING, int CT, int WAVE, int WPS, bool SPLIT = false, bool LATE = false>
static hipError_t td_launch_w(const TdParams &p, hipStream_t s)
{
    const uint32_t C = p.channels;
    const uint32_t S = p.s100;
    // chunk length: the batch's (whole tiles of whole chunks, occupancy) — or, where one call / one short segment is shared by eight
    // waves (streaming calls; split_batch == 2), the one that lets eight tiles cover it in ONE round
    const uint32_t L = (SPLIT && p.split_batch != 1u) ? td_split_chunk_frames(C, S) : td_chunk_frames(C, S);
    const uint32_t nch = 64u / C;
    const uint32_t cap = nch * L;                                   // frames one wave can scan at once
    const uint32_t pieces = (S + cap - 1) / cap;                    // equal tiles per sub-block
    uint32_t tile_len = (S + pieces - 1) / pieces;
    if (tile_len > cap) tile_len = cap;
    // per-wave LDS: halo + tile + slack + 64 peak slots
    const uint32_t halo = WAVE ? p.halo_frames : (uint32_t)kTdHaloFrames;
    uint32_t wave_floats = (halo + tile_len) * C + td_slack_floats(C) + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    // SPLIT: eight waves share the call where eight slices fit the LDS (up to 16 channels or so), four otherwise
    // (a batch's streams: four waves each — the grid is n_streams workgroups, and sixteen waves per CU are what the LDS slices allow;
    // split_batch == 2, a handful of streams cut into short segments: eight, latency is what counts there)
    uint32_t nwb = (SPLIT && p.split_batch != 1u) ? (uint32_t)kTdSplitWaves : (uint32_t)kTdWavesPerBlock;
    if (SPLIT && (size_t)wave_floats * 4 * nwb + sizeof(TdShare) > 160 * 1024) nwb = (uint32_t)kTdWavesPerBlock;
    const size_t lds = (size_t)wave_floats * 4 * nwb + (SPLIT ? sizeof(TdShare) : 0);
    auto fn = k_time_domain<FACTOR, RING, CT, WAVE, WPS, SPLIT, LATE>;
    static DevicePrep prepared;                     // one per kernel instantiation
    const hipError_t pe = prepare_on_device(prepared, [fn] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (pe != hipSuccess) return pe;
    const uint32_t waves = p.n_streams * (p.fixup ? p.nseg - 1u : p.nseg);
    const uint32_t blocks = SPLIT ? waves : (waves + kTdWavesPerBlock - 1) / kTdWavesPerBlock;      // SPLIT: a workgroup per (stream, segment)
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#ifdef SS_TUNING        // development builds only: name the instantiation a launch takes (tools/probe_td_wps.py)
    if (std::getenv("SS_TD_VERBOSE")) {
        hipFuncAttributes a{};
        (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(fn));
        std::fprintf(stderr, "k_time_domain<%d,%d,%d,%d,%d,%d,%d> grid %u x %u lds %zu scratch %zu B/lane vgpr %d\n", FACTOR, (int)RING, CT, WAVE, WPS,
                     (int)SPLIT, (int)LATE, blocks, 64 * nwb, lds, (size_t)a.localSizeBytes, a.numRegs);
    }
#endif
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * nwb), lds, s, p, L, tile_len, wave_floats, halo);
    return hipGetLastError();
}

// the tick launch (k_tick): possible when the call takes the SPLIT path on eight waves and the spectrum's static LDS fits beside
// the loudness call's slices; *fused says whether it happened (otherwise nothing was launched)
template <int FACTOR, int CT>
static hipError_t td_launch_tick(const TdParams &p, const FftBatchParams &fp, hipStream_t s, bool *fused)
{
    *fused = false;
    const uint32_t C = p.channels;
    const uint32_t S = p.s100;
    const uint32_t L = td_split_chunk_frames(C, S);
    const uint32_t cap = (64u / C) * L;
    const uint32_t pieces = (S + cap - 1) / cap;
    uint32_t tile_len = (S + pieces - 1) / pieces;
    if (tile_len > cap) tile_len = cap;
    const uint32_t halo = (uint32_t)kTdHaloFrames;
    uint32_t wave_floats = (halo + tile_len) * C + td_slack_floats(C) + kMaxChannels;
    wave_floats = (wave_floats + 3u) & ~3u;
    size_t lds = (size_t)wave_floats * 4 * kTdSplitWaves + sizeof(TdShare);
    if (lds < (size_t)kFft16kLdsBytes) lds = kFft16kLdsBytes;           // (the spectrum's workgroups use the same dynamic block)
    auto fn = k_tick<FACTOR, CT>;
    static DevicePrep prepared;
    static std::atomic<size_t> static_lds{0};
    const hipError_t pe = prepare_on_device(prepared, [fn] {
        hipFuncAttributes a{};
        hipError_t e = hipFuncGetAttributes(&a, reinterpret_cast<const void *>(fn));
        if (e != hipSuccess) return e;
        static_lds.store(a.sharedSizeBytes, std::memory_order_relaxed);
        return hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(160 * 1024 - a.sharedSizeBytes));
    });
    if (pe != hipSuccess) return pe;
    if (lds + static_lds.load(std::memory_order_relaxed) > 160 * 1024) return hipSuccess;     // not fused: the caller launches both
    const uint32_t fft_blocks = 2;                                      // mid, side
    hipLaunchKernelGGL(fn, dim3(fft_blocks + 1 + (p.st_out ? p.st_blocks : 0u)), dim3(64 * kTdSplitWaves), lds, s, p, L, tile_len, wave_floats,
                       halo, fp, fft_blocks);
    *fused = true;
    return hipGetLastError();
}

// compute units of the current device (256 on MI355X), asked once
static inline uint32_t td_device_cus()
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

// batches whose streams are walked by a whole workgroup (TdParams::split_batch): the same two register builds by grid size
template <int FACTOR, int CT, int WAVE>
static hipError_t td_launch_split_batch(const TdParams &p, hipStream_t s)
{
    const uint64_t blocks = (uint64_t)p.n_streams * (p.fixup ? p.nseg - 1u : p.nseg);
    // a handful of streams cut into short segments: eight waves per segment, the state applied behind the scan (LATE) — the chain
    // of a segment's tiles is what the launch takes
    if (p.split_batch == 2u) return td_launch_w<FACTOR, false, CT, WAVE, 2, true, true>(p, s);      // (two waves per SIMD: nothing spilled; the grid is small by definition)
    bool three = SS_TD_WAVES == 4 && blocks <= 3ull * td_device_cus();
#ifdef SS_TUNING        // development builds only: SS_TD_WPS=3|4 forces a register build
    if (const char *e = std::getenv("SS_TD_WPS")) three = SS_TD_WAVES == 4 && std::atoi(e) == 3;
#endif
    if (three) return td_launch_w<FACTOR, false, CT, WAVE, 3, true>(p, s);
    return td_launch_w<FACTOR, false, CT, WAVE, SS_TD_WAVES, true>(p, s);
}

template <int FACTOR, bool RING, int CT, int WAVE>
static hipError_t td_launch(const TdParams &p, hipStream_t s)
{
    // a workgroup is four waves, one per SIMD: three workgroups per CU hold the whole grid -> the spill-free build
    if constexpr (RING) return td_launch_w<FACTOR, RING, CT, WAVE, 3>(p, s);      // a streaming call is one stream: one workgroup
    else {                                             // (else: the four-waves build of a streaming form is never instantiated)
        const uint64_t waves = (uint64_t)p.n_streams * (p.fixup ? p.nseg - 1u : p.nseg);
        const uint64_t blocks = (waves + kTdWavesPerBlock - 1) / kTdWavesPerBlock;
        bool three = SS_TD_WAVES == 4 && blocks <= 3ull * td_device_cus();
#ifdef SS_TUNING        // development builds only: SS_TD_WPS=3|4 forces a register build
        if (const char *e = std::getenv("SS_TD_WPS")) three = SS_TD_WAVES == 4 && std::atoi(e) == 3;
#endif
        if (three) return td_launch_w<FACTOR, RING, CT, WAVE, 3>(p, s);
        return td_launch_w<FACTOR, RING, CT, WAVE, SS_TD_WAVES>(p, s);
    }
}

// Decimation fast path (WAVE = 2): samples per bin spp = len / W is an exact integer multiple of four (<= 128), so
// floor(i spp) / ceil((i+1) spp) are the integer products, and every tile starts on a multiple of four floats.
static inline int td_wave_int4(const TdParams &p)
{
    const uint64_t len = p.n_frames * p.channels;
    if (!p.wave_window || len % p.wave_window) return 0;
    const uint64_t spp = len / p.wave_window;
    if (spp < 4 || spp > 1000 || (spp & 3u)) return 0;         // the fused path itself stops at 1000 samples per bin
    const uint32_t C = p.channels, S = p.s100;
    const uint32_t L = p.split_batch == 2u ? td_split_chunk_frames(C, S) : td_chunk_frames(C, S);
    const uint32_t cap = (64u / C) * L;
    const uint32_t pieces = (S + cap - 1) / cap;
    uint32_t tile_len = (S + pieces - 1) / pieces;
    if (tile_len > cap) tile_len = cap;
    if (!(((uint64_t)S * C) % 4u == 0 && ((uint64_t)tile_len * C) % 4u == 0 && (p.halo_frames * C) % 4u == 0)) return 0;
    return spp <= 128 ? 2 : 3;
}

template <int FACTOR, bool RING>
static hipError_t td_launch_c(const TdParams &p, hipStream_t s, const FftBatchParams *tick_fft, bool *fused)
{
    if (!RING && p.split_batch) {  // (the host sets it only for the shapes instantiated here: stereo and eight channels, nseg == 1, no ragged lengths)
        if (p.frames_of || (p.channels != 2 && p.channels != 8)) return hipErrorInvalidValue;
        if (p.channels == 8) return p.wave_out ? td_launch_split_batch<FACTOR, 8, 1>(p, s) : td_launch_split_batch<FACTOR, 8, 0>(p, s);
        if (!p.wave_out) return td_launch_split_batch<FACTOR, 2, 0>(p, s);
        const int fast = td_wave_int4(p);
        if (fast == 2) return td_launch_split_batch<FACTOR, 2, 2>(p, s);
        if (fast == 3) return td_launch_split_batch<FACTOR, 2, 3>(p, s);
        return td_launch_split_batch<FACTOR, 2, 1>(p, s);
    }
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
    // streaming calls (the handle's add_samples, the session ticks) longer than one tile: the eight waves of a workgroup share
    // the call's tiles (SPLIT, see TdShare); the spill-free three-waves-per-SIMD build — a handful of workgroups at most
    bool split = RING && p.nseg == 1 && !p.frames_of;
#ifdef SS_TUNING        // development builds only: SS_TD_SPLIT=0 keeps streaming calls on one wave (A/B, drift measurements)
    if (const char *e = std::getenv("SS_TD_SPLIT")) split = split && std::atoi(e) != 0;
#endif
    if constexpr (RING) if (split) {            // (constexpr: the batch side never instantiates these forms)
        const uint32_t C = p.channels, S = p.s100;
        const uint32_t cap = (64u / C) * td_chunk_frames(C, S);
        const uint32_t pieces = (S + cap - 1) / cap;
        uint32_t tile_len = (S + pieces - 1) / pieces;
        if (tile_len > cap) tile_len = cap;
        if (p.n_frames > tile_len) {
            if (RING && tick_fft && p.n_streams == 1) {                 // a tick: the spectrum's workgroups ride the same launch
                const hipError_t e = p.channels == 2 ? td_launch_tick<FACTOR, 2>(p, *tick_fft, s, fused)
                                                     : td_launch_tick<FACTOR, 0>(p, *tick_fft, s, fused);
                if (e != hipSuccess || *fused) return e;
            }
            TdParams q = p;
            q.st_out = nullptr;                                          // (the reading needs k_tick's ring workgroups)
            return p.channels == 2 ? td_launch_w<FACTOR, RING, 2, 0, 3, true, true>(q, s) : td_launch_w<FACTOR, RING, 0, 0, 3, true, true>(q, s);
        }
    }
    if (!RING && p.channels == 8) return td_launch<FACTOR, false, 8, 0>(p, s);      // (config 5 without decimation; its fix-up launch)
    return p.channels == 2 ? td_launch<FACTOR, RING, 2, 0>(p, s) : td_launch<FACTOR, RING, 0, 0>(p, s);
}

// what ss_td_f<N>.hip defines: every launch of one oversampling factor (ring: the handle's / session's streaming calls)
hipError_t td_launch_f4(const TdParams &p, hipStream_t s, const FftBatchParams *tick_fft, bool *fused, bool ring);
hipError_t td_launch_f2(const TdParams &p, hipStream_t s, const FftBatchParams *tick_fft, bool *fused, bool ring);
hipError_t td_launch_f0(const TdParams &p, hipStream_t s, const FftBatchParams *tick_fft, bool *fused, bool ring);
#define SS_TD_DEFINE_FACTOR(N)                                                                                              \
    hipError_t td_launch_f##N(const TdParams &p, hipStream_t s, const FftBatchParams *tick_fft, bool *fused, bool ring)    \
    {                                                                                                                       \
        return ring ? td_launch_c<N, true>(p, s, tick_fft, fused) : td_launch_c<N, false>(p, s, nullptr, fused);           \
    }

}  // namespace ssk
