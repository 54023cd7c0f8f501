// ss_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the soundscope
// analyzer hot path.  Reference semantics: /root/reference/src/analyzer.rs
// (get_fft :55-105, get_waveform :107-137, add_samples/getters :139-164,
// calculate_integrated_lufs :170-182) and src/audio_player.rs:400-419, plus the
// arithmetic of ebur128 0.1.10 / spectrum-analyzer 1.7.0 / microfft 0.6.0 as
// restated in DESIGN.md.  Nothing here is translated from the reference: the
// reference has no GPU code.
#include "ss_kernels.h"

#ifndef SS_TD_WAVES
#define SS_TD_WAVES 4    // min waves per SIMD the time-domain kernel is register-allocated for
#endif
#ifndef SS_FFT_WAVES
#define SS_FFT_WAVES 2   // min waves per SIMD the N=4096 kernel is register-allocated for
#endif

namespace ssk {

// ============================================================================
//  small complex helpers (f32)
// ============================================================================
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
// a * (c - i s)
__device__ __forceinline__ float2 cmul_cs(float2 a, float c, float s) { return make_float2(a.x * c + a.y * s, a.y * c - a.x * s); }
// a * (-i)
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

// ---- packed-f32 complex arithmetic -------------------------------------------------------------
// A complex number lives in an even-aligned VGPR pair (re, im) and is processed with VOP3P packed
// f32 instructions, two flops per lane per instruction.  Measured on gfx950 (tools/ubench3.hip):
// v_pk_add_f32 issues in 5.7 cycles per wave-instruction at 2 waves/SIMD against 3.9 for v_add_f32,
// i.e. 27 % fewer issue cycles per complex add.  The swizzles a radix-4 butterfly needs (multiply by
// -i / +i, complex multiply) are expressed with op_sel / neg modifiers, which the compiler's SLP
// packer does not find (it pays ~30 % v_mov to pair registers instead — hence -fno-slp-vectorize).
typedef float v2f __attribute__((ext_vector_type(2)));
// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2f pk_sub_ib(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f pk_add_ib(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * w (complex): m = (a.y w.y, a.y w.x); r = (a.x w.x - m.x, a.x w.y + m.y)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w)
{
    v2f m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(m));
    return r;
}
// (a.x + a.y, a.y - a.x) = a - i a      [times R gives a * W16^2]
__device__ __forceinline__ v2f pk_w2pre(v2f a) { return pk_sub_ib(a, a); }
// (a.y - a.x, -(a.x + a.y))             [times R gives a * W16^6]
__device__ __forceinline__ v2f pk_w6pre(v2f a)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a));
    return r;
}
// a * (-i) = (a.y, -a.x)
__device__ __forceinline__ v2f pk_mul_mi(v2f a)
{
    v2f r;
    const v2f zero = {0.0f, 0.0f};
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(zero), "v"(a));
    return r;
}

// forward radix-4 butterfly on (a0,a1,a2,a3) in place: A_k = sum_j a_j (-i)^(jk)  — 8 packed adds
__device__ __forceinline__ void radix4(v2f &a0, v2f &a1, v2f &a2, v2f &a3)
{
    const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = pk_sub_ib(t1, t3);   // t1 - i t3
    a3 = pk_add_ib(t1, t3);   // t1 + i t3
}

// Forward 16-point DFT in registers (81 packed instructions).  Input a[j] natural order; output
// X[k] is left in a[R16(k)] with R16(k) = ((k & 3) << 2) | (k >> 2).
#define R16(k) ((((k) & 3) << 2) | ((k) >> 2))
__device__ __forceinline__ void fft16(v2f (&a)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    const v2f w1 = {C1, -S1}, w3 = {S1, -C1}, w9 = {-C1, S1};
    // stage 1: 4-point DFTs over q of a[r + 4q]; result p lands in a[r + 4p]
    radix4(a[0], a[4], a[8], a[12]);
    radix4(a[1], a[5], a[9], a[13]);
    radix4(a[2], a[6], a[10], a[14]);
    radix4(a[3], a[7], a[11], a[15]);
    // twiddle a[r + 4p] *= W16^(r p)
    a[5] = pk_cmul(a[5], w1);            // r=1,p=1: W^1
    a[9] = pk_w2pre(a[9]) * R;           // r=1,p=2: W^2
    a[13] = pk_cmul(a[13], w3);          // r=1,p=3: W^3
    a[6] = pk_w2pre(a[6]) * R;           // r=2,p=1: W^2
    a[10] = pk_mul_mi(a[10]);            // r=2,p=2: W^4 = -i
    a[14] = pk_w6pre(a[14]) * R;         // r=2,p=3: W^6 = (-R,-R)
    a[7] = pk_cmul(a[7], w3);            // r=3,p=1: W^3
    a[11] = pk_w6pre(a[11]) * R;         // r=3,p=2: W^6
    a[15] = pk_cmul(a[15], w9);          // r=3,p=3: W^9 = (-C1, +S1)
    // stage 2: 4-point DFTs over r of a[r + 4p]; result s lands in a[s + 4p] = X[p + 4s]
    radix4(a[0], a[1], a[2], a[3]);
    radix4(a[4], a[5], a[6], a[7]);
    radix4(a[8], a[9], a[10], a[11]);
    radix4(a[12], a[13], a[14], a[15]);
}

// dB of a squared magnitude q with dB = 10*log10(2)*log2(q) + off; q == 0 -> -150
// (scale_to_dbfs, analyzer.rs:11-27: val == 0.0 => -150.0)
__device__ __forceinline__ float db_from_sq(float q, float off)
{
    float r = fmaf(__log2f(q), 3.01029995663981195f, off);
    return q == 0.0f ? -150.0f : r;
}

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
constexpr int kX1Stride = 272;   // anyhop kernel: 256 + 16 de-phases the 4 ka-groups of a wave across banks
constexpr int kX2Stride = 17;    // anyhop kernel: row of 16 padded to 17, conflict-free b64 row reads
// pair kernel: both exchanges store rows of 16 complex padded to 18 (144 B): the reader's row is
// 16-B aligned and contiguous (8 x ds_read_b128), 16-lane write groups and 16-lane read groups both
// land on 16 distinct 4-bank slots (36*i mod 64 is a permutation of the multiples of 4).
constexpr int kRow = 18;
constexpr int kPlane = 16 * kRow;   // 288 complex per outer index; 16 planes = 4608 complex = 36864 B

// Published spectrum layout: bin k lives at k with bit 1 flipped when bit 6 is set.  A lane that owns four
// consecutive bins reads them as two aligned 16-byte pairs; the flip spreads the 16-lane groups of
// ds_read_b128 (and the 32-lane groups of the mirror's ds_read_b64) over distinct banks, while the
// publishing writes (16 consecutive k per 16-lane group) stay conflict-free.
#define SPEC_POS(k) ((k) ^ ((((k) >> 6) & 1) << 1))

// dB epilogue of one window.  xb holds the full spectrum Z[0..4095] in that order; a thread
// owns groups of FOUR consecutive retained bins (g = t, t + 256), so both output rows are written
// with 16-byte stores (rows are padded to a multiple of 4 floats): the 4-byte-per-lane stores of
// a stride-256 ownership were store-issue bound (1.6 ms of 4.7 ms at the config-3 size).
__device__ __forceinline__ void fft4096_epilogue(const v2f *xb, int t, uint32_t first_bin, uint32_t n_bins,
                                                 float db_offset, const float *__restrict__ offpink,
                                                 float *o_mid, float *o_side)
{
    const uint32_t ngroups = (n_bins + 3) >> 2;
    // dB = 10 log10(2) * log2(q) + (db_offset + pink[bin]); an exact zero reads -150 (+ pink): feed the
    // fma the log value that lands on -150 instead of selecting afterwards (one instruction less per bin)
    constexpr float kDb = 3.01029995663981195f;
    const float lg0 = (-150.0f - db_offset) / kDb;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t g = (uint32_t)t + 256u * i;
        if (g < ngroups) {
            const uint32_t k0 = first_bin + 4 * g;
            const float4 op = *reinterpret_cast<const float4 *>(offpink + 4 * g);   // table padded to the row stride
            const float opv[4] = {op.x, op.y, op.z, op.w};
            float rm[4], rs[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t k = k0 + e;                       // k <= 2051 < 4096: the mirror index stays positive
                const v2f zk = xb[SPEC_POS(k)];
                const v2f zm = xb[SPEC_POS(4096 - k)];           // Z[N - k]
                v2f m2, s2;                                      // 2*M = (zk.x+zm.x, zk.y-zm.y); 2*S ~ (zk.y+zm.y, zk.x-zm.x)
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(m2) : "v"(zk), "v"(zm));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(s2) : "v"(zk), "v"(zm));
                const float qm = fmaf(m2.x, m2.x, m2.y * m2.y);
                const float qs = fmaf(s2.x, s2.x, s2.y * s2.y);
                rm[e] = fmaf(qm == 0.0f ? lg0 : __log2f(qm), kDb, opv[e]);
                rs[e] = fmaf(qs == 0.0f ? lg0 : __log2f(qs), kDb, opv[e]);
            }
#if defined(SS_ABL) && SS_ABL == 1      /* ablation: no output stores */
            asm volatile("" ::"v"(rm[0]), "v"(rm[1]), "v"(rm[2]), "v"(rm[3]), "v"(rs[0]), "v"(rs[1]), "v"(rs[2]), "v"(rs[3]));
            (void)o_mid; (void)o_side;
#else
            reinterpret_cast<float4 *>(o_mid)[g] = make_float4(rm[0], rm[1], rm[2], rm[3]);
            reinterpret_cast<float4 *>(o_side)[g] = make_float4(rs[0], rs[1], rs[2], rs[3]);
#endif
        }
    }
}

// HS = hop / 256.  A workgroup iteration transforms TWO consecutive windows: they share the
// sliding sample registers (16 + HS slots) and every per-thread constant, and every barrier
// phase carries two independent radix-16 problems (half the barriers per window, twice the
// instruction-level parallelism to cover LDS latency).
#if defined(SS_ABL) && SS_ABL == 4      /* ablation: no butterflies */
#define SS_FFT16(z) asm volatile("" : "+v"(z[0]), "+v"(z[5]), "+v"(z[10]), "+v"(z[15]))
#else
#define SS_FFT16(z) fft16(z)
#endif
#if defined(SS_ABL) && SS_ABL == 5      /* ablation: no barriers (racy, timing only) */
#define SS_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define SS_SYNC() __syncthreads()
#endif
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

    for (uint32_t w = w_begin; w < w_end; w += 2) {
        const bool two = (w + 1 < w_end);
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
        for (int j = 0; j < 16; j++) z0[j] = v2f{sm[j] * hw[j], df[j] * hw[j]};
        // ---- pass 1 (the loop-end barrier has retired the previous pair's epilogue reads)
        SS_FFT16(z0);
        xbuf[0][X1W(0, tb, hi)] = z0[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) xbuf[0][X1W(ka, tb, hi)] = pk_cmul(z0[R16(ka)], tw1[ka]);
#pragma unroll
        for (int j = 0; j < 16; j++) z1[j] = v2f{sm[j + HS] * hw[j], df[j + HS] * hw[j]};
        SS_FFT16(z1);
        xbuf[1][X1W(0, tb, hi)] = z1[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) xbuf[1][X1W(ka, tb, hi)] = pk_cmul(z1[R16(ka)], tw1[ka]);
        SS_SYNC();
        // ---- pass 2 (thread = tb + 16 ka)
#pragma unroll
        for (int ta = 0; ta < 16; ta++) {
            z0[ta] = xbuf[0][X1W(hi, tb, ta)];
            z1[ta] = xbuf[1][X1W(hi, tb, ta)];
        }
        SS_SYNC();
        SS_FFT16(z0);
        xbuf[0][X2W(0, hi, tb)] = z0[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf[0][X2W(kb, hi, tb)] = pk_cmul(z0[R16(kb)], tw2s[tb * kb]);
        SS_FFT16(z1);
        xbuf[1][X2W(0, hi, tb)] = z1[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf[1][X2W(kb, hi, tb)] = pk_cmul(z1[R16(kb)], tw2s[tb * kb]);
        SS_SYNC();
        // ---- pass 3 (thread = ka + 16 kb): ka = tb, kb = hi
#pragma unroll
        for (int q = 0; q < 16; q++) {
            z0[q] = xbuf[0][X2W(hi, tb, q)];
            z1[q] = xbuf[1][X2W(hi, tb, q)];
        }
        SS_SYNC();
        SS_FFT16(z0);
        // ---- publish the whole spectrum in (swizzled) natural order: Z[t + 256 kc] at SPEC_POS(k)
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf[0][kc * 256 + tsw] = z0[R16(kc)];
        SS_FFT16(z1);
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf[1][kc * 256 + tsw] = z1[R16(kc)];
        SS_SYNC();
        // ---- epilogue: groups of four consecutive bins per thread, 16-byte stores
        float *o_mid = outp + (size_t)(w - w_begin) * out_win_stride;
        fft4096_epilogue(xbuf[0], t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride);
        if (two) fft4096_epilogue(xbuf[1], t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid + out_win_stride,
                                  o_mid + out_win_stride + p.bin_stride);
        // ---- slide the sample registers by two hops
        if (more) {
#pragma unroll
            for (int j = 0; j < NS - 2 * HS; j++) { sm[j] = sm[j + 2 * HS]; df[j] = df[j + 2 * HS]; }
#pragma unroll
            for (int q = 0; q < 2 * HS; q++) {
                if (NS - 2 * HS + q >= 0) { sm[NS - 2 * HS + q] = nx[q].x + nx[q].y; df[NS - 2 * HS + q] = nx[q].x - nx[q].y; }
            }
        }
        SS_SYNC();                             // epilogue reads are done before the next pair's pass-1 writes
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
#if defined(SS_FFT_PRIO_EPI)
#define SS_PRIO_EPI() 
#else
#define SS_PRIO_EPI() SS_PRIO_LO()
#endif
template <int HS, bool TW6>
__global__ __launch_bounds__(256, SS_FFT1_WAVES) void k_fft4096_ms1(FftBatchParams p)
{
    __shared__ __attribute__((aligned(16))) v2f xbuf[16 * kPlane];        // 36864 B
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];                //  2048 B
#define X1W(ka, tb_, ta_) ((ka) * kX1Stride + (tb_) + 16 * (ta_))
#define X2W(kb, ka_, tb_) ((kb) * kPlane + (ka_) * kRow + (tb_))
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
    float hw[16];
#pragma unroll
    for (int j = 0; j < 16; j++) hw[j] = p.half_window[t + 256 * j];
    v2f tw1[16];
    if (TW6) {
        tw1[1] = twn[t]; tw1[2] = twn[2 * t]; tw1[3] = twn[3 * t];
        tw1[4] = twn[4 * t]; tw1[8] = twn[8 * t]; tw1[12] = twn[12 * t];
    } else {
#pragma unroll
        for (int ka = 1; ka < 16; ka++) tw1[ka] = twn[t * ka];
    }
    tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[t];
    const int tb = t & 15, hi = t >> 4;
    const int tsw = SPEC_POS(t);
    const size_t out_win_stride = (size_t)2 * p.bin_stride;
    float *outp = p.out + ((size_t)stream * p.n_windows + w_begin) * out_win_stride;
    float sm[16], df[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float2 v = src[t + 256 * j];
        sm[j] = v.x + v.y;
        df[j] = v.x - v.y;
    }
    __syncthreads();
    for (uint32_t w = w_begin; w < w_end; ++w) {
        float2 nx[HS];
        const bool more = (w + 1 < w_end);
#pragma unroll
        for (int q = 0; q < HS; q++) {
            nx[q] = make_float2(0.f, 0.f);
            if (more) nx[q] = src[(size_t)(w + 1 - w_begin) * p.hop + t + 256 * (16 - HS + q)];
        }
        v2f z[16];
#pragma unroll
        for (int j = 0; j < 16; j++) z[j] = v2f{sm[j] * hw[j], df[j] * hw[j]};
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
        xbuf[X1W(0, tb, hi)] = z[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) {
            v2f v = z[R16(ka)];
            if (TW6) {
                if (ka & 3) v = pk_cmul(v, tw1[ka & 3]);
                if (ka & 12) v = pk_cmul(v, tw1[ka & 12]);
            } else {
                v = pk_cmul(v, tw1[ka]);
            }
            xbuf[X1W(ka, tb, hi)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ta = 0; ta < 16; ta++) z[ta] = xbuf[X1W(hi, tb, ta)];
        __syncthreads();
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
        xbuf[X2W(0, hi, tb)] = z[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf[X2W(kb, hi, tb)] = pk_cmul(z[R16(kb)], tw2s[tb * kb]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) z[q] = xbuf[X2W(hi, tb, q)];
        __syncthreads();
        SS_PRIO_LO();
        fft16(z);
        SS_PRIO_HI();
#pragma unroll
        for (int kc = 0; kc < 16; kc++)
            if ((p.publish_mask >> kc) & 1u) xbuf[kc * 256 + tsw] = z[R16(kc)];   // blocks with no retained bin or mirror are skipped
        __syncthreads();
        SS_PRIO_EPI();
        float *o_mid = outp + (size_t)(w - w_begin) * out_win_stride;
        fft4096_epilogue(xbuf, t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride);
        if (more) {
#pragma unroll
            for (int j = 0; j < 16 - HS; j++) { sm[j] = sm[j + HS]; df[j] = df[j + HS]; }
#pragma unroll
            for (int q = 0; q < HS; q++) { sm[16 - HS + q] = nx[q].x + nx[q].y; df[16 - HS + q] = nx[q].x - nx[q].y; }
        }
        __syncthreads();
    }
#undef X1W
#undef X2W
}

// generic hop (not a multiple of 256 or >= N/2 slots): one window per iteration, full reload
__global__ __launch_bounds__(256, SS_FFT_WAVES) void k_fft4096_ms_anyhop(FftBatchParams p)
{
    __shared__ __attribute__((aligned(16))) v2f xbuf[16 * kX1Stride];
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];
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
    for (uint32_t w = w_begin; w < w_end; ++w) {
        v2f z[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float2 v = src[(size_t)(w - w_begin) * p.hop + t + 256 * j];
            const float hwj = p.half_window[t + 256 * j];
            z[j] = v2f{(v.x + v.y) * hwj, (v.x - v.y) * hwj};
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
        fft4096_epilogue(xbuf, t, p.first_bin, p.n_bins, p.db_offset, p.offpink, o_mid, o_mid + p.bin_stride);
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
    __shared__ __attribute__((aligned(16))) v2f xbuf2[2][16 * kPlane];       // 2 x 36864 B
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];
#define X1W(ka, tb_, ta_) ((ka) * kX1Stride + (tb_) + 16 * (ta_))
#define X2W(kb, ka_, tb_) ((kb) * kPlane + (ka_) * kRow + (tb_))
    const int q = threadIdx.x >> 8;                 // which half-problem
    const int t = threadIdx.x & 255;
    v2f *xbuf = xbuf2[q];
    uint32_t bid = blockIdx.x;
    const uint32_t ch = bid % fft_ch; bid /= fft_ch;
    const uint32_t w = bid % p.n_windows;
    const uint32_t stream = bid / p.n_windows;
    if (p.windows_of && w >= p.windows_of[stream]) return;                          // ragged batches
    const size_t start = p.first_start + (size_t)w * p.hop;
    const float *base = p.pcm + ((size_t)stream * p.frames_per_stream + start) * p.channels;
    const v2f *tw16k = reinterpret_cast<const v2f *>(p.tw_n);       // W_16384^k, k < 8192
    const v2f *tw4k = reinterpret_cast<const v2f *>(p.tw_core);     // W_4096^k
    if (threadIdx.x < 256) tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[t];

    // windowed real samples 2i, 2i+1 as one complex value
    auto zload = [&](uint32_t i) -> v2f {
        float x0, x1;
        if (midside) {
            const float2 va = reinterpret_cast<const float2 *>(base)[2 * (size_t)i];       // frames 2i, 2i+1: (l,r)
            const float2 vb = reinterpret_cast<const float2 *>(base)[2 * (size_t)i + 1];
            x0 = ch == 0 ? (va.x + va.y) * 0.5f : (va.x - va.y) * 0.5f;
            x1 = ch == 0 ? (vb.x + vb.y) * 0.5f : (vb.x - vb.y) * 0.5f;
        } else {
            x0 = base[(size_t)(2 * i) * p.channels + ch];       // mono buffers have fft_ch == 1 => ch == 0
            x1 = base[(size_t)(2 * i + 1) * p.channels + ch];
        }
        const float2 hw = *reinterpret_cast<const float2 *>(p.window + 2 * (size_t)i);
        return v2f{x0 * hw.x, x1 * hw.y};
    };
    v2f z[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t i = (uint32_t)t + 256u * j;
        const v2f a = zload(i), b = zload(i + 4096u);
        z[j] = q ? pk_cmul(a - b, tw16k[2 * i]) : a + b;             // W_8192^i = W_16384^(2i)
    }
    const int tb = t & 15, hi = t >> 4;
    // ---- the 4096-point transform of y_q (same passes and LDS layouts as k_fft4096_ms)
    // pass-1 twiddles W^(t ka) from six gathered ones: W^(t ka) = W^(t (ka & 3)) * W^(t (ka & 12))
    // (scattered 8-byte gathers are the expensive part of this one-window-per-workgroup kernel)
    v2f twg[16];
    twg[1] = tw4k[t]; twg[2] = tw4k[2 * t]; twg[3] = tw4k[3 * t];
    twg[4] = tw4k[4 * t]; twg[8] = tw4k[8 * t]; twg[12] = tw4k[12 * t];
    fft16(z);
    xbuf[X1W(0, tb, hi)] = z[R16(0)];
#pragma unroll
    for (int ka = 1; ka < 16; ka++) {
        v2f v = z[R16(ka)];
        if (ka & 3) v = pk_cmul(v, twg[ka & 3]);
        if (ka & 12) v = pk_cmul(v, twg[ka & 12]);
        xbuf[X1W(ka, tb, hi)] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ta = 0; ta < 16; ta++) z[ta] = xbuf[X1W(hi, tb, ta)];
    __syncthreads();
    fft16(z);
    xbuf[X2W(0, hi, tb)] = z[R16(0)];
#pragma unroll
    for (int kb = 1; kb < 16; kb++) xbuf[X2W(kb, hi, tb)] = pk_cmul(z[R16(kb)], tw2s[tb * kb]);
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < 16; qq++) z[qq] = xbuf[X2W(hi, tb, qq)];
    __syncthreads();
    fft16(z);
    // publish Z_q[k] = Z[2k + q] at position k (natural order)
#pragma unroll
    for (int kc = 0; kc < 16; kc++) xbuf[kc * 256 + t] = z[R16(kc)];
    __syncthreads();
    // ---- real-FFT recombination + dB for the retained bins; consecutive threads own consecutive bins
    float *o = p.out + (((size_t)stream * p.n_windows + w) * fft_ch + ch) * p.bin_stride;
    for (uint32_t idx = threadIdx.x; idx < p.n_bins; idx += 512u) {
        const uint32_t b = p.first_bin + idx;
        float xr, xi;
        if (b == 8192u) {                                           // Nyquist of the real signal
            const v2f z0 = xbuf2[0][0];
            xr = z0.x - z0.y; xi = 0.0f;
        } else {
            const uint32_t qb = b & 1u, k = b >> 1;
            const uint32_t km = qb ? (4095u - k) : ((4096u - k) & 4095u);   // index of Z[8192 - b] in its half
            const v2f zk = xbuf2[qb][k];
            const v2f zc = xbuf2[qb][km];
            const float sr = (zk.x + zc.x) * 0.5f, si = (zk.y - zc.y) * 0.5f;
            const float dr = (zk.x - zc.x) * 0.5f, di = (zk.y + zc.y) * 0.5f;
            const v2f wv = tw16k[b];
            const float tr = wv.x * dr - wv.y * di;
            const float ti = wv.x * di + wv.y * dr;
            xr = sr + ti;
            xi = si - tr;
        }
        const float qv = fmaf(xr, xr, xi * xi);
        float r = fmaf(__log2f(qv), 3.01029995663981195f, p.db_offset);
        r = (qv == 0.0f) ? -150.0f : r;
        o[idx] = r + (p.pink ? p.pink[idx] : 0.0f);
    }
#undef X1W
#undef X2W
}

// ============================================================================
//  Spectrum, N = 16384 at hop 1024 for batches: one REAL channel per 512-thread workgroup that walks
//  `windows_per_block` consecutive windows with sliding sample registers (hop 1024 samples = one slot,
//  so a sample is fetched once per run instead of 16 times) and window weights rebuilt from two
//  resident twiddles, w[n0 + 1024 j] = 1/2 - 1/2 cos(a0 + j pi/8).
//  Decimation in time by four:  X[b] = A0[b] + W^b A1[b] + W^2b A2[b] + W^3b A3[b],  W = W_16384,
//  A_r = FFT_4096 of the real sequence xw[4i + r].  Half q (both are carried by every thread) transforms the complex
//  sequence z_q[i] = (xw[4i + 2q], xw[4i + 2q + 1]) on the radix-16 passes of k_fft4096_ms1; A_{2q}, A_{2q+1}
//  are its even / odd parts (the mid/side split of the N = 4096 kernel), combined per bin by Horner.
//  The two halves never exchange data before the epilogue.  MODE 0: mono buffer or channel `ch` of an
//  interleaved buffer, 1: stereo -> mid/side (audio_player.rs:400-419).
// ============================================================================
template <bool MIDSIDE>
__global__ __launch_bounds__(256, 2) void k_fft16k_run(FftBatchParams p, uint32_t fft_ch)
{
    __shared__ __attribute__((aligned(16))) v2f xbuf2[2][16 * kPlane];       // 2 x 36864 B
    __shared__ __attribute__((aligned(16))) v2f tw2s[256];                    //  2048 B
    __shared__ __attribute__((aligned(16))) float stage[4][256];              //  4096 B: 79872 B per workgroup, two per CU
#define X1W(ka, tb_, ta_) ((ka) * kX1Stride + (tb_) + 16 * (ta_))
#define X2W(kb, ka_, tb_) ((kb) * kPlane + (ka_) * kRow + (tb_))
    const int t = threadIdx.x;
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
    tw2s[t] = reinterpret_cast<const v2f *>(p.tw_256)[t];

    // samples n .. n + 3 of this workgroup's channel (n relative to the run's first window): the two
    // halves' complex inputs z_0 = (x[n], x[n+1]), z_1 = (x[n+2], x[n+3])
    auto ld4 = [&](size_t n, v2f &z0, v2f &z1) {
        if (MIDSIDE) {
            const float2 *f = reinterpret_cast<const float2 *>(base) + n;
            const float2 a = f[0], b = f[1], c = f[2], d = f[3];
            if (ch == 0) { z0 = v2f{(a.x + a.y) * 0.5f, (b.x + b.y) * 0.5f}; z1 = v2f{(c.x + c.y) * 0.5f, (d.x + d.y) * 0.5f}; }
            else { z0 = v2f{(a.x - a.y) * 0.5f, (b.x - b.y) * 0.5f}; z1 = v2f{(c.x - c.y) * 0.5f, (d.x - d.y) * 0.5f}; }
        } else {
            const float *f = base + n * C + ch;
            z0 = v2f{f[0], f[C]};
            z1 = v2f{f[2 * (size_t)C], f[3 * (size_t)C]};
        }
    };
    const uint32_t n0 = 4u * (uint32_t)t;           // slot j holds samples n0 + 1024 j .. +3 of the current window
    v2f raw0[16], raw1[16];
#pragma unroll
    for (int j = 0; j < 16; j++) ld4((size_t)n0 + 1024u * j, raw0[j], raw1[j]);
    // Hann weights of slot j: angle a_e + j pi/8, a_e = 2 pi (n0 + e) / 16384, e = 0..3 (table holds (cos, -sin))
    const v2f wa = tw16k[n0], wb = tw16k[n0 + 1], wc = tw16k[n0 + 2], wd = tw16k[n0 + 3];
    const v2f hc0 = {-0.5f * wa.x, -0.5f * wb.x}, hs0 = {-0.5f * wa.y, -0.5f * wb.y};   // -1/2 cos a_e, +1/2 sin a_e
    const v2f hc1 = {-0.5f * wc.x, -0.5f * wd.x}, hs1 = {-0.5f * wc.y, -0.5f * wd.y};
    const v2f half = {0.5f, 0.5f};
    v2f twg[16];
    twg[1] = tw4k[t]; twg[2] = tw4k[2 * t]; twg[3] = tw4k[3 * t];
    twg[4] = tw4k[4 * t]; twg[8] = tw4k[8 * t]; twg[12] = tw4k[12 * t];
    const int tb = t & 15, hi = t >> 4;
    const int tsw = SPEC_POS(t);
    const uint32_t ngroups = (p.n_bins + 3) >> 2;
    constexpr float kDb = 3.01029995663981195f;
    const float off2 = p.db_offset - 6.02059991327962390f;      // the epilogue carries 2 X
    __syncthreads();

    for (uint32_t w = w_begin; w < w_end; ++w) {
        const bool more = (w + 1 < w_end);
        v2f nx0 = {0.0f, 0.0f}, nx1 = {0.0f, 0.0f};
        if (more) ld4((size_t)(w + 1 - w_begin) * 1024u + n0 + 1024u * 15u, nx0, nx1);
        // the two halves are independent problems: every barrier phase carries both (half the barriers
        // per transform, two instruction streams to cover LDS latency)
        v2f z0[16], z1[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            // cos(j pi/8), sin(j pi/8)
            constexpr float cj[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                                      -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                                      -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f,
                                      0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
            constexpr float sj[16] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f,
                                      0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                                      -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                                      -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
            // w = 1/2 - 1/2 (cos a cj - sin a sj)
            z0[j] = raw0[j] * (half + hc0 * cj[j] + hs0 * sj[j]);
            z1[j] = raw1[j] * (half + hc1 * cj[j] + hs1 * sj[j]);
        }
        fft16(z0);
        xbuf2[0][X1W(0, tb, hi)] = z0[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) {
            v2f v = z0[R16(ka)];
            if (ka & 3) v = pk_cmul(v, twg[ka & 3]);
            if (ka & 12) v = pk_cmul(v, twg[ka & 12]);
            xbuf2[0][X1W(ka, tb, hi)] = v;
        }
        fft16(z1);
        xbuf2[1][X1W(0, tb, hi)] = z1[R16(0)];
#pragma unroll
        for (int ka = 1; ka < 16; ka++) {
            v2f v = z1[R16(ka)];
            if (ka & 3) v = pk_cmul(v, twg[ka & 3]);
            if (ka & 12) v = pk_cmul(v, twg[ka & 12]);
            xbuf2[1][X1W(ka, tb, hi)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ta = 0; ta < 16; ta++) { z0[ta] = xbuf2[0][X1W(hi, tb, ta)]; z1[ta] = xbuf2[1][X1W(hi, tb, ta)]; }
        __syncthreads();
        fft16(z0);
        xbuf2[0][X2W(0, hi, tb)] = z0[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf2[0][X2W(kb, hi, tb)] = pk_cmul(z0[R16(kb)], tw2s[tb * kb]);
        fft16(z1);
        xbuf2[1][X2W(0, hi, tb)] = z1[R16(0)];
#pragma unroll
        for (int kb = 1; kb < 16; kb++) xbuf2[1][X2W(kb, hi, tb)] = pk_cmul(z1[R16(kb)], tw2s[tb * kb]);
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 16; qq++) { z0[qq] = xbuf2[0][X2W(hi, tb, qq)]; z1[qq] = xbuf2[1][X2W(hi, tb, qq)]; }
        __syncthreads();
        fft16(z0);
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf2[0][kc * 256 + tsw] = z0[R16(kc)];   // Z_q[k] at SPEC_POS(k)
        fft16(z1);
#pragma unroll
        for (int kc = 0; kc < 16; kc++) xbuf2[1][kc * 256 + tsw] = z1[R16(kc)];
        __syncthreads();

        // ---- epilogue.  Iteration `it` covers 1024 retained bins, wave wv the 256 of them starting at
        // 1024 it + 256 wv.  Reading: lane l takes bins +l, +64+l, +128+l, +192+l, so every LDS read of the two
        // published spectra (and of their mirrors, descending) is stride-1 across the wave: conflict-free
        // whatever first_bin is.  Writing: the four dB values go through a wave-private 1 KB staging row so
        // that lane l stores bins +4l..+4l+3 with one 16-byte store.  From one iteration to the next the bin
        // index grows by 1024: positions move by +-1024 (SPEC_POS only looks at bits 1 and 6), twiddles turn by W_16^1.
        float *o = p.out + (((size_t)stream * p.n_windows + w) * fft_ch + ch) * p.bin_stride;
        {
            const uint32_t lane = (uint32_t)t & 63u, wv = (uint32_t)t >> 6;
            float *stg = stage[wv];
            uint32_t pb[4], pm[4];
            v2f wt[4];
            uint32_t fb = p.first_bin;
            asm volatile("" : "+s"(fb));        // per-window recomputation: hoisting these 16 registers out of the loop spills
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t b = fb + 256u * wv + 64u * e + lane;
                const uint32_t bq = b & 4095u, mq = (4096u - bq) & 4095u;
                pb[e] = SPEC_POS(bq); pm[e] = SPEC_POS(mq);
                wt[e] = tw16k[b & 8191u];
            }
            const v2f rho = {0.92387953251128674f, -0.38268343236508977f};
            const uint32_t n_iter = (4u * ngroups + 1023u) >> 10;
            for (uint32_t it = 0; it < n_iter; it++) {
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const v2f e0 = xbuf2[0][pb[e]], em = xbuf2[0][pm[e]];
                    const v2f o0 = xbuf2[1][pb[e]], om = xbuf2[1][pm[e]];
                    v2f a0, a1, a2, a3;        // 2 A_r[b]
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(a0) : "v"(e0), "v"(em));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(a1) : "v"(e0), "v"(em));
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(a2) : "v"(o0), "v"(om));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(a3) : "v"(o0), "v"(om));
                    v2f x = pk_cmul(a3, wt[e]) + a2;
                    x = pk_cmul(x, wt[e]) + a1;
                    x = pk_cmul(x, wt[e]) + a0;
                    const float qv = fmaf(x.x, x.x, x.y * x.y);
                    const float db = fmaf(__log2f(qv), kDb, off2);
                    r[e] = qv == 0.0f ? -150.0f : db;
                    pb[e] = (pb[e] + 1024u) & 4095u;
                    pm[e] = (pm[e] - 1024u) & 4095u;
                    wt[e] = pk_cmul(wt[e], rho);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) stg[64 * e + lane] = r[e];
                __builtin_amdgcn_wave_barrier();                      // LDS is in order per wave: ordering is all that is needed
                const float4 v = reinterpret_cast<const float4 *>(stg)[lane];
                __builtin_amdgcn_wave_barrier();
                const uint32_t g = 256u * it + 64u * wv + lane;        // group of four bins this lane stores
                if (g < ngroups) {
                    float4 pk = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.pink) pk = *reinterpret_cast<const float4 *>(p.pink + 4u * g);
                    reinterpret_cast<float4 *>(o)[g] = make_float4(v.x + pk.x, v.y + pk.y, v.z + pk.z, v.w + pk.w);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 15; j++) { raw0[j] = raw0[j + 1]; raw1[j] = raw1[j + 1]; }
            raw0[15] = nx0; raw1[15] = nx1;
        }
        __syncthreads();                    // epilogue reads are done before the next window's pass-1 writes
    }
#undef X1W
#undef X2W
}

// runs of windows at hop 1024 (batches); `mode` as launch_fft16k
hipError_t launch_fft16k_run(FftBatchParams p, int mode, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0 || p.n_bins == 0) return hipSuccess;
    const uint32_t fft_ch = (mode == 0) ? 1u : (mode == 1 ? 2u : p.channels);
    // enough workgroups to fill 256 CUs x 2, runs of at least 16 windows (the run's first window costs a full load)
    const uint64_t pairs = (uint64_t)p.n_streams * fft_ch;
    uint32_t groups = (uint32_t)((4096 + pairs - 1) / pairs);
    const uint32_t max_groups = p.n_windows / 16u ? p.n_windows / 16u : 1u;
    if (groups > max_groups) groups = max_groups;
    if (groups < 1) groups = 1;
    p.windows_per_block = (p.n_windows + groups - 1) / groups;
    groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    const dim3 grid((uint32_t)((pairs * groups + 7) & ~(uint64_t)7)), block(256);      // multiple of 8: see the XCD mapping in the kernel
    if (mode == 1) hipLaunchKernelGGL(k_fft16k_run<true>, grid, block, 0, s, p, fft_ch);
    else hipLaunchKernelGGL(k_fft16k_run<false>, grid, block, 0, s, p, fft_ch);
    return hipGetLastError();
}

hipError_t launch_fft16k(const FftBatchParams &p, int mode, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0 || p.n_bins == 0) return hipSuccess;
    const uint32_t fft_ch = (mode == 0) ? 1u : (mode == 1 ? 2u : p.channels);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft16k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 0);
        (void)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_fft16k, dim3(p.n_streams * p.n_windows * fft_ch), dim3(512), 0, s, p, mode == 1 ? 1 : 0, fft_ch);
    return hipGetLastError();
}

hipError_t launch_fft4096_ms(const FftBatchParams &p, hipStream_t s)
{
    if (p.n_windows == 0 || p.n_streams == 0) return hipSuccess;
    const uint32_t groups = (p.n_windows + p.windows_per_block - 1) / p.windows_per_block;
    dim3 grid(groups * p.n_streams), block(256);
    // hop 1024 (the reference's cadence): the single-window kernel at 3 workgroups per CU measured 2.5 %
    // faster than the window-pair kernel at 2 (A/B in one process, 3.48 vs 3.57 ms); -DSS_FFT_PAIR selects the latter
#if defined(SS_FFT_PAIR)
    if (p.hop == 1024) hipLaunchKernelGGL(k_fft4096_ms<4>, grid, block, 0, s, p);
#else
    if (p.hop == 1024) hipLaunchKernelGGL((k_fft4096_ms1<4, true>), grid, block, 0, s, p);
#endif
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
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_generic),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(p.n_streams * p.n_windows * fft_ch), block(256);
    hipLaunchKernelGGL(k_fft_generic, grid, block, lds, s, p, mode, log2m);
    return hipGetLastError();
}

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
                for (uint32_t gi = 0; gi < ngroups; gi++) {             // channel counts that do not divide 16
                    const uint32_t q = gi * 16 + (uint32_t)mrow;
                    const uint32_t bi = q / C, c = q - bi * C;
                    const bool col_ok = q < ncol;
                    const float *bp = tile + ((int)(bi * Cfg::BLK) - (Cfg::HIST - 1) + kq) * (int)C + (int)c;
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < Cfg::KSTEPS; s++)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], col_ok ? bp[4 * s * (int)C] : 0.0f, acc, 0, 0, 0);
                    float m = 0.0f;
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int row = 4 * kq + reg;
                        const bool ok = col_ok && (bi * Cfg::BLK + (uint32_t)(row % Cfg::BLK)) < seg;
                        m = fmaxf(m, ok ? fabsf(acc[reg]) : 0.0f);
                    }
                    if (col_ok) atomicMax(&tpk[c], __float_as_uint(m));
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
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
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
        if (p.channels == 2) {
            const int fast = td_wave_int4(p);
            if (fast == 2) return td_launch<FACTOR, false, 2, 2>(p, s);
            if (fast == 3) return td_launch<FACTOR, false, 2, 3>(p, s);
            return td_launch<FACTOR, false, 2, 1>(p, s);
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

// one wave evaluates gate and LRA on an LDS histogram pair (block, short-term)
__device__ void eval_hist(const unsigned long long *hb, const unsigned long long *hs,
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
        unsigned long long above = 0;
        for (int i = lane; i < kHistBins; i += 64) if ((uint32_t)i >= index) above += hs[i];
        above = wave_sum_u64(above);
        if (above != 0 && lane == 0) {
            const unsigned long long plow = (unsigned long long)((double)(above - 1) * 0.1 + 0.5);
            const unsigned long long phigh = (unsigned long long)((double)(above - 1) * 0.95 + 0.5);
            unsigned long long acc = 0; uint32_t j = index;
            while (acc <= plow) acc += hs[j++];
            const double l_en = en[j - 1];
            while (acc <= phigh) acc += hs[j++];
            const double h_en = en[j - 1];
            lra = (10.0 * log10(h_en) - 0.691) - (10.0 * log10(l_en) - 0.691);
        }
        lra = __shfl(lra, 0, 64);
    }
    if (lane == 0) { if (out_i) *out_i = integrated; if (out_lra) *out_lra = lra; }
}

__global__ __launch_bounds__(64) void k_finalize(FinalizeParams p)
{
    __shared__ unsigned long long hb[kHistBins];
    __shared__ unsigned long long hs[kHistBins];
    __shared__ unsigned int counts[2];
    const uint32_t stream = blockIdx.x;
    const int lane = threadIdx.x;
    unsigned long long *gh = reinterpret_cast<unsigned long long *>(p.hist) + (size_t)stream * 2 * kHistBins;
    unsigned long long *corpus = reinterpret_cast<unsigned long long *>(p.corpus_hist);
    for (int i = lane; i < kHistBins; i += 64) { hb[i] = gh[i]; hs[i] = gh[kHistBins + i]; }
    if (lane < 2) counts[lane] = 0;
    __syncthreads();

    const uint32_t C = p.channels;
    const double S = (double)p.k->s100;
    const double *P = p.subblocks + (size_t)stream * p.sub_stride;
    // gating block ending with sub-block j: j-3..j ; short-term block: j-29..j when (j-29) % 10 == 0
    const uint64_t sub_end = p.sub_end_of ? p.sub_end_of[stream] : p.sub_end;          // ragged batches
    for (uint64_t j = p.sub_begin + lane; j < sub_end; j += 64) {
        if (j >= 3) {
            double sum = 0.0;
            for (uint32_t c = 0; c < C; c++) {
                const double w = p.weights[c];
                if (w == 0.0) continue;
                double cs = 0.0;
                for (int q = 3; q >= 0; q--) cs += P[(size_t)((j - q) % p.sub_cap) * C + c];
                sum += w * cs;
            }
            sum /= 4.0 * S;
            atomicAdd(&counts[0], 1u);
            if (sum >= p.hist_bounds[0]) atomicAdd(&hb[hist_index(p.hist_bounds, sum)], 1ull);
        }
        if (j >= 29 && (j - 29) % 10 == 0) {
            double sum = 0.0;
            for (uint32_t c = 0; c < C; c++) {
                const double w = p.weights[c];
                if (w == 0.0) continue;
                double cs = 0.0;
                for (int q = 29; q >= 0; q--) cs += P[(size_t)((j - q) % p.sub_cap) * C + c];
                sum += w * cs;
            }
            sum /= 30.0 * S;
            atomicAdd(&counts[1], 1u);
            if (sum >= p.hist_bounds[0]) atomicAdd(&hs[hist_index(p.hist_bounds, sum)], 1ull);
        }
    }
    __syncthreads();
    // corpus contribution = what this call added
    for (int i = lane; i < kHistBins; i += 64) {
        const unsigned long long db = hb[i] - gh[i], ds = hs[i] - gh[kHistBins + i];
        if (corpus) {
            if (db) atomicAdd(&corpus[i], db);
            if (ds) atomicAdd(&corpus[kHistBins + i], ds);
        }
        gh[i] = hb[i];
        gh[kHistBins + i] = hs[i];
    }
    if (p.out_counts && lane < 2) p.out_counts[stream * 2 + lane] += counts[lane];
    eval_hist(hb, hs, p.hist_energies, p.hist_bounds,
              p.out_integrated ? &p.out_integrated[stream] : nullptr,
              p.out_lra ? &p.out_lra[stream] : nullptr);
}

// Streaming form (one handle, a few new sub-blocks per call, no per-call read-out): the same gating rules
// with the histogram updated in place by global atomics instead of a 16 KB round trip through LDS.
__global__ __launch_bounds__(64) void k_finalize_stream(FinalizeParams p)
{
    const int lane = threadIdx.x;
    unsigned long long *gh = reinterpret_cast<unsigned long long *>(p.hist);
    const uint32_t C = p.channels;
    const double S = (double)p.k->s100;
    const double *P = p.subblocks;
    uint32_t nb = 0, ns = 0;
    for (uint64_t j = p.sub_begin + lane; j < p.sub_end; j += 64) {
        if (j >= 3) {
            double sum = 0.0;
            for (uint32_t c = 0; c < C; c++) {
                const double w = p.weights[c];
                if (w == 0.0) continue;
                double cs = 0.0;
                for (int q = 3; q >= 0; q--) cs += P[(size_t)((j - q) % p.sub_cap) * C + c];
                sum += w * cs;
            }
            sum /= 4.0 * S;
            nb++;
            if (sum >= p.hist_bounds[0]) atomicAdd(&gh[hist_index(p.hist_bounds, sum)], 1ull);
        }
        if (j >= 29 && (j - 29) % 10 == 0) {
            double sum = 0.0;
            for (uint32_t c = 0; c < C; c++) {
                const double w = p.weights[c];
                if (w == 0.0) continue;
                double cs = 0.0;
                for (int q = 29; q >= 0; q--) cs += P[(size_t)((j - q) % p.sub_cap) * C + c];
                sum += w * cs;
            }
            sum /= 30.0 * S;
            ns++;
            if (sum >= p.hist_bounds[0]) atomicAdd(&gh[kHistBins + hist_index(p.hist_bounds, sum)], 1ull);
        }
    }
    if (p.out_counts) {
        if (nb) atomicAdd(&p.out_counts[0], nb);
        if (ns) atomicAdd(&p.out_counts[1], ns);
    }
}

hipError_t launch_finalize(const FinalizeParams &p, hipStream_t s)
{
    if (p.n_streams == 0) return hipSuccess;
    const bool streaming = p.n_streams == 1 && !p.corpus_hist && !p.out_integrated && !p.out_lra && p.sub_stride == 0;
    if (streaming) hipLaunchKernelGGL(k_finalize_stream, dim3(1), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(k_finalize, dim3(p.n_streams), dim3(64), 0, s, p);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void k_hist_eval(const unsigned long long *hist2000, const double *en,
                                                  const double *bd, double *out2)
{
    __shared__ unsigned long long hb[kHistBins];
    __shared__ unsigned long long hs[kHistBins];
    for (int i = threadIdx.x; i < kHistBins; i += 64) { hb[i] = hist2000[i]; hs[i] = hist2000[kHistBins + i]; }
    __syncthreads();
    eval_hist(hb, hs, en, bd, &out2[0], &out2[1]);
}

hipError_t launch_hist_eval(const uint64_t *hist2000, const double *energies, const double *bounds,
                            double *out2, hipStream_t s)
{
    hipLaunchKernelGGL(k_hist_eval, dim3(1), dim3(64), 0, s,
                       reinterpret_cast<const unsigned long long *>(hist2000), energies, bounds, out2);
    return hipGetLastError();
}

// mean square over the last `frames` frames of the filtered ring, channel-weighted
// (calc_gating_block on the ring "as is": loudness_shortterm / loudness_momentary).
// Two stages with a fixed reduction shape (bit-reproducible): kRingBlocks partial sums, then one block.
constexpr int kRingBlocks = 96;
__global__ __launch_bounds__(256) void k_ring_energy(const double *ring, uint64_t ring_frames, uint32_t C,
                                                     uint64_t end_frame, uint64_t frames,
                                                     const double *weights, double *partial)
{
    __shared__ double red[256];
    double acc = 0.0;
    const uint64_t total = frames * C;
    // ring position of absolute frame f is f % ring_frames; frames before 0 are the zeroed ring
    const uint64_t begin = end_frame + ring_frames * 4 - frames;    // keep the subtraction non-negative
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)kRingBlocks * 256) {
        const uint64_t f = i / C; const uint32_t c = (uint32_t)(i - f * C);
        const double w = weights[c];
        const double y = ring[((begin + f) % ring_frames) * C + c];
        acc = fma(w * y, y, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(128) void k_ring_final(const double *partial, uint64_t frames, double *out)
{
    __shared__ double red[128];
    red[threadIdx.x] = threadIdx.x < kRingBlocks ? partial[threadIdx.x] : 0.0;
    __syncthreads();
    for (int s = 64; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double e = red[0] / (double)frames;
        out[0] = e;
        out[1] = e <= 0.0 ? -INFINITY : 10.0 * log10(e) - 0.691;   // energy_to_loudness
    }
}

hipError_t launch_ring_energy(const double *ring, uint64_t ring_frames, uint32_t channels,
                              uint64_t end_frame, uint64_t frames, const double *weights,
                              double *out, double *scratch, hipStream_t s)
{
    hipLaunchKernelGGL(k_ring_energy, dim3(kRingBlocks), dim3(256), 0, s, ring, ring_frames, channels,
                       end_frame % ring_frames, frames, weights, scratch);
    hipLaunchKernelGGL(k_ring_final, dim3(1), dim3(128), 0, s, scratch, frames, out);
    return hipGetLastError();
}

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
    for (uint64_t j = start + lane16; j < end; j += 16) {
        const float v = x[j];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
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

}  // namespace ssk
