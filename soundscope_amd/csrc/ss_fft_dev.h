// ss_fft_dev.h: device-side pieces of the spectrum kernels that more than one translation unit compiles — the packed-f32 complex
// arithmetic, the register-resident 16-point DFT, and the one-window N = 16384 transform (k_fft16k of ss_fft.hip; the tick
// kernel of ss_time_domain.hip runs it in the workgroups beside the loudness chain's).
#pragma once
#include "ss_kernels.h"

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
// one ds_read_b64 (see kRowB): a volatile load in the LDS address space is neither merged with its neighbours nor widened
__device__ __forceinline__ v2f lds_ld64(const v2f *p)
{
    return *(const volatile __attribute__((address_space(3))) v2f *)p;
}
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
// (ONE asm statement for the dependent pair: behind every asm statement whose output the next VALU instruction reads, hipcc
// pads an s_nop — written as two statements a complex multiply carried two of them, eighty issue slots per window in
// k_fft4096_ms1; the hardware interlocks a VALU result read by the next VALU instruction by itself)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w)
{
    v2f m, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,0]"
        : "=v"(r), "=&v"(m) : "v"(a), "v"(w));
    return r;
}
// a * w for a compile-time constant w held in a scalar register pair (the three constants of fft16: left to the "v"
// constraint of pk_cmul the compiler copies them into vector registers in front of every use, ten v_mov_b64 per window)
__device__ __forceinline__ v2f pk_cmul_k(v2f a, v2f w)
{
    v2f m, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,0]"
        : "=v"(r), "=&v"(m) : "v"(a), "s"(w));
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

// a * w.lo (S = 0) or a * w.hi (S = 1) in both halves: one real weight out of a register pair that holds two
template <int S>
__device__ __forceinline__ v2f pk_mul_bcast(v2f a, v2f w)
{
    v2f r;
    if (S == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(w));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(w));
    return r;
}
// (l + r, l - r) of one stereo frame (l, r)
__device__ __forceinline__ v2f pk_sum_diff(v2f f)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(f));
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
    a[5] = pk_cmul_k(a[5], w1);            // r=1,p=1: W^1
    a[9] = pk_w2pre(a[9]) * R;           // r=1,p=2: W^2
    a[13] = pk_cmul_k(a[13], w3);          // r=1,p=3: W^3
    a[6] = pk_w2pre(a[6]) * R;           // r=2,p=1: W^2
    a[10] = pk_mul_mi(a[10]);            // r=2,p=2: W^4 = -i
    a[14] = pk_w6pre(a[14]) * R;         // r=2,p=3: W^6 = (-R,-R)
    a[7] = pk_cmul_k(a[7], w3);            // r=3,p=1: W^3
    a[11] = pk_w6pre(a[11]) * R;         // r=3,p=2: W^6
    a[15] = pk_cmul_k(a[15], w9);          // r=3,p=3: W^9 = (-C1, +S1)
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


constexpr int kX1Stride = 272;   // 256 + 16 de-phases the 4 ka-groups of a wave across banks
constexpr int kRow = 18;
constexpr int kPlane = 16 * kRow;   // 288 complex per outer index; 16 planes = 4608 complex = 36864 B

// ============================================================================
//  Spectrum, N = 16384 (the reference's native window, tui.rs:1488), ONE window of one REAL channel per
//  512-thread workgroup: real FFT through an 8192-point complex FFT, split by one radix-2
//  decimation-in-frequency step into two 4096-point problems that reuse the radix-16 machinery:
//    z[i] = (xw[2i], xw[2i+1]),  y_q[i] = (z[i] + (-1)^q z[i+4096]) W_8192^(i q),  Z[2k+q] = FFT_4096(y_q)[k]
//    X[b] = (Z[b] + conj Z[8192-b])/2 - (i/2) W_16384^b (Z[b] - conj Z[8192-b])
//  Threads 0-255 run q = 0, threads 256-511 run q = 1; the mirror 8192-b has the parity of b, so each
//  half only mirrors inside its own published spectrum.  midside 0: mono buffer or channel `ch` of an interleaved one,
//  1: stereo -> mid/side (audio_player.rs:400-419).  `bid`: (stream, window, channel) index of this workgroup.
//  `lds`: kFft16kLdsBytes of workgroup memory, 16-byte aligned (k_fft16k's own static array; the tick kernel's dynamic block, which
//  its other workgroups use for the loudness call's tiles — a kernel's LDS is the largest need, not the sum).
//  The epilogue asks for the twiddles (and pink values) of eight of a thread's bins before it uses the first: a tick runs
//  TWO of these workgroups on an otherwise idle chip, and one exposed memory round trip per bin was most of its 17 us.
// ============================================================================
constexpr int kFft16kLdsBytes = (2 * 16 * kPlane + 256) * 8;                // 75776
__device__ __forceinline__ void fft16k_window(const FftBatchParams &p, int midside, uint32_t fft_ch, uint32_t bid, void *lds)
{
    v2f (*const xbuf2)[16 * kPlane] = reinterpret_cast<v2f (*)[16 * kPlane]>(lds);      // [2][16 * kPlane]: 2 x 36864 B
    v2f *const tw2s = reinterpret_cast<v2f *>(lds) + 2 * 16 * kPlane;                    // [256]
#define X1W(ka, tb_, ta_) ((ka) * kX1Stride + (tb_) + 16 * (ta_))
#define X2W(kb, ka_, tb_) ((kb) * kPlane + (ka_) * kRow + (tb_))
    const int q = threadIdx.x >> 8;                 // which half-problem
    const int t = threadIdx.x & 255;
    v2f *xbuf = xbuf2[q];
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
        } else if (p.channels == 1) {
            const float2 v = reinterpret_cast<const float2 *>(base)[i];                     // mono buffer: samples 2i, 2i+1 in one load
            x0 = v.x; x1 = v.y;
        } else {
            x0 = base[(size_t)(2 * i) * p.channels + ch];
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
    constexpr int kAhead = 8;
    for (uint32_t idx0 = threadIdx.x; idx0 < p.n_bins; idx0 += 512u * kAhead) {
        v2f wv[kAhead];
        float pk[kAhead];
#pragma unroll
        for (int m = 0; m < kAhead; m++) {
            const uint32_t idx = idx0 + 512u * (uint32_t)m;
            const uint32_t ic = idx < p.n_bins ? idx : p.n_bins - 1u;      // (clamped: a load, never a store)
            const uint32_t b = p.first_bin + ic;
            wv[m] = tw16k[b < 8192u ? b : 0u];
            pk[m] = p.pink ? p.pink[ic] : 0.0f;
        }
#pragma unroll
        for (int m = 0; m < kAhead; m++) {
            const uint32_t idx = idx0 + 512u * (uint32_t)m;
            if (idx >= p.n_bins) break;
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
                const float tr = wv[m].x * dr - wv[m].y * di;
                const float ti = wv[m].x * di + wv[m].y * dr;
                xr = sr + ti;
                xi = si - tr;
            }
            const float qv = fmaf(xr, xr, xi * xi);
            float r = fmaf(__log2f(qv), 3.01029995663981195f, p.db_offset);
            r = (qv == 0.0f) ? -150.0f : r;
            o[idx] = r + pk[m];
        }
    }
    if (p.done_flag) {
        // every wave completes its own stores system-wide (a workgroup barrier alone does not wait for vector stores), then one
        // lane tells the host: whoever sees the flag sees the row
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(p.done_flag + ch, p.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#undef X1W
#undef X2W
}

}  // namespace ssk
