// A/B for north_star's "MFMA only if the FFT is recast as a dense twiddle GEMM and rocprof shows it beating the shuffle path":
// the radix-16 step of the spectrum kernels (a 16-point complex DFT per lane-thread) done
//   (A) the way k_fft4096_ms1 / k_fft16k_run do it: 81 packed-f32 VALU instructions in registers, and
//   (B) as a dense 16 x 16 complex DFT-matrix product on the matrix cores at f32-equivalent accuracy:
//       D = W X with W = Wr + i Wi constant, X = 16 points x 16 columns per MFMA tile,
//       Dr = Wr Xr - Wi Xi, Di = Wr Xi + Wi Xr: four real 16x16x16 products, each as the three-term f16 split
//       (W_hi X_hi + W_hi X_lo + W_lo X_hi, the split the true-peak kernel uses; f32 accumulate, ~2^-21 relative) ->
//       12 v_mfma_f32_16x16x16_f16 per 16 DFTs, plus the f32 -> (hi, lo) f16 conversion of X (2 v_fma_mix per real).
//   (B') the same with the operands already converted (upper bound for B: conversion not charged).
// Both run as throughput loops on every CU; reported: ns per 16-point DFT per CU and the ratio.  Accuracy of (B) is
// checked against (A) on random data.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench_dft16.hip -o tools/bin/ubench_dft16 && tools/bin/ubench_dft16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2f pk_sub_ib(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_add_ib(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w)
{
    v2f m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(m));
    return r;
}
__device__ __forceinline__ v2f pk_w2pre(v2f a) { return pk_sub_ib(a, a); }
__device__ __forceinline__ v2f pk_w6pre(v2f a)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ v2f pk_mul_mi(v2f a)
{
    v2f r;
    const v2f zero = {0.0f, 0.0f};
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(zero), "v"(a));
    return r;
}
__device__ __forceinline__ void radix4(v2f &a0, v2f &a1, v2f &a2, v2f &a3)
{
    const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
    a0 = t0 + t2; a2 = t0 - t2; a1 = pk_sub_ib(t1, t3); a3 = pk_add_ib(t1, t3);
}
#define R16(k) ((((k) & 3) << 2) | ((k) >> 2))
__device__ __forceinline__ void fft16(v2f (&a)[16])            // the product's radix-16 step (ss_fft.hip)
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    const v2f w1 = {C1, -S1}, w3 = {S1, -C1}, w9 = {-C1, S1};
    radix4(a[0], a[4], a[8], a[12]); radix4(a[1], a[5], a[9], a[13]); radix4(a[2], a[6], a[10], a[14]); radix4(a[3], a[7], a[11], a[15]);
    a[5] = pk_cmul(a[5], w1); a[9] = pk_w2pre(a[9]) * R; a[13] = pk_cmul(a[13], w3);
    a[6] = pk_w2pre(a[6]) * R; a[10] = pk_mul_mi(a[10]); a[14] = pk_w6pre(a[14]) * R;
    a[7] = pk_cmul(a[7], w3); a[11] = pk_w6pre(a[11]) * R; a[15] = pk_cmul(a[15], w9);
    radix4(a[0], a[1], a[2], a[3]); radix4(a[4], a[5], a[6], a[7]); radix4(a[8], a[9], a[10], a[11]); radix4(a[12], a[13], a[14], a[15]);
}

// (A) VALU: every lane-thread transforms its own 16 points; `iters` transforms back to back (outputs feed the next one,
// scaled so that nothing overflows), 64 DFTs per wave per iteration
__global__ __launch_bounds__(256) void k_valu(float2 *io, int iters)
{
    v2f a[16];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = v2f{io[base + j].x, io[base + j].y};
    for (int it = 0; it < iters; it++) {
        fft16(a);
#pragma unroll
        for (int j = 0; j < 16; j++) a[j] *= 0.25f;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) io[base + j] = make_float2(a[R16(j)].x, a[R16(j)].y);
}

// (B) MFMA: one wave transforms 16 columns per tile.  Lane (n = lane & 15, kq = lane >> 4) holds X[4 kq + j][n] (B operand)
// and W[n][4 kq + j] (A operand, row n of the DFT matrix); D[4 kq + r][n] comes back in the accumulator.
template <bool CONVERT>
__global__ __launch_bounds__(256) void k_mfma(float2 *io, const float2 *wtab, int iters, int check)
{
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    halfx4 wr_hi, wr_lo, wi_hi, wi_lo, nwi_hi, nwi_lo;            // W = Wr + i Wi (rows n, k = 4 kq + j), and -Wi
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 w = wtab[n * 16 + 4 * kq + j];
        wr_hi[j] = (_Float16)w.x; wr_lo[j] = (_Float16)(w.x - (float)wr_hi[j]);
        wi_hi[j] = (_Float16)w.y; wi_lo[j] = (_Float16)(w.y - (float)wi_hi[j]);
        nwi_hi[j] = -wi_hi[j]; nwi_lo[j] = -wi_lo[j];
    }
    const size_t tile = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 256;    // 16 points x 16 columns per wave
    float xr[4], xi[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const float2 v = io[tile + (size_t)n * 16 + 4 * kq + j]; xr[j] = v.x; xi[j] = v.y; }
    halfx4 xrh, xrl, xih, xil;
    auto split = [&]() {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            xrh[j] = (_Float16)xr[j]; xrl[j] = (_Float16)(xr[j] - (float)xrh[j]);
            xih[j] = (_Float16)xi[j]; xil[j] = (_Float16)(xi[j] - (float)xih[j]);
        }
    };
    split();
    floatx4 dr = {0, 0, 0, 0}, di = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        if (CONVERT) split();
        else asm volatile("" : "+v"(xrh), "+v"(xrl), "+v"(xih), "+v"(xil));      // pre-split operands, opaque so the loop is not hoisted
        dr = floatx4{0, 0, 0, 0}; di = floatx4{0, 0, 0, 0};
        // Dr = Wr Xr - Wi Xi ; Di = Wr Xi + Wi Xr ; each product = hi hi + hi lo + lo hi
        dr = __builtin_amdgcn_mfma_f32_16x16x16f16(wr_hi, xrh, dr, 0, 0, 0);
        di = __builtin_amdgcn_mfma_f32_16x16x16f16(wr_hi, xih, di, 0, 0, 0);
        dr = __builtin_amdgcn_mfma_f32_16x16x16f16(wr_hi, xrl, dr, 0, 0, 0);
        di = __builtin_amdgcn_mfma_f32_16x16x16f16(wr_hi, xil, di, 0, 0, 0);
        dr = __builtin_amdgcn_mfma_f32_16x16x16f16(wr_lo, xrh, dr, 0, 0, 0);
        di = __builtin_amdgcn_mfma_f32_16x16x16f16(wr_lo, xih, di, 0, 0, 0);
        dr = __builtin_amdgcn_mfma_f32_16x16x16f16(nwi_hi, xih, dr, 0, 0, 0);
        di = __builtin_amdgcn_mfma_f32_16x16x16f16(wi_hi, xrh, di, 0, 0, 0);
        dr = __builtin_amdgcn_mfma_f32_16x16x16f16(nwi_hi, xil, dr, 0, 0, 0);
        di = __builtin_amdgcn_mfma_f32_16x16x16f16(wi_hi, xrl, di, 0, 0, 0);
        dr = __builtin_amdgcn_mfma_f32_16x16x16f16(nwi_lo, xih, dr, 0, 0, 0);
        di = __builtin_amdgcn_mfma_f32_16x16x16f16(wi_lo, xrh, di, 0, 0, 0);
        if (!check) {
            // feed the next transform (timing loop only: D is in the accumulator layout, the values just have to change)
#pragma unroll
            for (int j = 0; j < 4; j++) { xr[j] = dr[j] * 0.25f; xi[j] = di[j] * 0.25f; }
            if (!CONVERT) { asm volatile("" :: "v"(xr[0]), "v"(xi[0]), "v"(xr[1]), "v"(xi[1]), "v"(xr[2]), "v"(xi[2]), "v"(xr[3]), "v"(xi[3])); }
        }
    }
    // D[4 kq + r][n]
#pragma unroll
    for (int r = 0; r < 4; r++) io[tile + (size_t)n * 16 + 4 * kq + r] = make_float2(dr[r], di[r]);
}

int main()
{
    const int blocks = 256 * 8, threads = 256;                    // 8 workgroups per CU
    const size_t n_valu = (size_t)blocks * threads * 16;
    std::vector<float2> h(n_valu), wt(256);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float2((float)std::sin(0.37 * i) * 0.7f, (float)std::cos(0.11 * i + 1) * 0.7f);
    for (int r = 0; r < 16; r++)
        for (int k = 0; k < 16; k++) wt[r * 16 + k] = make_float2((float)std::cos(-2 * M_PI * r * k / 16), (float)std::sin(-2 * M_PI * r * k / 16));
    float2 *d, *d2, *w;
    hipMalloc(&d, n_valu * 8); hipMalloc(&d2, n_valu * 8); hipMalloc(&w, 256 * 8);
    hipMemcpy(w, wt.data(), 256 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    // ---- accuracy: one transform of the same 16 columns x 16 points both ways
    hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipMemcpy(d2, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_valu, dim3(1), dim3(256), 0, 0, d, 1);              // thread t: points [16 t, 16 t + 16)
    hipLaunchKernelGGL(k_mfma<true>, dim3(4), dim3(256), 0, 0, d2, w, 1, 1);    // wave: columns n of a tile = same 16-point groups
    std::vector<float2> a(4096), b(4096);
    hipMemcpy(a.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d2, 4096 * 8, hipMemcpyDeviceToHost);
    double emax = 0, amax = 0;
    for (int i = 0; i < 4096; i++) {
        emax = std::fmax(emax, std::hypot((double)a[i].x * 4 - b[i].x, (double)a[i].y * 4 - b[i].y));   // (A) scales by 1/4
        amax = std::fmax(amax, std::hypot((double)b[i].x, (double)b[i].y));
    }
    printf("accuracy: max |MFMA f16x3 - VALU f32| = %.3g relative to the largest output %.3g -> %.3g\n", emax, amax, emax / amax);
    // ---- throughput
    const int iters = 2048;
    hipMemcpy(d, h.data(), n_valu * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(threads), 0, 0, d, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double dft_valu = (double)blocks * threads * iters;                // one DFT per thread per iteration
    const double ns_valu = ms * 1e6 / (dft_valu / 256.0);
    printf("(A) VALU fft16 (81 packed-f32 instr / DFT)            : %8.3f ms  %.3f ns per DFT-16 per CU\n", ms, ns_valu);
    double ns_b[2];
    for (int conv = 1; conv >= 0; conv--) {
        hipMemcpy(d2, h.data(), (size_t)blocks * 4 * 256 * 8, hipMemcpyHostToDevice);
        if (conv) hipLaunchKernelGGL(k_mfma<true>, dim3(blocks), dim3(threads), 0, 0, d2, w, 8, 0);
        else hipLaunchKernelGGL(k_mfma<false>, dim3(blocks), dim3(threads), 0, 0, d2, w, 8, 0);
        hipEventRecord(e0);
        if (conv) hipLaunchKernelGGL(k_mfma<true>, dim3(blocks), dim3(threads), 0, 0, d2, w, iters, 0);
        else hipLaunchKernelGGL(k_mfma<false>, dim3(blocks), dim3(threads), 0, 0, d2, w, iters, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double dft = (double)blocks * 4 * 16 * iters;                  // 16 DFTs per wave per iteration
        ns_b[conv] = ms * 1e6 / (dft / 256.0);
        printf("(B%s) MFMA f16x3 dense DFT matrix (12 MFMA / 16 DFTs)%s: %8.3f ms  %.3f ns per DFT-16 per CU  (%.2fx the VALU path)\n",
               conv ? " " : "'", conv ? ", f32->f16 split charged    " : ", operands pre-split (bound)", ms, ns_b[conv], ns_b[conv] / ns_valu);
    }
    return 0;
}
