// Issue cost of the f16 matrix instructions next to the f32 one, and whether f16 MFMA of one wave overlaps
// f64 / f32 VALU work of another wave on the same SIMD (gfx950).  cycles = ms * clock / (iters * 4 mfma).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define N_ITERS 4096
// role: 0 idle, 1 mfma f32 16x16x4, 2 f64 fma x8, 3 f32 fma x8, 4 mfma f16 16x16x16, 5 mfma f16 16x16x32, 6 mfma f64 16x16x4
__global__ __launch_bounds__(512) void k(float *out, int roleA, int roleB, int n)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    float r = 0.f;
    floatx4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    if (role == 1) {
        float x = threadIdx.x * 1e-3f, y = 1.0f + x;
        for (int i = 0; i < n; i++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        }
        r = a0[0] + a1[1];
    } else if (role == 4) {
        half4 x = {(_Float16)(threadIdx.x * 1e-3f), 1, 2, 3}, y = {1, 2, (_Float16)(threadIdx.x * 2e-3f), 4};
        for (int i = 0; i < n; i++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(y, x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(y, x, a1, 0, 0, 0);
        }
        r = a0[0] + a1[1];
    } else if (role == 5) {
        half8 x = {(_Float16)(threadIdx.x * 1e-3f), 1, 2, 3, 4, 5, 6, 7}, y = {1, 2, (_Float16)(threadIdx.x * 2e-3f), 4, 5, 6, 7, 8};
        for (int i = 0; i < n; i++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, a1, 0, 0, 0);
        }
        r = a0[0] + a1[1];
    } else if (role == 6) {
        typedef double doublex4 __attribute__((ext_vector_type(4)));
        doublex4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
        double x = threadIdx.x * 1e-3, y = 1.0 + x;
        for (int i = 0; i < n; i++) {
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, d1, 0, 0, 0);
        }
        r = (float)(d0[0] + d1[1]);
    } else if (role == 2) {
        double b0 = threadIdx.x, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7;
        const double c = 0.999999, d = 1e-9;
        for (int i = 0; i < n; i++) {
            b0 = fma(b0, c, d); b1 = fma(b1, c, d); b2 = fma(b2, c, d); b3 = fma(b3, c, d);
            b4 = fma(b4, c, d); b5 = fma(b5, c, d); b6 = fma(b6, c, d); b7 = fma(b7, c, d);
        }
        r = (float)(b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7);
    } else if (role == 3) {
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
        const float cf = 0.999999f, df = 1e-9f;
        for (int i = 0; i < n; i++) {
            f0 = fmaf(f0, cf, df); f1 = fmaf(f1, cf, df); f2 = fmaf(f2, cf, df); f3 = fmaf(f3, cf, df);
            f4 = fmaf(f4, cf, df); f5 = fmaf(f5, cf, df); f6 = fmaf(f6, cf, df); f7 = fmaf(f7, cf, df);
        }
        r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
static float run(int a, int b)
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, a, b, 16); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, a, b, N_ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(out); return ms;
}
int main()
{
    const char *nm[] = {"idle", "mfma_f32x4", "fma_f64", "fma_f32", "mfma_f16x16", "mfma_f16x32", "mfma_f64x4"};
    int pairs[][2] = {{1, 0}, {4, 0}, {5, 0}, {6, 0}, {2, 0}, {3, 0}, {4, 2}, {4, 3}, {5, 2}, {6, 2}, {6, 3}, {6, 4}, {2, 3}, {4, 4}, {6, 6}, {1, 1}};
    for (auto &p : pairs) printf("%-11s + %-11s : %.3f ms  (%.1f ns per inner iteration)\n", nm[p[0]], nm[p[1]], run(p[0], p[1]),
                                 run(p[0], p[1]) * 1e6 / N_ITERS);
    return 0;
}
