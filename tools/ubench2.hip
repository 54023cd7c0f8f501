// Does f32-input MFMA overlap with f64/f32 VALU work of ANOTHER wave on the same SIMD (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define N_ITERS 4096
// 512 threads = 8 waves: waves 0-3 take role A, waves 4-7 role B (wave w and w+4 share a SIMD)
// role: 0 idle, 1 = mfma f32 16x16x4 (2 chains), 2 = f64 fma x8, 3 = f32 fma x8
__global__ __launch_bounds__(512) void k(float *out, int roleA, int roleB, int n)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    float r = 0.f;
    if (role == 1) {
        floatx4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        float x = threadIdx.x * 1e-3f, y = 1.0f + x;
        for (int i = 0; i < n; i++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        }
        r = a0[0] + a1[1];
    } else if (role == 2) {
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const double c = 0.999999, d = 1e-9;
        for (int i = 0; i < n; i++) {
            a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = fma(a3, c, d);
            a4 = fma(a4, c, d); a5 = fma(a5, c, d); a6 = fma(a6, c, d); a7 = fma(a7, c, d);
        }
        r = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
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
    const char *nm[] = {"idle", "mfma_f32", "fma_f64", "fma_f32"};
    int pairs[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 2}, {1, 3}, {2, 3}, {1, 1}, {2, 2}, {3, 3}};
    for (auto &p : pairs) printf("%-9s + %-9s : %.3f ms\n", nm[p[0]], nm[p[1]], run(p[0], p[1]));
    return 0;
}
