// Instruction-rate microbenchmarks for gfx950 (calibrates DESIGN.md's cost model):
// hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double seed, int n)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double c = 0.999999, d = 1e-9;
    const float cf = 0.999999f, df = 1e-9f;
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {            // 8 independent f64 FMA
            a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = fma(a3, c, d);
            a4 = fma(a4, c, d); a5 = fma(a5, c, d); a6 = fma(a6, c, d); a7 = fma(a7, c, d);
        } else if (MODE == 1) {     // 8 independent f32 FMA
            f0 = fmaf(f0, cf, df); f1 = fmaf(f1, cf, df); f2 = fmaf(f2, cf, df); f3 = fmaf(f3, cf, df);
            f4 = fmaf(f4, cf, df); f5 = fmaf(f5, cf, df); f6 = fmaf(f6, cf, df); f7 = fmaf(f7, cf, df);
        } else if (MODE == 2) {     // 1 dependent f64 FMA chain (latency)
            a0 = fma(a0, c, d); a0 = fma(a0, c, d); a0 = fma(a0, c, d); a0 = fma(a0, c, d);
            a0 = fma(a0, c, d); a0 = fma(a0, c, d); a0 = fma(a0, c, d); a0 = fma(a0, c, d);
        } else if (MODE == 3) {     // f32 add (non-FMA)
            f0 += cf; f1 += cf; f2 += cf; f3 += cf; f4 += cf; f5 += cf; f6 += cf; f7 += cf;
        } else if (MODE == 4) {     // cvt f32->f64 + f64 add
            a0 += (double)f0; a1 += (double)f1; a2 += (double)f2; a3 += (double)f3;
            f0 += cf; f1 += cf; f2 += cf; f3 += cf;
        } else if (MODE == 5) {     // ds_bpermute (shfl)
            f0 = __shfl_up(f0, 2, 64); f1 = __shfl_up(f1, 2, 64); f2 = __shfl_up(f2, 2, 64); f3 = __shfl_up(f3, 2, 64);
            f4 = __shfl_up(f4, 2, 64); f5 = __shfl_up(f5, 2, 64); f6 = __shfl_up(f6, 2, 64); f7 = __shfl_up(f7, 2, 64);
        } else if (MODE == 6) {     // 1 dependent f32 FMA chain
            f0 = fmaf(f0, cf, df); f0 = fmaf(f0, cf, df); f0 = fmaf(f0, cf, df); f0 = fmaf(f0, cf, df);
            f0 = fmaf(f0, cf, df); f0 = fmaf(f0, cf, df); f0 = fmaf(f0, cf, df); f0 = fmaf(f0, cf, df);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int MODE>
void run(const char *name, int waves_per_simd, double ops_per_iter)
{
    double *out;
    const int blocks = 256 * waves_per_simd;     // 4 waves per block -> waves_per_simd per SIMD
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0, N_ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD
    const double instr = (double)waves_per_simd * N_ITERS * ops_per_iter;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.2GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.2);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("f64 fma x8 indep", w, 8);
        run<1>("f32 fma x8 indep", w, 8);
        run<3>("f32 add x8 indep", w, 8);
        run<2>("f64 fma dependent", w, 8);
        run<6>("f32 fma dependent", w, 8);
        run<4>("cvt+f64 add+f32 add (x4)", w, 12);
        run<5>("ds_bpermute x8", w, 8);
    }
    return 0;
}
