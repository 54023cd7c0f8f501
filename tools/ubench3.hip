// packed-f32 VALU rates on gfx950: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 vs scalar
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float seed, int n)
{
    v2f a0 = {seed + threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const v2f c = {0.999999f, 1.000001f}, d = {1e-9f, 2e-9f};
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {        // pk_fma
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (MODE == 1) { // pk_add
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
        } else if (MODE == 2) { // pk_mul
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 3) { // scalar add x8 (reference)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(d.x));
        } else if (MODE == 4) { // v_log_f32 x8
            asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                         "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));
        } else if (MODE == 5) { // v_mov_b32 x8
            asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                         "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(d.x));
        }
    }
    v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
}
template <int MODE> void run(const char *name, int wps)
{
    float *out; const int blocks = 256 * wps; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 16); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, N_ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)wps * N_ITERS * 8;
    printf("%-14s waves/SIMD=%d %.3f ms -> %.2f cycles/wave-instr @2.2GHz\n", name, wps, ms, ms * 1e6 / instr * 2.2);
    hipFree(out);
}
int main()
{
    for (int w : {2, 4}) {
        run<0>("v_pk_fma_f32", w); run<1>("v_pk_add_f32", w); run<2>("v_pk_mul_f32", w);
        run<3>("v_add_f32", w); run<4>("v_log_f32", w); run<5>("v_mov_b32", w);
    }
    return 0;
}
