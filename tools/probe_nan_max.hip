// NaN behaviour of the maxima the columns-only epilogue relies on: v_max_f32, v_max3_f32, ds_max_f32 (gfx950, IEEE mode of a HIP kernel)
// hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_nan_max tools/probe_nan_max.hip && tools/bin/probe_nan_max
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef __attribute__((address_space(3))) float lds_f32;
__global__ void k(const float *in, float *out)
{
    __shared__ float acc[4];
    const float a = in[0], b = in[1], c = in[2];      // NaN, +inf, 1
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); out[0] = r;     // (NaN, inf, 1)
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "v"(c)); out[1] = r;     // (inf, NaN, 1)
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(c), "v"(c), "v"(a)); out[2] = r;     // (1, 1, NaN)
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(a), "v"(a)); out[3] = r;     // (NaN, NaN, NaN)
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); out[4] = r;                  // (NaN, inf)
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(c), "v"(a)); out[5] = r;                  // (1, NaN)
    acc[0] = -INFINITY; acc[1] = -INFINITY; acc[2] = NAN; acc[3] = 2.0f;
    __syncthreads();
    lds_f32 *p = (lds_f32 *)acc;
    (void)__hip_atomic_fetch_max(p + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // -inf <- NaN
    (void)__hip_atomic_fetch_max(p + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // -inf <- +inf
    (void)__hip_atomic_fetch_max(p + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // NaN  <- 1
    (void)__hip_atomic_fetch_max(p + 3, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // 2    <- NaN
    __syncthreads();
    for (int i = 0; i < 4; i++) out[6 + i] = acc[i];
}
int main()
{
    float h[3] = {NAN, INFINITY, 1.0f}, *d, *o, r[10];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    const char *names[10] = {"v_max3(NaN, inf, 1)", "v_max3(inf, NaN, 1)", "v_max3(1, 1, NaN)", "v_max3(NaN, NaN, NaN)", "v_max(NaN, inf)", "v_max(1, NaN)",
                             "ds_max_f32: -inf <- NaN", "ds_max_f32: -inf <- +inf", "ds_max_f32: NaN <- 1", "ds_max_f32: 2 <- NaN"};
    for (int i = 0; i < 10; i++) printf("%-26s = %g\n", names[i], r[i]);
    return 0;
}
