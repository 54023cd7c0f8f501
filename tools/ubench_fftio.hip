// The spectrum kernel's HBM traffic pattern with no arithmetic: what would the chip take for just the loads and stores
// of k_fft4096_ms1 at the bench shape?  4096 workgroups x 256 threads, each walking 116 "windows": per window 4 float2
// loads per thread (8 KB per workgroup, contiguous 2 KB pieces) and two rows of 1708 floats stored as float4
// (13.7 KB per workgroup) — 10.4 GB per launch, 62 % of it stores.  Variants: loads only / stores only / both.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fftio.hip -o tools/bin/ubench_fftio && tools/bin/ubench_fftio
#include <hip/hip_runtime.h>
#include <cstdio>

// plain streaming stores: every thread one float4 per iteration, fully contiguous, no LDS (full occupancy)
template <bool NT>
__global__ __launch_bounds__(256) void k_flat(float4 *__restrict__ out, size_t n4)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(f4v{1.f, 2.f, 3.f, (float)i}, reinterpret_cast<f4v *>(out) + i);
        else out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
    }
}

// flat copy-shaped kernel: the same byte counts as the spectrum kernel (3.89 GB read, 6.49 GB written), fully contiguous, full
// occupancy, no LDS: what the memory system gives a MIXED read/write stream of this ratio whatever the access pattern
__global__ __launch_bounds__(256) void k_flat_mixed(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n_in4, size_t n_out4)
{
    float acc = 0.f;
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, j = i;
    // five stores per three loads (6.49 : 3.89)
    for (; j < n_out4; ) {
        for (int q = 0; q < 3 && i < n_in4; q++, i += step) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
        for (int q = 0; q < 5 && j < n_out4; q++, j += step) out[j] = make_float4(acc, 1.f, 2.f, 3.f);
    }
}

// LDSPAD = false: the same per-workgroup row pattern at FULL occupancy (no LDS footprint): separates the pattern from the
// three-workgroups-per-CU occupancy the real kernel runs at
template <int MODE, int STRIDE = 1708, bool LDSPAD = true>   // 1 loads, 2 stores, 3 both
__global__ __launch_bounds__(256, 3) void k_io(const float2 *__restrict__ in, float *__restrict__ out, int nwin, size_t frames_per_stream, int groups)
{
    __shared__ float lds[LDSPAD ? 9728 : 256];            // the real kernel's 38.9 KB: same occupancy (3 workgroups per CU)
    const int t = threadIdx.x;
    const uint32_t stream = blockIdx.x / groups, grp = blockIdx.x % groups;
    const float2 *src = in + (size_t)stream * frames_per_stream + 1024 + (size_t)grp * nwin * 1024;
    float *o = out + ((size_t)stream * groups * nwin + (size_t)grp * nwin) * 2 * STRIDE;
    float acc = 0.f;
    lds[t] = 0.f;
    for (int w = 0; w < nwin; w++) {
        if (MODE & 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) { const float2 v = src[(size_t)w * 1024 + t + 256 * (12 + q)]; acc += v.x + v.y; }
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int g = t + 256 * i;
                if (g < 427) {
                    const float4 v = make_float4(acc, acc + 1.f, acc + 2.f, (float)w);
                    reinterpret_cast<float4 *>(o + (size_t)w * 2 * STRIDE)[g] = v;
                    reinterpret_cast<float4 *>(o + (size_t)w * 2 * STRIDE + STRIDE)[g] = v;
                }
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc + lds[t];
}

int main()
{
    const int streams = 1024, groups = 4, nwin = 116;
    const size_t frames = 480000;
    float2 *in; float *out;
    hipMalloc(&in, (size_t)streams * frames * 8);
    hipMalloc(&out, (size_t)streams * groups * nwin * 2 * 1728 * 4);
    hipMemset(in, 0, (size_t)streams * frames * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double in_gb = (double)streams * groups * nwin * 1024 * 8 / 1e9, out_gb = (double)streams * groups * nwin * 2 * 1708 * 4 / 1e9;
    for (int mode = 1; mode <= 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 1) hipLaunchKernelGGL(k_io<1>, dim3(streams * groups), dim3(256), 0, 0, in, out, nwin, frames, groups);
            if (mode == 2) hipLaunchKernelGGL(k_io<2>, dim3(streams * groups), dim3(256), 0, 0, in, out, nwin, frames, groups);
            if (mode == 3) hipLaunchKernelGGL(k_io<3>, dim3(streams * groups), dim3(256), 0, 0, in, out, nwin, frames, groups);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double gb = (mode & 1 ? in_gb : 0) + (mode & 2 ? out_gb : 0);
        printf("%-12s %.3f ms  %.2f GB -> %.2f TB/s\n", mode == 1 ? "loads only" : mode == 2 ? "stores only" : "loads+stores", best, gb, gb / best);
    }
    // aligned rows (1728 floats = 54 cache lines)
    {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_io<2, 1728>), dim3(streams * groups), dim3(256), 0, 0, in, out, nwin, frames, groups);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-12s %.3f ms  %.2f GB -> %.2f TB/s (rows padded to 1728 floats, same useful bytes)\n", "stores/align", best, out_gb, out_gb / best);
    }
    // the row pattern at full occupancy (no LDS footprint)
    for (int mode = 2; mode <= 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 2) hipLaunchKernelGGL((k_io<2, 1708, false>), dim3(streams * groups), dim3(256), 0, 0, in, out, nwin, frames, groups);
            else hipLaunchKernelGGL((k_io<3, 1708, false>), dim3(streams * groups), dim3(256), 0, 0, in, out, nwin, frames, groups);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double gb = (mode & 1 ? in_gb : 0) + out_gb;
        printf("%-12s %.3f ms  %.2f GB -> %.2f TB/s (row pattern, no LDS footprint: full occupancy)\n", mode == 2 ? "stores/occ" : "both/occ", best, gb, gb / best);
    }
    // a flat, contiguous MIXED stream of the same read and write byte counts at full occupancy
    {
        const size_t n_in4 = (size_t)(in_gb * 1e9 / 16), n_out4 = (size_t)(out_gb * 1e9 / 16);
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_flat_mixed, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), n_in4, n_out4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-12s %.3f ms  %.2f GB -> %.2f TB/s (flat contiguous loads + stores, same byte counts, full occupancy)\n", "flat mixed", best, in_gb + out_gb, (in_gb + out_gb) / best);
    }
    // flat contiguous stores of the same byte count
    for (int nt = 0; nt < 2; nt++) {
        const size_t n4 = (size_t)(out_gb * 1e9 / 16);
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (nt) hipLaunchKernelGGL(k_flat<true>, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<float4 *>(out), n4);
            else hipLaunchKernelGGL(k_flat<false>, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<float4 *>(out), n4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-12s %.3f ms  %.2f GB -> %.2f TB/s\n", nt ? "flat nt" : "flat", best, out_gb, out_gb / best);
    }
    return 0;
}
