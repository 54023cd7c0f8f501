// Does a wave issuing v_mfma_f32_16x16x32_f16 (gfx950's K = 32 f16 form) disturb the arithmetic of ANOTHER wave on the same
// SIMD?  Round 2 saw isolated wrong spectrum windows when the time-domain kernel used that instruction beside the spectrum
// kernel (never with the K = 16 form); round 3 reproduced it on every pass (profiles/r03_race_k32_*.txt: two independent
// batches on two HIP streams are enough).  This isolates it: one workgroup of 8 waves, waves 0-3 = "victim" (a deterministic
// register-only or LDS-only instruction stream whose results are compared bit for bit with a run beside idle partners),
// waves 4-7 = "aggressor" (nothing but matrix instructions on constant operands: no memory, no LDS).  Waves w and w + 4 share
// a SIMD.  Prints the number of victim lanes whose results differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));

// victim: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_pk_add_f32 with op_sel / neg (complex rotate-add), 3 v_pk_mul_f32, 4 LDS b64 ring,
//         5 v_fma_f64, 6 v_log_f32 + v_exp_f32
// aggressor: 0 idle, 1 mfma f32 16x16x4 f32, 2 mfma f32 16x16x16 f16 (K = 16), 3 mfma f32 16x16x32 f16 (K = 32),
//            4 mfma f32 32x32x16 f16 (the other gfx950 double-K f16 form), 5 mfma f32 16x16x32 bf16
__global__ __launch_bounds__(512) void k(uint32_t *out, int victim, int aggr, int n)
{
    __shared__ v2f ring[4][64 * 9];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {
        floatx4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
        float r = 0.f;
        if (aggr == 1) {
            float x = lane * 1e-3f, y = 1.0f + x;
            for (int i = 0; i < n; i++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
            }
        } else if (aggr == 2) {
            half4 x = {(_Float16)(lane * 1e-3f), 1, 2, 3}, y = {1, 2, (_Float16)(lane * 2e-3f), 4};
            for (int i = 0; i < n; i++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x16f16(y, y, a3, 0, 0, 0);
            }
        } else if (aggr == 3) {
            half8 x = {(_Float16)(lane * 1e-3f), 1, 2, 3, 4, 5, 6, 7}, y = {1, 2, (_Float16)(lane * 2e-3f), 4, 5, 6, 7, 8};
            for (int i = 0; i < n; i++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, y, a3, 0, 0, 0);
            }
        } else if (aggr == 4) {
            half8 x = {(_Float16)(lane * 1e-3f), 1, 2, 3, 4, 5, 6, 7}, y = {1, 2, (_Float16)(lane * 2e-3f), 4, 5, 6, 7, 8};
            floatx16 c0 = {0}, c1 = {0};
            for (int i = 0; i < n; i++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, c1, 0, 0, 0);
            }
            a0[0] = c0[0] + c1[5];
        } else if (aggr == 5) {
            bf8 x, y;
            for (int j = 0; j < 8; j++) { x[j] = (__bf16)(float)(j + lane * 1e-3f); y[j] = (__bf16)(float)(8 - j); }
            for (int i = 0; i < n; i++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
        if (r == 1.2345e33f) out[0] = 1;          // keep the results alive
        return;
    }
    // ---- victim: eight independent chains per lane, values stay in [0.5, 2)
    const int gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t res[8];
    if (victim == 0 || victim == 2 || victim == 3) {
        v2f z[8];
        for (int j = 0; j < 8; j++) z[j] = v2f{1.0f + 0.001f * (float)((gid * 8 + j) % 997), 1.5f - 0.0007f * (float)((gid + 3 * j) % 613)};
        const v2f c = {0.99999f, 1.00001f}, d = {1e-6f, -1e-6f}, h = {0.5f, 0.5f};
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (victim == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(z[j]) : "v"(z[j]), "v"(c), "v"(d));
                else if (victim == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(z[j]) : "v"(z[j]), "v"(c));
                else {                              // (a - i b) / 2 ... keeps |z| bounded: a = z, b = z rotated
                    v2f t;
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(t) : "v"(z[j]), "v"(z[(j + 1) & 7]));
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(z[j]) : "v"(t), "v"(h));
                }
            }
        }
        for (int j = 0; j < 8; j++) res[j] = __float_as_uint(z[j].x) ^ (__float_as_uint(z[j].y) * 2654435761u);
    } else if (victim == 1) {
        float f[8];
        for (int j = 0; j < 8; j++) f[j] = 1.0f + 0.001f * (float)((gid * 8 + j) % 997);
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[j]) : "v"(f[j]), "v"(0.99999f), "v"(1e-6f)); }
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[j]) : "v"(f[j]), "v"(1.00001f), "v"(-1e-6f)); }
        }
        for (int j = 0; j < 8; j++) res[j] = __float_as_uint(f[j]);
    } else if (victim == 5) {
        double f[8];
        for (int j = 0; j < 8; j++) f[j] = 1.0 + 0.001 * (double)((gid * 8 + j) % 997);
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) f[j] = fma(f[j], 0.999999, 1e-9);
        }
        for (int j = 0; j < 8; j++) res[j] = (uint32_t)__double2loint(f[j]) ^ (uint32_t)__double2hiint(f[j]);
    } else if (victim == 6) {
        float f[8];
        for (int j = 0; j < 8; j++) f[j] = 1.0f + 0.001f * (float)((gid * 8 + j) % 997);
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) f[j] = __builtin_amdgcn_exp2f(__log2f(f[j]) * 0.99999f) + 1e-6f;
        }
        for (int j = 0; j < 8; j++) res[j] = __float_as_uint(f[j]);
    } else {                                        // 4: LDS ring, wave-private: write 8 b64 values, read them back rotated by 9 lanes
        v2f z[8];
        for (int j = 0; j < 8; j++) z[j] = v2f{(float)(gid * 8 + j), (float)(gid - j)};
        v2f *mine = ring[wave];
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) mine[lane * 9 + j] = z[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 8; j++) z[j] = mine[((lane + 9) & 63) * 9 + ((j + 3) & 7)];
            __builtin_amdgcn_wave_barrier();
        }
        for (int j = 0; j < 8; j++) res[j] = __float_as_uint(z[j].x) ^ (__float_as_uint(z[j].y) * 2654435761u);
    }
    for (int j = 0; j < 8; j++) out[(size_t)gid * 8 + j] = res[j];
}

int main(int argc, char **argv)
{
    const int blocks = 1024, n = argc > 1 ? atoi(argv[1]) : 20000, reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t words = (size_t)blocks * 256 * 8;
    uint32_t *out; hipMalloc(&out, words * 4);
    std::vector<uint32_t> ref(words), got(words);
    const char *vn[] = {"v_pk_fma_f32", "v_fma_f32", "v_pk_add_f32(op_sel)+pk_mul", "v_pk_mul_f32", "LDS b64 ring", "v_fma_f64", "v_log/v_exp f32"};
    const char *an[] = {"idle", "mfma 16x16x4 f32", "mfma 16x16x16 f16", "mfma 16x16x32 f16", "mfma 32x32x16 f16", "mfma 16x16x32 bf16"};
    for (int v = 0; v < 7; v++) {
        hipMemset(out, 0, words * 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, v, 0, n); hipDeviceSynchronize();
        hipMemcpy(ref.data(), out, words * 4, hipMemcpyDeviceToHost);
        for (int a = 0; a < 6; a++) {
            long bad_lanes = 0, bad_runs = 0;
            float ms_tot = 0;
            for (int r = 0; r < reps; r++) {
                hipMemset(out, 0, words * 4);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, v, a, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms_tot += ms;
                hipMemcpy(got.data(), out, words * 4, hipMemcpyDeviceToHost);
                long b = 0;
                for (size_t i = 0; i < words; i += 8) b += memcmp(&got[i], &ref[i], 32) != 0;
                bad_lanes += b; bad_runs += b != 0;
            }
            printf("victim %-28s beside %-18s : %ld of %d runs differ, %ld wrong lanes of %zu  (%.2f ms per run)\n", vn[v], an[a], bad_runs, reps,
                   bad_lanes, (size_t)reps * blocks * 256, ms_tot / reps);
            fflush(stdout);
        }
    }
    hipFree(out);
    return 0;
}
