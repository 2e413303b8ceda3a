// probe: MFMA + VALU co-issue on one SIMD.  Each wave runs REPS x [G groups of (1 MFMA + V independent VALU fma)] .
// variants: waves per SIMD 1 or 2; V = 0, 2, 4, 6, 8, 12, 16; also exp (transcendental) instead of fma.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, bool TRANS, bool AG>
__global__ __launch_bounds__(512) void k(float* out, int reps, long long* cyc) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    f32x16 acc0 = {}, acc1 = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b)); else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                if (TRANS) v[q % 16] = __builtin_amdgcn_exp2f(v[q % 16]);
                else v[q % 16] = __builtin_fmaf(v[q % 16], 1.0001f, 0.5f);
            }
            if (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b)); else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                if (TRANS) v[(q + 8) % 16] = __builtin_amdgcn_exp2f(v[(q + 8) % 16]);
                else v[(q + 8) % 16] = __builtin_fmaf(v[(q + 8) % 16], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V, bool TRANS, bool AG>
void run(int threads, float* out, long long* cyc, int grid = 256) {
    int reps = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, TRANS, AG>), dim3(grid), dim3(threads), 0, 0, out, 200, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, TRANS, AG>), dim3(grid), dim3(threads), 0, 0, out, reps, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double per_group = (double)c / (reps * 8.0);   // cycles per (2 MFMA + 2V VALU) per wave
    double ns_group = ms * 1e6 / (reps * 8.0);
    double tf = 2.0 * 32 * 32 * 16 * 2 * (reps * 8.0) * (threads / 64) * grid / (ms * 1e-3) / 1e12;
    printf("grid %3d %s waves/SIMD %d  V=%2d %s: %6.1f s_memtime ticks, %6.1f ns per [2 MFMA + %2d VALU] per wave -> tick rate %.2f GHz, %.0f TFLOP/s\n", grid, AG ? "AGPR" : "VGPR", threads / 256, V,
           TRANS ? "exp" : "fma", per_group, ns_group, 2 * V, per_group / ns_group, tf);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    for (int threads : {256, 512}) {
        run<0, false, false>(threads, out, cyc); run<4, false, false>(threads, out, cyc); run<8, false, false>(threads, out, cyc); run<16, false, false>(threads, out, cyc);
        run<0, false, true>(threads, out, cyc); run<4, false, true>(threads, out, cyc); run<8, false, true>(threads, out, cyc); run<16, false, true>(threads, out, cyc);
        run<8, true, false>(threads, out, cyc); run<8, true, true>(threads, out, cyc);
    }
    // the same on 8 workgroups only: far below the package power cap, so what is left is the issue behaviour alone
    for (int threads : {256, 512}) {
        run<0, false, false>(threads, out, cyc, 8); run<4, false, false>(threads, out, cyc, 8); run<8, false, false>(threads, out, cyc, 8);
        run<16, false, false>(threads, out, cyc, 8); run<8, true, false>(threads, out, cyc, 8);
    }
    return 0;
}
