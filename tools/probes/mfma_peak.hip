// Probe: what does a pure v_mfma_f32_32x32x2_f32 loop reach on this box (clock / power)?
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const float* in, float* out, int iters, long long* cyc) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 512];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = in[(threadIdx.x + r + i) & 1023];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    float h[1024];
    for (int z = 0; z < 2; ++z) {
        for (int i = 0; i < 1024; ++i) h[i] = z ? (rand() / (float)RAND_MAX - 0.5f) * 0.02f : 0.f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        for (int waves = 4; waves <= 8; waves += 4) {
            const int iters = 4000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (waves == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, in, out, iters, cyc);
                else hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, in, out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            double flop = 256.0 * waves * iters * 32 * 2.0 * 32 * 32 * 2;
            printf("%s data, %d waves/WG: %.3f ms  %.1f TFLOP/s  kernel cycles(s_memtime) %lld -> %.3f GHz-equivalent, cyc/mfma/SIMD %.1f\n",
                   z ? "random" : "zero", waves, ms, flop / ms / 1e9, c, c / (ms * 1e6), (double)c / (iters * 32.0 * (waves / 4)));
        }
    }
    return 0;
}
