// Probe: how much of the power-limited fp16-MFMA rate do the A-fragment LDS reads cost?
// 256 workgroups x 8 waves, random fp16 data; per group of 3 MFMAs (the split-fp16 pattern) issue R ds_read_b128
// (R = 0: fragments stay in registers, 1, 2 = the layer kernel's pattern).  Long runs so DVFS settles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int R>
__global__ __launch_bounds__(512) void k(const f16x8* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[4096];      // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8 bh = in[(threadIdx.x * 3 + 1) & 4095], bl = in[(threadIdx.x * 5 + 2) & 4095];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 ah = lds[lane], al = lds[2048 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            f16x8 nh = ah, nl = al;
            if (R >= 1) nh = lds[((g + it) & 31) * 64 + lane];
            if (R >= 2) nl = lds[2048 + ((g + it) & 31) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[g & 3], 0, 0, 0);
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[g & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ah = nh;
            al = nl;
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R>
void run(const f16x8* in, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<R>), dim3(256), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfma = 256.0 * 8 * iters * 32 * 3;
    printf("R=%d ds_read_b128 per 3 MFMAs: %.2f ms, %.1f ns per MFMA per SIMD-pair, %.0f TFLOP/s\n", R, ms,
           ms * 1e6 / (iters * 32.0 * 3 * 2), mfma * 2.0 * 32 * 32 * 16 / ms / 1e9);
}

int main() {
    f16x8* in;
    float* out;
    hipMalloc(&in, 4096 * 16);
    hipMalloc(&out, 256 * 512 * 4);
    _Float16* h = (_Float16*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
    run<0>(in, out);
    run<1>(in, out);
    run<2>(in, out);
    run<0>(in, out);
    return 0;
}
