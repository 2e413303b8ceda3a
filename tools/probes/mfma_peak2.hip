// Probe: f32 MFMA rate vs accumulator rotation depth and LDS-fed A operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDSA>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float b[8];
    for (int i = 0; i < 8; ++i) b[i] = in[(threadIdx.x + i * 7) & 1023];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = in[(threadIdx.x + r + i) & 1023];
    f32x4 a0 = *(const f32x4*)&lds[lane * 4], a1 = *(const f32x4*)&lds[4096 + lane * 4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            f32x4 n0 = a0, n1 = a1;
            if (LDSA) {
                n0 = *(const f32x4*)&lds[((g + 1) & 15) * 256 + lane * 4];
                n1 = *(const f32x4*)&lds[8192 + ((g + 1) & 15) * 256 + lane * 4];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[(2 * e) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], acc[(2 * e) % NACC], 0, 0, 0);
                acc[(2 * e + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[e + 4], acc[(2 * e + 1) % NACC], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0; a1 = n1;
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDSA>
void run(const float* in, float* out, const char* name) {
    const int iters = 250;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, LDSA>), dim3(256), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 256.0 * 4 * iters * 128 * 2.0 * 32 * 32 * 2;
    printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1 << 22);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<4, false>(in, out, "4 acc, reg A");
    run<2, false>(in, out, "2 acc, reg A");
    run<1, false>(in, out, "1 acc, reg A");
    run<4, true>(in, out, "4 acc, LDS A + barriers");
    run<2, true>(in, out, "2 acc, LDS A + barriers");
    return 0;
}
