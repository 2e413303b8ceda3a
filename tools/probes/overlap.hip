// Probe: MFMA (f16 32x32x16) and VALU work on one SIMD -- separate waves vs the same wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ROLE per wave half: 0 idle/exit, 1 MFMA only, 2 VALU only (VPM fma per step), 3 both interleaved
template <int ROLE_LO, int ROLE_HI, int VPM>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(threadIdx.x + i) & 1023]; b[i] = (_Float16)in[(threadIdx.x * 3 + i) & 1023]; }
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = in[(threadIdx.x + r) & 1023];
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = in[(threadIdx.x + 5 * i) & 1023];
    auto body = [&](auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (ROLE & 1) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 1], 0, 0, 0);
                if constexpr (ROLE & 2) {
#pragma unroll
                    for (int j = 0; j < VPM; ++j) v[j % 12] = __builtin_fmaf(v[j % 12], 1.0001f, 0.5f);
                }
            }
        }
    };
    if (wave < 4) { if constexpr (ROLE_LO != 0) body(std::integral_constant<int, ROLE_LO>{}); }
    else { if constexpr (ROLE_HI != 0) body(std::integral_constant<int, ROLE_HI>{}); }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int RL, int RH, int VPM>
void run(const float* in, float* out, const char* name) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<RL, RH, VPM>), dim3(256), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-58s %.3f ms -> %.1f cycles per step @2.3GHz\n", name, ms, ms * 1e-3 * 2.3e9 / (iters * 8.0));
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 1 << 22);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (i % 37) * 0.01f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<1, 0, 6>(in, out, "A: waves0-3 MFMA only");
    run<0, 2, 6>(in, out, "B: waves4-7 6 VALU/step only");
    run<1, 2, 6>(in, out, "A+B: waves0-3 MFMA, waves4-7 6 VALU/step (2 waves/SIMD)");
    run<3, 0, 6>(in, out, "4 waves each MFMA + 6 VALU interleaved");
    run<3, 3, 6>(in, out, "8 waves each MFMA + 6 VALU interleaved");
    run<1, 1, 6>(in, out, "8 waves MFMA only");
    run<2, 2, 6>(in, out, "8 waves 6 VALU/step only");
    run<3, 0, 12>(in, out, "4 waves each MFMA + 12 VALU interleaved");
    run<1, 2, 12>(in, out, "waves0-3 MFMA, waves4-7 12 VALU/step");
    return 0;
}
