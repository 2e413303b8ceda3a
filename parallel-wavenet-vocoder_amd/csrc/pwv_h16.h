// fp16-mode ("f16", PWV_PREC_F16) building blocks shared by the per-layer kernels (pwv_layer_h16.hip) and the persistent stack
// kernel (pwv_stack_persist.hip): the fp16 tile32 addressing, the fp32 -> fp16 fragment conversion, the k-step-pipelined GEMM on
// the `hi` halves of the split-fp16 packed weights, and the LDS map of one layer's weights.
#pragma once

#include "pwv_layer_common.h"

namespace pwv {

typedef float f32x2h __attribute__((ext_vector_type(2)));

// half offset of chunk (s, h) of flat row `row` in an fp16 tile32 buffer of C channels (C/8 chunks per row)
__device__ __forceinline__ size_t xoff(int row, int chunk, int C) {
    return (size_t)(row >> 5) * (32 * C) + (chunk * 32 + (row & 31)) * 8;
}

// 8 fp32 registers -> one fp16 fragment (round to nearest even)
template <int OFF, int N>
__device__ __forceinline__ f16x8 to_h8(const float (&x)[N]) {
    f16x8 r;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const f32x2h v = {x[OFF + q], x[OFF + q + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        r[q] = h[0];
        r[q + 1] = h[1];
    }
    return r;
}

// only the `hi` component of each split-fp16 section of a packed layer is staged into LDS (the first half of the section)

// GEMM over NS k-steps with NIT row tiles, fragments prefetched one k-step ahead
template <int NS, int NIT, int NACC, typename BF>
__device__ __forceinline__ void gemm_h(const f16x8* A, int lane, f32x16 (&acc)[NACC], BF&& bfrag) {
    f16x8 a[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) a[i] = A[(i * NS + 0) * 64 + lane];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f16x8 n[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) n[i] = a[i];
        if (s + 1 < NS) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) n[i] = A[(i * NS + s + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 b = bfrag(s);
#pragma unroll
        for (int i = 0; i < NIT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NIT; ++i) a[i] = n[i];
    }
}

// LDS map (16-byte units unless noted): A1 hi [4 it][8 s][64] | A2 hi [2][4][64] | AC hi [4][5][64] | BD 64 floats | counter
constexpr int kH_A1 = 0;
constexpr int kH_A2 = kH_A1 + 4 * 8 * 64;      // 2048
constexpr int kH_AC = kH_A2 + 2 * 4 * 64;      // 2560
constexpr int kH_END = kH_AC + 4 * 5 * 64;     // 3840 units = 61,440 B (with cond); 40,960 B without

}  // namespace pwv
