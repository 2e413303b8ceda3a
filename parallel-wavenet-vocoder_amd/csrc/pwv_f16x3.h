// Split-fp16 ("f16x3") MFMA building blocks shared by the per-layer kernels (pwv_layer_f16.hip) and the persistent
// stack kernel (pwv_stack_persist.hip): operand split, A-fragment addressing in LDS, and the k-step-pipelined GEMM.
#pragma once

#include "pwv_layer_common.h"

namespace pwv {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// registers x[OFF .. OFF+7] -> hi / lo fp16 fragments (one MFMA B operand each)
template <int OFF, int N>
__device__ __forceinline__ void split8(const float (&x)[N], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const f32x2 v = {x[OFF + q], x[OFF + q + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        // x - float(hi) with the fp16 halves read in place (round 6: v_fma_mix_f32, two instructions instead of two conversions and a packed subtract -- the
        // compiler folds fma(float(h), -1, x) back into the subtract, hence asm; same value: the difference is exact either way.  -60 VALU per 32-row unit:
        // C3 -0.9 %, C5 -0.9 %, 1 x 16000 -1.1 %, profiles/r06_ab_experiments.md r06_z17)
        // (The hi conversion above stays the compiler's: it is the FIRST reader of x, which may be an MFMA result, and the hazard recogniser does not see into
        //  asm.  The lo conversion stays the compiler's too: as asm it gave the gain back -- C3 -0.13 % instead of -0.86 %, r06_z19.)
        f32x2 r;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(h), "v"(v[0]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(h), "v"(v[1]));
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[q] = h[0];
        hi[q + 1] = h[1];
        lo[q] = l[0];
        lo[q + 1] = l[1];
    }
}

// A fragments live in LDS as 16-byte units: unit index ((comp*NITTOT + it)*NS + s)*64 + lane,
// comp 0 = hi, 1 = lo; each unit = the 8 k values of k-step s this lane multiplies.
template <int NS, int NITTOT>
__device__ __forceinline__ f16x8 frag16(const f16x8* A, int comp, int it, int s, int lane) {
    return A[((comp * NITTOT + it) * NS + s) * 64 + lane];
}

// One GEMM as NS groups (one k-step of 16 each) of NIT row tiles x 3 MFMAs.  Fragments of step
// s+1 are read while step s's MFMAs issue (sched_barrier-pinned, see pwv_layer.hip).
template <int NS, int NIT, int IT0, int ITSTEP, int NITTOT, int NACC, typename BH, typename BL, typename EF, typename TF>
__device__ __forceinline__ void gemm16(const f16x8* A, int lane, f32x16 (&acc)[NACC], f16x8 (&ah)[4], f16x8 (&al)[4],
                                       BH&& bh, BL&& bl, EF&& extra, TF&& tail) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f16x8 nh[4] = {ah[0], ah[1], ah[2], ah[3]};
        f16x8 nl[4] = {al[0], al[1], al[2], al[3]};
        if (s + 1 < NS) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                nh[i] = frag16<NS, NITTOT>(A, 0, IT0 + i * ITSTEP, s + 1, lane);
                nl[i] = frag16<NS, NITTOT>(A, 1, IT0 + i * ITSTEP, s + 1, lane);
            }
        } else {
            tail(nh, nl);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 b_h = bh(s);
        const f16x8 b_l = bl(s);
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            acc[IT0 + i * ITSTEP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b_h, acc[IT0 + i * ITSTEP], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            acc[IT0 + i * ITSTEP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b_l, acc[IT0 + i * ITSTEP], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            acc[IT0 + i * ITSTEP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b_h, acc[IT0 + i * ITSTEP], 0, 0, 0);
        extra(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = nh[i];
            al[i] = nl[i];
        }
    }
}

// The same over the k-steps [S0, S1) of a GEMM whose fragments are NS k-steps apart (round 6: the short-input instantiation of the persistent
// kernel runs the x[t] half of BOTH row-tile pairs before the x[t-d] half of either; per accumulator the order of the k-steps is unchanged)
template <int NS, int S0, int S1, int NIT, int IT0, int ITSTEP, int NITTOT, int NACC, typename BH, typename BL, typename EF, typename TF>
__device__ __forceinline__ void gemm16r(const f16x8* A, int lane, f32x16 (&acc)[NACC], f16x8 (&ah)[4], f16x8 (&al)[4],
                                        BH&& bh, BL&& bl, EF&& extra, TF&& tail) {
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        f16x8 nh[4] = {ah[0], ah[1], ah[2], ah[3]};
        f16x8 nl[4] = {al[0], al[1], al[2], al[3]};
        if (s + 1 < S1) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                nh[i] = frag16<NS, NITTOT>(A, 0, IT0 + i * ITSTEP, s + 1, lane);
                nl[i] = frag16<NS, NITTOT>(A, 1, IT0 + i * ITSTEP, s + 1, lane);
            }
        } else {
            tail(nh, nl);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 b_h = bh(s);
        const f16x8 b_l = bl(s);
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            acc[IT0 + i * ITSTEP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b_h, acc[IT0 + i * ITSTEP], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            acc[IT0 + i * ITSTEP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b_l, acc[IT0 + i * ITSTEP], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            acc[IT0 + i * ITSTEP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b_h, acc[IT0 + i * ITSTEP], 0, 0, 0);
        extra(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = nh[i];
            al[i] = nl[i];
        }
    }
}

template <int NS, int S, int NIT, int IT0, int ITSTEP, int NITTOT>
__device__ __forceinline__ void frags_at(const f16x8* A, int lane, f16x8 (&h)[4], f16x8 (&l)[4]) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        h[i] = frag16<NS, NITTOT>(A, 0, IT0 + i * ITSTEP, S, lane);
        l[i] = frag16<NS, NITTOT>(A, 1, IT0 + i * ITSTEP, S, lane);
    }
}

template <int NS, int NIT, int IT0, int ITSTEP, int NITTOT>
__device__ __forceinline__ void first_frags(const f16x8* A, int lane, f16x8 (&h)[4], f16x8 (&l)[4]) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        h[i] = frag16<NS, NITTOT>(A, 0, IT0 + i * ITSTEP, 0, lane);
        l[i] = frag16<NS, NITTOT>(A, 1, IT0 + i * ITSTEP, 0, lane);
    }
}

}  // namespace pwv
