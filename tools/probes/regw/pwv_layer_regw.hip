// Register-stationary variant of the split-fp16 gated-residual layer (modules.py:185-259): ONE wave per SIMD (4-wave
// workgroups, 512 registers per lane), the layer's filter|gate matrix -- 64 KB as hi | lo fp16 fragments -- held in the
// wave's registers (256 AGPRs) for all of its units instead of being re-read from LDS for every unit (64 of the 80
// ds_read_b128 of a unit); the dense matrix (16 KB) stays in LDS.  Same per-accumulator operation order as
// layer_f16x3_kernel, so the results are bit-identical (tests/test_gpu_parity.py runs both).
//
// Round 6: the unit body of the layer-stationary design (csrc/pwv_stack_systolic.hip) priced as a per-layer launch.
#include "pwv_f16x3.h"

namespace pwv {

// all 12 MFMAs of k-step S (4 row tiles x {hi*hi, hi*lo, lo*hi}), per accumulator in layer_f16x3_kernel's order
#define PWV_REGW_KSTEP(S, BH, BL)                                                                         \
    do {                                                                                                  \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[it][S], BH, acc[it], 0, 0, 0); \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[it][S], BL, acc[it], 0, 0, 0); \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[it][S], BH, acc[it], 0, 0, 0); \
    } while (0)

__global__ __launch_bounds__(256) void layer_f16x3_regw_kernel(const LayerParams p) {
    constexpr int WAVES = 4;
    constexpr int kL2 = 0, kLB = kA2Size, kLC = kA2Size + kBDSize;      // LDS: dense (hi | lo), dense bias, unit counter, P slots
    constexpr int kLP = kLC + 4;                                          // [4 waves][2][256]: the P rows of a unit's first and last sample
    __shared__ __attribute__((aligned(16))) float lds[kLP + WAVES * 2 * 256];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int net = blockIdx.x % p.G;
    const int wg = blockIdx.x / p.G;
    const int nwg = gridDim.x / p.G;
    int* unit_counter = reinterpret_cast<int*>(&lds[kLC]);
    if (tid == 0) *unit_counter = WAVES;
    const f16x8* A2 = reinterpret_cast<const f16x8*>(&lds[kL2]);

    const int rows = p.N * p.T;
    const int units = (rows + 31) / 32;
    const int per_wg = (units + nwg - 1) / nwg;
    const int u_begin = wg * per_wg;
    const int u_end = (u_begin + per_wg < units) ? u_begin + per_wg : units;
    auto grab = [&]() -> int {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(unit_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return u_begin + __builtin_amdgcn_readfirstlane(v);
    };
    const __amdgpu_buffer_rsrc_t out_rs = units_rsrc(p.x_out[net], u_begin, u_end, 32 * 64 * 4);

    auto load_x = [&](int unit, float (&xb)[32], float (&xc)[32]) {
        int row, rc, n, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        const bool has_prev = t >= p.dilation;
        load_tiled<8, 64>(p.x_in[net], rc, h, true, xc);
        if (__all(has_prev)) {
            load_tiled<8, 64>(p.x_in[net], rc - p.dilation, h, true, xb);
        } else {
            load_tiled<8, 64>(p.x_in[net], has_prev ? rc - p.dilation : rc, h, has_prev, xb);
        }
    };

    // dense matrix + bias -> LDS; filter|gate fragments -> registers (packed order: [comp][it][s][lane] 16-byte units)
    fill_lds_dma<(kA2Size + kBDSize) / 4, WAVES>(lds, p.packed[net] + kA2, wave, lane);
    f16x8 wh[4][8], wl[4][8];
    {
        const f16x8* A1g = reinterpret_cast<const f16x8*>(p.packed[net] + kA1);
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                wh[it][s] = A1g[((0 * 4 + it) * 8 + s) * 64 + lane];
                wl[it][s] = A1g[((1 * 4 + it) * 8 + s) * 64 + lane];
            }
        // pin the fragments into the accumulator half of the register file (AGPR class: an MFMA reads its A operand from there
        // directly; left to itself the allocator keeps them in VGPRs and shuttles them through v_accvgpr_read per use)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                asm volatile("" : "+a"(wh[it][s]));
                asm volatile("" : "+a"(wl[it][s]));
            }
    }
    __syncthreads();

    // The P row (the accumulators' start value, modules.py:218-221 folded to frame rate) a unit ahead: with ONE wave per SIMD nobody
    // covers a load issued at the top of a unit.  A unit's 32 samples lie in at most two frames when the hop is >= 32 samples:
    // lanes 0..31 fetch the row of the unit's first sample, lanes 32..63 that of its last (16 bytes each = 2 x 512 B), the wave
    // parks them in its LDS slot at the end of the unit before, and the next unit starts from 16 broadcast ds_read_b128.
    auto prow_of = [&](int nn, int tt) -> int {
        return p.cond_hop > 0 ? nn * p.cond_frames + fast_div(tt + p.cond_offset, p.hop_magic, p.hop_shift) : 0;
    };
    auto p_fetch = [&](int unit) -> f32x4 {
        int row, rc, n, t;
        bool valid;
        unit_rows(unit < u_end ? unit : u_end - 1, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        const int pr = prow_of(n, t);
        const int pa = __builtin_amdgcn_readlane(pr, 0), pb = __builtin_amdgcn_readlane(pr, 31);
        return *reinterpret_cast<const f32x4*>(p.proj[net] + (size_t)(h ? pb : pa) * p.proj_row_stride + (lane & 31) * 4);
    };
    float* pslot = lds + kLP + wave * 512;
    int unit = u_begin + wave;
    float rxb[32], rxc[32];
    load_x(unit, rxb, rxc);
    *reinterpret_cast<f32x4*>(pslot + lane * 4) = p_fetch(unit);
    int pbuf = 0;
    while (unit < u_end) {
        const int next = grab();
        int row, rc, n, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        const f32x4 pnext = p_fetch(next);
        f32x16 acc[4];
        {
            const int pr = prow_of(n, t);
            const int pa = __builtin_amdgcn_readlane(pr, 0), pb = __builtin_amdgcn_readlane(pr, 31);
            if (__builtin_expect(__any(pr != pa && pr != pb), 0)) {      // (a hop below 32 samples: straight from global memory)
                const float* g = p.proj[net] + (size_t)pr * p.proj_row_stride + h * 64;
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(g + it * 16 + q * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                    }
            } else {
                const float* g = pslot + pbuf * 256 + (pr == pa ? 0 : 128) + h * 64;
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(g + it * 16 + q * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                    }
            }
        }
        // ---- GEMM1, k-step-major: the B operand of a k-step is split while the previous k-step's 12 MFMAs issue ----
        f16x8 bh, bl, nh, nl;
        split8<0>(rxb, bh, bl);
#define PWV_REGW_STEP(S, SRC, OFF)                                \
        split8<OFF>(SRC, nh, nl);                                 \
        __builtin_amdgcn_sched_barrier(0);                        \
        PWV_REGW_KSTEP(S, bh, bl);                                \
        __builtin_amdgcn_sched_barrier(0);                        \
        bh = nh; bl = nl;
        PWV_REGW_STEP(0, rxb, 8)
        PWV_REGW_STEP(1, rxb, 16)
        PWV_REGW_STEP(2, rxb, 24)
        PWV_REGW_STEP(3, rxc, 0)
        PWV_REGW_STEP(4, rxc, 8)
        PWV_REGW_STEP(5, rxc, 16)
        PWV_REGW_STEP(6, rxc, 24)
#undef PWV_REGW_STEP
        f16x8 ah[4], al[4];
        first_frags<4, 2, 0, 1, 2>(A2, lane, ah, al);
        // GEMM2's accumulator starts at x[t] + dense_bias
        f32x16 acc2[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bd = *reinterpret_cast<const f32x4*>(&lds[kLB + h * 32 + it * 16 + q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = rxc[it * 16 + q * 4 + e] + bd[e];
            }
        __builtin_amdgcn_sched_barrier(0);
        PWV_REGW_KSTEP(7, bh, bl);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]));
        load_x(next, rxb, rxc);      // the next unit's rows: in flight under the gating + GEMM2 + stores
        __builtin_amdgcn_sched_barrier(0);
        float o[32];
        f16x8 oh[4], ol[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = gate_act(acc[0][r], acc[2][r]);
        split8<0>(o, oh[0], ol[0]);
        split8<8>(o, oh[1], ol[1]);
        gemm16<4, 2, 0, 1, 2>(
            A2, lane, acc2, ah, al, [&](int s) -> f16x8 { return oh[s]; }, [&](int s) -> f16x8 { return ol[s]; },
            [&](int s) {
                if (s < 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[16 + 8 * s + e] = gate_act(acc[1][8 * s + e], acc[3][8 * s + e]);
                    if (s == 0) split8<16>(o, oh[2], ol[2]);
                    else split8<24>(o, oh[3], ol[3]);
                    asm volatile("" : "+v"(oh[2 + (s & 1)]), "+v"(ol[2 + (s & 1)]));
                }
            },
            [&](f16x8(&)[4], f16x8(&)[4]) {});
        const int ooff = units_off(row, h, 64, u_begin);
        if (valid) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int it = g >> 2, q = g & 3;
                f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
                store_wt(out_rs, ooff + g * 1024, v);
            }
        }
        pbuf ^= 1;
        *reinterpret_cast<f32x4*>(pslot + pbuf * 256 + lane * 4) = pnext;
        __builtin_amdgcn_sched_barrier(0);
        unit = next;
    }
}

int launch_layer_f16x3_regw(const LayerParams& lp, int per_net, hipStream_t s) {
    hipLaunchKernelGGL(layer_f16x3_regw_kernel, dim3(per_net * lp.G), dim3(256), 0, s, lp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "regw layer kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

}  // namespace pwv
