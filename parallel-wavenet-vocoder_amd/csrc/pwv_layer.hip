// Fused gated-residual WaveNet layer and post-processing head for gfx950 (MI355X).
//
// Replaces, per layer, the ~15 TensorFlow ops of WaveNet._create_dilation_layer
// (/root/reference/modules.py:185-259) and, per net, the post-processing of
// WaveNet.__call__ (modules.py:145-165).
//
// Design (measured history in HISTORY.md section 4):
//   * GEMM orientation: MFMA rows = output channels, MFMA columns = time.  A wave owns 32
//     consecutive samples; lane (t = lane&31, h = lane>>5) owns, of every 64-channel row, the
//     16-byte chunks at float offsets 8g + 4h -- which is exactly the v_mfma 32x32 C/D layout
//     (pwv::chan_of).  Hence x[t], x[t-d] are loaded straight from HBM into the B operand,
//     the gated output o stays in the registers it was accumulated in and IS the B operand of
//     the dense / skip GEMMs, and the residual add is a register add.  No LDS round trip and
//     no transposes for activations.
//   * Weights are the A operand, pre-packed (pwv_pack_*) so each lane reads 4 consecutive
//     k-steps with one conflict-free ds_read_b128; a workgroup loads one layer's weights
//     (80-152 KB of the 160 KB LDS) once and walks many time tiles (persistent over tiles).
//   * The conditioning term and the filter/gate biases enter as the accumulator's initial
//     value (frame-rate projection P), so the hoisted 'repeat' conditioning costs no FLOPs
//     in this kernel; per-sample conditioning (transposed-conv upsampling) adds 40 k-steps.
//   * fp32 arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 fma chains, 157 TFLOP/s peak).
#include "pwv_layer_common.h"

#include <cstdlib>

namespace pwv {

template <bool SKIP, bool COND>
struct TileRegs {
    float xb[32];                 // x[t-d], this lane's 32 channels
    float xc[32];                 // x[t]
    float pj[64];                 // P[frame(t)] : accumulator init for the 4 F/G row tiles
    float cd[COND ? 40 : 4];      // cond[t], this lane's 40 channels
    float sk[SKIP ? 64 : 4];      // running skip sum (D layout)
    int row;                      // flattened row n*T + t of this lane
    int t;                        // position inside the utterance
    bool valid;
};

// the skip-sum row [128] of a unit: chunk for (it, q) = channel quad 8*it + 2*q + h
template <bool SKIP, bool COND>
__device__ __forceinline__ void load_skip_row(const LayerParams& p, int net, int lane, TileRegs<SKIP, COND>& r, bool skip_load) {
    if constexpr (SKIP) {
        const int h = lane >> 5;
        const int rows = p.N * p.T;
        const int rc = r.valid ? r.row : rows - 1;
        if (skip_load) {
            const float* srow = p.skip[net] + tile_off(rc, h, 128);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(srow + (8 * it + 2 * q) * 128);
#pragma unroll
                    for (int e = 0; e < 4; ++e) r.sk[it * 16 + q * 4 + e] = v[e];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 64; ++i) r.sk[i] = 0.f;
        }
    }
}

// SK_NOW = false: the skip-sum row is NOT loaded with the tile (8-wave kernels request it in front of GEMM2 / the gating instead:
// its 64 registers next to x[t-d], x[t], P and the per-sample condition at the top of the unit were the 27-35 spilled VGPRs of the
// SKIP + COND variants)
template <bool SKIP, bool COND, bool FIRST = false, bool SK_NOW = true>
__device__ __forceinline__ void load_tile(const LayerParams& p, int net, int unit, int lane,
                                          TileRegs<SKIP, COND>& r, bool skip_load) {
    const int h = lane >> 5;
    const int rows = p.N * p.T;
    const int row = unit * 32 + (lane & 31);
    r.row = row;
    r.valid = row < rows;

    const int rc = r.valid ? row : rows - 1;          // clamped: loads never leave the tensor
    const int n = rc / p.T;
    const int t = rc - n * p.T;
    const bool has_prev = t >= p.dilation;            // x[t-d] = 0 left of the utterance start
    r.t = t;
    if constexpr (FIRST) {
        // the four scalars the two rows are functions of: x[t], x[t-1], x[t-d], x[t-d-1] (zero left of the start);
        // rebuild_first() turns them into the rows
        const float* x1 = p.x_first;
        const int d = p.dilation;
        r.xc[0] = x1[rc];
        r.xc[1] = t >= 1 ? x1[rc - (t >= 1 ? 1 : 0)] : 0.f;
        r.xb[0] = has_prev ? x1[rc - (has_prev ? d : 0)] : 0.f;
        r.xb[1] = t >= d + 1 ? x1[rc - (t >= d + 1 ? d + 1 : 0)] : 0.f;
    } else {
        load_tiled<8, 64>(p.x_in[net], rc, h, true, r.xc);
        load_tiled<8, 64>(p.x_in[net], has_prev ? rc - p.dilation : rc, h, has_prev, r.xb);
    }
    int prow = 0;
    if (p.cond_hop > 0) prow = n * p.cond_frames + fast_div(t + p.cond_offset, p.hop_magic, p.hop_shift);
    load_contig<16>(p.proj[net] + (size_t)prow * p.proj_row_stride + h * 64, r.pj);
    if constexpr (COND) load_tiled<10, kCondC>(p.cond, rc, h, true, r.cd);
    if constexpr (SKIP && SK_NOW) load_skip_row<SKIP, COND>(p, net, lane, r, skip_load);
}

// Work is handed out in 32-row units (one MFMA column tile = one wave's worth of samples).
// WAVES = 4: one wave per SIMD with up to 512 registers; units are strided statically and the next
//            unit's operands are prefetched into registers under the current unit's GEMM2.
// WAVES = 8: two waves per SIMD (<= 256 registers each).  Waves pull units from a per-workgroup
//            LDS counter and waves 4..7 (the second wave of each SIMD) run at raised priority, so
//            the two waves of a SIMD drift out of phase: one wave's MFMAs cover the other's loads,
//            gating, address arithmetic and stores (identical streams at equal priority stay in
//            lockstep and hit their VALU sections together, leaving the matrix pipe idle).
// FIRST (layer 0 of a scalar-input net, 8 waves): the causal layer's output is not read but rebuilt from the flow input
//   with the front kernel's own two fp32 operations per channel (bit-identical to pwv_iaf_front_f32 + this kernel).
// HEAD  (plain gated last layer, 8 waves): head_f32_kernel<true>'s arithmetic runs behind the gate on the registers the
//   gated output was accumulated in; LDS holds exactly filter|gate + skip + postprocess1 = 160 KB, so the small vectors
//   come from global memory and units are handed out statically (no room for the counter).
// FOLD (with FIRST): layer 0's filter|gate convolution on the four scalars x[t-d-1], x[t-d], x[t-1], x[t] themselves
//   (pwv_pack_first_fold_f32; see layer_f16x3_kernel): two fp32 MFMA k-steps per row tile instead of 64.
template <int WAVES, bool SKIP, bool COND, bool GATED, bool FIRST = false, bool HEAD = false, bool FOLD = false>
__global__ __launch_bounds__(64 * WAVES) void layer_f32_kernel(const LayerParams p) {
    static_assert(!FOLD || FIRST, "FOLD: layer 0 of a scalar-input net only");
    static_assert(!(FIRST || HEAD) || WAVES == 8, "FIRST / HEAD: 8-wave kernels only");
    static_assert(!HEAD || (GATED && !SKIP && !COND && !FIRST), "HEAD: plain last layer only");
    static_assert(!FIRST || !SKIP, "FIRST: no skip accumulation");
    constexpr bool PREFETCH = WAVES == 4;
    constexpr int kLds = HEAD ? kA1Size + kASSize + kHA1Size : layer_floats(SKIP, COND);
    constexpr int kCF = kLds + 4;           // FIRST: the causal filter [2][64] behind the unit counter
    constexpr int kHS = kA1Size;            // HEAD: skip weights, then postprocess1
    constexpr int kH1 = kA1Size + kASSize;
    __shared__ __attribute__((aligned(16))) float lds[kLds + (HEAD ? 0 : 4) + (FIRST ? 128 : 0)];   // +4: the unit counter

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int net = blockIdx.x % p.G;
    const int wg = blockIdx.x / p.G;
    const int nwg = gridDim.x / p.G;
    int* unit_counter = reinterpret_cast<int*>(&lds[HEAD ? 0 : kLds]);      // unused with HEAD

    if constexpr (HEAD) {
        fill_lds_dma<kA1Size / 4, WAVES>(lds, p.packed[net] + kA1, wave, lane);
        fill_lds_dma<kASSize / 4, WAVES>(lds + kHS, p.packed_head[net] + kHAS, wave, lane);
        fill_lds_dma<kHA1Size / 4, WAVES>(lds + kH1, p.packed_head[net] + kHA1, wave, lane);
    } else {
        fill_lds_dma<kLds / 4, WAVES>(lds, p.packed[net], wave, lane);
        if (tid == 0) *unit_counter = 0;
        if constexpr (FIRST) {
            if (tid < 128) lds[kCF + tid] = p.cfilt[net][tid];
        }
    }
    __syncthreads();

    constexpr int kAS = kLayerBase;
    constexpr int kBS = kAS + kASSize;
    constexpr int kAC = kLayerBase + (SKIP ? kASSize + kBSSize : 0);
    constexpr int NC8 = kCondC / 8;

    const int rows = p.N * p.T;
    const int units = (rows + 31) / 32;
    const bool skip_load = SKIP && !p.skip_init;

    // this workgroup's contiguous share of the units
    const int per_wg = (units + nwg - 1) / nwg;
    const int u_begin = wg * per_wg;
    const int u_end = (u_begin + per_wg < units) ? u_begin + per_wg : units;

    auto grab = [&]() -> int {   // next unit for this wave (WAVES == 8)
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(unit_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return u_begin + __builtin_amdgcn_readfirstlane(v);
    };

    int unit;
    TileRegs<SKIP, COND> cur;
    if constexpr (PREFETCH) {
        unit = u_begin + wave;
        load_tile<SKIP, COND>(p, net, unit, lane, cur, skip_load);
    } else {
        if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
        unit = HEAD ? u_begin + wave : grab();
    }

    auto no_extra = [](int) {};
    const __amdgpu_buffer_rsrc_t out_rs = units_rsrc(p.x_out[net], u_begin, u_end, 32 * 64 * 4);      // unused with HEAD

    while (unit < u_end) {
        const int next = PREFETCH ? unit + WAVES : 0;
        if constexpr (!PREFETCH) load_tile<SKIP, COND, FIRST, false>(p, net, unit, lane, cur, skip_load);
        float first_x0 = 0.f, first_x1 = 0.f;      // FIRST: x[t], x[t-1]
        float fold_b0 = 0.f, fold_b1 = 0.f;        // FOLD: the B values of the two k-steps
        (void)first_x0;
        (void)first_x1;
        (void)fold_b0;
        (void)fold_b1;
        if constexpr (FIRST) {
            // this lane's 32 channels (8g + 4h + e) of h[t] and h[t-d] from the scalars; same operation order as
            // iaf_front_kernel: round(x[t-1] w0), then fma(x[t], w1, .)
            const float x0 = cur.xc[0], x1v = cur.xc[1], xd0 = cur.xb[0], xd1 = cur.xb[1];
            first_x0 = x0;
            first_x1 = x1v;
            const bool has_prev = cur.t >= p.dilation;
            if constexpr (FOLD) {      // (x[t-d], x[t-d-1] are already zero left of the start)
                fold_b0 = h ? xd0 : xd1;      // k = 0, 1
                fold_b1 = h ? x0 : x1v;       // k = 2, 3
            }
#pragma unroll
            for (int g = 0; g < (FOLD ? 0 : 8); ++g) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(&lds[kCF + 8 * g + 4 * h]);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(&lds[kCF + 64 + 8 * g + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cur.xc[4 * g + e] = fmaf(x0, w1[e], x1v * w0[e]);
                    const float vb = fmaf(xd0, w1[e], xd1 * w0[e]);
                    cur.xb[4 * g + e] = has_prev ? vb : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);      // one filter quad at a time: the scheduler otherwise front-loads all 16 reads
            }
        }
        // ---- GEMM1: [F;G][128 x 32t] = W1^T[128 x K] * [x[t-d]; x[t]; (cond[t])] -------------
        // accumulators start at P[frame(t)] (conditioning projection + filter/gate bias)
        f32x16 acc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[it][r] = cur.pj[it * 16 + r];
        float o[32];
        f32x4 a[4];
        auto bx = [&](int ks) -> float { return ks < 32 ? cur.xb[ks] : cur.xc[ks - 32]; };
        auto bc = [&](int ks) -> float { return cur.cd[COND ? ks : 0]; };
        // pair 0 = row tiles (0: F[0:32], 2: G[0:32]); pair 1 = (1: F[32:64], 3: G[32:64]).
        // pair 0 is gated on the VALU while pair 1's MFMAs run.
        if constexpr (FOLD) {
            // ---- layer 0, folded: the per-sample condition's GEMM (if any), then two k-steps on the scalars ------------------
            if constexpr (COND) {
                a[0] = frag(lds, kAC, 0, NC8, 0, lane);
                a[1] = frag(lds, kAC, 2, NC8, 0, lane);
                gemm_groups<NC8, 2, 0, 2>(lds, kAC, lane, acc, a, bc, no_extra, [&](f32x4(&n)[4]) {
                    n[0] = frag(lds, kAC, 1, NC8, 0, lane);
                    n[1] = frag(lds, kAC, 3, NC8, 0, lane);
                });
                gemm_groups<NC8, 2, 1, 2>(lds, kAC, lane, acc, a, bc, no_extra, [](f32x4(&)[4]) {});
            }
            const f32x4* F0 = reinterpret_cast<const f32x4*>(p.fold0[net]);
            f32x4 ff[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) ff[it] = F0[it * 64 + lane];
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ff[it][0], fold_b0, acc[it], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ff[it][1], fold_b1, acc[it], 0, 0, 0);
            if constexpr (!GATED) {
                a[0] = frag(lds, kA2, 0, 8, 0, lane);
                a[1] = frag(lds, kA2, 1, 8, 0, lane);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) o[g] = gate_act(acc[0][g], acc[2][g]);
        } else {
        if constexpr (COND) {
            a[0] = frag(lds, kAC, 0, NC8, 0, lane);
            a[1] = frag(lds, kAC, 2, NC8, 0, lane);
            gemm_groups<NC8, 2, 0, 2>(lds, kAC, lane, acc, a, bc, no_extra, [&](f32x4(&n)[4]) {
                n[0] = frag(lds, kA1, 0, 16, 0, lane);
                n[1] = frag(lds, kA1, 2, 16, 0, lane);
            });
        } else {
            a[0] = frag(lds, kA1, 0, 16, 0, lane);
            a[1] = frag(lds, kA1, 2, 16, 0, lane);
        }
        gemm_groups<16, 2, 0, 2>(lds, kA1, lane, acc, a, bx, no_extra, [&](f32x4(&n)[4]) {
            if constexpr (COND) {
                n[0] = frag(lds, kAC, 1, NC8, 0, lane);
                n[1] = frag(lds, kAC, 3, NC8, 0, lane);
            } else {
                n[0] = frag(lds, kA1, 1, 16, 0, lane);
                n[1] = frag(lds, kA1, 3, 16, 0, lane);
            }
        });
        if constexpr (COND) {
            gemm_groups<NC8, 2, 1, 2>(lds, kAC, lane, acc, a, bc, no_extra, [&](f32x4(&n)[4]) {
                n[0] = frag(lds, kA1, 1, 16, 0, lane);
                n[1] = frag(lds, kA1, 3, 16, 0, lane);
            });
        }
        gemm_groups<16, 2, 1, 2>(
            lds, kA1, lane, acc, a, bx,
            [&](int g) {
                o[g] = gate_act(acc[0][g], acc[2][g]);
                asm volatile("" : "+v"(o[g]));   // keep the gating inside this MFMA group (no sinking)
            },
            [&](f32x4(&n)[4]) {
                if constexpr (!GATED) {
                    n[0] = frag(lds, kA2, 0, 8, 0, lane);
                    n[1] = frag(lds, kA2, 1, 8, 0, lane);
                } else if constexpr (SKIP) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) n[i] = frag(lds, kAS, i, 8, 0, lane);
                } else if constexpr (HEAD) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) n[i] = frag(lds, kHS, i, 8, 0, lane);
                }
            });
        }

        // next tile's operands: in flight during GEMM2 / skip GEMM / stores
        TileRegs<SKIP, COND> nx;   // rows past the end are clamped inside load_tile
        if constexpr (PREFETCH) {
            load_tile<SKIP, COND>(p, net, next, lane, nx, skip_load);
            __builtin_amdgcn_sched_barrier(0);
        }

        const int ooff = units_off(cur.row, h, 64, u_begin);      // (rows past the end are never stored: `valid`)
        if constexpr (GATED && HEAD) {
            // ---- fused head: o (registers) -> skip -> relu -> postprocess1 -> relu -> postprocess2, the operations
            //      (and bits) of head_f32_kernel<true> ------------------------------------------------------------
            const float* hb = p.packed_head[net];
            f32x16 accs[4];
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(hb + kHBS + h * 64 + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accs[it][q * 4 + e] = v[e];
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) o[16 + r] = gate_act(acc[1][r], acc[3][r]);
            gemm_groups<8, 4, 0, 1>(lds, kHS, lane, accs, a, [&](int ks) -> float { return o[ks]; }, no_extra,
                                    [&](f32x4(&n)[4]) {
#pragma unroll
                                        for (int i = 0; i < 4; ++i) n[i] = frag(lds, kH1, i, 16, 0, lane);
                                    });
            f32x16 acc1[4];
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(hb + kHB1 + h * 64 + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc1[it][q * 4 + e] = v[e];
                }
            gemm_groups<16, 4, 0, 1>(
                lds, kH1, lane, acc1, a, [&](int ks) -> float { return fmaxf(accs[ks >> 4][ks & 15], 0.f); }, no_extra,
                [](f32x4(&)[4]) {});
            const int Q = p.head_q;
            for (int q = 0; q < Q; ++q) {
                float part = 0.f;
                const float* w2 = hb + kHW2 + (h * Q + q) * 64;
#pragma unroll
                for (int i4 = 0; i4 < 16; ++i4) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(w2 + 4 * i4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * i4 + e;
                        part = fmaf(fmaxf(acc1[i >> 4][i & 15], 0.f), w[e], part);
                    }
                }
                part += __shfl_xor(part, 32);
                part += hb[kHW2 + 2 * Q * 64 + q];
                if (cur.valid && h == 0) p.head_out[net][(size_t)cur.row * Q + q] = part;
            }
        } else if constexpr (GATED) {
            if constexpr (SKIP && !PREFETCH) load_skip_row<SKIP, COND>(p, net, lane, cur, skip_load);      // in flight under the gating
#pragma unroll
            for (int r = 0; r < 16; ++r) o[16 + r] = gate_act(acc[1][r], acc[3][r]);
            if (cur.valid) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                    store_wt(out_rs, ooff + g * 1024, v);
                }
            }
        } else {
            // ---- GEMM2: dense 64 -> 64, accumulator starts at x[t] + dense_bias ---------------
            f32x16 acc2[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bd = *reinterpret_cast<const f32x4*>(&lds[kBD + h * 32 + it * 16 + q * 4]);
                    if constexpr (FIRST) {
                        // x[t] row evaluated again from the two scalars (same operations, same bits) rather than kept live
                        const int g = it * 4 + q;
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(&lds[kCF + 8 * g + 4 * h]);
                        const f32x4 w1 = *reinterpret_cast<const f32x4*>(&lds[kCF + 64 + 8 * g + 4 * h]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = fmaf(first_x0, w1[e], first_x1 * w0[e]) + bd[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = cur.xc[it * 16 + q * 4 + e] + bd[e];
                    }
                }
            }
            if constexpr (SKIP && !PREFETCH) load_skip_row<SKIP, COND>(p, net, lane, cur, skip_load);      // in flight under GEMM2
            // k-steps 0..15 use o tile 0 (ready); pair 1 is gated under those MFMAs
            gemm_groups<8, 2, 0, 1>(
                lds, kA2, lane, acc2, a, [&](int ks) -> float { return o[ks]; },
                [&](int g) {
                    if (g < 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[16 + 4 * g + e] = gate_act(acc[1][4 * g + e], acc[3][4 * g + e]);
                            asm volatile("" : "+v"(o[16 + 4 * g + e]));
                        }
                    }
                },
                [&](f32x4(&n)[4]) {
                    if constexpr (SKIP) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) n[i] = frag(lds, kAS, i, 8, 0, lane);
                    }
                });
            if (cur.valid) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int it = g >> 2, q = g & 3;
                    f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
                    store_wt(out_rs, ooff + g * 1024, v);
                }
            }
        }

        if constexpr (SKIP) {
            // ---- skip 64 -> 128, accumulated across layers (modules.py:243-250, :147) ----------
            f32x16 accs[4];
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(&lds[kBS + h * 64 + it * 16 + q * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accs[it][q * 4 + e] = cur.sk[it * 16 + q * 4 + e] + bs[e];
                }
            gemm_groups<8, 4, 0, 1>(lds, kAS, lane, accs, a, [&](int ks) -> float { return o[ks]; }, no_extra,
                                    [](f32x4(&)[4]) {});
            if (cur.valid) {
                const __amdgpu_buffer_rsrc_t skip_rs = units_rsrc(p.skip[net], u_begin, u_end, 32 * 128 * 4);
                const int soff = units_off(cur.row, h, 128, u_begin);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {accs[it][q * 4], accs[it][q * 4 + 1], accs[it][q * 4 + 2], accs[it][q * 4 + 3]};
                        store_wt(skip_rs, soff + (8 * it + 2 * q) * 512, v);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PREFETCH) {
            cur = nx;
            unit = next;
        } else {
            unit = HEAD ? unit + WAVES : grab();
        }
    }
}

// --------------------------------------------------------------------------------------
// Head: (o @ skip + b | skip_sum) -> relu -> post1 -> relu -> post2     modules.py:145-165
// --------------------------------------------------------------------------------------
template <bool FROM_GATED>
__global__ __launch_bounds__(256) void head_f32_kernel(const HeadParams p) {
    constexpr int kLds = head_floats(kMaxQ);
    __shared__ __attribute__((aligned(16))) float lds[kLds];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int net = blockIdx.x % p.G;
    const int wg = blockIdx.x / p.G;
    const int nwg = gridDim.x / p.G;
    const int Q = p.Q;
    fill_lds<head_floats(kMaxQ) / 4, 256>(lds, p.packed[net], tid);   // buffers are sized for kMaxQ
    __syncthreads();

    auto no_extra = [](int) {};
    const int rows = p.N * p.T;
    const int ntiles = (rows + 127) / 128;
    for (int tile = wg; tile < ntiles; tile += nwg) {
        const int row = tile * 128 + wave * 32 + (lane & 31);
        const bool valid = row < rows;
        f32x16 accs[4];
        f32x4 a[4];
        if constexpr (FROM_GATED) {
            float o[32];
            load_tiled<8, 64>(p.in[net], valid ? row : rows - 1, h, true, o);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(&lds[kHBS + h * 64 + it * 16 + q * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accs[it][q * 4 + e] = bs[e];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag(lds, kHAS, i, 8, 0, lane);
            gemm_groups<8, 4, 0, 1>(lds, kHAS, lane, accs, a, [&](int ks) -> float { return o[ks]; }, no_extra,
                                    [&](f32x4(&n)[4]) {
#pragma unroll
                                        for (int i = 0; i < 4; ++i) n[i] = frag(lds, kHA1, i, 16, 0, lane);
                                    });
        } else {
            const float* srow = p.in[net] + tile_off(valid ? row : rows - 1, h, 128);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(srow + (8 * it + 2 * q) * 128);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accs[it][q * 4 + e] = v[e];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag(lds, kHA1, i, 16, 0, lane);
        }
        // relu -> post1 (128 -> 128)
        f32x16 acc1[4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(&lds[kHB1 + h * 64 + it * 16 + q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[it][q * 4 + e] = b1[e];
            }
        gemm_groups<16, 4, 0, 1>(
            lds, kHA1, lane, acc1, a, [&](int ks) -> float { return fmaxf(accs[ks >> 4][ks & 15], 0.f); }, no_extra,
            [](f32x4(&)[4]) {});
        // relu -> post2 (128 -> Q): per-lane partial dot over its 64 channels, then add the halves
        for (int q = 0; q < Q; ++q) {
            float part = 0.f;
            const float* w2 = &lds[kHW2 + (h * Q + q) * 64];
#pragma unroll
            for (int i4 = 0; i4 < 16; ++i4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(w2 + 4 * i4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * i4 + e;
                    part = fmaf(fmaxf(acc1[i >> 4][i & 15], 0.f), w[e], part);
                }
            }
            part += __shfl_xor(part, 32);
            part += lds[kHW2 + 2 * Q * 64 + q];
            if (valid && h == 0) p.out[net][(size_t)row * Q + q] = part;
        }
    }
}

// --------------------------------------------------------------------------------------
// weight packing (device-side gathers from TensorFlow layouts)
// --------------------------------------------------------------------------------------
// pwv_pack_first_fold_f32 (see pack_first_fold_f16_kernel): [4 row tiles][64 lanes] f32x4 = {M[oc][h], M[oc][2 + h], 0, 0},
// the A values of the two fp32 MFMA k-steps (k = h and k = 2 + h) of lane (i, h), oc = 32 it + i
__global__ void pack_first_fold_f32_kernel(const float* __restrict__ cf, const float* __restrict__ filter, const float* __restrict__ gate,
                                           f32x4* __restrict__ out) {
    const int u = threadIdx.x;      // (it, lane)
    const int it = u >> 6, lane = u & 63, h = lane >> 5, oc = 32 * it + (lane & 31);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < 2; ++e) {
        const int k = 2 * e + h, tap = k >> 1, c = k & 1;
        double acc = 0.0;
        for (int cin = 0; cin < 64; ++cin) {
            const float wv = oc < 64 ? filter[(tap * 64 + cin) * 64 + oc] : gate[(tap * 64 + cin) * 64 + oc - 64];
            acc += (double)cf[c * 64 + cin] * (double)wv;
        }
        v[e] = (float)((double)(oc < 64 ? kFScale : kGScale) * acc);
    }
    out[u] = v;
}

__global__ void pack_layer_kernel(const float* filter, const float* gate, const float* dense,
                                  const float* dense_bias, const float* skip, const float* skip_bias,
                                  const float* gc_filter, const float* gc_gate, int with_skip, int cond_c,
                                  float* out, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    float v = 0.f;
    int i = idx;
    if (i < kA1Size) {
        const int e = i & 3, lane = (i >> 2) & 63, ks4 = (i >> 8) & 15, it = i >> 12;
        const int ks = ks4 * 4 + e, tap = ks >> 5, r = ks & 31, h = lane >> 5;
        const int cin = 8 * (r >> 2) + 4 * h + (r & 3);
        const int oc = 32 * it + (lane & 31);
        v = oc < 64 ? kFScale * filter[(tap * 64 + cin) * 64 + oc] : kGScale * gate[(tap * 64 + cin) * 64 + oc - 64];
    } else if ((i -= kA1Size) < kA2Size) {
        const int e = i & 3, lane = (i >> 2) & 63, ks4 = (i >> 8) & 7, it = i >> 11;
        const int ks = ks4 * 4 + e, h = lane >> 5;
        const int c = chan_of(ks >> 4, ks & 15, h);
        v = dense[c * 64 + 32 * it + (lane & 31)];
    } else if ((i -= kA2Size) < kBDSize) {
        const int h = i >> 5, it = (i >> 4) & 1, r = i & 15;
        v = dense_bias ? dense_bias[chan_of(it, r, h)] : 0.f;
    } else {
        i -= kBDSize;
        bool done = false;
        if (with_skip) {
            if (i < kASSize) {
                const int e = i & 3, lane = (i >> 2) & 63, ks4 = (i >> 8) & 7, it = i >> 11;
                const int ks = ks4 * 4 + e, h = lane >> 5;
                const int c = chan_of(ks >> 4, ks & 15, h);
                v = skip[c * 128 + 32 * it + (lane & 31)];
                done = true;
            } else if ((i -= kASSize) < kBSSize) {
                const int h = i >> 6, it = (i >> 4) & 3, r = i & 15;
                v = skip_bias ? skip_bias[chan_of(it, r, h)] : 0.f;
                done = true;
            } else {
                i -= kBSSize;
            }
        }
        if (!done && cond_c > 0) {
            const int nc8 = cond_c / 8;
            const int e = i & 3, lane = (i >> 2) & 63;
            const int rest = i >> 8;  // it * nc8 + ks4
            const int ks4 = rest % nc8, it = rest / nc8;
            const int r = ks4 * 4 + e, h = lane >> 5;
            const int ci = 8 * (r >> 2) + 4 * h + (r & 3);
            const int oc = 32 * it + (lane & 31);
            v = oc < 64 ? kFScale * gc_filter[ci * 64 + oc] : kGScale * gc_gate[ci * 64 + oc - 64];
        }
    }
    out[idx] = v;
}

__global__ void pack_head_kernel(const float* skip, const float* skip_bias, const float* post1,
                                 const float* post1_bias, const float* post2, const float* post2_bias, int Q,
                                 float* out, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    float v = 0.f;
    int i = idx;
    if (i < kASSize) {
        const int e = i & 3, lane = (i >> 2) & 63, ks4 = (i >> 8) & 7, it = i >> 11;
        const int ks = ks4 * 4 + e, h = lane >> 5;
        v = skip ? skip[chan_of(ks >> 4, ks & 15, h) * 128 + 32 * it + (lane & 31)] : 0.f;
    } else if ((i -= kASSize) < kBSSize) {
        const int h = i >> 6, it = (i >> 4) & 3, r = i & 15;
        v = skip_bias ? skip_bias[chan_of(it, r, h)] : 0.f;
    } else if ((i -= kBSSize) < kHA1Size) {
        const int e = i & 3, lane = (i >> 2) & 63, ks4 = (i >> 8) & 15, it = i >> 12;
        const int ks = ks4 * 4 + e, h = lane >> 5;
        v = post1[chan_of(ks >> 4, ks & 15, h) * 128 + 32 * it + (lane & 31)];
    } else if ((i -= kHA1Size) < 128) {
        const int h = i >> 6, it = (i >> 4) & 3, r = i & 15;
        v = post1_bias ? post1_bias[chan_of(it, r, h)] : 0.f;
    } else if ((i -= 128) < 2 * Q * 64) {
        const int j = i & 63, hq = i >> 6;
        const int q = hq % Q, h = hq / Q;
        v = post2[chan_of(j >> 4, j & 15, h) * Q + q];
    } else {
        i -= 2 * Q * 64;
        v = (i < Q && post2_bias) ? post2_bias[i] : 0.f;
    }
    out[idx] = v;
}

// waves per workgroup of the layer kernel: 8 (two per SIMD) unless PWV_LAYER_WAVES=4
static int layer_waves() {
    static int w = 0;
    if (w == 0) {
        const char* e = getenv("PWV_LAYER_WAVES");
        w = (e && atoi(e) == 4) ? 4 : 8;
    }
    return w;
}

template <bool SKIP, bool COND, bool GATED>
static int launch_layer(const LayerParams& lp, int per_net4, int per_net8, hipStream_t s) {
    if (layer_waves() == 4)
        hipLaunchKernelGGL((layer_f32_kernel<4, SKIP, COND, GATED>), dim3(per_net4 * lp.G), dim3(256), 0, s, lp);
    else
        hipLaunchKernelGGL((layer_f32_kernel<8, SKIP, COND, GATED>), dim3(per_net8 * lp.G), dim3(512), 0, s, lp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "layer kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

// the two buffer-removing variants (always 8 waves)
template <bool COND, bool GATED, bool FIRST, bool HEAD, bool FOLD = false>
static int launch_layer_fused(const LayerParams& lp, int per_net8, hipStream_t s) {
    hipLaunchKernelGGL((layer_f32_kernel<8, false, COND, GATED, FIRST, HEAD, FOLD>), dim3(per_net8 * lp.G), dim3(512), 0, s, lp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "layer kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

}  // namespace pwv

using namespace pwv;

extern "C" {

size_t pwv_layer_packed_floats(int with_skip, int cond_channels) {
    return (size_t)layer_floats(with_skip != 0, cond_channels > 0);
}

size_t pwv_head_packed_floats(int Q) {
    (void)Q;   // one size for every Q <= kMaxQ: the kernels copy a fixed-size block into LDS
    return (size_t)head_floats(kMaxQ);
}

int pwv_proj_column_map(int* map128) {
    if (!map128) return set_error(PWV_EINVAL, "map128 is NULL");
    for (int h = 0; h < 2; ++h)
        for (int it = 0; it < 4; ++it)
            for (int r = 0; r < 16; ++r) map128[h * 64 + it * 16 + r] = chan_of(it, r, h);
    return PWV_OK;
}

int pwv_pack_layer_f32(const float* filter, const float* gate, const float* dense, const float* dense_bias,
                       const float* skip, const float* skip_bias, const float* gc_filter, const float* gc_gate,
                       int with_skip, int cond_channels, int precision, float* packed, pwv_stream_t stream) {
    PWV_CHECK_ARG(filter && gate && dense && packed, "pwv_pack_layer_f32: NULL weight pointer");
    PWV_CHECK_ARG(!with_skip || skip, "pwv_pack_layer_f32: with_skip needs skip weights");
    PWV_CHECK_ARG(cond_channels == 0 || cond_channels == kCondC,
                  "pwv_pack_layer_f32: per-sample conditioning supports %d channels, got %d", kCondC, cond_channels);
    PWV_CHECK_ARG(cond_channels == 0 || (gc_filter && gc_gate), "pwv_pack_layer_f32: gc weights missing");
    PWV_CHECK_ARG(precision >= PWV_PREC_F32 && precision <= PWV_PREC_F16, "pwv_pack_layer_f32: unsupported precision %d", precision);
    PWV_CHECK_ARG(precision != PWV_PREC_F16 || !with_skip, "pwv_pack_layer_f32: PWV_PREC_F16 does not support skip accumulation");
    if (precision != PWV_PREC_F32)   // the fp16 mode reads the `hi` halves of the split-fp16 layout
        return launch_pack_layer_f16x3(filter, gate, dense, dense_bias, skip, skip_bias, gc_filter, gc_gate, with_skip,
                                       cond_channels, packed, (hipStream_t)stream);
    const int total = layer_floats(with_skip != 0, cond_channels > 0);
    hipLaunchKernelGGL(pack_layer_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, filter, gate,
                       dense, dense_bias, skip, skip_bias, gc_filter, gc_gate, with_skip, cond_channels, packed, total);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_pack_first_fold_f32(const float* causal_filter, const float* filter, const float* gate, float* folded, pwv_stream_t stream) {
    PWV_CHECK_ARG(causal_filter && filter && gate && folded, "pwv_pack_first_fold_f32: NULL pointer");
    hipLaunchKernelGGL(pack_first_fold_f32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, causal_filter, filter, gate,
                       reinterpret_cast<f32x4*>(folded));
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_pack_first_fold_f16x3(const float* causal_filter, const float* filter, const float* gate, float* folded, pwv_stream_t stream) {
    PWV_CHECK_ARG(causal_filter && filter && gate && folded, "pwv_pack_first_fold_f16x3: NULL pointer");
    return launch_pack_first_fold_f16x3(causal_filter, filter, gate, folded, (hipStream_t)stream);
}

int pwv_pack_head_f32(const float* skip, const float* skip_bias, const float* post1, const float* post1_bias,
                      const float* post2, const float* post2_bias, int Q, int precision, float* packed,
                      pwv_stream_t stream) {
    PWV_CHECK_ARG(post1 && post2 && packed, "pwv_pack_head_f32: NULL weight pointer");
    PWV_CHECK_ARG(Q >= 1 && Q <= kMaxQ, "pwv_pack_head_f32: Q must be in [1,%d], got %d", kMaxQ, Q);
    PWV_CHECK_ARG(precision >= PWV_PREC_F32 && precision <= PWV_PREC_F16, "pwv_pack_head_f32: unsupported precision %d", precision);
    if (precision != PWV_PREC_F32)
        return launch_pack_head_f16x3(skip, skip_bias, post1, post1_bias, post2, post2_bias, Q, packed, (hipStream_t)stream);
    const int total = head_floats(Q);
    hipLaunchKernelGGL(pack_head_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, skip, skip_bias,
                       post1, post1_bias, post2, post2_bias, Q, packed, total);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_wavenet_layer_f32(const pwv_layer_args* a, pwv_stream_t stream) {
    PWV_CHECK_ARG(a, "pwv_wavenet_layer_f32: args is NULL");
    PWV_CHECK_ARG(a->G >= 1 && a->G <= PWV_MAX_NETS, "pwv_wavenet_layer_f32: G=%d out of range", a->G);
    PWV_CHECK_ARG(a->N >= 1 && a->T >= 1 && a->dilation >= 1, "pwv_wavenet_layer_f32: bad N/T/dilation");
    PWV_CHECK_ARG((long long)a->N * a->T < (1ll << 31) - 256, "pwv_wavenet_layer_f32: N*T too large");
    PWV_CHECK_ARG(a->precision >= PWV_PREC_F32 && a->precision <= PWV_PREC_F16, "pwv_wavenet_layer_f32: unsupported precision %d", a->precision);
    PWV_CHECK_ARG(a->cond_channels == 0 || a->cond_channels == kCondC,
                  "pwv_wavenet_layer_f32: per-sample conditioning supports %d channels", kCondC);
    PWV_CHECK_ARG((a->cond_channels > 0) == (a->cond != nullptr), "pwv_wavenet_layer_f32: cond / cond_channels mismatch");
    PWV_CHECK_ARG(a->proj_row_stride % 4 == 0, "pwv_wavenet_layer_f32: proj_row_stride must be a multiple of 4");
    PWV_CHECK_ARG(a->cond_hop >= 0, "pwv_wavenet_layer_f32: cond_hop < 0");
    LayerParams lp{};
    bool any_skip = false;
    for (int g = 0; g < a->G; ++g) {
        PWV_CHECK_ARG((a->x_in[g] || a->x_first) && (a->x_out[g] || a->head_packed[g]) && a->packed[g] && a->proj[g],
                      "pwv_wavenet_layer_f32: NULL buffer for net %d", g);
        PWV_CHECK_ARG(!a->head_packed[0] == !a->head_packed[g] && (!a->head_packed[g] || a->head_out[g]),
                      "pwv_wavenet_layer_f32: head_packed / head_out must be set for all nets or none");
        lp.packed_head[g] = a->head_packed[g];
        lp.head_out[g] = a->head_out[g];
        PWV_CHECK_ARG(!a->x_first || a->causal_filter[g], "pwv_wavenet_layer_f32: x_first needs causal_filter for net %d", g);
        lp.cfilt[g] = a->causal_filter[g];
        lp.fold0[g] = a->x_first ? a->first_fold[g] : nullptr;
        PWV_CHECK_ARG((lp.fold0[g] == nullptr) == (lp.fold0[0] == nullptr), "pwv_wavenet_layer_f32: first_fold must be set for all nets or for none");
        PWV_CHECK_ARG(a->x_in[g] != a->x_out[g], "pwv_wavenet_layer_f32: in-place layers are not supported (x[t-d] halo)");
        lp.x_in[g] = a->x_in[g];
        lp.x_out[g] = a->x_out[g];
        lp.packed[g] = a->packed[g];
        lp.proj[g] = a->proj[g];
        lp.skip[g] = a->skip[g];
        any_skip = any_skip || a->skip[g];
    }
    for (int g = 0; g < a->G; ++g)
        PWV_CHECK_ARG(!any_skip || a->skip[g], "pwv_wavenet_layer_f32: skip must be set for all nets or none");
    const bool half16 = a->precision == PWV_PREC_F16;      // (its fused head has room for the per-sample condition weights)
    PWV_CHECK_ARG(!a->x_first || !any_skip, "pwv_wavenet_layer_f32: x_first does not support skip accumulation");
    PWV_CHECK_ARG(!a->head_packed[0] || (a->out_mode == PWV_OUT_GATED && !any_skip && (!a->cond || half16) && !a->x_first && a->head_q >= 1 && a->head_q <= kMaxQ),
                  "pwv_wavenet_layer_f32: a fused head needs out_mode PWV_OUT_GATED, no skip accumulation, no per-sample condition (PWV_PREC_F16: "
                  "allowed) and head_q in [1,%d]", kMaxQ);
    lp.head_q = a->head_q;
    lp.x_first = a->x_first;
    lp.x_limit = a->x_limit;
    lp.range_flag = a->x_first ? a->range_flag : nullptr;
    lp.cond = a->cond;
    lp.proj_row_stride = a->proj_row_stride;
    lp.G = a->G;
    lp.N = a->N;
    lp.T = a->T;
    lp.dilation = a->dilation;
    lp.cond_hop = a->cond_hop;
    lp.cond_offset = a->cond_offset;
    lp.cond_frames = a->cond_frames;
    lp.skip_init = a->skip_init;
    make_magic((unsigned)a->T, lp.T_magic, lp.T_shift);
    make_magic((unsigned)(a->cond_hop > 0 ? a->cond_hop : 1), lp.hop_magic, lp.hop_shift);
    lp.trace = nullptr;
#ifdef PWV_TRACE
    { const char* e = getenv("PWV_TRACE_PTR"); if (e) lp.trace = (long long*)strtoull(e, nullptr, 0); }
#endif

    const int cus = device_cus();
    if (cus <= 0) return set_error(PWV_EHIP, "no HIP device");
    const long long rows = (long long)a->N * a->T;
    int per_net = (a->max_workgroups > 0 ? a->max_workgroups : cus) / a->G;
    if (per_net < 1) per_net = 1;
    const int nt4 = (int)((rows + 127) / 128), nt8 = (int)((rows + 255) / 256);   // >= 1 unit per wave
    const int g4 = per_net < nt4 ? per_net : nt4, g8 = per_net < nt8 ? per_net : nt8;
    // the write-through stores address a workgroup's own units with 32-bit byte offsets (units_rsrc / units_off): 16 KB per
    // unit of the widest buffer (skip accumulators) must stay below 2 GB per workgroup
    {
        const int gmin = g4 < g8 ? g4 : g8;
        PWV_CHECK_ARG(((rows + 31) / 32 + gmin - 1) / gmin < (1 << 17),
                      "pwv_wavenet_layer_f32: %lld rows on %d workgroups per net: more than 131071 units per workgroup (raise max_workgroups)",
                      rows, gmin);
    }
    hipStream_t s = (hipStream_t)stream;
    const bool cond = a->cond != nullptr, gated = a->out_mode == PWV_OUT_GATED;
    PWV_CHECK_ARG(a->out_mode == PWV_OUT_GATED || a->out_mode == PWV_OUT_RESIDUAL, "pwv_wavenet_layer_f32: bad out_mode");
    if (a->precision == PWV_PREC_F16X3) return launch_layer_f16x3(lp, any_skip, cond, gated, g8, s);
    if (a->precision == PWV_PREC_F16) {
        PWV_CHECK_ARG(!any_skip, "pwv_wavenet_layer_f32: PWV_PREC_F16 does not support skip accumulation");
        // 4-wave workgroups with 40 KB (60 KB with cond) of LDS: several per CU
        if (lp.packed_head[0]) return launch_layer_h16(lp, cond, gated, g8, s);      // fused head: one 8-wave workgroup per CU
        const int want = per_net * (cond ? 2 : 3);
        return launch_layer_h16(lp, cond, gated, want < nt4 ? want : nt4, s);
    }
    if (lp.packed_head[0]) return launch_layer_fused<false, true, false, true>(lp, g8, s);
    if (lp.x_first && lp.fold0[0]) {
        if (cond) return gated ? launch_layer_fused<true, true, true, false, true>(lp, g8, s) : launch_layer_fused<true, false, true, false, true>(lp, g8, s);
        return gated ? launch_layer_fused<false, true, true, false, true>(lp, g8, s) : launch_layer_fused<false, false, true, false, true>(lp, g8, s);
    }
    if (lp.x_first) {
        if (cond) return gated ? launch_layer_fused<true, true, true, false>(lp, g8, s) : launch_layer_fused<true, false, true, false>(lp, g8, s);
        return gated ? launch_layer_fused<false, true, true, false>(lp, g8, s) : launch_layer_fused<false, false, true, false>(lp, g8, s);
    }
    if (any_skip) {
        if (cond) return gated ? launch_layer<true, true, true>(lp, g4, g8, s) : launch_layer<true, true, false>(lp, g4, g8, s);
        return gated ? launch_layer<true, false, true>(lp, g4, g8, s) : launch_layer<true, false, false>(lp, g4, g8, s);
    }
    if (cond) return gated ? launch_layer<false, true, true>(lp, g4, g8, s) : launch_layer<false, true, false>(lp, g4, g8, s);
    return gated ? launch_layer<false, false, true>(lp, g4, g8, s) : launch_layer<false, false, false>(lp, g4, g8, s);
}

int pwv_wavenet_head_f32(const pwv_head_args* a, pwv_stream_t stream) {
    PWV_CHECK_ARG(a, "pwv_wavenet_head_f32: args is NULL");
    PWV_CHECK_ARG(a->G >= 1 && a->G <= PWV_MAX_NETS, "pwv_wavenet_head_f32: G=%d out of range", a->G);
    PWV_CHECK_ARG(a->N >= 1 && a->T >= 1, "pwv_wavenet_head_f32: bad N/T");
    PWV_CHECK_ARG((long long)a->N * a->T < (1ll << 31) - 256, "pwv_wavenet_head_f32: N*T too large");
    PWV_CHECK_ARG(a->Q >= 1 && a->Q <= kMaxQ, "pwv_wavenet_head_f32: Q must be in [1,%d]", kMaxQ);
    PWV_CHECK_ARG(a->precision >= PWV_PREC_F32 && a->precision <= PWV_PREC_F16, "pwv_wavenet_head_f32: unsupported precision %d", a->precision);
    PWV_CHECK_ARG(a->in_mode == PWV_HEAD_IN_GATED || a->in_mode == PWV_HEAD_IN_SKIPSUM, "pwv_wavenet_head_f32: bad in_mode");
    HeadParams hp{};
    for (int g = 0; g < a->G; ++g) {
        PWV_CHECK_ARG(a->in[g] && a->packed[g] && a->out[g], "pwv_wavenet_head_f32: NULL buffer for net %d", g);
        hp.in[g] = a->in[g];
        hp.packed[g] = a->packed[g];
        hp.out[g] = a->out[g];
    }
    hp.G = a->G;
    hp.N = a->N;
    hp.T = a->T;
    hp.Q = a->Q;
    const int cus = device_cus();
    if (cus <= 0) return set_error(PWV_EHIP, "no HIP device");
    const long long rows = (long long)a->N * a->T;
    int ntiles = (int)((rows + 127) / 128);
    int per_net = (a->max_workgroups > 0 ? a->max_workgroups : cus) / a->G;
    if (per_net < 1) per_net = 1;
    if (a->precision == PWV_PREC_F16X3) ntiles = (int)((rows + 255) / 256);   // 8-wave workgroups: >= 1 unit per wave
    if (a->precision == PWV_PREC_F16) {
        PWV_CHECK_ARG(a->in_mode == PWV_HEAD_IN_GATED, "pwv_wavenet_head_f32: PWV_PREC_F16 needs in_mode PWV_HEAD_IN_GATED");
        per_net *= 3;
        if (per_net > ntiles) per_net = ntiles;
        return launch_head_h16(hp, per_net * a->G, (hipStream_t)stream);
    }
    if (per_net > ntiles) per_net = ntiles;
    const int grid = per_net * a->G;
    if (a->precision == PWV_PREC_F16X3)
        return launch_head_f16x3(hp, a->in_mode == PWV_HEAD_IN_GATED, grid, (hipStream_t)stream);
    if (a->in_mode == PWV_HEAD_IN_GATED)
        hipLaunchKernelGGL((head_f32_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, hp);
    else
        hipLaunchKernelGGL((head_f32_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, hp);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_wavenet_stack_f32(const pwv_stack_args* a, pwv_stream_t const* streams) {
    PWV_CHECK_ARG(a && streams, "pwv_wavenet_stack_f32: NULL args / stream array");   // streams[0] may be the NULL (default) stream
    PWV_CHECK_ARG(a->G >= 1 && a->G <= PWV_MAX_NETS, "pwv_wavenet_stack_f32: G=%d out of range", a->G);
    PWV_CHECK_ARG(a->n_layers >= 1 && a->dilations, "pwv_wavenet_stack_f32: no layers");
    const bool two = a->G == 2 && streams[1] != nullptr;
    const int groups = two ? 2 : 1;           // launch groups per layer
    const int per_group = two ? 1 : a->G;     // nets per launch
    int wgs = a->max_workgroups;
    if (two && wgs == 0) {
        const int cus = device_cus();
        if (cus <= 0) return set_error(PWV_EHIP, "no HIP device");
        wgs = cus / 2 > 0 ? cus / 2 : 1;
    }
    const bool use_skip = a->skip[0] != nullptr;
    // split-fp16 and fp32 paths, plain last layer: the head runs inside the last layer's launch (pwv_layer_args.head_packed)
    const bool fuse_head = !a->separate_head && !use_skip && (!a->cond || a->precision == PWV_PREC_F16) && a->n_layers >= 2;
    int cur = 0;
    for (int j = 0; j < a->n_layers; ++j) {
        const bool last = j == a->n_layers - 1;
        for (int grp = 0; grp < groups; ++grp) {
            pwv_layer_args la{};
            la.G = per_group;
            for (int i = 0; i < per_group; ++i) {
                const int g = two ? grp : i;
                la.x_in[i] = cur ? a->buf1[g] : a->buf0[g];
                la.x_out[i] = cur ? a->buf0[g] : a->buf1[g];
                la.packed[i] = a->packed_layers[g] + (size_t)j * a->packed_layer_stride;
                la.proj[i] = a->proj[g] + (size_t)128 * j;
                la.skip[i] = use_skip ? a->skip[g] : nullptr;
            }
            la.proj_row_stride = a->proj_row_stride;
            la.cond = a->cond;
            la.cond_channels = a->cond_channels;
            la.skip_init = j == 0;
            la.N = a->N;
            la.T = a->T;
            la.dilation = a->dilations[j];
            la.cond_hop = a->cond_hop;
            la.cond_offset = a->cond_offset;
            la.cond_frames = a->cond_frames;
            la.out_mode = last ? PWV_OUT_GATED : PWV_OUT_RESIDUAL;
            la.precision = a->precision;
            la.max_workgroups = wgs;
            if (last && fuse_head) {
                for (int i = 0; i < per_group; ++i) {
                    const int g = two ? grp : i;
                    la.head_packed[i] = a->packed_head[g];
                    la.head_out[i] = a->out[g];
                }
                la.head_q = a->Q;
            }
            if (j == 0 && a->x_first) {
                la.x_first = a->x_first;
                la.x_limit = a->x_limit;
                la.range_flag = a->range_flag;
                for (int i = 0; i < per_group; ++i) {
                    la.causal_filter[i] = a->causal_filter[two ? grp : i];
                    la.first_fold[i] = a->first_fold[two ? grp : i];
                }
            }
            if (j == 0 && a->ev_begin[grp]) PWV_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_begin[grp], (hipStream_t)streams[grp]));
            const int rc = pwv_wavenet_layer_f32(&la, streams[grp]);
            if (rc != PWV_OK) return rc;
            if (j == a->n_layers - 2 && a->ev_end[grp]) PWV_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_end[grp], (hipStream_t)streams[grp]));
        }
        cur ^= 1;
    }
    for (int grp = 0; grp < groups && !fuse_head; ++grp) {
        pwv_head_args ha{};
        ha.G = per_group;
        for (int i = 0; i < per_group; ++i) {
            const int g = two ? grp : i;
            ha.in[i] = use_skip ? a->skip[g] : (cur ? a->buf1[g] : a->buf0[g]);
            ha.packed[i] = a->packed_head[g];
            ha.out[i] = a->out[g];
        }
        ha.N = a->N;
        ha.T = a->T;
        ha.Q = a->Q;
        ha.in_mode = use_skip ? PWV_HEAD_IN_SKIPSUM : PWV_HEAD_IN_GATED;
        ha.precision = a->precision;
        ha.max_workgroups = wgs;
        const int rc = pwv_wavenet_head_f32(&ha, streams[grp]);
        if (rc != PWV_OK) return rc;
    }
    return PWV_OK;
}

}  // extern "C"
