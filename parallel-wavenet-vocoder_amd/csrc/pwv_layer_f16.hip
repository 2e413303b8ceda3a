// Split-fp16 ("f16x3") variants of the fused layer / head kernels for gfx950.
//
// Same math, layouts in HBM, register ownership and launch structure as pwv_layer.hip (the
// reference call sites are /root/reference/modules.py:185-259 and :145-165), but the GEMMs run on
// v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate) with error compensation:
//
//     w*x  ~=  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi,     v_hi = fp16(v),  v_lo = fp16(v - v_hi)
//
// accumulated in fp32.  fp16 products are exact in the fp32 accumulator; the dropped term w_lo*x_lo
// is ~2^-22 |w x|, and each operand is represented to max(2^-22 |v|, 2^-25) (gfx950 keeps fp16
// subnormals in MFMA inputs and conversions -- tools/probes/f16_denorm.hip -- so no rescaling of the
// low parts is needed).  Activations stay fp32 in HBM; the split happens in registers.
// Per 32-sample unit: 120 MFMAs x 32 cycles = 3,840 matrix cycles instead of 320 x 64 = 20,480.
#include "pwv_f16x3.h"

namespace pwv {

// --------------------------------------------------------------------------------------
// fused gated-residual layer, split-fp16 arithmetic (8 waves, dynamic 32-sample units)
// --------------------------------------------------------------------------------------
// FIRST: layer 0 of a scalar-input net WITHOUT a materialised causal layer.  h[t] = x[t-1] w0 + x[t] w1
// (modules.py:179-180) is a rank-2 function of two scalars per row, so instead of writing [rows, 64] floats in a
// front kernel and reading them back twice (x[t], x[t-d]), the lane rebuilds its 32 channels of both rows from four
// scalars with the same two fp32 operations per channel the front kernel uses (bit-identical): 4 B instead of 768 B of
// traffic per sample for this layer, and no front launch.
//
// HEAD: the LAST layer with the post-processing head (modules.py:145-165) fused behind it.  The gated output o never
// leaves the registers it was accumulated in -- it is the B operand of the skip GEMM -- so the [rows, 64] round trip
// through HBM between the last layer and the head (512 B per sample and net) and the head launch disappear.  LDS holds
// exactly the three weight matrices (filter|gate 64 KB + skip 32 KB + postprocess1 64 KB = all 160 KB of the CU): the
// small vectors (skip / postprocess1 biases, postprocess2) are read from global memory per unit like P, and units are
// handed out statically (no room for a counter).
//
// FOLD (with FIRST): h[t] = x[t-1] w0 + x[t] w1 makes the filter|gate convolution of layer 0 a [4 -> 128] map of the scalars
// x[t-d-1], x[t-d], x[t-1], x[t] (pwv_pack_first_fold_f16x3): ONE MFMA k-step (4 of its 16 k values used) instead of eight,
// no LDS fragment reads and no operand splits for it.  The same function, rounded differently; the persistent kernel's
// folded layer 0 (pwv_stack_persist.hip) performs exactly these operations, so the two stay bit-identical.
template <bool SKIP, bool COND, bool GATED, bool FIRST = false, bool HEAD = false, bool FOLD = false>
__global__ __launch_bounds__(512) void layer_f16x3_kernel(const LayerParams p) {
    static_assert(!HEAD || (GATED && !SKIP && !COND && !FIRST), "HEAD: plain last layer only");
    static_assert(!FOLD || FIRST, "FOLD: layer 0 of a scalar-input net only");
    constexpr int WAVES = 8;
    constexpr int kLds = HEAD ? kA1Size + kASSize + kHA1Size : layer_floats(SKIP, COND);
    constexpr int kCF = kLds + 4;           // FIRST: the causal filter [2][64] behind the unit counter
    constexpr int kHS = kA1Size;            // HEAD: skip weights, then postprocess1
    constexpr int kH1 = kA1Size + kASSize;
    __shared__ __attribute__((aligned(16))) float lds[kLds + (HEAD ? 0 : 4) + (FIRST ? 128 : 0)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int net = blockIdx.x % p.G;
    const int wg = blockIdx.x / p.G;
    const int nwg = gridDim.x / p.G;
    int* unit_counter = reinterpret_cast<int*>(&lds[HEAD ? 0 : kLds]);      // unused with HEAD
#ifdef PWV_TRACE
    if (p.trace && tid == 0) {
        p.trace[4096 + blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        p.trace[4096 + 1024 + blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, chip-wide
    }
#endif
    // the first WAVES units are handed out statically (wave w takes unit w: no LDS round trip before the first rows can
    // be requested); the counter then starts at WAVES
    if constexpr (!HEAD) {
        if (tid == 0) *unit_counter = WAVES;
    }

    constexpr int kAS = kLayerBase;
    constexpr int kBS = kAS + kASSize;
    constexpr int kAC = kLayerBase + (SKIP ? kASSize + kBSSize : 0);
    const f16x8* A1 = reinterpret_cast<const f16x8*>(&lds[kA1]);
    const f16x8* A2 = reinterpret_cast<const f16x8*>(&lds[kA2]);
    const f16x8* AS = reinterpret_cast<const f16x8*>(&lds[kAS]);
    const f16x8* AC = reinterpret_cast<const f16x8*>(&lds[kAC]);

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
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
    auto no_extra = [](int) {};
    const __amdgpu_buffer_rsrc_t out_rs = units_rsrc(p.x_out[net], u_begin, u_end, 32 * 64 * 4);      // unused with HEAD

    // x[t-d] / x[t] rows of one unit -> registers (clamped addresses, zeros left of the utterance start)
    auto load_x = [&](int unit, float (&xb)[32], float (&xc)[32]) {
        int row, rc, n, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        const bool has_prev = t >= p.dilation;
        if constexpr (FIRST) {
            // the four scalars the two rows are functions of: x[t], x[t-1], x[t-d], x[t-d-1] (zero left of the start)
            const float* x1 = p.x_first;
            const int d = p.dilation;
            xc[0] = x1[rc];
            xc[1] = t >= 1 ? x1[rc - (t >= 1 ? 1 : 0)] : 0.f;
            xb[0] = has_prev ? x1[rc - (has_prev ? d : 0)] : 0.f;
            xb[1] = t >= d + 1 ? x1[rc - (t >= d + 1 ? d + 1 : 0)] : 0.f;
            return;
        }
        load_tiled<8, 64>(p.x_in[net], rc, h, true, xc);
        if (__all(has_prev)) {      // wave-uniform fast path: no per-register select
            load_tiled<8, 64>(p.x_in[net], rc - p.dilation, h, true, xb);
        } else {
            load_tiled<8, 64>(p.x_in[net], has_prev ? rc - p.dilation : rc, h, has_prev, xb);
        }
    };

    int tr_unit = -1;
    (void)tr_unit;
    int unit = u_begin + wave;
    float rxb[32], rxc[32];      // raw rows of the current unit (prefetched during the previous unit's GEMM2)
    if constexpr (FIRST) {
        if (tid < 128) lds[kCF + tid] = p.cfilt[net][tid];
    }
    if constexpr (HEAD) {
        fill_lds_dma<kA1Size / 4, WAVES>(lds, p.packed[net] + kA1, wave, lane);
        fill_lds_dma<kASSize / 4, WAVES>(lds + kHS, p.packed_head[net] + kHAS, wave, lane);
        fill_lds_dma<kHA1Size / 4, WAVES>(lds + kH1, p.packed_head[net] + kHA1, wave, lane);
    } else {
        fill_lds_dma<kLds / 4, WAVES>(lds, p.packed[net], wave, lane);
    }
    __syncthreads();
#ifdef PWV_TRACE
    if (p.trace && tid == 0) p.trace[4096 + blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif
    // The first unit's rows are requested BEHIND the weights, not ahead of them: vector-memory results return in order per
    // wave, so rows asked for first (HBM, microseconds under the start-of-kernel burst of all CUs) hold back the weight
    // transfers (L2 hits) and with them the barrier.  Measured with tools/trace_layer.py: barrier reached after 3.2k instead
    // of 9.1k cycles, first unit under way after 5.7k instead of 9.3k.  (Keeping the rows in flight ACROSS the barrier
    // with a partial `s_waitcnt vmcnt(16)` is not worth it: the compiler's wait-count bookkeeping does not credit a partial
    // wait while loads and LDS-DMA are both outstanding and then puts a vmcnt(0) in front of the first LDS access of
    // every unit.)
    load_x(unit, rxb, rxc);
    while (unit < u_end) {
        const int next = HEAD ? unit + WAVES : grab();
        ++tr_unit;
        PWV_STAMP(0);
        int row, rc, n, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        if (next >= u_end) __builtin_amdgcn_s_setprio(2);   // this wave's last unit: do not let it become the tail

        // accumulators start at P[frame(t)] (issued first: lands while x is being split)
        f32x16 acc[4];
        {
            int prow = 0;
            if (p.cond_hop > 0) prow = n * p.cond_frames + fast_div(t + p.cond_offset, p.hop_magic, p.hop_shift);
            const float* pr = p.proj[net] + (size_t)prow * p.proj_row_stride + h * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(pr + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                }
        }
        f16x8 ch[5], cl[5];      // per-sample condition, K = 80
        if constexpr (COND) {
            // pre-split fp16 planes (pwv_cond_split_f16): [hi | lo], each tile32-style [10 chunks][32 rows][8 halfs];
            // chunk 2s + h is this lane's B operand of k-step s
            const _Float16* c16 = reinterpret_cast<const _Float16*>(p.cond);
            const _Float16* cr = c16 + (size_t)(rc >> 5) * (32 * kCondC) + (h * 32 + (rc & 31)) * 8;
            const size_t plane = tile32_floats(rows, kCondC);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                ch[s] = *reinterpret_cast<const f16x8*>(cr + s * 512);
                cl[s] = *reinterpret_cast<const f16x8*>(cr + plane + s * 512);
            }
        }
        PWV_STAMP(1);
#ifdef PWV_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PWV_STAMP(2);
        f16x8 fb_h = {0, 0, 0, 0, 0, 0, 0, 0}, fb_l = {0, 0, 0, 0, 0, 0, 0, 0};      // FOLD: the scalars as ONE B operand (k = 0..3, lower half)
        if constexpr (FIRST) {
            // rebuild this lane's 32 channels (8g + 4h + e) of h[t] and h[t-d] from the scalars; same operation order
            // as iaf_front_kernel: round(x[t-1] w0), then fma(x[t], w1, .)
            const float x0 = rxc[0], x1v = rxc[1], xd0 = rxb[0], xd1 = rxb[1];
            const bool has_prev = t >= p.dilation;
            // range guard (include/pwv_hip.h): every row is some lane's x[t]; NaN fails the comparison too
            if (p.range_flag && !(fabsf(x0) <= p.x_limit)) __hip_atomic_store(p.range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if constexpr (FOLD) {
                const float sc[4] = {xd1, xd0, x1v, x0};      // (x[t-d], x[t-d-1] are already zero left of the start)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = h == 0 ? sc[q] : 0.f;
                    const _Float16 vh = (_Float16)v;
                    fb_h[q] = vh;
                    fb_l[q] = (_Float16)(v - (float)vh);
                }
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(&lds[kCF + 8 * g + 4 * h]);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(&lds[kCF + 64 + 8 * g + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rxc[4 * g + e] = fmaf(x0, w1[e], x1v * w0[e]);      // (h[t]: the residual add needs it in every form)
                    if constexpr (!FOLD) {
                        const float vb = fmaf(xd0, w1[e], xd1 * w0[e]);
                        rxb[4 * g + e] = has_prev ? vb : 0.f;
                    }
                }
            }
        }
        // B operands: bh[0..3] = x[t-d], bh[4..7] = x[t].  The packed K order is x[t] FIRST (k-steps 0..3), then x[t-d] (round 6): a unit's own
        // rows are what a stationary wave of the persistent kernel still holds in registers, the look-back row is what it waits for -- every
        // kernel accumulates in that order, so all paths stay bit-identical.  x[t-d] is split under the first four MFMA groups of pair 0.
        f16x8 bh[8], bl[8];
        float xc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) xc[i] = rxc[i];
        // (with a per-sample condition its GEMM comes first: x[t-d] is split up front, as before, and x[t] under the condition GEMM's first
        //  four groups -- any more operands in front of that GEMM are spilled registers)
        auto split_xc = [&](int s) {
            if (s == 0) { split8<0>(xc, bh[4], bl[4]); asm volatile("" : "+v"(bh[4]), "+v"(bl[4])); }
            if (s == 1) { split8<8>(xc, bh[5], bl[5]); asm volatile("" : "+v"(bh[5]), "+v"(bl[5])); }
            if (s == 2) { split8<16>(xc, bh[6], bl[6]); asm volatile("" : "+v"(bh[6]), "+v"(bl[6])); }
            if (s == 3) { split8<24>(xc, bh[7], bl[7]); asm volatile("" : "+v"(bh[7]), "+v"(bl[7])); }
        };
        auto split_xb = [&](int s) {
            if (s == 0) { split8<0>(rxb, bh[0], bl[0]); asm volatile("" : "+v"(bh[0]), "+v"(bl[0])); }
            if (s == 1) { split8<8>(rxb, bh[1], bl[1]); asm volatile("" : "+v"(bh[1]), "+v"(bl[1])); }
            if (s == 2) { split8<16>(rxb, bh[2], bl[2]); asm volatile("" : "+v"(bh[2]), "+v"(bl[2])); }
            if (s == 3) { split8<24>(rxb, bh[3], bl[3]); asm volatile("" : "+v"(bh[3]), "+v"(bl[3])); }
        };
        if constexpr (!FOLD) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (COND) split_xb(s);
                else split_xc(s);
            }
        }

        auto bxh = [&](int s) -> f16x8 { return bh[s ^ 4]; };
        auto bxl = [&](int s) -> f16x8 { return bl[s ^ 4]; };
        auto bch = [&](int s) -> f16x8 { return ch[COND ? s : 0]; };
        auto bcl = [&](int s) -> f16x8 { return cl[COND ? s : 0]; };

        PWV_STAMP(3);
        float o[32];
        f16x8 oh[4], ol[4];      // gated output as B operand: k-step s <-> o tile s>>1, regs 8*(s&1)..+7
        f16x8 ah[4], al[4];

        if constexpr (FOLD) {
            // ---- layer 0, folded: the per-sample condition's GEMM (if any), then ONE k-step on the scalars ----------------
            if constexpr (COND) {
                first_frags<5, 2, 0, 2, 4>(AC, lane, ah, al);
                gemm16<5, 2, 0, 2, 4>(AC, lane, acc, ah, al, bch, bcl, no_extra,
                                      [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<5, 2, 1, 2, 4>(AC, lane, nh, nl); });
                gemm16<5, 2, 1, 2, 4>(AC, lane, acc, ah, al, bch, bcl, no_extra, [](f16x8(&)[4], f16x8(&)[4]) {});
            }
            const f16x8* F0 = reinterpret_cast<const f16x8*>(p.fold0[net]);
            f16x8 fh[4], fl[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                fh[it] = F0[it * 64 + lane];
                fl[it] = F0[(4 + it) * 64 + lane];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[it], fb_h, acc[it], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[it], fb_l, acc[it], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[it], fb_h, acc[it], 0, 0, 0);
            if constexpr (!GATED) first_frags<4, 2, 0, 1, 2>(A2, lane, ah, al);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = gate_act(acc[0][r], acc[2][r]);
            split8<0>(o, oh[0], ol[0]);
            split8<8>(o, oh[1], ol[1]);
        } else {
        // ---- GEMM1, row-tile pair 0 = (F[0:32], G[0:32]) ----------------------------------------
        if constexpr (COND) {
            first_frags<5, 2, 0, 2, 4>(AC, lane, ah, al);
            gemm16<5, 2, 0, 2, 4>(AC, lane, acc, ah, al, bch, bcl, split_xc,
                                  [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 2, 0, 2, 4>(A1, lane, nh, nl); });
        } else {
            first_frags<8, 2, 0, 2, 4>(A1, lane, ah, al);
        }
        gemm16<8, 2, 0, 2, 4>(
            A1, lane, acc, ah, al, bxh, bxl,
            [&](int s) {
                if constexpr (!COND) split_xb(s);
            },
            [&](f16x8(&nh)[4], f16x8(&nl)[4]) {
                if constexpr (COND) first_frags<5, 2, 1, 2, 4>(AC, lane, nh, nl);
                else first_frags<8, 2, 1, 2, 4>(A1, lane, nh, nl);
            });
        PWV_STAMP(4);
        // ---- pair 1 = (F[32:64], G[32:64]); pair 0 is gated + split under these MFMAs -------------
        if constexpr (COND) {
            gemm16<5, 2, 1, 2, 4>(AC, lane, acc, ah, al, bch, bcl, no_extra,
                                  [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 2, 1, 2, 4>(A1, lane, nh, nl); });
        }
        gemm16<8, 2, 1, 2, 4>(
            A1, lane, acc, ah, al, bxh, bxl,
            [&](int s) {
                o[2 * s] = gate_act(acc[0][2 * s], acc[2][2 * s]);
                o[2 * s + 1] = gate_act(acc[0][2 * s + 1], acc[2][2 * s + 1]);
                asm volatile("" : "+v"(o[2 * s]), "+v"(o[2 * s + 1]));
                if (s == 3) {
                    split8<0>(o, oh[0], ol[0]);
                    asm volatile("" : "+v"(oh[0]), "+v"(ol[0]));
                }
                if (s == 7) {
                    split8<8>(o, oh[1], ol[1]);
                    asm volatile("" : "+v"(oh[1]), "+v"(ol[1]));
                }
            },
            [&](f16x8(&nh)[4], f16x8(&nl)[4]) {
                if constexpr (!GATED) first_frags<4, 2, 0, 1, 2>(A2, lane, nh, nl);
                else if constexpr (SKIP) first_frags<4, 2, 0, 1, 4>(AS, lane, nh, nl);
            });
        }

        PWV_STAMP(5);
        const int ooff = units_off(row, h, 64, u_begin);      // (rows past the end are never stored: `valid`)
        if constexpr (GATED && HEAD) {
            // ---- fused head: o (registers) -> skip -> relu -> postprocess1 -> relu -> postprocess2 -------------------
            const float* hb = p.packed_head[net];
            const f16x8* HS = reinterpret_cast<const f16x8*>(&lds[kHS]);
            const f16x8* H1 = reinterpret_cast<const f16x8*>(&lds[kH1]);
            f32x16 accs[4];      // starts at the skip bias (requested now, lands while pair 1 is gated)
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
            split8<16>(o, oh[2], ol[2]);
            split8<24>(o, oh[3], ol[3]);
            first_frags<4, 4, 0, 1, 4>(HS, lane, ah, al);
            gemm16<4, 4, 0, 1, 4>(HS, lane, accs, ah, al, [&](int s) -> f16x8 { return oh[s]; },
                                  [&](int s) -> f16x8 { return ol[s]; }, no_extra,
                                  [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 4, 0, 1, 4>(H1, lane, nh, nl); });
            f32x16 acc1[4];      // starts at the postprocess1 bias
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(hb + kHB1 + h * 64 + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc1[it][q * 4 + e] = v[e];
                }
            f16x8 sh[8], sl[8];
            {
                float r[64];
#pragma unroll
                for (int i = 0; i < 64; ++i) r[i] = fmaxf(accs[i >> 4][i & 15], 0.f);
                split8<0>(r, sh[0], sl[0]);
                split8<8>(r, sh[1], sl[1]);
                split8<16>(r, sh[2], sl[2]);
                split8<24>(r, sh[3], sl[3]);
                split8<32>(r, sh[4], sl[4]);
                split8<40>(r, sh[5], sl[5]);
                split8<48>(r, sh[6], sl[6]);
                split8<56>(r, sh[7], sl[7]);
            }
            gemm16<8, 4, 0, 1, 4>(H1, lane, acc1, ah, al, [&](int s) -> f16x8 { return sh[s]; },
                                  [&](int s) -> f16x8 { return sl[s]; }, no_extra, [](f16x8(&)[4], f16x8(&)[4]) {});
            load_x(next, rxb, rxc);      // the next unit's rows: in flight under the postprocess2 dot
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
                if (valid && h == 0) p.head_out[net][(size_t)row * Q + q] = part;
            }
        } else if constexpr (GATED) {
            if constexpr (!SKIP) load_x(next, rxb, rxc);      // (SKIP: requested in front of the skip GEMM, see below)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[16 + r] = gate_act(acc[1][r], acc[3][r]);
            if constexpr (SKIP) {
                split8<16>(o, oh[2], ol[2]);
                split8<24>(o, oh[3], ol[3]);
            }
            if (valid) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                    store_wt(out_rs, ooff + g * 1024, v);
                }
            }
        } else {
            // ---- GEMM2: dense 64 -> 64, accumulator starts at x[t] + dense_bias -------------------
            f32x16 acc2[2];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bd = *reinterpret_cast<const f32x4*>(&lds[kBD + h * 32 + it * 16 + q * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = xc[it * 16 + q * 4 + e] + bd[e];
                }
            // next unit's rows: in flight under GEMM2 + gating + stores (xc is dead from here on).  With skip accumulation they are
            // requested behind this layer's stores instead, in front of the skip GEMM (48 MFMAs of cover): their 64 registers on top
            // of GEMM2's working set were the 19-37 spilled VGPRs of the SKIP variants
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]));
            if constexpr (!SKIP) load_x(next, rxb, rxc);
            __builtin_amdgcn_sched_barrier(0);
            gemm16<4, 2, 0, 1, 2>(
                A2, lane, acc2, ah, al, [&](int s) -> f16x8 { return oh[s]; }, [&](int s) -> f16x8 { return ol[s]; },
                [&](int s) {
                    if (s < 2) {   // k-steps 0,1 use o tile 0; gate + split tile 1 under them
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[16 + 8 * s + e] = gate_act(acc[1][8 * s + e], acc[3][8 * s + e]);
                        if (s == 0) split8<16>(o, oh[2], ol[2]);
                        else split8<24>(o, oh[3], ol[3]);
                        asm volatile("" : "+v"(oh[2 + (s & 1)]), "+v"(ol[2 + (s & 1)]));
                    }
                },
                [&](f16x8(&nh)[4], f16x8(&nl)[4]) {
                    if constexpr (SKIP) first_frags<4, 2, 0, 1, 4>(AS, lane, nh, nl);
                });
            PWV_STAMP(6);
            if (valid) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int it = g >> 2, q = g & 3;
                    f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
                    store_wt(out_rs, ooff + g * 1024, v);
                }
            }
            PWV_STAMP(7);
        }

        if constexpr (SKIP) {
            // ---- skip 64 -> 128, accumulated across layers: two passes of two row tiles (64 outputs) each.  One pass over all four
            // tiles held 64 accumulator + 64 fragment registers (current + next k-step of four tiles) next to the gated operand and the
            // prefetched rows: 19-37 spilled VGPRs in every SKIP variant.  Same per-accumulator operation order.
            __builtin_amdgcn_sched_barrier(0);
            load_x(next, rxb, rxc);      // the next unit's rows: in flight under the skip GEMM (48 MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            f32x16 accs[4];
            float* srow = p.skip[net] + tile_off(rc, h, 128);
            const bool skip_load = !p.skip_init;
            const __amdgpu_buffer_rsrc_t skip_rs = units_rsrc(p.skip[net], u_begin, u_end, 32 * 128 * 4);
            const int soff = units_off(row, h, 128, u_begin);
            auto skip_init2 = [&](int it0) {
#pragma unroll
                for (int it = it0; it < it0 + 2; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bs = *reinterpret_cast<const f32x4*>(&lds[kBS + h * 64 + it * 16 + q * 4]);
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (skip_load) v = *reinterpret_cast<const f32x4*>(srow + (8 * it + 2 * q) * 128);
#pragma unroll
                        for (int e = 0; e < 4; ++e) accs[it][q * 4 + e] = v[e] + bs[e];
                    }
            };
            auto skip_store2 = [&](int it0) {
                if (valid) {
#pragma unroll
                    for (int it = it0; it < it0 + 2; ++it)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = {accs[it][q * 4], accs[it][q * 4 + 1], accs[it][q * 4 + 2], accs[it][q * 4 + 3]};
                            store_wt(skip_rs, soff + (8 * it + 2 * q) * 512, v);
                        }
                }
            };
            skip_init2(0);
            gemm16<4, 2, 0, 1, 4>(AS, lane, accs, ah, al, [&](int s) -> f16x8 { return oh[s]; }, [&](int s) -> f16x8 { return ol[s]; }, no_extra,
                                  [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<4, 2, 2, 1, 4>(AS, lane, nh, nl); });
            skip_store2(0);
            __builtin_amdgcn_sched_barrier(0);
            skip_init2(2);
            gemm16<4, 2, 2, 1, 4>(AS, lane, accs, ah, al, [&](int s) -> f16x8 { return oh[s]; }, [&](int s) -> f16x8 { return ol[s]; }, no_extra,
                                  [](f16x8(&)[4], f16x8(&)[4]) {});
            skip_store2(2);
        }
        __builtin_amdgcn_sched_barrier(0);
        unit = next;
    }
#ifdef PWV_TRACE
    if (p.trace && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_max(&p.trace[4096 + blockIdx.x * 4 + 2], (long long)__builtin_amdgcn_s_memtime(), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&p.trace[4096 + 1024 + blockIdx.x * 2 + 1], (long long)__builtin_amdgcn_s_memrealtime(),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

// --------------------------------------------------------------------------------------
// head, split-fp16 arithmetic (8 waves = 2 per SIMD, dynamic 32-row units out of a contiguous per-workgroup
// range like the layer kernel; post2 stays an fp32 VALU dot)
// --------------------------------------------------------------------------------------
template <bool FROM_GATED>
__global__ __launch_bounds__(512) void head_f16x3_kernel(const HeadParams p) {
    constexpr int WAVES = 8;
    constexpr int kLds = head_floats(kMaxQ);
    __shared__ __attribute__((aligned(16))) float lds[kLds + 4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int net = blockIdx.x % p.G;
    const int wg = blockIdx.x / p.G;
    const int nwg = gridDim.x / p.G;
    const int Q = p.Q;
    int* unit_counter = reinterpret_cast<int*>(&lds[kLds]);
    if (tid == 0) *unit_counter = WAVES;
    fill_lds<head_floats(kMaxQ) / 4, 64 * WAVES>(lds, p.packed[net], tid);   // buffers are sized for kMaxQ
    __syncthreads();
    const f16x8* HAS = reinterpret_cast<const f16x8*>(&lds[kHAS]);
    const f16x8* HA1 = reinterpret_cast<const f16x8*>(&lds[kHA1]);
    auto no_extra = [](int) {};
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
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
    for (int unit = u_begin + wave; unit < u_end; unit = grab()) {
        const int row = unit * 32 + (lane & 31);
        const bool valid = row < rows;
        const int rc = valid ? row : rows - 1;
        f32x16 accs[4];
        f16x8 ah[4], al[4];
        if constexpr (FROM_GATED) {
            f16x8 oh[4], ol[4];
            {
                float o[32];
                load_tiled<8, 64>(p.in[net], rc, h, true, o);
                split8<0>(o, oh[0], ol[0]);
                split8<8>(o, oh[1], ol[1]);
                split8<16>(o, oh[2], ol[2]);
                split8<24>(o, oh[3], ol[3]);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(&lds[kHBS + h * 64 + it * 16 + q * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accs[it][q * 4 + e] = bs[e];
                }
            first_frags<4, 4, 0, 1, 4>(HAS, lane, ah, al);
            gemm16<4, 4, 0, 1, 4>(HAS, lane, accs, ah, al, [&](int s) -> f16x8 { return oh[s]; },
                                  [&](int s) -> f16x8 { return ol[s]; }, no_extra,
                                  [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 4, 0, 1, 4>(HA1, lane, nh, nl); });
        } else {
            const float* srow = p.in[net] + tile_off(rc, h, 128);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(srow + (8 * it + 2 * q) * 128);
#pragma unroll
                    for (int e = 0; e < 4; ++e) accs[it][q * 4 + e] = v[e];
                }
            first_frags<8, 4, 0, 1, 4>(HA1, lane, ah, al);
        }
        // relu -> split -> post1 (128 -> 128)
        f16x8 sh[8], sl[8];
        {
            float r[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) r[i] = fmaxf(accs[i >> 4][i & 15], 0.f);
            split8<0>(r, sh[0], sl[0]);
            split8<8>(r, sh[1], sl[1]);
            split8<16>(r, sh[2], sl[2]);
            split8<24>(r, sh[3], sl[3]);
            split8<32>(r, sh[4], sl[4]);
            split8<40>(r, sh[5], sl[5]);
            split8<48>(r, sh[6], sl[6]);
            split8<56>(r, sh[7], sl[7]);
        }
        f32x16 acc1[4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(&lds[kHB1 + h * 64 + it * 16 + q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[it][q * 4 + e] = b1[e];
            }
        gemm16<8, 4, 0, 1, 4>(HA1, lane, acc1, ah, al, [&](int s) -> f16x8 { return sh[s]; },
                              [&](int s) -> f16x8 { return sl[s]; }, no_extra, [](f16x8(&)[4], f16x8(&)[4]) {});
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
// packing: TF layouts -> hi/lo fp16 A fragments (+ fp32 bias sections, same offsets as fp32 layout)
// --------------------------------------------------------------------------------------
__device__ __forceinline__ void put_split(f16x8* dst, int comp, const float (&w)[8]) {
    f16x8 v;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const _Float16 hi = (_Float16)w[q];
        v[q] = comp == 0 ? hi : (_Float16)(w[q] - (float)hi);
    }
    *dst = v;
}

// decode unit index within a [comp][NITTOT][NS][64] section
#define PWV_DECODE(u, NITTOT, NS)                                                     \
    const int lane = (u) & 63, s = ((u) >> 6) % (NS), it = ((u) >> 6) / (NS) % (NITTOT), \
              comp = ((u) >> 6) / ((NS) * (NITTOT));                                   \
    const int h = lane >> 5, i = lane & 31;                                          \
    (void)h; (void)i; (void)s; (void)it; (void)comp;

__global__ void pack_layer_f16_kernel(const float* filter, const float* gate, const float* dense, const float* dense_bias,
                                      const float* skip, const float* skip_bias, const float* gc_filter,
                                      const float* gc_gate, int with_skip, int cond_c, float* out, int total_units) {
    int u = blockIdx.x * blockDim.x + threadIdx.x;   // 16-byte units
    if (u >= total_units) return;
    f16x8* o16 = reinterpret_cast<f16x8*>(out);
    f32x4* o32 = reinterpret_cast<f32x4*>(out);
    float w[8];
    const int base = u;
    if (u < kA1Size / 4) {
        PWV_DECODE(u, 4, 8)
        const int tap = 1 - (s >> 2), oc = 32 * it + i;      // k-steps 0..3 = x[t] (tap 1), 4..7 = x[t-d] (tap 0): see the B operands
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cin = 16 * (s & 3) + 8 * (q >> 2) + 4 * h + (q & 3);
            w[q] = oc < 64 ? kFScale * filter[(tap * 64 + cin) * 64 + oc] : kGScale * gate[(tap * 64 + cin) * 64 + oc - 64];
        }
        put_split(&o16[base], comp, w);
        return;
    }
    u -= kA1Size / 4;
    if (u < kA2Size / 4) {
        PWV_DECODE(u, 2, 4)
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = dense[chan_of(s >> 1, 8 * (s & 1) + q, h) * 64 + 32 * it + i];
        put_split(&o16[base], comp, w);
        return;
    }
    u -= kA2Size / 4;
    if (u < kBDSize / 4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = u * 4 + e, hh = j >> 5, it = (j >> 4) & 1, r = j & 15;
            v[e] = dense_bias ? dense_bias[chan_of(it, r, hh)] : 0.f;
        }
        o32[base] = v;
        return;
    }
    u -= kBDSize / 4;
    if (with_skip) {
        if (u < kASSize / 4) {
            PWV_DECODE(u, 4, 4)
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = skip[chan_of(s >> 1, 8 * (s & 1) + q, h) * 128 + 32 * it + i];
            put_split(&o16[base], comp, w);
            return;
        }
        u -= kASSize / 4;
        if (u < kBSSize / 4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = u * 4 + e, hh = j >> 6, it = (j >> 4) & 3, r = j & 15;
                v[e] = skip_bias ? skip_bias[chan_of(it, r, hh)] : 0.f;
            }
            o32[base] = v;
            return;
        }
        u -= kBSSize / 4;
    }
    if (cond_c > 0) {
        PWV_DECODE(u, 4, 5)
        const int oc = 32 * it + i;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ci = 16 * s + 8 * (q >> 2) + 4 * h + (q & 3);
            w[q] = oc < 64 ? kFScale * gc_filter[ci * 64 + oc] : kGScale * gc_gate[ci * 64 + oc - 64];
        }
        put_split(&o16[base], comp, w);
    }
}

// pwv_pack_first_fold_f16x3: layer 0 of a scalar-input net, filter|gate folded onto the causal layer (modules.py:174-183 into
// :216-222).  h[t] = x[t-1] w0 + x[t] w1, so F|G(h[t-d], h[t]) = M [x[t-d-1], x[t-d], x[t-1], x[t]]^T with the [128, 4] matrix
// M[oc][2 tap + c] = sum_cin cf[c][cin] W[tap][cin][oc] (accumulated in fp64).  Output: the A fragments of ONE k-step,
// [hi | lo][4 row tiles][64 lanes] f16x8, k values 0..3 used (lanes with h = 0), the rest zero.
__global__ void pack_first_fold_f16_kernel(const float* __restrict__ cf, const float* __restrict__ filter, const float* __restrict__ gate,
                                           f16x8* __restrict__ out) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= 2 * 4 * 64) return;
    const int comp = u >> 8, it = (u >> 6) & 3, lane = u & 63, h = lane >> 5, i = lane & 31;
    const int oc = 32 * it + i;
    float w[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (h == 0) {
        for (int q = 0; q < 4; ++q) {
            const int tap = q >> 1, c = q & 1;
            double acc = 0.0;
            for (int cin = 0; cin < 64; ++cin) {
                const float wv = oc < 64 ? filter[(tap * 64 + cin) * 64 + oc] : gate[(tap * 64 + cin) * 64 + oc - 64];
                acc += (double)cf[c * 64 + cin] * (double)wv;
            }
            w[q] = (float)((double)(oc < 64 ? kFScale : kGScale) * acc);
        }
    }
    put_split(&out[u], comp, w);
}

__global__ void pack_head_f16_kernel(const float* skip, const float* skip_bias, const float* post1,
                                     const float* post1_bias, const float* post2, const float* post2_bias, int Q,
                                     float* out, int total_floats) {
    // units of 16 bytes up to kHW2, then plain floats (post2 / its bias)
    int u = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8* o16 = reinterpret_cast<f16x8*>(out);
    f32x4* o32 = reinterpret_cast<f32x4*>(out);
    const int base = u;
    float w[8];
    if (u < kASSize / 4) {
        PWV_DECODE(u, 4, 4)
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = skip ? skip[chan_of(s >> 1, 8 * (s & 1) + q, h) * 128 + 32 * it + i] : 0.f;
        put_split(&o16[base], comp, w);
        return;
    }
    u -= kASSize / 4;
    if (u < kBSSize / 4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = u * 4 + e, hh = j >> 6, it = (j >> 4) & 3, r = j & 15;
            v[e] = skip_bias ? skip_bias[chan_of(it, r, hh)] : 0.f;
        }
        o32[base] = v;
        return;
    }
    u -= kBSSize / 4;
    if (u < kHA1Size / 4) {
        PWV_DECODE(u, 4, 8)
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = post1[chan_of(s >> 1, 8 * (s & 1) + q, h) * 128 + 32 * it + i];
        put_split(&o16[base], comp, w);
        return;
    }
    u -= kHA1Size / 4;
    if (u < 128 / 4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = u * 4 + e, hh = j >> 6, it = (j >> 4) & 3, r = j & 15;
            v[e] = post1_bias ? post1_bias[chan_of(it, r, hh)] : 0.f;
        }
        o32[base] = v;
        return;
    }
    u -= 128 / 4;
    // post2 [2 h][Q][64] floats then bias (padded to 4)
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = u * 4 + e;
        float x = 0.f;
        if (j < 2 * Q * 64) {
            const int c = j & 63, hq = j >> 6, q = hq % Q, hh = hq / Q;
            x = post2[chan_of(c >> 4, c & 15, hh) * Q + q];
        } else if (j - 2 * Q * 64 < Q && post2_bias) {
            x = post2_bias[j - 2 * Q * 64];
        }
        v[e] = x;
    }
    if (base * 4 < total_floats) o32[base] = v;
}

template <bool SKIP, bool COND, bool GATED, bool FIRST = false, bool HEAD = false, bool FOLD = false>
static int launch16(const LayerParams& lp, int grid, hipStream_t s) {
    hipLaunchKernelGGL((layer_f16x3_kernel<SKIP, COND, GATED, FIRST, HEAD, FOLD>), dim3(grid), dim3(512), 0, s, lp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "f16x3 layer kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

int launch_layer_f16x3(const LayerParams& lp, bool skip, bool cond, bool gated, int per_net, hipStream_t s) {
    const int grid = per_net * lp.G;
    if (lp.packed_head[0]) {
        if (skip || cond || !gated || lp.x_first)
            return set_error(PWV_EINVAL, "fused head: plain last layer only (no skip accumulation, no per-sample condition, not layer 0)");
        return launch16<false, false, true, false, true>(lp, grid, s);
    }
    if (lp.x_first) {
        if (skip) return set_error(PWV_EINVAL, "x_first (layer 0 without a materialised causal layer) does not support skip accumulation");
        if (lp.fold0[0]) {      // layer 0 in its folded form
            if (cond) return gated ? launch16<false, true, true, true, false, true>(lp, grid, s) : launch16<false, true, false, true, false, true>(lp, grid, s);
            return gated ? launch16<false, false, true, true, false, true>(lp, grid, s) : launch16<false, false, false, true, false, true>(lp, grid, s);
        }
        if (cond) return gated ? launch16<false, true, true, true>(lp, grid, s) : launch16<false, true, false, true>(lp, grid, s);
        return gated ? launch16<false, false, true, true>(lp, grid, s) : launch16<false, false, false, true>(lp, grid, s);
    }
    if (skip) {
        if (cond) return gated ? launch16<true, true, true>(lp, grid, s) : launch16<true, true, false>(lp, grid, s);
        return gated ? launch16<true, false, true>(lp, grid, s) : launch16<true, false, false>(lp, grid, s);
    }
    if (cond) return gated ? launch16<false, true, true>(lp, grid, s) : launch16<false, true, false>(lp, grid, s);
    return gated ? launch16<false, false, true>(lp, grid, s) : launch16<false, false, false>(lp, grid, s);
}

int launch_head_f16x3(const HeadParams& hp, bool from_gated, int grid, hipStream_t s) {
    if (from_gated)
        hipLaunchKernelGGL((head_f16x3_kernel<true>), dim3(grid), dim3(512), 0, s, hp);
    else
        hipLaunchKernelGGL((head_f16x3_kernel<false>), dim3(grid), dim3(512), 0, s, hp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "f16x3 head kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

int launch_pack_layer_f16x3(const float* filter, const float* gate, const float* dense, const float* dense_bias,
                            const float* skip, const float* skip_bias, const float* gc_filter, const float* gc_gate,
                            int with_skip, int cond_c, float* out, hipStream_t s) {
    const int units = layer_floats(with_skip != 0, cond_c > 0) / 4;
    hipLaunchKernelGGL(pack_layer_f16_kernel, dim3((units + 255) / 256), dim3(256), 0, s, filter, gate, dense,
                       dense_bias, skip, skip_bias, gc_filter, gc_gate, with_skip, cond_c, out, units);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "f16x3 pack launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

int launch_pack_first_fold_f16x3(const float* cf, const float* filter, const float* gate, float* out, hipStream_t s) {
    hipLaunchKernelGGL(pack_first_fold_f16_kernel, dim3(2), dim3(256), 0, s, cf, filter, gate, reinterpret_cast<f16x8*>(out));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "first-fold pack launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

int launch_pack_head_f16x3(const float* skip, const float* skip_bias, const float* post1, const float* post1_bias,
                           const float* post2, const float* post2_bias, int Q, float* out, hipStream_t s) {
    const int total = head_floats(Q);
    const int units = (total + 3) / 4;
    hipLaunchKernelGGL(pack_head_f16_kernel, dim3((units + 255) / 256), dim3(256), 0, s, skip, skip_bias, post1,
                       post1_bias, post2, post2_bias, Q, out, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "f16x3 pack launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

}  // namespace pwv
