// fp16 mode ("f16") of the fused layer / head kernels for gfx950: the BUILD EXTENSION named by BASELINE.json
// config 5 ("fp16"); the reference is fp32 only (models.py:81-82), so this mode has no reference parity --
// its tolerance is stated against the fp64 oracle (tests/test_gpu_f16.py) and it is never the benchmark default.
//
// Same math and launch structure as pwv_layer_f16.hip (reference call sites modules.py:185-259, :145-165) but
//   * the residual stream lives in HBM as fp16 (128 B per sample instead of 256), tile32-style: block u = rows
//     32u..32u+31 as [8 chunks][32 rows][8 halfs], chunk s*2 + h holding channels 16s + 8(q>>2) + 4h + (q&3),
//     q < 8 -- exactly the eight values lane (t, h) feeds to the matrix pipe for k-step s, so ONE 16-byte load
//     per k-step goes from HBM into the B operand untouched, and each wave-level load is 1 KB contiguous;
//   * one fp16 product per term (weights = the `hi` halves of the split-fp16 packed buffers, read in place),
//     fp32 accumulation: 40 MFMAs per 32-sample unit instead of 120;
//   * ~41 KB of LDS per workgroup (4 waves), several workgroups per CU.
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

// only the `hi` component of each split-fp16 section is staged (fill_lds_dma on the first half of the section)

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

// head (modules.py:145-165) LDS map, in 16-byte units behind the layer's: skip hi [4][4][64] | postprocess1 hi [4][8][64] | floats
constexpr int kHH_AS = 0;
constexpr int kHH_A1 = kHH_AS + 4 * 4 * 64;
constexpr int kHH_END = kHH_A1 + 4 * 8 * 64;     // 3072 units = 49,152 B
constexpr int kHH_FLOATS = 128 + 128 + 2 * kMaxQ * 64 + 4;   // skip bias, post1 bias, post2 weights, post2 bias

// the head on a lane's gated output o (fp16 fragments oh[4], the B operand of the skip GEMM) -> Q fp32 outputs of row `row`
__device__ __forceinline__ void head_h16_unit(const f16x8* hl, const float* fl, const f16x8 (&ob)[4], int lane, int h, int Q, bool valid,
                                              float* out, int row) {
    const float* bs = fl;
    const float* b1 = fl + 128;
    const float* w2 = fl + 256;
    f32x16 accs[4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[it][r] = bs[h * 64 + it * 16 + r];
    gemm_h<4, 4>(&hl[kHH_AS], lane, accs, [&](int s) -> f16x8 { return ob[s]; });
    f16x8 sb[8];
    {
        float r[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) r[i] = fmaxf(accs[i >> 4][i & 15], 0.f);
        sb[0] = to_h8<0>(r);
        sb[1] = to_h8<8>(r);
        sb[2] = to_h8<16>(r);
        sb[3] = to_h8<24>(r);
        sb[4] = to_h8<32>(r);
        sb[5] = to_h8<40>(r);
        sb[6] = to_h8<48>(r);
        sb[7] = to_h8<56>(r);
    }
    f32x16 acc1[4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[it][r] = b1[h * 64 + it * 16 + r];
    gemm_h<8, 4>(&hl[kHH_A1], lane, acc1, [&](int s) -> f16x8 { return sb[s]; });
    for (int q = 0; q < Q; ++q) {
        float part = 0.f;
        const float* w = &w2[(h * Q + q) * 64];
#pragma unroll
        for (int i = 0; i < 64; ++i) part = fmaf(fmaxf(acc1[i >> 4][i & 15], 0.f), w[i], part);
        part += __shfl_xor(part, 32);
        part += w2[2 * Q * 64 + q];
        if (valid && h == 0) out[(size_t)row * Q + q] = part;
    }
}

__device__ __forceinline__ void stage_head_h16(f16x8* hl, float* fl, const float* packed, int Q, int wave, int lane, int tid, int waves) {
    if (waves == 8) {
        fill_lds_dma<4 * 4 * 64, 8>(reinterpret_cast<float*>(&hl[kHH_AS]), packed + kHAS, wave, lane);
        fill_lds_dma<4 * 8 * 64, 8>(reinterpret_cast<float*>(&hl[kHH_A1]), packed + kHA1, wave, lane);
    } else {
        fill_lds_dma<4 * 4 * 64, 4>(reinterpret_cast<float*>(&hl[kHH_AS]), packed + kHAS, wave, lane);
        fill_lds_dma<4 * 8 * 64, 4>(reinterpret_cast<float*>(&hl[kHH_A1]), packed + kHA1, wave, lane);
    }
    if (tid < 128) {
        fl[tid] = packed[kHBS + tid];
        fl[128 + tid] = packed[kHB1 + tid];
    }
    for (int i = tid; i < 2 * Q * 64 + 4; i += waves * 64) fl[256 + i] = packed[kHW2 + i];
}

#ifndef PWV_H16_MINWAVES
#define PWV_H16_MINWAVES 2
#endif
// FIRST: layer 0 of a scalar-input net rebuilds the causal layer's fp16 rows from four scalars per row with the operations of
// iaf_front_h16_kernel (fp32 fma, then ONE rounding to fp16): no front launch, no [rows, 64] fp16 buffer written and read twice.
// HEAD: the LAST layer with the head behind it -- the gated output's fp16 fragments are the B operand of the skip GEMM, exactly
// the bits head_h16_kernel would have read back from HBM.  103 KB of LDS with a per-sample condition: one 8-wave workgroup per CU.
// FOLD (with FIRST): layer 0's filter|gate convolution on the four scalars themselves (the `hi` fragments of
//   pwv_pack_first_fold_f16x3; see layer_f16x3_kernel): one MFMA k-step instead of eight.
template <bool COND, bool GATED, bool FIRST = false, bool HEAD = false, bool FOLD = false>
__global__ __launch_bounds__(HEAD ? 512 : 256, HEAD ? 1 : PWV_H16_MINWAVES) void layer_h16_kernel(const LayerParams p) {
    static_assert(!FOLD || FIRST, "FOLD: layer 0 of a scalar-input net only");
    static_assert(!HEAD || GATED, "HEAD: the last layer only");
    constexpr int WAVES = HEAD ? 8 : 4;
    constexpr int kUnits = COND ? kH_END : kH_AC;
    constexpr int kCF = kUnits + 64 / 4 + 1;                        // FIRST: causal filter [2][64] floats = 32 units
    constexpr int kHD = kCF + (FIRST ? 32 : 0);                     // HEAD: head weights + floats
    __shared__ __attribute__((aligned(16))) f16x8 lds[kHD + (HEAD ? kHH_END + (kHH_FLOATS + 3) / 4 : 0)];   // + BD (64 floats) + counter
    float* bd_lds = reinterpret_cast<float*>(&lds[kUnits]);
    int* unit_counter = reinterpret_cast<int*>(&lds[kUnits + 16]);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int net = blockIdx.x % p.G;
    const int wg = blockIdx.x / p.G;
    const int nwg = gridDim.x / p.G;
    const float* packed = p.packed[net];
    // the split-fp16 packed layout keeps [hi | lo] per section; only the hi halves are staged
    fill_lds_dma<4 * 8 * 64, WAVES>(reinterpret_cast<float*>(&lds[kH_A1]), packed + kA1, wave, lane);
    fill_lds_dma<2 * 4 * 64, WAVES>(reinterpret_cast<float*>(&lds[kH_A2]), packed + kA2, wave, lane);
    if constexpr (COND) fill_lds_dma<4 * 5 * 64, WAVES>(reinterpret_cast<float*>(&lds[kH_AC]), packed + kLayerBase, wave, lane);
    if (tid < 64) bd_lds[tid] = packed[kBD + tid];
    if (tid == 0) *unit_counter = WAVES;
    const float* cf = reinterpret_cast<const float*>(&lds[kCF]);
    if constexpr (FIRST) {
        if (tid < 128) reinterpret_cast<float*>(&lds[kCF])[tid] = p.cfilt[net][tid];
    }
    if constexpr (HEAD) stage_head_h16(&lds[kHD], reinterpret_cast<float*>(&lds[kHD + kHH_END]), p.packed_head[net], p.head_q, wave, lane, tid, WAVES);
    __syncthreads();

    const _Float16* xin = reinterpret_cast<const _Float16*>(p.x_in[net]);
    _Float16* xout = reinterpret_cast<_Float16*>(p.x_out[net]);
    const _Float16* cond = reinterpret_cast<const _Float16*>(p.cond);
    const int rows = p.N * p.T;
    const int units = (rows + 31) / 32;
    const int per_wg = (units + nwg - 1) / nwg;
    const int u_begin = wg * per_wg < units ? wg * per_wg : units;
    const int u_end = (u_begin + per_wg < units) ? u_begin + per_wg : units;
    auto grab = [&]() -> int {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(unit_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return u_begin + __builtin_amdgcn_readfirstlane(v);
    };

    const __amdgpu_buffer_rsrc_t out_rs = units_rsrc(xout, u_begin, u_end, 32 * 64 * 2);      // write-through stores, see pwv_layer_common.h
    for (int unit = u_begin + wave; unit < u_end; unit = grab()) {
        int row, rc, n, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        const bool has_prev = t >= p.dilation;
        const _Float16* xr = xin + xoff(rc, h, 64);
        const _Float16* xp = xin + xoff(has_prev ? rc - p.dilation : rc, h, 64);
        f16x8 b[8];          // k-steps 0..3 = x[t-d], 4..7 = x[t]: straight from HBM into the B operand
        f16x8 fold_b = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)fold_b;
        if constexpr (FIRST) {
            // the four scalars the two rows are functions of (zero left of the utterance start), then per k-step the lane's eight
            // channels 16s + 8(q>>2) + 4h + (q&3): round(x[t-1] w0), fma(x[t], w1, .), one rounding to fp16 (iaf_front_h16_kernel)
            const float* x1 = p.x_first;
            const int d = p.dilation;
            const float x0 = x1[rc], xm1 = t >= 1 ? x1[rc - (t >= 1 ? 1 : 0)] : 0.f;
            const float xd0 = has_prev ? x1[rc - (has_prev ? d : 0)] : 0.f, xd1 = t >= d + 1 ? x1[rc - (t >= d + 1 ? d + 1 : 0)] : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float vc[8], vb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = 16 * s + 8 * (q >> 2) + 4 * h + (q & 3);
                    vc[q] = fmaf(x0, cf[64 + c], xm1 * cf[c]);
                    vb[q] = fmaf(xd0, cf[64 + c], xd1 * cf[c]);
                }
                const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                b[4 + s] = to_h8<0>(vc);      // (x[t]: the residual add needs it in every form)
                b[s] = (has_prev && !FOLD) ? to_h8<0>(vb) : zero;
            }
            if constexpr (FOLD) {             // k = 0..3 of ONE k-step, lanes of the lower half: x[t-d-1], x[t-d], x[t-1], x[t]
                const float sc[4] = {xd1, xd0, xm1, x0};
#pragma unroll
                for (int q = 0; q < 4; ++q) fold_b[q] = (_Float16)(h == 0 ? sc[q] : 0.f);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                b[4 + s] = *reinterpret_cast<const f16x8*>(xr + s * 512);
                const f16x8 v = *reinterpret_cast<const f16x8*>(xp + s * 512);
                const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                b[s] = has_prev ? v : zero;
            }
        }
        f16x8 cb[5];
        if constexpr (COND) {
#pragma unroll
            for (int s = 0; s < 5; ++s) cb[s] = *reinterpret_cast<const f16x8*>(cond + xoff(rc, h, kCondC) + s * 512);
        }
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
        if constexpr (COND) gemm_h<5, 4>(&lds[kH_AC], lane, acc, [&](int s) -> f16x8 { return cb[s]; });
        if constexpr (FOLD) {
            const f16x8* F0 = reinterpret_cast<const f16x8*>(p.fold0[net]);
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F0[it * 64 + lane], fold_b, acc[it], 0, 0, 0);
        } else {
            gemm_h<8, 4>(&lds[kH_A1], lane, acc, [&](int s) -> f16x8 { return b[s ^ 4]; });      // (packed K order: x[t] first, see pack_layer_f16_kernel)
        }

        float o[32];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[tl * 16 + r] = gate_act(acc[tl][r], acc[tl + 2][r]);
        f16x8 oh[4];
        oh[0] = to_h8<0>(o);
        oh[1] = to_h8<8>(o);
        oh[2] = to_h8<16>(o);
        oh[3] = to_h8<24>(o);

        const int ooff = (int)((xoff(row, h, 64) - (size_t)u_begin * (32 * 64)) * 2);      // bytes from this workgroup's first unit
        if constexpr (HEAD) {
            head_h16_unit(&lds[kHD], reinterpret_cast<const float*>(&lds[kHD + kHH_END]), oh, lane, h, p.head_q, valid, p.head_out[net], row);
        } else if constexpr (GATED) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < 4; ++s) store_wt(out_rs, ooff + s * 1024, oh[s]);
            }
        } else {
            // dense 64 -> 64; accumulator starts at x[t] + dense_bias (register r of tile it <-> half q = r&7 of chunk 2it + (r>>3))
            f32x16 acc2[2];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc2[it][r] = (float)b[4 + 2 * it + (r >> 3)][r & 7] + bd_lds[h * 32 + it * 16 + r];
            gemm_h<4, 2>(&lds[kH_A2], lane, acc2, [&](int s) -> f16x8 { return oh[s]; });
            if (valid) {
                float y[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) y[i] = acc2[i >> 4][i & 15];
                store_wt(out_rs, ooff + 0 * 1024, to_h8<0>(y));
                store_wt(out_rs, ooff + 1 * 1024, to_h8<8>(y));
                store_wt(out_rs, ooff + 2 * 1024, to_h8<16>(y));
                store_wt(out_rs, ooff + 3 * 1024, to_h8<24>(y));
            }
        }
    }
}

// ---- head: o (fp16, permuted) -> skip -> relu -> post1 -> relu -> post2 (fp32 out) ----------------------

__global__ __launch_bounds__(256) void head_h16_kernel(const HeadParams p) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[kHH_END + (kHH_FLOATS + 3) / 4];
    float* fl = reinterpret_cast<float*>(&lds[kHH_END]);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int net = blockIdx.x % p.G;
    const int wg = blockIdx.x / p.G;
    const int nwg = gridDim.x / p.G;
    stage_head_h16(lds, fl, p.packed[net], p.Q, wave, lane, tid, 4);
    __syncthreads();
    const _Float16* in = reinterpret_cast<const _Float16*>(p.in[net]);
    const int rows = p.N * p.T;
    const int ntiles = (rows + 127) / 128;
    for (int tile = wg; tile < ntiles; tile += nwg) {
        const int row = tile * 128 + wave * 32 + (lane & 31);
        const bool valid = row < rows;
        const int rc = valid ? row : rows - 1;
        f16x8 ob[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ob[s] = *reinterpret_cast<const f16x8*>(in + xoff(rc, h, 64) + s * 512);
        head_h16_unit(lds, fl, ob, lane, h, p.Q, valid, p.out[net], row);
    }
}

// ---- IAF affine + causal layer with fp16 (permuted) output rows: modules.py:59, :179-180 ----------------------
struct FrontH16Params {
    const float* z;
    const float* s;
    const float* b;
    float* x_out;
    const float* filt[PWV_MAX_NETS];
    _Float16* hrow[PWV_MAX_NETS];
    int sb_stride, G, N, T, W;
};

constexpr int kFrontH16MaxTaps = 8;

// one thread per row (blockIdx.y = net), filter [W][64] in LDS (broadcast reads), 8 chunk stores of 16 bytes
__global__ __launch_bounds__(256) void iaf_front_h16_kernel(const FrontH16Params p) {
    __shared__ __attribute__((aligned(16))) float fl[kFrontH16MaxTaps * 64];
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < p.W * 64; i += 256) fl[i] = p.filt[g][i];
    __syncthreads();
    const unsigned rows = (unsigned)p.N * (unsigned)p.T;
    const unsigned row = blockIdx.x * 256u + threadIdx.x;
    if (row >= rows) return;
    const unsigned t = row % (unsigned)p.T;
    auto xval = [&](unsigned rr) -> float {
        const float zv = p.z[rr];
        return p.s ? fmaf(zv, p.s[(size_t)rr * p.sb_stride], p.b[(size_t)rr * p.sb_stride]) : zv;
    };
    const float xcur = xval(row);
    float xv[kFrontH16MaxTaps];
#pragma unroll
    for (int k = 0; k < kFrontH16MaxTaps; ++k) {
        const unsigned shift = (unsigned)(p.W - 1 - k);
        xv[k] = k + 1 == p.W ? xcur : ((k < p.W && shift <= t) ? xval(row - shift) : 0.f);
    }
    if (g == 0 && p.x_out) p.x_out[row] = xcur;
    _Float16* out = p.hrow[g] + xoff((int)row, 0, 64);
#pragma unroll
    for (int chunk = 0; chunk < 8; ++chunk) {
        const int s = chunk >> 1, h = chunk & 1;      // halves q <-> channels 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3}
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < kFrontH16MaxTaps; ++k) {
            if (k < p.W) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(&fl[k * 64 + 16 * s + 4 * h]);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(&fl[k * 64 + 16 * s + 8 + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = fmaf(xv[k], w0[e], acc[e]);
                    acc[4 + e] = fmaf(xv[k], w1[e], acc[4 + e]);
                }
            }
        }
        *reinterpret_cast<f16x8*>(out + chunk * 256) = to_h8<0>(acc);
    }
}

// [N,T,80] fp32 per-sample condition -> fp16 tile32 blocks in the B-operand order (5 k-steps); SPLIT adds the
// `lo` plane fp16(v - fp16(v)) behind the `hi` plane (plane_halfs apart) for the split-fp16 layer kernel, which
// then reads both operands of its condition GEMM straight from HBM instead of splitting 40 floats per unit
template <bool SPLIT>
__global__ void cond_to_h16_kernel(const float* __restrict__ cond, _Float16* __restrict__ out, unsigned rows, size_t plane_halfs) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;     // (block, chunk 0..9, row in block)
    const unsigned row = (idx / 320u) * 32u + (idx & 31u), chunk = (idx >> 5) % 10u;
    if (row >= rows) return;
    const int s = chunk >> 1, h = chunk & 1;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = cond[(size_t)row * kCondC + 16 * s + 8 * (q >> 2) + 4 * h + (q & 3)];
    const f16x8 hi = to_h8<0>(v);
    *reinterpret_cast<f16x8*>(out + xoff((int)row, (int)chunk, kCondC)) = hi;
    if constexpr (SPLIT) {
        float r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = v[q] - (float)hi[q];
        *reinterpret_cast<f16x8*>(out + plane_halfs + xoff((int)row, (int)chunk, kCondC)) = to_h8<0>(r);
    }
}

template <bool COND, bool GATED, bool FIRST, bool HEAD, bool FOLD = false>
static void launch_h16(const LayerParams& lp, int grid, hipStream_t s) {
    hipLaunchKernelGGL((layer_h16_kernel<COND, GATED, FIRST, HEAD, FOLD>), dim3(grid), dim3(HEAD ? 512 : 256), 0, s, lp);
}

int launch_layer_h16(const LayerParams& lp, bool cond, bool gated, int per_net, hipStream_t s) {
    const int grid = per_net * lp.G;
    const bool first = lp.x_first != nullptr, head = lp.packed_head[0] != nullptr;
    if (head && (!gated || first)) return set_error(PWV_EINVAL, "fp16 fused head: the last layer only (out_mode PWV_OUT_GATED, not layer 0)");
    if (head) {
        if (cond) launch_h16<true, true, false, true>(lp, grid, s);
        else launch_h16<false, true, false, true>(lp, grid, s);
    } else if (first && lp.fold0[0]) {
        if (cond) { if (gated) launch_h16<true, true, true, false, true>(lp, grid, s); else launch_h16<true, false, true, false, true>(lp, grid, s); }
        else { if (gated) launch_h16<false, true, true, false, true>(lp, grid, s); else launch_h16<false, false, true, false, true>(lp, grid, s); }
    } else if (first) {
        if (cond) { if (gated) launch_h16<true, true, true, false>(lp, grid, s); else launch_h16<true, false, true, false>(lp, grid, s); }
        else { if (gated) launch_h16<false, true, true, false>(lp, grid, s); else launch_h16<false, false, true, false>(lp, grid, s); }
    } else if (cond) {
        if (gated) launch_h16<true, true, false, false>(lp, grid, s);
        else launch_h16<true, false, false, false>(lp, grid, s);
    } else {
        if (gated) launch_h16<false, true, false, false>(lp, grid, s);
        else launch_h16<false, false, false, false>(lp, grid, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "fp16 layer kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

int launch_head_h16(const HeadParams& hp, int grid, hipStream_t s) {
    hipLaunchKernelGGL(head_h16_kernel, dim3(grid), dim3(256), 0, s, hp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "fp16 head kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

}  // namespace pwv

using namespace pwv;

extern "C" {

int pwv_iaf_front_f16(const float* z, const float* s, const float* b, int sb_stride, float* x_out, int G,
                      const float* const* filt, void* const* h16, int N, int T, int W, int R, pwv_stream_t stream) {
    PWV_CHECK_ARG(z && filt && h16, "pwv_iaf_front_f16: NULL pointer");
    PWV_CHECK_ARG((s == nullptr) == (b == nullptr), "pwv_iaf_front_f16: s and b must both be set or both NULL");
    PWV_CHECK_ARG(G >= 1 && G <= PWV_MAX_NETS && R == 64, "pwv_iaf_front_f16: G in [1,2] and R == 64 required");
    PWV_CHECK_ARG(N >= 1 && T >= 1 && W >= 1, "pwv_iaf_front_f16: bad N/T/W");
    PWV_CHECK_ARG((long long)N * T < (1ll << 31) - 256, "pwv_iaf_front_f16: N*T too large");
    PWV_CHECK_ARG(W <= kFrontH16MaxTaps, "pwv_iaf_front_f16: W must be <= %d", kFrontH16MaxTaps);
    FrontH16Params p{};
    p.z = z;
    p.s = s;
    p.b = b;
    p.x_out = x_out;
    p.sb_stride = sb_stride;
    p.G = G;
    p.N = N;
    p.T = T;
    p.W = W;
    for (int g = 0; g < G; ++g) {
        PWV_CHECK_ARG(filt[g] && h16[g], "pwv_iaf_front_f16: NULL filter / output for net %d", g);
        p.filt[g] = filt[g];
        p.hrow[g] = reinterpret_cast<_Float16*>(h16[g]);
    }
    hipLaunchKernelGGL(iaf_front_h16_kernel, dim3((unsigned)(((long long)N * T + 255) / 256), G), dim3(256), 0, (hipStream_t)stream, p);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_cond_to_f16(const float* cond, void* out16, int N, int T, int C, pwv_stream_t stream) {
    PWV_CHECK_ARG(cond && out16 && C == kCondC, "pwv_cond_to_f16: NULL pointer or C != %d", kCondC);
    const long long total = ((long long)N * T + 31) / 32 * 32 * 10;
    PWV_CHECK_ARG(N >= 0 && T >= 0 && total < (1ll << 31), "pwv_cond_to_f16: bad size");
    if (total == 0) return PWV_OK;
    hipLaunchKernelGGL(cond_to_h16_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cond,
                       reinterpret_cast<_Float16*>(out16), (unsigned)(N * T), (size_t)0);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_cond_split_f16(const float* cond, void* out16, int N, int T, int C, pwv_stream_t stream) {
    PWV_CHECK_ARG(cond && out16 && C == kCondC, "pwv_cond_split_f16: NULL pointer or C != %d", kCondC);
    const long long total = ((long long)N * T + 31) / 32 * 32 * 10;
    PWV_CHECK_ARG(N >= 0 && T >= 0 && total < (1ll << 31), "pwv_cond_split_f16: bad size");
    if (total == 0) return PWV_OK;
    hipLaunchKernelGGL(cond_to_h16_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cond,
                       reinterpret_cast<_Float16*>(out16), (unsigned)(N * T), tile32_floats((long long)N * T, kCondC));
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

}  // extern "C"
