// Layouts, kernel parameter blocks and device helpers shared by the fp32 (pwv_layer.hip) and
// split-fp16 (pwv_layer_f16.hip) variants of the fused layer / head kernels.
#pragma once

#include "pwv_common.h"

namespace pwv {

// ---- packed layer layout (floats) ----------------------------------------------------
constexpr int kA1 = 0;                    // [4 it][16 ks4][64 lane][4]   filter‖gate, K = 128
constexpr int kA1Size = 4 * 16 * 64 * 4;  // 16384
constexpr int kA2 = kA1 + kA1Size;        // [2 it][8 ks4][64][4]         dense, K = 64
constexpr int kA2Size = 2 * 8 * 64 * 4;   // 4096
constexpr int kBD = kA2 + kA2Size;        // [2 h][32]                    dense bias (D layout)
constexpr int kBDSize = 64;
constexpr int kLayerBase = kBD + kBDSize;  // 20544
constexpr int kASSize = 4 * 8 * 64 * 4;    // 8192   skip, K = 64, 128 outputs
constexpr int kBSSize = 128;               // [2 h][64]
constexpr int kCondC = 80;                 // per-sample conditioning channels supported
constexpr int kACSize = 4 * (kCondC / 8) * 64 * 4;  // 10240

constexpr int layer_floats(bool skip, bool cond) {
    return kLayerBase + (skip ? kASSize + kBSSize : 0) + (cond ? kACSize : 0);
}

// ---- packed head layout ----------------------------------------------------------------
constexpr int kHAS = 0;                       // skip weights (as above)
constexpr int kHBS = kHAS + kASSize;          // skip bias
constexpr int kHA1 = kHBS + kBSSize;          // post1 [4 it][16 ks4][64][4]
constexpr int kHA1Size = 4 * 16 * 64 * 4;
constexpr int kHB1 = kHA1 + kHA1Size;         // post1 bias [2 h][64]
constexpr int kHW2 = kHB1 + 128;              // post2 [2 h][Q][64], then bias [Q] (padded to 4)
constexpr int kMaxQ = 4;
constexpr int head_floats(int Q) { return kHW2 + 2 * Q * 64 + 4; }

struct LayerParams {
    const float* x_in[PWV_MAX_NETS];
    float* x_out[PWV_MAX_NETS];
    const float* packed[PWV_MAX_NETS];
    const float* proj[PWV_MAX_NETS];
    float* skip[PWV_MAX_NETS];
    const float* cond;
    int proj_row_stride;
    int G, N, T, dilation;
    int cond_hop, cond_offset, cond_frames;
    int skip_init;
    unsigned T_magic, T_shift;      // n / T  == umulhi(n, T_magic) >> T_shift   (T_magic == 0: T == 1)
    unsigned hop_magic, hop_shift;  // same for cond_hop
    long long* trace;   // debug builds only (-DPWV_TRACE): per-wave phase timestamps
    const float* x_first;                   // layer 0 without a materialised causal layer: the scalar input [rows] ...
    const float* cfilt[PWV_MAX_NETS];       // ... and each net's causal filter [2,1,64] (split-fp16 kernel only)
    const float* fold0[PWV_MAX_NETS];       // ... optional (split-fp16 kernel): layer 0's folded fragments (pwv_pack_first_fold_f16x3)
    const float* packed_head[PWV_MAX_NETS]; // last layer with the head fused behind it (split-fp16 kernel only)
    float* head_out[PWV_MAX_NETS];
    int head_q;
    float x_limit;                          // range guard of the split-fp16 arithmetic (x_first path): |x| <= x_limit ...
    int* range_flag;                        // ... else *range_flag = 1 (pinned host memory); NULL = no check
};

// exact n / d for n < 2^31 (Granlund-Montgomery): l = ceil(log2 d), magic = ceil(2^(31+l) / d), shift = l - 1
inline void make_magic(unsigned d, unsigned& magic, unsigned& shift) {
    if (d <= 1) { magic = 0; shift = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned long long k = 31 + l;
    magic = (unsigned)(((1ull << k) + d - 1) / d);
    shift = l - 1;
}
__device__ __forceinline__ int fast_div(int n, unsigned magic, unsigned shift) {
    return magic ? (int)(__umulhi((unsigned)n, magic) >> shift) : n;
}

// (utterance, time) of the 32 rows of a unit: one scalar division per unit when T >= 32
__device__ __forceinline__ void unit_rows(int unit, int lane, int rows, int N, int T, unsigned T_magic, unsigned T_shift,
                                          int& row, bool& valid, int& rc, int& n, int& t) {
    row = unit * 32 + (lane & 31);
    valid = row < rows;
    rc = valid ? row : rows - 1;
    if (T >= 32) {
        const int row0 = unit * 32;                       // wave-uniform: scalar mul_hi
        const int n0 = fast_div(row0, T_magic, T_shift);
        t = row0 - n0 * T + (lane & 31);
        n = n0;
        if (t >= T) { t -= T; n += 1; }
        if (!valid) { n = N - 1; t = T - 1; }
    } else {
        n = fast_div(rc, T_magic, T_shift);
        t = rc - n * T;
    }
}

// -DPWV_TRACE: waves of workgroup 0 record s_memtime at phase boundaries (tools/trace_layer.py)
#ifdef PWV_TRACE
#define PWV_STAMP(slot)                                                                   \
    do {                                                                                  \
        if (p.trace && blockIdx.x < 2 && lane == 0 && tr_unit < 8)                        \
            p.trace[((blockIdx.x * 8 + wave) * 8 + tr_unit) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PWV_STAMP(slot) do {} while (0)
#endif

struct HeadParams {
    const float* in[PWV_MAX_NETS];
    const float* packed[PWV_MAX_NETS];
    float* out[PWV_MAX_NETS];
    int G, N, T, Q;
};

// The packed filter / gate weights, the per-sample condition weights and the projection P are
// pre-multiplied by -2*log2(e) / -log2(e), so the GEMM directly yields the exp2 arguments
//     Fs = -2*log2(e) * F,   Gs = -log2(e) * G
// and tanh(F)*sigmoid(G) = (1 - 2^Fs) / ((1 + 2^Fs)(1 + 2^Gs)): two v_exp, one v_rcp, no scale
// multiplies.  Only the upper side needs a clamp (2^57.7 squared stays finite; 2^-inf = 0 is fine):
// tanh saturates to +-1 in fp32 beyond |F| = 20 (Fs = -+57.7), sigmoid(-40) = 4e-18.
constexpr float kFScale = -2.8853900817779268f;
constexpr float kGScale = -1.4426950408889634f;
__device__ __forceinline__ float gate_act(float fs, float gs) {
#pragma clang fp contract(off)
    // upper clamp as ONE v_med3_f32 each (fminf would get a canonicalising v_max in front of it).  It must be an
    // instruction the compiler knows: fs / gs are MFMA results, and the hazard recogniser does not insert the
    // MFMA-write -> VALU-read wait states in front of inline asm (an asm v_min here read a stale accumulator
    // register whenever the scheduler placed it right behind the last MFMA).
    fs = __builtin_amdgcn_fmed3f(fs, -3.0e38f, 57.7f);
    gs = __builtin_amdgcn_fmed3f(gs, -3.0e38f, 57.7f);
    // (1 + e1)(1 + e2) = t + e1 t with t = 1 + e2, written as ONE explicit fma and with contraction off: every
    // instantiation of every kernel then rounds the gate identically (the fused-head / fused-first-layer variants are
    // bit-identical to the separate launches, tests/test_gpu_parity.py) instead of leaving the choice to -ffp-contract
    const float e1 = __builtin_amdgcn_exp2f(fs);
    const float e2 = __builtin_amdgcn_exp2f(gs);
    const float t = 1.f + e2;
    return (1.f - e1) * __builtin_amdgcn_rcpf(__builtin_fmaf(e1, t, t));
}

// One layer's / head's packed weights -> LDS (packed order == LDS order).  All loads of a thread are
// issued before the first LDS write so the copy costs one memory round trip, not one per chunk.
template <int N4, int THREADS>
__device__ __forceinline__ void fill_lds(float* lds, const float* __restrict__ packed, int tid) {
    constexpr int ITERS = (N4 + THREADS - 1) / THREADS;
    const f32x4* src = reinterpret_cast<const f32x4*>(packed);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    f32x4 v[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int idx = tid + i * THREADS;
        v[i] = src[idx < N4 ? idx : N4 - 1];
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int idx = tid + i * THREADS;
        if (idx < N4) dst[idx] = v[i];
    }
}

// ---- "tile32" activation layout ---------------------------------------------------------------------
// Every [rows, C] activation buffer the fused kernels read or write (residual stream C = 64, skip sums
// C = 128, per-sample condition C = 80) is stored in blocks of 32 consecutive rows; block u holds rows
// 32u .. 32u+31 as [C/4 channel quads][32 rows][4 floats].  A wave owns 32 rows and lane (t, h) owns channel
// quads 2g + h, so each of its vector loads / stores covers 1 KB of CONTIGUOUS memory (32 rows x 16 B for
// h = 0, then the same for h = 1).  With plain channels-last rows the same instruction touched 32 B in each of
// 32 cache lines: measured 64 -> 50 us per layer launch at 160000 rows (HISTORY.md section 3).
// Buffers are padded to whole blocks: tile32_floats(rows, C).
__host__ __device__ inline size_t tile32_floats(long long rows, int C) { return (size_t)((rows + 31) / 32) * 32 * C; }
__device__ __forceinline__ size_t tile_off(int row, int quad, int C) {
    return (size_t)(row >> 5) * (32 * C) + quad * 128 + (row & 31) * 4;
}
// Residual-stream stores are WRITE-THROUGH (`sc1`): the line stays valid in the XCD's L2 for the next layer's reads, but
// it is no longer dirty, so the end-of-kernel release has nothing left to write back -- the ~40 MB a launch stores drain
// while it computes instead of between it and the dependent launch (measured: -2.6 % per step, HISTORY.md section 4, K1).  The cache
// policy bits of a store are only reachable through the buffer intrinsics; the descriptor covers exactly the workgroup's
// own units [u_begin, u_end), so out-of-range offsets are dropped by the hardware.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxWriteThrough = 16;      // sc1
// `unit_bytes` = bytes of one 32-row unit of the buffer (32 x C x element size)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t units_rsrc(const void* base, int u_begin, int u_end, int unit_bytes) {
    const unsigned long long a = (unsigned long long)base + (unsigned long long)u_begin * unit_bytes;
    const unsigned long long span = (unsigned long long)(u_end > u_begin ? u_end - u_begin : 0) * unit_bytes;
    // plain selects + readfirstlane: the descriptor must be provably wave-uniform or every access becomes a waterfall loop
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)(span > 0xffffffffull ? 0xffffffffull : span));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
// byte offset of (row, quad) of an fp32 tile32 buffer relative to the first unit of the descriptor
__device__ __forceinline__ int units_off(int row, int quad, int C, int u_begin) {
    return (((row >> 5) - u_begin) * (32 * C) + quad * 128 + (row & 31) * 4) * 4;
}
template <typename V16>      // any 16-byte vector (f32x4, f16x8)
__device__ __forceinline__ void store_wt(__amdgpu_buffer_rsrc_t r, int byte_off, V16 v) {
    static_assert(sizeof(V16) == 16, "store_wt moves 16 bytes");
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, kAuxWriteThrough);
}

// The same copy by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write pass): wave w moves the 1 KB chunks
// w, w + WAVES, ...; chunk c lands at lds + 256 c floats + 16 bytes x lane (the destination of an LDS-DMA is
// wave-uniform base + lane x size, so packed order == LDS order is exactly what it needs).  The transfers count on
// vmcnt; the __syncthreads() that follows waits for them and orders them for every reader.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N4, int WAVES>
__device__ __forceinline__ void fill_lds_dma(float* lds, const float* __restrict__ packed, int wave, int lane) {
    constexpr int CH = (N4 + 63) / 64;
#pragma unroll
    for (int i = 0; i < (CH + WAVES - 1) / WAVES; ++i) {
        const int c = wave + i * WAVES;
        if (c < CH && c * 64 + lane < N4)
            __builtin_amdgcn_global_load_lds((gptr_t)(packed + (size_t)(c * 64 + lane) * 4), (lptr_t)(lds + c * 256), 16, 0, 0);
    }
}

// lane (t,h) loads channel quads 2g + h, g < NCH, of row `row` (always a valid row: callers clamp);
// `keep == false` zeroes the result with v_cndmask instead of branching around the loads.
template <int NCH, int C>
__device__ __forceinline__ void load_tiled(const float* __restrict__ base, int row, int h, bool keep, float (&dst)[4 * NCH]) {
    const float* p0 = base + tile_off(row, h, C);
#pragma unroll
    for (int g = 0; g < NCH; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p0 + g * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * g + e] = keep ? v[e] : 0.f;
    }
}

// lane (t,h) loads its NCH 16-byte chunks (float offsets 8g + 4h) of one channels-last row.
// `row` is always a valid address (callers clamp); `keep == false` zeroes the result with
// v_cndmask instead of branching around the loads.
template <int NCH>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int h, bool keep, float (&dst)[4 * NCH]) {
#pragma unroll
    for (int g = 0; g < NCH; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * g + e] = keep ? v[e] : 0.f;
    }
}

template <int NCH>
__device__ __forceinline__ void load_contig(const float* __restrict__ p, float (&dst)[4 * NCH]) {
#pragma unroll
    for (int g = 0; g < NCH; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * g + e] = v[e];
    }
}


// ---- exact-fp32 MFMA building blocks (pwv_layer.hip, pwv_stack_persist.hip) ------------------------------------------
// lds fragment: 4 consecutive k-steps of row tile `it` for this lane (one ds_read_b128)
__device__ __forceinline__ f32x4 frag(const float* lds, int base, int it, int ngroups, int g, int lane) {
    return *reinterpret_cast<const f32x4*>(&lds[base + ((it * ngroups + g) * 64 + lane) * 4]);
}

// One GEMM as NG groups of (NIT row tiles x 4 k-steps) MFMAs.  The A fragments of group g+1
// are read while group g's MFMAs issue; sched_barrier(0) pins that order so the scheduler
// cannot hoist all ds_reads (which spills).  `a` enters holding group 0's fragments and leaves
// holding whatever `tail(a)` loaded during the last group (the next GEMM's group 0).
//   bval(ks): B register of k-step ks;  extra(g): VALU work to overlap with group g's MFMAs.
template <int NG, int NIT, int IT0, int ITSTEP, int NACC, typename BF, typename EF, typename TF>
__device__ __forceinline__ void gemm_groups(const float* lds, int base, int lane, f32x16 (&acc)[NACC], f32x4 (&a)[4],
                                            BF&& bval, EF&& extra, TF&& tail) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        f32x4 n[4] = {a[0], a[1], a[2], a[3]};
        if (g + 1 < NG) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) n[i] = frag(lds, base, IT0 + i * ITSTEP, NG, g + 1, lane);
        } else {
            tail(n);
        }
        __builtin_amdgcn_sched_barrier(0);   // reads first: a whole group of MFMAs covers their latency
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = bval(g * 4 + e);
#pragma unroll
            for (int i = 0; i < NIT; ++i)
                acc[IT0 + i * ITSTEP] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b, acc[IT0 + i * ITSTEP], 0, 0, 0);
        }
        extra(g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = n[i];
    }
}


// launchers of the split-fp16 variants (pwv_layer_f16.hip)
int launch_layer_f16x3(const LayerParams& lp, bool skip, bool cond, bool gated, int per_net, hipStream_t s);
int launch_head_f16x3(const HeadParams& hp, bool from_gated, int grid, hipStream_t s);
int launch_pack_layer_f16x3(const float* filter, const float* gate, const float* dense, const float* dense_bias,
                            const float* skip, const float* skip_bias, const float* gc_filter, const float* gc_gate,
                            int with_skip, int cond_c, float* out, hipStream_t s);
int launch_pack_first_fold_f16x3(const float* cf, const float* filter, const float* gate, float* out, hipStream_t s);
int launch_pack_head_f16x3(const float* skip, const float* skip_bias, const float* post1, const float* post1_bias,
                           const float* post2, const float* post2_bias, int Q, float* out, hipStream_t s);

// launchers of the fp16-storage variants (pwv_layer_h16.hip)
int launch_layer_h16(const LayerParams& lp, bool cond, bool gated, int per_net, hipStream_t s);
int launch_head_h16(const HeadParams& hp, int grid, hipStream_t s);

}  // namespace pwv
