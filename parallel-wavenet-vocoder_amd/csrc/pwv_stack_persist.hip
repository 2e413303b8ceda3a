// Persistent kernel for a run of consecutive gated-residual layers (modules.py:185-259, the loop at modules.py:138-143)
// of one or two nets -- ONE launch instead of one launch per layer, split-fp16 or exact-fp32 arithmetic.
//
// Why: a per-layer launch of layer_f16x3_kernel pays, besides its steady state, a weight-staging prologue (1.7 us), the
// first rows' latency (1.3 us), the spread of its workgroups' end times (~8 %: every workgroup waits for the slowest at
// EVERY layer) and the gap to the dependent launch (1.5 us): ~20 % of a 46 us launch (HISTORY.md section 4, K1p).
//
// Round 3 design -- STATIC ownership (the round-2 kernel dealt (layer, unit) tasks out of per-XCD pools with global
// atomics, recomputed the x[t-d] halo per XCD and was 3 % slower than the launches it replaced):
//
//  * Workgroup w of a net owns the units [u_begin, u_end) -- the SAME contiguous rows in every layer, what the per-layer
//    kernel gives it per launch -- and walks layer after layer over them.  Nothing is recomputed.  Its 8 waves take
//    (layer, unit) tasks from ONE LDS counter, layer-major, units in DESCENDING order; a wave that finishes its last unit
//    of layer j takes a unit of layer j+1 at once: no barrier between layers, inside the workgroup or across the grid.
//  * Dependencies are tracked, not assumed.  RAW: task (j, u) reads layer j-1's units u, u - ceil(d/32), (32u+31-d)>>5.
//    WAR: it overwrites ring slot j % 3, last read by layer j-2's tasks of the units u, u + floor(d'/32), u + ceil(d'/32).
//    Inside the own range: one byte per unit in LDS ("layers completed").  Across workgroups: one progress word per
//    workgroup in global memory ("layers completed for ALL my units"), needed only by the bottom units (left neighbour's
//    top rows) and the top units (right neighbour's reads of the slot being overwritten).  With descending order and a
//    ring of THREE buffers both cross-workgroup dependencies are a whole layer old when they are needed: a workgroup is
//    coupled to its neighbours only loosely, so the per-layer end-time noise averages out instead of adding up.
//  * Visibility (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16 R1): every x load
//    is `sc1` (L2-served, never the CU's L1, which other CUs' stores do not refresh); units a neighbour reads are stored
//    write-through (`sc1`); every wave drains `vmcnt(0)` before it publishes; the progress word is an agent-scope store
//    by the LAST wave to leave the layer (each wave counts itself out only after its drain).
//  * Weights: LDS holds layers j and j+1 (2 x 79 KB + 2 x 256 B of dense bias); the last wave to leave layer j refills
//    that slot with layer j+2 by LDS-DMA and announces it at its next drain.  The 2 KB that buys the control state come
//    from the LAST 1 KB fragment of each layer's dense matrix, which every unit reads from global memory (one 16-byte load
//    per lane, an L1 / L2 hit) instead.
//  * The launch leaves its workspace as it found it (zeros): the last workgroup to finish cleans up, so a caller that keeps the
//    workspace needs no zeroing kernel in front of the next launch (pwv_persist_args.workspace_clean).
//  * Every wait is bounded (20 ms, or until any wave of the launch has given up) and reports through a sticky status word in pinned host memory; the host then reruns on the
//    per-layer path.  All workgroups must be resident (grid <= CUs, one workgroup per CU by its LDS size).
// Results are bit-identical to the per-layer launches (same per-unit arithmetic): tests/test_gpu_persist.py.
#include "pwv_f16x3.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>

// cache policy of the stores of units no other workgroup reads (0 = plain: produced and consumed through one CU's L2)
#ifndef PWV_PERSIST_STORE_AUX
#define PWV_PERSIST_STORE_AUX 0
#endif

namespace pwv {

constexpr int kSlotFull = kA1Size + kA2Size;   // floats of a layer's matrices: filter|gate (hi+lo) + dense (hi+lo) = 81,920 B
constexpr int kSlot = kSlotFull - 256;         // held in LDS: everything but the last 1 KB fragment of the dense matrix
constexpr int kBiasF = 2 * kSlot;              // [2 slots][64] dense bias
constexpr int kCtlF = kBiasF + 128;            // control ints: [0] task counter, [1] abort, [2..3] flag bytes, [8 + j] waves that left layer j
constexpr int kCtlInts = 40;
constexpr int kLdsFloats = 40960;              // all 160 KB of the CU
constexpr int kCfF = kCtlF + kCtlInts;       // causal filter [2][64] (a run that starts with the net's layer 0, see x_first)
constexpr int kLeftN = 32;                                 // unit mode (short inputs): "layers completed" bytes of the 32 units LEFT of the range, as last seen
constexpr int kLeftB = (kCfF + 128) * 4;                   // ... in front of the own units' bytes, so one linear address serves both
constexpr int kDoneB = kLeftB + kLeftN;                    // byte offset of the per-unit "layers completed" bytes
constexpr int kMaxUnitsWg = kLdsFloats * 4 - kDoneB;       // 832 units per workgroup
#ifndef PWV_SHORT_MAX_UNITS
#define PWV_SHORT_MAX_UNITS 7      // (wave 7 is the loader.  Against the general kernel: 4 units per workgroup -17 %, 5: -8 %, 6: -11.5 %, 7: -11 %; profiles/r06_ab_experiments.md, r06_z2)
#endif
constexpr int kUnitModeMaxPerWg = PWV_SHORT_MAX_UNITS;
static_assert(kUnitModeMaxPerWg <= 7, "the short-input instantiation keeps wave 7 as its loader");
#ifndef PWV_MEDIUM_MAX_UNITS
#define PWV_MEDIUM_MAX_UNITS 0
#endif
constexpr int kMediumMaxPerWg = PWV_MEDIUM_MAX_UNITS;
constexpr int kMediumMode = (PWV_MEDIUM_MAX_UNITS) > 0 ? 1 : 0;      // (0: no such instantiation is built)
                      // unit mode up to this many units per workgroup and layer
constexpr int kUnitStride = 32;                            // ints between two units' words (own 128-byte lines: a poll asks for exactly the unit it waits for)
constexpr int kFlagB = (kCtlF + 2) * 4;        // flag bytes: +0 seenL, +1 seenR, +2 / +3 newest layer in LDS slot 0 / 1, +4 always 255
constexpr int kSeenLB = kFlagB, kSeenRB = kFlagB + 1, kWreadyB = kFlagB + 2, kTrueB = kFlagB + 4;
constexpr int kMaxPLayers = 32;
constexpr long long kWaitTicks = 2000000;      // a wave gives up after 20 ms of the chip-wide 100 MHz clock (normal waits: microseconds)
constexpr int kProgStride = 32;                // ints between two workgroups' progress words (own 128-byte lines)
constexpr int kMaxReachWgs = 60;               // neighbours polled by one wave instruction

struct PersistParams {
    float* ring[PWV_MAX_NETS];             // three full-size tile32 buffers, `ring_stride` floats apart; layer j reads buffer (j + 2 + rot) % 3, writes (j + rot) % 3
    long long ring_stride;
    const float* packed[PWV_MAX_NETS];     // packed layers of this launch, `packed_stride` floats apart
    const float* proj[PWV_MAX_NETS];       // P rows; this launch's first layer at column 0, layer j at 128 j
    int* prog;                             // [G][nwg] progress words, kProgStride ints apart, zeroed per launch
    int* uprog;                            // unit mode: [G][units] "layers completed" per UNIT (the left neighbours' top units are what a range's bottom units wait for)
    int unit_mode;
    int* abort;                            // one word behind them: != 0 once any wave of the launch has given up
    int* exited;                           // ... and one more: workgroups that have finished; the last one zeroes all of these words
    int active_wgs;                        // workgroups that own units (the others return at once)
    int* status;                           // pinned host word: != 0 after a give-up
    long long packed_stride;
    int proj_row_stride;
    int G, N, T, n_layers, units, per_wg, nwg, last_wg, reach_wgs, xcd_map, rot;
    int all_wt;                            // != 0: every ring store is write-through (a layer's rows exceed the L2s: see the launcher)
    int cond_hop, cond_offset, cond_frames;
    unsigned T_magic, T_shift, hop_magic, hop_shift;
    int dil[kMaxPLayers];
    const float* x_first;                  // != NULL: the run starts with the net's layer 0, which rebuilds the causal layer from the scalar input
    const float* cfilt[PWV_MAX_NETS];      // ... and each net's causal filter [2,1,64]
    const float* fold0[PWV_MAX_NETS];      // optional: layer 0's filter|gate GEMM folded onto the four scalars (pwv_pack_first_fold_f16x3 / _f32)
    float x_limit;                         // range guard of the split-fp16 arithmetic on x_first (include/pwv_hip.h)
    int* range_flag;
    long long* trace;                      // -DPWV_PTRACE builds: per-wave cycle accounting (tools/persist_trace.py)
    long long* trace_ev;                   // -DPWV_PTRACE builds: per-wave event timeline, [wave][64][4] = {s_memrealtime, code, layer, unit | bits << 32}
    // TAIL (tail_q > 0; both arithmetics since round 6): behind the run's layers every workgroup runs the net's LAST layer with the post-processing
    // head behind it on its own units (layer_f16x3_kernel's HEAD variant, the same operations) -- and, with `pair`, the IAF affine
    const float* tail_layer[PWV_MAX_NETS]; // the last layer's packed weights (its filter|gate fragments are used)
    const float* tail_head[PWV_MAX_NETS];  // the packed head
    float* tail_out[PWV_MAX_NETS];         // [rows, tail_q]
    int tail_q, tail_dil, tail_reach_wgs;
    const float* affine_x;                 // != NULL: out = fma(x, s, b) (modules.py:59) for the workgroup's rows, by whichever of the
    float* affine_out;                     // two nets' workgroups of a range finishes second (G = 2, Q = 1), or in place (G = 1, Q = 2)
    int* pair;                             // [nwg] arrival counters of the two nets' workgroups of a range (zero on entry and on exit)
};

// -DPWV_PTRACE: every wave accumulates s_memtime cycles: [0] whole loop, [1] drain at the top, [2] RAW spins, [3] WAR spins,
// [4] settle (publish / leave / refill), [5] units, [6] tasks that were not prefetched, [7] first task at, [8] last task done at;
// round 5, phases of a unit of the general loop (PT_PHASE: the time since the previous stamp goes to slot k): [9] top of the unit up
// to the drain (P row requested, next task located, look-back row split), [10] GEMM1, [11] GEMM2, [12] stores + moving on
#ifdef PWV_PTRACE
#define PT_DECL long long pt_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pt_t = 0, pt_p = 0; (void)pt_t; (void)pt_p;
#define PT_BEGIN() pt_t = __builtin_amdgcn_s_memtime()
#define PT_END(k) pt_acc[k] += __builtin_amdgcn_s_memtime() - pt_t
#define PT_ADD(k, v) pt_acc[k] += (v)
#define PT_MARK() do { __builtin_amdgcn_sched_barrier(0); pt_p = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PT_PHASE(k) do { __builtin_amdgcn_sched_barrier(0); const long long pt_n = __builtin_amdgcn_s_memtime(); pt_acc[k] += pt_n - pt_p; pt_p = pt_n; __builtin_amdgcn_sched_barrier(0); } while (0)
// event timeline (round 6, tools/persist_timeline.py): 1 unit computed, 2 stores issued, 3 own stores acknowledged (in front of a wait; 13 in front of a WAR wait), 4 dependencies
// satisfied (12: the WAR ones; bits << 32: what was missing at the first look), 5 rows arrived + look-back split, 6 top drained, 7 workgroup progress word published, 8 tail entered
#define PT_EV(code, jj, uu) do { if (p.trace_ev && pt_nev < 64) { if (lane == 0) { long long* e_ = p.trace_ev + (((size_t)blockIdx.x * 8 + wave) * 64 + pt_nev) * 4; \
    e_[0] = __builtin_amdgcn_s_memrealtime(); e_[1] = (code); e_[2] = (jj); e_[3] = (long long)(uu); } ++pt_nev; } } while (0)
#else
#define PT_DECL
#define PT_BEGIN() do {} while (0)
#define PT_END(k) do {} while (0)
#define PT_ADD(k, v) do {} while (0)
#define PT_MARK() do {} while (0)
#define PT_PHASE(k) do {} while (0)
#define PT_EV(code, jj, uu) do {} while (0)
#endif

// dependency bits of a task: [0] own x[t] rows, [1] [2] x[t-d] rows, [3] this layer's weights resident (RAW side);
// [4] [5] readers of the ring slot it overwrites (WAR)
constexpr unsigned kRawMask = 0xFu, kWarMask = 0x30u;

// GEMM2 (dense, K = 64) with the fragment source as a functor: the persistent kernel takes the last fragment from a register
template <typename FR, typename BH, typename BL, typename EF>
__device__ __forceinline__ void gemm16_dense(FR&& fr, f32x16 (&acc)[2], f16x8 (&ah)[4], f16x8 (&al)[4], BH&& bh, BL&& bl, EF&& extra) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f16x8 nh[2] = {ah[0], ah[1]};
        f16x8 nl[2] = {al[0], al[1]};
        if (s + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                nh[i] = fr(0, i, s + 1);
                nl[i] = fr(1, i, s + 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 b_h = bh(s);
        const f16x8 b_l = bl(s);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b_h, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b_l, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b_h, acc[i], 0, 0, 0);
        extra(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = nh[i];
            al[i] = nl[i];
        }
    }
}

// the same for the exact-fp32 arithmetic: 8 groups of (2 row tiles x 4 k-steps)
template <typename FR, typename BF, typename EF>
__device__ __forceinline__ void gemm_groups_dense(FR&& fr, f32x16 (&acc)[2], f32x4 (&a)[4], BF&& bval, EF&& extra) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        f32x4 n[2] = {a[0], a[1]};
        if (g + 1 < 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) n[i] = fr(i, g + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = bval(g * 4 + e);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b, acc[i], 0, 0, 0);
        }
        extra(g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = n[i];
    }
}

// SHORT (round 6): the instantiation for short inputs (at most kUnitModeMaxPerWg units per workgroup and layer: progress words per unit, stationary
// units, a loader wave) -- a template parameter, not a run-time mode: as run-time branches in the general task loop the additions cost the long-input
// launch 2.7 % of its step (SGPR spills reloaded per unit, profiles/r06_ab_experiments.md r06_r)
// MODE 1 (medium inputs, 8 ... PWV_MEDIUM_MAX_UNITS units per workgroup): the general kernel with the progress words per unit only.
template <bool F32, int MODE>
__global__ __launch_bounds__(512) void stack_persist_kernel(const PersistParams p) {
    constexpr bool SHORT = MODE == 2;      // stationary units, loader wave, ... (everything below that says SHORT)
    constexpr bool UNITW = MODE >= 1;      // progress words per unit
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    // block -> (net, range).  Observed, for speed only: block b runs on XCD b % 8 -- consecutive ranges of a net go to
    // blocks of one XCD, so most neighbour traffic stays inside one L2.  Nothing depends on it.
    int net, w;
    if (p.xcd_map) {
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
        net = s % p.G;
        w = x * (p.nwg >> 3) + s / p.G;
    } else {
        net = blockIdx.x % p.G;
        w = blockIdx.x / p.G;
    }
    const int rows = p.N * p.T;
    const int u_begin = w * p.per_wg;
    const int u_end = u_begin + p.per_wg < p.units ? u_begin + p.per_wg : p.units;
    const int n = u_end - u_begin;
    if (n <= 0) return;      // owns nothing; nobody waits for it (the neighbour sets stop at the last owning workgroup)
    const int L = p.n_layers;
    constexpr bool stat = SHORT;                        // stationary units (below, at the task loop): n <= kUnitModeMaxPerWg <= 8 units, one per wave
    constexpr bool loader_mode = SHORT;                 // ... leave wave 7 idle (n <= kUnitModeMaxPerWg = 7): it is the workgroup's loader
#ifdef PWV_PTRACE
    int pt_nev = 0;
#endif

    typedef __attribute__((address_space(3))) int* lds_ints_t;
    const lds_ints_t ctl = (lds_ints_t)(lds + kCtlF);
    // (an LDS-address-space pointer: through a generic one the byte accesses become flat_load / flat_store and count on vmcnt)
    typedef __attribute__((address_space(3))) volatile unsigned char* lds_bytes_t;
    const lds_bytes_t lb = (lds_bytes_t)lds;
    int* prog_n = p.prog + (size_t)net * p.nwg * kProgStride;
    int* uprog_n = p.uprog + (size_t)net * p.units * kUnitStride;
    const float* const proj_n = p.proj[net];
    const float* const packed_n = p.packed[net];
    // the per-layer dilations live in one VGPR (lane j holds entry j), read with v_readlane: a dynamically indexed kernel
    // argument is a scalar LOAD plus a wait each time
    const int v_dil = p.dil[lane & (kMaxPLayers - 1)];
    auto dil_of = [&](int j) -> int { return __builtin_amdgcn_readlane(v_dil, j); };
    // the layer that reads layer j's rows next: j + 1 of this launch, the tail's layer behind the last one, else (another launch
    // follows: anything) its own
    auto dil_next = [&](int j) -> int { return j + 1 < L ? dil_of(j + 1) : (p.tail_q > 0 ? p.tail_dil : dil_of(j)); };

    // ---- control state, then the weights of the first two layers (LDS-DMA, packed order == LDS order) -------------------
    for (int k = tid; k < (kLdsFloats - kCtlF); k += 512) ctl[k] = 0;
    __syncthreads();
    if (tid == 0) {
        lb[kSeenLB] = w > 0 ? 0 : 255;
        lb[kSeenRB] = w < p.last_wg ? 0 : 255;
        lb[kWreadyB] = 0;
        lb[kWreadyB + 1] = 1;
        lb[kTrueB] = 255;
    }
    auto fill_slot = [&](int slot, int layer, int first, int step) {
        const float* src = packed_n + (size_t)layer * p.packed_stride + lane * 4;
        float* dst = lds + slot * kSlot;
#pragma clang loop unroll(disable)
        for (int c = first; c < kSlot / 256; c += step)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + c * 256), (lptr_t)(dst + c * 256), 16, 0, 0);
        if (first == 0 && lane < 16)      // dense bias [2 h][32]
            __builtin_amdgcn_global_load_lds((gptr_t)(src + kSlotFull), (lptr_t)(lds + kBiasF + slot * 64), 16, 0, 0);
    };
    if (p.x_first && tid < 128) lds[kCfF + tid] = p.cfilt[net][tid];
    fill_slot(0, 0, wave, 8);
    if (L > 1) fill_slot(1, 1, wave, 8);
    __syncthreads();
#ifndef PWV_PERSIST_NOPRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

    // ONE buffer descriptor for the three ring buffers (they are one allocation); the buffer of a layer is selected by the
    // scalar offset operand of the load / store.  sc1 loads: L2-served, never the CU's L1.
    const __amdgpu_buffer_rsrc_t ring_rs = [&]() {
        const unsigned long long a = (unsigned long long)p.ring[net];
        const unsigned long long span = (unsigned long long)p.ring_stride * 8ull + (unsigned long long)p.units * 8192ull;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi2 << 32) | lo), 0, __builtin_amdgcn_readfirstlane((unsigned)span), 0x00020000);
    }();
    const __amdgpu_buffer_rsrc_t proj_rs = [&]() {      // (SHORT: the P rows through a buffer descriptor)
        const unsigned long long a = (unsigned long long)proj_n;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi2 << 32) | lo), 0, 0xFFFFFFFFu, 0x00020000);
    }();
    (void)proj_rs;
    const int slot_bytes = (int)(p.ring_stride * 4);
    auto in_soff = [&](int j) -> int { return ((j + 2 + p.rot) % 3) * slot_bytes; };
    auto out_soff = [&](int j) -> int { return ((j + p.rot) % 3) * slot_bytes; };
    auto toff = [&](int row) -> int { return ((row >> 5) * 2048 + h * 128 + (row & 31) * 4) * 4; };

    // x[t-d] / x[t] rows of one unit -> registers
    auto load_x = [&](int j, int unit, float (&xb)[32], float (&xc)[32]) {
        int row, rc, nn, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, nn, t);
        const int d = dil_of(j);
        const bool has_prev = t >= d;
        if (p.x_first && j == 0) {
            // layer 0 of the net: the four scalars its two rows are functions of (x[t], x[t-1], x[t-d], x[t-d-1]; zero left of
            // the utterance start); rebuilt into rows at the top of the unit (layer_f16x3_kernel's FIRST variant)
            const float* x1 = p.x_first;
            xc[0] = x1[rc];
            xc[1] = t >= 1 ? x1[rc - (t >= 1 ? 1 : 0)] : 0.f;
            xb[0] = has_prev ? x1[rc - (has_prev ? d : 0)] : 0.f;
            xb[1] = t >= d + 1 ? x1[rc - (t >= d + 1 ? d + 1 : 0)] : 0.f;
#pragma unroll
            for (int k = 2; k < 32; ++k) xb[k] = xc[k] = 0.f;      // (every element written on every path: the arrays stay in registers)
            return;
        }
        const int so = in_soff(j);
        const int oc = toff(rc), ob = toff(has_prev ? rc - d : rc);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring_rs, oc + g * 1024, so, 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) xc[4 * g + e] = v[e];
        }
        auto load_b = [&](bool keep) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring_rs, ob + g * 1024, so, 16));
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[4 * g + e] = keep ? v[e] : 0.f;
            }
        };
        if (__all(has_prev)) load_b(true);      // wave-uniform fast path: no select behind the loads, they stay in flight
        else load_b(has_prev);
    };

    // the unit's own rows alone (stationary units: once, in front of the task loop)
    auto load_xc = [&](int j, int unit, float (&xc)[32]) {
        int row, rc, nn, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, nn, t);
        const int so = in_soff(j);
        const int oc = toff(rc);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring_rs, oc + g * 1024, so, 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) xc[4 * g + e] = v[e];
        }
    };
    // the look-back row alone (stationary units, layers >= 1)
    auto load_xb = [&](int j, int unit, float (&xb)[32]) {
        int row, rc, nn, t;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, nn, t);
        const int d = dil_of(j);
        const bool has_prev = t >= d;
        const int so = in_soff(j);
        const int ob = toff(has_prev ? rc - d : rc);
        // (no select here: rows left of the utterance start are zeroed where the row is USED -- a select, or two paths that the register
        //  allocator joins with copies, behind these loads is a wait for them right here)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring_rs, ob + g * 1024, so, 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) xb[4 * g + e] = v[e];
        }
    };

    // ---- dependencies: lane k < 6 of a wave looks at ONE byte of LDS ------------------------------------------------------
    //   k = 0: own x[t] rows, 1 / 2: the x[t-d] rows (units u - ceil(d/32), u - floor(d/32)), 3: this layer's weights resident,
    //   4 / 5: the readers of the ring slot the task overwrites (layer j-2's tasks of the units u + floor(d'/32), u + ceil(d'/32)).
    // Per layer: the unit offsets and the values the bytes must have reached (two VGPRs); per task: one address VGPR.
    int vpack = 0;                     // need << 16 | (unit offset & 0xffff)
    auto layer_vectors = [&](int j) {
        const int d = dil_of(j), d2 = dil_of(j >= 2 ? j - 2 : 0);
        const int off = lane == 1 ? -((d + 31) >> 5) : (lane == 2 ? -(d >> 5) : (lane == 4 ? (d2 >> 5) : (lane == 5 ? ((d2 + 31) >> 5) : 0)));
        const int raw = j >= 1 ? j : 0, wts = j >= 2 ? j : 0, war = j >= 2 ? j - 1 : 0;
        const int need = lane < 3 ? raw : (lane == 3 ? wts : (lane < 6 ? war : 0));
        vpack = (need << 16) | (off & 0xffff);
    };
    auto dep_addr = [&](int j, int u) -> int {
        const int v = u + (int)(short)vpack;
        int a = kDoneB - u_begin + v;
        a = (v < u_begin && !UNITW) ? kSeenLB : a;      // (unit mode: the byte of that very unit, kLeftN bytes in front of the own ones)
        a = v >= u_end ? kSeenRB : a;
        a = (v < 0 || v >= p.units) ? kTrueB : a;
        a = lane == 3 ? kWreadyB + (j & 1) : a;
        return lane >= 6 ? kTrueB : a;
    };
    // bit k set: dependency k is NOT yet satisfied
    auto eval = [&](int addr) -> unsigned { return (unsigned)__ballot((int)lb[addr] < (vpack >> 16)); };
    // neighbours' progress words -> the cached "seen" byte of that side (only ever raised to a value that was observed)
    auto wave_min = [&](int v) -> int {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(v, o); v = t < v ? t : v; }
        return v;
    };
    // (round 6: the byte takes the value that was OBSERVED, not just `need` -- a neighbour is usually a layer further on than what is
    //  asked for, and a poll is a 1.5 us round trip to the fabric for a word another CU wrote through)
    auto poll_side = [&](int side, int need) {
        const int w0 = side ? w + 1 : (w - p.reach_wgs > 0 ? w - p.reach_wgs : 0);
        const int cnt = side ? (w + p.reach_wgs < p.last_wg ? p.reach_wgs : p.last_wg - w) : w - w0;
        int v = 255;
        int lo = lane;
        asm volatile("" : "+v"(lo));      // (address made here, not hoisted out of the task loop into a spilled register pair)
        if (lo < cnt) v = __hip_atomic_load(prog_n + (size_t)(w0 + lo) * kProgStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (UNITW) {
            const int m = wave_min(v);
            if (m >= need) lb[side ? kSeenRB : kSeenLB] = (unsigned char)m;
        } else {
            if (__ballot(v < need) == 0) lb[side ? kSeenRB : kSeenLB] = (unsigned char)need;
        }
    };
    // unit mode: lanes 1 / 2 ask for exactly the unit they wait for (own 128-byte line each), lanes 32.. for the right neighbours'
    // workgroup words in the same instruction (the WAR side's byte is refreshed on the way, so the top unit's stores rarely have to poll)
    auto poll_units = [&](int addr, int jw) {
        const int cnt_r = w + p.reach_wgs < p.last_wg ? p.reach_wgs : p.last_wg - w;
        int lo = lane;
        asm volatile("" : "+v"(lo));      // (the addresses are made HERE: hoisted out of the task loop as loop-invariant per-lane pointers they are spilled registers)
        const bool left = (lo == 1 || lo == 2) && addr >= kLeftB && addr < kDoneB;
        int v = 255;
        if (left) {
            v = __hip_atomic_load(uprog_n + (size_t)(u_begin - kLeftN + addr - kLeftB) * kUnitStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lo >= 32 && lo - 32 < cnt_r) {
            v = __hip_atomic_load(prog_n + (size_t)(w + 1 + lo - 32) * kProgStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        v = v > 255 ? 255 : v;
        if (left) lb[addr] = (unsigned char)v;
        // (the right neighbours are normally at layer jw or one further: two ballots instead of a reduction)
        if (cnt_r > 0) {
            const bool r = lo >= 32;
            if (__ballot(r && v < jw + 1) == 0) lb[kSeenRB] = (unsigned char)(jw + 1);
            else if (__ballot(r && v < jw) == 0 && (int)lb[kSeenRB] < jw) lb[kSeenRB] = (unsigned char)jw;
        }
    };

    // what this wave owes the others: the unit it has just stored and a weight refill it has issued (true at a vmcnt(0))
    int prev_addr = -1, prev_j = 0;      // LDS byte of the unit stored last, its layer
    int dma_pending = -1;                // layer whose LDS-DMA this wave issued and has not yet announced
    int left_upto = 0;                   // layers [0, left_upto) this wave has counted itself out of
    bool dead = false;
    auto publish = [&]() {               // (all lanes store the same byte: no exec juggling)
        if (prev_addr >= 0) {
            lb[prev_addr] = (unsigned char)(prev_j + 1);
            if (UNITW && lane == 0) __hip_atomic_store(uprog_n + (size_t)(prev_addr - kDoneB + u_begin) * kUnitStride, prev_j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev_addr = -1;
        }
        if (dma_pending >= 0) { lb[kWreadyB + (dma_pending & 1)] = (unsigned char)dma_pending; dma_pending = -1; }
    };
    auto flush_owed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish();
    };
    // leaving layer jj (after a drain: this wave's layer-jj stores are complete).  The LAST of the 8 waves publishes the
    // workgroup's progress and refills the LDS slot with layer jj + 2.
    auto leave_layers = [&](int upto) {
        for (; left_upto < upto; ++left_upto) {
            const int jj = left_upto;
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(&ctl[8 + jj], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == 7 && !loader_mode) {
                if (lane == 0) __hip_atomic_store(prog_n + (size_t)w * kProgStride, jj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                PT_EV(7, jj, -1);
                if (jj + 2 < L) {
                    if (dma_pending >= 0) flush_owed();       // (last twice in a row: announce the earlier refill first)
                    fill_slot(jj & 1, jj + 2, 0, 1);
                    dma_pending = jj + 2;
                }
            }
        }
    };
    // bounded wait for the dependencies `mask` of task (j, u) (rare: everything a task needs is normally a layer old);
    // `lv` = the layer vpack describes on entry and again on return
    auto wait_deps = [&](int j, int u, unsigned mask, int lv, int code) {
        // never spin while holding unpublished work -- and "work" includes leaving the layers this wave has moved past: with few
        // units per workgroup a wave's next task can be two layers on, and the weights it then waits for are refilled by the
        // LAST wave to leave the layer it has just finished
        if (!SHORT || prev_addr >= 0 || dma_pending >= 0) {
            flush_owed();
            PT_EV(mask == kWarMask ? 13 : 3, j, u);
        }
        leave_layers(j);
        if (dma_pending >= 0) flush_owed();
        if (lv != j) layer_vectors(j);
        const int addr = dep_addr(j, u);
        bool ok = false;
        const long long t0 = __builtin_amdgcn_s_memrealtime();
#ifdef PWV_PTRACE
        unsigned pt_bad0 = 0, pt_badl = 0;
        int pt_polls = 0;
#endif
        for (int k = 0; !ok; ++k) {
            const unsigned bad = eval(addr) & mask;
#ifdef PWV_PTRACE
            if (k == 0) pt_bad0 = bad;
            if (bad) pt_badl = bad;
            pt_polls = k;
#endif
            if (!bad) { ok = true; break; }
            // somebody has given up (this workgroup: LDS word; any workgroup of the launch: the word behind the progress words,
            // looked at every 64th poll), or this wait has lasted 20 ms: give up too.  (readfirstlane: the loop stays wave-uniform)
            if (__builtin_amdgcn_readfirstlane(*(__attribute__((address_space(3))) volatile int*)&ctl[1])) break;
            if ((k & 63) == 63 && (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ||
                                   __builtin_amdgcn_s_memrealtime() - t0 > kWaitTicks)) break;
            if constexpr (UNITW) {
                if ((bad & 0x6u) && __ballot((lane == 1 || lane == 2) && addr < kDoneB && addr >= kLeftB)) poll_units(addr, j);
            } else if ((bad & 0x6u) && __ballot(addr == kSeenLB && (lane == 1 || lane == 2))) poll_side(0, j);
            if ((bad & 0x30u) && __ballot(addr == kSeenRB && (lane == 4 || lane == 5))) poll_side(1, j - 1);
            __builtin_amdgcn_s_sleep(4);
        }
        if (lv != j) layer_vectors(lv);
        PT_EV(mask == kWarMask ? 12 : 4, j, (long long)u | ((long long)pt_bad0 << 32) | ((long long)pt_badl << 40) | ((long long)pt_polls << 48));
        if (ok) return;
        // (every lane stores the same words: a lane-0 branch here makes the compiler treat `dead`, and with it the whole
        // task loop, as divergent -- scalar bookkeeping in VGPRs, a waterfall loop around every buffer access)
        __hip_atomic_store(p.status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *(__attribute__((address_space(3))) volatile int*)&ctl[1] = 1;
        dead = true;
    };

    // ---- tasks: index i = layer * n + k, unit = u_end - 1 - k; claimed from the LDS counter one iteration ahead ---------
    auto claim = [&]() -> int {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;          // lane 0's value; readfirstlane at the point of use
    };
    auto locate = [&](int i, int& j) -> int {      // j: a layer at or before the task's (tasks are claimed in increasing order)
#pragma clang loop unroll(disable) vectorize(disable)
        while (j < L && i >= (j + 1) * n) ++j;      // (normally zero or one step: keep it a three-instruction scalar loop)
        return j < L ? u_end - 1 - (i - j * n) : -1;
    };
    // STATIONARY units (round 6; unit mode with at most one unit per wave): wave k owns unit u_end - 1 - k in EVERY layer, the other waves
    // have no tasks.  The unit's own rows x[t] then never travel: they are the accumulators the wave has just stored, and what it has to
    // fetch between two layers is the look-back row alone -- half the bytes in the CU's memory queue at the one moment a short layer waits
    // for (a CU loads freshly written rows at ~ 30 GB/s, latency-bound: 2.2 us for its four units' 64 KB; profiles/r06_short_timeline.md).
    int j = 0;
    int u = stat ? (wave < n ? u_end - 1 - wave : -1) : locate(__builtin_amdgcn_readfirstlane(claim()), j);
    int claim_v = stat ? 0 : claim();      // the task after that
    auto next_task = [&](int jc, int uc, int& jn) -> int {
        if (stat) { jn = jc + 1; return jn < L ? uc : -1; }
        return locate(__builtin_amdgcn_readfirstlane(claim_v), jn);
    };
    float rxb[32], rxc[32];
    bool war_ok = true;                    // (of the task in hand; its RAW side is satisfied when it starts)
    PT_DECL
#ifdef PWV_PTRACE
    const long long pt_start = __builtin_amdgcn_s_memtime();
    const long long pt_start_rt = __builtin_amdgcn_s_memrealtime();
    pt_acc[7] = pt_start;
#endif
    // ---- the loader (stationary units with an idle wave): it waits for the n active waves to have left layer jj, publishes the workgroup's
    // progress word and refills the LDS slot with layer jj + 2.  Left to the last wave to leave, as in the general scheme, the 80 KB of LDS-DMA
    // sit in THAT wave's memory queue in front of its next look-back row: 3 us on the top unit of every layer (r06_m timeline).
    if (loader_mode && wave == 7) {
        __builtin_amdgcn_s_setprio(0);      // (it shares its SIMD with an active wave)
        bool gone = false;
        for (int jj = 0; jj < L && !gone; ++jj) {
            const long long t0 = __builtin_amdgcn_s_memrealtime();
            for (int k = 0;; ++k) {
                if (__builtin_amdgcn_readfirstlane(*(__attribute__((address_space(3))) volatile int*)&ctl[8 + jj]) >= n) break;
                if (__builtin_amdgcn_readfirstlane(*(__attribute__((address_space(3))) volatile int*)&ctl[1])) { gone = true; break; }
                if ((k & 63) == 63 && (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ||
                                       __builtin_amdgcn_s_memrealtime() - t0 > kWaitTicks)) {
                    __hip_atomic_store(p.status, 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(p.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *(__attribute__((address_space(3))) volatile int*)&ctl[1] = 1;
                    gone = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (gone) break;
            if (lane == 0) __hip_atomic_store(prog_n + (size_t)w * kProgStride, jj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PT_EV(7, jj, -1);
            if (jj + 2 < L) {
                fill_slot(jj & 1, jj + 2, 0, 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lb[kWreadyB + (jj & 1)] = (unsigned char)(jj + 2);
            }
        }
    }
    // ---- layer 0 in its folded form (pwv_persist_args.first_fold), a loop of its own in front of the general one.
    // h[t] = x[t-1] w0 + x[t] w1 (modules.py:179-180) makes filter|gate(h[t-d], h[t]) a [4 -> 128] map of the scalars
    // x[t-d-1], x[t-d], x[t-1], x[t]: ONE split-fp16 MFMA k-step (4 of its 16 k values used) instead of eight -- two fp32
    // k-steps instead of 64 -- with no LDS fragment reads and no operand splits.  Layer 0 depends on nothing inside the launch
    // (its input was complete before it started and the ring slot it writes has no earlier reader), so this loop never waits;
    // tasks are layer-major, so it ends when the wave's next task is a layer-1 one, and the general loop's first-task code
    // takes over.  (As a branch INSIDE the general loop the two accumulator sets cost it 60-80 spilled registers.)
    if (p.x_first && p.fold0[net]) {
        const float* Af = lds;                                                    // layer 0's weights: slot 0
        const f16x8* A2 = reinterpret_cast<const f16x8*>(lds + kA1Size);
        (void)Af;
        (void)A2;
        const float* bias = lds + kBiasF + h * 32;
        const char* F0 = reinterpret_cast<const char*>(p.fold0[net]);
        const float* lastfrag = packed_n + kSlot + lane * 4;
        const int d = dil_of(0);
        const int dn = dil_next(0);
        typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
        const lds_f4_t cfb = (lds_f4_t)(lds + kCfF + 4 * h);
        // the folded fragments and the dense tail are the same for every unit: registers for the whole loop
        f16x8 fh[4], fl[4];      // split-fp16: [hi | lo][4 row tiles][64 lanes] f16x8
        f32x4 ff[4];             // fp32: [4 row tiles][64 lanes] {k = h, k = 2 + h, 0, 0}
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if constexpr (F32) {
                ff[it] = *reinterpret_cast<const f32x4*>(F0 + it * 1024 + lane * 16);
            } else {
                fh[it] = *reinterpret_cast<const f16x8*>(F0 + it * 1024 + lane * 16);
                fl[it] = *reinterpret_cast<const f16x8*>(F0 + (4 + it) * 1024 + lane * 16);
            }
        }
        const f32x4 lf32 = *reinterpret_cast<const f32x4*>(lastfrag);      // (16 bytes either way)
        while (u >= 0 && j == 0) {
            int row, rc, nn, t;
            bool valid;
            unit_rows(u, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, nn, t);
            // the four scalars (zero left of the utterance start) and the P row
            const float* x1 = p.x_first;
            const bool has_prev = t >= d;
            const float x0 = x1[rc];
            const float x1v = t >= 1 ? x1[rc - (t >= 1 ? 1 : 0)] : 0.f;
            const float xd0 = has_prev ? x1[rc - (has_prev ? d : 0)] : 0.f;
            const float xd1 = t >= d + 1 ? x1[rc - (t >= d + 1 ? d + 1 : 0)] : 0.f;
            f32x16 acc[4];
            {
                int prow = 0;
                if (p.cond_hop > 0) prow = nn * p.cond_frames + fast_div(t + p.cond_offset, p.hop_magic, p.hop_shift);
                const float* pr = proj_n + (size_t)prow * p.proj_row_stride + h * 64;
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(pr + it * 16 + q * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                    }
            }
            int j2 = 0;
            const int u2 = next_task(0, u, j2);
            // drain (the loads above, the previous unit's stores), publish that unit, claim the task after the next
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish();
            if (u2 >= 0 && !stat) claim_v = claim();
            if (p.range_flag && !(fabsf(x0) <= p.x_limit)) __hip_atomic_store(p.range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if constexpr (F32) {
                const float b0 = h ? xd0 : xd1, b1 = h ? x0 : x1v;      // k = 0, 1 | k = 2, 3
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ff[it][0], b0, acc[it], 0, 0, 0);
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ff[it][1], b1, acc[it], 0, 0, 0);
            } else {
                f16x8 b_h = {0, 0, 0, 0, 0, 0, 0, 0}, b_l = {0, 0, 0, 0, 0, 0, 0, 0};      // k = 0..3: lanes of the lower half
                const float sc[4] = {xd1, xd0, x1v, x0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = h == 0 ? sc[q] : 0.f;
                    const _Float16 vh = (_Float16)v;
                    b_h[q] = vh;
                    b_l[q] = (_Float16)(v - (float)vh);
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[it], b_h, acc[it], 0, 0, 0);
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[it], b_l, acc[it], 0, 0, 0);
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[it], b_h, acc[it], 0, 0, 0);
            }
            // GEMM2's accumulator starts at h[t] + dense_bias; h[t] with the operations of iaf_front_kernel (same bits as unfolded)
            f32x16 acc2[2];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w0 = cfb[2 * (4 * it + q)];
                    const f32x4 w1 = cfb[16 + 2 * (4 * it + q)];
                    const f32x4 bd = *reinterpret_cast<const f32x4*>(bias + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = fmaf(x0, w1[e], x1v * w0[e]) + bd[e];
                }
            float o[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[r] = gate_act(acc[0][r], acc[2][r]);
                o[16 + r] = gate_act(acc[1][r], acc[3][r]);
            }
            if constexpr (F32) {
                f32x4 a[4];
                a[0] = frag(Af, kA1Size, 0, 8, 0, lane);
                a[1] = frag(Af, kA1Size, 1, 8, 0, lane);
                gemm_groups_dense([&](int it, int g) -> f32x4 { return (it == 1 && g == 7) ? lf32 : frag(Af, kA1Size, it, 8, g, lane); },
                                  acc2, a, [&](int ks) -> float { return o[ks]; }, [](int) {});
            } else {
                const f16x8 lf = __builtin_bit_cast(f16x8, lf32);
                f16x8 ah[4], al[4];
                first_frags<4, 2, 0, 1, 2>(A2, lane, ah, al);
                f16x8 oh[4], ol[4];
                split8<0>(o, oh[0], ol[0]);
                split8<8>(o, oh[1], ol[1]);
                split8<16>(o, oh[2], ol[2]);
                split8<24>(o, oh[3], ol[3]);
                gemm16_dense(
                    [&](int comp, int it, int s) -> f16x8 { return (comp == 1 && it == 1 && s == 3) ? lf : frag16<4, 2>(A2, comp, it, s, lane); },
                    acc2, ah, al, [&](int s) -> f16x8 { return oh[s]; }, [&](int s) -> f16x8 { return ol[s]; }, [](int) {});
            }
            {
                const int so = out_soff(0);
                const int oo = toff(row);
                const bool shared = p.all_wt || u + ((dn + 31) >> 5) >= u_end;      // units the right neighbour reads in layer 1: write-through
                if (valid) {
                    if (shared) {
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            const int it = g >> 2, q = g & 3;
                            const f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ring_rs, oo + g * 1024, so, kAuxWriteThrough);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            const int it = g >> 2, q = g & 3;
                            const f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ring_rs, oo + g * 1024, so, PWV_PERSIST_STORE_AUX);
                        }
                    }
                }
            }
            PT_EV(2, 0, u);
            prev_addr = kDoneB + (u - u_begin);
            prev_j = 0;
            j = j2;
            u = u2;
        }
    }
    if (u >= 0) {
        // the first task of the general loop: nothing was prefetched (after the folded loop: drain and publish its last unit first)
        flush_owed();
        leave_layers(j);
        layer_vectors(j);
        if constexpr (SHORT) {
            // the unit's own rows: this wave's own layer-0 output (stationary units; complete: the drain above) or the run's input
            load_xc(j, u, rxc);
        } else {
            const unsigned bad = eval(dep_addr(j, u));
            if (bad & kRawMask) wait_deps(j, u, kRawMask, j, 4);
            war_ok = (bad & kWarMask) == 0;
            if (!dead) load_x(j, u, rxb, rxc);
        }
    }
    int lv_j = j;                          // layer voff / vneed currently describe

    // SHORT (stationary units): a unit is NOT software-pipelined over the previous one.  Its top: what the previous unit owes (drain, publish), then
    // its P row, then its dependencies, then its look-back row.  The loads are issued and consumed in ONE iteration: across the back-edge the
    // compiler's vmcnt bookkeeping is conservative, and a P row carried over as 64 accumulator registers costs GEMM1 fifty spilled ones.  (The packed
    // K order is x[t] first so that GEMM1 could start on the unit's own rows while the look-back row is in flight -- the "early half" -- which the
    // measurements of round 6 did not reward in any form the compiler or inline asm allows; see the comment at the dependency check below.)
    while (u >= 0 && !dead) {
        PT_MARK();
        // ---- TOP: P row requested; the rows of this unit were requested during the previous one ---------------------------
        int row, rc, nn, t;
        bool valid;
        unit_rows(u, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, nn, t);
        if constexpr (SHORT) {
            flush_owed();
            PT_EV(3, j, u);
            leave_layers(j);
        }
        f32x16 acc[4];
        if constexpr (SHORT) {
            // (buffer loads, like the rows: the compiler's scoreboard takes "all but the last 8 loads have landed" for the P row only if both are the
            //  same kind of vector-memory instruction; the launcher keeps the P rows of a short launch inside a descriptor's 4 GB)
            int prow = 0;
            if (p.cond_hop > 0) prow = nn * p.cond_frames + fast_div(t + p.cond_offset, p.hop_magic, p.hop_shift);
            const int po = (prow * p.proj_row_stride + j * 128 + h * 64) * 4;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(proj_rs, po + (it * 16 + q * 4) * 4, 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                }
        } else {
            int prow = 0;
            if (p.cond_hop > 0) prow = nn * p.cond_frames + fast_div(t + p.cond_offset, p.hop_magic, p.hop_shift);
            const float* pr = proj_n + (size_t)prow * p.proj_row_stride + j * 128 + h * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(pr + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                }
        }
        if constexpr (SHORT) {
            // (An "early half" -- GEMM1's x[t] k-steps of pair 0 run while the look-back row is still in flight -- needs the compiler's scoreboard to know that
            //  the P row has landed when the look-back loads go out: a vmcnt(0) BUILTIN directly behind the P loads does that, inline asm or a wait further
            //  down does not, and then the first MFMA gets a vmcnt(0), look-back row included.  Priced, profiles/r06_ab_experiments.md r06_u ... r06_x: with
            //  that builtin the wait for the P row (0.7 us, this path's latency, not a miss: touching the rows from the loader wave two layers ahead changes
            //  nothing) stands in front of the dependency check and costs more than the 24 MFMAs it frees: 0.457 against 0.452 ms at 1 x 16000; the P row
            //  requested in FRONT of the drain delays the publication the neighbours wait for by 1 us: 0.462 ms.  So: P row and look-back row in one queue,
            //  one wait in front of the first MFMA.)
            PT_EV(15, j, u);
            if (lv_j != j) { layer_vectors(j); lv_j = j; }
            const unsigned bad = eval(dep_addr(j, u)) & ~1u;      // (its own rows are this wave's previous output: program order)
            war_ok = (bad & kWarMask) == 0;
            if (bad & kRawMask) {
                wait_deps(j, u, kRawMask & ~1u, j, 4);
                if (dead) break;
            }
            PT_EV(16, j, u);
            load_xb(j, u, rxb);
            PT_EV(18, j, u);
        }
        // the next task (claimed an iteration ago) and the bytes it depends on
        int j2 = j;
        const int u2 = next_task(j, u, j2);
        unsigned bad2 = 0;                     // its dependency bits (one LDS byte per lane, read here under the P loads)
        if (u2 >= 0) {
            if (j2 != lv_j) { layer_vectors(j2); lv_j = j2; }
            bad2 = eval(dep_addr(j2, u2));
            if (stat) bad2 &= ~1u;             // (its own rows are this very task's output: program order)
        }

        if (!SHORT && p.x_first && j == 0) {
            // rebuild this lane's 32 channels (8g + 4h + e) of h[t] and h[t-d] from the scalars; the operation order of
            // iaf_front_kernel / the FIRST variant of the per-layer kernel: round(x[t-1] w0), then fma(x[t], w1, .)
            const float x0 = rxc[0], x1v = rxc[1], xd0 = rxb[0], xd1 = rxb[1];
            const bool has_prev = t >= dil_of(0);
            if (p.range_flag && !(fabsf(x0) <= p.x_limit)) __hip_atomic_store(p.range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(&lds[kCfF + 8 * g + 4 * h]);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(&lds[kCfF + 64 + 8 * g + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rxc[4 * g + e] = fmaf(x0, w1[e], x1v * w0[e]);
                    const float vb = fmaf(xd0, w1[e], xd1 * w0[e]);
                    rxb[4 * g + e] = has_prev ? vb : 0.f;
                }
            }
        }
        const float* bias = lds + kBiasF + (j & 1) * 64 + h * 32;
        float o[32];
        f32x16 acc2[2];
        // drain + publish + leave, behind the first operand work of the unit (the P row and the previous unit's stores land
        // meanwhile); then the verdict on the next task's dependencies
        auto settle_top = [&]() {
            PT_PHASE(9);
            if (!SHORT) PT_EV(5, j, u);
            PT_BEGIN();
            if (!SHORT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (SHORT: nothing is owed here -- the top of this unit has drained and published)
            PT_END(1);
            PT_EV(6, j, u);
            if constexpr (!SHORT) {      // (SHORT: the top of the unit has done all of it)
                publish();
                if (left_upto < j) leave_layers(j);
                if (u2 >= 0) claim_v = claim();
            }
            PT_ADD(5, 1);
            PT_MARK();
        };
        // the next task's rows: requested between GEMM1 and GEMM2, in flight under GEMM2 + gating + stores -- if their
        // producers are done (normally they are a layer-sweep old); otherwise behind this unit's stores, after a wait
        // (stationary units) the word of a LEFT NEIGHBOUR's unit the next task waits for: asked for here, under GEMM2 -- a poll is a
        // 2 us round trip, and that unit, its workgroup's top one, is usually through by now; the answer goes into its byte before the wait
        int early_v = -1;
        auto prefetch_next = [&]() {
            PT_PHASE(10);
            if (u2 >= 0 && !(bad2 & kRawMask) && !stat) {
                load_x(j2, u2, rxb, rxc);
            } else {      // (ends the old rows' live ranges: without it they would occupy 64 registers through both GEMMs)
#pragma unroll
                for (int k = 0; k < 32; ++k) rxb[k] = rxc[k] = 0.f;
            }
            if (stat && u2 >= 0 && (bad2 & 0x36u)) {
                // (lanes 1 / 2: the left neighbour's unit; lanes 32..: the RIGHT neighbours' workgroup words when the next task's stores will have
                //  to know that the readers of their ring slot are through -- the top unit's WAR side, a 2 - 3 us poll in front of its stores otherwise)
                const int a2 = dep_addr(j2, u2);
                const int cnt_r = w + p.reach_wgs < p.last_wg ? p.reach_wgs : p.last_wg - w;
                int lo = lane;
                asm volatile("" : "+v"(lo));
                if ((lo == 1 || lo == 2) && a2 >= kLeftB && a2 < kDoneB)
                    early_v = __hip_atomic_load(uprog_n + (size_t)(u_begin - kLeftN + a2 - kLeftB) * kUnitStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if ((bad2 & 0x30u) && lo >= 32 && lo - 32 < cnt_r)
                    early_v = __hip_atomic_load(prog_n + (size_t)(w + 1 + lo - 32) * kProgStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // the last fragment of the dense matrix (not in LDS): global memory, 16 bytes per lane
        const float* lastfrag = packed_n + (size_t)j * p.packed_stride + kSlot + lane * 4;

        if constexpr (F32) {
            // ---- exact-fp32 arithmetic: v_mfma_f32_32x32x2_f32, the operands are the rows as loaded (pwv_layer.hip) ----
            float xc[32], xb[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) { xc[k] = rxc[k]; xb[k] = rxb[k]; }
            if constexpr (SHORT) {      // (SHORT: the look-back row arrives unselected, see load_xb; rows left of the utterance start are zero, modules.py:24-28)
                if (!__all(t >= dil_of(j))) {
                    const bool hp = t >= dil_of(j);
#pragma unroll
                    for (int k = 0; k < 32; ++k) xb[k] = hp ? xb[k] : 0.f;
                }
            }
            settle_top();
            const float* Af = lds + (j & 1) * kSlot;                 // [kA1 | kA2 minus its last fragment]
            f32x4 a[4];
            f32x4 lf = {0.f, 0.f, 0.f, 0.f};
            auto bx = [&](int ks) -> float { return ks < 32 ? xb[ks] : xc[ks - 32]; };
            a[0] = frag(Af, 0, 0, 16, 0, lane);
            a[1] = frag(Af, 0, 2, 16, 0, lane);
            gemm_groups<16, 2, 0, 2>(Af, 0, lane, acc, a, bx, [](int) {}, [&](f32x4(&nf)[4]) {
                nf[0] = frag(Af, 0, 1, 16, 0, lane);
                nf[1] = frag(Af, 0, 3, 16, 0, lane);
            });
            gemm_groups<16, 2, 1, 2>(
                Af, 0, lane, acc, a, bx,
                [&](int g) {
                    o[g] = gate_act(acc[0][g], acc[2][g]);
                    asm volatile("" : "+v"(o[g]));   // keep the gating inside this MFMA group (no sinking)
                    if (g == 10) lf = *reinterpret_cast<const f32x4*>(lastfrag);
                },
                [&](f32x4(&nf)[4]) {
                    nf[0] = frag(Af, kA1Size, 0, 8, 0, lane);
                    nf[1] = frag(Af, kA1Size, 1, 8, 0, lane);
                });
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bd = *reinterpret_cast<const f32x4*>(bias + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = xc[it * 16 + q * 4 + e] + bd[e];
                }
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(lf));
            prefetch_next();
            gemm_groups_dense(
                [&](int it, int g) -> f32x4 { return (it == 1 && g == 7) ? lf : frag(Af, kA1Size, it, 8, g, lane); }, acc2, a,
                [&](int ks) -> float { return o[ks]; },
                [&](int g) {
                    if (g < 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[16 + 4 * g + e] = gate_act(acc[1][4 * g + e], acc[3][4 * g + e]);
                            asm volatile("" : "+v"(o[16 + 4 * g + e]));
                        }
                    }
                });
        } else {
            const f16x8* A1 = reinterpret_cast<const f16x8*>(lds + (j & 1) * kSlot);
            const f16x8* A2 = reinterpret_cast<const f16x8*>(lds + (j & 1) * kSlot + kA1Size);

            f16x8 bh[8], bl[8];      // B operands: bh[0..3] = x[t-d], bh[4..7] = x[t]; packed K order: x[t] first (pwv_layer_f16.hip)
            float xc[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) xc[k] = rxc[k];
            split8<0>(xc, bh[4], bl[4]);
            split8<8>(xc, bh[5], bl[5]);
            split8<16>(xc, bh[6], bl[6]);
            split8<24>(xc, bh[7], bl[7]);
            settle_top();
            auto bxh = [&](int s) -> f16x8 { return bh[s ^ 4]; };
            auto bxl = [&](int s) -> f16x8 { return bl[s ^ 4]; };
            f16x8 oh[4], ol[4];
            f16x8 ah[4], al[4];
            f16x8 lf = {0, 0, 0, 0, 0, 0, 0, 0};

            {
            // ---- GEMM1, row-tile pair 0 = (F[0:32], G[0:32]); x[t-d] is split under its first four MFMA groups ----------
            first_frags<8, 2, 0, 2, 4>(A1, lane, ah, al);
            gemm16<8, 2, 0, 2, 4>(
                A1, lane, acc, ah, al, bxh, bxl,
                [&](int s) {
                    if constexpr (SHORT) {      // the look-back row (no select behind its loads, see load_xb) is zeroed left of the utterance start and split HERE, in one piece
                        if (s == 3) {
                            PT_EV(14, j, u);
                            if (!__all(t >= dil_of(j))) {      // (rows left of the utterance start: zero, modules.py:24-28)
                                const bool hp = t >= dil_of(j);
#pragma unroll
                                for (int k = 0; k < 32; ++k) rxb[k] = hp ? rxb[k] : 0.f;
                            }
                            split8<0>(rxb, bh[0], bl[0]);
                            split8<8>(rxb, bh[1], bl[1]);
                            split8<16>(rxb, bh[2], bl[2]);
                            split8<24>(rxb, bh[3], bl[3]);
                            asm volatile("" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]), "+v"(bh[2]), "+v"(bl[2]), "+v"(bh[3]), "+v"(bl[3]));
                            PT_EV(5, j, u);
                        }
                    } else {
                        if (s == 0) { split8<0>(rxb, bh[0], bl[0]); asm volatile("" : "+v"(bh[0]), "+v"(bl[0])); }
                        if (s == 1) { split8<8>(rxb, bh[1], bl[1]); asm volatile("" : "+v"(bh[1]), "+v"(bl[1])); }
                        if (s == 2) { split8<16>(rxb, bh[2], bl[2]); asm volatile("" : "+v"(bh[2]), "+v"(bl[2])); }
                        if (s == 3) { split8<24>(rxb, bh[3], bl[3]); asm volatile("" : "+v"(bh[3]), "+v"(bl[3])); }
                    }
                },
                [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 2, 1, 2, 4>(A1, lane, nh, nl); });
            // ---- pair 1 = (F[32:64], G[32:64]); pair 0 is gated + split under these MFMAs -----------------------------------
            gemm16<8, 2, 1, 2, 4>(
                A1, lane, acc, ah, al, bxh, bxl,
                [&](int s) {
                    o[2 * s] = gate_act(acc[0][2 * s], acc[2][2 * s]);
                    o[2 * s + 1] = gate_act(acc[0][2 * s + 1], acc[2][2 * s + 1]);
                    asm volatile("" : "+v"(o[2 * s]), "+v"(o[2 * s + 1]));
                    if (s == 3) { split8<0>(o, oh[0], ol[0]); asm volatile("" : "+v"(oh[0]), "+v"(ol[0])); }
                    if (s == 7) { split8<8>(o, oh[1], ol[1]); asm volatile("" : "+v"(oh[1]), "+v"(ol[1])); }
                    if (s == 5) lf = *reinterpret_cast<const f16x8*>(lastfrag);      // lands under the last two k-steps
                },
                [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<4, 2, 0, 1, 2>(A2, lane, nh, nl); });
            }

            // ---- GEMM2: dense 64 -> 64, accumulator starts at x[t] + dense_bias ---------------------------------------------
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bd = *reinterpret_cast<const f32x4*>(bias + it * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = xc[it * 16 + q * 4 + e] + bd[e];
                }
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(lf));
            prefetch_next();      // (xc is dead from here on)
            gemm16_dense(
                [&](int comp, int it, int s) -> f16x8 { return (comp == 1 && it == 1 && s == 3) ? lf : frag16<4, 2>(A2, comp, it, s, lane); },
                acc2, ah, al, [&](int s) -> f16x8 { return oh[s]; }, [&](int s) -> f16x8 { return ol[s]; },
                [&](int s) {
                    if (s < 2) {   // k-steps 0,1 use o tile 0; gate + split tile 1 under them
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[16 + 8 * s + e] = gate_act(acc[1][8 * s + e], acc[3][8 * s + e]);
                        if (s == 0) split8<16>(o, oh[2], ol[2]);
                        else split8<24>(o, oh[3], ol[3]);
                        asm volatile("" : "+v"(oh[2 + (s & 1)]), "+v"(ol[2 + (s & 1)]));
                    }
                });
        }
        if (dead) break;
        PT_PHASE(11);
        PT_EV(1, j, u);
        // ---- stores (after the readers of the ring slot they overwrite are known to be done) -------------------------------
        if (!war_ok) {
            // (that verdict is a task old: look again before the machinery of a wait -- drain, leave, poll -- is set in motion; 1 us per unit on
            //  short inputs, where every unit's verdict is stale, profiles/r06_short_timeline.md)
            unsigned badw = kWarMask;
            if constexpr (UNITW) {
                if (lv_j != j) layer_vectors(j);
                badw = eval(dep_addr(j, u)) & kWarMask;
                if (lv_j != j) layer_vectors(lv_j);
            }
            if (badw) {
                PT_BEGIN();
                wait_deps(j, u, kWarMask, lv_j, 5);
                PT_END(3);
                if (dead) break;
            }
        }
        {
            const int so = out_soff(j);
            const int oo = toff(row);
            // units the right neighbour reads as x[t-d] in the next layer are stored write-through
            const int dn = dil_next(j);
            const bool shared = p.all_wt || u + ((dn + 31) >> 5) >= u_end;
            if (valid) {
                if (shared) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int it = g >> 2, q = g & 3;
                        const f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ring_rs, oo + g * 1024, so, kAuxWriteThrough);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int it = g >> 2, q = g & 3;
                        const f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ring_rs, oo + g * 1024, so, PWV_PERSIST_STORE_AUX);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        PT_EV(2, j, u);

        // ---- move on ------------------------------------------------------------------------------------------------------
        prev_addr = kDoneB + (u - u_begin);
        prev_j = j;
        if (stat && u2 >= 0 && (bad2 & 0x36u)) {      // the early poll's answer (lv_j == j2 here)
            const int a2 = dep_addr(j2, u2);
            if (early_v >= 0 && lane < 32) lb[a2] = (unsigned char)(early_v > 255 ? 255 : early_v);
            if ((bad2 & 0x30u) && w < p.last_wg && __ballot(lane >= 32 && early_v >= 0 && early_v < j2 - 1) == 0 && (int)lb[kSeenRB] < j2 - 1)
                lb[kSeenRB] = (unsigned char)(j2 - 1);
        }
        if constexpr (SHORT) {
            // (the next unit's top drains, publishes, waits and loads; its own rows are these accumulators)
#pragma unroll
            for (int k = 0; k < 32; ++k) rxc[k] = acc2[k >> 4][k & 15];
        } else if (u2 >= 0 && (bad2 & kRawMask)) {
            // the next task's producers were still at work when this unit looked: publish what this wave owes (a wave never
            // spins while holding unpublished work), wait, then load with the latency exposed
            PT_BEGIN();
            PT_ADD(6, 1);
            wait_deps(j2, u2, kRawMask, lv_j, 4);
            PT_END(2);
            if (dead) break;
            load_x(j2, u2, rxb, rxc);
        }
        j = j2;
        u = u2;
        if constexpr (!SHORT) war_ok = (bad2 & kWarMask) == 0;
        PT_PHASE(12);
    }
    // the last unit's stores, a refill this wave still owes, and the layers it has not yet counted itself out of
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!dead && !(loader_mode && wave >= n)) {      // (with a loader the idle waves are not counted: it waits for the n active ones)
        publish();
        leave_layers(L);
        if (dma_pending >= 0) flush_owed();
    }
    // ---- TAIL: the net's LAST layer with the head behind it (modules.py:145-165), then the IAF affine (modules.py:59) -------------
    // What used to be two more launches per flow (layer_f16x3_kernel<..., HEAD> and the affine) runs here on the workgroup's own
    // units as soon as ITS eight waves have left the run's last layer -- no grid-wide drain, no launch ramp.  The operations are
    // those of the HEAD variant in the same order (bit-identical, tests/test_gpu_persist.py).  The head's three matrices take
    // the whole LDS (filter|gate 64 KB + skip 32 KB + postprocess1 64 KB), so the control state above is gone from here on:
    // units are handed out statically, the left neighbours' progress words are polled directly, and the exit accounting at
    // the bottom is done by one thread behind a barrier.
    bool tail_done = false;
    {
        if (p.tail_q > 0) {
            tail_done = true;
            PT_EV(8, L, -1);
            __syncthreads();                       // every wave of the workgroup is out of the task loop (its stores drained, its layers left)
            const int wg_dead = __builtin_amdgcn_readfirstlane(*(__attribute__((address_space(3))) volatile int*)&ctl[1]);
            __syncthreads();                       // ... and has read that word before the weights overwrite it
            bool tail_ok = !wg_dead;
            if (tail_ok) {
                constexpr int kHS = kA1Size, kH1 = kA1Size + kASSize;
                fill_lds_dma<kA1Size / 4, 8>(lds, p.tail_layer[net] + kA1, wave, lane);
                fill_lds_dma<kASSize / 4, 8>(lds + kHS, p.tail_head[net] + kHAS, wave, lane);
                fill_lds_dma<kHA1Size / 4, 8>(lds + kH1, p.tail_head[net] + kHA1, wave, lane);
                // the look-back of this wave's first unit (u_begin + wave) reaches into the left neighbours iff wave < reach: they must
                // have completed the run's last layer ("layers completed for all my units" == L), bounded like every other wait
                const int td = p.tail_dil;
                if (w > 0 && wave < ((td + 31) >> 5)) {
                    const int w0 = w - p.tail_reach_wgs > 0 ? w - p.tail_reach_wgs : 0;
                    const int cnt = w - w0;
                    const long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (int k = 0;; ++k) {
                        int v = 1 << 20;
                        if (lane < cnt) v = __hip_atomic_load(prog_n + (size_t)(w0 + lane) * kProgStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__ballot(v < L) == 0) break;
                        if ((k & 63) == 63 && (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ||
                                               __builtin_amdgcn_s_memrealtime() - t0 > kWaitTicks)) {
                            __hip_atomic_store(p.status, 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(p.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            tail_ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                   // the head's weights are resident
                PT_EV(9, L, -1);
                const f16x8* A1 = reinterpret_cast<const f16x8*>(lds);
                const f16x8* HS = reinterpret_cast<const f16x8*>(lds + kHS);
                const f16x8* H1 = reinterpret_cast<const f16x8*>(lds + kH1);
                (void)A1; (void)HS; (void)H1;
                const float* hb = p.tail_head[net];
                const int Q = p.tail_q;
                const int so = in_soff(L);         // the run's last layer wrote buffer (L - 1 + rot) % 3
                const __amdgpu_buffer_rsrc_t out_rs = [&]() {
                    const unsigned long long a = (unsigned long long)p.tail_out[net];
                    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
                    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi2 << 32) | lo), 0,
                                                             __builtin_amdgcn_readfirstlane((unsigned)rows * (unsigned)Q * 4u), 0x00020000);
                }();
                auto load_tail = [&](int unit, float (&xb)[32], float (&xc)[32]) {
                    int row, rc, nn, t;
                    bool valid;
                    unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, nn, t);
                    const bool has_prev = t >= td;
                    const int oc = toff(rc), ob = toff(has_prev ? rc - td : rc);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring_rs, oc + g * 1024, so, 16));
#pragma unroll
                        for (int e = 0; e < 4; ++e) xc[4 * g + e] = v[e];
                    }
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring_rs, ob + g * 1024, so, 16));
#pragma unroll
                        for (int e = 0; e < 4; ++e) xb[4 * g + e] = has_prev ? v[e] : 0.f;
                    }
                };
                auto no_extra = [](int) {};
                int unit = u_begin + wave;
                float txb[32], txc[32];
                load_tail(unit, txb, txc);      // (unconditional, like every load of rows here: clamped addresses, and registers that are
                                                // written on every path do not stay live across the GEMMs)
                while (tail_ok && unit < u_end) {
                    // (compiler barrier: the head's small vectors -- skip / postprocess1 biases, postprocess2 -- are read from global
                    // memory per unit like P; hoisted out of the loop they are 192 loop-invariant registers, i.e. spills)
                    asm volatile("" ::: "memory");
                    const int next = unit + 8;
                    int row, rc, nn, t;
                    bool valid;
                    unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, nn, t);
                    f32x16 acc[4];
                    {
                        int prow = 0;
                        if (p.cond_hop > 0) prow = nn * p.cond_frames + fast_div(t + p.cond_offset, p.hop_magic, p.hop_shift);
                        const float* pr = proj_n + (size_t)prow * p.proj_row_stride + L * 128 + h * 64;
#pragma unroll
                        for (int it = 0; it < 4; ++it)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(pr + it * 16 + q * 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                            }
                    }
                    f32x16 acc1[4];      // postprocess1's accumulators: what both arithmetics hand to the postprocess2 dot below
                    if constexpr (F32) {
                        // ---- exact fp32 (round 6): the operations of layer_f32_kernel<8, ..., GATED, HEAD> in the same order ----------
                        auto bx = [&](int ks) -> float { return ks < 32 ? txb[ks] : txc[ks - 32]; };
                        float o[32];
                        f32x4 a[4];
                        a[0] = frag(lds, 0, 0, 16, 0, lane);
                        a[1] = frag(lds, 0, 2, 16, 0, lane);
                        gemm_groups<16, 2, 0, 2>(lds, 0, lane, acc, a, bx, no_extra, [&](f32x4(&nf)[4]) {
                            nf[0] = frag(lds, 0, 1, 16, 0, lane);
                            nf[1] = frag(lds, 0, 3, 16, 0, lane);
                        });
                        gemm_groups<16, 2, 1, 2>(
                            lds, 0, lane, acc, a, bx,
                            [&](int g) {
                                o[g] = gate_act(acc[0][g], acc[2][g]);
                                asm volatile("" : "+v"(o[g]));
                            },
                            [&](f32x4(&nf)[4]) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) nf[i] = frag(lds, kHS, i, 8, 0, lane);
                            });
                        asm volatile("" ::: "memory");      // (the bias loads below stay behind GEMM1: hoisted to the top of the unit they are 64 + 64 registers too many)
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
                        gemm_groups<8, 4, 0, 1>(lds, kHS, lane, accs, a, [&](int ks) -> float { return o[ks]; }, no_extra, [&](f32x4(&nf)[4]) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) nf[i] = frag(lds, kH1, i, 16, 0, lane);
                        });
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int it = 0; it < 4; ++it)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(hb + kHB1 + h * 64 + it * 16 + q * 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc1[it][q * 4 + e] = v[e];
                            }
                        gemm_groups<16, 4, 0, 1>(lds, kH1, lane, acc1, a, [&](int ks) -> float { return fmaxf(accs[ks >> 4][ks & 15], 0.f); }, no_extra,
                                                 [](f32x4(&)[4]) {});
                    } else {
                        f16x8 bh[8], bl[8];
                        float xc[32];
#pragma unroll
                        for (int i = 0; i < 32; ++i) xc[i] = txc[i];
                        split8<0>(xc, bh[4], bl[4]);
                        split8<8>(xc, bh[5], bl[5]);
                        split8<16>(xc, bh[6], bl[6]);
                        split8<24>(xc, bh[7], bl[7]);
                        auto bxh = [&](int s) -> f16x8 { return bh[s ^ 4]; };
                        auto bxl = [&](int s) -> f16x8 { return bl[s ^ 4]; };
                        float o[32];
                        f16x8 oh[4], ol[4];
                        f16x8 ah[4], al[4];
                        first_frags<8, 2, 0, 2, 4>(A1, lane, ah, al);
                        gemm16<8, 2, 0, 2, 4>(
                            A1, lane, acc, ah, al, bxh, bxl,
                            [&](int s) {
                                if (s == 0) { split8<0>(txb, bh[0], bl[0]); asm volatile("" : "+v"(bh[0]), "+v"(bl[0])); }
                                if (s == 1) { split8<8>(txb, bh[1], bl[1]); asm volatile("" : "+v"(bh[1]), "+v"(bl[1])); }
                                if (s == 2) { split8<16>(txb, bh[2], bl[2]); asm volatile("" : "+v"(bh[2]), "+v"(bl[2])); }
                                if (s == 3) { split8<24>(txb, bh[3], bl[3]); asm volatile("" : "+v"(bh[3]), "+v"(bl[3])); }
                            },
                            [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 2, 1, 2, 4>(A1, lane, nh, nl); });
                        gemm16<8, 2, 1, 2, 4>(
                            A1, lane, acc, ah, al, bxh, bxl,
                            [&](int s) {
                                o[2 * s] = gate_act(acc[0][2 * s], acc[2][2 * s]);
                                o[2 * s + 1] = gate_act(acc[0][2 * s + 1], acc[2][2 * s + 1]);
                                asm volatile("" : "+v"(o[2 * s]), "+v"(o[2 * s + 1]));
                                if (s == 3) { split8<0>(o, oh[0], ol[0]); asm volatile("" : "+v"(oh[0]), "+v"(ol[0])); }
                                if (s == 7) { split8<8>(o, oh[1], ol[1]); asm volatile("" : "+v"(oh[1]), "+v"(ol[1])); }
                            },
                            [](f16x8(&)[4], f16x8(&)[4]) {});
                        // ---- head: o (registers) -> skip -> relu -> postprocess1 -> relu -> postprocess2 ---------------------------
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
                        split8<16>(o, oh[2], ol[2]);
                        split8<24>(o, oh[3], ol[3]);
                        first_frags<4, 4, 0, 1, 4>(HS, lane, ah, al);
                        gemm16<4, 4, 0, 1, 4>(HS, lane, accs, ah, al, [&](int s) -> f16x8 { return oh[s]; }, [&](int s) -> f16x8 { return ol[s]; }, no_extra,
                                              [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 4, 0, 1, 4>(H1, lane, nh, nl); });
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
                        gemm16<8, 4, 0, 1, 4>(H1, lane, acc1, ah, al, [&](int s) -> f16x8 { return sh[s]; }, [&](int s) -> f16x8 { return sl[s]; }, no_extra,
                                              [](f16x8(&)[4], f16x8(&)[4]) {});
                    }
                    load_tail(next, txb, txc);      // the next unit's rows: in flight under the postprocess2 dot
                    float outv[kMaxQ] = {0.f, 0.f, 0.f, 0.f};
                    // (the addresses of the postprocess2 dot are made here, behind the GEMMs, from opaque copies: hoisted out of the unit loop as
                    // loop-invariant per-lane pointers they are the registers the exact-fp32 instantiation has to spill)
                    int out_idx = row * Q, hq = h * Q;
                    asm volatile("" : "+v"(out_idx), "+v"(hq));
                    for (int q = 0; q < Q; ++q) {
                        float part = 0.f;
                        const float* w2 = hb + kHW2 + (hq + q) * 64;
#pragma unroll
                        for (int i4 = 0; i4 < 16; ++i4) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(w2 + 4 * i4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int i = 4 * i4 + e;
                                part = fmaf(fmaxf(acc1[i >> 4][i & 15], 0.f), wv[e], part);
                            }
                        }
                        part += __shfl_xor(part, 32);
                        part += hb[kHW2 + 2 * Q * 64 + q];
                        if (q < kMaxQ) outv[q] = part;
                        // (write-through: with `pair` the other net's workgroup of this range may be the one that reads it)
                        if (valid && h == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part), out_rs, (out_idx + q) * 4, 0, kAuxWriteThrough);
                    }
                    if (p.affine_x && p.G == 1 && Q == 2) {      // one net with two outputs (scale, shift): the affine right here
                        if (valid && h == 0) p.affine_out[row] = fmaf(p.affine_x[row], outv[0], outv[1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    PT_EV(10, L, unit);
                    unit = next;
                }
            }
            // ---- the IAF affine for this range, by whichever of the two nets' workgroups arrives second ---------------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's outputs are written through
            __syncthreads();
            if (p.affine_x && p.pair && p.G == 2) {
                if (tid == 0) {
                    const int old = __hip_atomic_fetch_add(p.pair + w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *(__attribute__((address_space(3))) volatile int*)&ctl[0] = old;
                }
                __syncthreads();
                const int old = __builtin_amdgcn_readfirstlane(*(__attribute__((address_space(3))) volatile int*)&ctl[0]);
                if (old == 1) {
                    const int r_end = u_end * 32 < rows ? u_end * 32 : rows;
                    for (int row = u_begin * 32 + tid; row < r_end; row += 512) {
                        const float sv = __hip_atomic_load(p.tail_out[0] + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float bv = __hip_atomic_load(p.tail_out[1] + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        p.affine_out[row] = fmaf(p.affine_x[row], sv, bv);
                    }
                    if (tid == 0) __hip_atomic_store(p.pair + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // exit accounting (below) without the LDS counter: one thread, behind the barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            PT_EV(11, L, -1);
            if (wave == 0) {
                int done = 0;
                if (lane == 0) done = __hip_atomic_fetch_add(p.exited, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(done) == p.active_wgs - 1) {
                    for (int k = lane; k < p.G * p.nwg; k += 64) __hip_atomic_store(p.prog + (size_t)k * kProgStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (UNITW) for (int k = lane; k < p.G * p.units; k += 64) __hip_atomic_store(p.uprog + (size_t)k * kUnitStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0) {
                        __hip_atomic_store(p.abort, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p.exited, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
    }
    // The launch cleans up after itself: the LAST workgroup to finish zeroes every word a later launch polls (progress, abort,
    // this counter), so a launch that is handed this workspace again needs no zeroing kernel in front of it.  A wave counts
    // itself out only when its own global stores are complete (vmcnt(0)): no progress word can land after the zeroing.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!tail_done) {
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&ctl[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_readfirstlane(old) == 7) {
            int done = 0;
            if (lane == 0) done = __hip_atomic_fetch_add(p.exited, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_readfirstlane(done) == p.active_wgs - 1) {
                for (int k = lane; k < p.G * p.nwg; k += 64) __hip_atomic_store(p.prog + (size_t)k * kProgStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (UNITW) for (int k = lane; k < p.G * p.units; k += 64) __hip_atomic_store(p.uprog + (size_t)k * kUnitStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) {
                    __hip_atomic_store(p.abort, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.exited, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
#ifdef PWV_PTRACE
    if (p.trace && lane == 0) {
        pt_acc[8] = __builtin_amdgcn_s_memtime();
        pt_acc[0] = pt_acc[8] - pt_start;
        long long* tr = p.trace + ((size_t)blockIdx.x * 8 + wave) * 24;
        for (int k = 0; k < 13; ++k) tr[k] = pt_acc[k];
        tr[16] = net; tr[17] = w; tr[18] = dead ? 1 : 0; tr[19] = __builtin_amdgcn_s_memrealtime(); tr[20] = pt_start_rt;
    }
#endif
}

// every polled word starts at zero on EVERY call: either the caller says the workspace is clean (fresh zeros, or left by an
// earlier launch, which cleans up after itself) or this kernel runs first.  A kernel, not hipMemsetAsync: under stream capture the
// memset node of a torch-captured graph did not reset the words on replay (tests/test_gpu_persist.py::test_whole_model_persistent_eager_and_graph_replay)
__global__ void persist_zero_kernel(int4* p, size_t n16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = int4{0, 0, 0, 0};
}

}  // namespace pwv

using namespace pwv;

extern "C" {

int pwv_persist_status(int** status) {
    static int* g_status = nullptr;      // process lifetime; pinned + mapped: one pointer valid on host and device
    PWV_CHECK_ARG(status, "pwv_persist_status: NULL argument");
    if (!g_status) {
        PWV_CHECK_HIP(hipHostMalloc((void**)&g_status, sizeof(int), hipHostMallocMapped | hipHostMallocPortable));
        *g_status = 0;
    }
    *status = g_status;
    return PWV_OK;
}

struct PersistPlan { int units, nwg, per_wg, last_wg, reach_wgs, xcd_map, tail_reach_wgs, unit_mode; };

static int persist_plan(int G, long long rows, int n_layers, const int* dil, int cus, int max_wgs, int min_units, int tail_q, int tail_dil, PersistPlan& pl) {
    PWV_CHECK_ARG(G >= 1 && G <= PWV_MAX_NETS, "persistent stack: G=%d out of range", G);
    PWV_CHECK_ARG(n_layers >= 2 && n_layers <= kMaxPLayers && dil, "persistent stack: 2..%d layers per launch, got %d", kMaxPLayers, n_layers);
    PWV_CHECK_ARG(rows >= 1 && rows < (1ll << 31) - 256, "persistent stack: bad N*T");
    PWV_CHECK_ARG(cus >= G, "persistent stack: %d CUs for %d nets", cus, G);
    int dmax = 1;
    for (int j = 0; j < n_layers; ++j) {
        PWV_CHECK_ARG(dil[j] >= 1, "persistent stack: bad dilation");
        dmax = dil[j] > dmax ? dil[j] : dmax;
    }
    pl.units = (int)((rows + 31) / 32);
    PWV_CHECK_ARG((long long)pl.units * 8192 < (1ll << 32), "persistent stack: buffers beyond the 4 GB reach of a buffer descriptor");
    int wgs = cus / G;                                  // every workgroup must be resident: one per CU (its LDS is the whole CU's)
    if (max_wgs > 0 && max_wgs / G < wgs) wgs = max_wgs / G;
    PWV_CHECK_ARG(wgs >= 1, "persistent stack: no workgroups");
    // short inputs: at least `min_units` units per workgroup and layer: fewer workgroups instead of ranges of one or two units
    // (default 4 = one per SIMD; measured at 16000 rows x 2 nets: 0.64 ms per forward with 4, 0.76 ms with 8)
    if (min_units <= 0) min_units = 4;
    int want = (pl.units + min_units - 1) / min_units;
    if (want < 1) want = 1;
    pl.nwg = wgs < want ? wgs : want;
    pl.per_wg = (pl.units + pl.nwg - 1) / pl.nwg;
    PWV_CHECK_ARG(pl.per_wg <= kMaxUnitsWg, "persistent stack: %d units per workgroup (max %d)", pl.per_wg, kMaxUnitsWg);
    pl.last_wg = (pl.units - 1) / pl.per_wg;
    const int reach = (dmax + 31) / 32;      // units a task looks back (RAW) / is looked back at from (WAR)
    pl.reach_wgs = (reach + pl.per_wg - 1) / pl.per_wg;
    PWV_CHECK_ARG(pl.reach_wgs <= kMaxReachWgs, "persistent stack: dilation %d reaches over %d workgroups (max %d)", dmax, pl.reach_wgs, kMaxReachWgs);
    const int grid = G * pl.nwg;
    pl.xcd_map = (pl.nwg % 8 == 0 && grid % 8 == 0) ? 1 : 0;
    // Short inputs (a handful of units per workgroup and layer: every layer is ONE unit's latency per SIMD): progress per UNIT instead of per
    // workgroup.  A range's bottom unit waits for the left neighbour's TOP units, which that workgroup computes FIRST; its workgroup word
    // appears only when its slowest unit -- its own bottom one, which waited for ITS left neighbour -- is through: with workgroup words every
    // layer of the chain costs a cross-workgroup hop (profiles/r06_short_timeline.md).  PWV_PERSIST_UNITWORDS=0 keeps the workgroup words (A/B).
    static const int unit_words_env = [] { const char* e = getenv("PWV_PERSIST_UNITWORDS"); return e ? atoi(e) : 1; }();
    pl.unit_mode = 0;      // 2: the short-input instantiation, 1: the general kernel with unit words (medium inputs)
    if (unit_words_env && reach <= kLeftN && pl.nwg > 1) {
        if (pl.per_wg <= kUnitModeMaxPerWg) pl.unit_mode = 2;      // (the launcher: and a folded layer 0, if the run starts there)
        else if (pl.per_wg <= kMediumMaxPerWg) pl.unit_mode = 1;
    }
    // the tail's layer looks back too (a stack whose LAST dilation is its largest passed the probe and failed at the launch)
    pl.tail_reach_wgs = 0;
    if (tail_q > 0) {
        PWV_CHECK_ARG(tail_q <= kMaxQ && tail_dil >= 1, "persistent stack: tail_q 1..%d, tail_dilation >= 1", kMaxQ);
        PWV_CHECK_ARG(rows * tail_q * 4 < (1ll << 32), "persistent stack: tail_out beyond the 4 GB reach of a buffer descriptor");
        pl.tail_reach_wgs = ((tail_dil + 31) / 32 + pl.per_wg - 1) / pl.per_wg;
        PWV_CHECK_ARG(pl.tail_reach_wgs <= kMaxReachWgs, "persistent stack: tail dilation %d reaches over %d workgroups (max %d)", tail_dil, pl.tail_reach_wgs, kMaxReachWgs);
    }
    return PWV_OK;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// the short-input instantiation: the plan's verdict (units per workgroup, look-back reach), and a folded layer 0 if the run starts with one (it has
// no unfolded form), and the P rows inside the 2 GB its buffer descriptor's 32-bit offsets reach.  (The workspace is sized by the plan alone.)
static int short_input_mode(const pwv_persist_args* a, const PersistPlan& pl) {
    if (pl.unit_mode != 2) return pl.unit_mode;
    const int fallback = kMediumMaxPerWg > 0 ? 1 : 0;
    if (a->x_first && !a->first_fold[0]) return fallback;
    const long long p_rows = a->cond_hop > 0 ? (long long)a->N * a->cond_frames : 1;
    if (p_rows * a->proj_row_stride * 4 >= (1ll << 31)) return fallback;
    return 2;
}

// The caller's struct, as far as the caller knows it (struct_size), in front of zeros: a client compiled against an earlier minor version
// of the header passes a shorter struct, and what lies behind it in its memory is not ours to read (a garbage `status`
// pointer would be written through on a give-up, a garbage tail_q would switch the tail on)
static int persist_args_copy(const pwv_persist_args* a, pwv_persist_args& out, const char* who) {
    PWV_CHECK_ARG(a, "%s: NULL argument", who);
    const size_t need = offsetof(pwv_persist_args, min_units_per_workgroup) + sizeof(int);
    PWV_CHECK_ARG(a->struct_size >= need, "%s: pwv_persist_args.struct_size = %zu (set it to sizeof(pwv_persist_args); at least %zu)", who,
                  (size_t)a->struct_size, need);
    memset(&out, 0, sizeof(out));
    memcpy(&out, a, a->struct_size < sizeof(out) ? a->struct_size : sizeof(out));
    return PWV_OK;
}

size_t pwv_persist_workspace_bytes(const pwv_persist_args* args) {
    PersistPlan pl;
    const int cus = device_cus();
    pwv_persist_args copy;
    if (persist_args_copy(args, copy, "pwv_persist_workspace_bytes") != PWV_OK) return 0;
    const pwv_persist_args* a = &copy;
    if (persist_plan(a->G, (long long)a->N * a->T, a->n_layers, a->dilations, cus, a->max_workgroups, a->min_units_per_workgroup, a->tail_q, a->tail_dilation, pl) != PWV_OK) return 0;
    // progress words + the abort word + the exit counter (one 256-byte line) + one arrival counter per range (the tail's affine)
    // (+ in unit mode one word per unit and net)
    return align256((size_t)a->G * pl.nwg * kProgStride * 4 + 256) + align256((size_t)pl.nwg * 4) + (pl.unit_mode ? align256((size_t)a->G * pl.units * kUnitStride * 4) : 0);
}

int pwv_persist_short_input(const pwv_persist_args* args) {
    PersistPlan pl;
    pwv_persist_args copy;
    if (persist_args_copy(args, copy, "pwv_persist_short_input") != PWV_OK) return -1;
    const pwv_persist_args* a = &copy;
    if (persist_plan(a->G, (long long)a->N * a->T, a->n_layers, a->dilations, device_cus(), a->max_workgroups, a->min_units_per_workgroup, a->tail_q, a->tail_dilation, pl) != PWV_OK) return -1;
    return short_input_mode(a, pl) == 2 ? 1 : 0;
}

int pwv_wavenet_stack_persist_f32(const pwv_persist_args* args, pwv_stream_t stream) {
    pwv_persist_args copy;
    if (int rc0 = persist_args_copy(args, copy, "pwv_wavenet_stack_persist_f32")) return rc0;
    const pwv_persist_args* a = &copy;
    PWV_CHECK_ARG(a->workspace, "pwv_wavenet_stack_persist_f32: NULL workspace");
    const int cus = device_cus();
    if (cus <= 0) return set_error(PWV_EHIP, "no HIP device");
    PersistParams p{};
    PersistPlan pl;
    int rc = persist_plan(a->G, (long long)a->N * a->T, a->n_layers, a->dilations, cus, a->max_workgroups, a->min_units_per_workgroup, a->tail_q, a->tail_dilation, pl);
    if (rc != PWV_OK) return rc;
    PWV_CHECK_ARG(a->N >= 1 && a->T >= 1, "pwv_wavenet_stack_persist_f32: bad N/T");
    PWV_CHECK_ARG(a->precision == PWV_PREC_F16X3 || a->precision == PWV_PREC_F32, "pwv_wavenet_stack_persist_f32: precision must be PWV_PREC_F16X3 or PWV_PREC_F32");
    PWV_CHECK_ARG(a->proj_row_stride % 4 == 0 && a->cond_hop >= 0, "pwv_wavenet_stack_persist_f32: bad projection arguments");
    PWV_CHECK_ARG(a->workspace_bytes >= pwv_persist_workspace_bytes(a), "pwv_wavenet_stack_persist_f32: workspace too small");
    PWV_CHECK_ARG(((uintptr_t)a->workspace & 255) == 0, "pwv_wavenet_stack_persist_f32: workspace must be 256-byte aligned");
    p.prog = (int*)a->workspace;
    p.abort = p.prog + (size_t)a->G * pl.nwg * kProgStride;
    p.exited = p.abort + 1;
    int* const pair_words = (int*)((char*)a->workspace + align256((size_t)a->G * pl.nwg * kProgStride * 4 + 256));
    p.uprog = (int*)((char*)pair_words + align256((size_t)pl.nwg * 4));
    pl.unit_mode = short_input_mode(a, pl);
    p.unit_mode = pl.unit_mode;
    p.active_wgs = a->G * (pl.last_wg + 1);
    for (int g = 0; g < a->G; ++g) {
        PWV_CHECK_ARG(a->x_ring[g] && a->packed_layers[g] && a->proj[g], "pwv_wavenet_stack_persist_f32: NULL buffer for net %d", g);
        p.ring[g] = a->x_ring[g];
        p.packed[g] = a->packed_layers[g];
        p.proj[g] = a->proj[g];
    }
    if (a->status) p.status = a->status;      // the caller's own sticky word (pwv_status_words_alloc): concurrent callers stay apart
    else PWV_CHECK_HIP(pwv_persist_status(&p.status) == PWV_OK ? hipSuccess : hipErrorUnknown);
    PWV_CHECK_ARG(a->ring_stride >= (size_t)pl.units * 2048 && a->ring_stride % 4 == 0,
                  "pwv_wavenet_stack_persist_f32: ring_stride smaller than one tile32 buffer (%lld floats)", (long long)pl.units * 2048);
    p.ring_stride = (long long)a->ring_stride;
    p.packed_stride = (long long)a->packed_layer_stride;
    p.proj_row_stride = a->proj_row_stride;
    p.G = a->G;
    p.N = a->N;
    p.T = a->T;
    p.n_layers = a->n_layers;
    p.units = pl.units;
    p.per_wg = pl.per_wg;
    p.nwg = pl.nwg;
    p.last_wg = pl.last_wg;
    p.reach_wgs = pl.reach_wgs;
    p.xcd_map = pl.xcd_map;
    // Ring stores that no other workgroup reads are plain (the line stays in the XCD's L2 for the next layer's loads) -- as long as a
    // layer's rows FIT the L2s (8 x 4 MB).  Beyond that the lines are evicted before they are re-read anyway, and what is still
    // dirty when the launch ends (dead data: the ring is scratch) is written back between this launch and the next one: all
    // write-through measures -0.8 % / -0.2 % on C3 on two boxes (82 MB per layer), -0.2 % on C4, but +5 % on C1 and level at 16000 rows
    // (4 - 8 MB per layer) (profiles/r05_ab_experiments.md, r05_h / r05_i).  Same bits either way.
    p.all_wt = (long long)a->N * a->T * a->G * 256 > (32ll << 20) ? 1 : 0;
    PWV_CHECK_ARG(a->ring_rotation >= 0 && a->ring_rotation < 3, "pwv_wavenet_stack_persist_f32: ring_rotation must be 0, 1 or 2");
    p.rot = a->ring_rotation;
    p.cond_hop = a->cond_hop;
    p.cond_offset = a->cond_offset;
    p.cond_frames = a->cond_frames;
    make_magic((unsigned)a->T, p.T_magic, p.T_shift);
    make_magic((unsigned)(a->cond_hop > 0 ? a->cond_hop : 1), p.hop_magic, p.hop_shift);
    for (int j = 0; j < a->n_layers; ++j) p.dil[j] = a->dilations[j];
    p.x_first = a->x_first;
    if (a->x_first) {
        for (int g = 0; g < a->G; ++g) {
            PWV_CHECK_ARG(a->causal_filter[g], "pwv_wavenet_stack_persist_f32: x_first needs every net's causal filter");
            p.cfilt[g] = a->causal_filter[g];
        }
        p.x_limit = a->x_limit;
        p.range_flag = a->range_flag;
        for (int g = 0; g < a->G; ++g) p.fold0[g] = a->first_fold[g];
        for (int g = 1; g < a->G; ++g)
            PWV_CHECK_ARG((p.fold0[g] == nullptr) == (p.fold0[0] == nullptr), "pwv_wavenet_stack_persist_f32: first_fold must be set for all nets or for none");
    }
    // the tail: last layer + head (+ affine) behind the run's layers
    p.tail_q = 0;
    if (a->tail_q > 0) {
        for (int g = 0; g < a->G; ++g) {
            PWV_CHECK_ARG(a->tail_layer[g] && a->tail_head[g] && a->tail_out[g], "pwv_wavenet_stack_persist_f32: NULL tail buffer for net %d", g);
            p.tail_layer[g] = a->tail_layer[g];
            p.tail_head[g] = a->tail_head[g];
            p.tail_out[g] = a->tail_out[g];
        }
        p.tail_q = a->tail_q;
        p.tail_dil = a->tail_dilation;
        p.tail_reach_wgs = pl.tail_reach_wgs;
        if (a->affine_x) {
            PWV_CHECK_ARG(a->affine_out && ((a->G == 2 && a->tail_q == 1) || (a->G == 1 && a->tail_q == 2)),
                          "pwv_wavenet_stack_persist_f32: the fused affine needs affine_out and (G = 2, tail_q = 1) or (G = 1, tail_q = 2)");
            p.affine_x = a->affine_x;
            p.affine_out = a->affine_out;
            p.pair = pair_words;
        }
    } else {
        PWV_CHECK_ARG(!a->affine_x, "pwv_wavenet_stack_persist_f32: affine_x without a tail");
    }
    (void)pair_words;
    p.trace = nullptr;
#ifdef PWV_PTRACE
    { const char* e = getenv("PWV_PTRACE_PTR"); if (e) p.trace = (long long*)strtoull(e, nullptr, 0); }
    p.trace_ev = nullptr;
    { const char* e = getenv("PWV_PTRACE_EV_PTR"); if (e) p.trace_ev = (long long*)strtoull(e, nullptr, 0); }
#endif
    hipStream_t s = (hipStream_t)stream;
    const size_t n16 = pwv_persist_workspace_bytes(a) / 16;      // (progress words, abort / exit line, pair counters)
    if (!a->workspace_clean)
        hipLaunchKernelGGL(persist_zero_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (int4*)a->workspace, n16);
    const dim3 grid(a->G * pl.nwg), block(512);
    if (a->precision == PWV_PREC_F32) {
        if (pl.unit_mode == 2) hipLaunchKernelGGL((stack_persist_kernel<true, 2>), grid, block, 0, s, p);
        else if (pl.unit_mode == 1 && kMediumMaxPerWg > 0) hipLaunchKernelGGL((stack_persist_kernel<true, kMediumMode>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((stack_persist_kernel<true, 0>), grid, block, 0, s, p);
    } else {
        if (pl.unit_mode == 2) hipLaunchKernelGGL((stack_persist_kernel<false, 2>), grid, block, 0, s, p);
        else if (pl.unit_mode == 1 && kMediumMaxPerWg > 0) hipLaunchKernelGGL((stack_persist_kernel<false, kMediumMode>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((stack_persist_kernel<false, 0>), grid, block, 0, s, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "persistent stack kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

}  // extern "C"
