// Persistent dataflow kernel for a run of consecutive gated-residual layers (modules.py:185-259) of one or two nets,
// split-fp16 arithmetic -- ONE launch instead of one launch per layer.
//
// Why: a per-layer launch of layer_f16x3_kernel spends ~60 % of its life in steady state; the rest is weight staging
// (82 KB per workgroup, serial at the start), the ragged start / finish of the wave population and the ~7 us between
// dependent kernels ("What bounds it").  Here the waves never stop between layers:
//
//  * The chip is treated as 8 independent XCDs.  XCD x owns 1/8 of the rows (a contiguous time range) of every net and
//    recomputes, per layer, the few units left of its range that its later layers look back to (the x[t-d] halo; whole
//    32-row units accumulated from the last layer backwards, so each layer's inputs are contained in what the layer
//    before produced).  No data ever crosses an XCD boundary inside the launch: the L2 an XCD's workgroups share is
//    the only coherence point needed.  Intermediate layers live in per-XCD "strips" (3 rotating buffers); only the
//    launch's last layer writes the ordinary full-size buffer.
//  * Inside an XCD the (layer, unit) tasks of a net are dealt round-robin, layer-major, to the waves of its workgroups:
//    wave w takes tasks w, w + NW, ...  Every dependency of a task is (almost) a whole layer-sweep old, so the per-unit
//    progress flags (global memory, L2-served, agent-scope relaxed atomics) are satisfied on the first poll in steady
//    state; they make the schedule correct, not fast.  RAW: task (j, u) needs units u, (32u-d)>>5, (32u+31-d)>>5 of layer
//    j-1 complete.  WAR: its output replaces layer j-3's in the strip ring, whose readers are layer j-2's tasks of the
//    units u+floor(d'/32), u+ceil(d'/32).
//  * Producer: plain 16-byte stores (they stay in the XCD's L2) -> the wave's next natural `s_waitcnt vmcnt(0)` -> flag.
//    Consumer: flag poll -> `sc1` 16-byte loads (bypass the CU's vector L1, which other CUs' stores never refresh; served
//    by the shared L2).  Workgroups find their XCD with s_getreg(HW_REG_XCC_ID) and take a slot from a per-XCD counter, so
//    nothing depends on the dispatcher's block -> XCD placement; every spin is bounded and reports through a sticky
//    status word in pinned host memory (the host then uses the per-layer path).
//  * Weights: both LDS halves (2 x 80 KB = all of the CU's LDS) hold the packed filter|gate + dense matrices of layers j
//    and j+1; when the last of a workgroup's 8 waves leaves layer j (a counter in global memory -- LDS is full), that wave
//    refills the half with layer j+2 by LDS-DMA while the others already compute layer j+1.  The dense bias (64 floats)
//    is read from global memory per unit.
#include "pwv_f16x3.h"

#include <cstdlib>

// cache policy of the strip stores: 0 = plain (the strips are produced and consumed through the XCD's own L2)
#ifndef PWV_PERSIST_STORE_AUX
#define PWV_PERSIST_STORE_AUX 0
#endif

namespace pwv {

constexpr int kSlot = kA1Size + kA2Size;   // floats per LDS half: filter|gate (hi+lo) + dense (hi+lo) = 81,920 B
#ifndef PWV_RING
#define PWV_RING 2
#endif
constexpr int kRing = PWV_RING;            // strip buffers per (net, XCD)
constexpr int kMaxPLayers = 32;
constexpr int kSpinLimit = 1 << 17;        // polls before a wave gives up (~0.1-0.3 s)
constexpr int kCtlWg = 64;                 // ints of control state per workgroup: [p] newest layer resident in LDS half p, [2 + j] done[j]
constexpr int kCtlXcd = 64;                // ints of control state per XCD (own cache lines): [0] workgroup slot counter, [16 + 16 g] task counter of net g
constexpr int kCtlHead = 8 * kCtlXcd;      // ints in front of the per-workgroup blocks

struct PersistParams {
    const float* x_in[PWV_MAX_NETS];       // full-size tile32: input of this launch's first layer
    float* x_out[PWV_MAX_NETS];            // full-size tile32: output of this launch's last layer
    const float* packed[PWV_MAX_NETS];     // packed layers of this launch, `packed_stride` floats apart
    const float* proj[PWV_MAX_NETS];       // P rows; this launch's first layer at column 0, layer j at 128 j
    float* strips[PWV_MAX_NETS];           // [8 XCD][kRing][strip_units * 2048 floats]
    int* flags[PWV_MAX_NETS];              // [8 XCD][strip_units] layers completed per unit
    int* ctl;
    int* status;                           // pinned host word: != 0 after a give-up
    long long packed_stride;
    int proj_row_stride;
    int G, N, T, n_layers, units, upx, strip_units, wpx;     // wpx: workgroups per (XCD, net)
    int cond_hop, cond_offset, cond_frames;
    unsigned T_magic, T_shift, hop_magic, hop_shift;
    int dil[kMaxPLayers];
    int hu[kMaxPLayers];                   // halo units of layer j: hu[last] = 0, hu[j-1] = hu[j] + ceil(dil[j] / 32)
    long long* trace;                      // -DPWV_PTRACE builds: per-wave cycle accounting (tools/persist_trace.py)
};


// -DPWV_PTRACE: every wave accumulates s_memtime cycles per phase: [0] whole loop, [1] TOP wait (vmcnt(0)), [2] RAW spins,
// [3] WAR spins, [4] leave_layer, [5] weight-ready spins, [6] units, [7] RAW spins taken, [8] first task at, [9] last task done at
#ifdef PWV_PTRACE
#define PT_DECL long long pt_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pt_t = 0; (void)pt_t; long long pt_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pt_p = 0; long long pt_top[6] = {0, 0, 0, 0, 0, 0}; long long pt_q = 0;
#define PT_TOP0() pt_q = __builtin_amdgcn_s_memtime()
#define PT_TOP(k) do { const long long n_ = __builtin_amdgcn_s_memtime(); pt_top[k] += n_ - pt_q; pt_q = n_; } while (0)
// phase stamps: pt_ph[k] accumulates the cycles between PT_PHASE(k-1) and PT_PHASE(k) (PT_PHASE0 opens an iteration)
#define PT_PHASE0() pt_p = __builtin_amdgcn_s_memtime()
#define PT_PHASE(k) do { const long long n_ = __builtin_amdgcn_s_memtime(); pt_ph[k] += n_ - pt_p; pt_p = n_; } while (0)
#define PT_BEGIN() pt_t = __builtin_amdgcn_s_memtime()
#define PT_END(k) pt_acc[k] += __builtin_amdgcn_s_memtime() - pt_t
#define PT_ADD(k, v) pt_acc[k] += (v)
#else
#define PT_DECL
#define PT_BEGIN() do {} while (0)
#define PT_END(k) do {} while (0)
#define PT_ADD(k, v) do {} while (0)
#define PT_PHASE0() do {} while (0)
#define PT_PHASE(k) do {} while (0)
#define PT_TOP0() do {} while (0)
#define PT_TOP(k) do {} while (0)
#endif

__device__ __forceinline__ int ld_word(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_word(int* p, int v, int lane) {
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifdef PWV_PTRACE
static __device__ long long* g_diag_base = nullptr;
#define g_diag (g_diag_base ? g_diag_base + (size_t)(2048 + blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 : nullptr)
#endif
__device__ __forceinline__ bool spin_ge(const int* p, int need, int* status, int code, int lane) {
    for (int k = 0; k < kSpinLimit; ++k) {
        if (ld_word(p) >= need) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    if (lane == 0) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef PWV_PTRACE
    if (lane == 0 && g_diag) { g_diag[0] = code; g_diag[1] = need; g_diag[2] = ld_word(p); g_diag[3] = (long long)p; }
#endif
    return false;
}

template <bool F32>
__global__ __launch_bounds__(512) void stack_persist_kernel(const PersistParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * kSlot];
#ifdef PWV_PTRACE
    const long long pt_entry_rt = __builtin_amdgcn_s_memrealtime();      // 100 MHz, chip-wide
    g_diag_base = p.trace;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    // ---- which XCD am I on, and which of its workgroups -------------------------------------------------------
    unsigned xcc_reg;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_reg));
    const int xcc = __builtin_amdgcn_readfirstlane((int)(xcc_reg & 7u));     // (tells the compiler it is wave-uniform)
    int* lds_i = reinterpret_cast<int*>(lds);
    if (tid == 0) lds_i[0] = __hip_atomic_fetch_add(&p.ctl[xcc * kCtlXcd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int slot = __builtin_amdgcn_readfirstlane(lds_i[0]);
    __syncthreads();
    // The dispatcher usually puts gridDim / 8 workgroups on every XCD, but it is free not to (CUs still held by the previous
    // kernel's tail, a concurrent stream): however many arrive here, they alternate between the nets and share the XCD's
    // tasks dynamically.  What must hold is that every net has at least one workgroup on every XCD: the exit census checks it.
    const int net = slot % p.G;
    const int wgi = slot / p.G;
    int* wgctl = p.ctl + kCtlHead + blockIdx.x * kCtlWg;
    int* task_ctr = p.ctl + xcc * kCtlXcd + 16 + 16 * net;     // next unclaimed task of this (XCD, net)
    const int w = wgi * 8 + wave;
    const int L = p.n_layers;
    // the per-layer tables live in two VGPRs (lane j holds entry j) and are read with v_readlane: a dynamically indexed
    // kernel argument is a scalar LOAD plus a wait each time, and the task bookkeeping at the top of every unit needs ~10
    const int v_dil = p.dil[lane & (kMaxPLayers - 1)], v_hu = p.hu[lane & (kMaxPLayers - 1)];
    auto dil_of = [&](int j) -> int { return __builtin_amdgcn_readlane(v_dil, j); };
    auto hu_of = [&](int j) -> int { return __builtin_amdgcn_readlane(v_hu, j); };
    const float* const proj_n = p.proj[net];
    const float* const packed_n = p.packed[net];
    const float* const xin_n = p.x_in[net];
    float* const xout_n = p.x_out[net];

    // ---- weights of the first two layers (LDS-DMA, packed order == LDS order) ---------------------------------------
    fill_lds_dma<kSlot / 4, 8>(lds, p.packed[net], wave, lane);
    if (L > 1) fill_lds_dma<kSlot / 4, 8>(lds + kSlot, p.packed[net] + p.packed_stride, wave, lane);
    __syncthreads();
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    // ---- this XCD's share ----------------------------------------------------------------------------------------
    const int rows = p.N * p.T;
    const int own_lo = xcc * p.upx;
    const int hi = own_lo + p.upx < p.units ? own_lo + p.upx : p.units;
    // exit census: the last workgroup of the grid to get here checks that every XCD that owns rows had >= G workgroups
    auto census = [&]() {
        __syncthreads();
        if (tid != 0) return;
        const int done_wgs = __hip_atomic_fetch_add(&p.ctl[kCtlXcd - 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done_wgs != (int)gridDim.x - 1) return;
        for (int x = 0; x < 8; ++x)
            if (x * p.upx < p.units && __hip_atomic_load(&p.ctl[x * kCtlXcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.G)
                __hip_atomic_store(p.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    if (own_lo >= hi) { census(); return; }
    const int strip_u0 = own_lo - hu_of(0);                        // unit held by strip position 0 (may be negative)
    int* flags = p.flags[net] + (size_t)xcc * p.strip_units - strip_u0;      // indexed by GLOBAL unit
    float* strip_base = p.strips[net] + (size_t)xcc * kRing * p.strip_units * 2048;
    const unsigned strip_bytes = (unsigned)p.strip_units * 8192u;
    const unsigned full_bytes = (unsigned)(((long long)rows + 31) / 32) * 8192u;

    auto lo_of = [&](int j) -> int { const int v = own_lo - hu_of(j); return v > 0 ? v : 0; };
    // task index -> (layer, unit); pure function of i (round-robin, layer-major)
    auto locate = [&](int i, int& j, int& base) -> int {
        while (j < L && i >= base + (hi - lo_of(j))) { base += hi - lo_of(j); ++j; }
        return j < L ? lo_of(j) + (i - base) : -1;
    };
    // buffers of layer j: input = full-size x_in (first layer) or strip ring (j-1) % kRing; output likewise.
    // (plain selects + readfirstlane: the descriptor must be provably wave-uniform or every buffer access becomes a
    // waterfall loop)
    auto make_rsrc = [&](const float* base, unsigned bytes) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long a = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi2 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi2 << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    auto in_rsrc = [&](int j, int& shift) -> __amdgpu_buffer_rsrc_t {
        const bool first = j == 0;
        const int slotj = first ? 0 : (j - 1) % kRing;
        shift = first ? 0 : 32 * strip_u0;
        return make_rsrc(first ? xin_n : strip_base + (size_t)slotj * p.strip_units * 2048, first ? full_bytes : strip_bytes);
    };
    auto out_rsrc = [&](int j, int& shift) -> __amdgpu_buffer_rsrc_t {
        const bool last = j == L - 1;
        shift = last ? 0 : 32 * strip_u0;
        return make_rsrc(last ? xout_n : strip_base + (size_t)(j % kRing) * p.strip_units * 2048, last ? full_bytes : strip_bytes);
    };
    // byte offset of lane (row, h)'s first 16-byte chunk inside a tile32 buffer whose row 0 is global row `shift`
    auto toff = [&](int row, int shift) -> int { const int r = row - shift; return ((r >> 5) * 2048 + h * 128 + (r & 31) * 4) * 4; };

    // x[t-d] / x[t] rows of one unit -> registers through sc1 loads (L2-served, never the CU's L1)
    auto load_x = [&](int j, int unit, float (&xb)[32], float (&xc)[32]) {
        int row, rc, n, t, shift;
        bool valid;
        unit_rows(unit, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        const __amdgpu_buffer_rsrc_t r = in_rsrc(j, shift);
        const int d = dil_of(j);
        const bool has_prev = t >= d;
        const int oc = toff(rc, shift), ob = toff(has_prev ? rc - d : rc, shift);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, oc + g * 1024, 0, 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) xc[4 * g + e] = v[e];
        }
        auto load_b = [&](bool keep) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ob + g * 1024, 0, 16));
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[4 * g + e] = keep ? v[e] : 0.f;
            }
        };
#ifdef PWV_ABL_NOXB
#pragma unroll
        for (int k = 0; k < 32; ++k) xb[k] = xc[k];
#else
        if (__all(has_prev)) load_b(true);      // wave-uniform fast path: no select behind the loads, they stay in flight
        else load_b(has_prev);
#endif
    };

    // flags a task waits for.  RAW (j >= 1): units u, (32u-d)>>5, (32u+31-d)>>5 have completed layer j-1 (count >= j).
    // WAR (kRing <= j < L-1): the readers of the ring slot it overwrites -- layer j-2's tasks of the units
    // u+floor(d'/32), u+ceil(d'/32), d' = dil[j-2], where they exist -- are done (count >= j-1).
    struct Deps { int ra, rb, rc, raw_need, wa, wb, war_need; };      // flag indices (global units) and required counts
    auto deps_of = [&](int j, int u) -> Deps {
        Deps q;
        const int d = dil_of(j);
        const int ua = (32 * u - d) >> 5, ub = (32 * u + 31 - d) >> 5;
        q.ra = u; q.rb = ua < 0 ? u : ua; q.rc = ub < 0 ? u : ub;
        q.raw_need = j;                                   // j == 0: always satisfied (flags start at 0)
        // the ring slot it overwrites held layer j - kRing's output, read by layer j - kRing + 1's tasks
        q.war_need = (j >= kRing && j < L - 1) ? j - kRing + 2 : 0;
        const int d2 = dil_of(j >= kRing - 1 ? j - kRing + 1 : 0);
        const int wa = u + (d2 >> 5), wb = u + ((d2 + 31) >> 5);
        q.wa = wa > hi - 1 ? u : wa; q.wb = wb > hi - 1 ? u : wb;
        return q;
    };
    auto wait_raw = [&](const Deps& q) -> bool {
        return spin_ge(flags + q.ra, q.raw_need, p.status, 4, lane) && spin_ge(flags + q.rb, q.raw_need, p.status, 4, lane) &&
               spin_ge(flags + q.rc, q.raw_need, p.status, 4, lane);
    };

    // Before a wave waits for anybody it publishes everything it owes: the flag of the unit it has just stored and a
    // weight refill it has issued.  (Claims can run more than a layer ahead when an XCD has few units per wave; without this
    // a wave could wait for weights whose refill it has itself not announced, or for siblings that wait for its flag.)
    int prev_u = -1, prev_j = 0;
    int dma_pending = -1;          // layer whose LDS-DMA this wave issued and has not yet published
    auto flush_owed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (prev_u >= 0) { st_word(flags + prev_u, prev_j + 1, lane); prev_u = -1; }
        if (dma_pending >= 0) { st_word(&wgctl[dma_pending & 1], dma_pending, lane); dma_pending = -1; }
    };

    // leaving layer jj: count this wave out; the LAST of the workgroup's 8 waves refills the LDS half with layer jj + 2
    auto leave_layer = [&](int jj) {
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&wgctl[2 + jj], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == 7 && jj + 2 < L) {
            if (dma_pending >= 0) flush_owed();       // (a wave that is last twice in a row: announce the earlier refill first)
            // a rolled loop: one running per-lane address (unrolled, the 80 address pairs cost 40 VGPRs at this point)
            const float* src = packed_n + (size_t)(jj + 2) * p.packed_stride + lane * 4;
            float* dst = lds + (jj & 1) * kSlot;
#pragma clang loop unroll(disable)
            for (int c = 0; c < kSlot / 256; ++c)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + c * 256), (lptr_t)(dst + c * 256), 16, 0, 0);
            dma_pending = jj + 2;
        }
    };

    // ---- tasks are claimed dynamically, in the global (layer-major) order, from one counter per (XCD, net): a wave that
    // runs slower (the low-priority half of a SIMD pair, a CU with a busier memory path) simply takes fewer of them.
    // Claims are returning atomics issued one iteration before their result is needed.
    auto claim = [&]() -> int {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(task_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;          // lane 0's value; readfirstlane at the point of use
    };
    int i = __builtin_amdgcn_readfirstlane(claim());
    int claim_v = claim();                 // the task after that
    int j = 0, base = 0;
    int u = locate(i, j, base);
    bool dead = false;
    for (int jj = 0; jj < (u >= 0 ? j : L); ++jj) leave_layer(jj);       // layers this wave has no task in
    float rxb[32], rxc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) rxb[k] = rxc[k] = 0.f;
    bool war_ok = true, need_load = true;       // need_load: the rows of the task in hand were NOT prefetched
    int cur_wa = 0, cur_wb = 0, cur_wneed = 0;      // WAR flags of the task in hand
    Deps curd{};
    if (u >= 0) {
        if (j >= 2) { flush_owed(); dead = !spin_ge(&wgctl[j & 1], j, p.status, 3, lane); }
        curd = deps_of(j, u);
        cur_wa = curd.wa; cur_wb = curd.wb; cur_wneed = curd.war_need;
        war_ok = cur_wneed == 0;
    }
    PT_DECL
#ifdef PWV_PTRACE
    const long long pt_start = __builtin_amdgcn_s_memtime();
    const long long pt_start_rt = __builtin_amdgcn_s_memrealtime();
    pt_acc[8] = pt_start;
#endif

#ifdef PWV_ABL_SHIFT
#pragma unroll
    for (int k_ = 0; k_ < PWV_ABL_SHIFT; ++k_) asm volatile("s_nop 0");      // code-placement probe: 4 bytes each
#endif
    while (u >= 0 && !dead) {
        // ---- TOP: P row, the next task's flags; then everything this wave has in flight has landed ------------------
        PT_PHASE0();
        PT_TOP0();
        int row, rc, n, t;
        bool valid;
        unit_rows(u, lane, rows, p.N, p.T, p.T_magic, p.T_shift, row, valid, rc, n, t);
        f32x16 acc[4];
        {
            int prow = 0;
            if (p.cond_hop > 0) prow = n * p.cond_frames + fast_div(t + p.cond_offset, p.hop_magic, p.hop_shift);
            const float* pr = proj_n + (size_t)prow * p.proj_row_stride + j * 128 + h * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#ifdef PWV_ABL_NOP
                    const f32x4 v = {0.f, 0.f, 0.f, 0.f}; (void)pr;
#else
                    const f32x4 v = *reinterpret_cast<const f32x4*>(pr + it * 16 + q * 4);
#endif
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[it][q * 4 + e] = v[e];
                }
        }
        PT_TOP(0);      // unit_rows + P loads issued
        int j2 = j, base2 = base;
        const int i2 = __builtin_amdgcn_readfirstlane(claim_v);      // claimed an iteration ago
        const int u2 = locate(i2, j2, base2);
        Deps nxt{};
        int f_ra = 0, f_rb = 0, f_rc = 0, f_wa = 0, f_wb = 0, f_ld = 0;
        if (u2 >= 0) nxt = deps_of(j2, u2);
#ifndef PWV_ABL_NOFLAGS
        if (u2 >= 0) {
            f_ra = __hip_atomic_load(flags + nxt.ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_rb = __hip_atomic_load(flags + nxt.rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_rc = __hip_atomic_load(flags + nxt.rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_wa = __hip_atomic_load(flags + nxt.wa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_wb = __hip_atomic_load(flags + nxt.wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f_ld = __hip_atomic_load(&wgctl[j2 & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#else
        f_ra = f_rb = f_rc = f_wa = f_wb = f_ld = 1 << 20;
#endif
        PT_TOP(1);      // claim read, locate, deps, flag loads issued
        PT_BEGIN();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PT_END(1);
        PT_TOP(2);      // the wait
        PT_ADD(6, 1);
        // the previous unit's stores have reached the L2 (and a refill this wave issued has landed): publish
        if (prev_u >= 0) { st_word(flags + prev_u, prev_j + 1, lane); prev_u = -1; }     // (once: a later re-publish would LOWER a count)
        if (dma_pending >= 0) { st_word(&wgctl[dma_pending & 1], dma_pending, lane); dma_pending = -1; }
        if (u2 >= 0) claim_v = claim();
        if (need_load) {
            // rows not prefetched (first task, or their producers were not done when the previous iteration looked): wait
            // HERE, where everything this wave has produced is published -- a wave never spins on a RAW flag while it
            // holds unpublished work, so the claim order cannot tie a knot
            PT_BEGIN();
            if (curd.raw_need > 0) dead = !wait_raw(curd);
            PT_END(2);
            PT_ADD(7, 1);
            if (dead) break;
            load_x(j, u, rxb, rxc);
        }
        PT_TOP(3);      // publish, claim, deferred loads
        const bool raw_ok2 = __builtin_amdgcn_readfirstlane(f_ra) >= nxt.raw_need && __builtin_amdgcn_readfirstlane(f_rb) >= nxt.raw_need &&
                             __builtin_amdgcn_readfirstlane(f_rc) >= nxt.raw_need;
        const bool war_ok2 = __builtin_amdgcn_readfirstlane(f_wa) >= nxt.war_need && __builtin_amdgcn_readfirstlane(f_wb) >= nxt.war_need;
        const bool ld_ok2 = j2 < 2 || __builtin_amdgcn_readfirstlane(f_ld) >= j2;

        PT_TOP(4);      // flag evaluation
        PT_PHASE(0);      // TOP: loads issued, waited, published
        const float* bdp = packed_n + (size_t)j * p.packed_stride + kBD + h * 32;
        float bdr[32];           // dense bias of this lane's 32 output channels (global memory: the LDS is full of weights)
        float o[32];
        f32x16 acc2[2];
        // the next task's rows: requested between GEMM1 and GEMM2, in flight under GEMM2 + gating + stores
        auto prefetch_next = [&]() {
            if (u2 >= 0 && raw_ok2) {
                load_x(j2, u2, rxb, rxc);
            } else {      // nothing prefetched: last task of this wave, or the next task's producers are still at work
#pragma unroll
                for (int k = 0; k < 32; ++k) rxb[k] = rxc[k] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (F32) {
            // ---- exact-fp32 arithmetic: v_mfma_f32_32x32x2_f32, the operands are the rows as loaded (pwv_layer.hip) ----
            const float* Af = lds + (j & 1) * kSlot;                 // [kA1 | kA2] of this layer's packed block
            f32x4 a[4];
            auto bx = [&](int ks) -> float { return ks < 32 ? rxb[ks] : rxc[ks - 32]; };
            PT_PHASE(1);
            a[0] = frag(Af, 0, 0, 16, 0, lane);
            a[1] = frag(Af, 0, 2, 16, 0, lane);
            gemm_groups<16, 2, 0, 2>(Af, 0, lane, acc, a, bx, [](int) {}, [&](f32x4(&n)[4]) {
                n[0] = frag(Af, 0, 1, 16, 0, lane);
                n[1] = frag(Af, 0, 3, 16, 0, lane);
            });
            PT_PHASE(2);
            gemm_groups<16, 2, 1, 2>(
                Af, 0, lane, acc, a, bx,
                [&](int g) {
                    o[g] = gate_act(acc[0][g], acc[2][g]);
                    asm volatile("" : "+v"(o[g]));   // keep the gating inside this MFMA group (no sinking)
                    if (g == 10) load_contig<8>(bdp, bdr);
                },
                [&](f32x4(&n)[4]) {
                    n[0] = frag(Af, kA1Size, 0, 8, 0, lane);
                    n[1] = frag(Af, kA1Size, 1, 8, 0, lane);
                });
            PT_PHASE(3);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[it][r] = rxc[it * 16 + r] + bdr[it * 16 + r];
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]));
            prefetch_next();
            PT_PHASE(4);
            gemm_groups<8, 2, 0, 1>(
                Af, kA1Size, lane, acc2, a, [&](int ks) -> float { return o[ks]; },
                [&](int g) {
                    if (g < 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[16 + 4 * g + e] = gate_act(acc[1][4 * g + e], acc[3][4 * g + e]);
                            asm volatile("" : "+v"(o[16 + 4 * g + e]));
                        }
                    }
                },
                [](f32x4(&)[4]) {});
        } else {
            const f16x8* A1 = reinterpret_cast<const f16x8*>(lds + (j & 1) * kSlot);
            const f16x8* A2 = reinterpret_cast<const f16x8*>(lds + (j & 1) * kSlot + kA1Size);

            f16x8 bh[8], bl[8];      // B operands: k-steps 0..3 = x[t-d], 4..7 = x[t]
            float xc[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) xc[k] = rxc[k];
            split8<0>(rxb, bh[0], bl[0]);
            split8<8>(rxb, bh[1], bl[1]);
            split8<16>(rxb, bh[2], bl[2]);
            split8<24>(rxb, bh[3], bl[3]);
            auto bxh = [&](int s) -> f16x8 { return bh[s]; };
            auto bxl = [&](int s) -> f16x8 { return bl[s]; };
            f16x8 oh[4], ol[4];
            f16x8 ah[4], al[4];

            PT_PHASE(1);      // x[t-d] split
            // ---- GEMM1, row-tile pair 0 = (F[0:32], G[0:32]); x[t] is split under its first four MFMA groups ------------
            first_frags<8, 2, 0, 2, 4>(A1, lane, ah, al);
            gemm16<8, 2, 0, 2, 4>(
                A1, lane, acc, ah, al, bxh, bxl,
                [&](int s) {
                    if (s == 0) { split8<0>(xc, bh[4], bl[4]); asm volatile("" : "+v"(bh[4]), "+v"(bl[4])); }
                    if (s == 1) { split8<8>(xc, bh[5], bl[5]); asm volatile("" : "+v"(bh[5]), "+v"(bl[5])); }
                    if (s == 2) { split8<16>(xc, bh[6], bl[6]); asm volatile("" : "+v"(bh[6]), "+v"(bl[6])); }
                    if (s == 3) { split8<24>(xc, bh[7], bl[7]); asm volatile("" : "+v"(bh[7]), "+v"(bl[7])); }
                },
                [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<8, 2, 1, 2, 4>(A1, lane, nh, nl); });
            PT_PHASE(2);      // GEMM1 pair 0
            // ---- pair 1 = (F[32:64], G[32:64]); pair 0 is gated + split under these MFMAs -----------------------------------
            gemm16<8, 2, 1, 2, 4>(
                A1, lane, acc, ah, al, bxh, bxl,
                [&](int s) {
                    o[2 * s] = gate_act(acc[0][2 * s], acc[2][2 * s]);
                    o[2 * s + 1] = gate_act(acc[0][2 * s + 1], acc[2][2 * s + 1]);
                    asm volatile("" : "+v"(o[2 * s]), "+v"(o[2 * s + 1]));
                    if (s == 3) { split8<0>(o, oh[0], ol[0]); asm volatile("" : "+v"(oh[0]), "+v"(ol[0])); }
                    if (s == 7) { split8<8>(o, oh[1], ol[1]); asm volatile("" : "+v"(oh[1]), "+v"(ol[1])); }
#ifdef PWV_ABL_NOBD
                    if (s == 5) { for (int k = 0; k < 32; ++k) bdr[k] = 0.f; }
#else
                    if (s == 5) load_contig<8>(bdp, bdr);
#endif      // lands under the last two k-steps (x[t-d]'s operands are dead by now)
                },
                [&](f16x8(&nh)[4], f16x8(&nl)[4]) { first_frags<4, 2, 0, 1, 2>(A2, lane, nh, nl); });

            PT_PHASE(3);      // GEMM1 pair 1
            // ---- GEMM2: dense 64 -> 64, accumulator starts at x[t] + dense_bias ---------------------------------------------
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[it][q * 4 + e] = xc[it * 16 + q * 4 + e] + bdr[it * 16 + q * 4 + e];
                }
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]));
            prefetch_next();      // (xc is dead from here on)
            PT_PHASE(4);      // acc2 init + prefetch issue
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
                [](f16x8(&)[4], f16x8(&)[4]) {});

        }
        PT_PHASE(5);      // GEMM2
        // ---- stores (after the readers of the ring slot they overwrite are known to be done) -------------------------------
        PT_BEGIN();
        if (!war_ok) flush_owed();
        if (!war_ok) dead = dead || !(spin_ge(flags + cur_wa, cur_wneed, p.status, 5, lane) && spin_ge(flags + cur_wb, cur_wneed, p.status, 5, lane));
        PT_END(3);
        {
            int shift;
            const __amdgpu_buffer_rsrc_t ro = out_rsrc(j, shift);
            const int oo = toff(row, shift);
#ifdef PWV_ABL_NOSTORE
            if (valid && acc2[0][0] == 1.2345e-30f) {      // keeps GEMM2 alive, never true
#else
            if (valid) {
#endif
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int it = g >> 2, q = g & 3;
                    const f32x4 v = {acc2[it][q * 4], acc2[it][q * 4 + 1], acc2[it][q * 4 + 2], acc2[it][q * 4 + 3]};
#ifdef PWV_ABL_NTSTORE
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, oo + g * 1024, 0, 2);
#else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, oo + g * 1024, 0, PWV_PERSIST_STORE_AUX);
#endif
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        PT_PHASE(6);      // stores
        // ---- move on ------------------------------------------------------------------------------------------------------
        PT_BEGIN();
        for (int jj = j; jj < (u2 >= 0 ? j2 : L); ++jj) leave_layer(jj);
        PT_END(4);
        prev_u = u; prev_j = j;
        PT_BEGIN();
        if (u2 >= 0 && j2 != j && !ld_ok2) { flush_owed(); dead = dead || !spin_ge(&wgctl[j2 & 1], j2, p.status, 3, lane); }
        PT_END(5);
        i = i2; j = j2; base = base2; u = u2; war_ok = war_ok2;
        cur_wa = nxt.wa; cur_wb = nxt.wb; cur_wneed = nxt.war_need;
        need_load = !raw_ok2;
        curd = nxt;
        PT_PHASE(7);      // leave_layer / bookkeeping
    }
    // the last unit's stores, and a refill this wave still owes its workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PWV_PTRACE
    if (p.trace && lane == 0) {
        pt_acc[9] = __builtin_amdgcn_s_memtime();
        pt_acc[0] = pt_acc[9] - pt_start;
        long long* tr = p.trace + ((size_t)blockIdx.x * 8 + wave) * 32;
        for (int k = 0; k < 8; ++k) tr[16 + k] = pt_ph[k];
        for (int k = 0; k < 5; ++k) { tr[24 + k] = pt_top[k]; }
        for (int k = 0; k < 10; ++k) tr[k] = pt_acc[k];
        tr[10] = net; tr[11] = xcc; tr[12] = w;
        tr[13] = pt_entry_rt; tr[14] = pt_start_rt; tr[15] = __builtin_amdgcn_s_memrealtime();
        tr[30] = dead ? 1 : 0; tr[31] = ((long long)j << 32) | (unsigned)u;

    }
#endif
    if (prev_u >= 0 && !dead) st_word(flags + prev_u, prev_j + 1, lane);
    if (dma_pending >= 0) st_word(&wgctl[dma_pending & 1], dma_pending, lane);
    census();
}

// every polled word starts at zero on EVERY call.  A kernel, not hipMemsetAsync: under stream capture the memset node of a
// torch-captured graph did not reset the words on replay (the replays then found every flag already satisfied and every
// task already claimed -- fast and wrong; tests/test_gpu_persist.py::test_whole_model_persistent_eager_and_graph_replay)
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

static int persist_plan(int G, long long rows, int n_layers, const int* dil, int cus, int& units, int& upx, int& strip_units,
                        int& wpx, int* hu) {
    PWV_CHECK_ARG(G >= 1 && G <= PWV_MAX_NETS, "persistent stack: G=%d out of range", G);
    PWV_CHECK_ARG(n_layers >= 2 && n_layers <= kMaxPLayers && dil, "persistent stack: 2..%d layers per launch, got %d", kMaxPLayers, n_layers);
    PWV_CHECK_ARG(rows >= 1 && rows < (1ll << 31) - 256, "persistent stack: bad N*T");
    PWV_CHECK_ARG(cus >= 8 && cus % 8 == 0 && (cus / 8) % G == 0, "persistent stack: %d CUs do not split into 8 XCDs x %d nets", cus, G);
    units = (int)((rows + 31) / 32);
    upx = (units + 7) / 8;
    wpx = cus / 8 / G;
    hu[n_layers - 1] = 0;
    for (int j = n_layers - 1; j > 0; --j) {
        PWV_CHECK_ARG(dil[j] >= 1, "persistent stack: bad dilation");
        hu[j - 1] = hu[j] + (dil[j] + 31) / 32;
    }
    PWV_CHECK_ARG(dil[0] >= 1, "persistent stack: bad dilation");
    strip_units = upx + hu[0];
    PWV_CHECK_ARG((long long)strip_units * 8192 < (1ll << 32) && (long long)units * 8192 < (1ll << 32),
                  "persistent stack: buffers beyond the 4 GB reach of a buffer descriptor");
    return PWV_OK;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t pwv_persist_workspace_bytes(int G, int N, int T, int n_layers, const int* dilations) {
    int units, upx, strip_units, wpx, hu[kMaxPLayers];
    const int cus = device_cus();
    if (persist_plan(G, (long long)N * T, n_layers, dilations, cus, units, upx, strip_units, wpx, hu) != PWV_OK) return 0;
    const size_t ctl = align256((size_t)(kCtlHead + cus * kCtlWg) * 4 + (size_t)G * 8 * strip_units * 4);
    return ctl + (size_t)G * 8 * kRing * strip_units * 2048 * 4;
}

int pwv_wavenet_stack_persist_f32(const pwv_persist_args* a, pwv_stream_t stream) {
    PWV_CHECK_ARG(a && a->workspace, "pwv_wavenet_stack_persist_f32: NULL args / workspace");
    const int cus = device_cus();
    if (cus <= 0) return set_error(PWV_EHIP, "no HIP device");
    PersistParams p{};
    int rc = persist_plan(a->G, (long long)a->N * a->T, a->n_layers, a->dilations, cus, p.units, p.upx, p.strip_units, p.wpx, p.hu);
    if (rc != PWV_OK) return rc;
    PWV_CHECK_ARG(a->N >= 1 && a->T >= 1, "pwv_wavenet_stack_persist_f32: bad N/T");
    PWV_CHECK_ARG(a->precision == PWV_PREC_F16X3 || a->precision == PWV_PREC_F32, "pwv_wavenet_stack_persist_f32: precision must be PWV_PREC_F16X3 or PWV_PREC_F32");
    PWV_CHECK_ARG(a->proj_row_stride % 4 == 0 && a->cond_hop >= 0, "pwv_wavenet_stack_persist_f32: bad projection arguments");
    PWV_CHECK_ARG(a->workspace_bytes >= pwv_persist_workspace_bytes(a->G, a->N, a->T, a->n_layers, a->dilations),
                  "pwv_wavenet_stack_persist_f32: workspace too small");
    PWV_CHECK_ARG(((uintptr_t)a->workspace & 255) == 0, "pwv_wavenet_stack_persist_f32: workspace must be 256-byte aligned");
    const size_t ctl_ints = (size_t)(kCtlHead + cus * kCtlWg);
    const size_t ctl_bytes = align256(ctl_ints * 4 + (size_t)a->G * 8 * p.strip_units * 4);
    char* ws = (char*)a->workspace;
    p.ctl = (int*)ws;
    for (int g = 0; g < a->G; ++g) {
        PWV_CHECK_ARG(a->x_in[g] && a->x_out[g] && a->packed_layers[g] && a->proj[g], "pwv_wavenet_stack_persist_f32: NULL buffer for net %d", g);
        PWV_CHECK_ARG(a->x_in[g] != a->x_out[g], "pwv_wavenet_stack_persist_f32: x_in and x_out must differ");
        p.x_in[g] = a->x_in[g];
        p.x_out[g] = a->x_out[g];
        p.packed[g] = a->packed_layers[g];
        p.proj[g] = a->proj[g];
        p.flags[g] = (int*)ws + ctl_ints + (size_t)g * 8 * p.strip_units;
        p.strips[g] = (float*)(ws + ctl_bytes) + (size_t)g * 8 * kRing * p.strip_units * 2048;
    }
    PWV_CHECK_HIP(pwv_persist_status(&p.status) == PWV_OK ? hipSuccess : hipErrorUnknown);
    p.packed_stride = (long long)a->packed_layer_stride;
    p.proj_row_stride = a->proj_row_stride;
    p.G = a->G;
    p.N = a->N;
    p.T = a->T;
    p.n_layers = a->n_layers;
    p.cond_hop = a->cond_hop;
    p.cond_offset = a->cond_offset;
    p.cond_frames = a->cond_frames;
    make_magic((unsigned)a->T, p.T_magic, p.T_shift);
    make_magic((unsigned)(a->cond_hop > 0 ? a->cond_hop : 1), p.hop_magic, p.hop_shift);
    for (int j = 0; j < a->n_layers; ++j) p.dil[j] = a->dilations[j];
    p.trace = nullptr;
#ifdef PWV_PTRACE
    { const char* e = getenv("PWV_PTRACE_PTR"); if (e) p.trace = (long long*)strtoull(e, nullptr, 0); }
#endif
    hipStream_t s = (hipStream_t)stream;
    const size_t n16 = ctl_bytes / 16;
    hipLaunchKernelGGL(persist_zero_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (int4*)ws, n16);
    if (a->precision == PWV_PREC_F32)
        hipLaunchKernelGGL(stack_persist_kernel<true>, dim3(cus), dim3(512), 0, s, p);
    else
        hipLaunchKernelGGL(stack_persist_kernel<false>, dim3(cus), dim3(512), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PWV_EHIP, "persistent stack kernel launch failed: %s", hipGetErrorString(e));
    return PWV_OK;
}

}  // extern "C"
