// probe (round 5): is a LAYER-STATIONARY pipeline feasible on MI355X?
//
// Today every CU owns a time range and walks the layers (csrc/pwv_stack_persist.hip).  The alternative keeps a layer's weights on
// one CU for the whole forward and streams the 32-row units (8 KB of fp32 residual rows) through the layers: stage s waits for unit
// u of stage s-1, loads it, computes, stores its own unit u, publishes.  What decides it is the price of the hand-over:
//   * sustained units per microsecond per stage: today's kernel retires one unit per 4.3 us per SIMD-pair of waves, i.e. a CU
//     with 8 waves takes a unit every ~0.54 us; a stage must sustain that, same-XCD (L2) and cross-XCD (fabric);
//   * the fill time of 128 stages (the forward's 120 net-layers + heads).
//
// S stages = S workgroups of W waves, one per CU (the 160 KB of LDS force that).  Wave w of stage s handles the units u = w, w + W, ...
// Hand-over = the R1 form of cdna_hip_programming.md Guideline 16: sc1 (write-through) 16-byte stores -> s_waitcnt vmcnt(0) ->
// relaxed agent-scope tag store {slot tag = u + 1}; consumer: relaxed agent-scope poll of the tag, then sc1 loads.  Ring of RING slots
// per stage with back-pressure (a producer waits until the consumer has taken the unit that used the slot before).  "work" = a spin
// of `work` ticks of the 100 MHz clock between load and store (0 = the bare hand-over; 430 = today's 4.3 us unit).  Every wait is
// bounded (50 ms) and any give-up stops the whole grid through one abort word.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o stage_handover stage_handover.hip
// run:   ./stage_handover            (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr long long kGiveUp = 5000000;      // 50 ms of the 100 MHz clock
constexpr int kSc1 = 16;

struct Params {
    float* ring;            // [S][RING][2048 floats]
    int* tags;              // [S][RING] x 32 ints (own 128-byte lines): u + 1 once unit u of stage s is in the slot
    int* taken;             // [S][RING] x 32 ints: u + 1 once stage s + 1 has loaded unit u of stage s
    int* abort;
    long long* stamps;      // [S][W][4]: start, first unit published, last unit published, ok
    int S, W, U, RING, work, same_xcd;
};

__device__ __forceinline__ bool wait_tag(const int* p, int want, int* abort_word) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0;; ++k) {
        const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(v) >= want) return true;
        if ((k & 31) == 31) {
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > kGiveUp) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

__global__ __launch_bounds__(512) void pipeline_kernel(const Params p) {
    __shared__ volatile float hog[40000];      // 160,000 B: one workgroup per CU
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) hog[0] = 0.f;
    // stage of this block: blocks of one XCD (b % 8, observed) take CONSECUTIVE stages (same_xcd) or every hop crosses XCDs
    const int b = blockIdx.x;
    const int s = p.same_xcd ? (b & 7) * (p.S >> 3) + (b >> 3) : b;
    if (s >= p.S || wave >= p.W) return;
    const size_t ring_bytes = (size_t)p.S * p.RING * 8192;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.ring, 0, (unsigned)ring_bytes, 0x00020000);
    int* my_tags = p.tags + (size_t)s * p.RING * 32;
    int* my_taken = p.taken + (size_t)s * p.RING * 32;
    int* in_tags = p.tags + (size_t)(s > 0 ? s - 1 : 0) * p.RING * 32;
    int* in_taken = p.taken + (size_t)(s > 0 ? s - 1 : 0) * p.RING * 32;
    const long long t_start = __builtin_amdgcn_s_memrealtime();
    long long t_first = 0, t_last = 0;
    bool ok = true;
    f32x4 v[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) v[g] = f32x4{(float)lane, (float)g, (float)s, 1.f};
    for (int u = wave; u < p.U; u += p.W) {
        const int slot = u % p.RING;
        if (s > 0) {
            ok = wait_tag(in_tags + slot * 32, u + 1, p.abort);
            if (!ok) break;
            const int off = ((s - 1) * p.RING + slot) * 8192 + lane * 16;
#pragma unroll
            for (int g = 0; g < 8; ++g) v[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + g * 1024, 0, kSc1));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the unit is in registers: the producer may reuse the slot
            __hip_atomic_store(in_taken + slot * 32, u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p.work > 0) {
            const long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < p.work) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = v[g] * 1.0001f + 0.5f;
        if (s + 1 < p.S) {
            // back-pressure: the consumer has taken the unit that used this slot RING units ago
            if (u >= p.RING) {
                ok = wait_tag(my_taken + slot * 32, u - p.RING + 1, p.abort);
                if (!ok) break;
            }
            const int off = (s * p.RING + slot) * 8192 + lane * 16;
#pragma unroll
            for (int g = 0; g < 8; ++g) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[g]), rs, off + g * 1024, 0, kSc1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(my_tags + slot * 32, u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        t_last = __builtin_amdgcn_s_memrealtime();
        if (u == wave) t_first = t_last;
    }
    if (lane == 0) {
        long long* st = p.stamps + ((size_t)s * 8 + wave) * 4;
        st[0] = t_start; st[1] = t_first; st[2] = t_last; st[3] = ok ? 1 : 0;
        if (v[0][0] == 123456.f || hog[0] == 7.f) st[3] = 2;      // (keeps the arithmetic and the LDS allocation alive)
    }
}

static void run(int S, int W, int U, int RING, int work, int same_xcd) {
    Params p{};
    p.S = S; p.W = W; p.U = U; p.RING = RING; p.work = work; p.same_xcd = same_xcd;
    const size_t ring_bytes = (size_t)S * RING * 8192, tag_bytes = (size_t)S * RING * 32 * 4;
    hipMalloc(&p.ring, ring_bytes); hipMalloc(&p.tags, tag_bytes); hipMalloc(&p.taken, tag_bytes);
    hipMalloc(&p.abort, 256); hipMalloc(&p.stamps, (size_t)S * 8 * 4 * 8);
    hipMemset(p.ring, 0, ring_bytes); hipMemset(p.tags, 0, tag_bytes); hipMemset(p.taken, 0, tag_bytes); hipMemset(p.abort, 0, 256);
    hipMemset(p.stamps, 0, (size_t)S * 8 * 4 * 8);
    hipDeviceSynchronize();
    const int grid = same_xcd ? ((S + 7) / 8) * 8 : S;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(pipeline_kernel, dim3(grid), dim3(512), 0, 0, p);
    hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st((size_t)S * 8 * 4);
    hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    int aborted = 0; hipMemcpy(&aborted, p.abort, 4, hipMemcpyDeviceToHost);
    long long start = 1ll << 62, first_out = 0, last_out = 0, s0_last = 0;
    for (int s = 0; s < S; ++s) for (int w = 0; w < W; ++w) start = std::min(start, st[((size_t)s * 8 + w) * 4]);
    for (int w = 0; w < W; ++w) {
        const long long* a = &st[((size_t)(S - 1) * 8 + w) * 4];
        if (w == 0) first_out = a[1];
        last_out = std::max(last_out, a[2]);
        s0_last = std::max(s0_last, st[((size_t)0 * 8 + w) * 4 + 2]);
    }
    const double fill_us = (first_out - start) / 100.0, total_us = (last_out - start) / 100.0;
    const double steady_us_per_unit = U > W ? (total_us - fill_us) / (U - 1) : 0.0;
    printf("S %3d W %d U %4d ring %2d work %4.2f us %s: kernel %.1f us, fill (unit 0 out of the last stage) %.1f us = %.3f us/stage, steady %.3f us per unit "
           "(%.2f units/us per stage, per wave one unit per %.2f us)%s%s\n",
           S, W, U, RING, work / 100.0, same_xcd ? "same-XCD" : "cross-XCD", ms * 1e3, fill_us, fill_us / S, steady_us_per_unit,
           steady_us_per_unit > 0 ? 1.0 / steady_us_per_unit : 0.0, steady_us_per_unit * W, aborted ? "  ** GAVE UP **" : "",
           err != hipSuccess ? "  ** HIP ERROR **" : "");
    hipFree(p.ring); hipFree(p.tags); hipFree(p.taken); hipFree(p.abort); hipFree(p.stamps);
}

int main() {
    // warm-up
    run(8, 1, 16, 8, 0, 1);
    for (int same = 1; same >= 0; --same) {
        // the bare hand-over, one wave per stage: latency per hop, and what a ring of 8 sustains
        run(128, 1, 256, 8, 0, same);
        // 8 waves per stage (each its own units): what a CU can take in and hand on
        run(128, 8, 512, 16, 0, same);
        // with today's unit time between load and store (4.3 us per unit and wave): does the hand-over hide behind it?
        run(128, 8, 512, 16, 430, same);
        run(128, 4, 512, 16, 430, same);
        // fill of a deeper / shallower pipeline
        run(32, 8, 256, 16, 430, same);
        run(256, 8, 512, 16, 430, same);
    }
    return 0;
}
