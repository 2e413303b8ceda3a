// Probe (round 3): can consecutive kernels of ONE HIP stream overlap (hipExtAnyOrderLaunch = AQL packets without the
// barrier bit), is their dispatch in queue order, does the flag survive stream capture, and what does a chain of
// layer-like launches gain when per-workgroup device flags replace the queue-level kernel boundary?
//
//   T1  K_wait (128 WGs, 82 KB LDS each => 1 per CU) spins on a flag that only the NEXT kernel of the stream sets.
//       Normal launch: must time out.  Any-order launch of the setter: the waiters see the flag.
//   T2  the same pair recorded by stream capture and replayed as a graph.
//   T3  three any-order kernels of 256 WGs with random durations: is a workgroup of kernel k+1 ever started before all
//       workgroups of kernel k were started (queue-order dispatch)?
//   T4  a chain of L "layers" (128 WGs, stage 82 KB into LDS, then ~40 us of timed spinning +-spread, then publish):
//       (a) ordinary launches, (b) any-order launches where WG w of layer j+1 waits for the flags of WGs w and w-1 of layer j.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kLdsFloats = 20480;      // 80 KB: one workgroup per CU

__device__ __forceinline__ long long now100() { return __builtin_amdgcn_s_memrealtime(); }      // 100 MHz

__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(512) void k_wait(const int* flag, int* seen, long long* when, int timeout_us) {
    __shared__ float lds[kLdsFloats];
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t0 = now100();
        int s = 0;
        while (now100() - t0 < (long long)timeout_us * 100) {
            if (ld_flag(flag)) { s = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        seen[blockIdx.x] = s;
        when[blockIdx.x] = now100() - t0;
    }
    if (lds[threadIdx.x + 1] < 0) seen[0] = -1;
}

__global__ void k_set(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned hash(unsigned a, unsigned b) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0xC2B2AE3Du; x ^= x >> 13;
    return x;
}

// T3: random-duration workgroups that stamp start / end
__global__ __launch_bounds__(512) void k_rand(long long* stamps, int kidx, int base_us, int spread_us) {
    __shared__ float lds[kLdsFloats];
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t0 = now100();
        const int dur = base_us * 100 + (int)(hash(kidx, blockIdx.x) % (unsigned)(spread_us * 100 + 1));
        while (now100() - t0 < dur) __builtin_amdgcn_s_sleep(4);
        stamps[(kidx * 1024 + blockIdx.x) * 2] = t0;
        stamps[(kidx * 1024 + blockIdx.x) * 2 + 1] = now100();
    }
    if (lds[threadIdx.x + 1] < 0) stamps[0] = -1;
}

// T4: one "layer": stage 80 KB of weights, (optionally) wait for the producers' flags, work, publish
template <bool FLAGS>
__global__ __launch_bounds__(512) void k_layer(const float* weights, int* flags, int layer, int nwg, int base_us, int spread_pct,
                                               float* sink, long long* stamps) {
    __shared__ float lds[kLdsFloats];
    const float4* src = reinterpret_cast<const float4*>(weights);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < kLdsFloats / 4; i += 512) dst[i] = src[i];
    __syncthreads();
    const int w = blockIdx.x;
    long long t_ready = 0;
    if (threadIdx.x == 0) {
        if (stamps) stamps[(layer * 256 + w) * 4 + 0] = now100();
        if (FLAGS && layer > 0) {
            const int* f = flags + (layer - 1) * nwg;
            const long long t0 = now100();
            while (!(ld_flag(f + w) && (w == 0 || ld_flag(f + w - 1)))) {
                if (now100() - t0 > 2000000) break;      // 20 ms: give up (the chain then reports nonsense, never hangs)
                __builtin_amdgcn_s_sleep(2);
            }
        }
        t_ready = now100();
        if (stamps) stamps[(layer * 256 + w) * 4 + 1] = t_ready;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int dur = base_us * 100 + (int)((long long)base_us * (hash(layer, w) % 1000u) * spread_pct / 1000);
        const long long t0 = now100();
        while (now100() - t0 < dur) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    if (lds[(threadIdx.x * 7) % kLdsFloats] < 0) sink[0] = 1.f;
    if (threadIdx.x == 0) {
        if (FLAGS) __hip_atomic_store(flags + layer * nwg + w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamps) stamps[(layer * 256 + w) * 4 + 2] = now100();
    }
}

int main(int argc, char** argv) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int *flag, *seen;
    long long* when;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&seen, 4096)); CK(hipMalloc(&when, 8192));
    std::vector<int> hseen(128);
    std::vector<long long> hwhen(128);

    auto report = [&](const char* name) {
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hseen.data(), seen, 128 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hwhen.data(), when, 128 * 8, hipMemcpyDeviceToHost));
        int n = 0; long long mx = 0;
        for (int i = 0; i < 128; ++i) { n += hseen[i] == 1; mx = std::max(mx, hwhen[i]); }
        printf("%-64s %3d / 128 waiters saw the flag, longest wait %.1f us\n", name, n, mx / 100.0);
    };

    // ---- T1 ----
    for (int any = 0; any < 2; ++any) {
        CK(hipMemsetAsync(flag, 0, 4, s));
        hipLaunchKernelGGL(k_wait, dim3(128), dim3(512), 0, s, flag, seen, when, 3000);
        hipExtLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, s, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, flag);
        CK(hipGetLastError());
        report(any ? "T1 eager, setter launched with hipExtAnyOrderLaunch:" : "T1 eager, ordinary launches (must time out):");
    }
    // ---- T2: stream capture ----
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(flag, 0, 4, s));
        hipLaunchKernelGGL(k_wait, dim3(128), dim3(512), 0, s, flag, seen, when, 3000);
        hipExtLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag);
        hipError_t e = hipStreamEndCapture(s, &g);
        if (e != hipSuccess) printf("T2 capture failed: %s\n", hipGetErrorString(e));
        else {
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s));
            report("T2 captured graph with an any-order setter:");
        }
    }
    // ---- T3: dispatch order ----
    {
        long long* stamps;
        CK(hipMalloc(&stamps, 3 * 1024 * 2 * 8));
        CK(hipMemset(stamps, 0, 3 * 1024 * 2 * 8));
        for (int k = 0; k < 3; ++k)
            hipExtLaunchKernelGGL(k_rand, dim3(256), dim3(512), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, stamps, k, 30, 30);
        CK(hipStreamSynchronize(s));
        std::vector<long long> h(3 * 1024 * 2);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = h[0];
        for (int k = 0; k < 3; ++k) for (int b = 0; b < 256; ++b) t0 = std::min(t0, h[(k * 1024 + b) * 2]);
        for (int k = 0; k < 3; ++k) {
            long long smin = 1ll << 62, smax = 0, emin = 1ll << 62, emax = 0;
            int inorder = 1;
            for (int b = 0; b < 256; ++b) {
                const long long st = h[(k * 1024 + b) * 2], en = h[(k * 1024 + b) * 2 + 1];
                smin = std::min(smin, st); smax = std::max(smax, st); emin = std::min(emin, en); emax = std::max(emax, en);
                if (b && st + 50 < h[(k * 1024 + b - 1) * 2]) inorder = 0;
            }
            printf("T3 kernel %d: starts %.1f .. %.1f us, ends %.1f .. %.1f us, starts in blockIdx order (0.5 us slack): %s\n", k,
                   (smin - t0) / 100.0, (smax - t0) / 100.0, (emin - t0) / 100.0, (emax - t0) / 100.0, inorder ? "yes" : "no");
        }
        // does any WG of kernel k+1 start before the LAST start of kernel k?
        for (int k = 0; k + 1 < 3; ++k) {
            long long last_start = 0, first_next = 1ll << 62;
            for (int b = 0; b < 256; ++b) {
                last_start = std::max(last_start, h[(k * 1024 + b) * 2]);
                first_next = std::min(first_next, h[((k + 1) * 1024 + b) * 2]);
            }
            printf("T3 first start of kernel %d is %.1f us AFTER the last start of kernel %d\n", k + 1, (first_next - last_start) / 100.0, k);
        }
    }
    // ---- T4: chain ----
    {
        const int L = 60, nwg = 128;
        float *weights, *sink;
        int* flags;
        long long* stamps;
        CK(hipMalloc(&weights, kLdsFloats * 4)); CK(hipMemset(weights, 0, kLdsFloats * 4));
        CK(hipMalloc(&sink, 4));
        CK(hipMalloc(&flags, L * nwg * 4));
        CK(hipMalloc(&stamps, L * 256 * 4 * 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int spread : {0, 10, 20}) {
            for (int mode = 0; mode < 3; ++mode) {      // 0 ordinary, 1 any-order + flags, 2 ordinary + flags (cost of the flags alone)
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(flags, 0, L * nwg * 4, s));
                    CK(hipEventRecord(e0, s));
                    for (int j = 0; j < L; ++j) {
                        if (mode == 0)
                            hipLaunchKernelGGL(k_layer<false>, dim3(nwg), dim3(512), 0, s, weights, flags, j, nwg, 40, spread, sink, (long long*)nullptr);
                        else
                            hipExtLaunchKernelGGL(k_layer<true>, dim3(nwg), dim3(512), 0, s, nullptr, nullptr,
                                                  mode == 1 ? hipExtAnyOrderLaunch : 0, weights, flags, j, nwg, 40, spread, sink, (long long*)nullptr);
                    }
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                const double mean = 40.0 * (1 + spread / 200.0), mx = 40.0 * (1 + spread / 100.0);
                printf("T4 spread %2d %%: %-34s %.1f us per layer (work mean %.1f, max %.1f)\n", spread,
                       mode == 0 ? "ordinary launches" : mode == 1 ? "any-order + per-WG flags" : "ordinary launches + flags", best * 1000 / L, mean, mx);
            }
        }
    }
    return 0;
}
