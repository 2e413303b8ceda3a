// Small kernels around the fused layer: generic causal_conv, the frame-rate GEMM,
// condition upsampling (repeat / crop), logistic noise, IAF affine + causal layer.
// gfx950 only.  Reference call sites are cited per kernel (file:line in /root/reference).
#include "pwv_common.h"

#include <cstring>

namespace pwv {

static thread_local char g_err[512] = {0};
char* error_buffer() { return g_err; }
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// --------------------------------------------------------------------------------------
// modules.causal_conv (modules.py:11-43), any W / Cin / Cout / dilation.
// One thread per (row, 4 output channels); x[row - shift] is a wave-broadcast read, the filter
// row read is coalesced over the output channels.  API-parity op (the fused layer kernel is
// the hot path); memory-bound for small Cin.
// --------------------------------------------------------------------------------------
__global__ void causal_conv_kernel(const float* __restrict__ x, const float* __restrict__ f, float* __restrict__ y,
                                   int N, int T, int Cin, int Cout, int W, int d) {
    const int co4 = (Cout + 3) / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * T * co4;
    if (idx >= total) return;
    const int c4 = (int)(idx % co4);
    const long long row = idx / co4;
    const int t = (int)(row % T);
    const int co = c4 * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < W; ++k) {
        const long long shift = (long long)(W - 1 - k) * d;
        if (shift > t) continue;
        const float* xr = x + (row - shift) * Cin;
        const float* fk = f + (size_t)k * Cin * Cout;
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = xr[ci];
            const float* fr = fk + (size_t)ci * Cout + co;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co + e < Cout) acc[e] = fmaf(xv, fr[e], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (co + e < Cout) y[row * Cout + co + e] = acc[e];
}

// --------------------------------------------------------------------------------------
// y[M,Nout] = act(x[M,K] @ w[K,Nout] + bias): fp32 MFMA (v_mfma_f32_32x32x2_f32).
// MFMA rows = output columns n, MFMA columns = rows m.  A workgroup owns one 128-wide column
// block: it stages w[:, n0:n0+128] once into LDS in A-fragment order ([row tile][4 k-steps][lane][4],
// one conflict-free ds_read_b128 per 4 MFMAs), then its 4 waves walk 32-row tiles keeping their x
// rows in registers (lane (m,h) holds k in [h*K/2, (h+1)*K/2)).
// models.py:110-120,128-130 and the hoisted modules.py:216-228.
// --------------------------------------------------------------------------------------
template <int KH4>  // K/2 in units of 4 floats (K = 8*KH4)
// conv_T > 0: the two-tap dilated causal convolution of modules.causal_conv (modules.py:11-43) as this GEMM: K = 2 Cin, the lower lane
// half (k < Cin, tap 0) reads row m - conv_d of x [M, Cin] (zeros left of the utterance start, t = m % conv_T < conv_d), the upper half row m
// (round 6: the composed WaveNet path's convolutions ran on the scalar causal_conv_kernel at 2.7 TFLOP/s)
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int M,
                                                     int K, int Nout, int relu, int conv_T = 0, int conv_d = 0) {
    __shared__ __attribute__((aligned(16))) float lds[4 * KH4 * 64 * 4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int kh = K / 2;
    const int n0 = blockIdx.y * 128;
    // stage the weight block: element (k, n0 + c) -> fragment slot of lane (c&31, k/kh), k-step k%kh.
    // Each thread moves float4s (4 consecutive columns of one k row); all loads are issued before the
    // first LDS write.  Nout % 4 == 0, so a float4 is either fully inside or fully outside.
    {
        constexpr int ITERS = (KH4 * 8 * 32 + 255) / 256;      // K*128/4 float4s over 256 threads
        f32x4 v[ITERS];
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tid + i * 256;
            const int k = idx >> 5, c = (idx & 31) * 4;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < K && n0 + c < Nout) v[i] = *reinterpret_cast<const f32x4*>(w + (size_t)k * Nout + n0 + c);
        }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int idx = tid + i * 256;
            const int k = idx >> 5, c = (idx & 31) * 4;
            if (k < K) {
                const int hh = k / kh, ks = k - hh * kh;
                float* dst = &lds[(((c >> 5) * KH4 + (ks >> 2)) * 64 + hh * 32 + (c & 31)) * 4 + (ks & 3)];
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[4 * e] = v[i][e];     // lanes c..c+3 are 4 floats apart
            }
        }
    }
    __syncthreads();
    const int row_tiles = (M + 31) / 32;
    for (int rt = blockIdx.x * 4 + wave; rt < row_tiles; rt += gridDim.x * 4) {
        const int m = rt * 32 + j;
        const bool mvalid = m < M;
        const int mc = mvalid ? m : M - 1;
        float xb[KH4 * 4];
        const float* xr = x + (size_t)mc * K + h * kh;
        bool keep = true;
        if (conv_T > 0) {
            keep = h == 1 || (mc % conv_T) >= conv_d;
            xr = x + (size_t)(mc - ((h == 0 && keep) ? conv_d : 0)) * kh;
        }
#pragma unroll
        for (int g = 0; g < KH4; ++g) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (4 * g < kh && keep) v = *reinterpret_cast<const f32x4*>(xr + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) xb[4 * g + e] = v[e];
        }
        f32x16 acc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + chan_of(it, r, h);
                acc[it][r] = (bias && n < Nout) ? bias[n] : 0.f;
            }
        f32x4 a[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) a[it] = *reinterpret_cast<const f32x4*>(&lds[((it * KH4) * 64 + lane) * 4]);
#pragma unroll
        for (int g = 0; g < KH4; ++g) {
            if (4 * g < kh) {
                f32x4 nx[4] = {a[0], a[1], a[2], a[3]};
                if (g + 1 < KH4) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        nx[it] = *reinterpret_cast<const f32x4*>(&lds[((it * KH4 + g + 1) * 64 + lane) * 4]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[it][e], xb[4 * g + e], acc[it], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 4; ++it) a[it] = nx[it];
            }
        }
        if (mvalid) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + 32 * it + 8 * q + 4 * h;
                    if (n < Nout) {
                        f32x4 v = {acc[it][q * 4], acc[it][q * 4 + 1], acc[it][q * 4 + 2], acc[it][q * 4 + 3]};
                        if (relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        *reinterpret_cast<f32x4*>(y + (size_t)m * Nout + n) = v;
                    }
                }
        }
    }
}

// --------------------------------------------------------------------------------------
// The same GEMM in split-fp16 arithmetic (w*x ~= wh*xh + wh*xl + wl*xh on v_mfma_f32_32x32x16_f16, fp32
// accumulate: the arithmetic of the default fused layer kernels, see pwv_layer_f16.hip) -- 5.3x fewer
// matrix-pipe cycles than the fp32 MFMA version for the same ~2^-22 products.  Used for the frame-rate
// projection P and the conditioning GEMMs when the net runs in 'f16x3' / 'f16' precision.
// A workgroup owns a 128-wide column block: w[:, n0:n0+128] is split and staged once into LDS as A fragments
// ([hi|lo][4 column tiles][NS k-steps][64 lanes] x 8 halfs: k = 16 s + 8 h + q), its 4 waves walk 32-row tiles;
// lane (m, h) loads the 8 contiguous floats x[m, 16 s + 8 h ..] of every k-step and splits them in registers.
// --------------------------------------------------------------------------------------
typedef float f32x2m __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8m(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const f32x2m v = {x[q], x[q + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f32x2m r = v - __builtin_convertvector(h, f32x2m);
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[q] = h[0];
        hi[q + 1] = h[1];
        lo[q] = l[0];
        lo[q + 1] = l[1];
    }
}

template <int NS>
__global__ __launch_bounds__(256) void linear_split_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int M,
                                                           int K, int Nout, int relu) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[2 * 4 * NS * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int n0 = blockIdx.y * 128;
    // stage: one item = (column c, k-step s, half h): 8 k values -> one hi and one lo fragment unit.
    // consecutive threads take consecutive columns (coalesced dword loads along n for each k)
    for (int item = tid; item < 128 * NS * 2; item += 256) {
        const int c = item & 127, sh = item >> 7, ss = sh >> 1, hh = sh & 1;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * ss + 8 * hh + q;
            v[q] = (k < K && n0 + c < Nout) ? w[(size_t)k * Nout + n0 + c] : 0.f;
        }
        f16x8 hi, lo;
        split8m(v, hi, lo);
        const int unit = ((c >> 5) * NS + ss) * 64 + hh * 32 + (c & 31);
        lds[unit] = hi;
        lds[4 * NS * 64 + unit] = lo;
    }
    __syncthreads();
    const f16x8* AH = lds;
    const f16x8* AL = lds + 4 * NS * 64;
    const int row_tiles = (M + 31) / 32;
    for (int rt = blockIdx.x * 4 + wave; rt < row_tiles; rt += gridDim.x * 4) {
        const int m = rt * 32 + j;
        const bool mvalid = m < M;
        const int mc = mvalid ? m : M - 1;
        f16x8 bh[NS], bl[NS];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            float v[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int k = 16 * ss + 8 * h + 4 * g;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (k < K) t = *reinterpret_cast<const f32x4*>(x + (size_t)mc * K + k);   // K % 8 == 0: whole float4s
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = t[e];
            }
            split8m(v, bh[ss], bl[ss]);
        }
        // MFMA rows = rows m (A = x fragments from registers), MFMA columns = output columns n (B = w fragments
        // from LDS): lane (j, h) then holds column n0 + 32 it + j of 16 rows, and every store instruction writes two
        // full 128-byte lines
        f32x16 acc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int n = n0 + 32 * it + j;
            const float bv = (bias && n < Nout) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[it][r] = bv;
        }
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            f16x8 ah[4], al[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                ah[it] = AH[(it * NS + ss) * 64 + lane];
                al[it] = AL[(it * NS + ss) * 64 + lane];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ss], ah[it], acc[it], 0, 0, 0);
                acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ss], ah[it], acc[it], 0, 0, 0);
                acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ss], al[it], acc[it], 0, 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int n = n0 + 32 * it + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = rt * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                float v = acc[it][r];
                if (relu) v = fmaxf(v, 0.f);
                // agent-scope relaxed store = a plain store with `sc1` (write-through): the consumer is the NEXT launch, and lines
                // that are already clean need no write-back at this kernel's end (see store_wt in pwv_layer_common.h)
                if (mr < M && n < Nout) __hip_atomic_store(&y[(size_t)mr * Nout + n], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The forward's prologue in ONE launch (round 5): the frame-rate condition c = relu(mel @ dense) (models.py:128-130, no bias), the
// frame-rate projections P = c @ [gc_filter | gc_gate of every layer of every net] + biases (hoisted modules.py:216-228,
// engine.project_all) and the range check of the mel (include/pwv_hip.h "Range guard") used to be three launches; at short inputs
// they were a third of a one-flow forward.  Every workgroup of the projection GEMM recomputes c for ITS row tiles with exactly the
// MFMA sequence of linear_split_kernel<5> (3 column tiles x 5 k-steps x 3 products: +45 MFMAs next to the tile's 60), passes it
// through a per-wave LDS tile into the operand layout (lane = row), and goes on as linear_split_kernel<5> does -- so c and P are
// bit-identical to the separate launches (tests/test_gpu_parity.py).  K0 = n_mels <= 80, K = condition channels <= 80.
constexpr int kCtStride = 84;
__global__ __launch_bounds__(256) void cond_proj_kernel(const float* __restrict__ x0, const float* __restrict__ w0, int K0,
                                                        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ frames, int M, int K, int Nout, float limit, int* flag) {
    constexpr int NS = 5;
    __shared__ __attribute__((aligned(16))) f16x8 lds[2 * 4 * NS * 64];       // this workgroup's 128 columns of the bank
    __shared__ __attribute__((aligned(16))) f16x8 lds0[2 * 4 * NS * 64];      // dense, staged as the condition GEMM stages it (n0 = 0)
    __shared__ __attribute__((aligned(16))) float ct[4][32 * kCtStride];      // per wave: c of the row tile in hand
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int n0 = blockIdx.y * 128;
    for (int item = tid; item < 128 * NS * 2; item += 256) {
        const int c = item & 127, sh = item >> 7, ss = sh >> 1, hh = sh & 1;
        float v[8], v0[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * ss + 8 * hh + q;
            v[q] = (k < K && n0 + c < Nout) ? w[(size_t)k * Nout + n0 + c] : 0.f;
            v0[q] = (k < K0 && c < K) ? w0[(size_t)k * K + c] : 0.f;
        }
        f16x8 hi, lo;
        split8m(v, hi, lo);
        const int unit = ((c >> 5) * NS + ss) * 64 + hh * 32 + (c & 31);
        lds[unit] = hi;
        lds[4 * NS * 64 + unit] = lo;
        split8m(v0, hi, lo);
        lds0[unit] = hi;
        lds0[4 * NS * 64 + unit] = lo;
    }
    __syncthreads();
    const f16x8* AH = lds;
    const f16x8* AL = lds + 4 * NS * 64;
    const f16x8* A0H = lds0;
    const f16x8* A0L = lds0 + 4 * NS * 64;
    float* const ctw = ct[wave];
    const int row_tiles = (M + 31) / 32;
    for (int rt = blockIdx.x * 4 + wave; rt < row_tiles; rt += gridDim.x * 4) {
        const int m = rt * 32 + j;
        const bool mvalid = m < M;
        const int mc = mvalid ? m : M - 1;
        f16x8 bh[NS], bl[NS];
        // ---- c = relu(mel @ dense) for the 32 rows of this tile (linear_split_kernel<5> on x0, relu = 1, bias = NULL) ----------
        bool bad = false;
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            float v[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int k = 16 * ss + 8 * h + 4 * g;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (k < K0) t = *reinterpret_cast<const f32x4*>(x0 + (size_t)mc * K0 + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[4 * g + e] = t[e];
                    bad = bad || !(fabsf(t[e]) <= limit);
                }
            }
            split8m(v, bh[ss], bl[ss]);
        }
        if (flag && bad) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        f32x16 acc0[3];
#pragma unroll
        for (int it = 0; it < 3; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[it][r] = 0.f;
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            f16x8 ah[3], al[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                ah[it] = A0H[(it * NS + ss) * 64 + lane];
                al[it] = A0L[(it * NS + ss) * 64 + lane];
            }
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                acc0[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ss], ah[it], acc0[it], 0, 0, 0);
                acc0[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ss], ah[it], acc0[it], 0, 0, 0);
                acc0[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ss], al[it], acc0[it], 0, 0, 0);
            }
        }
        // lane (j, h) holds column 32 it + j of the rows 8 (r >> 2) + 4 h + (r & 3): into the wave's LDS tile [row][column]
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int n = 32 * it + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = 8 * (r >> 2) + 4 * h + (r & 3);
                const float v = fmaxf(acc0[it][r], 0.f);
                if (n < K) {
                    ctw[ml * kCtStride + n] = v;
                    const int mr = rt * 32 + ml;
                    if (frames && blockIdx.y == 0 && mr < M) __hip_atomic_store(&frames[(size_t)mr * K + n], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (same wave writes and reads: program order through the LDS queue)
        // ---- P tile = c @ bank block + bias (linear_split_kernel<5> on c) --------------------------------------------------------
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            float v[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int k = 16 * ss + 8 * h + 4 * g;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (k < K) t = *reinterpret_cast<const f32x4*>(&ctw[j * kCtStride + k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = t[e];
            }
            split8m(v, bh[ss], bl[ss]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the tile is free for the next iteration's writes)
        f32x16 acc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int n = n0 + 32 * it + j;
            const float bv = (bias && n < Nout) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[it][r] = bv;
        }
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            f16x8 ah[4], al[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                ah[it] = AH[(it * NS + ss) * 64 + lane];
                al[it] = AL[(it * NS + ss) * 64 + lane];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ss], ah[it], acc[it], 0, 0, 0);
                acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ss], ah[it], acc[it], 0, 0, 0);
                acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ss], al[it], acc[it], 0, 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int n = n0 + 32 * it + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = rt * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (mr < M && n < Nout) __hip_atomic_store(&y[(size_t)mr * Nout + n], acc[it][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// models.py:131-133: out[n,t,:] = frames[n,(t+offset)/hop,:]
__global__ void upsample_repeat_kernel(const float* __restrict__ frames, float* __restrict__ out, int N, int t_mel,
                                       int C4, int T, int hop, int offset) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * T * C4;
    if (idx >= total) return;
    const int c = (int)(idx % C4);
    const long long row = idx / C4;
    const int t = (int)(row % T);
    const int n = (int)(row / T);
    int fr = (t + offset) / hop;
    fr = fr < t_mel ? fr : t_mel - 1;
    reinterpret_cast<f32x4*>(out)[idx] = reinterpret_cast<const f32x4*>(frames)[((size_t)n * t_mel + fr) * C4 + c];
}

// models.py:124: out[n,t,:] = in[n,t+offset,:]
__global__ void crop_time_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int T_in, int C4,
                                 int T_out, int offset) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * T_out * C4;
    if (idx >= total) return;
    const int c = (int)(idx % C4);
    const long long row = idx / C4;
    const int t = (int)(row % T_out);
    const int n = (int)(row / T_out);
    reinterpret_cast<f32x4*>(out)[idx] = reinterpret_cast<const f32x4*>(in)[((size_t)n * T_in + t + offset) * C4 + c];
}

// models.py:32-33: Logistic(0,1) sample = log u - log1p(-u); u from a splitmix64 counter hash
__device__ __forceinline__ float logistic_of_counter(unsigned long long seed, unsigned long long counter) {
    unsigned long long s = seed * 0x9E3779B97F4A7C15ull + counter;
    s += 0x9E3779B97F4A7C15ull;
    s = (s ^ (s >> 30)) * 0xBF58476D1CE4E5B9ull;
    s = (s ^ (s >> 27)) * 0x94D049BB133111EBull;
    s = s ^ (s >> 31);
    // 23 random bits -> u in [2^-24, 1 - 2^-24]: k + 0.5 is exact in fp32 for k < 2^23, so u is never 0 or 1 (with 24 bits
    // 16777215.5 rounds up to 2^24, u = 1 and z = +inf once in 2^24 samples -- found by the split-fp16 range guard)
    const float u = ((float)(unsigned)(s >> 41) + 0.5f) * (1.0f / 8388608.0f);
    return logf(u) - log1pf(-u);
}

__global__ void logistic_noise_kernel(float* __restrict__ z, long long n, unsigned long long seed,
                                      unsigned long long offset) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    z[i] = logistic_of_counter(seed, (unsigned long long)i + offset);
}

// The same sampler for a CAPTURED launch (a HIP graph replays the same kernel arguments: a counter range passed by value would
// repeat).  state = {seed, offset, blocks finished, skip} in device memory: every thread reads seed and offset, the LAST block to
// finish -- by then every block has read them -- moves the offset on by n and clears the ticket, so the next replay draws the next
// range: the stream of pwv_logistic_noise_f32(seed, offset), (seed, offset + n), ...  skip != 0: z is the caller's, nothing moves.
__global__ void logistic_noise_stream_kernel(float* __restrict__ z, long long n, unsigned long long* state) {
    // one thread reads {seed, offset, skip} with atomic loads (the last block of this very launch stores to state[1]: a plain load
    // could legally be re-materialised behind that store) and hands them to its block through LDS; the ticket is acq_rel: a block's
    // reads are ordered before its arrival, the last arriver's update behind everyone's
    __shared__ unsigned long long sh[3];
    if (threadIdx.x == 0) {
        sh[0] = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh[1] = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh[2] = __hip_atomic_load(&state[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned long long seed = sh[0], offset = sh[1];
    const bool skip = sh[2] != 0;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !skip) z[i] = logistic_of_counter(seed, (unsigned long long)i + offset);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long done = __hip_atomic_fetch_add(&state[2], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned long long)gridDim.x - 1ull) {
            if (!skip) __hip_atomic_store(&state[1], offset + (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[2], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// modules.py:59 (x = z*s + b) fused with the next flow's causal layer (modules.py:179-180):
// h_g[row, c] = sum_k x[t-(W-1-k)] * filt_g[k,0,c], written in the tile32 layout (pwv_layer_common.h).
// One thread per (block of 32 rows, net, channel quad, row in block): consecutive threads write consecutive 16 B.
struct FrontParams {
    const float* z;
    const float* s;
    const float* b;
    float* x_out;
    const float* filt[PWV_MAX_NETS];
    float* h[PWV_MAX_NETS];
    int sb_stride, G, N, T, W, R;
};

constexpr int kFrontMaxTaps = 8;         // filter widths the fused front end supports
constexpr int kFrontMaxFilt = 2048;      // W * R floats staged per workgroup

// One thread per row (blockIdx.y = net): x[t-(W-1)..t] once, then R/4 channel quads, each a 16-byte store that is
// contiguous across the 32 lanes of a tile32 block.  The filter (W*R floats) sits in LDS: every lane reads the same
// words (broadcast).  The previous thread-per-quad version spent its time in three 32-bit divisions per 16 bytes.
__global__ __launch_bounds__(256) void iaf_front_kernel(const FrontParams p) {
    __shared__ __attribute__((aligned(16))) float fl[kFrontMaxFilt];
    const int g = blockIdx.y;
    if (p.G > 0) {
        for (int i = threadIdx.x; i < p.W * p.R; i += 256) fl[i] = p.filt[g][i];
        __syncthreads();
    }
    const unsigned rows = (unsigned)p.N * (unsigned)p.T;
    const unsigned row = blockIdx.x * 256u + threadIdx.x;
    if (row >= rows) return;
    const unsigned t = row % (unsigned)p.T;
    auto xval = [&](unsigned rr) -> float {
        const float zv = p.z[rr];
        return p.s ? fmaf(zv, p.s[(size_t)rr * p.sb_stride], p.b[(size_t)rr * p.sb_stride]) : zv;
    };
    const float xcur = xval(row);
    float xv[kFrontMaxTaps];
#pragma unroll
    for (int k = 0; k < kFrontMaxTaps; ++k) {
        const unsigned shift = (unsigned)(p.W - 1 - k);      // tap k multiplies x[t - (W-1-k)], zero left of the utterance start
        xv[k] = k + 1 == p.W ? xcur : ((k < p.W && shift <= t) ? xval(row - shift) : 0.f);
    }
    if (g == 0 && p.x_out) p.x_out[row] = xcur;
    if (p.G == 0) return;
    float* out = p.h[g] + (size_t)(row >> 5) * (32u * p.R) + (row & 31u) * 4u;
    const int r4 = p.R / 4;
    for (int q = 0; q < r4; ++q) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kFrontMaxTaps; ++k) {
            if (k < p.W) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&fl[k * p.R + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv[k], w[e], acc[e]);
            }
        }
        *reinterpret_cast<f32x4*>(out + q * 128) = acc;
    }
}

// [rows, C] channels-last <-> tile32 ([block of 32 rows][C/4 quads][32 rows][4]); thread = (block, quad, row in block)
template <bool TO_TILE>
__global__ void tile32_kernel(const float* __restrict__ in, float* __restrict__ out, unsigned rows, unsigned c4) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned row = (idx / (32u * c4)) * 32u + (idx & 31u), quad = (idx >> 5) % c4;
    if (row >= rows) return;
    const size_t lin = (size_t)row * c4 * 4 + quad * 4, til = (size_t)(row >> 5) * (128u * c4) + quad * 128u + (row & 31u) * 4u;
    *reinterpret_cast<f32x4*>(out + (TO_TILE ? til : lin)) = *reinterpret_cast<const f32x4*>(in + (TO_TILE ? lin : til));
}

// range guard of the split-fp16 arithmetic (include/pwv_hip.h): one pass over a small tensor (flow input, mel)
__global__ void range_check_kernel(const float* __restrict__ x, long long n, float limit, int* flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(fabsf(x[i]) <= limit)) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static inline unsigned blocks_for(long long total, int threads) { return (unsigned)((total + threads - 1) / threads); }

}  // namespace pwv

using namespace pwv;

extern "C" {

const char* pwv_last_error(void) { return error_buffer(); }
int pwv_version(void) { return PWV_HIP_VERSION; }

int pwv_range_flag(int** flag) {
    static int* g_flag = nullptr;      // process lifetime; pinned + mapped: the same pointer is valid on host and device
    PWV_CHECK_ARG(flag, "pwv_range_flag: NULL argument");
    if (!g_flag) {
        PWV_CHECK_HIP(hipHostMalloc((void**)&g_flag, sizeof(int), hipHostMallocMapped | hipHostMallocPortable));
        *g_flag = 0;
    }
    *flag = g_flag;
    return PWV_OK;
}

int pwv_status_words_alloc(int** words) {
    PWV_CHECK_ARG(words, "pwv_status_words_alloc: NULL argument");
    int* w = nullptr;
    PWV_CHECK_HIP(hipHostMalloc((void**)&w, 2 * sizeof(int), hipHostMallocMapped | hipHostMallocPortable));
    w[0] = w[1] = 0;
    *words = w;
    return PWV_OK;
}

int pwv_status_words_free(int* words) {
    if (words) PWV_CHECK_HIP(hipHostFree(words));
    return PWV_OK;
}

int pwv_range_check_f32(const float* x, int64_t n, float limit, int* flag, pwv_stream_t stream) {
    PWV_CHECK_ARG(x && flag && n >= 0, "pwv_range_check_f32: bad arguments");
    if (n == 0) return PWV_OK;
    hipLaunchKernelGGL(range_check_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, limit, flag);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}
int pwv_device_cus(void) {
    const int c = device_cus();
    return c > 0 ? c : set_error(PWV_EHIP, "no HIP device available");
}

static int launch_linear(const float* x, const float* w, const float* bias, float* y, int M, int K, int Nout, int relu, int conv_T, int conv_d,
                         pwv_stream_t stream);

int pwv_causal_conv_f32(const float* x, const float* filt, float* y, int N, int T, int Cin, int Cout, int W,
                        int dilation, pwv_stream_t stream) {
    PWV_CHECK_ARG(x && filt && y, "pwv_causal_conv_f32: NULL pointer");
    PWV_CHECK_ARG(N >= 0 && T >= 0 && Cin >= 1 && Cout >= 1 && W >= 1 && dilation >= 1,
                  "pwv_causal_conv_f32: bad shape N=%d T=%d Cin=%d Cout=%d W=%d d=%d", N, T, Cin, Cout, W, dilation);
    PWV_CHECK_ARG(x != y, "pwv_causal_conv_f32: in-place not supported");
    const long long total = (long long)N * T * ((Cout + 3) / 4);
    if (total == 0) return PWV_OK;
    // the shapes of a WaveNet's own convolutions run on the fp32 MFMA GEMM (exact fp32 products, the k order of a GEMM instead of the
    // scalar kernel's tap-major fma chain): 1 x 1 convolutions as they are, two-tap dilated ones with the shifted row as the lower half of K
    if ((long long)N * T < (1ll << 31) && Cout % 4 == 0 && x != y) {
        if (W == 1 && Cin % 8 == 0 && Cin <= 128) return launch_linear(x, filt, nullptr, y, N * T, Cin, Cout, 0, 0, 0, stream);
        if (W == 2 && Cin % 4 == 0 && Cin <= 64) return launch_linear(x, filt, nullptr, y, N * T, 2 * Cin, Cout, 0, T, dilation, stream);
    }
    hipLaunchKernelGGL(causal_conv_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, filt, y,
                       N, T, Cin, Cout, W, dilation);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_linear_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int Nout, int relu,
                   pwv_stream_t stream) {
    PWV_CHECK_ARG(x && w && y, "pwv_linear_f32: NULL pointer");
    PWV_CHECK_ARG(M >= 0 && K >= 8 && K % 8 == 0 && K <= 128, "pwv_linear_f32: K must be a multiple of 8 in [8,128], got %d", K);
    PWV_CHECK_ARG(Nout >= 4 && Nout % 4 == 0, "pwv_linear_f32: Nout must be a multiple of 4, got %d", Nout);
    return launch_linear(x, w, bias, y, M, K, Nout, relu, 0, 0, stream);
}

static int launch_linear(const float* x, const float* w, const float* bias, float* y, int M, int K, int Nout, int relu, int conv_T, int conv_d,
                         pwv_stream_t stream) {
    if (M == 0) return PWV_OK;
    const int kh = K / 2;   // floats per lane half, loaded as float4 chunks
    // one workgroup per (row chunk, 128-column block); ~4 workgroups per CU overall
    const unsigned gy = (unsigned)((Nout + 127) / 128);
    unsigned gx = (unsigned)((M + 127) / 128);
    const unsigned cap = (1024u + gy - 1) / gy;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    dim3 grid(gx, gy), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (kh <= 32)
        hipLaunchKernelGGL((linear_kernel<8>), grid, block, 0, s, x, w, bias, y, M, K, Nout, relu, conv_T, conv_d);
    else if (kh <= 40)
        hipLaunchKernelGGL((linear_kernel<10>), grid, block, 0, s, x, w, bias, y, M, K, Nout, relu, conv_T, conv_d);
    else
        hipLaunchKernelGGL((linear_kernel<16>), grid, block, 0, s, x, w, bias, y, M, K, Nout, relu, conv_T, conv_d);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_linear_split_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int Nout, int relu,
                         pwv_stream_t stream) {
    PWV_CHECK_ARG(x && w && y, "pwv_linear_split_f32: NULL pointer");
    PWV_CHECK_ARG(M >= 0 && K >= 8 && K % 8 == 0 && K <= 128, "pwv_linear_split_f32: K must be a multiple of 8 in [8,128], got %d", K);
    PWV_CHECK_ARG(Nout >= 4 && Nout % 4 == 0, "pwv_linear_split_f32: Nout must be a multiple of 4, got %d", Nout);
    if (M == 0) return PWV_OK;
    const unsigned gy = (unsigned)((Nout + 127) / 128);
    unsigned gx = (unsigned)((M + 127) / 128);
    const unsigned cap = (1024u + gy - 1) / gy;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    dim3 grid(gx, gy), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (K <= 64)
        hipLaunchKernelGGL((linear_split_kernel<4>), grid, block, 0, s, x, w, bias, y, M, K, Nout, relu);
    else if (K <= 80)
        hipLaunchKernelGGL((linear_split_kernel<5>), grid, block, 0, s, x, w, bias, y, M, K, Nout, relu);
    else
        hipLaunchKernelGGL((linear_split_kernel<8>), grid, block, 0, s, x, w, bias, y, M, K, Nout, relu);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_cond_project_f32(const float* mel, const float* dense, int n_mels, const float* bank_w, const float* bank_b, float* frames,
                         float* P, int M, int C, int Nout, float limit, int* range_flag, pwv_stream_t stream) {
    PWV_CHECK_ARG(mel && dense && bank_w && P, "pwv_cond_project_f32: NULL pointer");
    PWV_CHECK_ARG(M >= 0 && n_mels >= 8 && n_mels % 8 == 0 && n_mels <= 80 && C >= 8 && C % 8 == 0 && C <= 80,
                  "pwv_cond_project_f32: n_mels and C must be multiples of 8 in [8, 80], got %d / %d", n_mels, C);
    PWV_CHECK_ARG(Nout >= 4 && Nout % 4 == 0, "pwv_cond_project_f32: Nout must be a multiple of 4, got %d", Nout);
    if (M == 0) return PWV_OK;
    const unsigned gy = (unsigned)((Nout + 127) / 128);
    unsigned gx = (unsigned)((M + 127) / 128);
    const unsigned cap = (1024u + gy - 1) / gy;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(cond_proj_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, mel, dense, n_mels, bank_w, bank_b, P, frames, M, C, Nout,
                       range_flag ? limit : 3.0e38f, range_flag);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_upsample_repeat_f32(const float* frames, float* out, int N, int t_mel, int C, int T, int hop, int offset,
                            pwv_stream_t stream) {
    PWV_CHECK_ARG(frames && out, "pwv_upsample_repeat_f32: NULL pointer");
    PWV_CHECK_ARG(N >= 0 && t_mel >= 1 && C >= 4 && C % 4 == 0 && T >= 0 && hop >= 1 && offset >= 0,
                  "pwv_upsample_repeat_f32: bad shape (C must be a multiple of 4)");
    PWV_CHECK_ARG(T == 0 || (T - 1 + offset) / hop < t_mel, "pwv_upsample_repeat_f32: T=%d needs more than t_mel=%d frames", T, t_mel);
    const long long total = (long long)N * T * (C / 4);
    if (total == 0) return PWV_OK;
    hipLaunchKernelGGL(upsample_repeat_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, frames,
                       out, N, t_mel, C / 4, T, hop, offset);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_crop_time_f32(const float* in, float* out, int N, int T_in, int C, int T_out, int offset, pwv_stream_t stream) {
    PWV_CHECK_ARG(in && out, "pwv_crop_time_f32: NULL pointer");
    PWV_CHECK_ARG(C >= 4 && C % 4 == 0 && offset >= 0 && T_out >= 0 && offset + T_out <= T_in,
                  "pwv_crop_time_f32: bad crop (T_in=%d T_out=%d offset=%d C=%d)", T_in, T_out, offset, C);
    const long long total = (long long)N * T_out * (C / 4);
    if (total == 0) return PWV_OK;
    hipLaunchKernelGGL(crop_time_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, N,
                       T_in, C / 4, T_out, offset);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_logistic_noise_f32(float* z, int64_t n, uint64_t seed, uint64_t offset, pwv_stream_t stream) {
    PWV_CHECK_ARG(z && n >= 0, "pwv_logistic_noise_f32: bad arguments");
    if (n == 0) return PWV_OK;
    hipLaunchKernelGGL(logistic_noise_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, z,
                       (long long)n, (unsigned long long)seed, (unsigned long long)offset);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_logistic_noise_stream_f32(float* z, int64_t n, uint64_t* state, pwv_stream_t stream) {
    PWV_CHECK_ARG(z && state && n >= 0, "pwv_logistic_noise_stream_f32: bad arguments");
    if (n == 0) return PWV_OK;
    hipLaunchKernelGGL(logistic_noise_stream_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, z, (long long)n,
                       (unsigned long long*)state);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_iaf_front_f32(const float* z, const float* s, const float* b, int sb_stride, float* x_out, int G,
                      const float* const* filt, float* const* h, int N, int T, int W, int R, pwv_stream_t stream) {
    PWV_CHECK_ARG(z, "pwv_iaf_front_f32: z is NULL");
    PWV_CHECK_ARG((s == nullptr) == (b == nullptr), "pwv_iaf_front_f32: s and b must both be set or both NULL");
    PWV_CHECK_ARG(G >= 0 && G <= PWV_MAX_NETS, "pwv_iaf_front_f32: G=%d out of range", G);
    PWV_CHECK_ARG(N >= 1 && T >= 1 && W >= 1, "pwv_iaf_front_f32: bad N/T/W");
    PWV_CHECK_ARG(G == 0 || (R >= 4 && R % 4 == 0), "pwv_iaf_front_f32: R must be a multiple of 4");
    PWV_CHECK_ARG(G > 0 || x_out, "pwv_iaf_front_f32: nothing to do");
    PWV_CHECK_ARG(sb_stride >= 1 || !s, "pwv_iaf_front_f32: sb_stride must be >= 1");
    FrontParams p{};
    p.z = z;
    p.s = s;
    p.b = b;
    p.x_out = x_out;
    p.sb_stride = sb_stride;
    p.G = G;
    p.N = N;
    p.T = T;
    p.W = W;
    p.R = R;
    for (int g = 0; g < G; ++g) {
        PWV_CHECK_ARG(filt && h && filt[g] && h[g], "pwv_iaf_front_f32: NULL filter / output for net %d", g);
        p.filt[g] = filt[g];
        p.h[g] = h[g];
    }
    PWV_CHECK_ARG((long long)N * T < (1ll << 31) - 256, "pwv_iaf_front_f32: N*T must stay below 2^31");
    PWV_CHECK_ARG(G == 0 || (W <= kFrontMaxTaps && W * R <= kFrontMaxFilt),
                  "pwv_iaf_front_f32: filter width %d / %d channels not supported (W <= %d, W*R <= %d)", W, R, kFrontMaxTaps, kFrontMaxFilt);
    PWV_CHECK_ARG(W <= kFrontMaxTaps, "pwv_iaf_front_f32: W must be <= %d", kFrontMaxTaps);
    hipLaunchKernelGGL(iaf_front_kernel, dim3(blocks_for((long long)N * T, 256), G > 0 ? G : 1), dim3(256), 0, (hipStream_t)stream, p);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

size_t pwv_tile32_floats(int64_t rows, int C) { return rows > 0 && C > 0 ? (size_t)((rows + 31) / 32) * 32 * (size_t)C : 0; }

static int tile32_convert(bool to_tile, const float* in, float* out, int64_t rows, int C, pwv_stream_t stream) {
    PWV_CHECK_ARG(in && out && in != out, "pwv tile32 conversion: NULL or aliased buffers");
    PWV_CHECK_ARG(rows >= 0 && C >= 4 && C % 4 == 0, "pwv tile32 conversion: C must be a positive multiple of 4");
    const long long total = (rows + 31) / 32 * 32 * (C / 4);
    PWV_CHECK_ARG(total < (1ll << 31), "pwv tile32 conversion: rows*C/4 must stay below 2^31");
    if (total == 0) return PWV_OK;
    if (to_tile)
        hipLaunchKernelGGL(tile32_kernel<true>, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                           (unsigned)rows, (unsigned)(C / 4));
    else
        hipLaunchKernelGGL(tile32_kernel<false>, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                           (unsigned)rows, (unsigned)(C / 4));
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_rows_to_tile32_f32(const float* in, float* out, int64_t rows, int C, pwv_stream_t stream) {
    return tile32_convert(true, in, out, rows, C, stream);
}

int pwv_tile32_to_rows_f32(const float* in, float* out, int64_t rows, int C, pwv_stream_t stream) {
    return tile32_convert(false, in, out, rows, C, stream);
}

}  // extern "C"
