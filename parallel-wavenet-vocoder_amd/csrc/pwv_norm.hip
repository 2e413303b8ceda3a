// Normalisers and the small elementwise ops of the un-fused WaveNet path, gfx950.
//   instance_normalization                       /root/reference/modules.py:274-284  (moments over the TIME axis, eps 1e-8)
//   batch_normalization at inference             /root/reference/modules.py:266      (a per-channel affine; the host folds it
//                                                 into the packed weights wherever a GEMM follows -- this file only has the
//                                                 affine for the places where no GEMM does)
//   tanh(filter) * sigmoid(gate), adds, relu     /root/reference/modules.py:236,148,157,251
// All HBM-bound elementwise / reduction kernels over channels-last [rows, C] tensors (and tile32 buffers for the affine).
#include "pwv_common.h"

namespace pwv {

// y[r, c] = act(x[r, c] * scale[c] + bias[c]); scale / bias may be NULL (1 / 0).  tile32: the buffer is in the fused
// kernels' 32-row tiled layout (index -> channel = ((i / 128) % (C/4)) * 4 + i % 4).
__global__ void channel_affine_kernel(const float* __restrict__ x, float* __restrict__ y, long long total4, int C,
                                      const float* __restrict__ scale, const float* __restrict__ bias, int tile32, int relu) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4
    if (i >= total4) return;
    const int c4 = C / 4;
    const int c = tile32 ? (int)((i / 32) % c4) * 4 : (int)(i % c4) * 4;
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = v[e];
        if (scale) t *= scale[c + e];
        if (bias) t += bias[c + e];
        v[e] = relu ? fmaxf(t, 0.f) : t;
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
}

// scalar-channel variant (C not a multiple of 4, e.g. the [N, T, 1] flow output)
__global__ void channel_affine1_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int C,
                                       const float* __restrict__ scale, const float* __restrict__ bias, int relu) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    float t = x[i];
    if (scale) t *= scale[c];
    if (bias) t += bias[c];
    y[i] = relu ? fmaxf(t, 0.f) : t;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

// out = tanh(f) * sigmoid(g)  (modules.py:236), libm-accurate forms: this is the reference-shaped slow path
__global__ void gate_kernel(const float* __restrict__ f, const float* __restrict__ g, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = tanhf(f[i]) * (1.f / (1.f + expf(-g[i])));
}

// ---- instance normalisation: per (utterance, channel) moments over time ----------------------------------------------
// Pass 1: block (chunk, n) sums x and x^2 of its time chunk for every channel in fp64 (the one-pass E[x^2] - E[x]^2 form
// is safe there) and writes them to partial[n][chunk][c][2] -- no atomics, so the result is bitwise repeatable.
// Pass 2: every block first reduces the partials of its channels (<= kInMaxChunks), then normalises its rows.
constexpr int kInMaxChunks = 512;

__global__ __launch_bounds__(256) void in_stats_kernel(const float* __restrict__ x, double* __restrict__ partial, int T, int C, int chunk_len,
                                                       int chunks) {
    __shared__ double red[2][256];
    const int n = blockIdx.y, ch = blockIdx.x;
    const int t0 = ch * chunk_len, t1 = (t0 + chunk_len < T) ? t0 + chunk_len : T;
    // thread = (time lane, channel lane): consecutive threads read consecutive channels (coalesced along C)
    const int cl = C < 256 ? C : 256;              // channels handled per sweep
    const int tl = 256 / cl > 0 ? 256 / cl : 1;    // time lanes
    const int tc = threadIdx.x % cl, tt = threadIdx.x / cl;
    for (int c0 = 0; c0 < C; c0 += cl) {
        const int c = c0 + tc;
        double s = 0.0, q = 0.0;
        if (c < C && tt < tl)
            for (int t = t0 + tt; t < t1; t += tl) {
                const double v = (double)x[((size_t)n * T + t) * C + c];
                s += v;
                q += v * v;
            }
        red[0][threadIdx.x] = s;
        red[1][threadIdx.x] = q;
        __syncthreads();
        if (tt == 0 && c < C) {
            for (int k = 1; k < tl; ++k) { s += red[0][k * cl + tc]; q += red[1][k * cl + tc]; }
            double* dst = partial + (((size_t)n * chunks + ch) * C + c) * 2;
            dst[0] = s;
            dst[1] = q;
        }
        __syncthreads();
    }
}

// Pass 1b (round 6): ONE block per utterance reduces the partials to the per-channel scale / shift (fp64, fixed order: bitwise
// repeatable) -- every block of the apply pass used to repeat this reduction serially over up to 512 partials, which was 78 us of its
// 80 at 16000 samples (profiles/r06_z_in_kernel_stats.md of the build before).  256 threads = 256 / cl k-lanes per channel.
__global__ __launch_bounds__(256) void in_finalize_kernel(const double* __restrict__ partial, float* __restrict__ ab, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int T, int C, int chunks, float eps) {
    __shared__ double red[2][256];
    const int n = blockIdx.x;
    const int cl = C < 256 ? C : 256;
    const int kl = 256 / cl > 0 ? 256 / cl : 1;
    const int tc = threadIdx.x % cl, tk = threadIdx.x / cl;
    for (int c0 = 0; c0 < C; c0 += cl) {
        const int c = c0 + tc;
        double s = 0.0, q = 0.0;
        if (c < C && tk < kl)
            for (int k = tk; k < chunks; k += kl) {
                const double* src = partial + (((size_t)n * chunks + k) * C + c) * 2;
                s += src[0];
                q += src[1];
            }
        red[0][threadIdx.x] = s;
        red[1][threadIdx.x] = q;
        __syncthreads();
        if (tk == 0 && c < C) {
            for (int k = 1; k < kl; ++k) { s += red[0][k * cl + tc]; q += red[1][k * cl + tc]; }
            const double mean = s / T;
            double var = q / T - mean * mean;               // tf.nn.moments: the biased variance
            if (var < 0.0) var = 0.0;
            const double inv = 1.0 / sqrt(var + (double)eps);     // (variance + epsilon) ** .5, modules.py:282
            const double gm = gamma ? (double)gamma[c] : 1.0;
            ab[(size_t)n * 2 * C + c] = (float)(gm * inv);
            ab[(size_t)n * 2 * C + C + c] = (float)((beta ? (double)beta[c] : 0.0) - gm * inv * mean);
        }
        __syncthreads();
    }
}

// Pass 2: y = x * scale[c] + shift[c]
__global__ __launch_bounds__(256) void in_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ ab_all,
                                                       int T, int C, int rows_per_block) {
    extern __shared__ float ab[];      // [C] scale, [C] shift
    const int n = blockIdx.y;
    for (int c = threadIdx.x; c < 2 * C; c += 256) ab[c] = ab_all[(size_t)n * 2 * C + c];
    __syncthreads();
    const int t0 = blockIdx.x * rows_per_block;
    const int t1 = t0 + rows_per_block < T ? t0 + rows_per_block : T;
    const size_t base = ((size_t)n * T + t0) * C;
    const size_t total = (size_t)(t1 - t0) * C;
    for (size_t i = threadIdx.x; i < total; i += 256) {
        const int c = (int)((base + i) % C);
        y[base + i] = fmaf(x[base + i], ab[c], ab[C + c]);
    }
}

// One layer's slice of the frame-rate projection weights: column `col` of the kernels' P layout is channel
// map(col) of [filter (0..63) | gate (64..127)], pre-multiplied by the exp2 scale of its half (pwv_layer_common.h).
//   proj_w[c, col0 + col] = k(col) * gc_{filter|gate}[c, ch]      (c < C; skipped when gc_filter == NULL)
//   proj_b[col0 + col]    = k(col) * {filter|gate}_bias[ch]       (0 when the biases are NULL)
__global__ void pack_proj_kernel(const float* __restrict__ gc_filter, const float* __restrict__ gc_gate,
                                 const float* __restrict__ filter_bias, const float* __restrict__ gate_bias, int C, int col0, int row_stride,
                                 float* __restrict__ proj_w, float* __restrict__ proj_b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (c, col), c == C is the bias row
    if (i >= (C + 1) * 128) return;
    const int c = i >> 7, col = i & 127;
    const int h = col >> 6, it = (col >> 4) & 3, r = col & 15;
    const int ch = chan_of(it, r, h);                          // 0..127 in [filter | gate]
    const bool is_gate = ch >= 64;
    const float k = is_gate ? -1.4426950408889634f : -2.8853900817779268f;
    if (c == C) {
        const float* b = is_gate ? gate_bias : filter_bias;
        proj_b[col0 + col] = b ? k * b[ch & 63] : 0.f;
    } else if (gc_filter) {
        const float* w = is_gate ? gc_gate : gc_filter;
        proj_w[(size_t)c * row_stride + col0 + col] = k * w[c * 64 + (ch & 63)];
    }
}

// Pack-time statistics of one layer's weights for the range analysis of the split-fp16 arithmetic (include/pwv_hip.h,
// "Range guard"): 8 workgroups (one per statistic), out[0..7] = max|filter|, max|gate|, max|dense|, max|skip|, max|gc_filter|, max|gc_gate|,
// max_out sum_in |dense[in, out]| + max|dense_bias|, max_out sum_in |skip[in, out]| + max|skip_bias|.
__global__ __launch_bounds__(256) void range_stats_kernel(const float* filter, const float* gate, const float* dense, const float* dense_bias,
                                                          const float* skip, const float* skip_bias, const float* gc_filter,
                                                          const float* gc_gate, int C, float* out) {
    __shared__ float red[256];
    auto block_max = [&](float v) -> float {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    auto max_abs = [&](const float* p, int n) -> float {
        float m = 0.f;
        if (p)
            for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(p[i]));
        return block_max(m);
    };
    auto colsum_max = [&](const float* w, int rows, int cols) -> float {      // w [rows, cols]: max over columns of sum |.|
        float m = 0.f;
        for (int c = threadIdx.x; c < cols; c += 256) {
            float a = 0.f;
            for (int r = 0; r < rows; ++r) a += fabsf(w[r * cols + c]);
            m = fmaxf(m, a);
        }
        return block_max(m);
    };
    float v = 0.f;      // one workgroup per statistic (blockIdx.x)
    switch (blockIdx.x) {
        case 0: v = max_abs(filter, 2 * 64 * 64); break;
        case 1: v = max_abs(gate, 2 * 64 * 64); break;
        case 2: v = max_abs(dense, 64 * 64); break;
        case 3: v = max_abs(skip, 64 * 128); break;
        case 4: v = max_abs(gc_filter, C * 64); break;
        case 5: v = max_abs(gc_gate, C * 64); break;
        case 6: v = colsum_max(dense, 64, 64) + max_abs(dense_bias, 64); break;
        default: v = colsum_max(skip, 64, 128) + max_abs(skip_bias, 128); break;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

static inline unsigned nblocks(long long total, int threads) { return (unsigned)((total + threads - 1) / threads); }

}  // namespace pwv

using namespace pwv;

extern "C" {

int pwv_channel_affine_f32(const float* x, float* y, int64_t rows, int C, const float* scale, const float* bias, int tile32, int relu,
                           pwv_stream_t stream) {
    PWV_CHECK_ARG(x && y && rows >= 0 && C >= 1, "pwv_channel_affine_f32: bad arguments");
    PWV_CHECK_ARG(!tile32 || C % 4 == 0, "pwv_channel_affine_f32: a tile32 buffer has a multiple of 4 channels");
    if (rows == 0) return PWV_OK;
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0) {
        const long long total4 = (tile32 ? (rows + 31) / 32 * 32 : rows) * (C / 4);
        hipLaunchKernelGGL(channel_affine_kernel, dim3(nblocks(total4, 256)), dim3(256), 0, s, x, y, total4, C, scale, bias, tile32, relu);
    } else {
        const long long total = rows * C;
        hipLaunchKernelGGL(channel_affine1_kernel, dim3(nblocks(total, 256)), dim3(256), 0, s, x, y, total, C, scale, bias, relu);
    }
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_add_f32(const float* a, const float* b, float* out, int64_t n, pwv_stream_t stream) {
    PWV_CHECK_ARG(a && b && out && n >= 0 && n % 4 == 0, "pwv_add_f32: bad arguments (n must be a multiple of 4)");
    if (n == 0) return PWV_OK;
    hipLaunchKernelGGL(add_kernel, dim3(nblocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)(n / 4));
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_gate_f32(const float* f, const float* g, float* out, int64_t n, pwv_stream_t stream) {
    PWV_CHECK_ARG(f && g && out && n >= 0, "pwv_gate_f32: bad arguments");
    if (n == 0) return PWV_OK;
    hipLaunchKernelGGL(gate_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, f, g, out, (long long)n);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_pack_proj_f32(const float* gc_filter, const float* gc_gate, const float* filter_bias, const float* gate_bias, int C, int layer,
                      int n_layers, float* proj_w, float* proj_b, pwv_stream_t stream) {
    PWV_CHECK_ARG(proj_b && layer >= 0 && layer < n_layers && C >= 0, "pwv_pack_proj_f32: bad arguments");
    PWV_CHECK_ARG((gc_filter == nullptr) == (gc_gate == nullptr) && (filter_bias == nullptr) == (gate_bias == nullptr),
                  "pwv_pack_proj_f32: filter / gate tensors come in pairs");
    PWV_CHECK_ARG(!gc_filter || (proj_w && C >= 1), "pwv_pack_proj_f32: proj_w / C missing");
    hipLaunchKernelGGL(pack_proj_kernel, dim3(nblocks((long long)(C + 1) * 128, 256)), dim3(256), 0, (hipStream_t)stream, gc_filter, gc_gate,
                       filter_bias, gate_bias, gc_filter ? C : 0, layer * 128, n_layers * 128, proj_w, proj_b);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

int pwv_range_stats_f32(const float* filter, const float* gate, const float* dense, const float* dense_bias, const float* skip,
                        const float* skip_bias, const float* gc_filter, const float* gc_gate, int C, float* out8, pwv_stream_t stream) {
    PWV_CHECK_ARG(filter && gate && dense && skip && out8, "pwv_range_stats_f32: NULL pointer");
    PWV_CHECK_ARG((gc_filter == nullptr) == (gc_gate == nullptr) && C >= 0, "pwv_range_stats_f32: gc tensors come in pairs");
    hipLaunchKernelGGL(range_stats_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, filter, gate, dense, dense_bias, skip, skip_bias,
                       gc_filter, gc_gate, gc_filter ? C : 0, out8);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

size_t pwv_instance_norm_workspace_bytes(int N, int T, int C) {
    if (N < 1 || T < 1 || C < 1) return 0;
    // time chunks of >= 64 rows, as many as it takes to put a block on every CU (round 6: chunks of 2048 rows left a 16000-sample
    // utterance to 8 blocks -- 165 us per call on a chip that reads the tensor in 2; profiles/r06_configs.md, bench/in)
    int chunks = (T + 63) / 64;
    if (chunks > kInMaxChunks) chunks = kInMaxChunks;
    return (size_t)N * chunks * C * 2 * sizeof(double) + (size_t)N * 2 * C * sizeof(float);      // partial sums, then scale | shift per (n, c)
}

int pwv_instance_norm_f32(const float* x, float* y, int N, int T, int C, const float* gamma, const float* beta, float eps,
                          void* workspace, size_t workspace_bytes, pwv_stream_t stream) {
    PWV_CHECK_ARG(x && y && workspace, "pwv_instance_norm_f32: NULL pointer");
    PWV_CHECK_ARG(N >= 1 && T >= 1 && C >= 1 && C <= 4096, "pwv_instance_norm_f32: bad shape N=%d T=%d C=%d", N, T, C);
    PWV_CHECK_ARG(workspace_bytes >= pwv_instance_norm_workspace_bytes(N, T, C), "pwv_instance_norm_f32: workspace too small");
    // time chunks of >= 64 rows, as many as it takes to put a block on every CU (round 6: chunks of 2048 rows left a 16000-sample
    // utterance to 8 blocks -- 165 us per call on a chip that reads the tensor in 2; profiles/r06_configs.md, bench/in)
    int chunks = (T + 63) / 64;
    if (chunks > kInMaxChunks) chunks = kInMaxChunks;
    const int chunk_len = (T + chunks - 1) / chunks;
    chunks = (T + chunk_len - 1) / chunk_len;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(in_stats_kernel, dim3(chunks, N), dim3(256), 0, s, x, (double*)workspace, T, C, chunk_len, chunks);
    float* ab = (float*)((double*)workspace + (size_t)N * chunks * C * 2);
    hipLaunchKernelGGL(in_finalize_kernel, dim3(N), dim3(256), 0, s, (const double*)workspace, ab, gamma, beta, T, C, chunks, eps);
    const int rows_per_block = 64;
    hipLaunchKernelGGL(in_apply_kernel, dim3((T + rows_per_block - 1) / rows_per_block, N), dim3(256), 2 * C * sizeof(float), s, x, y,
                       (const float*)ab, T, C, rows_per_block);
    PWV_CHECK_HIP(hipGetLastError());
    return PWV_OK;
}

}  // extern "C"
