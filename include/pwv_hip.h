/*
 * pwv_hip.h -- C ABI of libpwv_hip.so: the MI355X (gfx950) kernels of the IAF-WaveNet
 * student generation path of andabi/parallel-wavenet-vocoder.
 *
 * The reference has NO FFI / plugin interface for this path: it is Python calling stock
 * TensorFlow ops.  Each entry point below therefore cites the
 * reference *Python call site* (file:line in /root/reference) whose arithmetic it replaces.
 * The Python host (parallel-wavenet-vocoder_amd/{modules,models}.py) binds them with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (never retained or freed here);
 *     tensors that cross the boundary (mel, z, waveform, weights, the inputs / outputs of the
 *     standalone ops) are float32, channels-last [N, T, C], weights in TensorFlow layout
 *     [width, Cin, Cout]; the scratch activations BETWEEN the fused kernels use the 32-row tiled
 *     layout "tile32" documented below;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it
 *     (no hidden synchronisation, no internal streams or threads; re-entrant per stream);
 *   - return 0 on success, a negative PWV_E* code otherwise; pwv_last_error() returns a
 *     thread-local message for the last failing call on this thread;
 *   - "packed" buffers hold weights re-laid-out for the MFMA A-operand; their size comes
 *     from the matching pwv_*_packed_floats() call and their content from pwv_pack_*().
 */
#ifndef PWV_HIP_H_
#define PWV_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWV_OK 0
#define PWV_EINVAL (-1)      /* bad argument / unsupported shape */
#define PWV_EHIP (-2)        /* HIP runtime error (message in pwv_last_error) */

#define PWV_MAX_NETS 2       /* nets evaluated side by side in one launch (scalar, shifter) */

/* arithmetic of the fused layer / head kernels */
#define PWV_PREC_F32 0       /* v_mfma_f32_32x32x2_f32: exact fp32 fma chains */
#define PWV_PREC_F16X3 1     /* 3-term split-fp16 MFMA (hi*hi + hi*lo + lo*hi), fp32 accumulate */
#define PWV_PREC_F16 2       /* BUILD EXTENSION (the reference is fp32 only): fp16 residual stream in HBM,
                              * one fp16 MFMA product, fp32 accumulate.  Activation buffers (x_in/x_out,
                              * buf0/buf1, head `in`, `cond`) then hold fp16 tile32 blocks as written by
                              * pwv_iaf_front_f16 / pwv_cond_to_f16 and are passed through the same
                              * pointer fields; P, net outputs and the waveform chain stay fp32.
                              * No skip accumulation.  ~1e-3 of the fp32 result, not 2e-5. */

typedef void* pwv_stream_t;

const char* pwv_last_error(void);
/* PWV_HIP_VERSION of the library that was loaded: major * 100 + minor.  A change of the major number changes the layout of an
 * argument struct: 3xx = pwv_persist_args begins with `struct_size`.  A client compiled against this header checks
 * pwv_version() / 100 == PWV_HIP_VERSION / 100 once after loading the library. */
#define PWV_HIP_VERSION 301
int pwv_version(void);
/* number of compute units of the current device (grid sizing); <0 on error */
int pwv_device_cus(void);

/* ---------------------------------------------------------------------------------------
 * modules.causal_conv(value, filter_, dilation)                      modules.py:11-43
 *   y[n,t,:] = sum_k x[n, t-(W-1-k)*d, :] @ f[k],  x[t<0] = 0,  len(y) == len(x)
 *   x [N,T,Cin], f [W,Cin,Cout], y [N,T,Cout].  Any W, Cin, Cout, dilation >= 1.
 * ------------------------------------------------------------------------------------- */
int pwv_causal_conv_f32(const float* x, const float* filt, float* y,
                        int N, int T, int Cin, int Cout, int W, int dilation,
                        pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * y[M,Nout] = act(x[M,K] @ w[K,Nout] + bias)   (bias may be NULL; relu: 0/1)
 * The 1x1 convolutions of the path that are plain GEMMs: cond `dense` (models.py:128-130),
 * each conv2d_transpose stage, whose kernel width == stride makes it a per-frame GEMM
 * (models.py:110-120), and the frame-rate projection of the condition through every layer's
 * gc_filter / gc_gate (+ filter_bias / gate_bias) (modules.py:216-228, hoisted).
 * K % 8 == 0, K <= 128, Nout % 4 == 0.
 * ------------------------------------------------------------------------------------- */
int pwv_linear_f32(const float* x, const float* w, const float* bias, float* y,
                   int M, int K, int Nout, int relu, pwv_stream_t stream);

/* same contract in split-fp16 arithmetic (3 fp16 MFMA products per term, fp32 accumulate, ~2^-22
 * relative per product: the arithmetic of PWV_PREC_F16X3); 5x fewer matrix-pipe cycles */
int pwv_linear_split_f32(const float* x, const float* w, const float* bias, float* y,
                         int M, int K, int Nout, int relu, pwv_stream_t stream);
/* The prologue of a 'repeat'-conditioned forward as ONE launch (PWV_PREC_F16X3 / PWV_PREC_F16 arithmetic of pwv_linear_split_f32):
 *   frames[M, C] = relu(mel[M, n_mels] @ dense[n_mels, C])                      models.py:128-130 (no bias)
 *   P[M, Nout]   = frames @ bank_w[C, Nout] + bank_b[Nout]                      the hoisted projections of modules.py:216-228
 *   *range_flag  = 1 if any mel value is non-finite or beyond `limit`           (range guard; range_flag may be NULL)
 * bit-identical to pwv_range_check_f32 + pwv_linear_split_f32 (relu) + pwv_linear_split_f32; n_mels, C multiples of 8, <= 80;
 * `frames` may be NULL when only P is wanted. */
int pwv_cond_project_f32(const float* mel, const float* dense, int n_mels, const float* bank_w, const float* bank_b, float* frames,
                         float* P, int M, int C, int Nout, float limit, int* range_flag, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * IAFVocoder._upsample_cond, 'repeat' branch: tile + reshape + crop   models.py:131-133
 *   out[n, t, :] = frames[n, (t + offset) / hop, :],  t in [0, T)
 *   frames [N, t_mel, C] (already relu(mel @ dense)), out [N, T, C].
 * ------------------------------------------------------------------------------------- */
int pwv_upsample_repeat_f32(const float* frames, float* out, int N, int t_mel, int C,
                            int T, int hop, int offset, pwv_stream_t stream);

/* out[n, t, :] = in[n, t + offset, :]  (the crop at models.py:124) */
int pwv_crop_time_f32(const float* in, float* out, int N, int T_in, int C, int T_out, int offset,
                      pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Logistic(0,1).sample                                               models.py:32-33
 *   z = log(u) - log1p(-u),  u ~ U(0,1) from a counter-based generator keyed by
 *   (seed, element index + offset): reproducible and shardable.
 * ------------------------------------------------------------------------------------- */
int pwv_logistic_noise_f32(float* z, int64_t n, uint64_t seed, uint64_t offset, pwv_stream_t stream);
/* The same sampler for a launch that is CAPTURED into a HIP graph and replayed (arguments passed by value would repeat the counter
 * range).  state = four uint64 in DEVICE memory: {seed, offset, 0, skip}.  A launch draws z[i] = the sample of counter offset + i and
 * -- its last block to finish -- advances state[1] by n, so replay k draws what pwv_logistic_noise_f32(seed, offset + k*n) draws;
 * state[3] != 0: z is left as the caller filled it and nothing advances; state[2] is the launch's own ticket (zero between launches).
 * ONE launch per `state` in flight at a time: two launches that share a state must be ordered (one stream, or one graph replayed on
 * one stream) -- overlapping ones would share the ticket. */
int pwv_logistic_noise_stream_f32(float* z, int64_t n, uint64_t* state, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Range guard of the split-fp16 arithmetic (PWV_PREC_F16X3).  The reference computes in fp32 (models.py:81-82);
 * the split-fp16 kernels convert GEMM operands to fp16 hi/lo pairs, which keeps fp32's ~22-bit products but has
 * fp16's EXPONENT range: an operand beyond 65504 would become inf where fp32 is fine.  The host bounds every GEMM
 * operand from the weights (pack time) and from two run-time maxima -- the flow input and the mel input -- which
 * the kernels check against the limits the host derived:
 *   pwv_range_flag(&p)      process-wide sticky int32 in pinned, device-visible host memory (0 = in range);
 *                           the host reads it after any synchronisation, no device -> host copy needed.
 *   pwv_range_check_f32     sets *flag = 1 when any x[i] is non-finite or |x[i]| > limit.
 * A raised flag means: the result of that forward is not trustworthy in PWV_PREC_F16X3 -- rerun it in
 * PWV_PREC_F32 (the Python host raises PwvRangeError / reruns).
 * ------------------------------------------------------------------------------------- */
int pwv_range_flag(int** flag);
/* A caller's OWN pair of sticky words in pinned, device-visible host memory (both zero): words[0] for pwv_persist_args.status,
 * words[1] as the range flag of any entry point that takes one.  One pair per (device, stream) or per thread keeps concurrent
 * callers apart (the process-wide words above are shared by everybody who does not bring their own).  Free with
 * pwv_status_words_free when no launch that was handed them can still be running. */
int pwv_status_words_alloc(int** words);
int pwv_status_words_free(int* words);
int pwv_range_check_f32(const float* x, int64_t n, float limit, int* flag, pwv_stream_t stream);
/* pack-time statistics of one layer (R = D = 64, S = 128; TF layouts) for the host's bound on the residual stream:
 * out8 = max|filter|, max|gate|, max|dense|, max|skip|, max|gc_filter|, max|gc_gate| (0 when NULL),
 *        max_out sum_in |dense| + max|dense_bias|, max_out sum_in |skip| + max|skip_bias|;  C = rows of gc_* */
int pwv_range_stats_f32(const float* filter, const float* gate, const float* dense, const float* dense_bias, const float* skip,
                        const float* skip_bias, const float* gc_filter, const float* gc_gate, int C, float* out8, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * "tile32": the layout of every [rows = N*T, C] ACTIVATION buffer the fused kernels below read or
 * write (residual stream C = 64, skip sums C = 128, per-sample condition C = 80).  Rows are stored
 * in blocks of 32; block u holds rows 32u..32u+31 as [C/4 channel quads][32 rows][4 floats]:
 *     float index of (row, c) = (row/32)*32*C + (c/4)*128 + (row%32)*4 + c%4
 * so that a wavefront (32 rows, lane (t,h) owning channel quads 2g+h) moves 1 KB of contiguous
 * memory per vector load / store.  Buffers hold pwv_tile32_floats(rows, C) floats (whole blocks).
 * These are scratch buffers between pwv_iaf_front_f32 and pwv_wavenet_head_f32; the converters are
 * for callers that bring their own channels-last tensors (multi-channel input, per-sample condition).
 * ------------------------------------------------------------------------------------- */
size_t pwv_tile32_floats(int64_t rows, int C);
int pwv_rows_to_tile32_f32(const float* in, float* out, int64_t rows, int C, pwv_stream_t stream);
int pwv_tile32_to_rows_f32(const float* in, float* out, int64_t rows, int C, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * LinearIAFLayer affine + the next flow's causal layer, fused.
 *   x[r] = z[r]*s[r*sb_stride] + b[r*sb_stride]          modules.py:59   (x = z if s == NULL)
 *   h_g[n,t,:] = sum_k x[n, t-(W-1-k), 0] * filt_g[k,0,:]  modules.py:179-180 (no bias)
 * z [N*T]; s, b strided views of the previous flow's net outputs; x_out [N*T] (may be NULL
 * when s == NULL); for g < G: filt[g] is [W,1,R], h[g] is a tile32 buffer of N*T rows x R channels
 * (see below).  G may be 0 (affine only).  W <= 8 and W*R <= 2048 (the filter is staged in LDS).
 * ------------------------------------------------------------------------------------- */
int pwv_iaf_front_f32(const float* z, const float* s, const float* b, int sb_stride,
                      float* x_out, int G, const float* const* filt, float* const* h,
                      int N, int T, int W, int R, pwv_stream_t stream);

/* PWV_PREC_F16 front end: same arithmetic as pwv_iaf_front_f32 (R == 64, G >= 1) but h16[g] is an fp16
 * tile32 buffer: block u = rows 32u..32u+31 as [8 chunks][32 rows][8 halfs]; chunk s*2+h, half q holds
 * channel 16s + 8(q>>2) + 4h + (q&3) (the order the MFMA B operand consumes);
 * pwv_tile32_floats(rows, 64) HALFS. */
int pwv_iaf_front_f16(const float* z, const float* s, const float* b, int sb_stride,
                      float* x_out, int G, const float* const* filt, void* const* h16,
                      int N, int T, int W, int R, pwv_stream_t stream);

/* PWV_PREC_F16: per-sample condition [N,T,80] fp32 channels-last (models.py:110-124) -> fp16 tile32
 * (10 chunks per row, same channel order), pwv_tile32_floats(rows, 80) halfs */
int pwv_cond_to_f16(const float* cond, void* out16, int N, int T, int C, pwv_stream_t stream);

/* PWV_PREC_F16X3: the same conversion plus the `lo` plane fp16(v - fp16(v)) stored behind the `hi` plane
 * (2 x pwv_tile32_floats(rows, 80) halfs in total): the split-fp16 layer kernel reads both operands of its
 * condition GEMM from these planes (the condition is constant over the 120 net-layers of a forward, so it is
 * split once instead of 120 times) */
int pwv_cond_split_f16(const float* cond, void* out16, int N, int T, int C, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused gated-residual layer: WaveNet._create_dilation_layer          modules.py:185-259
 * for R = D = 64, S = 128, filter_width 2 (hparams/default.yaml:22-25), G <= 2 nets per launch.
 *
 *   F‖G = [x[t-d] ‖ x[t]] @ [[filter0‖gate0],[filter1‖gate1]] + P[frame(t)]   (+ cond[t] @ gc)
 *   o   = tanh(F) * sigmoid(G)                                                  modules.py:236
 *   out = x[t] + o @ dense + dense_bias        (out_mode PWV_OUT_RESIDUAL)     modules.py:239-251
 *   out = o                                    (out_mode PWV_OUT_GATED: last layer -> head)
 *   skip (+)= o @ skip_w + skip_bias           (when skip != NULL)              modules.py:243-250
 *
 * P is the frame-rate projection of the condition through gc_filter‖gc_gate plus
 * filter_bias‖gate_bias, in the kernel's column order (pwv_proj_column_map); the row used by
 * sample t of utterance n is  n*cond_frames + (t + cond_offset)/cond_hop  (cond_hop == 0:
 * always row 0, i.e. biases only / no conditioning).  `cond` (per-sample condition, 80 channels,
 * transposed-conv upsampling) is NULL in hoisted mode; its form depends on `precision`: fp32 tile32
 * (PWV_PREC_F32), pwv_cond_split_f16 planes (PWV_PREC_F16X3), pwv_cond_to_f16 blocks (PWV_PREC_F16).
 * ------------------------------------------------------------------------------------- */
#define PWV_OUT_RESIDUAL 0
#define PWV_OUT_GATED 1

/* floats of one layer's packed buffer; with_skip / with_cond select optional sections */
size_t pwv_layer_packed_floats(int with_skip, int cond_channels);

/* gather TF-layout weights (device) into the packed layout (device).
 * filter, gate [2,64,64]; dense [1,64,64]; dense_bias [64] or NULL;
 * skip [1,64,128] + skip_bias [128]/NULL when with_skip; gc_filter, gc_gate [1,C,64] when
 * cond_channels > 0 (per-sample conditioning). */
int pwv_pack_layer_f32(const float* filter, const float* gate, const float* dense,
                       const float* dense_bias, const float* skip, const float* skip_bias,
                       const float* gc_filter, const float* gc_gate, int with_skip,
                       int cond_channels, int precision, float* packed, pwv_stream_t stream);

/* column order of P: map[col] = channel index into [filter(0..63) ‖ gate(64..127)] */
int pwv_proj_column_map(int* map128);

/* One layer's 128 columns of the frame-rate projection operands, packed on the device:
 *   proj_w [C, 128*n_layers] (+ column 128*layer): gc_filter ‖ gc_gate [1,C,64] each, in the P column order, pre-multiplied by
 *   the exp2 scales of the gate (-2 log2 e for filter columns, -log2 e for gate columns); proj_b [128*n_layers]: the same
 *   for filter_bias ‖ gate_bias [64] (zeros when NULL).  gc_* may be NULL (no conditioning: only proj_b is written).
 *   P = frames @ proj_w + proj_b is then one pwv_linear_*_f32 call per net (modules.py:216-228, hoisted to frame rate). */
int pwv_pack_proj_f32(const float* gc_filter, const float* gc_gate, const float* filter_bias, const float* gate_bias, int C,
                      int layer, int n_layers, float* proj_w, float* proj_b, pwv_stream_t stream);

typedef struct pwv_layer_args {
    int G;                                 /* nets in this launch (1 or 2) */
    const float* x_in[PWV_MAX_NETS];       /* tile32, N*T rows x 64 */
    float* x_out[PWV_MAX_NETS];            /* tile32, N*T rows x 64 */
    const float* packed[PWV_MAX_NETS];     /* pwv_pack_layer_f32 output */
    const float* proj[PWV_MAX_NETS];       /* P rows for THIS layer (128 floats each) */
    int proj_row_stride;                   /* floats between consecutive P rows */
    const float* cond;                     /* per-sample condition in the form `precision` wants, or NULL */
    int cond_channels;                     /* 0 or 80 */
    float* skip[PWV_MAX_NETS];             /* tile32, N*T rows x 128: accumulators, or NULL */
    int skip_init;                         /* 1: skip = ..., 0: skip += ... */
    int N, T, dilation;
    int cond_hop, cond_offset, cond_frames;
    int out_mode;                          /* PWV_OUT_RESIDUAL / PWV_OUT_GATED */
    int precision;                         /* PWV_PREC_* */
    int max_workgroups;                    /* 0 = one per CU */
    /* Layer 0 of a scalar-input net without a materialised causal layer (PWV_PREC_F16X3 / PWV_PREC_F32, no skip accumulation,
     * filter width 2): when x_first != NULL, x_in is ignored and the kernel evaluates
     *     h[t] = x_first[t-1] * causal_filter[0,0,:] + x_first[t] * causal_filter[1,0,:]      modules.py:179-180
     * for rows t and t-d itself, with the operations (and bits) of pwv_iaf_front_f32: 4 B instead of 768 B of
     * traffic per sample for this layer and no front launch.  x_first is [N*T] float32 (the flow's input). */
    const float* x_first;
    const float* causal_filter[PWV_MAX_NETS];   /* [2,1,64] each */
    /* optional, with x_first: pwv_pack_first_fold_f16x3's (PWV_PREC_F16X3, PWV_PREC_F16) / _f32's (PWV_PREC_F32) output per net (all or none).  Layer 0's
     * filter|gate convolution then runs on the four scalars x[t-d-1], x[t-d], x[t-1], x[t] themselves -- one MFMA k-step
     * instead of eight: the same function (h[t] is linear in x[t-1], x[t]), rounded differently: within the path's
     * tolerance of the unfolded form, not bit-identical to it.  pwv_persist_args.first_fold does exactly the same. */
    const float* first_fold[PWV_MAX_NETS];
    /* The LAST layer with the post-processing head fused behind it (PWV_PREC_F16X3 / PWV_PREC_F32, out_mode PWV_OUT_GATED, no skip
     * accumulation, no per-sample condition): when head_packed[g] != NULL the gated output stays in registers and
     * feeds pwv_wavenet_head_f32's arithmetic directly; head_out[g] receives [N,T,head_q]; x_out is not written. */
    const float* head_packed[PWV_MAX_NETS];     /* pwv_pack_head_f32 output */
    float* head_out[PWV_MAX_NETS];
    int head_q;
    /* PWV_PREC_F16X3 range guard (see pwv_range_check_f32): with x_first set and range_flag != NULL, layer 0 sets
     * *range_flag = 1 when a flow-input sample is non-finite or |x| > x_limit. */
    float x_limit;
    int* range_flag;
} pwv_layer_args;

int pwv_wavenet_layer_f32(const pwv_layer_args* args, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * WaveNet post-processing head                                         modules.py:145-165
 *   total = o @ skip_w + skip_bias   (in_mode PWV_HEAD_IN_GATED: use_skip_connection False,
 *                                     only the last layer's skip is live, modules.py:147)
 *   total = skip_sum                 (in_mode PWV_HEAD_IN_SKIPSUM)
 *   y = relu(relu(total) @ post1 + b1) @ post2 + b2                    y [N,T,Q], Q <= 4
 * ------------------------------------------------------------------------------------- */
#define PWV_HEAD_IN_GATED 0
#define PWV_HEAD_IN_SKIPSUM 1

size_t pwv_head_packed_floats(int Q);
/* Layer 0 of a scalar-input net (W = 2 causal filter [2,1,64] in front of a [2,64,64] filter / gate pair, modules.py:174-183 and
 * :216-222) folded onto its four input scalars: 2048 floats (8 KB) of split-fp16 A fragments for pwv_persist_args.first_fold.
 * The fold is accumulated in fp64. */
#define PWV_FIRST_FOLD_FLOATS 2048
int pwv_pack_first_fold_f16x3(const float* causal_filter, const float* filter, const float* gate, float* folded, pwv_stream_t stream);
/* the same fold for PWV_PREC_F32 (two fp32 MFMA k-steps per row tile instead of 64): 1024 floats of `folded` are written */
int pwv_pack_first_fold_f32(const float* causal_filter, const float* filter, const float* gate, float* folded, pwv_stream_t stream);

/* skip [1,64,128], skip_bias [128]/NULL, post1 [1,128,128], post1_bias [128]/NULL,
 * post2 [1,128,Q], post2_bias [Q]/NULL */
int pwv_pack_head_f32(const float* skip, const float* skip_bias, const float* post1,
                      const float* post1_bias, const float* post2, const float* post2_bias,
                      int Q, int precision, float* packed, pwv_stream_t stream);

typedef struct pwv_head_args {
    int G;
    const float* in[PWV_MAX_NETS];         /* tile32: gated o (64 channels) or skip sum (128) */
    const float* packed[PWV_MAX_NETS];
    float* out[PWV_MAX_NETS];              /* [N,T,Q] */
    int N, T, Q;
    int in_mode;
    int precision;
    int max_workgroups;
} pwv_head_args;

int pwv_wavenet_head_f32(const pwv_head_args* args, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Whole dilated stack + head of G structurally identical nets in one call: the loop of
 * WaveNet.__call__ (modules.py:138-165) without a host round trip per layer.
 *   buf0[g] holds the causal layer's output on entry; buf0/buf1 ping-pong through the layers;
 *   out[g] receives the net output [N,T,Q].
 * With PWV_PREC_F16X3 or PWV_PREC_F32, no skip accumulation and no per-sample condition the head runs inside the last layer's
 * launch (pwv_layer_args.head_packed) unless `separate_head` is set.
 * streams[0] only (streams[1] == NULL): every layer is one launch covering all G nets.
 * Two streams and G == 2: net g's chain runs on streams[g] with `max_workgroups` workgroups per
 * launch (0 = half of the CUs), launches interleaved -- the two independent chains then hide each
 * other's launch gaps and tails.  The caller orders the streams against its own work.
 * ------------------------------------------------------------------------------------- */
typedef struct pwv_stack_args {
    int G;
    int n_layers;
    const int* dilations;                         /* HOST array [n_layers] */
    float* buf0[PWV_MAX_NETS];                    /* tile32, N*T rows x 64 */
    float* buf1[PWV_MAX_NETS];                    /* tile32, N*T rows x 64 */
    const float* packed_layers[PWV_MAX_NETS];     /* n_layers packed layer buffers, back to back */
    size_t packed_layer_stride;                   /* floats between consecutive layers */
    const float* proj[PWV_MAX_NETS];              /* P rows holding all layers: layer j at +128*j */
    int proj_row_stride;
    const float* cond;                            /* per-sample condition in the form `precision` wants, or NULL */
    int cond_channels;
    float* skip[PWV_MAX_NETS];                    /* tile32, N*T rows x 128, or NULL (use_skip_connection) */
    const float* packed_head[PWV_MAX_NETS];
    float* out[PWV_MAX_NETS];
    int Q;
    int N, T;
    int cond_hop, cond_offset, cond_frames;
    int precision;
    int max_workgroups;
    /* optional timing hooks (hipEvent_t, may be NULL): ev_begin[c] is recorded on chain c's stream in front of its
     * first layer launch, ev_end[c] behind its last RESIDUAL layer launch (layer n_layers-2) -- elapsed / (n_layers-1)
     * is the mean launch duration of the dominant kernel on the production launch path (bench.py's roofline).
     * One chain (c = 0) in single-stream mode, one per net with two streams. */
    void* ev_begin[PWV_MAX_NETS];
    void* ev_end[PWV_MAX_NETS];
    /* optional: layer 0 evaluates the causal layer itself (see pwv_layer_args.x_first); buf0 then only serves as the
     * ping-pong partner of buf1 and need not be initialised */
    const float* x_first;
    const float* causal_filter[PWV_MAX_NETS];
    int separate_head;                            /* 1: never fuse the head into the last layer's launch */
    float x_limit;                                /* forwarded to layer 0 (pwv_layer_args.x_limit / range_flag) */
    int* range_flag;
    const float* first_fold[PWV_MAX_NETS];        /* optional, forwarded to layer 0 (pwv_layer_args.first_fold) */
} pwv_stack_args;

int pwv_wavenet_stack_f32(const pwv_stack_args* args, pwv_stream_t const* streams);

/* ---------------------------------------------------------------------------------------
 * Normalisers (modules.normalize, modules.py:263-284) and the elementwise ops of the un-fused WaveNet path.
 *   pwv_instance_norm_f32   method 'in' (modules.py:274-284): per (utterance, channel) mean / biased variance over the
 *                           TIME axis, y = gamma (x - mean) / sqrt(var + eps) + beta, eps = 1e-8 in the reference;
 *                           x, y [N, T, C] channels-last; gamma / beta [C] or NULL; three launches (fp64 partial sums per
 *                           time chunk, no atomics: bitwise repeatable), workspace from pwv_instance_norm_workspace_bytes.
 *   pwv_channel_affine_f32  y = act(x * scale[c] + bias[c]) -- method 'bn' at inference (modules.py:266: scale =
 *                           gamma / sqrt(moving_variance + 1e-3), bias = beta - moving_mean * scale) where no GEMM follows
 *                           that the host could fold it into; also bias adds and relu.  scale / bias may be NULL;
 *                           tile32 != 0: x, y are tile32 buffers (C % 4 == 0); in place (x == y) allowed.
 *   pwv_add_f32, pwv_gate_f32   out = a + b;  out = tanh(f) * sigmoid(g) (modules.py:236).
 * ------------------------------------------------------------------------------------- */
size_t pwv_instance_norm_workspace_bytes(int N, int T, int C);
int pwv_instance_norm_f32(const float* x, float* y, int N, int T, int C, const float* gamma, const float* beta, float eps,
                          void* workspace, size_t workspace_bytes, pwv_stream_t stream);
int pwv_channel_affine_f32(const float* x, float* y, int64_t rows, int C, const float* scale, const float* bias, int tile32,
                           int relu, pwv_stream_t stream);
int pwv_add_f32(const float* a, const float* b, float* out, int64_t n, pwv_stream_t stream);
int pwv_gate_f32(const float* f, const float* g, float* out, int64_t n, pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * wav -> (normalised) dB mel-spectrogram, the conditioning input of generate.py           data_load.py:51-54
 *   audio.wav2melspec_db (audio.py:341-356): |STFT| (librosa.stft: centred, reflect-padded; `window` is the analysis
 *   window already zero-padded to n_fft) -> mel_basis [n_mels, 1 + n_fft/2] (librosa.filters.mel, audio.py:241) ->
 *   amplitude_to_db (10 log10 max(amin^2, .^2), then clipped to top_db below the utterance maximum) -> if `normalise`:
 *   (clip((db - min_db) / (max_db - min_db), 0, 1) - 0.5) * 2  (audio.py:254-286).
 *   wav [N, L] -> mel [N, 1 + L/hop, n_mels]; fp64 accumulation inside; n_fft even, <= 2048.
 * ------------------------------------------------------------------------------------- */
int pwv_wav_to_mel_db_f32(const float* wav, const float* window, const float* mel_basis, float* mel, int N, int L, int n_fft,
                          int hop, int n_mels, float amin, float top_db, float max_db, float min_db, int normalise,
                          pwv_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * A run of consecutive RESIDUAL layers (out_mode PWV_OUT_RESIDUAL, no skip accumulation, no per-sample condition,
 * PWV_PREC_F16X3 or PWV_PREC_F32) of G nets as ONE persistent launch: the inner iterations of the loop in WaveNet.__call__
 * (modules.py:138-143) without a kernel boundary, a weight-staging phase and a ramp-up / ramp-down per layer.
 *   x_ring[g]  three full-size tile32 buffers (N*T rows x 64) per net, `ring_stride` floats apart (>= pwv_tile32_floats(N*T, 64)).
 *   Layer j of the run reads buffer (j + 2 + r) % 3 and writes buffer (j + r) % 3, r = ring_rotation: the input of the run is
 *   in buffer (2 + r) % 3 on entry, its output in buffer (n_layers - 1 + r) % 3 on return; the third buffer holds
 *   intermediate layers (a stack can be cut into several runs that hand the ring on: r' = (output buffer + 1) % 3).  Layer j reads
 *   packed_layers[g] + j*packed_layer_stride and the P columns proj[g] + 128*j.
 *   Results are bit-identical to n_layers calls of pwv_wavenet_layer_f32.
 * Every workgroup owns the same contiguous rows in every layer and walks layer after layer over them; what it needs of
 * its neighbours' rows (the x[t-d] look-back) is handed over through per-workgroup progress words in `workspace`;
 * csrc/pwv_stack_persist.hip has the protocol.  The call enqueues (unless `workspace_clean`) a kernel that zeroes the control words and one kernel
 * (grid <= one workgroup per CU; max_workgroups > 0 limits it further, e.g. to share the chip with another stream).
 *   pwv_persist_workspace_bytes   size of `workspace` (device memory, 256-byte aligned, contents don't care) for the shape in
 *                                 `args` (G, N, T, n_layers, dilations, max_workgroups, min_units_per_workgroup, tail_q, tail_dilation are read);
 *                                 0 = this shape cannot run as a persistent launch (pwv_last_error says why): use the
 *                                 per-layer launches
 *   pwv_persist_status(&p)        process-wide sticky int32 in pinned host memory: 0, or != 0 once a launch gave up
 *                                 (a poll that ran into its bound: the workgroups were not all resident, e.g. because
 *                                 another process held CUs); the outputs of that launch are then invalid and the caller
 *                                 uses the per-layer path.
 * ------------------------------------------------------------------------------------- */
typedef struct pwv_persist_args {
    size_t struct_size;                           /* = sizeof(pwv_persist_args) as the CALLER was compiled.  The library reads that many bytes and
                                                   * treats every field behind them as zero / NULL, so fields appended in later minor versions
                                                   * (the status word, the tail, the affine) cannot be read out of a shorter caller's memory;
                                                   * 0 or less than the fields up to `min_units_per_workgroup` is PWV_EINVAL.  Zero-initialise
                                                   * the struct (`pwv_persist_args a = {0}; a.struct_size = sizeof a;`) before filling it in. */
    int G;
    int n_layers;                                 /* 2..32 layers in this launch */
    const int* dilations;                         /* HOST array [n_layers] */
    float* x_ring[PWV_MAX_NETS];
    size_t ring_stride;                           /* floats between two of a net's three buffers */
    int ring_rotation;                            /* 0, 1 or 2 */
    const float* packed_layers[PWV_MAX_NETS];     /* the run's first layer */
    size_t packed_layer_stride;
    const float* proj[PWV_MAX_NETS];              /* the run's first layer's columns */
    int proj_row_stride;
    int N, T;
    int cond_hop, cond_offset, cond_frames;
    void* workspace;
    size_t workspace_bytes;
    int workspace_clean;                          /* != 0: `workspace` is all zero on entry (fresh, or last used by this entry point, which
                                                   * zeroes it again before it returns the chip): no zeroing kernel is enqueued */
    int precision;                                /* PWV_PREC_F16X3 or PWV_PREC_F32 (packed_layers packed accordingly) */
    int max_workgroups;                           /* 0 = one per CU */
    int min_units_per_workgroup;                  /* short inputs: use fewer workgroups rather than ranges below this (0 = 4) */
    /* optional: the run starts with the net's layer 0, which evaluates the causal layer itself from the scalar input [N*T]
     * (pwv_layer_args.x_first: same operations, same bits); buffer (2 + r) % 3 is then not read */
    const float* x_first;
    const float* causal_filter[PWV_MAX_NETS];
    float x_limit;                                /* range guard on x_first (pwv_layer_args.x_limit / range_flag) */
    int* range_flag;
    /* optional, with x_first: pwv_pack_first_fold_f16x3's (PWV_PREC_F16X3) or pwv_pack_first_fold_f32's (PWV_PREC_F32) output per net (all nets or none).  Layer 0's
     * filter|gate GEMM then runs on the four scalars x[t-d-1], x[t-d], x[t-1], x[t] themselves (one MFMA k-step instead of
     * eight): the same function (h[t] is linear in x[t-1], x[t]), rounded differently -- within the path's tolerance of the
     * unfolded form, not bit-identical to it */
    const float* first_fold[PWV_MAX_NETS];
    /* optional: this call's own sticky give-up word (pwv_status_words_alloc: words[0]); NULL = the process-wide word of
     * pwv_persist_status.  Two threads / streams with their own words cannot consume or clear each other's flags. */
    int* status;
    /* optional TAIL (tail_q > 0, either arithmetic; the run must be the LAST run of the stack): behind the run's layers every
     * workgroup runs the net's last layer (dilation tail_dilation, packed weights tail_layer[g], P columns proj[g] + 128*n_layers)
     * with the post-processing head behind it (tail_head[g] = pwv_pack_head_f32's output; modules.py:145-165) on its own rows and
     * writes tail_out[g] [N*T, tail_q] -- what pwv_wavenet_layer_f32 with head_packed / head_out computes, bit for bit, without
     * the two kernel boundaries.  With affine_x (the flow's input [N*T]) the IAF affine out = x*s + b (modules.py:59) is evaluated
     * too: s = net 0, b = net 1 (G = 2, tail_q = 1; by whichever of the two nets' workgroups of a row range finishes second) or
     * s, b = the two outputs of one net (G = 1, tail_q = 2); affine_out [N*T].  A flow is then ONE launch. */
    const float* tail_layer[PWV_MAX_NETS];
    const float* tail_head[PWV_MAX_NETS];
    float* tail_out[PWV_MAX_NETS];
    int tail_q;
    int tail_dilation;
    const float* affine_x;
    float* affine_out;
} pwv_persist_args;

size_t pwv_persist_workspace_bytes(const pwv_persist_args* args);
/* 1 if pwv_wavenet_stack_persist_f32 takes its SHORT-INPUT instantiation for `args` (round 6; at most 7 units of 32 rows per workgroup and layer, i.e. up to
 * about 28000 rows for two nets on 256 CUs: progress words per unit instead of per workgroup, a unit stays on one wave through all layers -- its own rows never
 * leave the registers --, a loader wave refills the weights; same results bit for bit), 0 if the general one, -1 on arguments the launch would refuse.
 * Reads what pwv_persist_workspace_bytes reads, and x_first, first_fold, cond_hop, cond_frames, proj_row_stride.  PWV_PERSIST_UNITWORDS=0 in the
 * environment keeps every launch on the general instantiation (A/B). */
int pwv_persist_short_input(const pwv_persist_args* args);
int pwv_persist_status(int** status);
int pwv_wavenet_stack_persist_f32(const pwv_persist_args* args, pwv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PWV_HIP_H_ */
