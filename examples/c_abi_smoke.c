/* Plain C client of libpwv_hip.so: no Python, no torch -- raw hipMalloc'd pointers through the C ABI.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/c_abi_smoke.c -Iinclude -I/opt/rocm/include \
 *       -Lparallel-wavenet-vocoder_amd -lpwv_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/parallel-wavenet-vocoder_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_smoke && /tmp/c_abi_smoke
 * (tests/test_abi.py compiles it on every run, tests/test_gpu_parity.py runs it on the GPU)
 *
 * Runs modules.causal_conv (modules.py:11-43) and the tile32 round trip on the GPU and checks them against loops
 * written here; then the round-5 entry points a C caller needs for a serving loop: its OWN pair of sticky words
 * (pwv_status_words_alloc) with the range guard reporting into it, and the capturable logistic sampler
 * (pwv_logistic_noise_stream_f32) against the by-value one (models.py:32-33).  Exit code 0 = match. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pwv_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_PWV(x) do { int r_ = (x); if (r_ != PWV_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, pwv_last_error()); return 3; } } while (0)

int main(void) {
    enum { N = 2, T = 77, CIN = 3, COUT = 8, W = 2, D = 5, C = 64 };
    static float x[N * T * CIN], f[W * CIN * COUT], y[N * T * COUT], want[N * T * COUT];
    static float rows[N * T * C], back[N * T * C];
    unsigned s = 12345u;
    int i, n, t, k, ci, co;
    for (i = 0; i < N * T * CIN; ++i) { s = s * 1664525u + 1013904223u; x[i] = (float)(s >> 8) / 16777216.0f - 0.5f; }
    for (i = 0; i < W * CIN * COUT; ++i) { s = s * 1664525u + 1013904223u; f[i] = (float)(s >> 8) / 16777216.0f - 0.5f; }
    for (i = 0; i < N * T * C; ++i) { s = s * 1664525u + 1013904223u; rows[i] = (float)(s >> 8) / 16777216.0f; }
    /* y[n,t,:] = sum_k x[n, t-(W-1-k)*d, :] @ f[k],  x[t<0] = 0 */
    for (n = 0; n < N; ++n)
        for (t = 0; t < T; ++t)
            for (co = 0; co < COUT; ++co) {
                double acc = 0;
                for (k = 0; k < W; ++k) {
                    const int ts = t - (W - 1 - k) * D;
                    if (ts < 0) continue;
                    for (ci = 0; ci < CIN; ++ci) acc += (double)x[(n * T + ts) * CIN + ci] * f[(k * CIN + ci) * COUT + co];
                }
                want[(n * T + t) * COUT + co] = (float)acc;
            }
    float *dx, *df, *dy, *drows, *dtile, *dback;
    const size_t tile_floats = pwv_tile32_floats((int64_t)N * T, C);
    CHECK_HIP(hipMalloc((void**)&dx, sizeof x));
    CHECK_HIP(hipMalloc((void**)&df, sizeof f));
    CHECK_HIP(hipMalloc((void**)&dy, sizeof y));
    CHECK_HIP(hipMalloc((void**)&drows, sizeof rows));
    CHECK_HIP(hipMalloc((void**)&dtile, tile_floats * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dback, sizeof rows));
    CHECK_HIP(hipMemcpy(dx, x, sizeof x, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(df, f, sizeof f, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(drows, rows, sizeof rows, hipMemcpyHostToDevice));
    printf("libpwv_hip version %d on a device with %d CUs\n", pwv_version(), pwv_device_cus());
    CHECK_PWV(pwv_causal_conv_f32(dx, df, dy, N, T, CIN, COUT, W, D, NULL));
    CHECK_PWV(pwv_rows_to_tile32_f32(drows, dtile, (int64_t)N * T, C, NULL));
    CHECK_PWV(pwv_tile32_to_rows_f32(dtile, dback, (int64_t)N * T, C, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(y, dy, sizeof y, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(back, dback, sizeof back, hipMemcpyDeviceToHost));
    double err = 0;
    for (i = 0; i < N * T * COUT; ++i) { const double e = fabs((double)y[i] - want[i]); if (e > err) err = e; }
    int bad = 0;
    for (i = 0; i < N * T * C; ++i) bad += back[i] != rows[i];
    /* error behaviour: bad arguments come back as a negative code plus a message, nothing is launched */
    const int rc = pwv_causal_conv_f32(NULL, df, dy, N, T, CIN, COUT, W, D, NULL);
    printf("causal_conv max |err| = %.3g, tile32 round trip mismatches = %d, NULL input -> %d (\"%s\")\n", err, bad, rc, pwv_last_error());

    /* this caller's own sticky words: words[0] would go into pwv_persist_args.status, words + 1 is its range flag */
    int* words = NULL;
    CHECK_PWV(pwv_status_words_alloc(&words));
    CHECK_PWV(pwv_range_check_f32(drows, (int64_t)N * T * C, 2.0f, words + 1, NULL));      /* rows are in [0, 1): in range */
    CHECK_HIP(hipDeviceSynchronize());
    const int flag_clean = words[1];
    CHECK_PWV(pwv_range_check_f32(drows, (int64_t)N * T * C, 0.5f, words + 1, NULL));      /* ... and not below 0.5 */
    CHECK_HIP(hipDeviceSynchronize());
    const int flag_raised = words[1], give_up_word = words[0];
    words[1] = 0;

    /* the sampler in its capturable form: state = {seed, offset, ticket, skip} in device memory; launch k draws the counters
     * [offset + k n, offset + (k + 1) n) of the stream pwv_logistic_noise_f32(seed, .) draws by value */
    enum { NZ = 1000 };
    static float za[NZ], zb[NZ];
    float *dza, *dzb;
    uint64_t state[4] = {77u, 12345u, 0u, 0u}, *dstate;
    int noise_bad = 0, kk;
    CHECK_HIP(hipMalloc((void**)&dza, sizeof za));
    CHECK_HIP(hipMalloc((void**)&dzb, sizeof zb));
    CHECK_HIP(hipMalloc((void**)&dstate, sizeof state));
    CHECK_HIP(hipMemcpy(dstate, state, sizeof state, hipMemcpyHostToDevice));
    for (kk = 0; kk < 3; ++kk) {
        CHECK_PWV(pwv_logistic_noise_stream_f32(dza, NZ, dstate, NULL));
        CHECK_PWV(pwv_logistic_noise_f32(dzb, NZ, 77u, 12345u + (uint64_t)kk * NZ, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(za, dza, sizeof za, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(zb, dzb, sizeof zb, hipMemcpyDeviceToHost));
        for (i = 0; i < NZ; ++i) noise_bad += za[i] != zb[i];
    }
    CHECK_HIP(hipMemcpy(state, dstate, sizeof state, hipMemcpyDeviceToHost));
    const int state_ok = state[0] == 77u && state[1] == 12345u + 3u * NZ && state[2] == 0u && state[3] == 0u;
    printf("own status words: range flag %d -> %d (give-up word %d); capturable sampler: %d mismatches over 3 launches, state %s\n",
           flag_clean, flag_raised, give_up_word, noise_bad, state_ok ? "advanced by 3 n" : "WRONG");
    CHECK_PWV(pwv_status_words_free(words));
    return (err <= 1e-5 && bad == 0 && rc == PWV_EINVAL && flag_clean == 0 && flag_raised == 1 && give_up_word == 0 && noise_bad == 0 && state_ok) ? 0 : 1;
}
