/* Plain C client of libpwv_hip.so: no Python, no torch -- raw hipMalloc'd pointers through the C ABI.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/c_abi_smoke.c -Iinclude -I/opt/rocm/include \
 *       -Lparallel-wavenet-vocoder_amd -lpwv_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/parallel-wavenet-vocoder_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_smoke && /tmp/c_abi_smoke
 * (tests/test_abi.py compiles it on every run, tests/test_gpu_parity.py runs it on the GPU)
 *
 * Runs modules.causal_conv (modules.py:11-43) and the tile32 round trip on the GPU and checks them against loops
 * written here.  Exit code 0 = match. */
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
    return (err <= 1e-5 && bad == 0 && rc == PWV_EINVAL) ? 0 : 1;
}
