// Probe: do v_mfma_f32_32x32x16_f16 inputs keep fp16 subnormals?  does v_cvt_pkrtz produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

__global__ void k(float* out, float tiny, float big) {
    // A[i][k]: lane (i = l&31, h = l>>5) holds k = 8h..8h+7.  Put `tiny` at A[i][0], `big` at B[0][j].
    const int lane = threadIdx.x;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    fp16x2 t = __builtin_amdgcn_cvt_pkrtz(tiny, big);
    if (lane < 32) { a[0] = (_Float16)t[0]; b[0] = (_Float16)t[1]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) { out[0] = c[0]; out[1] = (float)(_Float16)t[0]; out[2] = (float)(_Float16)tiny; }
}

int main() {
    float* d; hipMalloc(&d, 64);
    const float tinies[] = {1e-3f, 6.1e-5f, 3e-5f, 1e-6f, 6e-8f};
    for (float tiny : tinies) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny, 1024.f);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("tiny=%g  mfma(tiny*1024)=%g (exact %g)  cvt_pkrtz->f32=%g  cvt_rne->f32=%g\n", tiny, h[0], tiny * 1024, h[1], h[2]);
    }
    return 0;
}
