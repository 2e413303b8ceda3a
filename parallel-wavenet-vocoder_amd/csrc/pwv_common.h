// Shared helpers for libpwv_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "pwv_hip.h"

namespace pwv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// thread-local error message (pwv_last_error)
char* error_buffer();
int set_error(int code, const char* fmt, ...);

#define PWV_CHECK_ARG(cond, ...)                           \
    do {                                                   \
        if (!(cond)) return pwv::set_error(PWV_EINVAL, __VA_ARGS__); \
    } while (0)

#define PWV_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return pwv::set_error(PWV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Channel owned by accumulator register r (0..15) of MFMA row-tile `it` for the lane half h:
// the v_mfma_f32_32x32x* C/D layout is  row = (r&3) + 8*(r>>2) + 4*h, col = lane&31.
// The whole pipeline keeps activations in exactly this order (lane half h owns the 16-byte
// chunks at float offsets 8*g + 4*h of a 64-channel row), so that the D registers of one GEMM
// are the B operand of the next with no data movement.
__host__ __device__ inline int chan_of(int it, int r, int h) { return 32 * it + 8 * (r >> 2) + 4 * h + (r & 3); }

inline int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

}  // namespace pwv
