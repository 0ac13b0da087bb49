// fp32 reductions in the exact operation order of the reference's SSE kernels, so
// that device results are bit-identical to the CPU library:
//   four lane accumulators s[l] += x[4i+l]*y[4i+l] (multiply and add NOT fused),
//   zero-padded tail, then (s0+s1)+(s2+s3)      -- utils.cpp:481-556.
// __fmul_rn/__fadd_rn/__fsub_rn are never contracted into FMAs by hipcc.
#pragma once
#include <hip/hip_runtime.h>

namespace vlq {

// fvec_inner_product, utils.cpp:509-533 (tail product added unconditionally)
template <typename LoadX, typename LoadY>
__device__ __forceinline__ float ip_sse_order(LoadX x, LoadY y, int d) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 4 <= d; i += 4) {
        s0 = __fadd_rn(s0, __fmul_rn(x(i + 0), y(i + 0)));
        s1 = __fadd_rn(s1, __fmul_rn(x(i + 1), y(i + 1)));
        s2 = __fadd_rn(s2, __fmul_rn(x(i + 2), y(i + 2)));
        s3 = __fadd_rn(s3, __fmul_rn(x(i + 3), y(i + 3)));
    }
    const int r = d - i;
    s0 = __fadd_rn(s0, r > 0 ? __fmul_rn(x(i + 0), y(i + 0)) : 0.f);
    s1 = __fadd_rn(s1, r > 1 ? __fmul_rn(x(i + 1), y(i + 1)) : 0.f);
    s2 = __fadd_rn(s2, r > 2 ? __fmul_rn(x(i + 2), y(i + 2)) : 0.f);
    s3 = __fadd_rn(s3, 0.f);
    return __fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3));
}

// fvec_norm_L2sqr, utils.cpp:538-556
template <typename LoadX>
__device__ __forceinline__ float norm_sse_order(LoadX x, int d) {
    return ip_sse_order(x, x, d);
}

// fvec_L2sqr, utils.cpp:481-506 (tail only when d % 4 != 0)
template <typename LoadX, typename LoadY>
__device__ __forceinline__ float l2sqr_sse_order(LoadX x, LoadY y, int d) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 4 <= d; i += 4) {
        float a0 = __fsub_rn(x(i + 0), y(i + 0)), a1 = __fsub_rn(x(i + 1), y(i + 1));
        float a2 = __fsub_rn(x(i + 2), y(i + 2)), a3 = __fsub_rn(x(i + 3), y(i + 3));
        s0 = __fadd_rn(s0, __fmul_rn(a0, a0));
        s1 = __fadd_rn(s1, __fmul_rn(a1, a1));
        s2 = __fadd_rn(s2, __fmul_rn(a2, a2));
        s3 = __fadd_rn(s3, __fmul_rn(a3, a3));
    }
    const int r = d - i;
    if (r > 0) {
        float a0 = __fsub_rn(x(i + 0), y(i + 0));
        s0 = __fadd_rn(s0, __fmul_rn(a0, a0));
        if (r > 1) { float a1 = __fsub_rn(x(i + 1), y(i + 1)); s1 = __fadd_rn(s1, __fmul_rn(a1, a1)); }
        else s1 = __fadd_rn(s1, 0.f);
        if (r > 2) { float a2 = __fsub_rn(x(i + 2), y(i + 2)); s2 = __fadd_rn(s2, __fmul_rn(a2, a2)); }
        else s2 = __fadd_rn(s2, 0.f);
        s3 = __fadd_rn(s3, 0.f);
    }
    return __fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3));
}

}  // namespace vlq
