// VALU issue-rate probe: N dependent-free v_mul/v_add per wave, at 1, 2, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/valu_rate.hip -o tools/micro/bin/valu_rate ; gpurun -- tools/micro/bin/valu_rate
// measured (MI355X): 3.1 cycles per wave64 instruction at 1 wave per SIMD, 2.7 at 2, 2.5 at 4, 2.3 at 8 (at a nominal 2.4 GHz)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNUSED>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x0 = __fadd_rn(__fmul_rn(x0, a), b); x1 = __fadd_rn(__fmul_rn(x1, a), b); x2 = __fadd_rn(__fmul_rn(x2, a), b); x3 = __fadd_rn(__fmul_rn(x3, a), b);
            x4 = __fadd_rn(__fmul_rn(x4, a), b); x5 = __fadd_rn(__fmul_rn(x5, a), b); x6 = __fadd_rn(__fmul_rn(x6, a), b); x7 = __fadd_rn(__fmul_rn(x7, a), b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
    float* out; hipMalloc(&out, 1 << 26);
    const int iters = 4000;
    for (int wpc : {4, 8, 16, 32}) {           // waves per CU: 1, 2, 4, 8 per SIMD
        const int threads = 64 * wpc > 1024 ? 1024 : 64 * wpc;
        const int blocks = 256 * (64 * wpc / threads);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * 8 * 8 * 2;
        const double waves_per_simd = wpc / 4.0;
        // cycles per instruction per SIMD at 2.4 GHz nominal
        printf("waves/SIMD %.0f: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", waves_per_simd, ms,
               ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4);
    }
    return 0;
}
