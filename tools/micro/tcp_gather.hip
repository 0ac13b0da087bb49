// Second gather engine probe: can the vector L1 (TCP) serve some of the list scan's table look-ups beside the LDS pipe?
// NG of the 16 look-ups of a code go to a per-workgroup table in global memory (NG KB, L1-resident), 16-NG to LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tcp_gather.hip -o tools/micro/bin/tcp_gather ; gpurun -- tools/micro/bin/tcp_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int NG>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ gtab, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const float* gt = gtab + (size_t)(blockIdx.x & 1023) * (NG ? NG : 1) * 256;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f;
    auto mix = [](uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; };
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            x += 0x9E3779B9u;
            const uint32_t c = mix(x);            // four uniform, independent index bytes (see lds_gather.hip on why not LCG bytes)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t j = (c >> (8 * m)) & 255u;
                const int s = u * 4 + m;
                if (s < NG) acc += gt[s * 256 + j];
                else acc += lds[s * 256 + j];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int NG> void run(float* out, const float* gtab) {
    const int iters = 2000, blocks = 256 * 4 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NG>, dim3(blocks), dim3(256), 16384, 0, out, gtab, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NG>, dim3(blocks), dim3(256), 16384, 0, out, gtab, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gathers = (double)blocks * 256 * iters * 16;
    printf("%2d of 16 look-ups through the L1: %.3f ms, %.2f lane-gathers per clock and CU (2.4 GHz nominal)\n", NG, ms,
           gathers / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    float *out, *gtab; hipMalloc(&out, 1 << 26); hipMalloc(&gtab, 1024 * 16 * 1024);
    hipMemset(gtab, 0, 1024 * 16 * 1024);
    run<0>(out, gtab); run<1>(out, gtab); run<2>(out, gtab); run<3>(out, gtab); run<4>(out, gtab); run<6>(out, gtab); run<8>(out, gtab); run<16>(out, gtab);
    return 0;
}
