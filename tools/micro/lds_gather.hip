// LDS gather-rate probe: every lane reads table entries at pseudo-random indices (the list scan's access pattern),
// 16 waves per CU, as 32-bit gathers from a [4096] float table and as 64-bit gathers from a [4096] float2 table.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_gather.hip -o tools/micro/bin/lds_gather ; gpurun -- tools/micro/bin/lds_gather
// measured (MI355X, round 3), uniform independent index bytes: 32-bit 9.98 lane-gathers per clock and CU (6.1 T/s) -- the
// bank arithmetic of 32 lanes on 32 banks, 64 / (2 x 3.15) = 10.15 --, 64-bit 3.6 (2.2 T/s): a random 64-bit gather costs
// 2.8 x a 32-bit one, so a table of pairs {tabA, tabB} serving two queries per gather LOSES (a two-queries-per-workgroup
// scan built on it -- bit-identical rows -- ran 0.733 ms against 0.680 on the bench batch and was dropped).  With index
// bytes taken from a linear congruential state (this probe's first version, kept as the second mode) the figures are
// 12.5 / 4.5: correlated low-order bits spread a lane group over the banks better than chance -- not a rate data reaches.
// The list scan's 8.9 look-ups per clock and CU are 0.89 of the uniform 32-bit figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int WIDE, int HASHED>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f, acc2 = 0.f;
    // HASHED = 1: every index byte comes from a 32-bit finalizer (uniform, independent across lanes);
    // HASHED = 0: bytes of a linear congruential state, as this probe was first written -- its low-order bits are
    // correlated across lanes, which happens to spread a lane group over the banks better than chance
    auto mix = [](uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; };
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t c;
            if (HASHED) { x += 0x9E3779B9u; c = mix(x); } else { x = x * 1664525u + 1013904223u; c = x >> 8; }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t j = HASHED ? (c >> (8 * m)) & 255u : ((c >> (8 * (m % 3))) + 37u * m) & 255u;
                if (WIDE) { const f32x2 v = reinterpret_cast<const f32x2*>(lds)[(u * 4 + m) % 16 * 256 + j]; acc += v.x; acc2 += v.y; }
                else acc += lds[(u * 4 + m) % 16 * 256 + j];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + acc2;
}
int main() {
    float* out; hipMalloc(&out, 1 << 26);
    const int iters = 2000;
    for (int hashed = 0; hashed < 2; hashed++)
    for (int wide = 0; wide < 2; wide++) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int blocks = 256 * 4 * 8;        // 8 rounds of 4 workgroups per CU
        const size_t smem = 32768;
        auto launch = [&](int it) {
            if (wide && hashed) hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(256), smem, 0, out, it);
            else if (wide) hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256), smem, 0, out, it);
            else if (hashed) hipLaunchKernelGGL((k<0, 1>), dim3(blocks), dim3(256), smem, 0, out, it);
            else hipLaunchKernelGGL((k<0, 0>), dim3(blocks), dim3(256), smem, 0, out, it);
        };
        launch(10);
        (void)hipEventRecord(e0);
        launch(iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double gathers = (double)blocks * 256 * iters * 16;
        printf("%s gathers, %s indices: %.3f ms, %.2f lane-gathers per clock and CU (2.4 GHz nominal), %.2f T gathers/s\n", wide ? "64-bit" : "32-bit",
               hashed ? "hashed (uniform)" : "LCG-byte", ms, gathers / (ms * 1e-3) / 256 / 2.4e9, gathers / (ms * 1e-3) / 1e12);
    }
    return 0;
}
