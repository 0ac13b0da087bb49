// The list scan's inner loop in isolation: the half-block pipeline of scan16.hip (same macros, same selection) over a
// fixed table, with the 16-byte codes from registers (SRC 0) or streamed from global memory (SRC 1), with a running
// minimum (SEL 0) or the real 64-key wave selection (SEL 1).  Prints look-ups per clock and CU beside
// tools/micro/lds_gather.hip's 9.98 for bare uniformly random gathers: what the loop's own structure costs.
// measured (MI355X, round 3): codes from memory + selection (= the kernel's loop) 8.9-9.0, with a running minimum 9.3, the same
// from HBM, Infinity Cache or L2 and one or two trips ahead; (codes hashed in registers cost VALU time: not a clean upper bound)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vector_line_quantization_amd/csrc tools/micro/scan_loop.hip -o /tmp/scan_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>
#include "scan16_common.cuh"
using namespace vlq;

template <int SRC, int SEL, int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ codes, float* out, uint32_t len, int nl, uint32_t wg_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);
    u64* queue = reinterpret_cast<u64*>(smraw + 32768);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int NT = 256;
    for (int i = t; i < 8192; i += 256) lut[i] = (float)((i * 2654435761u) >> 20) * 1e-3f + 1.f;
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    __syncthreads();
    WaveSelect<1, 1, false> sel;
    sel.init(10, queue + wave * 64, lane);
    float best = 3.4e38f;
    uint32_t dummy = 0;
    const uint4* cp = codes + (size_t)blockIdx.x * wg_stride;
    uint32_t x = t * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto mix = [](uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; };
    auto gen = [&]() { uint4 c; x += 0x9E3779B9u; c.x = mix(x); c.y = mix(x ^ 0x68E31DA4u); c.z = mix(x ^ 0xB5297A4Du); c.w = mix(x ^ 0x1B56C4E9u); return c; };
    for (int l = 0; l < nl; l++) {
        const float dis0 = (float)l;
        const uint32_t pos0 = l * len;
        uint4 cc, cd, ce, cf;
        if (SRC) { cc = cp[min((uint32_t)t, len - 1)]; cd = cp[min((uint32_t)t + NT, len - 1)]; } else { cc = gen(); cd = gen(); }
        if (SRC == 2) { ce = cp[min((uint32_t)t + 2 * NT, len - 1)]; cf = cp[min((uint32_t)t + 3 * NT, len - 1)]; }
        uint32_t j0 = (uint32_t)wave * 64;
        for (; j0 + NT < len; j0 += 2 * NT) {
            const uint32_t ja = j0 + lane, jb = ja + NT;
            const uint4 ca = cc, cb = cd;
            if (SRC >= 3) {       // codes from registers, but the loads are issued all the same (16 or 4 bytes per lane) and only folded into a dummy
                cc = gen(); cd = gen();
                if (SRC == 3) { const uint4 u = cp[min(jb + NT, len - 1)], w = cp[min(jb + 2 * NT, len - 1)]; dummy ^= u.x ^ u.y ^ u.z ^ u.w ^ w.x ^ w.y ^ w.z ^ w.w; }
                else { const uint32_t u = reinterpret_cast<const uint32_t*>(cp)[min(jb + NT, len - 1)], w = reinterpret_cast<const uint32_t*>(cp)[min(jb + 2 * NT, len - 1)]; dummy ^= u ^ w; }
            } else if (SRC == 2) { cc = ce; cd = cf; ce = cp[min(jb + 3 * NT, len - 1)]; cf = cp[min(jb + 4 * NT, len - 1)]; }
            else if (SRC) { cc = cp[min(jb + NT, len - 1)]; cd = cp[min(jb + 2 * NT, len - 1)]; } else { cc = gen(); cd = gen(); }
            float da, db;
            if (MODE == 0) {          // the kernel's half-block pipeline
                float h1[8], h2[8], h3[8], h4[8];
                { { float (&v)[8] = h1; VLQ_G8LO_NW(0, ca.x, ca.y); } { float (&v)[8] = h2; VLQ_G8HI_NW(0, ca.z, ca.w); } }
                VLQ_WAIT8(8, h1);
                da = dis0;
#pragma unroll
                for (int m = 0; m < 8; m++) da = __fadd_rn(da, h1[m]);
                asm volatile("" : "+v"(da));
                { float (&v)[8] = h3; VLQ_G8LO_NW(0, cb.x, cb.y); }
                VLQ_WAIT8(8, h2);
#pragma unroll
                for (int m = 0; m < 8; m++) da = __fadd_rn(da, h2[m]);
                asm volatile("" : "+v"(da));
                { float (&v)[8] = h4; VLQ_G8HI_NW(0, cb.z, cb.w); }
                VLQ_WAIT8(8, h3);
                db = dis0;
#pragma unroll
                for (int m = 0; m < 8; m++) db = __fadd_rn(db, h3[m]);
                asm volatile("" : "+v"(db));
                VLQ_WAIT8(0, h4);
#pragma unroll
                for (int m = 0; m < 8; m++) db = __fadd_rn(db, h4[m]);
            } else if (MODE == 1) {   // all 16 of a code at once, one wait (adc16_fixed)
                da = adc16_fixed<0>(ca, dis0, two);
                db = adc16_fixed<0>(cb, dis0, two);
            } else {                  // compiler-scheduled plain C++
                da = adc16(lut, ca, dis0);
                db = adc16(lut, cb, dis0);
            }
            if (SEL) {
                const bool hit_a = __builtin_amdgcn_ballot_w64(da < sel.thr) != 0;
                if (hit_a) sel.offer(da, pos0 + ja, true);
                sel.offer(db, pos0 + jb, jb < len);
            } else best = fminf(best, fminf(da, db));
        }
    }
    if (SEL) { sel.flush(); out[blockIdx.x * 256 + t] = (float)(sel.best[0] >> 32); }
    else out[blockIdx.x * 256 + t] = best;
    if (dummy == 0x12345u) out[0] = 1.f;
}

template <int SRC, int SEL, int MODE> void run(const uint4* codes, float* out, const char* what, uint32_t wg_stride = 16384) {
    const uint32_t len = 16384; const int nl = 8, blocks = 256 * 4 * 2;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t smem = 32768 + 4 * 64 * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<SRC, SEL, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((k<SRC, SEL, MODE>), dim3(blocks), dim3(256), smem, 0, codes, out, len, 1, wg_stride);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SRC, SEL, MODE>), dim3(blocks), dim3(256), smem, 0, codes, out, len, nl, wg_stride);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double lookups = (double)blocks * nl * len * 16;
    printf("%-64s %.3f ms, %.2f look-ups per clock and CU, %.2f TB/s of code bytes\n", what, ms, lookups / (ms * 1e-3) / 256 / 2.4e9,
           lookups / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t n = (size_t)2048 * 16384;
    uint4* codes; float* out; (void)hipMalloc(&codes, n * 16); (void)hipMalloc(&out, 1 << 24);
    uint32_t* h = (uint32_t*)malloc(n * 16); uint32_t x = 1;
    for (size_t i = 0; i < n * 4; i++) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 15); }
    (void)hipMemcpy(codes, h, n * 16, hipMemcpyHostToDevice);
    run<0, 0, 0>(codes, out, "codes from registers, running minimum, half-block pipeline");
    run<0, 0, 1>(codes, out, "codes from registers, running minimum, 16 reads + one wait");
    run<0, 0, 2>(codes, out, "codes from registers, running minimum, compiler-scheduled");
    run<0, 1, 0>(codes, out, "codes from registers, wave selection, half-block pipeline");
    run<1, 0, 0>(codes, out, "codes from memory, running minimum, half-block pipeline");
    run<1, 1, 0>(codes, out, "codes from memory, wave selection, half-block pipeline (= kernel)");
    run<1, 1, 1>(codes, out, "codes from memory, wave selection, 16 reads + one wait");
    run<1, 1, 0>(codes, out, "codes from L2 (one 256 KB list for all), wave selection, half-block pipeline", 0);
    run<1, 1, 0>(codes, out, "codes from L2 / MALL (64 lists of 256 KB), wave selection, half-block pipeline", 256);
    run<3, 1, 0>(codes, out, "codes from registers + the two 16-byte loads issued and discarded, wave selection");
    run<4, 1, 0>(codes, out, "codes from registers + two 4-byte loads issued and discarded, wave selection");
    run<2, 1, 0>(codes, out, "codes from memory two trips ahead, wave selection, half-block pipeline");
    run<2, 0, 0>(codes, out, "codes from memory two trips ahead, running minimum, half-block pipeline");
    return 0;
}
