// What the per-probe table stores cost the list scan, and whether another store instruction is cheaper: the headline
// shape in miniature -- per "probe" a 256-thread workgroup writes a 16 KB table into one of two LDS buffers, passes a
// barrier and looks up 768 codes (3 chunks per wave, adc16_fixed) streamed from memory; 4 workgroups per CU.
//   ST 0: 4 x ds_write_b128 per thread (the kernel's build_lut16)      ST 1: 16 x ds_write_addtid_b32 (M0 + offset + 4 * lane)
//   ST 2: 8 x ds_write_b64                                             ST 3: no stores (bound)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vector_line_quantization_amd/csrc tools/micro/lut_store.hip -o /tmp/lut_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>
#include "scan16_common.cuh"
using namespace vlq;

template <int ST, int BUF>
__device__ __forceinline__ void store_table(float* lds, int t, int lane, int wave, const float4 (&v)[4]) {
    if (ST == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) reinterpret_cast<float4*>(lds + BUF * 4096)[i * 256 + t] = v[i];
    } else if (ST == 2) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            reinterpret_cast<float2*>(lds + BUF * 4096)[(i * 256 + t) * 2] = make_float2(v[i].x, v[i].y);
            reinterpret_cast<float2*>(lds + BUF * 4096)[(i * 256 + t) * 2 + 1] = make_float2(v[i].z, v[i].w);
        }
    } else if (ST == 1) {
        // lane l owns entries l, l + 64, l + 128, l + 192 of its wave's 256-entry rows: four 256-byte pieces per row
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(BUF * 16384 + wave * 1024);
        uint32_t save;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %17\n\t"
            "ds_write_addtid_b32 %1 offset:0\n\t"
            "ds_write_addtid_b32 %2 offset:256\n\t"
            "ds_write_addtid_b32 %3 offset:512\n\t"
            "ds_write_addtid_b32 %4 offset:768\n\t"
            "ds_write_addtid_b32 %5 offset:4096\n\t"
            "ds_write_addtid_b32 %6 offset:4352\n\t"
            "ds_write_addtid_b32 %7 offset:4608\n\t"
            "ds_write_addtid_b32 %8 offset:4864\n\t"
            "ds_write_addtid_b32 %9 offset:8192\n\t"
            "ds_write_addtid_b32 %10 offset:8448\n\t"
            "ds_write_addtid_b32 %11 offset:8704\n\t"
            "ds_write_addtid_b32 %12 offset:8960\n\t"
            "ds_write_addtid_b32 %13 offset:12288\n\t"
            "ds_write_addtid_b32 %14 offset:12544\n\t"
            "ds_write_addtid_b32 %15 offset:12800\n\t"
            "ds_write_addtid_b32 %16 offset:13056\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(save)
            : "v"(v[0].x), "v"(v[0].y), "v"(v[0].z), "v"(v[0].w), "v"(v[1].x), "v"(v[1].y), "v"(v[1].z), "v"(v[1].w),
              "v"(v[2].x), "v"(v[2].y), "v"(v[2].z), "v"(v[2].w), "v"(v[3].x), "v"(v[3].y), "v"(v[3].z), "v"(v[3].w), "s"(base)
            : "memory");
    }
}

template <int ST>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ codes, float* out, int nprobe, uint32_t len) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lds = reinterpret_cast<float*>(smraw);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 8192; i += 256) lds[i] = 1.f;
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    __syncthreads();
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = make_float4(t * 0.001f + i, 1.f, 2.f, 3.f);
    float best = 3.4e38f;
    const uint4* cp = codes + (size_t)(blockIdx.x & 2047) * len;      // (the buffer holds 2048 lists)
    auto scan = [&](auto bufc, int p) {
        constexpr int B = decltype(bufc)::value;
        const uint32_t off = (uint32_t)(p * 768) % (len - 768);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint4 cc = cp[off + c * 256 + t];
            best = fminf(best, adc16_fixed<B>(cc, 0.f, two));
        }
    };
    for (int p = 0; p < nprobe; p += 2) {
        __builtin_amdgcn_s_setprio(2);
        store_table<ST, 0>(lds, t, lane, wave, v);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        scan(std::integral_constant<int, 0>{}, p);
        v[0].x += best * 1e-30f;
        __builtin_amdgcn_s_setprio(2);
        store_table<ST, 1>(lds, t, lane, wave, v);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        scan(std::integral_constant<int, 1>{}, p + 1);
        v[1].x += best * 1e-30f;
    }
    out[blockIdx.x * 256 + t] = best;
}

template <int ST> void run(const uint4* codes, float* out, const char* what) {
    const int nprobe = 32, blocks = 256 * 4 * 8; const uint32_t len = 16384;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t smem = 32768;
    hipLaunchKernelGGL((k<ST>), dim3(blocks), dim3(256), smem, 0, codes, out, 2, len);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<ST>), dim3(blocks), dim3(256), smem, 0, codes, out, nprobe, len);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double probes_per_cu = (double)blocks * nprobe / 256;
    printf("%-44s %.3f ms, %.0f cycles per probe and CU (2.4 GHz nominal), %.2f look-ups per clock and CU\n", what, ms,
           ms * 1e-3 * 2.4e9 / probes_per_cu, (double)blocks * nprobe * 768 * 16 / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    const size_t n = (size_t)2048 * 16384;
    uint4* codes; float* out; (void)hipMalloc(&codes, n * 16); (void)hipMalloc(&out, 1 << 24);
    uint32_t* h = (uint32_t*)malloc(n * 16); uint32_t x = 1;
    for (size_t i = 0; i < n * 4; i++) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 15); }
    (void)hipMemcpy(codes, h, n * 16, hipMemcpyHostToDevice);
    run<3>(codes, out, "no table stores");
    run<0>(codes, out, "4 x ds_write_b128 per thread (the kernel)");
    run<2>(codes, out, "8 x ds_write_b64 per thread");
    run<1>(codes, out, "16 x ds_write_addtid_b32 per thread");
    return 0;
}
