// Device-side list append (lists.h).  The only library primitive used is rocPRIM's radix
// sort, for the stable (list, input position) ordering of a batch.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "lists.h"

namespace vlq {

namespace {

template <typename T>
__global__ void count_and_key_kernel(const T* __restrict__ assign, int64_t n, int64_t nlist,
                                     int* __restrict__ cnt, unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t key = (int64_t)assign[i];
    const bool ok = key >= 0 && key < nlist;
    if (ok) atomicAdd(&cnt[key], 1);
    // sort key = (list, input position): any correct sort is then a stable one
    keys[i] = ((unsigned long long)(ok ? (uint32_t)key : 0xFFFFFFFFu) << 32) | (uint32_t)i;
}

// old list i moves from old_off[i] to new_off[i]; one workgroup walks lists b, b+grid, ...
__global__ __launch_bounds__(256) void relayout_kernel(const uint8_t* __restrict__ oc, const uint8_t* __restrict__ ol,
                                                       const int64_t* __restrict__ oi, const int64_t* __restrict__ old_off,
                                                       const int64_t* __restrict__ len, const int64_t* __restrict__ new_off,
                                                       int64_t nlist, int code_size, uint8_t* __restrict__ nc,
                                                       uint8_t* __restrict__ nl, int64_t* __restrict__ ni) {
    for (int64_t l = blockIdx.x; l < nlist; l += gridDim.x) {
        const int64_t n = len[l];
        if (n == 0) continue;
        const int64_t so = old_off[l], dn = new_off[l];
        const int64_t bytes = n * code_size;
        const uint8_t* src = oc + so * code_size;
        uint8_t* dst = nc + dn * code_size;
        if (((so * code_size) | (dn * code_size)) % 16 == 0) {
            const int64_t n16 = bytes >> 4;
            for (int64_t j = threadIdx.x; j < n16; j += 256)
                reinterpret_cast<uint4*>(dst)[j] = reinterpret_cast<const uint4*>(src)[j];
            for (int64_t j = (n16 << 4) + threadIdx.x; j < bytes; j += 256) dst[j] = src[j];
        } else {
            for (int64_t j = threadIdx.x; j < bytes; j += 256) dst[j] = src[j];
        }
        for (int64_t j = threadIdx.x; j < n; j += 256) ni[dn + j] = oi[so + j];
        if (ol) for (int64_t j = threadIdx.x; j < n; j += 256) nl[dn + j] = ol[so + j];
    }
}

// sorted position s holds (list, i): its rank inside the batch's share of the list is
// s - cstart[list]
__global__ void place_kernel(const unsigned long long* __restrict__ sorted, int64_t n,
                             const int64_t* __restrict__ cstart, const int64_t* __restrict__ off,
                             const int64_t* __restrict__ len, const uint8_t* __restrict__ new_codes,
                             const uint8_t* __restrict__ new_lambdas, const int64_t* __restrict__ xids,
                             int64_t id_base, int code_size, uint8_t* __restrict__ codes,
                             uint8_t* __restrict__ lambdas, int64_t* __restrict__ ids) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const unsigned long long k = sorted[s];
    const uint32_t list = (uint32_t)(k >> 32);
    if (list == 0xFFFFFFFFu) return;
    const int64_t i = (int64_t)(uint32_t)k;
    const int64_t dst = off[list] + len[list] + (s - cstart[list]);
    const uint8_t* src = new_codes + i * code_size;
    uint8_t* d = codes + dst * code_size;
    if (code_size == 16) {
        *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(src);
    } else {
        for (int b = 0; b < code_size; b++) d[b] = src[b];
    }
    if (lambdas) lambdas[dst] = new_lambdas[i];
    ids[dst] = xids ? xids[i] : id_base + i;
}

}  // namespace

// move every list to new_off[i] (capacities new_off[i+1] - new_off[i] >= len[i]); synchronises
int lists_relayout(ListStore& ls, std::vector<int64_t>& new_off, hipStream_t s) {
    const int64_t nlist = ls.nlist;
    const int64_t cap = new_off[(size_t)nlist];
    DevBuf nc, nl, ni, noff;
    int rc = nc.reserve((size_t)cap * ls.code_size + 16);
    if (rc == VLQ_OK) rc = ni.reserve((size_t)cap * 8 + 16);
    if (rc == VLQ_OK && ls.lambdas) rc = nl.reserve((size_t)cap + 16);
    if (rc == VLQ_OK) rc = noff.reserve(((size_t)nlist + 1) * 8);
    if (rc != VLQ_OK) { nc.release(); nl.release(); ni.release(); noff.release(); return rc; }
    hipError_t e = hipMemcpyAsync(noff.p, new_off.data(), ((size_t)nlist + 1) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && ls.codes->p) {
        const unsigned g = (unsigned)std::min<int64_t>(nlist, 65535 * 16);
        hipLaunchKernelGGL(relayout_kernel, dim3(g), dim3(256), 0, s, ls.codes->as<uint8_t>(),
                           ls.lambdas ? ls.lambdas->as<uint8_t>() : nullptr, ls.ids->as<int64_t>(),
                           ls.off->as<int64_t>(), ls.len->as<int64_t>(), noff.as<int64_t>(), nlist,
                           ls.code_size, nc.as<uint8_t>(), ls.lambdas ? nl.as<uint8_t>() : nullptr,
                           ni.as<int64_t>());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // new_off (host) and the old buffers die below
    if (e != hipSuccess) {
        nc.release(); nl.release(); ni.release(); noff.release();
        return fail(VLQ_ERR_HIP, "list relayout failed: %s", hipGetErrorString(e));
    }
    std::swap(*ls.codes, nc);
    std::swap(*ls.ids, ni);
    if (ls.lambdas) std::swap(*ls.lambdas, nl);
    std::swap(*ls.off, noff);
    nc.release(); nl.release(); ni.release(); noff.release();
    ls.h_off->swap(new_off);
    return VLQ_OK;
}

// room for num_vecs / nlist vectors in every list (IVFBase::reserveMemory, gpu/impl/IVFBase.cu:62-89)
int lists_reserve(ListStore& ls, int64_t num_vecs, hipStream_t s) {
    const int64_t per = num_vecs / ls.nlist;
    if (per < 1) return VLQ_OK;
    const std::vector<int64_t>& h_off = *ls.h_off;
    bool change = false;
    std::vector<int64_t> new_off((size_t)ls.nlist + 1, 0);
    for (int64_t i = 0; i < ls.nlist; i++) {
        const int64_t cap = h_off[(size_t)i + 1] - h_off[(size_t)i];
        if (cap < per) change = true;
        new_off[(size_t)i + 1] = new_off[(size_t)i] + std::max(cap, per);
    }
    return change ? lists_relayout(ls, new_off, s) : VLQ_OK;
}

// capacity == length for every list (IVFBase::reclaimMemory, gpu/impl/IVFBase.cu:136-166);
// *bytes = device bytes given back
int lists_reclaim(ListStore& ls, uint64_t* bytes, hipStream_t s) {
    const std::vector<int64_t>& h_len = *ls.h_len;
    const int64_t old_cap = (*ls.h_off)[(size_t)ls.nlist];
    std::vector<int64_t> new_off((size_t)ls.nlist + 1, 0);
    for (int64_t i = 0; i < ls.nlist; i++) new_off[(size_t)i + 1] = new_off[(size_t)i] + h_len[(size_t)i];
    const int64_t new_cap = new_off[(size_t)ls.nlist];
    if (bytes) *bytes = (uint64_t)(old_cap - new_cap) * (uint64_t)(ls.code_size + 8 + (ls.lambdas ? 1 : 0));
    if (new_cap == old_cap) return VLQ_OK;
    return lists_relayout(ls, new_off, s);
}

int lists_append(ListStore& ls, AppendWorkspace& ws, int64_t n, const int64_t* assign64,
                 const int32_t* assign32, const uint8_t* new_codes, const uint8_t* new_lambdas,
                 const int64_t* xids, int64_t id_base, hipStream_t s) {
    if (n <= 0) return VLQ_OK;
    if (n > 0x7FFFFFFFll) return fail(VLQ_ERR_INVALID, "add(): at most 2^31-1 vectors per call");
    if (ls.nlist >= 0xFFFFFFFFll) return fail(VLQ_ERR_UNSUPPORTED, "more than 2^32-2 lists");
    const int64_t nlist = ls.nlist;
    std::vector<int64_t>& h_off = *ls.h_off;
    std::vector<int64_t>& h_len = *ls.h_len;

    // 1. how many new vectors per list, and the (list, position) sort keys
    TRY(ws.cnt.reserve((size_t)nlist * 4));
    TRY(ws.keys_in.reserve((size_t)n * 8));
    TRY(ws.keys_out.reserve((size_t)n * 8));
    HIP_TRY(hipMemsetAsync(ws.cnt.p, 0, (size_t)nlist * 4, s));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (assign32)
        hipLaunchKernelGGL(count_and_key_kernel<int32_t>, dim3(grid), dim3(256), 0, s, assign32, n, nlist,
                           ws.cnt.as<int>(), ws.keys_in.as<unsigned long long>());
    else
        hipLaunchKernelGGL(count_and_key_kernel<int64_t>, dim3(grid), dim3(256), 0, s, assign64, n, nlist,
                           ws.cnt.as<int>(), ws.keys_in.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, ws.keys_in.as<unsigned long long>(),
                                     ws.keys_out.as<unsigned long long>(), (size_t)n, 0, 64, s));
    TRY(ws.sort_tmp.reserve(tmp_bytes ? tmp_bytes : 16));
    HIP_TRY(rocprim::radix_sort_keys(ws.sort_tmp.p, tmp_bytes, ws.keys_in.as<unsigned long long>(),
                                     ws.keys_out.as<unsigned long long>(), (size_t)n, 0, 64, s));
    std::vector<int> cnt((size_t)nlist);
    HIP_TRY(hipMemcpyAsync(cnt.data(), ws.cnt.p, (size_t)nlist * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));

    // 2. room?  If any list overflows its capacity the layout is rebuilt with 25 % slack.
    bool grow = false;
    std::vector<int64_t> cstart((size_t)nlist);
    int64_t run = 0;
    for (int64_t i = 0; i < nlist; i++) {
        cstart[(size_t)i] = run;
        run += cnt[(size_t)i];
        if (h_len[(size_t)i] + cnt[(size_t)i] > h_off[(size_t)i + 1] - h_off[(size_t)i]) grow = true;
    }
    if (grow) {
        std::vector<int64_t> new_off((size_t)nlist + 1, 0);
        for (int64_t i = 0; i < nlist; i++) {
            const int64_t need = h_len[(size_t)i] + cnt[(size_t)i];
            // never shrink a list that was given room by reserve
            const int64_t cap = std::max(need + need / 4, h_off[(size_t)i + 1] - h_off[(size_t)i]);
            new_off[(size_t)i + 1] = new_off[(size_t)i] + cap;
        }
        TRY(lists_relayout(ls, new_off, s));
    }

    // 3. place the batch, then publish the new lengths
    TRY(ws.cstart.reserve((size_t)nlist * 8));
    HIP_TRY(hipMemcpyAsync(ws.cstart.p, cstart.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(256), 0, s, ws.keys_out.as<unsigned long long>(), n,
                       ws.cstart.as<int64_t>(), ls.off->as<int64_t>(), ls.len->as<int64_t>(), new_codes,
                       new_lambdas, xids, id_base, ls.code_size, ls.codes->as<uint8_t>(),
                       ls.lambdas ? ls.lambdas->as<uint8_t>() : nullptr, ls.ids->as<int64_t>());
    HIP_TRY(hipGetLastError());
    for (int64_t i = 0; i < nlist; i++) h_len[(size_t)i] += cnt[(size_t)i];
    HIP_TRY(hipMemcpyAsync(ls.len->p, h_len.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));      // cstart / h_len host buffers are read by the copies above
    return VLQ_OK;
}

}  // namespace vlq
