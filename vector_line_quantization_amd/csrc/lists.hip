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

// old list i moves from old_off[i] to new_off[i].  A workgroup owns 256 consecutive lists and reads
// their metadata coalesced: short lists (the many-list indexes: a handful of codes each) are copied
// by their own thread, long ones are queued in LDS and copied by the whole workgroup.
constexpr int kShortList = 32;
__global__ __launch_bounds__(256) void relayout_kernel(const uint8_t* __restrict__ oc, const uint8_t* __restrict__ ol,
                                                       const int64_t* __restrict__ oi, const int64_t* __restrict__ old_off,
                                                       const int64_t* __restrict__ len, const int64_t* __restrict__ new_off,
                                                       int64_t nlist, int code_size, uint8_t* __restrict__ nc,
                                                       uint8_t* __restrict__ nl, int64_t* __restrict__ ni) {
    __shared__ int longs[256];
    __shared__ int nlong;
    if (threadIdx.x == 0) nlong = 0;
    __syncthreads();
    const int64_t l0 = (int64_t)blockIdx.x * 256;
    {
        const int64_t l = l0 + threadIdx.x;
        const int64_t n = l < nlist ? len[l] : 0;
        if (n > kShortList) longs[atomicAdd(&nlong, 1)] = threadIdx.x;
        else if (n > 0) {
            const int64_t so = old_off[l], dn = new_off[l];
            if (code_size == 16) {
                for (int64_t j = 0; j < n; j++)
                    reinterpret_cast<uint4*>(nc)[dn + j] = reinterpret_cast<const uint4*>(oc)[so + j];
            } else {
                for (int64_t j = 0; j < n * code_size; j++) nc[dn * code_size + j] = oc[so * code_size + j];
            }
            for (int64_t j = 0; j < n; j++) ni[dn + j] = oi[so + j];
            if (ol) for (int64_t j = 0; j < n; j++) nl[dn + j] = ol[so + j];
        }
    }
    __syncthreads();
    for (int k = 0; k < nlong; k++) {
        const int64_t l = l0 + longs[k];
        const int64_t n = len[l];
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

// append bookkeeping on the device (one thread per list)
__global__ void overflow_kernel(const int* __restrict__ cnt, const int64_t* __restrict__ len,
                                const int64_t* __restrict__ off, int64_t nlist, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlist) return;
    const int c = cnt[i];
    if (c > 0 && len[i] + c > off[i + 1] - off[i]) *flag = 1;
}
// capacities after growth: 25 % slack, never below the current capacity (reserve); [nlist] = 0
__global__ void grow_caps_kernel(const int* __restrict__ cnt, const int64_t* __restrict__ len,
                                 const int64_t* __restrict__ off, int64_t nlist, int64_t* __restrict__ cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nlist) return;
    if (i == nlist) { cap[i] = 0; return; }
    const int64_t need = len[i] + cnt[i];
    const int64_t have = off[i + 1] - off[i];
    const int64_t want = need + need / 4;
    cap[i] = want > have ? want : have;
}
__global__ void add_counts_kernel(const int* __restrict__ cnt, int64_t nlist, int64_t* __restrict__ len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nlist && cnt[i]) len[i] += cnt[i];
}
struct IntToI64 {
    __host__ __device__ int64_t operator()(int v) const { return (int64_t)v; }
};

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

// refresh the host copies of the list starts / lengths after device-side appends
int lists_sync_host(ListStore& ls, hipStream_t s) {
    if (!ls.h_stale || !*ls.h_stale) return VLQ_OK;
    ls.h_off->resize((size_t)ls.nlist + 1);
    ls.h_len->resize((size_t)ls.nlist);
    HIP_TRY(hipMemcpyAsync(ls.h_off->data(), ls.off->p, ((size_t)ls.nlist + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ls.h_len->data(), ls.len->p, (size_t)ls.nlist * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *ls.h_stale = false;
    return VLQ_OK;
}

// move every list to noff[i] (device array [nlist+1], capacities noff[i+1] - noff[i] >= len[i],
// noff[nlist] = cap); takes ownership of noff; synchronises
static int relayout_dev(ListStore& ls, DevBuf& noff, int64_t cap, hipStream_t s) {
    const int64_t nlist = ls.nlist;
    DevBuf nc, nl, ni;
    int rc = nc.reserve((size_t)cap * ls.code_size + 16);
    if (rc == VLQ_OK) rc = ni.reserve((size_t)cap * 8 + 16);
    if (rc == VLQ_OK && ls.lambdas) rc = nl.reserve((size_t)cap + 16);
    if (rc != VLQ_OK) { nc.release(); nl.release(); ni.release(); noff.release(); return rc; }
    hipError_t e = hipSuccess;
    if (ls.codes->p) {
        const unsigned g = (unsigned)((nlist + 255) / 256);
        hipLaunchKernelGGL(relayout_kernel, dim3(g), dim3(256), 0, s, ls.codes->as<uint8_t>(),
                           ls.lambdas ? ls.lambdas->as<uint8_t>() : nullptr, ls.ids->as<int64_t>(),
                           ls.off->as<int64_t>(), ls.len->as<int64_t>(), noff.as<int64_t>(), nlist,
                           ls.code_size, nc.as<uint8_t>(), ls.lambdas ? nl.as<uint8_t>() : nullptr,
                           ni.as<int64_t>());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // the old buffers die below
    if (e != hipSuccess) {
        nc.release(); nl.release(); ni.release(); noff.release();
        return fail(VLQ_ERR_HIP, "list relayout failed: %s", hipGetErrorString(e));
    }
    std::swap(*ls.codes, nc);
    std::swap(*ls.ids, ni);
    if (ls.lambdas) std::swap(*ls.lambdas, nl);
    std::swap(*ls.off, noff);
    nc.release(); nl.release(); ni.release(); noff.release();
    return VLQ_OK;
}

// same, from a host array (reserve / reclaim); the host copies must be current
int lists_relayout(ListStore& ls, std::vector<int64_t>& new_off, hipStream_t s) {
    const int64_t nlist = ls.nlist;
    DevBuf noff;
    TRY(noff.reserve(((size_t)nlist + 1) * 8));
    hipError_t e = hipMemcpyAsync(noff.p, new_off.data(), ((size_t)nlist + 1) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { noff.release(); return fail(VLQ_ERR_HIP, "list relayout failed: %s", hipGetErrorString(e)); }
    TRY(relayout_dev(ls, noff, new_off[(size_t)nlist], s));
    ls.h_off->swap(new_off);
    return VLQ_OK;
}

// room for num_vecs / nlist vectors in every list (IVFBase::reserveMemory, gpu/impl/IVFBase.cu:62-89)
int lists_reserve(ListStore& ls, int64_t num_vecs, hipStream_t s) {
    const int64_t per = num_vecs / ls.nlist;
    if (per < 1) return VLQ_OK;
    TRY(lists_sync_host(ls, s));
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
    TRY(lists_sync_host(ls, s));
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
                 const int64_t* xids, int64_t id_base, hipStream_t s, int64_t* placed, bool* relaid) {
    if (placed) *placed = 0;
    if (relaid) *relaid = false;
    if (n <= 0) return VLQ_OK;
    if (n > 0x7FFFFFFFll) return fail(VLQ_ERR_INVALID, "add(): at most 2^31-1 vectors per call");
    if (ls.nlist >= 0xFFFFFFFFll) return fail(VLQ_ERR_UNSUPPORTED, "more than 2^32-2 lists");
    const int64_t nlist = ls.nlist;

    // 1. how many new vectors per list, and the (list, position) sort keys
    TRY(ws.cnt.reserve((size_t)nlist * 4 + 16));
    TRY(ws.cstart.reserve(((size_t)nlist + 1) * 8));
    TRY(ws.keys_in.reserve((size_t)n * 8));
    TRY(ws.keys_out.reserve((size_t)n * 8));
    HIP_TRY(hipMemsetAsync(ws.cnt.p, 0, (size_t)nlist * 4 + 16, s));       // [nlist] counts | overflow flag
    int* flag = ws.cnt.as<int>() + nlist;
    const unsigned grid = (unsigned)((n + 255) / 256);
    const unsigned lgrid = (unsigned)((nlist + 1 + 255) / 256);
    if (assign32)
        hipLaunchKernelGGL(count_and_key_kernel<int32_t>, dim3(grid), dim3(256), 0, s, assign32, n, nlist,
                           ws.cnt.as<int>(), ws.keys_in.as<unsigned long long>());
    else
        hipLaunchKernelGGL(count_and_key_kernel<int64_t>, dim3(grid), dim3(256), 0, s, assign64, n, nlist,
                           ws.cnt.as<int>(), ws.keys_in.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
    auto cnt64 = rocprim::make_transform_iterator(ws.cnt.as<int>(), IntToI64());
    size_t sort_bytes = 0, scan_bytes = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, sort_bytes, ws.keys_in.as<unsigned long long>(),
                                     ws.keys_out.as<unsigned long long>(), (size_t)n, 0, 64, s));
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, cnt64, ws.cstart.as<int64_t>(), (int64_t)0,
                                    (size_t)nlist + 1, rocprim::plus<int64_t>(), s));
    TRY(ws.sort_tmp.reserve(std::max(std::max(sort_bytes, scan_bytes), (size_t)16)));
    HIP_TRY(rocprim::radix_sort_keys(ws.sort_tmp.p, sort_bytes, ws.keys_in.as<unsigned long long>(),
                                     ws.keys_out.as<unsigned long long>(), (size_t)n, 0, 64, s));
    // cstart[i] = batch vectors in lists < i; cstart[nlist] = vectors placed (cnt[nlist] is the zero flag)
    HIP_TRY(rocprim::exclusive_scan(ws.sort_tmp.p, scan_bytes, cnt64, ws.cstart.as<int64_t>(), (int64_t)0,
                                    (size_t)nlist + 1, rocprim::plus<int64_t>(), s));

    // 2. room?  If any list overflows its capacity the layout is rebuilt with 25 % slack -- the
    //    per-list arithmetic stays on the device; the host sees one flag and two totals.
    hipLaunchKernelGGL(overflow_kernel, dim3(lgrid), dim3(256), 0, s, ws.cnt.as<int>(), ls.len->as<int64_t>(),
                       ls.off->as<int64_t>(), nlist, flag);
    HIP_TRY(hipGetLastError());
    int grow = 0;
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&grow, flag, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&total, ws.cstart.as<int64_t>() + nlist, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (grow) {
        DevBuf caps, noff;
        int rc = caps.reserve(((size_t)nlist + 1) * 8);
        if (rc == VLQ_OK) rc = noff.reserve(((size_t)nlist + 1) * 8);
        if (rc != VLQ_OK) { caps.release(); noff.release(); return rc; }
        hipLaunchKernelGGL(grow_caps_kernel, dim3(lgrid), dim3(256), 0, s, ws.cnt.as<int>(), ls.len->as<int64_t>(),
                           ls.off->as<int64_t>(), nlist, caps.as<int64_t>());
        size_t b2 = 0;
        hipError_t e = rocprim::exclusive_scan(nullptr, b2, caps.as<int64_t>(), noff.as<int64_t>(), (int64_t)0,
                                               (size_t)nlist + 1, rocprim::plus<int64_t>(), s);
        if (e == hipSuccess && ws.sort_tmp.reserve(std::max(b2, (size_t)16)) != VLQ_OK) e = hipErrorOutOfMemory;
        if (e == hipSuccess)
            e = rocprim::exclusive_scan(ws.sort_tmp.p, b2, caps.as<int64_t>(), noff.as<int64_t>(), (int64_t)0,
                                        (size_t)nlist + 1, rocprim::plus<int64_t>(), s);
        int64_t cap = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&cap, noff.as<int64_t>() + nlist, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        caps.release();
        if (e != hipSuccess) { noff.release(); return fail(VLQ_ERR_HIP, "list growth failed: %s", hipGetErrorString(e)); }
        TRY(relayout_dev(ls, noff, cap, s));
        if (relaid) *relaid = true;
    }

    // 3. place the batch, then publish the new lengths
    hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(256), 0, s, ws.keys_out.as<unsigned long long>(), n,
                       ws.cstart.as<int64_t>(), ls.off->as<int64_t>(), ls.len->as<int64_t>(), new_codes,
                       new_lambdas, xids, id_base, ls.code_size, ls.codes->as<uint8_t>(),
                       ls.lambdas ? ls.lambdas->as<uint8_t>() : nullptr, ls.ids->as<int64_t>());
    hipLaunchKernelGGL(add_counts_kernel, dim3(lgrid), dim3(256), 0, s, ws.cnt.as<int>(), nlist, ls.len->as<int64_t>());
    HIP_TRY(hipGetLastError());
    if (ls.h_stale) *ls.h_stale = true;
    if (placed) *placed = total;
    return VLQ_OK;
}

}  // namespace vlq
