// Device-resident inverted lists: one flat array per field (codes, optional lambda bytes,
// ids), list i at [off[i], off[i] + len[i]) with capacity off[i+1] - off[i].  Lists loaded
// with set_lists are packed (capacity == length); lists grown by add() get 25 % slack when
// the layout has to be rebuilt, so appending is amortised O(batch) and never leaves the
// device: counts, prefix sums, the overflow test and the new layout are computed there, the host
// sees one flag and two totals per call.  Replaces the per-list growable device vectors of the reference
// (gpu/impl/IVFBase.cuh:109-143, gpu/impl/InvertedListAppend.cu:122-247).
#pragma once
#include "handle.h"

namespace vlq {

struct ListStore {
    int64_t nlist = 0;
    int code_size = 0;
    DevBuf* codes = nullptr;        // [cap_total][code_size]
    DevBuf* lambdas = nullptr;      // [cap_total] or nullptr (plain IVFPQ)
    DevBuf* ids = nullptr;          // [cap_total] int64
    DevBuf* off = nullptr;          // [nlist+1] int64 list starts (off[nlist] = cap_total)
    DevBuf* len = nullptr;          // [nlist] int64
    // host copies of off / len for the per-list accessors; *h_stale after a device-side append
    std::vector<int64_t>* h_off = nullptr;
    std::vector<int64_t>* h_len = nullptr;
    bool* h_stale = nullptr;
};

typedef AppendWs AppendWorkspace;   // handle.h
inline void release(AppendWorkspace& w) { w.cnt.release(); w.cstart.release(); w.keys_in.release(); w.keys_out.release(); w.sort_tmp.release(); }

// Append n encoded vectors: vector i goes to the end of list assign[i] (assign[i] < 0: dropped,
// IndexIVFPQ.cpp:238-243), vectors of one list keep their input order (:236-248).  All
// pointers are device pointers; id of vector i = xids ? xids[i] : id_base + i (:244).
// assign32 != nullptr: int32 list ids (VLQ lines) instead of assign64.
int lists_append(ListStore& ls, AppendWorkspace& ws, int64_t n, const int64_t* assign64,
                 const int32_t* assign32, const uint8_t* new_codes, const uint8_t* new_lambdas,
                 const int64_t* xids, int64_t id_base, hipStream_t s, int64_t* placed = nullptr, bool* relaid = nullptr);
// (*relaid: the append rebuilt the layout -- every list may have moved; otherwise the batch sits behind the old ends of the
// lists and ws.cnt[i] (int) is the number of vectors list i received)

int lists_sync_host(ListStore& ls, hipStream_t s);
int lists_relayout(ListStore& ls, std::vector<int64_t>& new_off, hipStream_t s);
int lists_reserve(ListStore& ls, int64_t num_vecs, hipStream_t s);
int lists_reclaim(ListStore& ls, uint64_t* bytes, hipStream_t s);

}  // namespace vlq
