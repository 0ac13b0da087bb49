/*
 * vlq_ivfpq.h -- C ABI of the MI355X-native IVF(PQ) list-scan search path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * Each entry point names the reference interface it replaces (paths relative to
 * the reference tree).  The C++ shell that mirrors faiss::Index / IndexIVFPQ /
 * gpu::GpuIndexIVFPQ on top of these calls is include/faiss_amd/ ; the binding a
 * maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions (Index.h:25-36, :62): vectors are row-major float32 x[i*d+j];
 * labels are int64; results are ascending squared-L2 distances; missing results
 * are label -1 / distance FLT_MAX (Heap.h:76-78,318-321).
 *
 * Pointers marked [h|d] may be host or device memory (the reference accepts
 * either, gpu/utils/CopyUtils.cuh); device pointers must belong to the index's
 * device.  All work is issued on the index's stream (vlq_ivfpq_set_stream); calls
 * with host output buffers return after the results have landed, calls whose
 * buffers are all device-resident are asynchronous on that stream.  Device inputs
 * produced on ANOTHER stream must be complete (or that stream must be the index's
 * stream) before the call: the private stream is non-blocking and does not wait
 * for the caller's streams -- the reference's GpuResources contract
 * (gpu/GpuResources.h:36, gpu/StandardGpuResources.cpp).
 *
 * Every function returns VLQ_OK or an error code; vlq_last_error() gives the
 * message of the last failure on the calling thread.  There is no CPU fallback:
 * without a HIP device every compute entry point fails with VLQ_ERR_HIP.
 */
#ifndef VLQ_IVFPQ_H
#define VLQ_IVFPQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vlq_ivfpq_s* vlq_ivfpq_t;

enum {
    VLQ_OK = 0,
    VLQ_ERR_INVALID = 1,      /* bad argument (FAISS_THROW_* in the reference)      */
    VLQ_ERR_HIP = 2,          /* HIP runtime failure (CUDA_VERIFY in the reference) */
    VLQ_ERR_UNSUPPORTED = 3,  /* outside the implemented envelope                   */
    VLQ_ERR_STATE = 4         /* index not trained / lists not loaded               */
};

#define VLQ_MAX_K 1024        /* gpu/impl/IVFPQ.cu:966-967 */
#define VLQ_MAX_NPROBE 1024
/* coarse stage of a multi-index quantizer (vlq_ivfpq_set_imi_centroids): MultiIndexQuantizer::search has no limit on k
 * (IndexPQ.cpp:804-857) and the reference's own drivers ask for 2048 cells (tests/sift1b_imi_pq.cpp:363) */
#define VLQ_MAX_IMI_NPROBE 4096

int vlq_version(void);
const char* vlq_last_error(void);
/* number of HIP devices visible (0 if none / no driver) */
int vlq_device_count(void);

/* GpuIndexIVFPQ(resources, dims, nlist, subQuantizers, bitsPerCode, METRIC_L2, cfg)
 * gpu/GpuIndexIVFPQ.h:52-58 ; IndexIVFPQ(quantizer, d, nlist, M, nbits) IndexIVFPQ.h:49-51.
 * Requires d % M == 0, 1 <= nbits <= 8 (IndexIVFPQ.cpp:51). */
int vlq_ivfpq_create(vlq_ivfpq_t* out, int device, int d, int nlist, int M, int nbits);
void vlq_ivfpq_destroy(vlq_ivfpq_t h);

/* GpuResources::getDefaultStream (gpu/GpuResources.h:36): run on the caller's
 * hipStream_t.  Until this is called the index uses a private non-blocking stream;
 * NULL selects the HIP null (legacy default) stream, e.g. torch's default stream. */
int vlq_ivfpq_set_stream(vlq_ivfpq_t h, void* hip_stream);

/* GpuIndexIVF::copyFrom: coarse centroids = IndexFlatL2::xb of the quantizer
 * (gpu/GpuIndexIVF.cu:120-150).  centroids: [nlist*d] [h|d]. */
int vlq_ivfpq_set_coarse_centroids(vlq_ivfpq_t h, const float* centroids);

/* MultiIndexQuantizer coarse quantizer with 2 sub-quantizers of imi_nbits each
 * (IndexPQ.h:124-160; "IMI2x14" of tests/sift1b_imi_pq.cpp): replaces the flat coarse
 * centroids.  The index must have been created with nlist = 4^imi_nbits; list key =
 * i0 | i1 << imi_nbits; the precomputed table becomes type 2 (IndexIVFPQ.cpp:430-457).
 * centroids: [2][2^imi_nbits][d/2] = MultiIndexQuantizer::pq.centroids.  [h|d] */
int vlq_ivfpq_set_imi_centroids(vlq_ivfpq_t h, int imi_nbits, const float* centroids);

/* ProductQuantizer::centroids [M][ksub][dsub] (ProductQuantizer.h:51-60),
 * GpuIndexIVFPQ::copyFrom gpu/GpuIndexIVFPQ.cu:168-231. [h|d] */
int vlq_ivfpq_set_pq_centroids(vlq_ivfpq_t h, const float* centroids);

/* IndexIVFPQ public search-time fields (IndexIVFPQ.h:30-41):
 * by_residual, use_precomputed_table (0 or 1; 1 builds the table on the device,
 * = IndexIVFPQ::precompute_table IndexIVFPQ.cpp:392-429), max_codes (0 = off). */
int vlq_ivfpq_set_search_options(vlq_ivfpq_t h, int by_residual, int use_precomputed_table,
                                 int64_t max_codes);

/* Bulk load of the inverted lists (copyFrom of IndexIVF::ids / IndexIVFPQ::codes,
 * IndexIVF.h:55, IndexIVFPQ.h:43) in list-contiguous form:
 *   codes[ntotal][M] u8, ids[ntotal] i64, list_offsets[nlist+1] i64.  [h|d] */
int vlq_ivfpq_set_lists(vlq_ivfpq_t h, const uint8_t* codes, const int64_t* ids,
                        const int64_t* list_offsets);

/* IndexIVFPQ::add_with_ids (IndexIVFPQ.cpp:186-272) on the device: coarse 1-NN,
 * residual, per-sub-quantizer argmin (first minimum wins), append in input order.
 * xids may be NULL (ids ntotal..ntotal+n-1).  x [h|d], xids [h|d]. */
int vlq_ivfpq_add(vlq_ivfpq_t h, int64_t n, const float* x, const int64_t* xids);

/* GpuIndexIVFPQ::reserveMemory (gpu/GpuIndexIVFPQ.cu:283-290 -> IVFBase::reserveMemory,
 * gpu/impl/IVFBase.cu:62-89): room for num_vecs / nlist vectors in every list, so the adds that
 * follow append in place.  GpuIndexIVFPQ::reclaimMemory (:323-330 -> IVFBase.cu:136-166): give
 * the slack back (capacity == length); *bytes_reclaimed (may be NULL) = device bytes freed. */
int vlq_ivfpq_reserve_memory(vlq_ivfpq_t h, int64_t num_vecs);
int vlq_ivfpq_reclaim_memory(vlq_ivfpq_t h, uint64_t* bytes_reclaimed);

/* Encode only: assign[n] i64 and codes[n][M] u8 (IndexIVFPQ::encode_multiple with
 * compute_keys=true, IndexIVFPQ.cpp:150-167).  Outputs [h|d]. */
int vlq_ivfpq_encode(vlq_ivfpq_t h, int64_t n, const float* x, int64_t* assign, uint8_t* codes);
/* The same with the lists given: codes[n][M] of x for assign[n] (IndexIVFPQ::encode_multiple with
 * compute_keys = false, IndexIVFPQ.cpp:150-167; the `precomputed_idx` path of add_core_o, :192-211).  A vector
 * with assign < 0 is encoded as if its residual were zero (IndexIVFPQ.cpp:219-221).  All buffers [h|d]. */
int vlq_ivfpq_encode_preassigned(vlq_ivfpq_t h, int64_t n, const float* x, const int64_t* assign, uint8_t* codes);

int64_t vlq_ivfpq_ntotal(vlq_ivfpq_t h);
/* GpuIndexIVFPQ::getListLength / getListCodes / getListIndices
 * (gpu/GpuIndexIVFPQ.h:120-131).  codes_out/ids_out are host buffers. */
int vlq_ivfpq_list_length(vlq_ivfpq_t h, int list_id, int64_t* len);
int vlq_ivfpq_get_list(vlq_ivfpq_t h, int list_id, uint8_t* codes_out, int64_t* ids_out);

/* GpuIndexIVFPQConfig::useFloat16LookupTables (gpu/GpuIndexIVFPQ.h:24-38) for the plain IVFPQ search, M = 16 x 8
 * bit in table mode 1, k <= 256 (other shapes and larger k keep fp32 tables).  As in the reference's kernel
 * (gpu/impl/PQScanMultiPassPrecomputed.cu:30-114,375-477 with LookupT = half): term 2 and term 3 are kept as half
 * (impl/IVFPQ.cu:599-684, :1599-1680), a list's table is their half sum, looked-up entries are added in float.
 * Off by default: the fp32 tables are the parity build against the CPU library; with half tables the
 * reference's own GPU-vs-CPU bar applies (gpu/test/TestGpuIndexIVFPQ.cpp:89-99: relative distance error
 * <= 0.015, <= 30 % of the results differing in fp16 mode).  The search returns VLQ_ERR_UNSUPPORTED when a
 * term-2 entry lies outside the half range (|value| > 65504: byte-valued data such as SIFT; fine for
 * normalised descriptors) -- the reference would silently hold infinities there. */
int vlq_ivfpq_set_float16_tables(vlq_ivfpq_t h, int enable);

/* Scheduling of the 16-byte-code scan kernel -- SPEED ONLY, results are identical in every mode:
 *   0  automatic: query-major today -- the list-owned schedule moves fewer bytes but is slower on every data
 *      set measured, so nothing selects it by itself
 *   1  query-major: one workgroup per query walks all its probes
 *   2  list-owned: the lists are cut into 8 partitions of neighbouring lists, one per XCD; one workgroup
 *      per (query, partition) scans the query's probes of that partition, a merge joins the parts.  Keeps
 *      the term2 rows (IndexIVFPQ.cpp:641-644) of a partition in that XCD's L2.
 * Replaces nothing in the reference (its CPU loop has no such choice). */
int vlq_ivfpq_set_scan_schedule(vlq_ivfpq_t h, int mode);

/* Float16 screen of the coarse stage -- SPEED ONLY, results are identical with and without it (csrc/coarse_screen.hip):
 * batches of 2048 queries and more on coarse quantizers of 256 .. 2^20 centroids (flat, or each half of a multi-index;
 * d <= 128, nprobe 1 .. 128: searches, and the assignment of add / encode) first get an
 * APPROXIMATE distance matrix from half copies of queries and centroids; a rigorous bound on |approximate - exact| keeps
 * every column that can still be among a row's nprobe nearest, and only those get the exact fp32 distance of the matrix
 * path (the same k-ascending fmaf chain), then the same (distance, column) selection.  Rows the bound cannot decide are
 * done exactly in full; an index where that happens to more than 0.5 % of the rows drops the screen by itself.
 *   mode 0: off (f32 MFMA distance matrix for every batch), 1: on where the shape allows (the default).
 * vlq_ivfpq_coarse_screen_state: *enabled = the screen is (still) in use for this index, *rows = rows that went through it,
 * *undecided = rows it handed to the exact path (as of the last batch whose counters have reached the host).
 * Replaces nothing in the reference (knn_L2sqr computes every distance, utils.cpp:935-946). */
int vlq_ivfpq_set_coarse_screen(vlq_ivfpq_t h, int mode);
int vlq_ivfpq_coarse_screen_state(vlq_ivfpq_t h, int* enabled, uint64_t* rows, uint32_t* undecided);

/* IndexIVFPQ::search (IndexIVFPQ.cpp:1063-1081) = GpuIndexIVFPQ::search.
 * x[n*d], D[n*k], I[n*k]   [h|d].   nprobe <= 1024 (multi-index quantizer: <= VLQ_MAX_IMI_NPROBE, scanned in runs of 1024
 * like vlq_ivfpq_search_preassigned below; max_codes must be 0 then), k <= 1024.
 * Host D / I: the call returns with the rows in place.  PAGE-LOCKED host D / I (hipHostMalloc / hipHostRegister:
 * what GpuResources::getPinnedMemory hands out, gpu/GpuResources.h:40) are written by the scan kernel itself --
 * no staging buffer, no copy-out; pageable ones are staged through device memory and copied. */
int vlq_ivfpq_search(vlq_ivfpq_t h, int64_t n, const float* x, int nprobe, int k,
                     float* D, int64_t* I);

/* The parity seam: IndexIVFPQ::search_knn_with_key (IndexIVFPQ.h:140-146,
 * IndexIVFPQ.cpp:964-1060).  keys[n*nprobe] (-1 = skip), coarse_dis[n*nprobe].
 * store_pairs != 0 returns list<<32|offset as label.  All buffers [h|d].
 * nprobe may exceed VLQ_MAX_NPROBE here (the CPU class has no limit: tests/sift1b_imi_pq.cpp asks for 2048; up to 64 x
 * VLQ_MAX_NPROBE): the probe list is then scanned in runs of 1024 in coarse order and the rows joined by (distance, run,
 * place in the run's row) -- the (distance, scan position) order of one long scan.
 * A key >= nlist aborts the reference's search (IndexIVFPQ.cpp:1008-1011): with a host D or I
 * the call synchronises and returns VLQ_ERR_INVALID itself; with device D and I the call stays
 * asynchronous, the offending probe is skipped and the error is returned by the next
 * vlq_ivfpq_stats() (which also clears it). */
int vlq_ivfpq_search_preassigned(vlq_ivfpq_t h, int64_t n, const float* x, const int64_t* keys,
                                 const float* coarse_dis, int nprobe, int k, float* D,
                                 int64_t* I, int store_pairs);

/* quantizer->search (IndexIVFPQ.cpp:1073 -> IndexFlat.cpp:42-56; MultiIndexQuantizer::search IndexPQ.cpp:804-857):
 * top-nprobe coarse centroids, ascending (multi-index: the cells in MinSumK's order, nprobe <= VLQ_MAX_IMI_NPROBE).
 * Outputs [h|d]. */
int vlq_ivfpq_coarse_search(vlq_ivfpq_t h, int64_t n, const float* x, int nprobe,
                            float* coarse_dis, int64_t* keys);

/* Merge of per-shard results for indexes whose inverted lists are split over GPUs / ranks
 * (GpuIndexIVFPQ::merge, gpu/GpuIndexIVFPQ.cu:1467-1591, used by gpu/test/deep1b16_query.cpp
 * after the gather; IndexShards::search merge, MetaIndexes.cpp:486-557).  D_parts / I_parts
 * are [nparts][nq][k] DEVICE buffers (e.g. the output of an all-gather); D / I [nq][k]
 * device.  Ties go to the lower part, then the lower rank. */
int vlq_merge_topk(int device, void* hip_stream, int64_t nq, int k, int nparts, const float* D_parts,
                   const int64_t* I_parts, float* D, int64_t* I);

/* Introspection for parity tests (host output buffers):
 *   query tables  = ProductQuantizer::compute_inner_prod_table /
 *                   compute_distance_table (ProductQuantizer.cpp:410-436): out[n][M][ksub]
 *   precomputed   = IndexIVFPQ::precomputed_table [nlist][M][ksub]               */
int vlq_ivfpq_query_tables(vlq_ivfpq_t h, int64_t n, const float* x, int inner_product, float* out);
int vlq_ivfpq_get_precomputed_table(vlq_ivfpq_t h, float* out);

/* indexIVFPQ_stats (IndexIVFPQ.h:169-195): codes visited since the last reset. */
int vlq_ivfpq_stats(vlq_ivfpq_t h, uint64_t* nq, uint64_t* ncode, int reset);

/* Introspection, speed only (replaces nothing in the reference): what the last search's list scan was -- the kernel shape
 * chosen for the batch, the walking order of a query's probes (csrc/walk_order.cuh: decided on the device from the lists
 * neighbouring queries share) and the clock period the walk ran with -- as text:
 *   "kernel=scan16_kernel<1, 2, 1, false, false, false> order=list-id walk first=1 shared=123/8192 limit=2457 period_ticks=9876".
 * Synchronises the stream.  vlq_ivfpq_reset_walk_state forgets the measured walk times: the next search runs as the first
 * search of a fresh handle does (what a driver that calls search once gets; bench.py's cold_walk_ms). */
int vlq_ivfpq_last_scan_info(vlq_ivfpq_t h, char* buf, int cap);
int vlq_ivfpq_reset_walk_state(vlq_ivfpq_t h);

/* HIP-event timing of the stages of the search calls issued since the last
 * reset, on the index's stream: ms[0]=coarse (norms+GEMM+select), ms[1]=query
 * tables, ms[2]=list scan (+top-k), calls = number of timed scan launches.
 * enable: 0 = off, 1 = every stage (six event records per call), 2 = the scan kernel only
 * (two event records per call), 3 = the scan kernel of every 4th call, starting with the
 * next one (an event record is a queue barrier: it costs the call more than it measures). */
int vlq_ivfpq_profile(vlq_ivfpq_t h, int enable);
int vlq_ivfpq_profile_read(vlq_ivfpq_t h, double ms[3], int64_t* calls, int reset);

#ifdef __cplusplus
}
#endif
#endif /* VLQ_IVFPQ_H */
