// faiss::gpu::GpuIndexIVFPQ construction and search surface
// (gpu/GpuIndexIVFPQ.h:24-234, gpu/GpuIndexIVF.h:25-95, gpu/GpuIndex.h:19-102,
// gpu/GpuIndexFlat.h:27-49) over the MI355X library, including the fork's VLQ
// constructor (nedge, nLambda) with its train / add / search path (vlq_line.h).
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "compat.h"
#include "../../vlq_line.h"
#include "GpuIndicesOptions.h"
#include "GpuResources.h"

namespace faiss { namespace gpu {

enum class MemorySpace { Device = 1, Unified = 2 };   // gpu/utils/MemorySpace.h

struct GpuIndexConfig {
  GpuIndexConfig() : device(0), memorySpace(MemorySpace::Device) {}
  int device;
  MemorySpace memorySpace;
};
struct GpuIndexFlatConfig : public GpuIndexConfig {
  GpuIndexFlatConfig() : useFloat16(false), useFloat16Accumulator(false), storeTransposed(false) {}
  bool useFloat16;              ///< accepted; the coarse quantizer stays fp32 here (superset precision)
  bool useFloat16Accumulator;
  bool storeTransposed;         ///< layout hint of the reference's cuBLAS call; no effect here
};
struct GpuIndexIVFConfig : public GpuIndexConfig {
  GpuIndexIVFConfig() : indicesOptions(INDICES_64_BIT) {}
  IndicesOptions indicesOptions;
  GpuIndexFlatConfig flatConfig;
};
struct GpuIndexIVFPQConfig : public GpuIndexIVFConfig {
  GpuIndexIVFPQConfig() : useFloat16LookupTables(false), usePrecomputedTables(false) {}
  bool useFloat16LookupTables;  ///< VLQ search (16 x 8 bit): half tables as in the reference; IVFPQ search: fp32 (superset precision)
  bool usePrecomputedTables;
};

class GpuIndex : public faiss::Index {
 public:
  GpuIndex(GpuResources* resources, int dims, faiss::MetricType metric, GpuIndexConfig config)
      : Index(dims, metric), resources_(resources), device_(config.device), memorySpace_(config.memorySpace) {
    FAISS_THROW_IF_NOT_MSG(resources_, "null GpuResources");
    FAISS_THROW_IF_NOT_MSG(metric == METRIC_L2, "only METRIC_L2 is built");
    resources_->initializeForDevice(device_);
  }
  int getDevice() const { return device_; }
  GpuResources* getResources() { return resources_; }

 protected:
  GpuResources* resources_;
  const int device_;
  const MemorySpace memorySpace_;
};

class GpuIndexIVFPQ : public GpuIndex {
 public:
  /// copy-construct from a trained CPU index (gpu/GpuIndexIVFPQ.h:45-49)
  GpuIndexIVFPQ(GpuResources* resources, const faiss::IndexIVFPQ* index,
                GpuIndexIVFPQConfig config = GpuIndexIVFPQConfig())
      : GpuIndex(resources, index->d, index->metric_type, config), ivfpqConfig_(config),
        nlist_((int)index->nlist), nprobe_(1), subQuantizers_(0), bitsPerCode_(0), reserveMemoryVecs_(0) {
    verifyConfig_();
    copyFrom(index);
  }
  /// empty index (gpu/GpuIndexIVFPQ.h:52-58)
  GpuIndexIVFPQ(GpuResources* resources, int dims, int nlist, int subQuantizers, int bitsPerCode,
                faiss::MetricType metric, GpuIndexIVFPQConfig config = GpuIndexIVFPQConfig())
      : GpuIndex(resources, dims, metric, config), ivfpqConfig_(config), nlist_(nlist), nprobe_(1),
        subQuantizers_(subQuantizers), bitsPerCode_(bitsPerCode), reserveMemoryVecs_(0) {
    verifyConfig_();
    FAISS_THROW_IF_NOT_MSG(bitsPerCode_ >= 1 && bitsPerCode_ <= 8, "Bits per code must be <= 8");
    FAISS_THROW_IF_NOT_MSG(dims % subQuantizers_ == 0, "Number of sub-quantizers must be an even divisor of the dimensions");
    is_trained = false;
    create_();
  }
  /// the fork's VLQ index (gpu/GpuIndexIVFPQ.h:60-68): lines between each coarse centroid
  /// and its `nedge` nearest centroids, `nLambda` scalar positions along a line
  GpuIndexIVFPQ(GpuResources* resources, int dims, int nlist, int subQuantizers, int bitsPerCode, int nedge,
                int nLambda, faiss::MetricType metric, GpuIndexIVFPQConfig config = GpuIndexIVFPQConfig())
      : GpuIndex(resources, dims, metric, config), nLambda_(nLambda), numedge_(nedge), begin_(0), end_(0),
        w1_(1), ivfpqConfig_(config), nlist_(nlist), nprobe_(1), subQuantizers_(subQuantizers),
        bitsPerCode_(bitsPerCode), reserveMemoryVecs_(0) {
    verifyConfig_();
    FAISS_THROW_IF_NOT_MSG(bitsPerCode_ >= 1 && bitsPerCode_ <= 8, "Bits per code must be <= 8");
    FAISS_THROW_IF_NOT_MSG(dims % subQuantizers_ == 0, "Number of sub-quantizers must be an even divisor of the dimensions");
    is_trained = false;
    VLQ_CHECK(vlq_line_create(&line_, device_, d, nlist_, subQuantizers_, bitsPerCode_, numedge_, nLambda_));
    VLQ_CHECK(vlq_line_set_stream(line_, (void*)resources_->getDefaultStream(device_)));
    // the VLQ search honours float16 look-up tables for the drivers' shape (16 x 8-bit codes), built as the
    // reference builds them (vlq_line.h); other shapes and the plain IVFPQ path compute in fp32
    if (ivfpqConfig_.useFloat16LookupTables && subQuantizers_ == 16 && bitsPerCode_ == 8)
      VLQ_CHECK(vlq_line_set_float16_tables(line_, 1));
    edgeInfoV_.resize((size_t)nlist_ * numedge_);
    edgeDistInfoV_.resize((size_t)nlist_ * numedge_);
    lambdaInfoV_.resize(nLambda_);
    edgeInfo_ = edgeInfoV_.data(); edgeDistInfo_ = edgeDistInfoV_.data(); lambdaInfo_ = lambdaInfoV_.data();
  }
  ~GpuIndexIVFPQ() override { if (h_) vlq_ivfpq_destroy(h_); if (line_) vlq_line_destroy(line_); }

  // public VLQ state of the reference class (gpu/GpuIndexIVFPQ.h:70-81)
  int nLambda_ = 0;
  int numedge_ = 0;
  int begin_ = 0, end_ = 0;   ///< MPI list range of the fork's drivers; unused (query sharding instead)
  int w1_ = 1;                ///< lines kept per query
  int* edgeInfo_ = nullptr;
  float* edgeDistInfo_ = nullptr;
  float* lambdaInfo_ = nullptr;
  bool isVLQ() const { return line_ != nullptr; }
  GpuIndexIVFPQ(const GpuIndexIVFPQ&) = delete;
  GpuIndexIVFPQ& operator=(const GpuIndexIVFPQ&) = delete;

  /// gpu/GpuIndexIVFPQ.cu:168-231
  void copyFrom(const faiss::IndexIVFPQ* index) {
    FAISS_THROW_IF_NOT_MSG(index->pq.byte_per_idx == 1, "GPU: only pq.byte_per_idx == 1 is supported");
    FAISS_THROW_IF_NOT_MSG(index->by_residual, "GPU: only by_residual = true is supported");
    FAISS_THROW_IF_NOT_MSG(index->polysemous_ht == 0, "GPU: polysemous codes not supported");
    const IndexFlat* flat = dynamic_cast<const IndexFlat*>(index->quantizer);
    FAISS_THROW_IF_NOT_MSG(flat && flat->metric_type == METRIC_L2,
                           "Only IndexFlatL2 is supported as the coarse quantizer (gpu/GpuIndexIVF.cu:131-133)");
    d = index->d; metric_type = index->metric_type;
    nlist_ = (int)index->nlist; nprobe_ = (int)index->nprobe;
    subQuantizers_ = (int)index->pq.M; bitsPerCode_ = (int)index->pq.nbits;
    if (h_) { vlq_ivfpq_destroy(h_); h_ = nullptr; }
    create_();
    is_trained = index->is_trained;
    ntotal = 0;
    if (!index->is_trained) return;
    FAISS_THROW_IF_NOT(flat->ntotal == (idx_t)index->nlist);
    coarse_ = flat->xb;
    pqCentroids_ = index->pq.centroids;
    VLQ_CHECK(vlq_ivfpq_set_coarse_centroids(h_, coarse_.data()));
    VLQ_CHECK(vlq_ivfpq_set_pq_centroids(h_, pqCentroids_.data()));
    usePrecomputed_ = ivfpqConfig_.usePrecomputedTables || index->use_precomputed_table == 1;
    VLQ_CHECK(vlq_ivfpq_set_search_options(h_, 1, usePrecomputed_ ? 1 : 0, (int64_t)index->max_codes));
    std::vector<int64_t> off(nlist_ + 1, 0);
    for (int i = 0; i < nlist_; i++) off[i + 1] = off[i] + (int64_t)index->ids[i].size();
    std::vector<uint8_t> fc((size_t)off[nlist_] * subQuantizers_);
    std::vector<int64_t> fi((size_t)off[nlist_]);
    for (int i = 0; i < nlist_; i++) {
      if (index->ids[i].empty()) continue;
      memcpy(&fc[(size_t)off[i] * subQuantizers_], index->codes[i].data(), index->codes[i].size());
      for (size_t j = 0; j < index->ids[i].size(); j++) fi[off[i] + j] = index->ids[i][j];
    }
    VLQ_CHECK(vlq_ivfpq_set_lists(h_, fc.data(), fi.data(), off.data()));
    ntotal = index->ntotal;
  }

  /// gpu/GpuIndexIVFPQ.cu:233-290: overwrite a CPU index with our state
  void copyTo(faiss::IndexIVFPQ* index) const {
    FAISS_THROW_IF_NOT_MSG(ivfpqConfig_.indicesOptions != INDICES_IVF, "Cannot copy to CPU as GPU index doesn't retain indices (INDICES_IVF)");
    IndexFlat* flat = dynamic_cast<IndexFlat*>(index->quantizer);
    FAISS_THROW_IF_NOT_MSG(flat, "target quantizer must be an IndexFlat");
    index->d = d; index->metric_type = metric_type; index->is_trained = is_trained;
    index->nlist = nlist_; index->nprobe = nprobe_; index->ntotal = ntotal;
    index->by_residual = true; index->use_precomputed_table = 0;
    index->pq = faiss::ProductQuantizer(d, subQuantizers_, bitsPerCode_);
    index->code_size = subQuantizers_;
    index->ids.assign(nlist_, std::vector<long>());
    index->codes.assign(nlist_, std::vector<uint8_t>());
    flat->reset();
    if (!is_trained) return;
    flat->add(nlist_, coarse_.data());
    index->pq.centroids = pqCentroids_;
    for (int i = 0; i < nlist_; i++) {
      index->ids[i] = getListIndices(i);
      index->codes[i] = getListCodes(i);
    }
    if (usePrecomputed_) index->precompute_table();
  }

  void reserveMemory(size_t numVecs) {
    reserveMemoryVecs_ = numVecs;
    if (h_) VLQ_CHECK(vlq_ivfpq_reserve_memory(h_, (int64_t)numVecs));
  }
  size_t reclaimMemory() {
    uint64_t bytes = 0;
    if (h_) VLQ_CHECK(vlq_ivfpq_reclaim_memory(h_, &bytes));
    return (size_t)bytes;
  }
  void setPrecomputedCodes(bool enable) {
    usePrecomputed_ = enable;
    VLQ_CHECK(vlq_ivfpq_set_search_options(h_, 1, enable ? 1 : 0, 0));
  }
  bool getPrecomputedCodes() const { return usePrecomputed_; }
  /// what the caller asked for (honoured for 16 x 8-bit codes, plain and VLQ; other shapes compute in fp32)
  bool getFloat16LookupTables() const { return ivfpqConfig_.useFloat16LookupTables; }
  int getNumSubQuantizers() const { return subQuantizers_; }
  int getBitsPerCode() const { return bitsPerCode_; }
  int getCentroidsPerSubQuantizer() const { return 1 << bitsPerCode_; }
  int getNumLists() const { return nlist_; }
  /// gpu/GpuIndexIVF.cu:201-207
  void setNumProbes(int nprobe) {
    FAISS_THROW_IF_NOT_MSG(nprobe > 0 && nprobe <= VLQ_MAX_NPROBE, "nprobe must be in 1..1024");
    nprobe_ = nprobe;
  }
  int getNumProbes() const { return nprobe_; }

  void reset() override {
    if (line_) {       // VLQ index: the lists are the lines
      std::vector<int64_t> off((size_t)nlist_ * numedge_ + 1, 0);
      VLQ_CHECK(vlq_line_set_lists(line_, nullptr, nullptr, nullptr, off.data()));
    } else if (h_) {
      std::vector<int64_t> off(nlist_ + 1, 0);
      VLQ_CHECK(vlq_ivfpq_set_lists(h_, nullptr, nullptr, off.data()));
    }
    ntotal = 0;
  }

  /// k-means of the coarse quantizer + PQ on residuals, on the device
  /// (GpuIndexIVFPQ::train gpu/GpuIndexIVFPQ.cu:1160-1178 minus the fork's graph build)
  void train(Index::idx_t n, const float* x) override {
    if (is_trained) return;
    if (line_) { trainVLQ_(n, x); return; }
    faiss::IndexFlatL2 flat(d);
    faiss::IndexIVFPQ cpu(&flat, d, nlist_, subQuantizers_, bitsPerCode_);
#ifndef VLQ_WITH_REFERENCE_FAISS
    flat.device = device_;   // our CPU-named classes train on the device too
    cpu.device = device_;
#endif
    cpu.verbose = verbose;
    cpu.train(n, x);
    cpu.nprobe = nprobe_;
    copyFrom(&cpu);
  }
  void add(Index::idx_t n, const float* x) override { add_with_ids(n, x, nullptr); }
  void add_with_ids(Index::idx_t n, const float* x, const Index::idx_t* ids) override {
    FAISS_THROW_IF_NOT_MSG(is_trained, "Index not trained");
    if (line_) VLQ_CHECK(vlq_line_add(line_, n, x, (const int64_t*)ids));
    else VLQ_CHECK(vlq_ivfpq_add(h_, n, x, (const int64_t*)ids));
    ntotal += n;
  }
  void search(Index::idx_t n, const float* x, Index::idx_t k, float* distances, Index::idx_t* labels) const override {
    FAISS_THROW_IF_NOT_MSG(is_trained, "Index not trained");
    FAISS_THROW_IF_NOT_MSG(k >= 1 && k <= VLQ_MAX_K, "k outside 1..1024 (gpu/impl/IVFPQ.cu:966-967)");
    if (line_) {   // searchImpl_ -> IVFPQ::queryGraph (gpu/GpuIndexIVFPQ.cu:1400-1460)
      VLQ_CHECK(vlq_line_search(line_, n, x, nprobe_, w1_, (int)k, distances, (int64_t*)labels, nullptr));
      return;
    }
    if (ivfpqConfig_.indicesOptions == INDICES_IVF) {
      std::vector<int64_t> keys((size_t)n * nprobe_);
      std::vector<float> cd((size_t)n * nprobe_);
      VLQ_CHECK(vlq_ivfpq_coarse_search(h_, n, x, nprobe_, cd.data(), keys.data()));
      VLQ_CHECK(vlq_ivfpq_search_preassigned(h_, n, x, keys.data(), cd.data(), nprobe_, (int)k, distances,
                                             (int64_t*)labels, 1));
    } else {
      int rc = vlq_ivfpq_search(h_, n, x, nprobe_, (int)k, distances, (int64_t*)labels);
      if (rc == VLQ_ERR_UNSUPPORTED && fp16TablesOn_) {
        // useFloat16LookupTables is a speed hint of the reference (its drivers pass co.useFloat16 unconditionally): data whose
        // table entries leave the half range (byte-valued descriptors: term 2 reaches 1e5) are searched with fp32 tables
        // instead of failing the caller's search()
        fprintf(stderr, "WARNING GpuIndexIVFPQ: float16 look-up tables cannot hold this index's table entries (%s); "
                        "using fp32 tables\n", vlq_last_error());
        VLQ_CHECK(vlq_ivfpq_set_float16_tables(h_, 0));
        fp16TablesOn_ = false;
        rc = vlq_ivfpq_search(h_, n, x, nprobe_, (int)k, distances, (int64_t*)labels);
      }
      if (rc != VLQ_OK) { FAISS_THROW_FMT("vlq_ivfpq_search -> %d: %s", rc, vlq_last_error()); }
    }
  }

  /// gpu/GpuIndexIVFPQ.h:120-131; a VLQ index's lists are its lines (listId = centroid * nedge + edge, as in the fork)
  int getListLength(int listId) const {
    int64_t len = 0;
    if (line_) VLQ_CHECK(vlq_line_list_length(line_, listId, &len));
    else VLQ_CHECK(vlq_ivfpq_list_length(h_, listId, &len));
    return (int)len;
  }
  std::vector<unsigned char> getListCodes(int listId) const {
    std::vector<unsigned char> c((size_t)getListLength(listId) * subQuantizers_);
    if (c.empty()) return c;
    if (line_) VLQ_CHECK(vlq_line_get_list(line_, listId, c.data(), nullptr, nullptr));
    else VLQ_CHECK(vlq_ivfpq_get_list(h_, listId, c.data(), nullptr));
    return c;
  }
  std::vector<long> getListIndices(int listId) const {
    std::vector<int64_t> v((size_t)getListLength(listId));
    if (v.empty()) return std::vector<long>();
    if (line_) VLQ_CHECK(vlq_line_get_list(line_, listId, nullptr, nullptr, v.data()));
    else VLQ_CHECK(vlq_ivfpq_get_list(h_, listId, nullptr, v.data()));
    return std::vector<long>(v.begin(), v.end());
  }
  vlq_ivfpq_t handle() const { return h_; }

  /// GpuIndexIVFPQ::merge (gpu/GpuIndexIVFPQ.cu:1519-1591): nns / dist hold the per-process
  /// results [nprocess][nq][k] (as gathered by the fork's MPI drivers); host buffers
  void merge(faiss::Index::idx_t* nns, float* dist, int k, int nq, int nprocess, float* distances,
             faiss::Index::idx_t* labels) const {
    const size_t np = (size_t)nprocess * nq * k, no = (size_t)nq * k;
    void *dD = nullptr, *dI = nullptr, *oD = nullptr, *oI = nullptr;
    auto chk = [](hipError_t e) { if (e != hipSuccess) FAISS_THROW_FMT("HIP error %s", hipGetErrorString(e)); };
    chk(hipSetDevice(device_));
    chk(hipMalloc(&dD, np * 4)); chk(hipMalloc(&dI, np * 8)); chk(hipMalloc(&oD, no * 4)); chk(hipMalloc(&oI, no * 8));
    hipStream_t st = resources_->getDefaultStream(device_);
    chk(hipMemcpyAsync(dD, dist, np * 4, hipMemcpyHostToDevice, st));
    chk(hipMemcpyAsync(dI, nns, np * 8, hipMemcpyHostToDevice, st));
    const int rc = vlq_merge_topk(device_, (void*)st, nq, k, nprocess, (const float*)dD, (const int64_t*)dI,
                                  (float*)oD, (int64_t*)oI);
    if (rc == VLQ_OK) {
      chk(hipMemcpyAsync(distances, oD, no * 4, hipMemcpyDeviceToHost, st));
      chk(hipMemcpyAsync(labels, oI, no * 8, hipMemcpyDeviceToHost, st));
      chk(hipStreamSynchronize(st));
    }
    (void)hipFree(dD); (void)hipFree(dI); (void)hipFree(oD); (void)hipFree(oI);
    VLQ_CHECK(rc);
  }

  // ---- the fork's raw VLQ files (gpu/GpuIndexIVFPQ.cu:1731-1844) -------------------------
  /// <name>.ppqt = centroids | pq centroids | edgeInfo | edgeDistInfo | lambdaInfo | constInfo
  void writeCodebookToFile(const std::string& name) {
    FAISS_THROW_IF_NOT_MSG(line_ && is_trained, "VLQ index not trained");
    std::ofstream f((name + ".ppqt").c_str(), std::ofstream::out | std::ofstream::binary);
    FAISS_THROW_IF_NOT_MSG(f.good(), "cannot open .ppqt for writing");
    std::vector<float> zeros(nLambda_, 0.f);   // constInfo_: the abandoned "constq" experiment (:722-731)
    f.write((const char*)coarse_.data(), coarse_.size() * sizeof(float));
    f.write((const char*)pqCentroids_.data(), pqCentroids_.size() * sizeof(float));
    f.write((const char*)edgeInfo_, (size_t)nlist_ * numedge_ * sizeof(int));
    f.write((const char*)edgeDistInfo_, (size_t)nlist_ * numedge_ * sizeof(float));
    f.write((const char*)lambdaInfo_, nLambda_ * sizeof(float));
    f.write((const char*)zeros.data(), nLambda_ * sizeof(float));
  }
  void readCodebookFromFile(const std::string& name) {
    FAISS_THROW_IF_NOT_MSG(line_, "not a VLQ index");
    std::ifstream f((name + ".ppqt").c_str(), std::ifstream::in | std::ifstream::binary);
    FAISS_THROW_IF_NOT_MSG(f.good(), "cannot open .ppqt");
    coarse_.resize((size_t)nlist_ * d);
    pqCentroids_.resize((size_t)(1 << bitsPerCode_) * d);
    f.read((char*)coarse_.data(), coarse_.size() * sizeof(float));
    f.read((char*)pqCentroids_.data(), pqCentroids_.size() * sizeof(float));
    f.read((char*)edgeInfo_, (size_t)nlist_ * numedge_ * sizeof(int));
    f.read((char*)edgeDistInfo_, (size_t)nlist_ * numedge_ * sizeof(float));
    f.read((char*)lambdaInfo_, nLambda_ * sizeof(float));
    FAISS_THROW_IF_NOT_MSG(f.good(), ".ppqt file too short");
    VLQ_CHECK(vlq_line_set_coarse_centroids(line_, coarse_.data()));
    VLQ_CHECK(vlq_line_set_graph(line_, edgeInfo_, edgeDistInfo_));
    VLQ_CHECK(vlq_line_set_lambda_codebook(line_, lambdaInfo_));
    VLQ_CHECK(vlq_line_set_pq_centroids(line_, pqCentroids_.data()));
    is_trained = true;
  }
  /// <name>.dbIdx (ids) / .dblas (lambda bytes) / .dbcodes / .dbcount (int per line), line-major
  void writeDbToFile(const std::string& name) {
    FAISS_THROW_IF_NOT_MSG(line_, "not a VLQ index");
    std::ofstream fi((name + ".dbIdx").c_str(), std::ofstream::binary), fl((name + ".dblas").c_str(), std::ofstream::binary),
        fc((name + ".dbcodes").c_str(), std::ofstream::binary), fn((name + ".dbcount").c_str(), std::ofstream::binary);
    FAISS_THROW_IF_NOT_MSG(fi.good() && fl.good() && fc.good() && fn.good(), "cannot open db files for writing");
    const int64_t nl = (int64_t)nlist_ * numedge_;
    std::vector<int> counts(nl);
    std::vector<uint8_t> c, l;
    std::vector<int64_t> ids;
    for (int64_t i = 0; i < nl; i++) {
      int64_t len = 0;
      VLQ_CHECK(vlq_line_list_length(line_, i, &len));
      counts[i] = (int)len;
      if (len == 0) continue;
      c.resize((size_t)len * subQuantizers_); l.resize(len); ids.resize(len);
      VLQ_CHECK(vlq_line_get_list(line_, i, c.data(), l.data(), ids.data()));
      fi.write((const char*)ids.data(), len * sizeof(long));
      fl.write((const char*)l.data(), len);
      fc.write((const char*)c.data(), c.size());
    }
    fn.write((const char*)counts.data(), counts.size() * sizeof(int));
  }
  /// readDbFromFile (gpu/GpuIndexIVFPQ.cu:1847-1904): every line of the .dbIdx / .dblas / .dbcodes / .dbcount set
  void readDbFromFile(const std::string& name) { readDb_(name, 1, 0); }
  /// :1918-2010: the same with the caller's total count (the reference sizes its device buffers with it and then walks
  /// the whole .dbcount table): nb must cover the stored vectors
  void readDbFromFile(const std::string& name, size_t nb) {
    readDb_(name, 1, 0);
    FAISS_THROW_IF_NOT_FMT((size_t)ntotal <= nb, "readDbFromFile: nb=%zu but the files hold %ld vectors", nb, ntotal);
  }
  /// :2214-2309 / :2106-2212 -- the per-rank loaders of the fork's MPI drivers (gpu/test/deep1b16_query.cpp:270,
  /// sift1b16_query.cpp:323: `index.readDbFromFile(prename, 0, numproces, rank)`): rank r of pronum keeps the lines
  /// [(nl / pronum) * r, (nl / pronum) * (r + 1)) with nl = nlist * nedge -- the reference's own arithmetic (:2132-2141),
  /// remainder lines of a non-dividing pronum are dropped there too -- every other line stays empty; begin_ / end_ =
  /// the centroid range (:2127-2134).  nb is ignored on input, as in the reference (it overwrites it with the count)
  void readDbFromFile(const std::string& name, int pronum, int rank) { readDb_(name, pronum, rank); }
  void readDbFromFile(const std::string& name, size_t /*nb*/, int pronum, int rank) { readDb_(name, pronum, rank); }

  /// lambda bytes of one line (gpu/GpuIndexIVFPQ.cu:2332-2338)
  std::vector<unsigned char> getListLambdas(int listId) const {
    FAISS_THROW_IF_NOT_MSG(line_, "not a VLQ index");
    int64_t len = 0;
    VLQ_CHECK(vlq_line_list_length(line_, listId, &len));
    std::vector<unsigned char> l((size_t)len);
    if (len > 0) VLQ_CHECK(vlq_line_get_list(line_, listId, nullptr, l.data(), nullptr));
    return l;
  }

  /// coarse centroids as a .umem file (gpu/GpuIndexIVFPQ.cu:1760-1771 -> filehelper.cpp:253-280, 344-351): the text
  /// lines "<num>\n<dim>\n", the float rows from byte 20 (the reference's writer appends the rows a second time behind
  /// them -- filehelper.cpp:277 -- which no reader looks at; they are written once here)
  void writeCentroidsToFile(const std::string& name) {
    FAISS_THROW_IF_NOT_MSG(!coarse_.empty(), "no coarse centroids");
    std::ofstream f((name + ".umem").c_str(), std::ofstream::binary);
    FAISS_THROW_IF_NOT_MSG(f.good(), "cannot open the centroid file for writing");
    f << (size_t)nlist_ << std::endl << (unsigned)d << std::endl;
    f.seekp(20, std::ios::beg);
    f.write((const char*)coarse_.data(), (std::streamsize)((size_t)nlist_ * d * sizeof(float)));
  }

 private:
  void readDb_(const std::string& name, int pronum, int rank) {
    FAISS_THROW_IF_NOT_MSG(line_, "not a VLQ index");
    FAISS_THROW_IF_NOT_FMT(pronum >= 1 && rank >= 0 && rank < pronum, "bad process rank %d of %d", rank, pronum);
    std::ifstream fi((name + ".dbIdx").c_str(), std::ifstream::binary), fl((name + ".dblas").c_str(), std::ifstream::binary),
        fc((name + ".dbcodes").c_str(), std::ifstream::binary), fn((name + ".dbcount").c_str(), std::ifstream::binary);
    FAISS_THROW_IF_NOT_MSG(fi.good() && fl.good() && fc.good() && fn.good(), "cannot open db files");
    const int64_t nl = (int64_t)nlist_ * numedge_;
    std::vector<int> counts(nl);
    fn.read((char*)counts.data(), counts.size() * sizeof(int));
    FAISS_THROW_IF_NOT_MSG(fn.good(), "db count file too short");
    begin_ = (nlist_ / pronum) * rank;
    end_ = rank == pronum - 1 ? nlist_ - 1 : begin_ + nlist_ / pronum - 1;
    const int64_t start = (nl / pronum) * rank, stop = pronum == 1 ? nl : start + nl / pronum;   // lines [start, stop)
    std::vector<int64_t> off(nl + 1, 0);
    int64_t skip = 0, take = 0;
    for (int64_t i = 0; i < nl; i++) {
      const bool mine = i >= start && i < stop;
      off[i + 1] = off[i] + (mine ? counts[i] : 0);
      if (i < start) skip += counts[i];
      if (mine) take += counts[i];
    }
    std::vector<int64_t> ids(take);
    std::vector<uint8_t> l(take), c((size_t)take * subQuantizers_);
    fi.seekg(skip * (int64_t)sizeof(long)); fl.seekg(skip); fc.seekg(skip * subQuantizers_);
    fi.read((char*)ids.data(), take * sizeof(long));
    fl.read((char*)l.data(), take);
    fc.read((char*)c.data(), c.size());
    FAISS_THROW_IF_NOT_MSG(fi.good() && fl.good() && fc.good(), "db files too short");
    VLQ_CHECK(vlq_line_set_lists(line_, c.data(), l.data(), ids.data(), off.data()));
    ntotal = take;
  }

  /// GpuIndexIVFPQ::train of the fork (gpu/GpuIndexIVFPQ.cu:1160-1178): coarse k-means,
  /// centroid graph, then trainResidualQuantizer_ (:346-403): lines + lambdas of a
  /// training subset, 1-D k-means of the lambdas, PQ on the residuals to the anchors
  void trainVLQ_(Index::idx_t n, const float* x) {
    {
      faiss::IndexFlatL2 flat(d);
#ifndef VLQ_WITH_REFERENCE_FAISS
      flat.device = device_;
#endif
      ClusteringParameters cp;
      cp.niter = 10;
      Clustering clus(d, nlist_, cp);
      clus.verbose = verbose;
      clus.train(n, x, flat);
      coarse_ = clus.centroids;
    }
    VLQ_CHECK(vlq_line_set_coarse_centroids(line_, coarse_.data()));
    VLQ_CHECK(vlq_line_build_graph(line_, edgeInfo_, edgeDistInfo_));
    const Index::idx_t nt = std::min<Index::idx_t>(n, (Index::idx_t)(1 << bitsPerCode_) * 128);
    std::vector<int32_t> line(nt);
    std::vector<float> lambdaf(nt);
    VLQ_CHECK(vlq_line_assign(line_, nt, x, line.data(), lambdaf.data()));
    {
      ClusteringParameters cp;
      Clustering clus(1, nLambda_, cp);
      faiss::IndexFlatL2 flat1(1);
#ifndef VLQ_WITH_REFERENCE_FAISS
      flat1.device = device_;
#endif
      clus.train(nt, lambdaf.data(), flat1);
      memcpy(lambdaInfo_, clus.centroids.data(), sizeof(float) * nLambda_);
    }
    VLQ_CHECK(vlq_line_set_lambda_codebook(line_, lambdaInfo_));
    std::vector<float> residuals((size_t)nt * d);
    VLQ_CHECK(vlq_line_residuals(line_, nt, x, residuals.data()));
    faiss::ProductQuantizer pq(d, subQuantizers_, bitsPerCode_);
    pq.verbose = verbose;
    pq.train((int)nt, residuals.data());
    pqCentroids_ = pq.centroids;
    VLQ_CHECK(vlq_line_set_pq_centroids(line_, pqCentroids_.data()));
    is_trained = true;
  }
  // The reference's own drivers ask for float16 look-up tables / coarse storage
  // (gpu/test/deep1b16_query.cpp:239-243: co.useFloat16 = true -> config.useFloat16LookupTables), a
  // speed/memory option of its CUDA kernels (gpu/GpuIndexIVFPQ.h:24-38, impl/IVFPQ.cu:1442 toHalf).
  // They are ACCEPTED: the VLQ search (the drivers' path) builds float16 tables the way the reference
  // does (vlq_line_set_float16_tables), and so does the plain IVFPQ search for 16 x 8-bit codes with precomputed
  // tables (vlq_ivfpq_set_float16_tables; data that leave the half range fall back to fp32 tables with a warning);
  // other shapes and the coarse quantizer compute the same quantities in fp32 -- every result the fp16
  // configuration could return is returned at higher precision, and fp32 is the path pinned bit for bit to the CPU index.  Only options that would
  // change SEMANTICS are rejected.
  void verifyConfig_() const {
    FAISS_THROW_IF_NOT_MSG(ivfpqConfig_.memorySpace == MemorySpace::Device || ivfpqConfig_.memorySpace == MemorySpace::Unified,
                           "unknown memory space");
    FAISS_THROW_IF_NOT_MSG(ivfpqConfig_.indicesOptions >= INDICES_CPU && ivfpqConfig_.indicesOptions <= INDICES_64_BIT,
                           "unknown indicesOptions");
  }
  void create_() {
    VLQ_CHECK(vlq_ivfpq_create(&h_, device_, d, nlist_, subQuantizers_, bitsPerCode_));
    VLQ_CHECK(vlq_ivfpq_set_stream(h_, (void*)resources_->getDefaultStream(device_)));
    usePrecomputed_ = ivfpqConfig_.usePrecomputedTables;
    VLQ_CHECK(vlq_ivfpq_set_search_options(h_, 1, usePrecomputed_ ? 1 : 0, 0));
    // useFloat16LookupTables: half tables for 16 x 8-bit codes with precomputed tables, built as the reference
    // builds them (vlq_ivfpq_set_float16_tables); other shapes compute in fp32 (superset precision)
    if (ivfpqConfig_.useFloat16LookupTables && subQuantizers_ == 16 && bitsPerCode_ == 8) {
      VLQ_CHECK(vlq_ivfpq_set_float16_tables(h_, 1));
      fp16TablesOn_ = true;
    }
  }
  GpuIndexIVFPQConfig ivfpqConfig_;
  int nlist_, nprobe_, subQuantizers_, bitsPerCode_;
  size_t reserveMemoryVecs_;
  bool usePrecomputed_ = false;
  mutable bool fp16TablesOn_ = false;   ///< half tables in use for the plain IVFPQ search (dropped if the data do not fit the half range)
  std::vector<float> coarse_, pqCentroids_;
  std::vector<int> edgeInfoV_;
  std::vector<float> edgeDistInfoV_, lambdaInfoV_;
  vlq_ivfpq_t h_ = nullptr;
  vlq_line_t line_ = nullptr;
};

} }
