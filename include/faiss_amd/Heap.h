// Result container of search_knn_with_key: the reference passes caller buffers as a
// HeapArray (Heap.h:349-398).  Only the plain-data view is needed here: selection
// runs on the device (wave64 select), results come back sorted ascending and padded
// with -1 / FLT_MAX exactly as heap_reorder leaves them (Heap.h:296-323).
#pragma once
#include <cstddef>

namespace faiss {

template <typename T_, typename TI_>
struct CMax { typedef T_ T; typedef TI_ TI; };
template <typename T_, typename TI_>
struct CMin { typedef T_ T; typedef TI_ TI; };

template <typename C>
struct HeapArray {
  typedef typename C::TI TI;
  typedef typename C::T T;
  size_t nh;   ///< number of heaps (queries)
  size_t k;    ///< entries per heap
  TI* ids;     ///< nh * k
  T* val;      ///< nh * k
  T* get_val(size_t key) { return val + key * k; }
  TI* get_ids(size_t key) { return ids + key * k; }
};

typedef HeapArray<CMax<float, long> > float_maxheap_array_t;
typedef HeapArray<CMin<float, long> > float_minheap_array_t;

}  // namespace faiss
