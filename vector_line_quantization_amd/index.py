"""Python handle over the C ABI.  Arguments may be numpy arrays (host) or torch
tensors (host or device: only .data_ptr() is taken -- torch is plumbing for device
memory and streams, never compute).  The library issues its work on the index's own stream:
hand over device tensors only after the stream that produced them has finished, or make the
index use that stream (`set_stream(torch.cuda.current_stream().cuda_stream)`)."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def _is_torch(a):
    return hasattr(a, "data_ptr") and hasattr(a, "is_contiguous")


def _ptr(a, np_dtype=None):
    """(pointer, keepalive) of a numpy array or torch tensor; None -> NULL."""
    if a is None:
        return None, None
    if _is_torch(a):
        if not a.is_contiguous():
            a = a.contiguous()
        return C.c_void_p(a.data_ptr()), a
    a = np.ascontiguousarray(a, dtype=np_dtype)
    return a.ctypes.data_as(C.c_void_p), a


def _out_ptr(a, np_dtype):
    """Pointer of an OUTPUT buffer: it must be written in place, so a buffer that would need a
    contiguous (or dtype-converted) copy is an error, never a silent temporary."""
    if _is_torch(a):
        if not a.is_contiguous():
            raise ValueError("output tensor must be contiguous (the library writes it in place)")
        want = {np.float32: "torch.float32", np.int64: "torch.int64"}.get(np_dtype)
        if want is not None and str(a.dtype) != want:
            raise ValueError("output tensor must be %s, got %s" % (want, a.dtype))
        return C.c_void_p(a.data_ptr()), a
    if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"] or a.dtype != np.dtype(np_dtype) or not a.flags["WRITEABLE"]:
        raise ValueError("output array must be a writeable C-contiguous numpy array of dtype %s" % np.dtype(np_dtype).name)
    return a.ctypes.data_as(C.c_void_p), a


class GpuIVFPQ:
    """Mirror of the data-carrying surface of faiss::gpu::GpuIndexIVFPQ
    (gpu/GpuIndexIVFPQ.h:41-234) / faiss::IndexIVFPQ (IndexIVFPQ.h:29-164) for the
    search path: construct, copy trained state in, search."""

    def __init__(self, d, nlist, M, nbits, device=0):
        self.d, self.nlist, self.M, self.nbits = d, nlist, M, nbits
        self.ksub = 1 << nbits
        self._h = C.c_void_p()
        check(lib().vlq_ivfpq_create(C.byref(self._h), C.c_int(device), C.c_int(d), C.c_int(nlist),
                                     C.c_int(M), C.c_int(nbits)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().vlq_ivfpq_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state ------------------------------------------------------------
    def set_stream(self, stream_ptr):
        check(lib().vlq_ivfpq_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def set_coarse_centroids(self, c):
        p, _k = _ptr(c, np.float32)
        check(lib().vlq_ivfpq_set_coarse_centroids(self._h, p))

    def set_imi_centroids(self, imi_nbits, c):
        p, _k = _ptr(c, np.float32)
        check(lib().vlq_ivfpq_set_imi_centroids(self._h, C.c_int(imi_nbits), p))

    def set_pq_centroids(self, c):
        p, _k = _ptr(c, np.float32)
        check(lib().vlq_ivfpq_set_pq_centroids(self._h, p))

    def set_search_options(self, by_residual=True, use_precomputed_table=1, max_codes=0):
        check(lib().vlq_ivfpq_set_search_options(self._h, C.c_int(int(by_residual)),
                                                 C.c_int(use_precomputed_table), C.c_int64(max_codes)))

    def set_float16_tables(self, enable=True):
        """GpuIndexIVFPQConfig::useFloat16LookupTables for the plain IVFPQ search (include/vlq_ivfpq.h)"""
        check(lib().vlq_ivfpq_set_float16_tables(self._h, C.c_int(int(enable))))

    def set_coarse_screen(self, mode):
        """0 off, 1 on (default): float16 screen in front of the exact coarse distances (speed only; include/vlq_ivfpq.h)"""
        check(lib().vlq_ivfpq_set_coarse_screen(self._h, C.c_int(int(mode))))

    def coarse_screen_state(self):
        """(enabled, rows screened so far, rows the screen handed to the exact path)"""
        en, rows, und = C.c_int(0), C.c_uint64(0), C.c_uint32(0)
        check(lib().vlq_ivfpq_coarse_screen_state(self._h, C.byref(en), C.byref(rows), C.byref(und)))
        return bool(en.value), int(rows.value), int(und.value)

    def set_scan_schedule(self, mode):
        """0 automatic, 1 query-major, 2 list-owned (speed only; include/vlq_ivfpq.h)"""
        check(lib().vlq_ivfpq_set_scan_schedule(self._h, C.c_int(int(mode))))

    def set_lists(self, codes, ids, list_offsets):
        pc, _a = _ptr(codes, np.uint8)
        pi, _b = _ptr(ids, np.int64)
        po, _c = _ptr(list_offsets, np.int64)
        check(lib().vlq_ivfpq_set_lists(self._h, pc, pi, po))

    @property
    def ntotal(self):
        return int(lib().vlq_ivfpq_ntotal(self._h))

    def list_length(self, i):
        n = C.c_int64()
        check(lib().vlq_ivfpq_list_length(self._h, C.c_int(i), C.byref(n)))
        return n.value

    def get_list(self, i):
        n = self.list_length(i)
        codes = np.empty((n, self.M), np.uint8)
        ids = np.empty((n,), np.int64)
        check(lib().vlq_ivfpq_get_list(self._h, C.c_int(i), codes.ctypes.data_as(C.c_void_p),
                                       ids.ctypes.data_as(C.c_void_p)))
        return codes, ids

    # --- add --------------------------------------------------------------
    def add(self, x, xids=None):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        pi, _b = _ptr(xids, np.int64)
        check(lib().vlq_ivfpq_add(self._h, C.c_int64(n), px, pi))

    def reserve_memory(self, num_vecs):
        """GpuIndexIVFPQ::reserveMemory: room for num_vecs / nlist vectors in every list."""
        check(lib().vlq_ivfpq_reserve_memory(self._h, C.c_int64(num_vecs)))

    def reclaim_memory(self):
        """GpuIndexIVFPQ::reclaimMemory: drop the append slack; returns the device bytes freed."""
        n = C.c_uint64()
        check(lib().vlq_ivfpq_reclaim_memory(self._h, C.byref(n)))
        return n.value

    def encode(self, x):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        assign = np.empty((n,), np.int64)
        codes = np.empty((n, self.M), np.uint8)
        check(lib().vlq_ivfpq_encode(self._h, C.c_int64(n), px, assign.ctypes.data_as(C.c_void_p),
                                     codes.ctypes.data_as(C.c_void_p)))
        return assign, codes

    # --- search -----------------------------------------------------------
    def encode_preassigned(self, x, assign):
        """codes of x for the given lists (IndexIVFPQ::encode_multiple, compute_keys = false)"""
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        pa, _b = _ptr(assign, np.int64)
        codes = np.empty((n, self.M), np.uint8)
        check(lib().vlq_ivfpq_encode_preassigned(self._h, C.c_int64(n), px, pa, codes.ctypes.data_as(C.c_void_p)))
        return codes

    def _out(self, out, shape, dtype, like):
        if out is not None:
            return out
        if _is_torch(like) and like.is_cuda:
            import torch
            return torch.empty(shape, dtype={np.float32: torch.float32, np.int64: torch.int64}[dtype],
                               device=like.device)
        return np.empty(shape, dtype)

    def search(self, x, nprobe, k, D=None, I=None):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        D = self._out(D, (n, k), np.float32, x)
        I = self._out(I, (n, k), np.int64, x)
        pD, _b = _out_ptr(D, np.float32)
        pI, _c = _out_ptr(I, np.int64)
        check(lib().vlq_ivfpq_search(self._h, C.c_int64(n), px, C.c_int(nprobe), C.c_int(k), pD, pI))
        return D, I

    def search_preassigned(self, x, keys, coarse_dis, k, store_pairs=False, D=None, I=None):
        n = x.shape[0]
        nprobe = keys.shape[1]
        px, _a = _ptr(x, np.float32)
        pk, _b = _ptr(keys, np.int64)
        pc, _c = _ptr(coarse_dis, np.float32)
        D = self._out(D, (n, k), np.float32, x)
        I = self._out(I, (n, k), np.int64, x)
        pD, _d = _out_ptr(D, np.float32)
        pI, _e = _out_ptr(I, np.int64)
        check(lib().vlq_ivfpq_search_preassigned(self._h, C.c_int64(n), px, pk, pc, C.c_int(nprobe),
                                                 C.c_int(k), pD, pI, C.c_int(int(store_pairs))))
        return D, I

    def coarse_search(self, x, nprobe, cdis=None, keys=None):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        cdis = self._out(cdis, (n, nprobe), np.float32, x)
        keys = self._out(keys, (n, nprobe), np.int64, x)
        pc, _b = _out_ptr(cdis, np.float32)
        pk, _c = _out_ptr(keys, np.int64)
        check(lib().vlq_ivfpq_coarse_search(self._h, C.c_int64(n), px, C.c_int(nprobe), pc, pk))
        return cdis, keys

    # --- introspection ----------------------------------------------------
    def query_tables(self, x, inner_product=True):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        out = np.empty((n, self.M, self.ksub), np.float32)
        check(lib().vlq_ivfpq_query_tables(self._h, C.c_int64(n), px, C.c_int(int(inner_product)),
                                           out.ctypes.data_as(C.c_void_p)))
        return out

    def precomputed_table(self, rows=None):
        out = np.empty((rows or self.nlist, self.M, self.ksub), np.float32)
        check(lib().vlq_ivfpq_get_precomputed_table(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self, reset=False):
        nq, ncode = C.c_uint64(), C.c_uint64()
        check(lib().vlq_ivfpq_stats(self._h, C.byref(nq), C.byref(ncode), C.c_int(int(reset))))
        return nq.value, ncode.value

    def last_scan_info(self):
        """what the last search's list scan was (kernel shape, walking order, clock period): include/vlq_ivfpq.h"""
        buf = C.create_string_buffer(256)
        check(lib().vlq_ivfpq_last_scan_info(self._h, buf, C.c_int(256)))
        return buf.value.decode()

    def reset_walk_state(self):
        check(lib().vlq_ivfpq_reset_walk_state(self._h))

    def profile(self, enable=True):
        """0 / False: off; 1 / True: every stage; 2: the scan kernel only (two event records per call);
        3: the scan kernel of every 4th call, starting with the next one"""
        check(lib().vlq_ivfpq_profile(self._h, C.c_int(int(enable))))

    def profile_read(self, reset=True):
        ms = (C.c_double * 3)()
        calls = C.c_int64()
        check(lib().vlq_ivfpq_profile_read(self._h, ms, C.byref(calls), C.c_int(int(reset))))
        return {"coarse_ms": ms[0], "tables_ms": ms[1], "scan_ms": ms[2], "scan_calls": calls.value}


class GpuVLQ:
    """The fork's vector-and-line-quantization index (include/vlq_line.h):
    GpuIndexIVFPQ(resources, dims, nlist, M, nbits, nedge, nLambda, ...) of
    gpu/GpuIndexIVFPQ.h:60-68 -- construct, load trained state, add, search."""

    def __init__(self, d, nlist, M, nbits, nedge, nlambda, device=0):
        self.d, self.nlist, self.M, self.nbits = d, nlist, M, nbits
        self.nedge, self.nlambda, self.ksub = nedge, nlambda, 1 << nbits
        self._h = C.c_void_p()
        check(lib().vlq_line_create(C.byref(self._h), C.c_int(device), C.c_int(d), C.c_int(nlist),
                                    C.c_int(M), C.c_int(nbits), C.c_int(nedge), C.c_int(nlambda)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().vlq_line_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(lib().vlq_line_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def set_coarse_centroids(self, c):
        p, _k = _ptr(c, np.float32)
        check(lib().vlq_line_set_coarse_centroids(self._h, p))

    def set_pq_centroids(self, c):
        p, _k = _ptr(c, np.float32)
        check(lib().vlq_line_set_pq_centroids(self._h, p))

    def set_lambda_codebook(self, li):
        p, _k = _ptr(li, np.float32)
        check(lib().vlq_line_set_lambda_codebook(self._h, p))

    def set_float16_tables(self, enable=True):
        """GpuIndexIVFPQConfig::useFloat16LookupTables for the VLQ search (include/vlq_line.h)"""
        check(lib().vlq_line_set_float16_tables(self._h, C.c_int(int(enable))))

    def set_row_mode(self, mode):
        """0 automatic, 1 term-2 rows read from the stored table, 2 rows rebuilt in the scan kernel
        (speed only, identical results; include/vlq_line.h)"""
        check(lib().vlq_line_set_row_mode(self._h, C.c_int(int(mode))))

    def set_scan_parts(self, parts):
        """workgroups per query of the default 16-byte scan (0 = automatic); never changes a result"""
        check(lib().vlq_line_set_scan_parts(self._h, C.c_int(int(parts))))

    def set_graph(self, edge_info, edge_dist):
        pe, _a = _ptr(edge_info, np.int32)
        pd, _b = _ptr(edge_dist, np.float32)
        check(lib().vlq_line_set_graph(self._h, pe, pd))

    def build_graph(self):
        ei = np.empty((self.nlist, self.nedge), np.int32)
        ed = np.empty((self.nlist, self.nedge), np.float32)
        check(lib().vlq_line_build_graph(self._h, ei.ctypes.data_as(C.c_void_p), ed.ctypes.data_as(C.c_void_p)))
        return ei, ed

    def assign(self, x):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        line = np.empty((n,), np.int32)
        lam = np.empty((n,), np.float32)
        check(lib().vlq_line_assign(self._h, C.c_int64(n), px, line.ctypes.data_as(C.c_void_p),
                                    lam.ctypes.data_as(C.c_void_p)))
        return line, lam

    def residuals(self, x):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        out = np.empty((n, self.d), np.float32)
        check(lib().vlq_line_residuals(self._h, C.c_int64(n), px, out.ctypes.data_as(C.c_void_p)))
        return out

    def encode(self, x):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        line = np.empty((n,), np.int32)
        lam = np.empty((n,), np.uint8)
        codes = np.empty((n, self.M), np.uint8)
        check(lib().vlq_line_encode(self._h, C.c_int64(n), px, line.ctypes.data_as(C.c_void_p),
                                    lam.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p)))
        return line, lam, codes

    def add(self, x, xids=None):
        px, _a = _ptr(x, np.float32)
        pi, _b = _ptr(xids, np.int64)
        check(lib().vlq_line_add(self._h, C.c_int64(x.shape[0]), px, pi))

    def set_lists(self, codes, lambdas, ids, line_offsets):
        pc, _a = _ptr(codes, np.uint8)
        pl, _b = _ptr(lambdas, np.uint8)
        pi, _c = _ptr(ids, np.int64)
        po, _d = _ptr(line_offsets, np.int64)
        check(lib().vlq_line_set_lists(self._h, pc, pl, pi, po))

    @property
    def ntotal(self):
        return int(lib().vlq_line_ntotal(self._h))

    def get_list(self, line):
        n = C.c_int64()
        check(lib().vlq_line_list_length(self._h, C.c_int64(line), C.byref(n)))
        codes = np.empty((n.value, self.M), np.uint8)
        lam = np.empty((n.value,), np.uint8)
        ids = np.empty((n.value,), np.int64)
        check(lib().vlq_line_get_list(self._h, C.c_int64(line), codes.ctypes.data_as(C.c_void_p),
                                      lam.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p)))
        return codes, lam, ids

    def search(self, x, nprobe, w1, k, D=None, I=None, return_lines=False):
        n = x.shape[0]
        px, _a = _ptr(x, np.float32)
        if D is None:
            D = GpuIVFPQ._out(self, None, (n, k), np.float32, x)
        if I is None:
            I = GpuIVFPQ._out(self, None, (n, k), np.int64, x)
        pD, _b = _out_ptr(D, np.float32)
        pI, _c = _out_ptr(I, np.int64)
        lines = np.empty((n, w1), np.int32) if return_lines else None
        pl = lines.ctypes.data_as(C.c_void_p) if return_lines else None
        check(lib().vlq_line_search(self._h, C.c_int64(n), px, C.c_int(nprobe), C.c_int(w1), C.c_int(k),
                                    pD, pI, pl))
        return (D, I, lines) if return_lines else (D, I)

    def stats(self, reset=False):
        nc = C.c_uint64()
        check(lib().vlq_line_stats(self._h, C.byref(nc), C.c_int(int(reset))))
        return nc.value

    def profile(self, enable=True):
        check(lib().vlq_line_profile(self._h, C.c_int(int(enable))))

    def profile_read(self, reset=True):
        """(scan-kernel milliseconds summed over the launches since the last reset, number of launches)"""
        ms, n = C.c_double(), C.c_int64()
        check(lib().vlq_line_profile_read(self._h, C.byref(ms), C.byref(n), C.c_int(int(reset))))
        return ms.value, n.value
