"""Checks for indexes too large for a whole-index oracle (the 1 B-vector runs of tools/time_imi.py,
tools/time_vlq.py, tools/big_index.py): a SAMPLE of queries is verified end to end against the oracle
on exactly the lists / lines those queries touch, fetched back from the device.

Test infrastructure: this module lives under tests/ because it drives the oracle; the tools import it
only to VERIFY what they time (never inside a timed region)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def check_ivfpq_sample(g, xs, nprobe, k, pq, coarse=None, imi=None, imi_nbits=0):
    """g: GpuIVFPQ holding the big index; xs [ns, d] float32 host queries.  Returns a dict of booleans
    / counts: coarse stage (keys + coarse_dis) and full search (D, I, ncode) bit-exact vs the oracle."""
    from oracle.pyoracle import OracleIndex
    xs = np.ascontiguousarray(xs, np.float32)
    d, nlist, M = g.d, g.nlist, g.M
    cd, keys = g.coarse_search(xs, nprobe)
    ox0 = OracleIndex(d, nlist, M, 8, coarse, pq, imi_centroids=imi, imi_nbits=imi_nbits)
    cdo, keyso = ox0.coarse_search(xs, nprobe, canonical=True)
    out = {"queries": int(xs.shape[0]), "coarse_keys_equal": bool(np.array_equal(keys, keyso)),
           "coarse_dis_bits_equal": bool(np.array_equal(_bits(cd), _bits(cdo)))}
    # the probed lists, fetched from the device, as a sparse copy of the index
    uniq = np.unique(keys[keys >= 0])
    off = np.zeros(nlist + 1, np.int64)
    codes, ids, lens = [], [], np.zeros(uniq.shape[0], np.int64)
    for j, key in enumerate(uniq):
        c, i = g.get_list(int(key))
        codes.append(c)
        ids.append(i)
        lens[j] = i.shape[0]
    cnt = np.zeros(nlist, np.int64)
    cnt[uniq] = lens
    np.cumsum(cnt, out=off[1:])
    del cnt
    ox = OracleIndex(d, nlist, M, 8, coarse, pq, imi_centroids=imi, imi_nbits=imi_nbits,
                     codes=np.concatenate(codes) if codes else None, ids=np.concatenate(ids) if ids else None,
                     list_offsets=off, precomputed_table=ox0.precomputed_table)
    Do, Io = ox.search_preassigned(xs, keys, cd, k, canonical=True)
    g.stats(reset=True)
    D, I = g.search(xs, nprobe, k)
    _n, ncode = g.stats(reset=True)
    out.update({"lists_fetched": int(uniq.shape[0]), "codes_in_fetched_lists": int(lens.sum()),
                "distance_bits_equal": bool(np.array_equal(_bits(D), _bits(Do))),
                "labels_equal": bool(np.array_equal(I, Io)), "ncode_equal": bool(ncode == ox.last_ncode),
                "ncode_per_query": ncode / float(xs.shape[0])})
    out["ok"] = all(out[f] for f in ("coarse_keys_equal", "coarse_dis_bits_equal", "distance_bits_equal",
                                     "labels_equal", "ncode_equal"))
    return out


def check_vlq_sample(g, xs, nprobe, w1, k, coarse, pq, lambda_info, edge_info, edge_dist, fp16=False):
    """g: GpuVLQ holding the big index.  Lines selected for the sample queries are fetched from the
    device; the VLQ oracle then searches the same queries on that sparse copy."""
    from oracle.pyoracle import OracleVLQ
    xs = np.ascontiguousarray(xs, np.float32)
    g.stats(reset=True)
    D, I, lines = g.search(xs, nprobe, w1, k, return_lines=True)
    ncode = g.stats(reset=True)
    nl = g.nlist * g.nedge
    uniq = np.unique(lines[lines >= 0])
    off = np.zeros(nl + 1, np.int64)
    cnt = np.zeros(nl, np.int64)
    codes, lams, ids = [], [], []
    for line in uniq:
        c, l, i = g.get_list(int(line))
        codes.append(c)
        lams.append(l)
        ids.append(i)
        cnt[line] = i.shape[0]
    np.cumsum(cnt, out=off[1:])
    v = OracleVLQ(g.d, g.nlist, g.M, g.nbits, g.nedge, g.nlambda, coarse, pq_centroids=pq, edge_info=edge_info,
                  edge_dist=edge_dist, lambda_info=lambda_info)
    v.codes = np.ascontiguousarray(np.concatenate(codes)) if codes else v.codes
    v.lambdas = np.ascontiguousarray(np.concatenate(lams)) if lams else v.lambdas
    v.ids = np.ascontiguousarray(np.concatenate(ids)) if ids else v.ids
    v.line_off = off
    Do, Io, lo = v.search(xs, nprobe, w1, k, return_lines=True, fp16=fp16)
    # the oracle sees only the fetched lines: every line it selects must be one the device selected too
    out = {"queries": int(xs.shape[0]), "lines_fetched": int(uniq.shape[0]), "codes_in_fetched_lines": int(cnt.sum()),
           "lines_equal": bool(np.array_equal(lines, lo)), "distance_bits_equal": bool(np.array_equal(_bits(D), _bits(Do))),
           "labels_equal": bool(np.array_equal(I, Io)), "ncode_equal": bool(ncode == v.last_ncode),
           "ncode_per_query": ncode / float(xs.shape[0])}
    out["ok"] = all(out[f] for f in ("lines_equal", "distance_bits_equal", "labels_equal", "ncode_equal"))
    return out


def self_hit(I, first_id=0):
    """queries = the stored vectors first_id .. first_id+nq-1: fraction found first / anywhere in the row"""
    ids = np.arange(first_id, first_id + I.shape[0])
    return float((I[:, 0] == ids).mean()), float((I == ids[:, None]).any(axis=1).mean())
