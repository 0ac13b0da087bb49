"""Shared helpers: golden fixture loading and the tie-canonical comparison."""
import glob
import hashlib
import os

import numpy as np

from oracle import pyoracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.z = {k: z[k] for k in z.files}
        c = self.z["cfg"]
        (self.d, self.nlist, self.M, self.nbits, self.nt, self.nb, self.nq, self.nprobe, self.k,
         self.max_codes, self.n_small, _, _, self.by_residual, _) = [int(v) for v in c[:15]]
        self.imi_nbits = int(c[15]) if len(c) > 15 else 0
        self.mode = int(self.z["meta"][0])
        self.xq = self.z["xq"]
        self.xb = self.z["xb_u8"].astype(np.float32) if "xb_u8" in self.z else self.z["xb"]
        self.xids = self.z.get("xids")

    def __getitem__(self, k):
        return self.z[k]

    def oracle_index(self, with_lists=True):
        z = self.z
        return pyoracle.OracleIndex(
            self.d, self.nlist, self.M, self.nbits, z.get("coarse_centroids"), z["pq_centroids"],
            imi_centroids=z.get("imi_centroids"), imi_nbits=self.imi_nbits,
            codes=z["codes"] if with_lists else None, ids=z["ids"] if with_lists else None,
            list_offsets=z["list_offsets"] if with_lists else None,
            by_residual=self.by_residual, use_precomputed_table=self.mode, max_codes=self.max_codes)


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_topk(D, I, Dref, Iref, what=""):
    """Distances bit-identical; labels identical except for permutations inside
    groups of exactly equal distance, and -- for the group that straddles the k-th
    place -- a possibly different choice among equally distant candidates (the rule
    of the reference's own select test, gpu/test/TestGpuSelect.cu:82-114).  Returns
    the number of rows whose boundary group could not be checked label-by-label."""
    assert D.shape == Dref.shape and I.shape == Iref.shape
    assert np.array_equal(bits(D), bits(Dref)), "%s: distances differ bitwise" % what
    n, k = D.shape
    loose = 0
    for r in range(n):
        if np.array_equal(I[r], Iref[r]):
            continue
        d = D[r]
        start = 0
        while start < k:
            end = start + 1
            while end < k and d[end] == d[start]:
                end += 1
            a, b = sorted(I[r, start:end]), sorted(Iref[r, start:end])
            if a != b:
                # only legal for the tie group touching the k-th place
                assert end == k and end - start >= 1 and d[start] != np.float32(np.finfo(np.float32).max), \
                    "%s: row %d labels differ outside a boundary tie group" % (what, r)
                loose += 1
            start = end
    return loose


def tie_canonical(D, I):
    """Sort labels inside groups of exactly equal distance (rows are ascending in D)."""
    out = I.copy()
    for r in range(D.shape[0]):
        order = np.lexsort((I[r], D[r]))
        out[r] = I[r][order]
    return out


def label_agreement(D, I, Dr, Ir):
    """Fraction of result slots that agree with the reference, where labels inside a
    group of exactly equal distance are compared as sets and, in the group touching
    the k-th place, equally distant alternatives count as agreeing."""
    n, k = D.shape
    ok = 0
    for r in range(n):
        s = 0
        while s < k:
            e = s + 1
            while e < k and Dr[r, e] == Dr[r, s]:
                e += 1
            a, b = set(I[r, s:e].tolist()), set(Ir[r, s:e].tolist())
            if np.array_equal(D[r, s:e], Dr[r, s:e]):
                ok += (e - s) if (e == k or a == b) else len(a & b)
            else:
                ok += len(a & b)
            s = e
    return ok / float(n * k)
