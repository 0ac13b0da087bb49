"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

The CPU-baseline leg of bench.py: run the REFERENCE's own CPU IndexIVFPQ (compiled from
/root/reference by oracle/ref.mk into oracle/_ref/, which travels to the GPU box as a built
artefact) on the host cores, on the same index and queries as the MI355X run.  The index
goes over in the reference's on-disk format (index_io.cpp:226-317: "IvPQ" with an "IxF2"
quantizer), the queries and results as tagged arrays (tests/golden/tagged.py)."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from tagged import read_tagged, write_tagged  # noqa: E402

DRIVER = os.path.join(HERE, "_ref", "ref_driver")


def available():
    return os.path.exists(DRIVER) and os.path.exists(os.path.join(HERE, "_ref", "libfaiss_ref.so"))


def _header(f, d, ntotal):
    # write_index_header (index_io.cpp:147-155): d, ntotal, 2 dummies, is_trained, metric (L2 = 1)
    f.write(struct.pack("<iqqq?i", d, ntotal, 1 << 20, 1 << 20, True, 1))


def _vec(f, a):
    a = np.ascontiguousarray(a)
    f.write(struct.pack("<Q", a.size))
    f.write(a.tobytes())


def write_ivfpq_index(path, coarse, pq_centroids, nbits, codes, ids, list_offsets, nprobe=1):
    """IndexIVFPQ over IndexFlatL2 in the reference's file format."""
    nlist, d = coarse.shape
    M = pq_centroids.shape[0]
    ntotal = int(list_offsets[-1])
    with open(path, "wb") as f:
        f.write(b"IvPQ")
        _header(f, d, ntotal)
        f.write(struct.pack("<QQ", nlist, nprobe))
        f.write(b"IxF2")
        _header(f, d, nlist)
        _vec(f, coarse.astype(np.float32))
        for i in range(nlist):
            _vec(f, ids[list_offsets[i]:list_offsets[i + 1]].astype(np.int64))
        f.write(struct.pack("<?", False))            # maintain_direct_map
        _vec(f, np.empty(0, np.int64))                # direct_map
        f.write(struct.pack("<?Q", True, M))          # by_residual, code_size
        f.write(struct.pack("<QQQ", d, M, nbits))     # ProductQuantizer
        _vec(f, pq_centroids.astype(np.float32))
        for i in range(nlist):
            _vec(f, codes[list_offsets[i]:list_offsets[i + 1]].astype(np.uint8))


def run_reference(index_path, xq, nprobe, k, reps, threads, timeout=600):
    """-> D, I of the last run, seconds per run, (use_precomputed_table, ncode per run, omp threads)"""
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(HERE, "_ref", "mkl") + ":" + os.path.join(HERE, "_ref") + ":" + env.get("LD_LIBRARY_PATH", "")
    env["OMP_NUM_THREADS"] = str(threads)
    env["MKL_NUM_THREADS"] = str(threads)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        write_tagged(fin, {"xq": np.ascontiguousarray(xq, np.float32)})
        subprocess.run([DRIVER, "bench", index_path, fin, fout, str(nprobe), str(k), str(reps), str(threads)],
                       check=True, env=env, timeout=timeout, cwd=ROOT)   # the driver names libfaiss_ref.so relative to the repo root
        out = read_tagged(fout)
    return out["D"], out["I"], out["seconds"].astype(np.float64), out["meta"]
