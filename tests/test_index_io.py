"""CPU: the C++ shell's read_index / write_index against index files written by the
reference's own write_index (stored as data in the golden fixtures): every field is
recovered and writing the index back is byte-identical to the reference's file."""
import os
import subprocess

import numpy as np
import pytest

from util import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


@pytest.mark.parametrize("name", ["tiny_padding", "imi_sse_tables"])
def test_reference_file_round_trips_byte_identically(tmp_path, name):
    case = Case(name)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "vector_line_quantization_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", CPP, "io_roundtrip"])
    fin, fout = tmp_path / "ref.faissindex", tmp_path / "ours.faissindex"
    case["faissindex_file"].tofile(fin)
    p = subprocess.run([os.path.join(CPP, "io_roundtrip"), str(fin), str(fout)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    fields = dict(kv.split("=") for kv in p.stdout.split())
    assert int(fields["d"]) == case.d and int(fields["nlist"]) == case.nlist
    assert int(fields["M"]) == case.M and int(fields["nbits"]) == case.nbits
    assert int(fields["ntotal"]) == case.nb and int(fields["nvec"]) == case["ids"].shape[0]
    assert int(fields["ncodes"]) == case["codes"].size
    assert fields["quantizer"] == ("Imiq" if case.imi_nbits else "IxF2")
    assert int(fields["qntotal"]) == case.nlist
    assert np.array_equal(np.fromfile(fout, dtype=np.uint8), case["faissindex_file"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_padding", "imi_sse_tables"])
def test_reference_written_file_searches_on_gpu_like_the_reference(tmp_path, name):
    """a `*_populated_index.faissindex` written by the REFERENCE (the fixture's
    `faissindex_file` bytes) is read by the shell's read_index, its lists go to HBM, and the search returns
    the reference's own D / I of that fixture -- through the IndexIVFPQ object and, for a flat quantizer,
    through GpuIndexIVFPQ::copyFrom (gpu/GpuIndexIVFPQ.cu:168-231; an IMI quantizer is refused there as in the
    reference, gpu/GpuIndexIVF.cu:131-133)."""
    from util import assert_same_topk, label_agreement
    case = Case(name)
    exe = os.path.join(CPP, "search_from_file")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", CPP, "search_from_file"])
    fin, fq, out = tmp_path / "ref.faissindex", tmp_path / "q.f32", str(tmp_path / "res")
    case["faissindex_file"].tofile(fin)
    np.ascontiguousarray(case.xq, dtype=np.float32).tofile(fq)
    p = subprocess.run([exe, str(fin), str(fq), str(case.nq), str(case.nprobe), str(case.k), out],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ntotal=%d" % case.nb in p.stdout
    variants = [""] if case.imi_nbits else ["", ".gpu"]
    assert ("gpu: copyFrom ntotal=%d" % case.nb in p.stdout) == (not case.imi_nbits)
    for v in variants:
        D = np.fromfile(out + v + ".D", dtype=np.float32).reshape(case.nq, case.k)
        I = np.fromfile(out + v + ".I", dtype=np.int64).reshape(case.nq, case.k)
        if np.array_equal(D.view(np.uint32), case["D"].view(np.uint32)):
            assert_same_topk(D, I, case["D"], case["I"], name + v)       # bit-equal distances, labels up to ties
        else:       # the >= 20-query coarse stage is BLAS on the reference side: rounding at the nprobe-th place
            assert label_agreement(D, I, case["D"], case["I"]) >= 0.99
            m = (I == case["I"]) & (I >= 0)
            assert (np.abs(D[m] - case["D"][m]) <= 1e-4 * np.abs(case["D"][m])).all()
