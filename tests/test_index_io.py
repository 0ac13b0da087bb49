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
