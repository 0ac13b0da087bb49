"""CPU: the C-ABI library loads and exports every symbol include/vlq_ivfpq.h
declares; without a GPU every compute entry point fails loudly (no fallback)."""
import os
import re

import pytest

import vector_line_quantization_amd as vlq
from vector_line_quantization_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    syms = set()
    for hdr in ("vlq_ivfpq.h", "vlq_line.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(vlq_[a-z0-9_]+)\s*\(", txt))
    return sorted(syms)


def test_header_and_loader_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_all_symbols_exported():
    L = vlq.lib()
    for s in header_symbols():
        assert hasattr(L, s), s
    assert L.vlq_version() >= 100


def test_fails_loudly_without_gpu():
    if vlq.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(vlq.VlqError) as e:
        vlq.GpuIVFPQ(64, 16, 8, 8)
    assert "no HIP device" in str(e.value)


def test_product_does_not_import_oracle():
    """The product package must not import, link, load or execute anything of oracle/."""
    pkg = os.path.join(ROOT, "vector_line_quantization_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|pyoracle|libivfpq_oracle|#include\s+[\"<][^\n]*oracle|orc_[a-z_]+\s*\()")
    for dp, _dn, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cuh", ".cpp")) or fn == "Makefile":
                src = open(os.path.join(dp, fn), errors="replace").read()
                assert not bad.search(src), os.path.join(dp, fn)
