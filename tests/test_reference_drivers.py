"""INTEGRATION.md section B executed: the REFERENCE's own driver tests/demo_sift1M.cpp, compiled in place from
/root/reference/tests (tests/cpp/Makefile `ref_drivers`, build container only) and linked with the reference's own
CPU library plus integration/reference_interposer.cpp in front of it, runs on the MI355X without a changed line:
index_factory("IVF4096,PQ8+16") -> IndexIVFPQR -> IndexIVFPQ::search_knn_with_key lands in
vlq_ivfpq_search_preassigned.  The SAME binary with VLQ_INTERPOSE=off is the CPU-only run it is compared with.

The driver auto-tunes (ParameterSpace::explore): which operating points it visits and which one it selects depends
on measured times, so the comparison is made where both runs did the same thing -- every operating point both
runs report must have the same 1-recall@1, and when both select the same configuration the final R@1 / R@10 /
R@100 lines must be equal.  The generated data have real-valued coordinates (no exact distance ties)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = os.path.join(ROOT, "tests", "cpp", "ref_drivers")
# database vectors of the multi-index driver runs: 2^28 lists are the drivers' (4.3 GB index files whatever the data); with
# 400 000 vectors a query's 2048 probed cells hold ~60 codes, so that most result rows are full (at 100 000 most were padding)
NB = 400000


def _env(extra):
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = os.path.join(ROOT, "oracle/_ref/mkl") + ":" + e.get("LD_LIBRARY_PATH", "")
    e["OMP_NUM_THREADS"] = "8"            # the reference's OpenMP loops: never one spinning thread per logical CPU
    e["OMP_WAIT_POLICY"] = "passive"
    e.update(extra)
    return e


def _parse(out):
    pts = {m.group(1): float(m.group(2)) for m in re.finditer(r"cno=\d+ key=(\S+) perf=([0-9.]+)", out)}
    sel = re.search(r'Setting parameter configuration "([^"]*)"', out)
    rec = tuple(re.findall(r"R@(?:1|10|100) = ([0-9.]+)", out)[-3:])
    return pts, (sel.group(1) if sel else None), rec


def _run_cached_pair(exe, data, run, populate_first=False):
    """Two runs of the SAME binary from the driver's cached populated index: CPU-only, then the device.  The cache is written
    by tools/make_driver_data.py, or -- populate_first -- by a first device run of the driver itself: its own add loop
    (add_with_ids -> the interposed add_core_o: residuals and codes computed on the device), which then writes the cache the
    two compared runs start from (a populating run's adds would leave the host's OpenMP / MKL pools warm for one side only)."""
    outs = {}
    for mode in (("populate",) if populate_first else ()) + ("off", "on"):
        p = subprocess.run([exe], env=_env({"VLQ_DATA_ROOT": data, "VLQ_INTERPOSE": "on" if mode == "populate" else mode}),
                           capture_output=True, text=True, timeout=1500, cwd=run)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs[mode] = (p.stdout, p.stderr)
        print(mode, "\n".join(p.stdout.splitlines()[-6:]), p.stderr.splitlines()[-1])
    if populate_first:
        m = re.search(r"adds=(\d+) vectors=(\d+)", outs["populate"][1])
        assert m and int(m.group(1)) >= 1 and int(m.group(2)) == NB, outs["populate"][1][-400:]       # the driver's add ran on the device
        m = re.search(r"adds=(\d+) vectors=(\d+)", outs["on"][1])
        assert m and int(m.group(1)) == 0                                                            # ... and the compared run read the cache
    return outs


def _real_rows_agree(ic, idd):
    """Printed neighbour rows of the two runs: the reference pads a short row with id -1 -- the padding must be equally long in
    both runs, and the REAL ids must be the same up to one near-tie per row (padding never counts as agreement)."""
    assert len(ic) == len(idd) and len(ic) > 0
    full = 0
    for a, b in zip(ic, idd):
        ra, rb = [v for v in a if v >= 0], [v for v in b if v >= 0]
        assert len(ra) == len(rb), (a, b)
        assert len(set(ra) & set(rb)) >= max(len(set(ra)) - 1, min(len(ra), 1)), (a, b)
        full += len(ra) == len(a)
    return full


def _faster_on_the_device(outs):
    """The driver's OWN clock around its search call (`avg. query time`): the device run's is below the CPU-only run's.  The
    index went to the device in the interposed precompute_table (read_index calls it, index_io.cpp:492-495) together with the
    one-time launch and workspace costs -- the interposer's report line says so (`list_uploads=1`); the timed call is
    IndexIVFPQ::search served whole (coarse stage and scan on the device, `whole_searches=1`): nothing of the reference's
    MultiIndexQuantizer::search is left in it (`multi_index_search_seconds` of the device run <= 10 % of the CPU run's)."""
    t, knn, miq = {}, {}, {}
    for mode in ("off", "on"):
        m = re.search(r"avg\. query time\s+([0-9.eE+-]+)\s*ms", outs[mode][0])
        assert m, outs[mode][0][-600:]
        t[mode] = float(m.group(1))
        m = re.search(r"knn_with_key_seconds=([0-9.]+) multi_index_search_seconds=([0-9.]+)", outs[mode][1])
        assert m, outs[mode][1][-400:]
        knn[mode], miq[mode] = float(m.group(1)), float(m.group(2))
    whole = re.search(r"whole_searches=(\d+) whole_search_seconds=([0-9.]+)", outs["on"][1])
    assert whole and int(whole.group(1)) == 1, outs["on"][1][-400:]
    print("avg. query time: CPU-only %.4f ms, device %.4f ms (x %.1f); CPU-only run: search_knn_with_key %.4f s, MultiIndexQuantizer::search "
          "%.4f s; device run: whole search %.4f s, of the reference's MultiIndexQuantizer::search %.4f s" % (
              t["off"], t["on"], t["off"] / max(t["on"], 1e-9), knn["off"], miq["off"], float(whole.group(2)), miq["on"]))
    up = re.search(r"list_uploads=(\d+) cpu_fallbacks=(\d+)", outs["on"][1])
    assert up and int(up.group(1)) == 1 and int(up.group(2)) == 0, outs["on"][1][-400:]
    assert miq["on"] <= 0.1 * miq["off"], (miq, t)
    assert t["on"] < t["off"], (t, knn, miq)            # the driver's own printed time


def test_interposer_exports_the_reference_symbols():
    """CPU: the prebuilt interposer defines the three member functions under the reference's mangled names."""
    so = os.path.join(RD, "libvlq_interpose.so")
    if not os.path.exists(so):
        pytest.skip("tests/cpp/ref_drivers was not prebuilt (needs the reference tree at build time)")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    for s in ("_ZNK5faiss10IndexIVFPQ19search_knn_with_keyEmPKfPKlS2_PNS_9HeapArrayINS_4CMaxIflEEEEb",
              "_ZN5faiss10IndexIVFPQ10add_core_oElPKfPKlPfS4_", "_ZN5faiss10IndexIVFPQ16precompute_tableEv"):
        assert s in syms, s
    for exe in ("demo_sift1M", "sift1b_imi_pq", "deep1b_imi_pq", "deep1b16_imi_pq"):
        assert os.access(os.path.join(RD, exe), os.X_OK)


@pytest.mark.gpu
def test_demo_sift1M_unchanged_on_the_device(tmp_path):
    exe = os.path.join(RD, "demo_sift1M")
    if not (os.path.exists(exe) and os.path.exists(os.path.join(ROOT, "oracle/_ref/libfaiss_ref.so"))):
        pytest.skip("tests/cpp/ref_drivers was not prebuilt (needs the reference tree at build time)")
    data = str(tmp_path / "data")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_driver_data.py"), data, "40000", "200000", "1000"])
    # The driver's AutoTune exploration (ParameterSpace::explore) times every operating point and prunes by what it measured,
    # so WHICH points a run visits depends on the box's load: the counts asserted below are what a quiet box gives.  A run
    # that visited too few points is repeated (twice at most); a recall that differs at a common point fails every time.
    def attempt():
        runs = {}
        for mode in ("off", "on"):
            p = subprocess.run([exe], env=_env({"VLQ_DATA_ROOT": data, "VLQ_INTERPOSE": mode}), capture_output=True, text=True,
                               timeout=900, cwd=str(tmp_path))
            assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
            runs[mode] = (p.stdout, p.stderr)
        pts_c, sel_c, rec_c = _parse(runs["off"][0])
        pts_d, sel_d, rec_d = _parse(runs["on"][0])
        print("cpu  :", sel_c, rec_c, len(pts_c), "operating points")
        print("device:", sel_d, rec_d, len(pts_d), "operating points")
        summ = re.search(r"\[vlq-interpose\] device searches=(\d+) queries=(\d+) ncode=(\d+) .*cpu_fallbacks=(\d+)", runs["on"][1])
        assert summ, runs["on"][1][-500:]
        print(summ.group(0))
        searches, queries, ncode = (int(summ.group(i)) for i in (1, 2, 3))
        assert searches >= 5 and queries >= 5000 and ncode > 10 ** 7        # the plain (ht = 64) operating points ran on the device
        off = re.search(r"\[vlq-interpose\] device searches=(\d+)", runs["off"][1])
        assert off and int(off.group(1)) == 0                                # ... and nothing did in the CPU-only run
        common = sorted(set(pts_c) & set(pts_d) - {""})
        assert len(common) >= 2, (pts_c, pts_d)
        for key in common:
            assert pts_c[key] == pts_d[key], (key, pts_c[key], pts_d[key])
        assert any("ht=64" in k for k in common)
        assert len(rec_c) == 3 and len(rec_d) == 3
        if sel_c and sel_c == sel_d:
            assert rec_c == rec_d

    for tries_left in (2, 1, 0):
        try:
            attempt()
            break
        except AssertionError as e:
            print("attempt failed (%d left): %s" % (tries_left, str(e)[:600]))
            if tries_left == 0:
                raise


@pytest.mark.gpu
def test_sift1b_imi_pq_unchanged_on_the_device(tmp_path):
    """tests/sift1b_imi_pq.cpp as shipped (inverted multi-index 2 x 14 bits = 2^28 lists, 8-byte codes, nprobe 2048,
    k 128), compiled in place: CPU-only run, then the device run of the SAME binary, both on the driver's cached populated index.  The 2^28 lists are the driver's (two 4.3 GB index files whatever the data): the database is kept
    at 400 000 vectors (2000 queries) so that the test is about a minute (VLQ_SKIP_HEAVY_DRIVERS=1 leaves it out).
    The driver's training step (2 M vectors, k-means into 2 x 16 384 centroids on the host) is skipped through the
    driver's own cache branch (:237-251): tools/make_driver_data.py writes the trained-index file it looks for."""
    if os.environ.get("VLQ_SKIP_HEAVY_DRIVERS") == "1":
        pytest.skip("VLQ_SKIP_HEAVY_DRIVERS=1")
    exe = os.path.join(RD, "sift1b_imi_pq")
    if not (os.path.exists(exe) and os.path.exists(os.path.join(ROOT, "oracle/_ref/libfaiss_ref.so"))):
        pytest.skip("tests/cpp/ref_drivers was not prebuilt (needs the reference tree at build time)")
    data, run = str(tmp_path / "data"), str(tmp_path / "run")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_driver_data.py"), data, "sift1b", run, str(NB), "2000", "1"])
    outs = _run_cached_pair(exe, data, run)

    def parse(out):
        rec = [float(v) for v in re.findall(r"R@(?:1|10|100) = ([0-9.]+)", out)[-3:]]
        ids = [tuple(int(v) for v in ln.split(":", 1)[1].split()) for ln in out.splitlines() if re.match(r"query\s+\d+:", ln)]
        dis = [tuple(ln.split(":", 1)[1].split()) for ln in out.splitlines() if ln.strip().startswith("dis:")]
        return rec, ids, dis
    rc, ic, dc = parse(outs["off"][0])
    rd, idd, dd = parse(outs["on"][0])
    summ = re.search(r"\[vlq-interpose\] device searches=(\d+) queries=(\d+) ncode=(\d+)", outs["on"][1])
    assert summ and int(summ.group(1)) >= 1 and int(summ.group(2)) == 2000 and int(summ.group(3)) > 0
    _faster_on_the_device(outs)
    # 8-byte codes on 128 dimensions = 16-dimensional sub-vectors: there the reference computes its tables through
    # BLAS (ProductQuantizer.cpp:445-461,470-492; vendor-defined rounding), so the two runs agree to
    # rounding -- the north star's 1e-4 relative -- not digit for digit; the neighbours are the same up to near-ties
    assert len(rc) == 3 and len(dc) == 10 and len(dd) == 10
    for a, b in zip(dc, dd):
        assert len(a) == len(b) == 10
        assert max(abs(float(x) - float(y)) / max(1e-9, abs(float(x))) for x, y in zip(a, b)) <= 1e-4
    assert _real_rows_agree(ic, idd) >= 8              # most printed rows are full: 10 real neighbours
    assert max(abs(a - b) for a, b in zip(rc, rd)) <= 0.004


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["deep1b_imi_pq", "deep1b16_imi_pq"])
def test_deep1b_drivers_unchanged_on_the_device(tmp_path, driver):
    """tests/deep1b_imi_pq.cpp and tests/deep1b16_imi_pq.cpp as shipped (BASELINE configs[3] / [4] name them: 96 dimensions,
    inverted multi-index 2 x 14 bits = 2^28 lists, 8- / 16-byte codes, nprobe 2048, k 128), compiled in place: CPU-only run,
    then the device run of the SAME binary, both on the driver's cached populated index; 8-byte codes are served by
    scanm_kernel<8, ..., IMI>, 16-byte codes by scan16's table type 2.  Like the sift1b driver: 2^28 lists (4.3 GB index
    files), 400 000 database vectors, 2000 queries, one to two minutes each (VLQ_SKIP_HEAVY_DRIVERS=1 leaves them out)."""
    if os.environ.get("VLQ_SKIP_HEAVY_DRIVERS") == "1":
        pytest.skip("VLQ_SKIP_HEAVY_DRIVERS=1")
    exe = os.path.join(RD, driver)
    if not (os.path.exists(exe) and os.path.exists(os.path.join(ROOT, "oracle/_ref/libfaiss_ref.so"))):
        pytest.skip("tests/cpp/ref_drivers was not prebuilt (needs the reference tree at build time)")
    data, run = str(tmp_path / "data"), str(tmp_path / "run")
    # deep1b16_imi_pq populates its index itself (the driver's own add loop on the device); deep1b_imi_pq starts from a written cache
    own_add = driver == "deep1b16_imi_pq"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_driver_data.py"), data, "deep1b", run, str(NB), "2000",
                           "-16" if own_add else "8"])
    outs = _run_cached_pair(exe, data, run, populate_first=own_add)

    def parse(out):
        rec = [float(v) for v in re.findall(r"R@(?:1|10|100) = ([0-9.]+)", out)[-3:]]
        ids = [tuple(int(v) for v in ln.split(":", 1)[1].split()) for ln in out.splitlines() if re.match(r"query\s+\d+:", ln)]
        dis = [tuple(ln.split(":", 1)[1].split()) for ln in out.splitlines() if ln.strip().startswith("dis:")]
        return rec, ids, dis
    rc, ic, dc = parse(outs["off"][0])
    rd, idd, dd = parse(outs["on"][0])
    summ = re.search(r"\[vlq-interpose\] device searches=(\d+) queries=(\d+) ncode=(\d+) .*cpu_fallbacks=(\d+)", outs["on"][1])
    assert summ and int(summ.group(1)) >= 1 and int(summ.group(2)) == 2000 and int(summ.group(3)) > 0 and int(summ.group(4)) == 0
    _faster_on_the_device(outs)
    # sub-vectors of 12 / 6 dimensions: the reference's PQ tables take its SSE path (no BLAS); the coarse stage's 48-dimensional
    # half tables go through the BLAS vendor's sgemm in the CPU run and through the k-ordered fmaf chain on the device --
    # distances to the north star's 1e-4, the same neighbours up to near-ties
    assert len(rc) == 3 and len(dc) == 10 and len(dd) == 10
    for a, b in zip(dc, dd):
        assert max(abs(float(x) - float(y)) / max(1e-9, abs(float(x))) for x, y in zip(a, b)) <= 1e-4
    assert _real_rows_agree(ic, idd) >= 8              # most printed rows are full: 10 real neighbours
    assert max(abs(a - b) for a, b in zip(rc, rd)) <= 0.004


@pytest.mark.gpu
@pytest.mark.parametrize("d", [16, 32])
def test_quantizer_and_whole_search_calls_of_a_user_program(d):
    """tests/cpp/miq_search_calls.cpp (reference headers, linked like the drivers): MultiIndexQuantizer::search with k > 1 called by
    the program itself lands in the device coarse stage of the handle that holds the quantizer's sub-centroids, and
    IndexIVFPQ::search with up to 2048 probes is served whole -- each against the reference's own definitions reached through
    dlsym: bit for bit with 8-dimensional sub-vectors (the reference's SSE tables), to rounding with 16-dimensional ones (its BLAS)."""
    exe = os.path.join(RD, "miq_search_calls")
    if not (os.path.exists(exe) and os.path.exists(os.path.join(ROOT, "oracle/_ref/libfaiss_ref.so"))):
        pytest.skip("tests/cpp/ref_drivers was not prebuilt (needs the reference tree at build time)")
    p = subprocess.run([exe, str(d)], env=_env({"VLQ_INTERPOSE": "on"}), capture_output=True, text=True, timeout=900)
    print(p.stdout[-3000:], p.stderr[-600:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    m = re.search(r"whole_searches=(\d+) .*device_coarse_searches=(\d+)", p.stderr)
    assert m and int(m.group(1)) == 5 and int(m.group(2)) == 4, p.stderr[-400:]        # 3 multi-index + 2 flat whole searches
