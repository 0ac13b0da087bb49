#!/usr/bin/env python3
"""Copy the summaries of a profiles/refresh.sh run (gpurun_out/<round>/) into profiles/<round>_*.
   python profiles/collect.py r01"""
import glob, json, os, re, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), "gpurun_out", R)
dst = lambda n: os.path.join(HERE, "%s_%s" % (R, n))
one = lambda pat: max(glob.glob(os.path.join(SRC, pat), recursive=True), key=os.path.getmtime)   # gpurun merges runs: newest wins

shutil.copy(os.path.join(SRC, "bench.json"), dst("bench.json"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), dst("bench_under_rocprof.json"))
shutil.copy(one("trace/**/*kernel_stats.csv"), dst("bench_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "pmc_summary.txt"), dst("pmc_summary.txt"))
# r01-r03: the second pass ran the low-intrinsic-dimension data beside a G1 headline; since r04 the headline IS that data
# and the second pass is G1 (bench.py's first_dataset leg)
shutil.copy(os.path.join(SRC, "pmc2_summary.txt"), dst("pmc_summary_first_dataset.txt" if R >= "r04" else "pmc_summary_second_dataset.txt"))
if os.path.exists(os.path.join(SRC, "pmc_owned_summary.txt")):
    shutil.copy(os.path.join(SRC, "pmc_owned_summary.txt"), dst("pmc_summary_owned_schedule.txt"))
shutil.copy(os.path.join(SRC, "sweep.md"), dst("sweep.md"))
shutil.copy(one("imi10/**/*kernel_stats.csv"), dst("imi_kernel_stats.csv"))
shutil.copy(one("imi14/**/*kernel_stats.csv"), dst("imi14_kernel_stats.csv"))
shutil.copy(one("vlq/**/*kernel_stats.csv"), dst("vlq16m_kernel_stats.csv" if R >= "r04" else "vlq_kernel_stats.csv"))   # r04: vlq_kernel_stats = the 998 M-code run
for extra in ("vlq_fp16.log", "sched_ab.txt", "host_buffers.txt", "slices.txt", "slices_g1.txt"):
    if os.path.exists(os.path.join(SRC, extra)):
        shutil.copy(os.path.join(SRC, extra), dst(extra if extra.endswith(".txt") else extra.replace(".log", ".txt")))
with open(dst("long_lists.txt"), "w") as f:
    f.write("# tools/long_lists.py: random codes straight into the lists, 10 000 queries; K / DIM / NPROBE as in profiles/refresh.sh\n")
    f.write(open(os.path.join(SRC, "long_lists.txt")).read())
    f.write("# tools/large_k.py (bench index, nprobe 32, 10 000 queries)\n")
    f.write(open(os.path.join(SRC, "large_k.txt")).read())
with open(dst("imi_vlq.txt"), "w") as f:
    for n in ("imi10.log", "imi14.log", "vlq4m.log", "vlq.log", "vlq_fp16.log", "vlq_c5_1b.log", "vlq_c5_1b_fp16.log", "vlq_c5_1b_rows2.log"):
        if not os.path.exists(os.path.join(SRC, n)):
            continue
        lines = [l for l in open(os.path.join(SRC, n)).read().splitlines()
                 if l.startswith(("added", "loaded", "search", "look-up", "self-hit", "oracle sample", "VERIFIED")) ]
        f.write("# %s\n%s\n" % (n, "\n".join(lines)))

clean = lambda path: [l for l in open(path).read().splitlines() if l and not re.match(r"[EWI]20\d\d", l) and not l.startswith("[bench]")]
# round 4: code sizes (engineered lines re-measured, the generic-kernel block of the file kept), table mode 0, coarse stage
# alone, the VLQ scan at the driver's geometry (kernel stats + counter passes -> r04_vlq_traffic.json)
if os.path.exists(os.path.join(SRC, "code_sizes.log")):
    old = open(dst("code_sizes.txt")).read().splitlines() if os.path.exists(dst("code_sizes.txt")) else []
    head = [l for l in old if l.startswith("#") and "generic kernel" not in l]
    generic = [l for l in old if "GENERIC KERNEL" in l or "generic kernel" in l]
    with open(dst("code_sizes.txt"), "w") as f:
        f.write("\n".join(head + [l for l in clean(os.path.join(SRC, "code_sizes.log")) if " M=" in l] +
                          sorted(generic, key=lambda l: not l.startswith("#"))) + "\n")
        if os.path.exists(os.path.join(SRC, "mode0.log")):
            f.write("# tools/time_mode0.py: table mode 0 (no precomputed table) against mode 1, headline data\n")
            f.write("\n".join(l for l in clean(os.path.join(SRC, "mode0.log")) if "use_precomputed_table" in l) + "\n")
    shutil.copy(one("code_sizes/**/*kernel_stats.csv"), dst("code_sizes_kernel_stats.csv"))
if os.path.exists(os.path.join(SRC, "coarse.log")):
    with open(dst("coarse_f32.txt"), "w") as f:
        f.write("# tools/time_coarse.py under rocprofv3 --kernel-trace --stats: the coarse stage with the float16 screen off (f32 MFMA matrix + select)\n")
        for n in ("coarse", "coarse_vlq"):
            f.write("\n".join(l for l in clean(os.path.join(SRC, n + ".log")) if l.startswith("coarse")) + "\n")
            import csv
            for r in csv.reader(open(one(n + "/**/*kernel_stats.csv"))):
                if "coarse_" in r[0]:
                    f.write("    %-60s calls %s average %.1f us\n" % (r[0].split("(")[0][:60], r[1], float(r[3]) / 1e3))
if os.path.exists(os.path.join(SRC, "pmc_vlq", "summary.txt")):
    shutil.copy(one("vlq1b/**/*kernel_stats.csv"), dst("vlq_kernel_stats.csv"))
    shutil.copy(one("vlq1b_fp16/**/*kernel_stats.csv"), dst("vlq_fp16_kernel_stats.csv"))
    import hashlib
    srcs = ["line16c.hip", "line.h", "scan16_common.cuh", "wave_topk.cuh", "scan_common.cuh"]
    hh = hashlib.sha256()
    for fn in srcs:
        hh.update(open(os.path.join(os.path.dirname(HERE), "vector_line_quantization_amd", "csrc", fn), "rb").read())
    rec = {"what": "HBM bytes per scan launch of the VLQ scan at the reference driver's geometry with 998 M synthetic codes and 2000 "
                   "queries: rocprofv3 --pmc FETCH_SIZE [KB] x 1024 x 2 (gfx950 correction), profiles/%s_pmc_vlq.txt" % R,
           "sources": srcs, "sources_sha": hh.hexdigest()}
    body = ["# VLQ at the reference driver's geometry (65 536 x 64 lines, nprobe 64, w1 1024, k 128, 998 M synthetic codes, 2000 queries per launch), round 4",
            "# SYNTH=1 NLIST=65536 NEDGE=64 NB=1000000000 [FP16=1] bash profiles/pmc_cmd.sh <out> line tools/time_vlq.py 2000 3   (4 counter passes, no tracing domains; profiles/refresh.sh r04 d)",
            "# HBM bytes per launch = FETCH_SIZE [KB] x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section):"]
    parts = []
    for name, sub, tf in (("fp32_tables", "pmc_vlq", "false"), ("float16_tables", "pmc_vlq_fp16", "true")):
        t = open(os.path.join(SRC, sub, "summary.txt")).read()
        m = re.search(r"line16c_scan_kernel<2, 8, 2, %s> grid=1024000\n((?:    [^\n]*\n)+)" % tf, t)
        fs = float(re.search(r"FETCH_SIZE\s+n=\s*\d+ mean=\s*([0-9.]+)", m.group(1)).group(1))
        rec[name] = {"kernel": "line16c_scan_kernel<2, 8, 2, %s>" % tf, "fetch_size_kb": fs, "bytes": fs * 2048}
        body.append("#   line16c_scan_kernel<2,8,2,%s> (%s): %.1f KB -> %.2f GB (algorithmic 17 B x 487.4 M codes = 8.29 GB)" % (tf, name, fs, fs * 2048 / 1e9))
        parts.append("\n===== %s\n%s" % (name, t))
    open(dst("pmc_vlq.txt"), "w").write("\n".join(body) + "\n" + "".join(parts))
    json.dump(rec, open(dst("vlq_traffic.json"), "w"), indent=1)
    with open(dst("vlq_c5_synth.txt"), "w") as f:
        f.write("# SYNTH=1 NLIST=65536 NEDGE=64 NB=1000000000 [FP16=1] python tools/time_vlq.py 2000 5 (under rocprofv3 --kernel-trace --stats; profiles/refresh.sh r04 d)\n")
        for n in ("vlq1b_trace.log", "vlq1b_fp16_trace.log"):
            f.write("# %s\n%s\n" % (n, "\n".join(clean(os.path.join(SRC, n)))))

# HBM traffic of the scan kernel per launch: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 counts 64 B per
# 128-B request of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> FETCH_SIZE x 2
txt = open(dst("pmc_summary.txt")).read()
blk = re.search(r"void vlq::scan16_kernel<1[^\n]*\n((?:    [^\n]*\n)+)", txt)
vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+n=\s*\d+ mean=\s*([0-9.]+)", blk.group(1))}
mk = re.search(r"void (vlq::scan16_kernel<1[^\n]*?) grid=", txt)
out = {"kernel": mk.group(1) if mk else "vlq::scan16_kernel<1, ...> (the summary truncates kernel names to 48 characters)",
       "source": "profiles/%s_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 3)" % R,
       "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
       "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request of wide coalesced reads, MI355X_MICROARCH.md §HBM)",
       "hbm_bytes_per_launch": (vals["FETCH_SIZE"] * 2 + vals["WRITE_SIZE"]) * 1024,
       "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])}
# bench.py reports this figure only while the scan-kernel sources are the ones that were profiled
sys.path.insert(0, os.path.dirname(HERE))
import bench
out["sources_sha256"] = bench.sources_sha()
# ... and the same setting of the data generator (bench.py defaults at the time of the passes)
# ... taken from what bench.py recorded in that run (rounds before 5 did not record it: their defaults)
try:
    out["generator"] = json.load(open(dst("bench.json")))["dataset"]["generator_flags"]
except Exception:
    out["generator"] = [0.005, 12, 0.4] if R >= "r04" else list(bench.G1_FLAGS)
# the kernel's average duration in the kernel trace of the same refresh run (bench_kernel_stats.csv): the time these bytes go with
try:
    import csv as _csv
    for r in _csv.reader(open(dst("bench_kernel_stats.csv"))):
        if r and out["kernel"][:40] in r[0]:
            out["kernel_ms_in_same_refresh_run"] = float(r[3]) / 1e6
            break
except Exception:
    pass
json.dump(out, open(dst("scan_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
