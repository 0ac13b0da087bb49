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
shutil.copy(one("vlq/**/*kernel_stats.csv"), dst("vlq_kernel_stats.csv"))
for extra in ("vlq_fp16.log", "sched_ab.txt", "host_buffers.txt", "slices.txt"):
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
out["generator"] = [0.005, 12, 0.4] if R >= "r04" else list(bench.G1_FLAGS)
json.dump(out, open(dst("scan_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
