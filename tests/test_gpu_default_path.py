"""GPU: what bench.py measures is the library's DEFAULT path.  With no VLQ_* variable in the environment, the headline shape
(BASELINE configs[1]: 10 000 queries, nprobe 32, k 10 on the bench index) must launch the two-wave one-buffer 16-byte scan,
scan16_kernel<1, 2, 1, false, false, false>, with the list-id walk of a query's probes, and the generator setting of rounds 1-3
(dense clusters: neighbouring queries share most lists) must keep the reference's coarse-distance order -- so the measured
default cannot drift silently behind the A/B switches of DESIGN.md section 9.  Runs in a fresh process (the switches are read
once per process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CODE = r"""
import json, sys, types
sys.path.insert(0, ".")
import torch
import bench
dev = torch.device("cuda", 0)
out = {}
for name, kw in (("headline", dict(sigma=0.005, rank=12, spread=0.4)), ("g1", dict(sigma=0.03, rank=0, spread=0.0))):
    a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, gmm_centres=2000, **kw)
    g, centres, coarse, pq, xb = bench.build_index(a, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(33)
    xq = bench.gmm(torch, gen, centres, 10000, a.sigma, dev, a.rank, a.spread)
    D = torch.empty((10000, 10), dtype=torch.float32, device=dev); I = torch.empty((10000, 10), dtype=torch.int64, device=dev)
    for _ in range(3): g.search(xq, 32, 10, D=D, I=I)
    out[name] = g.last_scan_info()
    g.search(xq[:1250].contiguous(), 32, 10, D=D[:1250], I=I[:1250])
    out[name + "_slice"] = g.last_scan_info()
print("INFO " + json.dumps(out))
"""


def test_headline_shape_runs_the_default_kernel_and_order():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("VLQ_")}
    p = subprocess.run([sys.executable, "-c", CODE], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("INFO ")][-1]
    info = json.loads(line[5:])
    assert "kernel=scan16_kernel<1, 2, 1, false, false, false>" in info["headline"], info
    assert "order=list-id walk" in info["headline"], info
    assert "period_ticks=0" not in info["headline"], info             # the clock period has been measured by the third call
    assert "kernel=scan16_kernel<1, 2, 1, false, false, false>" in info["g1"], info
    assert "order=coarse-distance order" in info["g1"], info
    # the per-GPU slice of a batch sharded over 8 GPUs: four waves, one buffer
    assert "kernel=scan16_kernel<1, 4, 1, false, false, false>" in info["headline_slice"], info
