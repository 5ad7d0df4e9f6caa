#!/usr/bin/env python3
"""Same-box A/B of two builds of the library: fused-kernel time (best of 3) and whole-step time on
the usual shapes, each build in a process of its own, alternating.
    python tools/exp/ab_lib.py tools/exp/libscmiac_old.so scmiac_amd/csrc/libscmiac_hip.so"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, json, time
from pathlib import Path
sys.path.insert(0, %r)
from scmiac_amd import _hip
_hip.LIB_PATH = Path(sys.argv[1]).resolve()
import torch, bench
out = {}
for name, n in (("c2", None), ("c4", None), ("c3", 125000), ("c3", None), ("cca30", 250000)):
    U, V, _, _, d = bench.make_latents(name, n, None)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    best, step = 1e9, 1e9
    for _ in range(4 if q.shape[0] * r.shape[0] < 5e11 else 2):
        _hip.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _hip.knn_search(q, r, 30)
        torch.cuda.synchronize(); step = min(step, (time.perf_counter() - t0) * 1e3)
        ms, _ = _hip.fused_ms()
        _hip.set_timing(False)
        best = min(best, ms)
    out[f"{name} n={q.shape[0]}"] = (round(best, 3), round(step, 3))
print(json.dumps(out))
''' % str(ROOT)
libs = sys.argv[1:3]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        p = subprocess.run([sys.executable, "-c", CHILD, l], capture_output=True, text=True, cwd=ROOT)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(p.stderr[-2000:]); sys.exit(1)
        res[l].append(json.loads(line[-1]))
for shape in res[libs[0]][0]:
    print(shape, " | ".join(f"{Path(l).name}: fused {min(r[shape][0] for r in res[l]):.3f} ms, step {min(r[shape][1] for r in res[l]):.3f} ms" for l in libs), flush=True)
