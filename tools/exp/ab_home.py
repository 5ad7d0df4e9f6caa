#!/usr/bin/env python3
"""Same-box A/B of the home window (8192 / 4096 rows) on short-stream shapes with the library as built."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402
for name, n, m in (("c2", None, None), ("c4", None, None), ("c2", 50000, None), ("c2", 40000, 60000), ("c3", 60000, 500000), ("c3", 125000, None), ("c2", 200000, 200000)):
    U, V, _, _, d = bench.make_latents(name, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    out = []
    for home in (8192, 4096, 6144):
        _hip.set_tuning(home_rows=home)
        best = 1e9
        for _ in range(5 if q.shape[0] * r.shape[0] < 5e10 else 3):
            _hip.set_timing(True)
            _hip.knn_search(q, r, 30, exact_fallback=False)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
            best = min(best, ms)
        out.append(f"home {home}: {best:.3f} ms {flops / best / 1e9 / 157.3 * 100:.1f}%")
    print(f"{name} n={q.shape[0]} m={r.shape[0]}: " + " | ".join(out), flush=True)
_hip.set_tuning()
