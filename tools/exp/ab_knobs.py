#!/usr/bin/env python3
"""Same-box A/B of partition knob sets on the usual shapes (fused-kernel time, best of 3)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

sets = [("default", {}), ("seed6144", dict(seed_rows=6144)), ("seed6144 cell2048", dict(seed_rows=6144, cell_rows=2048)),
        ("seed6144 cell1024", dict(seed_rows=6144, cell_rows=1024)), ("seed9216", dict(seed_rows=9216)),
        ("seed6144 home16384", dict(seed_rows=6144, home_rows=16384))]
for name, n in (("c2", None), ("c4", None), ("c3", 125000), ("c3", 250000), ("c3", None), ("c5shard", None), ("cca30", None)):
    U, V, _, _, d = bench.make_latents(name, n, None)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    out = []
    for label, kw in sets:
        _hip.set_tuning(**kw)
        best = 1e9
        for _ in range(3 if q.shape[0] * r.shape[0] < 5e11 else 2):
            _hip.set_timing(True)
            _hip.knn_search(q, r, 30, exact_fallback=False)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
            best = min(best, ms)
        out.append(f"{label}: {best:.3f} ms {flops / best / 1e9 / 157.3 * 100:.1f}%")
    print(f"{name} n={q.shape[0]}: " + " | ".join(out), flush=True)
_hip.set_tuning()
