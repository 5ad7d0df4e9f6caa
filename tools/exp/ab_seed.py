#!/usr/bin/env python3
"""Same-box sweep of the seed pass's length (rows of a block's own cells it looks at) on the
short-stream shapes (fused-kernel time, best of 4)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

for name, n in (("c2", None), ("c4", None), ("c3", 125000)):
    U, V, _, _, d = bench.make_latents(name, n, None)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    out = []
    for seed in (320, 960, 1920, 3200, 4480, 6144, 8000):
        for home in (8192, 4096):
            _hip.set_tuning(seed_rows=seed, home_rows=home)
            best = 1e9
            for _ in range(4):
                _hip.set_timing(True)
                _hip.knn_search(q, r, 30, exact_fallback=False)
                ms, _ = _hip.fused_ms()
                _hip.set_timing(False)
                best = min(best, ms)
            out.append(f"seed {seed} home {home}: {best:.3f} ms {flops / best / 1e9 / 157.3 * 100:.1f}%")
    print(f"{name} n={q.shape[0]}: " + " | ".join(out), flush=True)
_hip.set_tuning()
