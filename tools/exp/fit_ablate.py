#!/usr/bin/env python3
"""Block time of the fused kernel (one full round, n = 98304, splits=1) under the diagnostic
kernel's ablations, for two stream lengths: what is stream, what is fixed, what is selection."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

U, V, _, _, d = bench.make_latents("c3", 98304, 400_000)
q = torch.from_numpy(U).cuda()
for label, kw in (("product kernel", {}), ("diag kernel, nothing ablated", dict(diag=True)), ("no selection", dict(ablate=1)), ("no selection, no per-subtile test", dict(ablate=129)),
                  ("no selection, no test, no seed pass", dict(ablate=129 + 32)), ("no selection, no test, no barrier", dict(ablate=129 + 4)),
                  ("no selection, no test, no staging", dict(ablate=129 + 2)),
                  ("empty candidate path", dict(ablate=8)), ("candidate arithmetic, no stores", dict(ablate=16)),
                  ("caller order, product", dict(no_locality=True))):
    _hip.set_tuning(**kw)
    out = []
    for m in (100_000, 400_000):
        r = torch.from_numpy(V[:m]).cuda()
        best = 1e9
        for _ in range(3):
            _hip.set_timing(True)
            _hip.knn_search(q, r, 30, splits=1, exact_fallback=False)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
            best = min(best, ms)
        out.append(best * 2.4)
    slope = (out[1] - out[0]) / 300_000 * 1e6
    print(f"{label:34s}: 100k rows {out[0]:7.3f} Mcycles, 400k rows {out[1]:7.3f} Mcycles; {slope * 32:6.0f} cycles per subtile, "
          f"fixed {out[0] - slope * 0.1:6.3f} Mcycles", flush=True)
_hip.set_tuning()
