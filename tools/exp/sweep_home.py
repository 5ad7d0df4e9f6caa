#!/usr/bin/env python3
"""Fused-kernel time over the partition's knobs (home window, seed rows, cell size) on a few shapes."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

shapes = [("c2", None), ("c4", None), ("c3", 125000)] + ([("c3", None)] if "--c3" in sys.argv else [])
for name, n in shapes:
    U, V, _, _, d = bench.make_latents(name, n, None)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    for home in (8192, 16384, 32768):
        for seed in (3072, 6144):
            for cell in (512, 1024, 2048):
                _hip.set_tuning(home_rows=home, seed_rows=seed, cell_rows=cell)
                best = 1e9
                for _ in range(3):
                    _hip.set_timing(True)
                    _hip.knn_search(q, r, 30, exact_fallback=False)
                    ms, _ = _hip.fused_ms()
                    _hip.set_timing(False)
                    best = min(best, ms)
                print(f"{name} n={q.shape[0]} home={home} seed={seed} cell={cell}: {best:.3f} ms  {flops / best / 1e9 / 157.3 * 100:.1f} %", flush=True)
_hip.set_tuning()
