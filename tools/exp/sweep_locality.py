#!/usr/bin/env python3
"""Sweep of the locality knobs (cell size, home rows, seed rows): whole-search time."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

CELLS = [int(v) for v in sys.argv[1].split(",")]
HOMES = [int(v) for v in sys.argv[2].split(",")]
SEEDS = [int(v) for v in sys.argv[3].split(",")]
for wl, n, m in (("c2", None, None), ("c3", 250_000, None), ("c5shard", None, 250_000)):
    U, V, _, _, d = bench.make_latents(wl, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    for cell in CELLS:
        for home in HOMES:
            for seed in SEEDS:
                _hip.set_tuning(cell_rows=cell, home_rows=home, seed_rows=seed)
                best, best_all = 1e9, 1e9
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _hip.set_timing(True)
                    st = _hip.knn_search(q, r, 30)
                    ms, _ = _hip.fused_ms()
                    _hip.set_timing(False)
                    torch.cuda.synchronize()
                    best, best_all = min(best, ms), min(best_all, (time.perf_counter() - t0) * 1e3)
                print(f"{wl} {q.shape[0]}x{r.shape[0]}: cell {cell:5d} home {home:5d} seed {seed:4d}: fused {flops / best / 1e9 / 157.3 * 100:5.1f} %  "
                      f"whole {flops / best_all / 1e9 / 157.3 * 100:5.1f} % ({best_all:8.3f} ms)", flush=True)
_hip.set_tuning()
