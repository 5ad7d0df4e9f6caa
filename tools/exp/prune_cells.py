#!/usr/bin/env python3
"""Pruned search against the number of pivot cells (whole search, best of 3; kept = share of the
dense stream's stages that were still scored)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

for wl, n, m in (("c4", None, None), ("c2", None, None), ("c3", 250000, None), ("cca30", 250000, None)):
    U, V, _, _, d = bench.make_latents(wl, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    out = []
    for mode, cell_rows in (("dense", 0), ("pruned", 0), ("pruned", 768), ("pruned", 512), ("pruned", 384), ("pruned", 256), ("pruned", 128)):
        _hip.set_tuning(cell_rows=cell_rows)
        if mode == "pruned" and not _hip.knn_grid_plan(q.shape[0], r.shape[0], d, 30, mode=mode)["prunes"]:
            continue
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = _hip.knn_search(q, r, 30, mode=mode)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        cells = _hip.knn_grid_plan(q.shape[0], r.shape[0], d, 30, mode=mode)["cells"]
        kept = ""
        if mode == "pruned":
            k_, d_ = _hip.knn_prune_stats(q.shape[0], r.shape[0], d, 30)
            kept = f" kept {k_ / max(d_, 1):.3f}"
        out.append(f"{mode} cells {cells}: {best:.2f} ms{kept}")
    print(f"{wl} n={q.shape[0]} m={r.shape[0]}: " + " | ".join(out), flush=True)
_hip.set_tuning()
