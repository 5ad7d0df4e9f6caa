#!/usr/bin/env python3
"""Where a block of the fused kernel spends its cycles: seed pass, home stages, the rest of the
stream, closing merges -- from the diagnostic instantiation's per-wave timestamps, for the main
region and the tail region of the grid apart.

    python tools/exp/phase_times.py [workload[:n[:pruned]] ...]      (default: c2 c4 c3:125000)
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402


def report(name, n_override, tuning=None, mode="dense"):
    U, V, _, _, d = bench.make_latents(name, n_override, None)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    n, m, k = q.shape[0], r.shape[0], 30
    tuning = dict(tuning or {})
    _hip.set_tuning(**tuning)
    for _ in range(2):
        _hip.set_timing(True)
        _hip.knn_search(q, r, k, mode=mode)
        ms_prod, _ = _hip.fused_ms()
        _hip.set_timing(False)
    _hip.set_tuning(diag=True, **tuning)
    _hip.set_timing(True)
    _hip.knn_search(q, r, k, mode=mode)
    ms_diag, _ = _hip.fused_ms()
    _hip.set_timing(False)
    plan = _hip.knn_grid_plan(n, m, d, k, mode=mode)
    dg = _hip.knn_diag(n, m, d, k, mode=mode).astype(np.float64)
    if plan["prunes"]:
        print("   pruned search: stages streamed / dense", _hip.knn_prune_stats(n, m, d, k))
    _hip.set_tuning()
    nw = plan["queries_per_block"] // 32
    blocks = dg.reshape(-1, nw, 16)
    block0 = plan["main_query_blocks"] * plan["S"]
    flops = 2.0 * n * m * d
    print(f"== {name} n={n} m={m} d={d} {mode} {tuning or ''}: product kernel {ms_prod:.3f} ms = {flops / ms_prod / 1e9 / 157.3 * 100:.1f} % of peak, "
          f"diagnostic kernel {ms_diag:.3f} ms; plan {plan}")
    for label, b in (("main", blocks[:block0]), ("tail", blocks[block0:])):
        if len(b) == 0:
            continue
        tot = b[:, :, 0].max(axis=1)                 # a block is done when its slowest wave is
        seed = b[:, :, 10].mean(axis=1)
        home = b[:, :, 11].mean(axis=1) - seed
        rest = b[:, :, 12].mean(axis=1) - b[:, :, 11].mean(axis=1)
        close = b[:, :, 0].mean(axis=1) - b[:, :, 12].mean(axis=1)
        T, oN, SN = b[:, 0, 13], b[:, 0, 14], b[:, 0, 15]
        rest_stages = np.maximum(T - oN, 1)
        print(f"  {label}: {len(b)} blocks; block Mcycles min/median/max {tot.min() / 1e6:.2f}/{np.median(tot) / 1e6:.2f}/{tot.max() / 1e6:.2f}; "
              f"stages {np.median(T):.0f} (home {np.median(oN):.0f}, min {oN.min():.0f}, max {oN.max():.0f}; seed {np.median(SN):.0f})")
        print(f"        mean Mcycles: seed {seed.mean() / 1e6:.3f}  home {home.mean() / 1e6:.3f}  rest {rest.mean() / 1e6:.3f}  closing {close.mean() / 1e6:.3f}; "
              f"per stage: seed {seed.sum() / max(SN.sum(), 1) / 1e3:.1f}k  home {home.sum() / max(oN.sum(), 1) / 1e3:.1f}k  rest {rest.sum() / rest_stages.sum() / 1e3:.1f}k")
        slow = np.argsort(-tot)[:4]
        fast = np.argsort(tot)[:2]
        for tag, ids in (("slowest", slow), ("fastest", fast)):
            for i in ids:
                print(f"        {tag} block {i}: {tot[i] / 1e6:.2f} M; seed {seed[i] / 1e6:.2f} home {home[i] / 1e6:.2f} rest {rest[i] / 1e6:.2f} "
                      f"closing {close[i] / 1e6:.2f}; per wave entries {b[i, :, 3].mean():.0f} merges {b[i, :, 4].mean():.0f} "
                      f"candidates {b[i, :, 5].mean():.0f}; stages streamed {T[i]:.0f} (home {oN[i]:.0f}, seed {SN[i]:.0f}); "
                      f"candidates per wave min/max {b[i, :, 5].min():.0f}/{b[i, :, 5].max():.0f}")
        w = b.reshape(-1, 16)
        print(f"        per wave: entries {w[:, 3].mean():.0f}  merges {w[:, 4].mean():.0f}  candidates {w[:, 5].mean():.0f}  general-path entries {w[:, 9].mean():.0f}  "
              f"candidate-path cycles {w[:, 1].mean() / 1e3:.0f}k of which merges {w[:, 2].mean() / 1e3:.0f}k")


if __name__ == "__main__":
    for spec in (sys.argv[1:] or ["c2", "c4", "c3:125000"]):
        name, n, mode = (spec.split(":") + ["", ""])[:3]
        report(name, int(n) if n else None, mode=mode or "dense")
