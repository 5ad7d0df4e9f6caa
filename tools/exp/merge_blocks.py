#!/usr/bin/env python3
"""Ring-block rehearsal on one GPU: 500k queries against 8 blocks of 250k unit-norm rows merged
one after the other; per-block fused-kernel time with the locality partition on every block, or
only on the first (later blocks start from the running k-th distance anyway)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

U, V, _, _, d = bench.make_latents("c5shard")
q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
n, m, per = q.shape[0], r.shape[0], r.shape[0] // 8
for mode in ("locality on every block", "locality on the first block only"):
    for rep in range(2):
        state, ms_all = None, []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(8):
            _hip.set_tuning(no_locality=(mode.endswith("only") and b > 0))
            _hip.set_timing(True)
            state = _hip.knn_search(q, r[b * per:(b + 1) * per], 30, col_offset=b * per, state=state, exact_fallback=False)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
            ms_all.append(ms)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
    _hip.set_tuning()
    fl = 2.0 * n * per * d
    print(f"{mode}: per block " + " ".join(f"{fl / x / 1e9 / 157.3 * 100:5.1f}" for x in ms_all) +
          f" % of peak; fused total {sum(ms_all):8.2f} ms, wall {wall:8.2f} ms; uncertified {int(state.n_uncertified.item())}", flush=True)
