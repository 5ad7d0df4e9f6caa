#!/usr/bin/env python3
"""A/B: fused kernel time and whole-search time with and without the locality partition."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

CASES = [("c2", None, None), ("c3", 125_000, None), ("c3", 250_000, None), ("c3", None, None), ("c5shard", None, 250_000),
         ("c5shard", None, None)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]
for wl, n, m in CASES:
    U, V, _, _, d = bench.make_latents(wl, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    ref = None
    for name, kw in (("caller order", dict(no_locality=True)), ("locality", dict())):
        _hip.set_tuning(**kw)
        best, best_all = 1e9, 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _hip.set_timing(True)
            st = _hip.knn_search(q, r, 30)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
            torch.cuda.synchronize()
            best, best_all = min(best, ms), min(best_all, (time.perf_counter() - t0) * 1e3)
        same = "" if ref is None else f" same table: {bool(torch.equal(ref.idx, st.idx) and torch.equal(ref.dist2, st.dist2))}"
        ref = ref or st
        print(f"{wl} {q.shape[0]}x{r.shape[0]} d={d}: {name:13s} fused {best:9.3f} ms = {flops / best / 1e9 / 157.3 * 100:5.1f} % of peak;"
              f" whole search {best_all:9.3f} ms = {flops / best_all / 1e9 / 157.3 * 100:5.1f} %; uncertified {int(st.n_uncertified.item())}{same}",
              flush=True)
    _hip.set_tuning()
