#!/usr/bin/env python3
"""Same-box A/B of stream sharing (fused-kernel time, best of 5): off, the library's choice and
other (helper lists, stages per run, stages left) settings, on single-round shapes."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

sets = [("off", dict(lists=-1)), ("default", {}), ("2/4/8", dict(lists=2, run=4, min_left=8)),
        ("3/8/12", dict(lists=3, run=8, min_left=12)), ("2/16/24", dict(lists=2, run=16, min_left=24)),
        ("1/8/12", dict(lists=1, run=8, min_left=12)), ("4/4/6", dict(lists=4, run=4, min_left=6))]
shapes = [("c2", None, None), ("c4", None, None), ("c2", 40000, 60000), ("c2", 50000, None), ("c3", 60000, 500000)]
if len(sys.argv) > 1 and sys.argv[1] == "shards":
    shapes = [("c3", 125000, None), ("c3", 250000, None), ("c5shard", 125000, None)]
    sets = [("off", dict(lists=-1)), ("2/4/8", dict(lists=2, run=4, min_left=8)), ("2/8/16", dict(lists=2, run=8, min_left=16)),
            ("3/16/32", dict(lists=3, run=16, min_left=32))]
for name, n, m in shapes:
    U, V, _, _, d = bench.make_latents(name, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    out, ref = [], None
    for label, kw in sets:
        _hip.set_sharing(**kw)
        best = 1e9
        for _ in range(5 if q.shape[0] * r.shape[0] < 5e10 else 2):
            _hip.set_timing(True)
            st = _hip.knn_search(q, r, 30, exact_fallback=False)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
            best = min(best, ms)
        info = _hip.knn_share_info(q.shape[0], r.shape[0], d, 30, read_stats=True)
        same = ""
        if ref is None:
            ref = (st.idx.clone(), st.dist2.clone())
        else:
            same = " same" if torch.equal(ref[0], st.idx) and torch.equal(ref[1], st.dist2) else " DIFFERENT"
        out.append(f"{label}: {best:.3f} ms {flops / best / 1e9 / 157.3 * 100:.1f}% joined {info.get('joined', 0)} "
                   f"stages {info.get('helper_stages', 0)}{same}")
    print(f"{name} n={q.shape[0]} m={r.shape[0]}: " + " | ".join(out), flush=True)
_hip.set_sharing()
