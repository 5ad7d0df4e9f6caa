"""Pending buffers of 64 entries per query (the library's choice) against 32, now that a merge keeps
the threshold a query had: fused-kernel time, best of 3, same box.  (Sharing is switched off for the
comparison: the stream-sharing kernels exist for the library's own variants only.)
Run on the GPU box: python tools/exp/ab_pcap.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

shapes = [("c2", None, None), ("c4", None, None), ("c3", 125000, None), ("c3", 250000, None), ("c3", None, None)]
for name, n, m in shapes:
    U, V, _, _, d = bench.make_latents(name, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    flops = 2.0 * q.shape[0] * r.shape[0] * d
    out, ref = [], None
    _hip.set_sharing(lists=-1)
    for label, variant in (("12x320x64", None), ("12x320x32", "12x320x32x1"), ("12x256x32", "12x256x32x1")):
        _hip.set_tuning(variant=variant)
        best = 1e9
        for _ in range(3):
            _hip.set_timing(True)
            st = _hip.knn_search(q, r, 30, exact_fallback=False)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
            best = min(best, ms)
        same = ""
        if ref is None:
            ref = (st.idx.clone(), st.dist2.clone())
        else:
            same = " same" if torch.equal(ref[0], st.idx) and torch.equal(ref[1], st.dist2) else " DIFFERENT"
        out.append(f"{label}: {best:.3f} ms {flops / best / 1e9 / 157.3 * 100:.1f}%{same}")
    _hip.set_tuning()
    _hip.set_sharing()
    print(f"{name} n={q.shape[0]} m={r.shape[0]}: " + " | ".join(out), flush=True)
