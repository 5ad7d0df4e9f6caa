"""What outlier queries cost the pruned search: C3's mixture at 250 000 x 1 000 000 with a share of the
queries moved far away from every cluster (their neighbours are then far in every direction, their
reach covers most cells, and their whole block streams what they need).
Run on the GPU box: python tools/exp/prune_outliers.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from scmiac_amd import _hip


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


U, V, _, _, d = bench.make_latents("c3", 250000, None)
r = torch.from_numpy(V).cuda()
n, m = U.shape[0], V.shape[0]
rng = np.random.default_rng(77)
for share in (0.0, 1e-5, 1e-4, 1e-3, 1e-2):
    Q = U.copy()
    k_out = int(round(share * n))
    if k_out:
        rows = rng.choice(n, k_out, replace=False)
        Q[rows] = (2.5 * Q[rows] + 4.0 * rng.normal(size=(k_out, d))).astype(np.float32)
    q = torch.from_numpy(Q).cuda()
    t_dense, sd = timed(lambda: _hip.knn_search(q, r, 30, mode="dense"))
    t_pruned, sp = timed(lambda: _hip.knn_search(q, r, 30, mode="pruned"))
    kept, dense = _hip.knn_prune_stats(n, m, d, 30)
    t_auto, sa = timed(lambda: _hip.knn_search(q, r, 30, mode="auto"))
    same = torch.equal(sp.idx, sd.idx) and torch.equal(sp.dist2, sd.dist2) and torch.equal(sa.idx, sd.idx)
    print(f"{k_out:5d} outlier queries of {n}: dense {t_dense:.1f} ms, pruned {t_pruned:.1f} ms (kept {kept / max(dense, 1):.4f}), "
          f"auto {t_auto:.1f} ms -> {sa.auto_mode}; identical {same}", flush=True)
