"""mode="auto": the probe's estimate of the kept share against the real one, and what the probe costs,
per shape.  Run on the GPU box: python tools/exp/auto_mode.py"""
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


shapes = [("c2", None, None), ("c4", None, None), ("c2", 30000, 500000), ("c3", 250000, None), ("c3", None, None),
          ("cca30", None, None), ("c5shard", None, None), ("noise", 100000, 100000), ("noise", 200000, 500000)]
for wl, n, m in shapes:
    if wl == "noise":
        g = torch.Generator(device="cuda").manual_seed(3)
        q = torch.randn((n, 50), device="cuda", generator=g)
        r = torch.randn((m, 50), device="cuda", generator=g)
    else:
        U, V, _, _, d = bench.make_latents(wl, n, m)
        q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    n_, m_, d_ = q.shape[0], r.shape[0], q.shape[1]
    t_probe, est = timed(lambda: _hip.knn_kept_fraction(q, r, 30))
    t_dense, sd = timed(lambda: _hip.knn_search(q, r, 30, mode="dense"))
    t_pruned, sp = timed(lambda: _hip.knn_search(q, r, 30, mode="pruned"))
    kept, dense = _hip.knn_prune_stats(n_, m_, d_, 30)
    t_auto, sa = timed(lambda: _hip.knn_search(q, r, 30, mode="auto"))
    same = torch.equal(sa.idx, sd.idx) and torch.equal(sa.dist2, sd.dist2) and torch.equal(sp.idx, sd.idx)
    print(f"{wl} {n_}x{m_} d={d_}: probe says {est:.3f} in {t_probe:.2f} ms, real kept {kept / max(dense, 1):.3f}; "
          f"dense {t_dense:.2f} ms, pruned {t_pruned:.2f} ms, auto {t_auto:.2f} ms -> {_hip.choose_mode(q, r, 30)}; identical {same}",
          flush=True)
