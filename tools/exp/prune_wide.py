"""Pruned against dense search on the wide-row workloads (d = 100 ... 256), and what mode="auto" does.
Run on the GPU box: python tools/exp/prune_wide.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
import bench
from scmiac_amd import _hip


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for wl in ("wide", "wide128", "wide200", "wide256"):
    U, V, _, _, d = bench.make_latents(wl, None, None)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    n, m = q.shape[0], r.shape[0]
    t_dense, sd = timed(lambda: _hip.knn_search(q, r, 30, mode="dense"))
    t_pruned, sp = timed(lambda: _hip.knn_search(q, r, 30, mode="pruned"))
    kept, dense = _hip.knn_prune_stats(n, m, d, 30)
    t_auto, sa = timed(lambda: _hip.knn_search(q, r, 30, mode="auto"))
    same = torch.equal(sp.idx, sd.idx) and torch.equal(sp.dist2, sd.dist2) and torch.equal(sa.idx, sd.idx)
    print(f"{wl} {n}x{m} d={d}: dense {t_dense:.2f} ms, pruned {t_pruned:.2f} ms (kept {kept / max(dense, 1):.4f}, "
          f"{_hip.knn_kernel_name(n, m, d, 30, mode='pruned')}), auto {t_auto:.2f} ms -> {_hip.choose_mode(q, r, 30)}; "
          f"identical {same}; uncertified {int(sd.n_uncertified.item())}/{int(sp.n_uncertified.item())}", flush=True)
