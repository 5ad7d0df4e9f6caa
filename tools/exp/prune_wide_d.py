"""Pruned against dense search at widths between the wide workloads of bench.py (mixture of 100 clusters,
200 000 x 500 000).  Run on the GPU box: python tools/exp/prune_wide_d.py 160 192 224"""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from scmiac_amd import _hip


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for d in [int(a) for a in sys.argv[1:]] or [160, 192, 224]:
    g = torch.Generator(device="cuda").manual_seed(7000 + d)
    centres = 3.0 * torch.randn((100, d), device="cuda", generator=g)
    n, m = 200000, 500000
    q = centres[torch.randint(0, 100, (n,), device="cuda", generator=g)] + torch.randn((n, d), device="cuda", generator=g)
    r = centres[torch.randint(0, 100, (m,), device="cuda", generator=g)] + torch.randn((m, d), device="cuda", generator=g)
    t_dense, sd = timed(lambda: _hip.knn_search(q, r, 30, mode="dense"))
    t_pruned, sp = timed(lambda: _hip.knn_search(q, r, 30, mode="pruned"))
    kept, dense = _hip.knn_prune_stats(n, m, d, 30)
    same = torch.equal(sp.idx, sd.idx) and torch.equal(sp.dist2, sd.dist2)
    print(f"d={d}: dense {t_dense:.2f} ms, pruned {t_pruned:.2f} ms (kept {kept / max(dense, 1):.4f}, "
          f"{_hip.knn_kernel_name(n, m, d, 30, mode='pruned')}); identical {same}", flush=True)
