"""The pruned search's slowest block: fused-kernel time against the home window, on the shapes where one
block streams everything.  Run on the GPU box: python tools/exp/prune_straggler.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from scmiac_amd import _hip


def data(spec):
    if isinstance(spec, str):
        U, V, _, _, d = bench.make_latents(spec, None, None)
        return torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    d = spec
    g = torch.Generator(device="cuda").manual_seed(7000 + d)
    centres = 3.0 * torch.randn((100, d), device="cuda", generator=g)
    n, m = 200000, 500000
    q = centres[torch.randint(0, 100, (n,), device="cuda", generator=g)] + torch.randn((n, d), device="cuda", generator=g)
    r = centres[torch.randint(0, 100, (m,), device="cuda", generator=g)] + torch.randn((m, d), device="cuda", generator=g)
    return q, r


for spec in ("wide256", 72, 160):
    q, r = data(spec)
    n, m, d = q.shape[0], r.shape[0], q.shape[1]
    line = []
    for home in (0, 16384, 32768, 65536):
        _hip.set_tuning(**({"home_rows": home} if home else {}))
        _hip.knn_search(q, r, 30, mode="pruned"); torch.cuda.synchronize()
        _hip.set_timing(True)
        st = _hip.knn_search(q, r, 30, mode="pruned"); torch.cuda.synchronize()
        ms, _ = _hip.fused_ms()
        _hip.set_timing(False)
        kept, dense = _hip.knn_prune_stats(n, m, d, 30)
        _hip.set_tuning(diag=True, **({"home_rows": home} if home else {}))
        _hip.knn_search(q, r, 30, mode="pruned"); torch.cuda.synchronize()
        plan = _hip.knn_grid_plan(n, m, d, 30, mode="pruned")
        dg = _hip.knn_diag(n, m, d, 30, mode="pruned").astype(np.float64)
        nw = plan["queries_per_block"] // 32
        b = dg.reshape(-1, nw, 16)
        T = b[:, 0, 13]
        _hip.set_tuning()
        line.append(f"home {home or 'default'}: fused {ms:.1f} ms, kept {kept / dense:.4f}, stages per block median {np.median(T):.0f} max {T.max():.0f} "
                    f"(blocks above 4x median: {(T > 4 * np.median(T)).sum()})")
    print(f"{spec} d={d}: " + " | ".join(line), flush=True)
