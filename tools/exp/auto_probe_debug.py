import sys
sys.path.insert(0, '/root/repo')
import torch
import bench
from scmiac_amd import _hip

U, V, _, _, d = bench.make_latents("c3", 250000, None)
q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
n, m = q.shape[0], r.shape[0]
groups, group_rows = 8, 384
centres = q[torch.arange(groups, device=q.device) * (n // groups) + n // (2 * groups)]
score = (q * q).sum(1)[None, :] - 2.0 * (centres @ q.T)
pool = score[:, ::max(1, n // 65536)]
rank = max(1, min(pool.shape[1], round(group_rows * pool.shape[1] / n)))
radius = torch.kthvalue(pool, rank, dim=1).values
member = score <= radius[:, None]
print("pool", pool.shape, "rank", rank, "members per centre", member.sum(1).tolist())
tight = q[member.any(0)].contiguous()
top = torch.topk(score, group_rows, dim=1, largest=False, sorted=False).indices.reshape(-1)
exact = q[torch.unique(top)].contiguous()
for name, probe in (("radius", tight), ("topk", exact)):
    for splits in (0, 1, 4, 32):
        _hip.knn_search(probe, r, 30, mode="pruned", exact_fallback=False, splits=splits)
        kept, dense = _hip.knn_prune_stats(probe.shape[0], m, d, 30, splits=splits)
        print(name, probe.shape[0], "splits", splits, "kept", kept, "dense", dense, _hip.knn_grid_plan(probe.shape[0], m, d, 30, splits=splits, mode="pruned"), flush=True)
