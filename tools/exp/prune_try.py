import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from scmiac_amd import _hip
for wl, n, m in (("c2", None, None), ("c4", None, None), ("c2", 30000, 500000), ("c3", 250000, None), ("c3", None, None), ("cca30", None, None), ("c5shard", None, None)):
    U, V, _, _, d = bench.make_latents(wl, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    res = {}
    for name, mode in (("dense", "dense"), ("prune", "pruned")):
        st = _hip.knn_search(q, r, 30, mode=mode); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            st = _hip.knn_search(q, r, 30, mode=mode)
        torch.cuda.synchronize()
        res[name] = ((time.perf_counter() - t0) / 3 * 1e3, st.idx.clone(), st.dist2.clone(), int(st.n_uncertified.item()))
    _hip.set_tuning()
    same = torch.equal(res["dense"][1], res["prune"][1]) and torch.equal(res["dense"][2], res["prune"][2])
    print(f"{wl}: dense(diag=0) {res['dense'][0]:.2f} ms  prune {res['prune'][0]:.2f} ms  identical {same}  uncertified {res['dense'][3]}/{res['prune'][3]}", flush=True)
