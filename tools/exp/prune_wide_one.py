import sys
sys.path.insert(0, '/root/repo')
import torch
import bench
from scmiac_amd import _hip
wl = sys.argv[1] if len(sys.argv) > 1 else "wide256"
U, V, _, _, d = bench.make_latents(wl, None, None)
q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
for _ in range(3):
    _hip.knn_search(q, r, 30, mode="pruned")
torch.cuda.synchronize()
