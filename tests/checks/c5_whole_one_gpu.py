"""The whole 4M x 2M, d=64 cosine atlas (config 5) on one GPU: dense and pruned search, a row sample
against the CPU oracle.  Prints one JSON line."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
import bench
from scmiac_amd import _hip
U, V, _, _, d = bench.make_latents("c5")
q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
out = {"workload": "c5 whole on one GPU", "n": U.shape[0], "m": V.shape[0], "d": d}
st = _hip.knn_search(q, r, 30, mode="pruned"); torch.cuda.synchronize()
t0 = time.perf_counter(); st = _hip.knn_search(q, r, 30, mode="pruned"); torch.cuda.synchronize()
out["pruned_ms_per_table"] = round((time.perf_counter() - t0) * 1e3, 1)
kept, dense = _hip.knn_prune_stats(U.shape[0], V.shape[0], d, 30)
out["kept_stage_fraction"] = round(kept / dense, 4)
_hip.set_timing(True)
t0 = time.perf_counter(); sd = _hip.knn_search(q, r, 30); torch.cuda.synchronize()
out["dense_ms_per_table"] = round((time.perf_counter() - t0) * 1e3, 1)
ms, _ = _hip.fused_ms(); _hip.set_timing(False)
out["dense_fused_kernel_ms"] = round(ms, 1)
out["dense_frac_of_fp32_mfma_peak"] = round(2.0 * U.shape[0] * V.shape[0] * d / (ms * 1e-3) / 157.3e12, 4)
out["pruned_identical_to_dense"] = bool(torch.equal(st.idx, sd.idx) and torch.equal(st.dist2, sd.dist2))
out["uncertified_rows_dense"] = int(sd.n_uncertified.item())
from oracle import oracle
rows = np.sort(np.random.default_rng(77).choice(U.shape[0], 1000, replace=False))
idx_o, d2_o = oracle.knn_dist2(U[rows], V, 30)
rt = torch.from_numpy(rows).cuda()
out["rows_vs_oracle"] = len(rows)
out["index_mismatches"] = int((sd.idx[rt].cpu().numpy() != idx_o).any(axis=1).sum())
out["dist2_bit_identical"] = bool((sd.dist2[rt].cpu().numpy() == d2_o).all())
print(json.dumps(out))
