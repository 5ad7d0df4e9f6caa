"""Every row of a bench.py workload's k-NN table against the CPU oracle (not a sample): the GPU
table is computed once, the oracle goes through the query rows in slices and the comparison is
bit for bit (indices and squared distances).  Stops after --budget seconds and says how far it got.

    python tests/checks/full_parity.py [--workload c3] [--budget 1000] [--out gpurun_out/full_parity_c3.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from oracle import oracle  # noqa: E402  (checker)
from scmiac_amd import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--budget", type=float, default=1000.0)
    ap.add_argument("--slice", type=int, default=20000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--out", default="")
    ap.add_argument("--prune", action="store_true", help="the pruned search (mode='pruned') instead of the dense one")
    a = ap.parse_args()
    t_start = time.time()
    U, V, _, _, d = bench.make_latents(a.workload)
    if a.prune:
        _hip.set_default_mode("pruned")
    q_dev, r_dev = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    st = _hip.knn_search(q_dev, r_dev, a.k)
    torch.cuda.synchronize()
    t0 = time.time()
    st = _hip.knn_search(q_dev, r_dev, a.k)
    torch.cuda.synchronize()
    print(f"GPU search: {(time.time() - t0) * 1e3:.1f} ms per table", flush=True)
    _hip.set_default_mode("dense")
    idx_g, d2_g = st.idx.cpu().numpy(), st.dist2.cpu().numpy()
    oracle.build()
    n = U.shape[0]
    # slices in a seeded random order: wherever the budget ends, the rows checked are a fair sample
    order = np.random.default_rng(7).permutation((n + a.slice - 1) // a.slice)
    checked = bad_idx = bad_d2 = 0
    for s in order:
        lo, hi = s * a.slice, min((s + 1) * a.slice, n)
        idx_o, d2_o = oracle.knn_dist2(U[lo:hi], V, a.k)
        bad_idx += int((idx_g[lo:hi] != idx_o).any(axis=1).sum())
        bad_d2 += int((d2_g[lo:hi] != d2_o).any(axis=1).sum())
        checked += hi - lo
        print(f"{checked} rows checked, {bad_idx} with index differences, {bad_d2} with distance differences "
              f"({time.time() - t_start:.0f} s)", flush=True)
        if time.time() - t_start > a.budget:
            break
    out = {"workload": a.workload, "n": int(n), "m": int(V.shape[0]), "d": int(d), "k": a.k, "rows_checked": int(checked),
           "rows_with_index_mismatch": bad_idx, "rows_with_dist2_mismatch": bad_d2,
           "uncertified_rows_sent_to_direct_evaluation": int(st.n_uncertified.item()),
           "all_rows": bool(checked == n), "seconds": round(time.time() - t_start, 1),
           "oracle_threads": oracle.host_threads()}
    print(json.dumps(out))
    if a.out:
        Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
    sys.exit(1 if (bad_idx or bad_d2) else 0)


if __name__ == "__main__":
    main()
