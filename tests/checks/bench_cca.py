"""Times the device CCA (matrix-free randomized SVD) against the CPU restatement that
forms the dense cell x cell product like the reference (anchors.py:171-242).

    python tests/checks/bench_cca.py [--n1 10412 --n2 10412 --genes 3000 --components 30]
                              [--max-cc-cell 50000] [--no-cpu]

Synthetic log-normalised counts; prints one JSON line."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=10412)
    ap.add_argument("--n2", type=int, default=10412)
    ap.add_argument("--genes", type=int, default=3000)
    ap.add_argument("--components", type=int, default=30)
    ap.add_argument("--max-cc-cell", type=int, default=50000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()

    from scmiac_amd.cca_device import cca_device

    rng = np.random.default_rng(1)
    prof = rng.gamma(2.0, 1.0, size=(12, args.genes)).astype(np.float32)
    X1 = np.log1p(rng.poisson(prof[rng.integers(0, 12, args.n1)] * 0.5)).astype(np.float64)
    X2 = np.log1p(rng.poisson(prof[rng.integers(0, 12, args.n2)] * 0.3)).astype(np.float64)
    kw = dict(n_components=args.components, max_cc_cell=args.max_cc_cell)

    times = []
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            U, V, _ = cca_device(X1, X2, **kw)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        # device-only part: inputs already resident
        dev = torch.device("cuda")
        from scmiac_amd import cca_device as cd
        t1 = cd.DeviceScaler(torch.from_numpy(X1[:args.max_cc_cell]).to(dev))
        a = t1.transform(torch.from_numpy(X1[:args.max_cc_cell]).to(dev))
        t2 = cd.DeviceScaler(torch.from_numpy(X2[:args.max_cc_cell]).to(dev))
        b = t2.transform(torch.from_numpy(X2[:args.max_cc_cell]).to(dev))
        svd_t = []
        for _ in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cd.randomized_svd_of_product(a, b, args.components)
            torch.cuda.synchronize()
            svd_t.append(time.perf_counter() - t0)
    out = {"n1": args.n1, "n2": args.n2, "genes": args.genes, "components": args.components,
           "max_cc_cell": args.max_cc_cell,
           "device_cca_s_incl_upload": round(float(np.median(times[1:])), 4),
           "device_svd_s_resident": round(float(np.median(svd_t[1:])), 4)}
    if not args.no_cpu:
        from oracle import oracle
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            U0, V0, _ = oracle.cca(X1, X2, **kw)
        out["cpu_cca_s"] = round(time.perf_counter() - t0, 2)
        out["cpu_threads"] = len(os.sched_getaffinity(0))
        out["max_abs_diff_U"] = float(np.abs(U - U0).max())
        out["max_abs_diff_V"] = float(np.abs(V - V0).max())
        out["speedup"] = round(out["cpu_cca_s"] / out["device_cca_s_incl_upload"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
