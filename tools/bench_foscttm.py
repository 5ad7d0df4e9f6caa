"""Times the closer-than-match count (FOSCTTM) on the GPU: one direction, inputs resident.

    python tools/bench_foscttm.py [--n 1000000 --d 30 --noise 1.5 --unit] [--dense-n 20000]

Prints one JSON line: ms per direction, TFLOP/s of the 2 n m d contraction, fraction of
the fp32 MFMA peak, mean FOSCTTM, and (for --dense-n rows) the time of the reference's
dense formulation (benchmark.py:99-106) in eager PyTorch on the same GPU."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scmiac_amd import _hip  # noqa: E402

PEAK_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--d", type=int, default=30)
    ap.add_argument("--noise", type=float, default=1.5)
    ap.add_argument("--unit", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dense-n", type=int, default=20000)
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(1)
    centres = torch.randn(100, args.d, device="cuda", generator=g) * 2
    lab = torch.randint(0, 100, (args.n,), device="cuda", generator=g)
    x = centres[lab] + torch.randn(args.n, args.d, device="cuda", generator=g)
    y = x + args.noise * torch.randn(args.n, args.d, device="cuda", generator=g)
    if args.unit:
        x = torch.nn.functional.normalize(x, dim=1)
        y = torch.nn.functional.normalize(y, dim=1)
    counts = _hip.closer_count(x, y)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        counts = _hip.closer_count(x, y)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) * 1e3
    tf = 2.0 * args.n * args.n * args.d / (ms * 1e-3) / 1e12
    out = {"n": args.n, "d": args.d, "unit_rows": args.unit, "ms_per_direction": round(ms, 2),
           "tflops": round(tf, 1), "frac_fp32_mfma_peak": round(tf / PEAK_TFLOPS, 3),
           "mean_foscttm": round(float(counts.double().mean().item()) / args.n, 5)}
    if args.dense_n:
        k = min(args.dense_n, args.n)
        xs, ys = x[:k], y[:k]

        def dense():
            d2 = (xs ** 2).sum(1)[:, None] + (ys ** 2).sum(1)[None, :] - 2 * (xs @ ys.t())
            dd = torch.sqrt(torch.clamp(d2, min=0.0))
            diag = torch.diag(dd)
            return (dd < diag[:, None]).float().mean(1), (dd < diag[None, :]).float().mean(0)
        dense()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dense()
        torch.cuda.synchronize()
        t_dense = time.perf_counter() - t0
        _hip.closer_count(xs, ys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _hip.closer_count(xs, ys)
        _hip.closer_count(ys, xs)
        torch.cuda.synchronize()
        out["dense_eager_both_directions_ms_at_%d" % k] = round(t_dense * 1e3, 2)
        out["fused_both_directions_ms_at_%d" % k] = round((time.perf_counter() - t0) * 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
