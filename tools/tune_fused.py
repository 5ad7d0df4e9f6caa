#!/usr/bin/env python3
"""Tuning aid: time the fused distance + top-k kernel for several (waves per block x
stage rows) variants on one GPU, interleaved in one process.

    python tools/tune_fused.py --n 262144 --m 1000000 --variants 8x128 4x64 --rounds 3
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--m", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--variants", nargs="+", default=["default"], help='"WAVESxSTAGE_ROWSxPENDINGxGLOBAL" or "default"')
    ap.add_argument("--workload", default="c3", help="bench.py workload the latents are drawn from")
    args = ap.parse_args()
    U, V, _, _, d = bench.make_latents(args.workload, args.n, args.m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    res = {v: [] for v in args.variants}
    ref_idx = None
    # the library reads no environment: this tool turns its own variables into tuning calls
    ablate = int(os.environ.get("SCMIAC_FUSED_ABLATE", "0"))
    diag = os.environ.get("SCMIAC_FUSED_DIAG") == "1"
    no_tail = os.environ.get("SCMIAC_FUSED_NO_TAIL") == "1"
    no_locality = os.environ.get("SCMIAC_NO_LOCALITY") == "1"
    for rnd in range(args.rounds + 1):
        for v in args.variants:
            _hip.set_tuning(variant=None if v == "default" else v, ablate=ablate, diag=diag, no_tail=no_tail,
                            no_locality=no_locality)
            _hip.set_timing(True)
            # ablated kernels leave poor shortlists: skip the direct-evaluation pass they would trigger
            ablated = ablate != 0
            st = _hip.knn_search(q, r, args.k, splits=args.splits, exact_fallback=not ablated)
            ms, cnt = _hip.fused_ms()
            _hip.set_timing(False)
            if ablated:
                pass
            elif ref_idx is None:
                ref_idx = st.idx.clone()
            elif not torch.equal(ref_idx, st.idx):
                print(f"variant {v}: RESULT DIFFERS from the first variant", flush=True)
            if rnd > 0:
                res[v].append(ms)
    if diag:
        dg = _hip.knn_diag(args.n, args.m, d, args.k, args.splits).astype(np.float64)
        tot = dg[:, 0]
        print(f"diag ({args.variants[-1]}): waves {len(dg)}  wave cycles mean {tot.mean():.3e}  "
              f"candidate path {100 * dg[:, 1].sum() / tot.sum():.2f} %  of which merges {100 * dg[:, 2].sum() / tot.sum():.2f} %  "
              f"entries/wave {dg[:, 3].mean():.0f}  merges/wave {dg[:, 4].mean():.0f}  candidates/wave {dg[:, 5].mean():.0f}  "
              f"cycles/entry {dg[:, 1].sum() / max(dg[:, 3].sum(), 1):.0f}  cycles/merge {dg[:, 2].sum() / max(dg[:, 4].sum(), 1):.0f}  "
              f"register steps/wave {dg[:, 6].mean():.0f}  cycles to duplicate check/entry {dg[:, 7].sum() / max(dg[:, 3].sum(), 1):.0f}  "
              f"one-candidate store cycles/entry {dg[:, 8].sum() / max((dg[:, 3] - dg[:, 9]).sum(), 1):.0f}  "
              f"general-path entries/wave {dg[:, 9].mean():.0f}")
    flops = 2.0 * args.n * args.m * d
    for v in args.variants:
        ms = np.array(res[v])
        print(f"{v:>8}: median {np.median(ms):9.3f} ms  min {ms.min():9.3f} ms  "
              f"{flops / (np.median(ms) * 1e-3) / 1e12:7.2f} TFLOP/s  ({flops / (np.median(ms) * 1e-3) / 157.3e12 * 100:5.1f} % of fp32 MFMA peak)", flush=True)


if __name__ == "__main__":
    main()
