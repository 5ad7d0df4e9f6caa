#!/usr/bin/env python3
"""Fit the fused kernel's block time: one full round of 256 blocks (n = 98304), every block
streaming all m rows (splits=1), for several m and with S chunks per block."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

U, V, _, _, d = bench.make_latents("c3", 98304 * 2, 400_000)
for n in (98304,):
    q = torch.from_numpy(U[:n]).cuda()
    for m in (16384, 32768, 65536, 100_000, 200_000, 400_000):
        r = torch.from_numpy(V[:m]).cuda()
        for splits in (1, 2, 4):
            best = 1e9
            for _ in range(4):
                _hip.set_timing(True)
                st = _hip.knn_search(q, r, 30, splits=splits)
                ms, _ = _hip.fused_ms()
                _hip.set_timing(False)
                best = min(best, ms)
            rounds = splits * 1.0
            print(f"n={n} m={m} splits={splits}: {best:8.3f} ms  = {best * 2.4e6 / rounds / 1e6:7.3f} Mcycles per block of {m // splits} rows; "
                  f"{2.0 * n * m * d / best / 1e9 / 157.3 * 100:5.1f} % of peak", flush=True)
