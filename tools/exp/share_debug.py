#!/usr/bin/env python3
"""One slow stream among fast ones (the GPU test's construction): kernel time and helper counters
with sharing off and on."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402

rng = np.random.default_rng(41)
d, m = 50, 60000
R = rng.normal(size=(m, d)).astype(np.float32)
R *= np.linspace(6.0, 0.5, m, dtype=np.float32)[:, None]
Q = (R[rng.integers(0, 2000, 8 * 384)] + 0.01 * rng.normal(size=(8 * 384, d))).astype(np.float32)
Q[:384] = 1e-3 * rng.normal(size=(384, d)).astype(np.float32)
q, r = torch.from_numpy(Q).cuda(), torch.from_numpy(R).cuda()
for label, kw in (("off", dict(lists=-1)), ("2/4/4", dict(lists=2, run=4, min_left=4)), ("4/2/1", dict(lists=4, run=2, min_left=1))):
    _hip.set_tuning(slots=8, tail_min_rows=1024, no_locality=True)
    _hip.set_sharing(**kw)
    for _ in range(2):
        _hip.set_timing(True)
        st = _hip.knn_search(q, r, 30, splits=1, exact_fallback=False)
        ms, _ = _hip.fused_ms()
        _hip.set_timing(False)
    print(label, f"{ms:.3f} ms", _hip.knn_kernel_name(len(Q), m, d, 30, splits=1), _hip.knn_grid_plan(len(Q), m, d, 30, splits=1),
          _hip.knn_share_info(len(Q), m, d, 30, splits=1, read_stats=True), flush=True)
_hip.set_tuning(); _hip.set_sharing()
