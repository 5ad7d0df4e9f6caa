"""``scmiac_find_mnn_noalloc`` -- the entry point for callers that cannot allocate -- against
``scmiac_find_mnn`` through the C ABI: two calls in a row on one GPU, the second on a smaller
problem (the static scratch keeps the first call's contents), then the size limit."""
import ctypes

import numpy as np
import pytest
import torch

from scmiac_amd import _hip

pytestmark = pytest.mark.gpu

MAX_ITEMS = 8 * 1024 * 1024  # SCMIAC_MNN_NOALLOC_MAX_ITEMS in include/scmiac_hip.h


def _tables(seed, n1, n2, k, ka):
    rng = np.random.default_rng(seed)
    g12 = rng.integers(0, n2, size=(n1, k), dtype=np.int32)
    g21 = rng.integers(0, n1, size=(n2, k), dtype=np.int32)
    rows = rng.integers(0, n1, n1 // 2)               # planted mutual pairs: the result is not tiny
    g21[g12[rows, ka // 2], ka - 1] = rows
    return torch.from_numpy(g12).cuda(), torch.from_numpy(g21).cuda()


def _call_noalloc(g12, g21, ka):
    lib = _hip.require_gpu()
    n1, ld12 = g12.shape
    n2, ld21 = g21.shape
    pairs = torch.full((max(n1 * ka, 1), 2), -7, dtype=torch.int64, device="cuda")
    n_pairs = torch.full((1,), -7, dtype=torch.int64, device="cuda")
    status = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    before = torch.cuda.memory_allocated()
    rc = lib.scmiac_find_mnn_noalloc(_hip._ptr(g12), n1, ld12, _hip._ptr(g21), n2, ld21, ka, _hip._ptr(pairs),
                                     _hip._ptr(n_pairs), _hip._ptr(status), _hip._stream())
    assert torch.cuda.memory_allocated() == before
    return rc, pairs, n_pairs, status


def test_noalloc_find_mnn_matches_the_workspace_entry_point_twice_in_a_row():
    ka = 5
    for seed, (n1, n2, k) in enumerate([(300_000, 250_000, 30), (4_000, 5_000, 12), (1, 3, 5)]):
        g12, g21 = _tables(seed, n1, n2, k, ka)
        want_pairs, want_n, want_status = _hip.find_mnn(g12, g21, ka)      # scmiac_find_mnn, caller workspace
        rc, pairs, n_pairs, status = _call_noalloc(g12, g21, ka)
        torch.cuda.synchronize()
        assert rc == 0, _hip.load_library().scmiac_last_error()
        n = int(want_n.item())
        assert int(n_pairs.item()) == n and int(status.item()) == int(want_status.item()) == 0
        assert n > 0 or n1 == 1
        assert torch.equal(pairs[:n], want_pairs[:n])                        # same pairs, same (i, j) order
        assert bool((pairs[n:] == -7).all())                                # nothing written past the count


def test_noalloc_find_mnn_rejects_what_the_static_scratch_cannot_hold():
    lib = _hip.require_gpu()
    one = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    g = torch.zeros((2, 5), dtype=torch.int32, device="cuda")
    n1 = MAX_ITEMS // 5 + 1
    rc = lib.scmiac_find_mnn_noalloc(_hip._ptr(g), n1, 5, _hip._ptr(g), 2, 5, 5, _hip._ptr(one), _hip._ptr(one),
                                     _hip._ptr(st), _hip._stream())
    assert rc == 1 and b"static scratch" in lib.scmiac_last_error()
    torch.cuda.synchronize()
    # out-of-range neighbour entries are reported through `status`, as by scmiac_find_mnn
    bad = torch.full((4, 5), 99, dtype=torch.int32, device="cuda")
    rc, _, n_pairs, status = _call_noalloc(bad, g, 5)
    torch.cuda.synchronize()
    assert rc == 0 and int(status.item()) == 1
