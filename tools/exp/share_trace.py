#!/usr/bin/env python3
"""Timeline of a stream-sharing launch: when the blocks finish their own streams, when they end,
how much they scored for others (from the trace the kernel leaves beside its counters)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from scmiac_amd import _hip  # noqa: E402
import bench  # noqa: E402

for name, n, m in (("c2", None, None), ("c4", None, None), ("c2", 50000, None), ("c3", 60000, 500000)):
    U, V, _, _, d = bench.make_latents(name, n, m)
    q, r = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    for label, kw in (("tail pieces first", dict(order=1)), ("tail pieces last", dict(order=2)), ("plain dense kernel", dict(lists=-1))):
        _hip.set_sharing(**kw)
        for _ in range(3):
            _hip.set_timing(True)
            _hip.knn_search(q, r, 30, exact_fallback=False)
            ms, _ = _hip.fused_ms()
            _hip.set_timing(False)
        info = _hip.knn_share_info(q.shape[0], r.shape[0], d, 30, read_stats=True)
        if info["lists"] == 0:
            print(f"== {name} n={q.shape[0]} {label}: kernel {ms:.3f} ms", flush=True)
            continue
        plan = _hip.knn_grid_plan(q.shape[0], r.shape[0], d, 30)
        tr = info["trace"].astype(np.float64)
        t0 = tr[:, 0].min()
        start, own, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0   # microseconds
        ntail = plan["grid_blocks"] - plan["main_query_blocks"] * plan["S"]
        tail_first = kw.get("order", 0) == 1
        if not tail_first:   # put the rows in tail-first order for the prints below
            tr = np.concatenate([tr[plan["grid_blocks"] - ntail:], tr[:plan["grid_blocks"] - ntail]])
            t0 = tr[:, 0].min()
            start, own, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0
        main = slice(ntail, None)
        pct = lambda a: " ".join(f"{np.percentile(a, p):7.0f}" for p in (0, 10, 50, 90, 99, 100))
        print(f"== {name} n={q.shape[0]} {label}: kernel {ms:.3f} ms, joined {info['joined']}, helper stages {info['helper_stages']}, tail blocks {ntail}")
        print(f"   main blocks: start   us (min p10 p50 p90 p99 max) {pct(start[main])}")
        print(f"                own end us                            {pct(own[main])}")
        print(f"                own dur us                            {pct(own[main] - start[main])}")
        print(f"                end     us                            {pct(end[main])}")
        if ntail:
            print(f"   tail blocks: dur us                                {pct(end[:ntail] - start[:ntail])}")
        if info["joined"]:
            order = np.argsort(-own)[:8]           # the blocks whose own streams ended last
            swz = lambda b: b                        # (stream numbers of main blocks are swizzled block numbers: matched through the helpers' records)
            first = tr[:, 5].astype(int) - 1
            for b in order:
                stream_guess = [int(x) for x in np.nonzero(first >= 0)[0] if False]
                print(f"     block {b:3d}: start {start[b]:6.0f} own end {own[b]:6.0f} end {end[b]:6.0f} us; scored {int(tr[b, 4])} stages of its own stream")
            # the queues as the launch left them (the control words lie just before the counters)
            up = lambda x: (x + 255) // 256 * 256
            S_ = info["streams"]
            off = _hip._share_stats_offset(q.shape[0], r.shape[0], d, 30)
            ws = _hip._ws_cache[torch.device("cuda", torch.cuda.current_device())]
            base = off - 3 * up(4 * S_)
            cur = ws[base:base + 4 * S_].view(torch.int32).cpu().numpy()
            pub = ws[base + up(4 * S_):base + up(4 * S_) + 4 * S_].view(torch.int32).cpu().numpy()
            cl = ws[base + 2 * up(4 * S_):base + 2 * up(4 * S_) + 4 * S_].view(torch.int32).cpu().numpy()
            nmain = plan["main_query_blocks"] * plan["S"]
            def stream_of(blk):
                if blk < ntail:
                    return nmain + blk
                b = blk - ntail
                return ((b & ~255) | ((b & 7) << 5) | ((b & 255) >> 3)) if (b | 255) < nmain else b
            print("   claims histogram over main streams:", np.bincount(cl[:nmain], minlength=4)[:6])
            for b in order:
                v_ = stream_of(int(b))
                helpers_of = [int(x) for x in np.nonzero(first == v_)[0]]
                print(f"     block {b:3d} = stream {v_:3d}: claims {cl[v_]} cursor {cur[v_]} pub {pub[v_]}; blocks whose FIRST join was this stream: {helpers_of}")
            late = {stream_of(b) for b in range(ntail, plan["grid_blocks"]) if start[b] > 100}
            ev = []
            for b in np.nonzero(tr[:, 7] > 0)[0]:
                for i in range(4):
                    if tr[b, 8 + i] > 0:
                        ev.append(((tr[b, 6] - t0) / 100.0, int(b), int(tr[b, 8 + i]) - 1, int(tr[b, 12 + i]), i))
            ev.sort()
            print("   joins (helper's first-join time us, helper block, stream, late-started stream?, stages scored, item):")
            for e in ev[:12] + ev[len(ev) // 2:len(ev) // 2 + 6] + ev[-8:]:
                print(f"      {e[0]:6.0f} block {e[1]:3d} -> stream {e[2]:3d} {'LATE' if e[2] in late else '    '} {e[3]:3d} stages (item {e[4]})")
            helped_late = sum(1 for e in ev if e[2] in late)
            print(f"   joins on late-started streams: {helped_late} of {len(ev)}; late-started streams: {len(late)}; of them with claims: {sum(1 for v_ in late if cl[v_] > 0)}")
            hs = (tr[:, 6] - t0) / 100.0
            hj = tr[:, 7] > 0
            print(f"   helpers: {int(hj.sum())} blocks joined something; first join at us {pct(hs[hj])}; stages scored for others per helper {pct(tr[hj, 3])}")
            # streams by how many stages their owners scored themselves
            print(f"   own streams: stages scored by the owner itself {pct(tr[main, 4])}")
        print(f"   sum of block busy time / 256 = {(end - start).sum() / 256:.0f} us; sum of own durations / 256 = {(own - start).sum() / 256:.0f} us", flush=True)
_hip.set_sharing()
