#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats of one run; SQ / FETCH_SIZE / WRITE_SIZE counter
passes of three more) into the small files kept under profiles/.

    python tools/summarize_profiles.py --stats DIR --pmc DIR [DIR ...] --bench LOG --tag r01_c3
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--pmc", nargs="+", required=True)
    ap.add_argument("--bench", default=None, help="log of a bench.py run whose JSON line becomes <tag>_bench.json")
    ap.add_argument("--tag", default="r01_c3")
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--n-local", type=int, default=None, help="query rows per launch (default: the workload's, or --n)")
    ap.add_argument("--out-dir", default=str(ROOT / "profiles"))
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--m", type=int, default=None)
    a, _ = ap.parse_known_args()  # further bench.py options are passed through by the caller and ignored here
    out_dir = Path(a.out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    import sys
    sys.path.insert(0, str(ROOT))
    import bench
    if a.n_local is None:
        a.n_local = a.n or bench.WORKLOADS[a.workload][0]
    overridden = bool(a.n or a.m)
    def newest(pattern):  # a directory may hold the outputs of earlier runs as well
        return max(glob.glob(pattern), key=os.path.getmtime)

    shutil.copy(newest(f"{a.stats}/*/*_kernel_stats.csv"), out_dir / f"{a.tag}_kernel_stats.csv")
    out = {}
    for d in a.pmc:
        for r in csv.DictReader(open(newest(f"{d}/*/*_counter_collection.csv"))):
            kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "scmiac" in kn:
                out.setdefault(kn, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    out = {k: dict(sorted(v.items())) for k, v in sorted(out.items())}
    fk = next(k for k in out if "fused" in k)
    v = out[fk]
    cyc = v["GRBM_GUI_ACTIVE"] / 8  # the counter is summed over the 8 XCDs
    hbm = int((v["FETCH_SIZE"] * 2 + v["WRITE_SIZE"]) * 1024)
    out["_derived"] = {
        "kernel": fk, "launches_per_pass": 1, "gpu_cycles_per_xcd": cyc,
        "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
        "hbm_bytes_per_launch": hbm,
        "l2_hit_rate": (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])) if "TCC_HIT_sum" in v else None,
        "l2_read_requests_to_fabric": v.get("TCC_EA0_RDREQ_sum"),
        "l2_read_requests_marked_dram": v.get("TCC_EA0_RDREQ_DRAM_sum"),
        "note": "one bench.py step (--steps 1 --warmup 0 --no-cpu --no-full) per pass; SQ counters, FETCH_SIZE and "
                "WRITE_SIZE each in a pass of their own; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); "
                "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: what left the XCDs' L2s for the fabric (reads served by the "
                "Infinity Cache included: no counter of this set tells them from reads of the HBM); L2 hit rate = "
                "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)",
    }
    json.dump(out, open(out_dir / f"{a.tag}_pmc_summary.json", "w"), indent=1)
    # (bench.py looks through profiles/*_hbm_traffic.json for its workload, shard size, kernel and sources)
    traffic_file = out_dir / f"{a.tag}_hbm_traffic.json"
    m_local = a.m or bench.WORKLOADS[a.workload][1]
    json.dump({"workload": a.workload, "n_local": a.n_local, "m_local": m_local, "kernel": fk, "source_sha256": bench.kernel_source_hash(),
               "FETCH_SIZE_KB": v["FETCH_SIZE"],
               "WRITE_SIZE_KB": v["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled as the gfx950 "
                       "correction for wide coalesced reads prescribes; one launch = one table of this workload"},
              open(traffic_file, "w"), indent=1)
    if a.bench:
        line = [l for l in open(a.bench).read().splitlines() if l.startswith("{")][-1]
        (out_dir / f"{a.tag}_bench.json").write_text(line + "\n")
    print(json.dumps(out["_derived"], indent=1))


if __name__ == "__main__":
    main()
