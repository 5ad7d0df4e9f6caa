#!/bin/bash
# One FETCH_SIZE counter pass over one bench.py step: prints the kernel's FETCH_SIZE sum (KiB).
#   tools/fetch_pass.sh <tag> [bench.py options]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/fetch_$TAG
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-full "$@" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 - <<PY
import csv, glob
tot = 0.0
for f in glob.glob("$O/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "fused_knn_kernel" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
            tot += float(row["Counter_Value"])
print("$TAG FETCH_SIZE KiB", tot, " x2 GB:", round(2 * tot * 1024 / 1e9, 1))
PY
