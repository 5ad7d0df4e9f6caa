#!/bin/bash
# Quick look at the fused kernel on the usual shapes: bench lines without the CPU sample and the
# full search.  Output: gpurun_out/quick_<tag>.jsonl
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/quick_$TAG.jsonl; : > $O
run() { timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-full "$@" >> $O 2>gpurun_out/quick_err.log || { tail -3 gpurun_out/quick_err.log; exit 1; }; }
run --workload c3
run --workload c2
run --workload c4
run --workload c3 --n 125000
run --workload c3 --n 250000
run --workload c5shard
run --workload cca30
python3 - <<PY
import json
for l in open("$O"):
    r=json.loads(l); print(r["config"]["workload"][:60], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"])
PY
