#!/bin/bash
# A look at the usual shapes after the library restructuring: dense bench lines (quick_bench),
# the wide-row workloads with the partition, and pruned against dense.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout -k 10 500 bash tools/quick_bench.sh r03 || exit 1
O=gpurun_out/quick_r03_wide.jsonl; : > $O
for w in wide wide128 wide200 wide256 c4; do
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-full --workload $w >> $O 2>gpurun_out/quick_err.log || { tail -3 gpurun_out/quick_err.log; exit 1; }
done
python3 - <<PY
import json
for l in open("$O"):
    r=json.loads(l); print(r["config"]["workload"][:60], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"])
PY
timeout -k 10 300 python3 -u tools/exp/prune_try.py > gpurun_out/r03_prune_try.txt 2>&1 || { tail -5 gpurun_out/r03_prune_try.txt; exit 1; }
cat gpurun_out/r03_prune_try.txt
