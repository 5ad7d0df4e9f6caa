#!/bin/bash
# GPU session: whole GPU test suite, single-rank rehearsal of the RCCL code path,
# C2 bench + kernel stats of the kernel as it stands.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/gpu_tests.log 2>&1
rc=$?; tail -5 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 SCMIAC_BENCH_FORCE_DIST=1 \
  timeout -k 10 300 python bench.py --steps 3 --no-full > $O/force_dist_1gpu.json 2> $O/force_dist_1gpu.err || exit $?
tail -c 600 $O/force_dist_1gpu.json
timeout -k 10 300 python bench.py --workload c2 --steps 10 --warmup 2 > $O/c2_bench.json 2> $O/c2_bench.err || exit $?
tail -c 900 $O/c2_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/c2_stats -- python3 $R/bench.py --workload c2 --steps 10 --warmup 2 --no-cpu --no-full > $O/c2_stats.log 2>&1 || exit $?
echo done
