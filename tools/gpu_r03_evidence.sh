#!/bin/bash
# One JSON line + rocprofv3 kernel stats for every quoted number that lies outside the bench.py
# workloads: CCA producer, loss / loaders, drop-in API at PBMC size, FOSCTTM count kernel.
#   results: gpurun_out/profiles/r03_<name>.json and r03_<name>_kernel_stats.csv
set -o pipefail
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/profiles; mkdir -p $P
run() {  # name, then the command
  local name=$1; shift
  echo "== $name"
  (cd $R && timeout -k 10 400 "$@" > $P/r03_$name.json 2> $R/gpurun_out/ev_$name.err) || { tail -3 $R/gpurun_out/ev_$name.err; return 1; }
  tail -c 400 $P/r03_$name.json; echo
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev_$name -- "$@" > $R/gpurun_out/ev_$name.log 2>&1) || { tail -3 $R/gpurun_out/ev_$name.log; return 1; }
  cp $(ls -t $R/gpurun_out/ev_$name/*/*_kernel_stats.csv | head -1) $P/r03_${name}_kernel_stats.csv
}
run cca_pbmc python3 $R/tests/checks/bench_cca.py || exit 1
run cca_200k python3 $R/tests/checks/bench_cca.py --n1 200000 --n2 150000 --no-cpu || exit 1
run consumer python3 $R/tools/bench_consumer.py || exit 1
run dropin_pbmc python3 $R/tests/checks/bench_dropin.py || exit 1
run foscttm_d30_unit python3 $R/tools/bench_foscttm.py --d 30 --unit --noise 0.3 || exit 1
run foscttm_d50_n1p5 python3 $R/tools/bench_foscttm.py --d 50 --noise 1.5 || exit 1
run foscttm_d30_n1p5 python3 $R/tools/bench_foscttm.py --d 30 --noise 1.5 || exit 1
run foscttm_d30_n3 python3 $R/tools/bench_foscttm.py --d 30 --noise 3 || exit 1
run foscttm_d30_n6 python3 $R/tools/bench_foscttm.py --d 30 --noise 6 || exit 1
run foscttm_d30_n20 python3 $R/tools/bench_foscttm.py --d 30 --noise 20 || exit 1
echo evidence done
