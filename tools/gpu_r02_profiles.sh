#!/bin/bash
# rocprofv3 evidence (bench line, kernel statistics, counter passes) for a list of bench.py workloads
set -o pipefail
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  IFS=: read -r wl tag extra <<< "$spec"
  echo "== $wl $tag $extra"
  bash tools/profile_workload.sh $wl $tag $extra > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
  tail -8 gpurun_out/prof_$tag.log | head -7
done
