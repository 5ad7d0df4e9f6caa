#!/bin/bash
# Profile battery of the twelve bench shapes, one group per gpurun call:  tools/gpu_r03_profiles.sh a|b|c|d
# Summaries land under gpurun_out/profiles/ (copy into profiles/ and commit).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
P=tools/profile_workload.sh
case "$1" in
  a) bash $P c3 r03_c3 && bash $P c2 r03_c2 && bash $P c4 r03_c4 ;;
  b) bash $P c3 r03_c3_n500k --n 500000 && bash $P c3 r03_c3_n250k --n 250000 && bash $P c3 r03_c3_n125k --n 125000 ;;
  c) bash $P c5shard r03_c5shard && bash $P cca30 r03_cca30 ;;
  d) bash $P wide r03_wide100 && bash $P wide128 r03_wide128 && bash $P wide200 r03_wide200 && bash $P wide256 r03_wide256 ;;
  *) echo "usage: $0 a|b|c|d"; exit 2 ;;
esac
