#!/bin/bash
# rocprofv3 kernel stats + four counter passes + one bench line (taken last, so that it finds this run's own
# traffic record) -- formerly: one bench line + rocprofv3 kernel stats + four counter passes (SQ, FETCH_SIZE, WRITE_SIZE, L2 hit / miss
# and fabric-side read requests, each a run of its own) for one bench.py workload, condensed into profiles/<tag>_* by summarize_profiles.py.
#   tools/profile_workload.sh <workload> <tag> [bench.py options, e.g. --n 125000]
# Runs on the GPU box; raw outputs under gpurun_out/<tag>/, summaries under gpurun_out/profiles/<tag>_*
# (copy those into profiles/ and commit).
set -o pipefail
W=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O $R/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
ONE="--workload $W --steps 1 --warmup 0 --no-cpu --no-full"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu --no-full "$@" > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/bench.py $ONE "$@" > $O/pmc_sq.log 2>&1 || { tail -5 $O/pmc_sq.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $ONE "$@" > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $ONE "$@" > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O/pmc_tcc -- python3 $R/bench.py $ONE "$@" > $O/pmc_tcc.log 2>&1 || { tail -5 $O/pmc_tcc.log; exit 1; }
cd $R
python3 tools/summarize_profiles.py --stats $O/stats --pmc $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_tcc --tag $TAG \
    --workload $W --out-dir $R/gpurun_out/profiles "$@" || exit 1
# the bench line of the same library, with the traffic record just taken where bench.py looks for it
cp $R/gpurun_out/profiles/${TAG}_hbm_traffic.json $R/profiles/
timeout -k 10 400 python3 bench.py --workload $W --steps 5 --warmup 1 "$@" > $O/bench.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
grep '^{' $O/bench.log | tail -1 > $R/gpurun_out/profiles/${TAG}_bench.json
