#!/bin/bash
# parity tests of the k-NN path, then timings of the fused kernel with and without the locality partition
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_knn.py tests/test_gpu_full_size.py -x -q --timeout 400 --timeout-method thread > $O/parity_b.log 2>&1
rc=$?; tail -15 $O/parity_b.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/exp/ab_locality.py 2>&1 | tee $O/ab_locality.log
