#!/bin/bash
# Round 6, call 5: gn_apply_pstats_kernel with its first data round trip requested before the statistics fold -- parity tests of
# the GroupNorm family, then same-box A/B against the previous build (variants/base_r6a.so = HEAD before the change).
set -u
O=gpurun_out/r06_call5; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_kernels_fullsize.py -x -q -m gpu -k "groupnorm or gn or norm" -p no:cacheprovider > $O/pytest_gn.log 2>&1; echo "pytest groupnorm rc=$?"; tail -2 $O/pytest_gn.log
bash tools/ab_bench.sh $O 3 base_r6a head
