#!/bin/bash
# Round 6, call 33: GroupNorm inside the split-K reduction also for the 640-channel norms of the 32^2 level (20 channels per group)
set -u
O=gpurun_out/r06_call33; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$?: $(tail -1 $O/selftest_r5.log)"; grep -E "^FAIL" $O/selftest_r5.log | head -8
timeout 300 python tools/gn_paths.py > $O/gn_paths.log 2>&1; tail -12 $O/gn_paths.log
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?: $(tail -1 $O/pytest.log)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
bash tools/ab_bench.sh $O 4 base head
