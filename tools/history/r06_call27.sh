#!/bin/bash
# Round 6, call 27: split-K slabs as f16 partial sums (fp32 through round 5): selftests, launch-list replay (bits twice, time), kernel tests, same-box A/B
set -u
O=gpurun_out/r06_call27; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
timeout 600 $S > $O/selftest_all.log 2>&1; echo "selftest (all) rc=$?: $(tail -1 $O/selftest_all.log)"; grep -E "^FAIL" $O/selftest_all.log | head -8
timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$?: $(tail -1 $O/selftest_r5.log)"; grep -E "^FAIL" $O/selftest_r5.log | head -8
PFD_REPLAY_DET=1 timeout 300 $S --replay $L 2>&1 | tail -2
timeout 200 $S --replay-time $L > $O/replay_time_f16slabs.log 2>&1; tail -1 $O/replay_time_f16slabs.log
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_kernels_fullsize.py -x -q -m gpu -p no:cacheprovider > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?: $(tail -1 $O/pytest_kernels.log)"; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_kernels.log | head -10
bash tools/ab_bench.sh $O 3 base head
