#!/bin/bash
set -u
O=gpurun_out/r04_call9; mkdir -p $O
( cd prompt-free-diffusion_amd/csrc && timeout 300 ./build/selftest --r4 > ../../$O/selftest_r4.log 2>&1; echo "selftest --r4 rc=$?"; tail -1 ../../$O/selftest_r4.log; grep FAIL ../../$O/selftest_r4.log | head )
for i in 1 2; do timeout 300 python tools/determinism_gemm.py 2>/dev/null | tee -a $O/determinism_gemm.log; done
timeout 300 python tools/determinism_check.py 2>/dev/null | tee $O/determinism_check.log
timeout 300 python tools/determinism_layers.py 2>/dev/null | grep "alone\|ops per" | tee $O/determinism_layers.log
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_hip_kernels_fullsize.py tests/test_hip_trajectory.py -m gpu -x -q -s -k "repeated_launches or groupnorm_statistics or lanes or c3_trajectory or unet_c2_batch8 or zero_uncond or cfg_prefix or full_size_properties or controlnet_c3 or wide_512x768" > $O/pytest_quick.log 2>&1; echo "pytest quick rc=$?"; tail -5 $O/pytest_quick.log; grep -n "parity\] producer\|lanes 2 vs" $O/pytest_quick.log | cut -c1-200
