#!/bin/bash
# Round 6, call 6: conv3x3_narrow_kernel (UNet head 320 -> 4, VAE conv_out) -- selftest cases + bench against the round-5 kernel,
# the stage tests that contain it, same-box A/B end to end.
set -u
O=gpurun_out/r06_call6; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --narrow 2>&1 | tee $O/selftest_narrow.log | tail -14
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest parity+kernels rc=$?"; tail -2 $O/pytest.log
bash tools/ab_bench.sh $O 3 base_r6a head
