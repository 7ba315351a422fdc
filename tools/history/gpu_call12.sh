#!/bin/bash
# remaining GPU seconds of round 4: the two test files the closing call did not reach
set -u
O=gpurun_out/r04_last; mkdir -p $O
T0=$(date +%s)
timeout 150 python -m pytest tests/test_hip_parity.py -m gpu -q -s > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_parity.log
timeout 120 python -m pytest tests/test_hip_kernels_fullsize.py -m gpu -q -s -k "unet_c5_shape or wide_512x768" > $O/pytest_fullsize_rest.log 2>&1; echo "pytest fullsize rest rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_fullsize_rest.log
