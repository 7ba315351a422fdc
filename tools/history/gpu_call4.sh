#!/bin/bash
# round-4 GPU session 4: GroupNorm statistics from the producers (selftest, tests, end-to-end A/B), Winograd prototype, background oracle jobs
set -u
O=gpurun_out/r04_call4; mkdir -p $O
( cd prompt-free-diffusion_amd/csrc && timeout 300 ./build/selftest --r4 > ../../$O/selftest_r4.log 2>&1; echo "selftest --r4 rc=$?"; tail -1 ../../$O/selftest_r4.log; grep FAIL ../../$O/selftest_r4.log | head -20
  timeout 300 ./build/selftest --winograd > ../../$O/winograd.log 2>&1; echo "winograd rc=$?"; cat ../../$O/winograd.log | cut -c1-330 )
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels_fullsize.py tests/test_hip_trajectory.py -m gpu -x -q -s -k "groupnorm_statistics or lanes or c3_trajectory or unet_c2_batch8 or zero_uncond or cfg_prefix or full_size_properties or end_to_end or controlnet_c3 or wide_512x768" > $O/pytest_quick.log 2>&1; echo "pytest quick rc=$?"; tail -5 $O/pytest_quick.log; grep -n "parity\] producer\|lanes 2 vs" $O/pytest_quick.log | cut -c1-200
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof"
for rep in 1 2; do
  PFD_GN_PSTATS=0 timeout 400 $B > $O/bench_ps0_$rep.json 2>/dev/null; echo "pstats0 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ps0_$rep.json)"
  PFD_GN_PSTATS=1 timeout 400 $B > $O/bench_ps1_$rep.json 2>/dev/null; echo "pstats1 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ps1_$rep.json)"
done
