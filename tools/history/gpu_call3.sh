#!/bin/bash
# round-4 GPU session 3: A-fragment prefetch of the patch kernel (correctness + cold-replay A/B + end to end), then the whole GPU suite
set -u
O=gpurun_out/r04_call3; mkdir -p $O
cd prompt-free-diffusion_amd/csrc
timeout 300 ./build/selftest --r4 > ../../$O/selftest_r4.log 2>&1; echo "selftest --r4 rc=$?"; tail -1 ../../$O/selftest_r4.log
timeout 300 ./build/selftest --patch-wide > ../../$O/selftest_patch_wide.log 2>&1; echo "patch-wide rc=$?"; tail -1 ../../$O/selftest_patch_wide.log
L=../../profiles/unet_c2_gemm_shapes.txt
for rep in 1 2; do
  PFD_PATCH_PF=0 timeout 300 ./build/selftest --replay-time $L > ../../$O/replay_pf0_$rep.log 2>&1
  PFD_PATCH_PF=1 timeout 300 ./build/selftest --replay-time $L > ../../$O/replay_pf1_$rep.log 2>&1
done
for f in pf0_1 pf1_1 pf0_2 pf1_2; do tail -n1 ../../$O/replay_$f.log; done
cd ../..
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof"
PFD_PATCH_PF=0 timeout 400 $B > $O/bench_pf0.json 2>/dev/null; echo "pf0 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_pf0.json)"
PFD_PATCH_PF=1 timeout 400 $B > $O/bench_pf1.json 2>/dev/null; echo "pf1 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_pf1.json)"
timeout 1500 python -m pytest tests/ -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 $O/pytest_gpu.log
