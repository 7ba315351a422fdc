#!/bin/bash
# round-4 GPU session 1: new-kernel correctness, ring A/B on the cold replay, lanes A/B, quick tests, one bench line
set -u
O=gpurun_out/r04_call1; mkdir -p $O
cd prompt-free-diffusion_amd/csrc
timeout 300 ./build/selftest --r4 > ../../$O/selftest_r4.log 2>&1; echo "selftest --r4 rc=$?"
timeout 200 ./build/selftest --attn512 > ../../$O/selftest_attn512.log 2>&1; echo "attn512 rc=$?"
L=../../profiles/unet_c2_gemm_shapes.txt
for rep in 1 2; do
  PFD_PATCH_RING=0 PFD_WS_RING=0 timeout 300 ./build/selftest --replay-time $L > ../../$O/replay_ring0_$rep.log 2>&1
  PFD_PATCH_RING=1 PFD_WS_RING=1 timeout 300 ./build/selftest --replay-time $L > ../../$O/replay_ring1_$rep.log 2>&1
done
tail -1 ../../$O/replay_ring0_1.log ../../$O/replay_ring1_1.log ../../$O/replay_ring0_2.log ../../$O/replay_ring1_2.log
cd ../..
timeout 900 python tools/lanes_ab.py --lanes 1,2,4 --rounds 2 > $O/lanes_ab.log 2>$O/lanes_ab.err; echo "lanes rc=$?"; tail -8 $O/lanes_ab.log
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -m gpu -x -q -s -k "attention or layernorm_fold or full_tensors or conv_layers or gemm or native" > $O/pytest_quick.log 2>&1; echo "pytest quick rc=$?"; tail -3 $O/pytest_quick.log
timeout 900 python -m pytest tests/test_hip_trajectory.py -m gpu -x -q -s -k "c3_trajectory" > $O/pytest_c3.log 2>&1; echo "pytest c3 rc=$?"; tail -3 $O/pytest_c3.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2>$O/bench_c2.err; echo "bench rc=$?"; cut -c1-400 $O/bench_c2.json
