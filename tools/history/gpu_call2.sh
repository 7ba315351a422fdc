#!/bin/bash
# round-4 GPU session 2: ABI 8 (k_split / zero_rows) correctness, end-to-end A/B of the fused forms, GroupNorm chunking sweep,
# ATen ops riding in the loop
set -u
O=gpurun_out/r04_call2; mkdir -p $O
( cd prompt-free-diffusion_amd/csrc && timeout 300 ./build/selftest --r4 > ../../$O/selftest_r4.log 2>&1; echo "selftest --r4 rc=$?"; tail -1 ../../$O/selftest_r4.log; grep FAIL ../../$O/selftest_r4.log | head )
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels_fullsize.py -m gpu -x -q -s -k "zero_uncond or cfg_prefix or unet_c2_batch8 or unet_eps or end_to_end or hipgraph or layernorm_fold or controlnet_c3" > $O/pytest_quick.log 2>&1; echo "pytest quick rc=$?"; tail -3 $O/pytest_quick.log
timeout 300 python tools/aten_ops_in_loop.py > $O/aten_ops.log 2>$O/aten_ops.err; echo "aten rc=$?"; tail -25 $O/aten_ops.log
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof"
for rep in 1 2; do
  PFD_GEMM_FUSE=0 timeout 400 $B > $O/bench_fuse0_$rep.json 2>/dev/null; echo "fuse0 $(cut -c1-120 $O/bench_fuse0_$rep.json | grep -o '"ms_per_step": [0-9.]*' )"
  PFD_GEMM_FUSE=1 timeout 400 $B > $O/bench_fuse1_$rep.json 2>/dev/null; echo "fuse1 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_fuse1_$rep.json)"
done
for nb in 1024 2048; do
  PFD_GN_BLOCKS=$nb timeout 400 $B > $O/bench_gnblocks_$nb.json 2>/dev/null; echo "gn_blocks $nb $(grep -o '"ms_per_step": [0-9.]*' $O/bench_gnblocks_$nb.json)"
done
PFD_PATCH_RING=0 PFD_WS_RING=0 timeout 400 $B > $O/bench_ring0.json 2>/dev/null; echo "ring0 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ring0.json)"
( cd prompt-free-diffusion_amd/csrc && ./build/selftest --bench-gn > ../../$O/bench_gn_512.log 2>&1; PFD_GN_BLOCKS=1024 ./build/selftest --bench-gn > ../../$O/bench_gn_1024.log 2>&1; PFD_GN_BLOCKS=2048 ./build/selftest --bench-gn > ../../$O/bench_gn_2048.log 2>&1; paste -d'\n' ../../$O/bench_gn_512.log ../../$O/bench_gn_2048.log | grep groupnorm | cut -c1-130 )
