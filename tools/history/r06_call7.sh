#!/bin/bash
# Round 6, call 7: attention3_kernel with two 64-key tiles per barrier (K ring 5, V^T ring 4) -- correctness, micro-bench, same-box
# A/B against variants/base_r6b.so (= one tile per barrier).
set -u
O=gpurun_out/r06_call7; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
PFD_ATTN3_FORCE=1 timeout 300 $S --attn > $O/selftest_attn_force.log 2>&1; echo "selftest --attn (forced attention3) rc=$?: $(tail -1 $O/selftest_attn_force.log)"
timeout 300 $S --attn > $O/selftest_attn.log 2>&1; echo "selftest --attn rc=$?: $(tail -1 $O/selftest_attn.log)"
timeout 120 $S --bench-attn 2>&1 | tee $O/bench_attn.log | head -3
PFD_HIP_LIB=$(pwd)/variants/base_r6b.so timeout 120 $S --bench-attn 2>&1 | head -3
bash tools/ab_bench.sh $O 3 base_r6b head
