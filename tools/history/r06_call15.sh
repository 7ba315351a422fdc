#!/bin/bash
# Round 6, call 15: attention2_kernel<D, 4, false, false, 3> -- every tile of a short key stream (Nk <= 192: cross-attention, 8^2
# self-attention) resident in LDS after ONE round trip -- correctness, micro-bench, same-box A/B against variants/base_r6c.so
set -u
O=gpurun_out/r06_call15; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --attn > $O/selftest_attn.log 2>&1; echo "selftest --attn rc=$?: $(tail -1 $O/selftest_attn.log)"
PFD_ATTN_FORCE8=1 timeout 300 $S --attn > $O/selftest_attn8.log 2>&1; echo "selftest --attn (8-wave forced) rc=$?: $(tail -1 $O/selftest_attn8.log)"
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest kernels+parity rc=$?: $(tail -1 $O/pytest.log)"
timeout 120 $S --bench-attn 2>&1 | tee $O/bench_attn_new.log | tail -4
cp prompt-free-diffusion_amd/libpfd_hip.so /tmp/head.so; cp variants/base_r6c.so prompt-free-diffusion_amd/libpfd_hip.so
timeout 120 $S --bench-attn 2>&1 | tee $O/bench_attn_old.log | tail -4
cp /tmp/head.so prompt-free-diffusion_amd/libpfd_hip.so
bash tools/ab_bench.sh $O 3 base_r6c head
