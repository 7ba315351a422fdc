#!/bin/bash
# Round 5, GPU call 1: the WHOLE `pytest -m gpu` at HEAD in one piece (no -x: every failure of the call is seen), then the
# hardware confirmation of the forced-only kernels (`selftest --r5`, bounded), then one bench line of the default path.
#   usage (on the GPU box): bash tools/r05_call1.sh   -> gpurun_out/r05_call1/
set -u
O=gpurun_out/r05_call1; mkdir -p $O
T0=$(date +%s)
git rev-parse HEAD > $O/head.txt 2>/dev/null
timeout 1300 python -m pytest tests/ -q -s -m gpu --durations=25 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? after $(( $(date +%s) - T0 )) s"; tail -4 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 400 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$? after $(( $(date +%s) - T0 )) s"; tail -2 $O/selftest_r5.log; grep -c FAIL $O/selftest_r5.log; grep FAIL $O/selftest_r5.log | head
timeout 300 python bench.py --steps 6 --warmup 2 > $O/bench_base.json 2> $O/bench_base.err; echo "bench rc=$? after $(( $(date +%s) - T0 )) s"; cat $O/bench_base.json | head -c 1500
