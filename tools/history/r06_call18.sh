#!/bin/bash
# Round 6, call 18: the 256 x 320 GEGLU tile as a persistent kernel with the next tile's first stage prefetched under the epilogue
set -u
O=gpurun_out/r06_call18; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
PFD_GEGLU_PERSIST=2 timeout 600 $S > $O/selftest_persist2.log 2>&1; echo "selftest (persistent form forced) rc=$?: $(tail -1 $O/selftest_persist2.log)"; grep -E "^FAIL" $O/selftest_persist2.log | head -5
timeout 600 $S --ln > $O/selftest_ln.log 2>&1; echo "selftest --ln rc=$?: $(tail -1 $O/selftest_ln.log)"
PFD_REPLAY_DET=1 timeout 300 $S --replay $L 2>&1 | tail -1
for i in 1 2; do
  PFD_GEGLU_PERSIST=0 timeout 100 $S --replay-time profiles/r06_fixed_cost_probe_shapes.txt 2>&1 | grep -E "^ *(32768|16384|8192) +2560" | awk '{print "one tile per block:", $1,$2,$3,$16}' | tr '\n' ';'; echo
  timeout 100 $S --replay-time profiles/r06_fixed_cost_probe_shapes.txt 2>&1 | grep -E "^ *(32768|16384|8192) +2560" | awk '{print "persistent:        ", $1,$2,$3,$16}' | tr '\n' ';'; echo
done
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_kernels_fullsize.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest kernels rc=$?: $(tail -1 $O/pytest.log)"
for i in 1 2 3; do
  for f in 0 1; do
    PFD_GEGLU_PERSIST=$f timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $O/bench_p${f}_$i.json 2> $O/bench_p${f}_$i.err
    echo "PFD_GEGLU_PERSIST=$f run $i: $(python -c "import json; d=json.load(open('$O/bench_p${f}_$i.json')); print(round(d['ms_per_step'],2), 'ms/batch')" 2>&1 | tail -1)"
  done
done
