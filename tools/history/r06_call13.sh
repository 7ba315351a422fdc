#!/bin/bash
# Round 6, call 13: the patch kernel with per-wave progress words instead of a barrier per tap (forced variant 94 / PFD_PATCH_FL=1):
# correctness (selftest incl. forced 10400 cases, the C2 launch list twice = same bits), cold replay, same-box A/B by environment.
set -u
O=gpurun_out/r06_call13; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
timeout 600 $S > $O/selftest_all.log 2>&1; echo "selftest rc=$?: $(tail -1 $O/selftest_all.log)"; grep -E "^FAIL" $O/selftest_all.log | head
PFD_PATCH_FL=1 PFD_REPLAY_DET=1 timeout 300 $S --replay $L 2>&1 | tail -2
for i in 1 2; do
  timeout 200 $S --replay-time $L > $O/replay_bar_$i.log 2>&1; echo "barrier: $(tail -1 $O/replay_bar_$i.log)"
  PFD_PATCH_FL=1 timeout 200 $S --replay-time $L > $O/replay_fl_$i.log 2>&1; echo "flags:   $(tail -1 $O/replay_fl_$i.log)"
done
grep -E "^ *(32768|8192|2048) +(320|640|1280) +(2880|5760|8640|11520|17280|23040) 3 1 0" $O/replay_bar_2.log | head -8
grep -E "^ *(32768|8192|2048) +(320|640|1280) +(2880|5760|8640|11520|17280|23040) 3 1 0" $O/replay_fl_2.log | head -8
for i in 1 2 3; do
  for f in 0 1; do
    PFD_PATCH_FL=$f timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $O/bench_fl${f}_$i.json 2> $O/bench_fl${f}_$i.err
    echo "PFD_PATCH_FL=$f run $i: $(python -c "import json; d=json.load(open('$O/bench_fl${f}_$i.json')); print(round(d['ms_per_step'],2), 'ms/batch')" 2>&1 | tail -1)"
  done
done
