#!/bin/bash
# First GPU call of the next round (prepared at the end of round 4, never run): correctness of the forced-only candidates --
# 5-stage ring variants (26 = 64x160 on 4 waves, 46 = on 8 waves), the linear ring kernels whose activation fragments stay in
# registers with the weights alone on a 7-stage LDS ring (27 = 64x160 on 4 waves, 45 = on 8 waves, 85 = 128x160 on 8 waves:
# 6 K tiles in flight instead of 3 / 2; 29 = 27 at the depth of 26; 86 / 28 = the 2-stage tiles 82 / 22 with 3 weight stages and
# two blocks per CU) and the patch kernel that hands over through per-wave progress words in LDS instead of a barrier per
# tap (95; every poll is bounded, a protocol error shows as a FAIL, not a hang).  All of them are bit-identical to the kernels
# they would replace on the CPU emulation (tests/test_cpu_emulation.py); `selftest --r5` is the hardware confirmation, under a
# timeout FIRST, then their per-problem A/B on the cold replay of the C2 launch list
# (tile code 1000 + 100 * variant: a launch the forced variant does not serve falls back to the automatic choice).
#   usage (on the GPU box): bash tools/r05_first_call.sh   -> gpurun_out/r05_ring5/
set -u
O=gpurun_out/r05_ring5; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
timeout 400 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$?"; tail -1 $O/selftest_r5.log; grep FAIL $O/selftest_r5.log | head
for rep in 1 2; do
  for t in 0 3300 3600 3700 5300 5600 5500 9300 9500 9200 9600 3200 3800 3900; do     # 0 = automatic, 33 / 53 / 93 = today's rings (variants 23 / 43 / 83), 36 / 56 = 5 stages, 37 / 55 / 95 = A in registers + 7 weight stages, 92 = the 2-stage 128-row tile on 8 waves (variant 82), 96 = its register-operand form with 3 weight stages (86), 32 / 38 = the 2-stage 64-row tile (22) and its register-operand form (28), 39 = 37 with 5 weight stages (the depth of 36: what the register path costs at equal depth)
    timeout 60 $S --replay-time $L $t > $O/replay_t${t}_$rep.log 2>&1; echo "tile $t run $rep: $(tail -1 $O/replay_t${t}_$rep.log)"
  done
done
for rep in 1 2; do
  for t in 10800 10500; do               # patch kernel: 98 = barrier per tap with two weight stages (the like-for-like base), 95 = progress words in LDS
    timeout 60 $S --replay-time $L $t > $O/replay_t${t}_$rep.log 2>&1; echo "tile $t run $rep: $(tail -1 $O/replay_t${t}_$rep.log)"
  done
done
for t in 0 3300 3600 3700 5300 5600 5500 9300 9500 9200 9600 3200 3800 3900 10800 10500; do cp $O/replay_t${t}_2.log $O/replay_t$t.log; done
python tools/replay_merge.py $O 0 10800 10500 > $O/merge_patch.log 2>&1; tail -12 $O/merge_patch.log
python tools/replay_merge.py $O 0 3300 3600 3700 5300 5600 5500 9300 9500 9200 9600 3200 3800 3900 > $O/merge.log 2>&1; tail -40 $O/merge.log
# attention d = 40: mode 7 = today's default (6) + s_setprio(1) around the two MFMA clusters of a tile (built, unmeasured)
( cd prompt-free-diffusion_amd/csrc
  PFD_ATTN=7 PFD_ATTN_FORCE8=1 timeout 100 ./build/selftest --attn > ../../$O/selftest_attn_mode7.log 2>&1; echo "selftest --attn (mode 7) rc=$?"; tail -1 ../../$O/selftest_attn_mode7.log
  for rep in 1 2; do for m in 6 7 8; do PFD_ATTN=$m timeout 60 ./build/selftest --bench-attn > ../../$O/bench_attn_mode${m}_$rep.log 2>&1; echo "mode $m run $rep:"; grep -i "bench" ../../$O/bench_attn_mode${m}_$rep.log | head -8; done; done )

