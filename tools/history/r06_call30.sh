#!/bin/bash
# Round 6, call 30: split-K slabs written through the staging image + store pass (16-byte chunks of contiguous row segments) instead of
# 8-byte stores straight from the accumulators
set -u
O=gpurun_out/r06_call30; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
LIB=prompt-free-diffusion_amd/libpfd_hip.so
timeout 600 $S > $O/selftest_all.log 2>&1; echo "selftest (all) rc=$?: $(tail -1 $O/selftest_all.log)"; grep -E "^FAIL" $O/selftest_all.log | head -8
timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$?: $(tail -1 $O/selftest_r5.log)"; grep -E "^FAIL" $O/selftest_r5.log | head -8
PFD_REPLAY_DET=1 timeout 300 $S --replay $L 2>&1 | tail -1
cp $LIB $O/head.so
for i in 1 2; do
  for n in base head; do
    if [ $n = head ]; then cp $O/head.so $LIB; else cp variants/$n.so $LIB; fi
    timeout 200 $S --replay-time $L > $O/replay_time_${n}_$i.log 2>&1; echo "$n replay-time $i: $(tail -1 $O/replay_time_${n}_$i.log)"
  done
done
cp $O/head.so $LIB; rm -f $O/head.so
bash tools/ab_bench.sh $O 3 base head
