#!/bin/bash
# Round 6, call 29: the statistics-emitting / GroupNorm-fused split-K reductions with all their (now 16-byte / 8-byte f16) slab loads in ONE round trip
set -u
O=gpurun_out/r06_call29; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
LIB=prompt-free-diffusion_amd/libpfd_hip.so
cp $LIB $O/head.so
for n in head rgn6 u6; do
  if [ $n = head ]; then cp $O/head.so $LIB; else cp variants/$n.so $LIB; fi
  timeout 300 $S --r5 > $O/selftest_r5_$n.log 2>&1; echo "$n selftest --r5: $(tail -1 $O/selftest_r5_$n.log)"
  PFD_REPLAY_DET=1 timeout 300 $S --replay $L 2>&1 | tail -1
  for i in 1 2; do timeout 200 $S --replay-time $L > $O/replay_time_${n}_$i.log 2>&1; echo "$n replay-time $i: $(tail -1 $O/replay_time_${n}_$i.log)"; done
done
cp $O/head.so $LIB; rm -f $O/head.so
bash tools/ab_bench.sh $O 3 head u6
