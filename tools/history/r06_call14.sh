#!/bin/bash
# Round 6, call 14: flag hand-over patch kernel, readiness word prefetched under the previous tap's MFMAs
set -u
O=gpurun_out/r06_call14; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
PFD_PATCH_FL=1 PFD_REPLAY_DET=1 timeout 300 $S --replay $L 2>&1 | tail -1
for i in 1 2; do
  timeout 200 $S --replay-time $L > $O/replay_bar_$i.log 2>&1; echo "barrier: $(tail -1 $O/replay_bar_$i.log)"
  PFD_PATCH_FL=1 timeout 200 $S --replay-time $L > $O/replay_fl_$i.log 2>&1; echo "flags:   $(tail -1 $O/replay_fl_$i.log)"
done
grep -E "^ *(32768|8192|2048) +(320|640|1280) +(2880|5760|8640|11520|17280|23040) 3 1 0" $O/replay_bar_2.log | head -6
grep -E "^ *(32768|8192|2048) +(320|640|1280) +(2880|5760|8640|11520|17280|23040) 3 1 0" $O/replay_fl_2.log | head -6
