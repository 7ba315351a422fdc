#!/bin/bash
# Round 6, call 4: how much of the GEMM / conv time of a UNet pass is the cold weight stream?  The launch list of a C2 pass with
# every launch's weights from a fresh slice of a 2 GB pool (as in the sampler) vs from one cache-warm buffer.
set -u
O=gpurun_out/r06_call4; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
for i in 1 2; do
  timeout 200 $S --replay-time $L > $O/replay_cold_$i.log 2>&1; tail -1 $O/replay_cold_$i.log
  PFD_REPLAY_WARM=1 timeout 200 $S --replay-time $L > $O/replay_warm_$i.log 2>&1; tail -1 $O/replay_warm_$i.log
done
