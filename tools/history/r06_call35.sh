#!/bin/bash
# Round 6, call 35: launch list regenerated from the final tree (the 640-channel convolutions of the 32^2 level now carry the fused GroupNorm request),
# replayed twice (same bits), timed, then the rocprofv3 / PMC passes over it
set -u
O=gpurun_out/r06_call35; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 python tools/dump_unet_shapes.py 2>&1 | tail -1
cp profiles/unet_c2_gemm_shapes.txt $O/unet_c2_gemm_shapes.txt
PFD_REPLAY_DET=1 timeout 300 $S --replay profiles/unet_c2_gemm_shapes.txt 2>&1 | tail -1
timeout 200 $S --replay-time profiles/unet_c2_gemm_shapes.txt > $O/replay_time.log 2>&1; tail -1 $O/replay_time.log
bash tools/profile_round.sh r06_profile_f r06 > gpurun_out/profile_round_f.log 2>&1; tail -2 gpurun_out/profile_round_f.log
