#!/bin/bash
set -u
O=gpurun_out/r04_call6; mkdir -p $O
( cd prompt-free-diffusion_amd/csrc && PFD_REPLAY_DET=1 timeout 300 ./build/selftest --replay ../../profiles/unet_c2_gemm_shapes.txt > ../../$O/replay_det.log 2>&1; echo "replay det rc=$?"; grep -c NONDET ../../$O/replay_det.log; grep NONDET ../../$O/replay_det.log | sort | uniq -c | head -20; tail -1 ../../$O/replay_det.log )
timeout 300 python tools/determinism_layers.py > $O/det_layers.log 2>$O/det_layers.err; echo "layers rc=$?"; tail -15 $O/det_layers.log; tail -3 $O/det_layers.err
