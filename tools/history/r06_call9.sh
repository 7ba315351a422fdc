#!/bin/bash
# Round 6, call 9: mid-round checkpoint -- regenerate the C2 launch list with the ABI-9 fields (24 per line), replay it, the WHOLE GPU
# suite as one command (incl. the new RCCL one-rank and 1024x768 tests), bench C2 with the driver's arguments.
set -u
O=gpurun_out/r06_call9; mkdir -p $O
T0=$(date +%s)
timeout 300 python tools/dump_unet_shapes.py 2>&1 | tail -1
cp profiles/unet_c2_gemm_shapes.txt $O/unet_c2_gemm_shapes.txt
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --replay profiles/unet_c2_gemm_shapes.txt 2>&1 | tail -2
PFD_REPLAY_DET=1 timeout 300 $S --replay profiles/unet_c2_gemm_shapes.txt 2>&1 | tail -1
timeout 300 $S --replay-time profiles/unet_c2_gemm_shapes.txt > $O/replay_time.log 2>&1; tail -1 $O/replay_time.log
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)|rccl\]" $O/pytest_gpu.log | head
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"; head -c 300 $O/bench_c2.json; echo
