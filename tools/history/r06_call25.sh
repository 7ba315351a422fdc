#!/bin/bash
# Round 6, call 25: HIP runtime knobs that touch hipGraph replay / kernel-argument placement, same-box A/B on the C2 bench (short form)
set -u
O=gpurun_out/r06_call25; mkdir -p $O
run() {  # name, env assignment (or empty)
  local n=$1 e=$2 i=$3
  env $e timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $O/bench_${n}_$i.json 2> $O/bench_${n}_$i.err
  echo "$n [$e] run $i: $(python -c "import json; d=json.load(open('$O/bench_${n}_$i.json')); print(round(d['ms_per_step'],2), 'ms/batch')" 2>&1 | tail -1)"
}
for i in 1 2 3; do
  run base "PFD_NOP=1" $i
  run devkarg1 "HIP_FORCE_DEV_KERNARG=1" $i
  run devkarg0 "HIP_FORCE_DEV_KERNARG=0" $i
  run pktcap1 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" $i
  run pktcap0 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" $i
  run kacopy0 "DEBUG_HIP_KERNARG_COPY_OPT=0" $i
  run hwq1 "GPU_MAX_HW_QUEUES=1" $i
  run hwq2 "GPU_MAX_HW_QUEUES=2" $i
  run gbatch0 "DEBUG_HIP_GRAPH_BATCH_SIZE=0" $i
  run gbatch4k "DEBUG_HIP_GRAPH_BATCH_SIZE=4096" $i
done
