#!/bin/bash
# Same-box A/B of library builds: tools/ab_bench.sh <out dir> <rounds> <name> [<name> ...]   (name = variants/<name>.so, or "head" = the tree's own build)
# Alternates the builds <rounds> times with the driver's bench command (short form) and prints ms per batch of each run.
set -u
O=$1; R=$2; shift 2
mkdir -p $O
LIB=prompt-free-diffusion_amd/libpfd_hip.so
cp $LIB $O/head.so
for i in $(seq 1 $R); do
  for n in "$@"; do
    if [ "$n" = head ]; then cp $O/head.so $LIB; else cp variants/$n.so $LIB; fi
    timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof ${AB_ARGS:-} > $O/bench_${n}_$i.json 2> $O/bench_${n}_$i.err
    echo "$n run $i: $(python -c "import json; d=json.load(open('$O/bench_${n}_$i.json')); print(round(d['ms_per_step'],2), 'ms/batch', round(d['value'],3), 'images/s')" 2>&1 | tail -1)"
  done
done
cp $O/head.so $LIB; rm -f $O/head.so
