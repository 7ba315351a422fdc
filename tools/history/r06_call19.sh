#!/bin/bash
# Round 6, call 19: context encode / VAE decode of consecutive batches on side streams beside the DDIM loop (generate(overlap=True))
set -u
O=gpurun_out/r06_call19; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -p no:cacheprovider -k "overlapped or graphed_stages or concurrent or serving or rccl" > $O/pytest.log 2>&1; echo "pytest rc=$?: $(tail -1 $O/pytest.log)"
for i in 1 2 3; do
  for f in "--no-overlap" ""; do
    n=$([ -z "$f" ] && echo ovl || echo serial)
    timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof $f > $O/bench_${n}_$i.json 2> $O/bench_${n}_$i.err
    echo "$n run $i: $(python -c "import json; d=json.load(open('$O/bench_${n}_$i.json')); print(round(d['ms_per_step'],2), 'ms/batch', round(d['value'],3), 'images/s')" 2>&1 | tail -1)"
  done
done
for c in c3 c5; do
  for f in "--no-overlap" ""; do
    n=$([ -z "$f" ] && echo ovl || echo serial)
    timeout 600 python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-prof $f > $O/bench_${c}_${n}.json 2> $O/bench_${c}_${n}.err
    echo "$c $n: $(python -c "import json; d=json.load(open('$O/bench_${c}_${n}.json')); print(round(d['ms_per_step'],2), 'ms/batch', round(d['value'],3), 'images/s')" 2>&1 | tail -1)"
  done
done
