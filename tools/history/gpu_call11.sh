#!/bin/bash
# last GPU minutes of round 4: HEAD against the round-3 tree (variants/r03_tree = git archive of 835ff0f, built) alternating on ONE box,
# then the trajectory tests on the fixture-backed oracle (tests/golden/trajectories.npz)
set -u
O=gpurun_out/r04_ab; mkdir -p $O
B="--steps 6 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
pair() {
  timeout 150 python bench.py $B > $O/r04_$1.json 2> $O/r04_$1.err
  ( cd variants/r03_tree && timeout 120 python bench.py $B > ../../$O/r03_$1.json 2> ../../$O/r03_$1.err )
  echo "pair $1 done after $(( $(date +%s) - T0 )) s"
}
pair 1
timeout 160 python -m pytest tests/test_hip_trajectory.py -m gpu -q -s -x > $O/pytest_trajectory.log 2>&1; echo "pytest trajectory rc=$? after $(( $(date +%s) - T0 )) s"; grep "trajectory\]" $O/pytest_trajectory.log | cut -c1-230; tail -2 $O/pytest_trajectory.log
pair 2
for f in r04_1 r03_1 r04_2 r03_2; do python - <<P
import json
try:
    d = json.load(open("$O/$f.json")); print("%-6s %7.1f ms per batch  %.3f images/s  loop %s" % ("$f", d["ms_per_step"], d["value"], d.get("stage_ms_per_batch", {}).get("ddim_loop_ms")))
except Exception as e:
    print("$f", "no result:", e)
P
done
