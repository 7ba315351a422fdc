#!/bin/bash
# Round 5, GPU call 3: hardware check of the fused split-K reduction + GroupNorm (selftest --r5, kernel-level pytest), then the
# end-to-end A/B on one box: HEAD (losers deleted, winners adopted, fused reduction on) vs PFD_GNF=0 (fused reduction off)
# vs the remaining forced-only candidates under PFD_R5X (1: 5-stage ring 26, 2: 46, 4: patch kernel without the barrier per tap),
# then the whole GPU suite.
set -u
O=gpurun_out/r05_call3; mkdir -p $O
B="--steps 5 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/selftest_r5.log)"; grep FAIL $O/selftest_r5.log | head
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -p no:cacheprovider > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/pytest_kernels.log)"
run() {   # run <tag> <env...>
  local tag=$1; shift
  env "$@" timeout 170 python bench.py $B > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$? after $(( $(date +%s) - T0 )) s: $(python - <<P
import json
try:
    d = json.load(open("$O/$tag.json")); print("%.1f ms per batch, loop %s" % (d["ms_per_step"], d.get("stage_ms_per_batch", {}).get("ddim_loop_ms")))
except Exception as e:
    print("no result:", str(e)[:80])
P
)"
}
for rep in 1 2; do
  run head_$rep PFD_R5X=0
  run gnf0_$rep PFD_GNF=0
  run patch8off_$rep PFD_PATCH8=0
  run r5x1_$rep PFD_R5X=1
  run r5x2_$rep PFD_R5X=2
  run r5x4_$rep PFD_R5X=4
done
run head_3 PFD_R5X=0
timeout 900 python -m pytest tests/ -q -s -m gpu --durations=10 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
