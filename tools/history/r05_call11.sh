#!/bin/bash
# Round 5, GPU call 11: the cleaned library (env switches folded, PP / 5-stage / flag kernels gone) + the CFG-pair re-join
# without copies (PfdGemmDesc.res_rows): selftest, the whole GPU suite, then the A/B against the torch.cat form.
set -u
O=gpurun_out/r05_call11; mkdir -p $O
B="--steps 5 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/selftest_r5.log)"; grep FAIL $O/selftest_r5.log | head
timeout 300 $S > $O/selftest_all.log 2>&1; echo "selftest (all) rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/selftest_all.log)"; grep FAIL $O/selftest_all.log | head
timeout 900 python -m pytest tests/ -x -q -s -m gpu --durations=10 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu -x rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
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
for rep in 1 2 3; do
  run nocopy_$rep PFD_QUIET=1
  run cat_$rep PFD_PAIR_CAT=1
done
