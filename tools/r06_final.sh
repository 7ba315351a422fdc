#!/bin/bash
# Round 6: the WHOLE `pytest -m gpu` at HEAD in one piece (as the driver runs it: -x), smoke(), then the bench lines of the
# four BASELINE configurations with the driver's arguments.
set -u
O=gpurun_out/r06_final; mkdir -p $O
T0=$(date +%s)
timeout 1300 python -m pytest tests/ -x -q -s -m gpu --durations=15 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu -x rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
S=prompt-free-diffusion_amd/csrc/build/selftest; timeout 300 $S > $O/selftest_all.log 2>&1; echo "selftest (all) rc=$?: $(tail -1 $O/selftest_all.log)"; timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$?: $(tail -1 $O/selftest_r5.log)"; PFD_ATTN3_FORCE=1 timeout 300 $S --attn > $O/selftest_attn3.log 2>&1; echo "selftest --attn (attention3 forced) rc=$?: $(tail -1 $O/selftest_attn3.log)"; timeout 300 $S --narrow > $O/selftest_narrow.log 2>&1; echo "selftest --narrow rc=$?: $(tail -1 $O/selftest_narrow.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?: $(tail -1 $O/smoke.log)"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$? after $(( $(date +%s) - T0 )) s"; head -c 400 $O/bench_c2.json; echo
for c in c3 c4 c5; do
  timeout 400 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$? after $(( $(date +%s) - T0 )) s: $(head -c 260 $O/bench_$c.json | cut -c1-260)"
done
