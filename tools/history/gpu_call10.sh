#!/bin/bash
# round-4 closing call: the GPU suite on HEAD first, then the evidence (replay list, benches, rocprofv3 passes)
set -u
O=gpurun_out/r04_final; mkdir -p $O
T0=$(date +%s)
timeout 1150 python -m pytest tests/ -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log
# the launch list of one C2 UNet pass as it runs today (fused skip GEMMs, zero rows, GroupNorm producer statistics)
timeout 200 python tools/dump_unet_shapes.py > $O/dump_shapes.log 2>&1; cp profiles/unet_c2_gemm_shapes.txt $O/
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 120 $S --replay-time profiles/unet_c2_gemm_shapes.txt > $O/replay_time.log 2>&1; tail -1 $O/replay_time.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-400 $O/bench_c2.json
for c in c3 c5 c4; do
  timeout 200 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-260 $O/bench_$c.json
done
echo "benches done after $(( $(date +%s) - T0 )) s"
PFD_PROFILE_CONFIGS="" timeout 420 bash tools/profile_round.sh r04_final/prof r04 > $O/profile_round.log 2>&1
echo "profile_round done after $(( $(date +%s) - T0 )) s"
( cd prompt-free-diffusion_amd/csrc && timeout 200 ./build/selftest > ../../$O/selftest.log 2>&1; echo "selftest rc=$?"; tail -1 ../../$O/selftest.log
  timeout 100 ./build/selftest --r4 > ../../$O/selftest_r4.log 2>&1; echo "selftest --r4 rc=$?"; tail -1 ../../$O/selftest_r4.log )
PFD_PROFILE_CONFIGS="c3 c5" timeout 300 bash -c 'cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; for c in c3 c5; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_final/prof/kt_$c -o kt -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/r04_final/prof/kt_bench_$c.json 2> $R/gpurun_out/r04_final/prof/kt_$c.log; python $R/tools/rocpd_stats.py $(find $R/gpurun_out/r04_final/prof/kt_$c -name "*results.db" | head -1) $R/gpurun_out/r04_final/prof/r04_rocprof_kernel_stats_$c.md > /dev/null 2>&1; done; find $R/gpurun_out/r04_final -name "*results.db" -size +20M -delete'
echo "all done after $(( $(date +%s) - T0 )) s"; du -sh $O
