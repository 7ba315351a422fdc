#!/bin/bash
# Round 5, GPU call 4: where do the two new paths (fused reduction + GroupNorm, 8 x 8 patch tiles) spend their time?
# rocprofv3 kernel statistics of a short bench under each switch + the cold per-problem replay with / without the 8 x 8 tiles.
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
O=$REPO/gpurun_out/r05_call4; mkdir -p $O
S=$REPO/prompt-free-diffusion_amd/csrc/build/selftest
L=$REPO/profiles/unet_c2_gemm_shapes.txt
T0=$(date +%s)
for t in 1 0; do PFD_PATCH8=$t timeout 90 $S --replay-time $L > $O/replay_patch8_$t.log 2>&1; echo "replay PFD_PATCH8=$t: $(tail -1 $O/replay_patch8_$t.log)"; done
cd /tmp; export TMPDIR=/tmp
prof() {  # prof <tag> <env...>
  local tag=$1; shift
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o kt -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/kt_$tag.json 2> $O/kt_$tag.log
  python $REPO/tools/rocpd_stats.py $(find $O/kt_$tag -name '*results.db' | head -1) $O/stats_$tag.md > /dev/null 2>&1
  echo "$tag done after $(( $(date +%s) - T0 )) s: $(head -c 300 $O/kt_$tag.json | cut -c1-200)"
}
prof head PFD_R5X=0
prof gnf0 PFD_GNF=0
prof patch8off PFD_PATCH8=0
find $O -name '*results.db' -delete
ls $O
