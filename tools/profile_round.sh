#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats of the bench command + PMC passes (HBM traffic, SQ) over the
# torch-free replay of the UNet launch list.   usage: tools/profile_round.sh <out dir under gpurun_out> <tag>
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$1; TAG=$2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
S=$REPO/prompt-free-diffusion_amd/csrc/build/selftest
L=$REPO/profiles/unet_c2_gemm_shapes.txt
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $OUT/kt_bench.json 2> $OUT/kt.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- $S --replay $L > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- $S --replay $L > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_sq -o q -- $S --replay $L > $OUT/pmc_sq.log 2>&1
# attention / GroupNorm / LayerNorm: HBM traffic and GB/s from PMC passes over the torch-free kernel benches
for w in attn gn; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$w -o f -- $S --bench-$w > $OUT/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$w -o w -- $S --bench-$w > $OUT/pmc_write_$w.log 2>&1
done
# round 6: SQ counters of the attention kernels (two passes: 8 SQ slots each) over the torch-free attention bench
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_sq_attn1 -o q -- $S --bench-attn > $OUT/pmc_sq_attn1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq_attn2 -o q -- $S --bench-attn > $OUT/pmc_sq_attn2.log 2>&1
# kernel stats of the other BASELINE configs (C3 ControlNet + SeeCoder-PA, C5 768^2)
for c in ${PFD_PROFILE_CONFIGS-c3 c5}; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$c -o kt -- python $REPO/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $OUT/kt_bench_$c.json 2> $OUT/kt_$c.log
done
cd $REPO
for c in ${PFD_PROFILE_CONFIGS-c3 c5}; do
  python tools/rocpd_stats.py $(find $OUT/kt_$c -name '*results.db' | head -1) $OUT/${TAG}_rocprof_kernel_stats_$c.md > /dev/null 2>&1
done
python tools/rocpd_stats.py $(find $OUT/kt -name '*results.db' | head -1) $OUT/${TAG}_rocprof_kernel_stats.md > /dev/null 2>&1
python tools/pmc_bucket.py $(find $OUT/pmc_fetch -name '*results.db' | head -1) $(find $OUT/pmc_write -name '*results.db' | head -1) $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic_per_bucket.md > $OUT/pmc_bucket.log 2>&1
python tools/pmc_sq.py $(find $OUT/pmc_sq -name '*results.db' | head -1) $OUT/${TAG}_pmc_sq_gemm.md > /dev/null 2>&1
for w in attn gn; do
  python tools/pmc_kernels.py $(find $OUT/pmc_fetch_$w -name '*results.db' | head -1) $(find $OUT/pmc_write_$w -name '*results.db' | head -1) $OUT/${TAG}_pmc_hbm_$w.md $OUT/pmc_traffic.json > $OUT/pmc_kernels_$w.log 2>&1
done
{ echo "<!-- SQ counters of the attention kernels over selftest --bench-attn (two rocprofv3 --pmc passes; GRBM_GUI_ACTIVE is summed over the 8 XCDs) -->"
  python tools/pmc_dump.py $(find $OUT/pmc_sq_attn1 -name '*results.db' | head -1) "" attention
  echo
  python tools/pmc_dump.py $(find $OUT/pmc_sq_attn2 -name '*results.db' | head -1) "" attention; } > $OUT/${TAG}_pmc_sq_attn.md 2> $OUT/pmc_sq_attn_dump.log
find $OUT -name '*results.db' -size +20M -delete    # keep gpurun_out under its 64 MiB cap
ls -la $OUT
