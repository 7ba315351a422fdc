#!/bin/bash
# Round 6, call 2: attention3_kernel (software-pipelined d = 40 attention, 64 queries per wave) -- correctness on hardware, the
# micro-bench old vs new, SQ counters of the new kernel, end-to-end A/B (PFD_ATTN3=0/1 alternating on this box).
set -u
O=gpurun_out/r06_call2; mkdir -p $O
REPO=$(pwd)
S=$REPO/prompt-free-diffusion_amd/csrc/build/selftest
PFD_ATTN3_FORCE=1 timeout 300 $S --attn > $O/selftest_attn_force.log 2>&1; echo "selftest --attn (forced attention3) rc=$?"; grep -E "FAIL|SELFTEST" $O/selftest_attn_force.log | head
timeout 300 $S --attn > $O/selftest_attn.log 2>&1; echo "selftest --attn rc=$?"; grep -E "FAIL|SELFTEST" $O/selftest_attn.log | head
PFD_ATTN3=0 timeout 120 $S --bench-attn > $O/bench_attn_old.log 2>&1; cat $O/bench_attn_old.log
timeout 120 $S --bench-attn > $O/bench_attn_new.log 2>&1; cat $O/bench_attn_new.log
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "attention" -p no:cacheprovider > $O/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -2 $O/pytest_attn.log
for i in 1 2; do
  for a in 0 1; do
    PFD_ATTN3=$a timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $O/bench_a${a}_$i.json 2> $O/bench_a${a}_$i.err
    echo "PFD_ATTN3=$a run $i: $(python -c "import json,sys; d=json.load(open('$O/bench_a${a}_$i.json')); print(d['ms_per_step'], d['value'])")"
  done
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $REPO/$O/pmc1 -o q -- $S --bench-attn > $REPO/$O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $REPO/$O/pmc2 -o q -- $S --bench-attn > $REPO/$O/pmc2.log 2>&1
cd $REPO
python tools/pmc_dump.py $(find $O/pmc1 -name "*results.db" | head -1) $O/pmc_attn_pass1.md "attention"
python tools/pmc_dump.py $(find $O/pmc2 -name "*results.db" | head -1) $O/pmc_attn_pass2.md "attention"
find $O -name '*results.db' -delete
