#!/bin/bash
# Round 6, call 1: baseline of the round-5 tree on this round's first lease -- the whole GPU suite, one bench line,
# attention micro-bench and the SQ counters of the attention kernels.
set -u
O=gpurun_out/r06_call1; mkdir -p $O
REPO=$(pwd)
T0=$(date +%s)
timeout 1300 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.log
S=$REPO/prompt-free-diffusion_amd/csrc/build/selftest
timeout 120 $S --bench-attn > $O/bench_attn.log 2>&1; cat $O/bench_attn.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"; head -c 600 $O/bench_c2.json; echo
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $REPO/$O/pmc_sq_attn -o q -- $S --bench-attn > $REPO/$O/pmc_sq_attn.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $REPO/$O/pmc_sq_attn2 -o q -- $S --bench-attn > $REPO/$O/pmc_sq_attn2.log 2>&1
cd $REPO
python tools/pmc_sq.py $(find $O/pmc_sq_attn -name '*results.db' | head -1) $O/r06_pmc_sq_attn_base.md; cat $O/r06_pmc_sq_attn_base.md
python tools/pmc_dump.py $(find $O/pmc_sq_attn -name "*results.db" | head -1) $O/pmc_attn_pass1.md; python tools/pmc_dump.py $(find $O/pmc_sq_attn2 -name "*results.db" | head -1) $O/pmc_attn_pass2.md
echo "total $(( $(date +%s) - T0 )) s"
