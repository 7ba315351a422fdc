#!/bin/bash
# Second GPU call of the next round (prepared at the end of round 4, never run) -- ONLY after tools/r05_first_call.sh showed
# `selftest --r5` green: end-to-end A/B of the candidates that have an environment switch, alternating on ONE box
# (boxes differ by +-8 %), then the parity suites under the winning switches.
#   PFD_AREG=7   ring kernels with the activation fragments in registers (bit mask: 1 = 27 for 23, 2 = 45 for 43, 4 = 85 for 83)
#   PFD_AREG=31  ... and 86 for 82 (8), 28 for 22 (16): the 2-stage tiles -> 3 weight stages, two blocks per CU, two K tiles in flight each
#   PFD_ATTN=7   d = 40 attention with s_setprio around the MFMA clusters
#   PFD_ATTN=8   every attention launch clears its LDS image with 16-byte stores (the plain form: a rolled loop of ds_write_b16, ~2 us at d = 160)
#   PFD_GN_PAR=1 GroupNorm apply from producer statistics: the partials of eight slabs requested before the first add
#   PFD_WPREFETCH=1  every GEMM / conv weight matrix is read on a side stream one launch ahead of its consumer (warm instead of
#                    cold weight tiles for the latency-chain launches); parallel branches in the captured graph
#   PFD_GN_SMALL_FAST=1  single-launch GroupNorm of the 8^2 / 16^2 levels without per-chunk divisions / gamma-beta round trips
#   usage (on the GPU box): bash tools/r05_second_call.sh   -> gpurun_out/r05_e2e/
set -u
O=gpurun_out/r05_e2e; mkdir -p $O
B="--steps 6 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
run() {   # run <tag> <env assignments...>
  local tag=$1; shift
  env "$@" timeout 150 python bench.py $B > $O/$tag.json 2> $O/$tag.err
  echo "$tag done after $(( $(date +%s) - T0 )) s"
}
for rep in 1 2; do
  run base_$rep PFD_AREG=0
  run areg_$rep PFD_AREG=7
  run areg2_$rep PFD_AREG=31
  run attn7_$rep PFD_ATTN=7
  run attn8_$rep PFD_ATTN=8
  run gn_$rep PFD_GN_PAR=1 PFD_GN_SMALL_FAST=1
  run wpf_$rep PFD_WPREFETCH=1
  run all_$rep PFD_AREG=31 PFD_ATTN=7 PFD_GN_PAR=1 PFD_GN_SMALL_FAST=1
done
for f in base_1 areg_1 areg2_1 attn7_1 attn8_1 gn_1 wpf_1 all_1 base_2 areg_2 areg2_2 attn7_2 attn8_2 gn_2 wpf_2 all_2; do python - <<P
import json
try:
    d = json.load(open("$O/$f.json")); print("%-8s %7.1f ms per batch  %.3f images/s  loop %s" % ("$f", d["ms_per_step"], d["value"], d.get("stage_ms_per_batch", {}).get("ddim_loop_ms")))
except Exception as e:
    print("$f", "no result:", e)
P
done
# compile-time candidate: the same source built with -DPFD_FAST_PROLOGUE into a second library (PFD_HIP_LIB selects it)
( cd prompt-free-diffusion_amd/csrc && mkdir -p build_fast && for f in capi gemm_conv gemm_glds attention swin_attn norm elementwise; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DPFD_FAST_PROLOGUE -c $f.hip -o build_fast/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_fast/*.o -o build_fast/libpfd_hip_fast.so ) > $O/build_fast.log 2>&1
for rep in 1 2; do
  run base2_$rep PFD_AREG=0
  run fastpro_$rep PFD_HIP_LIB=$PWD/prompt-free-diffusion_amd/csrc/build_fast/libpfd_hip_fast.so
done
for f in base2_1 fastpro_1 base2_2 fastpro_2; do python - <<P
import json
try:
    d = json.load(open("$O/$f.json")); print("%-10s %7.1f ms per batch  %.3f images/s" % ("$f", d["ms_per_step"], d["value"]))
except Exception as e:
    print("$f", "no result:", e)
P
done
# parity under the switches (kernel-level suite + the C2 trajectory on the fixture-backed oracle)
PFD_AREG=31 PFD_ATTN=7 PFD_GN_PAR=1 PFD_GN_SMALL_FAST=1 timeout 400 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels_fullsize.py -m gpu -q -x > $O/pytest_switches.log 2>&1
echo "pytest (all four switches) rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_switches.log
PFD_AREG=31 PFD_ATTN=7 PFD_GN_PAR=1 PFD_GN_SMALL_FAST=1 timeout 200 python -m pytest tests/test_hip_trajectory.py -m gpu -q -s -x -k c2 > $O/pytest_trajectory_switches.log 2>&1
echo "pytest trajectory rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_trajectory_switches.log
