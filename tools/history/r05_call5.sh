#!/bin/bash
# Round 5, GPU call 5: A/B of the residual-tile touch (second library built without it), the XCD-ordered / deeper fused
# reduction + GroupNorm against PFD_GNF=0, and the split cap of the 8 x 8 patch tiles.
set -u
O=gpurun_out/r05_call5; mkdir -p $O
B="--steps 5 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
( cd prompt-free-diffusion_amd/csrc && mkdir -p build_nt && for f in capi gemm_conv gemm_glds attention swin_attn norm elementwise; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DPFD_RES_TOUCH=0 -c $f.hip -o build_nt/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_nt/*.o -o build_nt/libpfd_hip_nt.so ) > $O/build_nt.log 2>&1
NT=$PWD/prompt-free-diffusion_amd/csrc/build_nt/libpfd_hip_nt.so
echo "no-touch library after $(( $(date +%s) - T0 )) s: $(ls -la $NT 2>&1 | cut -c1-100)"
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/selftest_r5.log)"; grep FAIL $O/selftest_r5.log | head
PFD_PATCH8_SPLITS=16 timeout 300 $S --r5 > $O/selftest_r5_cap16.log 2>&1; echo "selftest --r5 (cap 16) rc=$?: $(tail -1 $O/selftest_r5_cap16.log)"; grep FAIL $O/selftest_r5_cap16.log | head -5
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
  run notouch_$rep PFD_HIP_LIB=$NT
  run gnf0_$rep PFD_GNF=0
  run p8off_$rep PFD_PATCH8=0
  run p8cap10_$rep PFD_PATCH8_SPLITS=10
  run p8cap16_$rep PFD_PATCH8_SPLITS=16
  run p8off_gnf0_$rep PFD_PATCH8=0 PFD_GNF=0
done
run head_3 PFD_R5X=0
