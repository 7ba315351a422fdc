#!/bin/bash
# Round 5, GPU call 7: the statistics-emitting split-K reduction on 1024 threads per (64-row slab, 160-column tile) against
# the same sources built with 256 (-DPFD_RGN_THREADS=256); 8 x 8 patch tiles off in both.
set -u
O=gpurun_out/r05_call7; mkdir -p $O
B="--steps 5 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
( cd prompt-free-diffusion_amd/csrc && mkdir -p build_r256 && for f in capi gemm_conv gemm_glds attention swin_attn norm elementwise; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DPFD_RGN_THREADS=256 -c $f.hip -o build_r256/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_r256/*.o -o build_r256/libpfd_hip_r256.so ) > $O/build_r256.log 2>&1
R256=$PWD/prompt-free-diffusion_amd/csrc/build_r256/libpfd_hip_r256.so
echo "256-thread library after $(( $(date +%s) - T0 )) s: $(ls -la $R256 2>&1 | cut -c1-100)"
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 300 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/selftest_r5.log)"; grep FAIL $O/selftest_r5.log | head
timeout 300 $S > $O/selftest_all.log 2>&1; echo "selftest (all) rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/selftest_all.log)"; grep FAIL $O/selftest_all.log | head
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
  run r1024_$rep PFD_PATCH8=0
  run r256_$rep PFD_PATCH8=0 PFD_HIP_LIB=$R256
done
