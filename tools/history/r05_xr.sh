#!/bin/bash
set -u
O=gpurun_out/r05_xr; mkdir -p $O
B="--steps 5 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
build() {  # build <dir> <defines...>
  local dir=$1; shift
  ( cd prompt-free-diffusion_amd/csrc && mkdir -p $dir && for f in capi gemm_conv gemm_glds attention swin_attn norm elementwise; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form "$@" -c $f.hip -o $dir/$f.o & done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $dir/*.o -o $dir/lib.so ) > $O/build_$dir.log 2>&1
}
build build_x1 -DPFD_XA_NK=40 &
build build_x2 -DPFD_XH_NK=32 &
build build_x3 -DPFD_XG_M=512 &
wait
L=$PWD/prompt-free-diffusion_amd/csrc
echo "libraries after $(( $(date +%s) - T0 )) s: $(ls $L/build_x1/lib.so $L/build_x2/lib.so $L/build_x3/lib.so 2>&1 | wc -l)"
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
  run base_$rep PFD_QUIET=1
  run x1_$rep PFD_HIP_LIB=$L/build_x1/lib.so
  run x2_$rep PFD_HIP_LIB=$L/build_x2/lib.so
  run x3_$rep PFD_HIP_LIB=$L/build_x3/lib.so
done
