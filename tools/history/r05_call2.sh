#!/bin/bash
# Round 5, GPU call 2: end-to-end A/B of every candidate that has a switch, on ONE box, alternating with the base (boxes
# differ by +-8 %).  Pass 1: base, each candidate once, base again; pass 2 (adaptive): base + every candidate that
# beat the mean base of pass 1 by >= 0.4 %.  Decisions (adopt as default / delete) are taken from gpurun_out/r05_call2/summary.txt.
#   usage (on the GPU box): bash tools/r05_call2.sh [candidate tags...]   -> gpurun_out/r05_call2/
set -u
O=gpurun_out/r05_call2; mkdir -p $O
B="--steps 5 --warmup 2 --no-cpu-baseline --no-prof"
T0=$(date +%s)
declare -A ENVS
ENVS[base]="PFD_AREG=0"
ENVS[areg7]="PFD_AREG=7"
ENVS[areg24]="PFD_AREG=24"
ENVS[areg31]="PFD_AREG=31"
ENVS[attn7]="PFD_ATTN=7"
ENVS[attn8]="PFD_ATTN=8"
ENVS[gnpar]="PFD_GN_PAR=1"
ENVS[gnfast]="PFD_GN_SMALL_FAST=1"
ENVS[wpf]="PFD_WPREFETCH=1"
ENVS[fastpro]="PFD_HIP_LIB=$PWD/prompt-free-diffusion_amd/csrc/build_fast/libpfd_hip_fast.so"
CANDS=${@:-areg7 areg24 attn7 attn8 gnpar gnfast wpf fastpro}
run() {   # run <tag> <suffix>
  local tag=$1 sfx=$2
  env ${ENVS[$tag]} timeout 170 python bench.py $B > $O/${tag}_$sfx.json 2> $O/${tag}_$sfx.err
  echo "${tag}_$sfx rc=$? after $(( $(date +%s) - T0 )) s: $(python - <<P
import json
try:
    d = json.load(open("$O/${tag}_$sfx.json")); print("%.1f ms per batch, loop %s" % (d["ms_per_step"], d.get("stage_ms_per_batch", {}).get("ddim_loop_ms")))
except Exception as e:
    print("no result:", str(e)[:80])
P
)"
}
# compile-time candidate: the same sources built with -DPFD_FAST_PROLOGUE into a second library (PFD_HIP_LIB selects it)
if [[ " $CANDS " == *" fastpro "* ]]; then
( cd prompt-free-diffusion_amd/csrc && mkdir -p build_fast && for f in capi gemm_conv gemm_glds attention swin_attn norm elementwise; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DPFD_FAST_PROLOGUE -c $f.hip -o build_fast/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_fast/*.o -o build_fast/libpfd_hip_fast.so ) > $O/build_fast.log 2>&1
echo "fast-prologue library built after $(( $(date +%s) - T0 )) s: $(ls -la prompt-free-diffusion_amd/csrc/build_fast/libpfd_hip_fast.so 2>&1 | cut -c1-120)"
fi
S=prompt-free-diffusion_amd/csrc/build/selftest
timeout 330 $S --r5 > $O/selftest_r5.log 2>&1; echo "selftest --r5 rc=$? after $(( $(date +%s) - T0 )) s: $(tail -1 $O/selftest_r5.log)"; grep FAIL $O/selftest_r5.log | head
run base 1a
for c in $CANDS; do run $c 1; done
run base 1b
python - $O $CANDS > $O/pass1.txt <<'P'
import json, sys
O, cands = sys.argv[1], sys.argv[2:]
def ms(tag):
    try: return json.load(open(f"{O}/{tag}.json"))["ms_per_step"]
    except Exception: return None
b = [v for v in (ms("base_1a"), ms("base_1b")) if v]
base = sum(b) / len(b) if b else None
print("base", b)
keep = []
for c in cands:
    v = ms(c + "_1")
    if v and base:
        print(f"{c:10s} {v:8.1f} ms  {100 * (v / base - 1):+.2f} %")
        if v < base * 0.996: keep.append(c)
    else:
        print(f"{c:10s} no result")
print("RERUN", " ".join(keep))
P
cat $O/pass1.txt
KEEP=$(grep '^RERUN' $O/pass1.txt | cut -d' ' -f2-)
if [ -n "$KEEP" ]; then
  for c in $KEEP; do
    if [ $(( $(date +%s) - T0 )) -gt ${PFD_CALL_BUDGET_S:-1350} ]; then echo "time budget reached: no second run for $c"; continue; fi
    run $c 2; run base 2_$c
  done
fi
python - $O $CANDS > $O/summary.txt <<'P'
import json, sys, glob, os
O, cands = sys.argv[1], sys.argv[2:]
def ms(path):
    try: return json.load(open(path))["ms_per_step"]
    except Exception: return None
bases = sorted(v for v in (ms(p) for p in glob.glob(f"{O}/base_*.json")) if v)
print("base runs (ms per batch):", [round(v, 1) for v in bases])
base = sum(bases) / len(bases)
for c in cands:
    vs = [v for v in (ms(p) for p in sorted(glob.glob(f"{O}/{c}_[12].json"))) if v]
    if vs:
        m = sum(vs) / len(vs)
        print(f"{c:10s} {[round(v, 1) for v in vs]}  mean {m:.1f} ms  vs base {base:.1f}: {100 * (m / base - 1):+.2f} %")
P
cat $O/summary.txt
echo "call 2 done after $(( $(date +%s) - T0 )) s"
