#!/bin/bash
# Round 6, call 3: why attention3_kernel is not faster -- timing ablations of its loop and the MFMA / VALU overlap micro-benchmark.
set -u
O=gpurun_out/r06_call3; mkdir -p $O
S=prompt-free-diffusion_amd/csrc/build/selftest
for a in 0 1 2 3 34 50 4 8 12 16 28; do
  echo "ABL=$a: $(PFD_ATTN3_ABL=$a timeout 60 $S --bench-attn 2>&1 | head -1)"
done | tee $O/attn3_ablation.log
timeout 300 tools/ubench/mfma_valu | tee $O/ubench_mfma_valu.log
