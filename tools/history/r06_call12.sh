S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/unet_c2_gemm_shapes.txt
timeout 200 $S --replay-time $L > gpurun_out/abl_head.log 2>&1; tail -1 gpurun_out/abl_head.log; grep -E "^ *(32768|8192|2048) +(320|640|1280) +(2880|5760|8640|11520|17280|23040) 3 1 0" gpurun_out/abl_head.log | head -12
cp prompt-free-diffusion_amd/libpfd_hip.so /tmp/head.so; cp variants/patch_nobar.so prompt-free-diffusion_amd/libpfd_hip.so
timeout 200 $S --replay-time $L > gpurun_out/abl_nobar.log 2>&1; tail -1 gpurun_out/abl_nobar.log; grep -E "^ *(32768|8192|2048) +(320|640|1280) +(2880|5760|8640|11520|17280|23040) 3 1 0" gpurun_out/abl_nobar.log | head -12
cp /tmp/head.so prompt-free-diffusion_amd/libpfd_hip.so
bash tools/ab_bench.sh gpurun_out/r06_call12 2 head patch_nobar
