S=prompt-free-diffusion_amd/csrc/build/selftest
L=profiles/r06_fixed_cost_probe_shapes.txt
timeout 100 $S --replay-time $L 2>&1 | tail -15 | cut -c1-110
cp prompt-free-diffusion_amd/libpfd_hip.so /tmp/head.so; cp variants/epi_nostore.so prompt-free-diffusion_amd/libpfd_hip.so
echo "---- no output stores"
timeout 100 $S --replay-time $L 2>&1 | tail -15 | cut -c1-110
cp /tmp/head.so prompt-free-diffusion_amd/libpfd_hip.so
