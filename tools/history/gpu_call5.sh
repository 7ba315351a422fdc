#!/bin/bash
set -u
O=gpurun_out/r04_call5; mkdir -p $O
run() { env "$@" timeout 300 python tools/determinism_check.py 2>/dev/null | tee -a $O/determinism.log; }
run PFD_X=1
run PFD_GN_PSTATS=0
run PFD_GN_PSTATS=0 PFD_PATCH_RING=0 PFD_WS_RING=0
run PFD_GN_PSTATS=0 PFD_GEMM_FUSE=0
run PFD_GN_PSTATS=0 PFD_PATCH_RING=0 PFD_WS_RING=0 PFD_GEMM_FUSE=0
