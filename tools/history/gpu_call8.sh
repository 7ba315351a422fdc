#!/bin/bash
for v in vC; do for r in 1 2; do PFD_HIP_LIB=$PWD/variants/$v/p/libpfd_hip.so timeout 300 python tools/determinism_gemm.py 2>/dev/null | grep "ln=True" | sed "s/^/$v: /"; done; done
