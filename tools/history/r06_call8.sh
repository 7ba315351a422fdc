#!/bin/bash
# Round 6, call 8: which GroupNorms of a C2 step still run a statistics pass?
set -u
O=gpurun_out/r06_call8; mkdir -p $O
timeout 600 python tools/gn_paths.py $O/gn_paths.log 2>&1 | tail -40
