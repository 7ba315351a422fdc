#!/bin/bash
set -u
O=gpurun_out/r04_call7; mkdir -p $O
( cd variants/old && timeout 300 python tools/determinism_check.py 2>/dev/null | tee ../../$O/old_check.log; timeout 300 python tools/determinism_layers.py 2>/dev/null | tail -6 | tee ../../$O/old_layers.log )
timeout 300 python tools/determinism_layers.py 2>/dev/null | tail -4 | tee $O/new_layers.log
