"""CPU (hipcc cross-compiles gfx950 without a GPU): the generated ISA of the hot kernels must stay free of the two
compiler behaviours that cost this code base +8 % / a dead operand ring when they went unnoticed:
global loads serialised behind `s_waitcnt vmcnt(0)` because they were written inside an `if`, LDS-DMA rings drained by a
`__syncthreads()` fence, and register spills."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="hipcc not available")


def test_hot_kernels_have_no_serialised_loads_no_drained_rings_no_scratch():
    import isa_audit
    bad, rows = isa_audit.findings()
    assert not bad, "\n".join(bad)
    names = " ".join(k for _, k, _ in rows)
    # the audit really saw the kernels it is about (a renamed kernel must not silently drop out of it)
    for must in ("conv3x3_patch_ws_kernel", "gemm160ws_kernel", "splitk_reduce_kernel", "gn_stats_kernel",
                 "layernorm_rows_kernel", "gemm_conv_kernel", "splitk_reduce_gnorm_kernel", "splitk_reduce_gn_kernel"):
        assert must in names, must
    rings = [k for _, k, v in rows if v["ring"]]
    assert len(rings) >= 4, rings          # 64-row / 128-row rings, linear + conv
    assert all(v["loads"] > 0 for _, k, v in rows if "gemm160" in k)
