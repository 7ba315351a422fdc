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
                 "layernorm_rows_kernel", "gemm_conv_kernel", "gemm160ar_kernel", "conv3x3_patch_fl_kernel"):
        assert must in names, must
    rings = [k for _, k, v in rows if v["ring"]]
    assert len(rings) >= 4, rings          # 64-row / 128-row rings, linear + conv
    assert all(v["loads"] > 0 for _, k, v in rows if "gemm160" in k)


def test_register_operand_ring_kernels_keep_only_their_counted_wait():
    """forced variants 27 / 45 / 85 (gemm160ar_kernel): the K loop's only VM waits are the hand-written counted wait and the
    tail drain -- hipcc must not have added a wait of its own for the activation registers (they are loaded by asm
    statements it does not count), which would drain the 6-deep ring once per trip"""
    import isa_audit
    res = isa_audit.areg_loop_waits(isa_audit.compile_asm(os.path.join(isa_audit.CSRC, "gemm_glds.hip")))
    assert len(res) == 10, sorted(res)   # four ring tiles x {linear, implicit-GEMM convolution} + the 3-stage 128- and 64-row linears
    for k, (keep, bad, n_keep, nbuf) in res.items():
        assert not bad, (k, bad)
        assert n_keep == nbuf - 1 or n_keep == nbuf, (k, n_keep)   # one counted wait per unrolled step
        assert 0 < keep < 64
    # and nothing but the consuming MFMAs reads a register one of those loads wrote (no compiler copy / spill of a value
    # that may not have landed yet)
    hyg = isa_audit.areg_register_hygiene(isa_audit.compile_asm(os.path.join(isa_audit.CSRC, "gemm_glds.hip")))
    assert sorted(hyg) == sorted(res)
    for k, (bad, nring) in hyg.items():
        assert not bad, (k, bad[:4])
        assert nring in (48, 56, 80, 112), (k, nring)   # ring stages x row blocks x 2 K halves x 4 registers
