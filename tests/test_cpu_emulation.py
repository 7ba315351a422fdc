"""CPU emulation of the wide-tile GEMM / convolution kernels (tools/cpu_emu): the kernel file csrc/gemm_glds.hip is compiled
for the HOST against a shim of the HIP runtime -- one OS thread per GPU thread of a block, MFMA / shuffles / ballot as
rendezvous of a wave's 64 threads, LDS as block-shared statics, real barriers -- and the library's own dispatcher
(pfd_gemm160_try, forced variants, split-K) runs small problems through it.  Checked against a double-precision reference,
and the split-K reduction with the fused GroupNorm (round 5) against the plain reduction + a double-precision GroupNorm.
What the model cannot see: s_waitcnt counts (tests/test_isa_audit.py covers the compiler's), register allocation, timing."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")

pytestmark = pytest.mark.skipif(not os.path.exists(CXX), reason="clang++ of the ROCm toolchain not available")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("pfd_cpu_emu"))
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "cpu_emu", "build.py"), out], check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(out, "emu_gemm")


def test_groupnorm_kernels_on_the_emulation(emu):
    """csrc/norm.hip on the same emulation: the single-launch small-slab GroupNorm and the apply from producer statistics
    (grouped partial loads, adopted in round 5) against a double-precision GroupNorm"""
    r = subprocess.run([os.path.join(os.path.dirname(emu), "emu_norm"), "--quick"], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert r.returncode == 0 and len(lines) == 5, r.stdout[-3000:] + r.stderr[-1000:]
    assert all(l.startswith("ok") for l in lines), "\n".join(lines)


def _run(emu, *filters):
    r = subprocess.run([emu, *filters], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert r.returncode == 0 and lines and not any(l.startswith("FAIL") for l in lines), r.stdout[-3000:] + r.stderr[-1000:]
    return lines


def test_emulation_reproduces_the_hardware_validated_kernels(emu):
    lines = _run(emu, "variant 23 (", "variant 83 (", "variant 98", "variant 96", "variant 99")
    assert len(lines) == 6                                               # (96 also with K-tile-contiguous weights)
    assert sum("== variant 98 bitwise" in l for l in lines) == 2      # 3-stage ring and 8-wave forms of the patch kernel


def test_statistics_emitting_splitk_reduce(emu):
    """split-K reduction that also emits the GroupNorm statistics of what it stores (three row sweeps of loads in flight):
    statistics == sums of the stored f16 values"""
    lines = _run(emu, "statistics")
    assert len(lines) == 2 and all("statistics err" in l for l in lines), "\n".join(lines)


def test_residual_stored_once_for_a_doubled_batch(emu):
    """PfdGemmDesc.res_rows (ABI 9): the store pass, the store pass behind zero rows (the cross-attention re-join of a CFG pair)
    and the plain split-K reduction read residual row m - res_rows for the second half"""
    lines = _run(emu, "residual read with one wrap")
    assert len(lines) == 3, "\n".join(lines)


def test_splitk_reduce_with_fused_groupnorm(emu):
    """PfdGemmDesc.gnf_y (ABI 9, round 5): the split-K reduction whose blocks own (sample, group) slabs and normalise them in
    the same launch -- ring kernel, 4-wave ring with residual and raw tensor kept, patch kernel: raw result bit for bit the
    plain reduction's (or not written at all), normalised tensor == GroupNorm(32) + SiLU of it in double precision, and the
    same request on the unsplit problem declined with nothing written"""
    # (one of the three cases of tools/cpu_emu/emu_gemm here -- a minute of emulation each; `emu_gemm "fused GroupNorm"` runs all,
    #  `selftest --r5` runs the real shapes on hardware)
    lines = _run(emu, "fused GroupNorm: split-K 2 conv 4x4x128")
    assert len(lines) == 1 and all("unsplit request declined, nothing written" in l for l in lines), "\n".join(lines)


def test_attention_kernels_on_the_emulation(emu):
    """csrc/attention.hip on the same emulation (v_mfma_f32_32x32x16_f16, v_permlane16_swap, 16-byte clears of the LDS image):
    the 8-wave d = 40 form and the 4-wave forms against a double-precision softmax(Q K^T) V at d = 40 / 80 / 96 / 160 with
    ragged query / key counts"""
    exe = os.path.join(os.path.dirname(emu), "emu_attn")
    # --a3 (round 6): attention3_kernel, the software-pipelined 64-queries-per-wave d = 40 form, incl. two deferred-rescale cases
    for extra, n in (([], 5), (["--quick", "--w4"], 2), (["--a3"], 7)):
        r = subprocess.run([exe] + extra, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
        assert r.returncode == 0 and len(lines) == n and all(l.startswith("ok") for l in lines), r.stdout[-2000:] + r.stderr[-1000:]
