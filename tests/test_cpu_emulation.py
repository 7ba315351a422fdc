"""CPU emulation of the wide-tile GEMM / convolution kernels (tools/cpu_emu): the kernel file csrc/gemm_glds.hip is compiled
for the HOST against a shim of the HIP runtime -- one OS thread per GPU thread of a block, MFMA / shuffles / ballot as
rendezvous of a wave's 64 threads, LDS as block-shared statics, real barriers -- and the library's own dispatcher
(pfd_gemm160_try, forced variants, split-K) runs small problems through it.  Checked against a double-precision reference,
and the round-5 candidates that have never run on hardware (register-operand rings 27 / 45 / 85 / 29 / 86 / 28, the patch
kernel that hands over through LDS progress words, 95) bit for bit against the hardware-validated kernels they would
replace (23 / 43 / 83 / 82 / 22 / 98).  What the model cannot see: s_waitcnt counts (tests/test_ring_protocol.py and
tests/test_isa_audit.py cover those), register allocation, timing."""
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


def test_groupnorm_candidates_match_the_plain_kernels_bit_for_bit(emu):
    """csrc/norm.hip on the same emulation: PFD_GN_SMALL_FAST=1 and PFD_GN_PAR=1 (round-5 candidates, never run on hardware)
    against the plain kernels and a double-precision GroupNorm"""
    r = subprocess.run([os.path.join(os.path.dirname(emu), "emu_norm"), "--quick"], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert r.returncode == 0 and len(lines) == 5, r.stdout[-3000:] + r.stderr[-1000:]
    assert all(l.startswith("ok") and "== plain form bitwise" in l for l in lines), "\n".join(lines)


def _run(emu, *filters):
    r = subprocess.run([emu, *filters], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert r.returncode == 0 and lines and not any(l.startswith("FAIL") for l in lines), r.stdout[-3000:] + r.stderr[-1000:]
    return lines


def test_emulation_reproduces_the_hardware_validated_kernels(emu):
    lines = _run(emu, "variant 23 (", "variant 83 (", "variant 98", "variant 96", "variant 99")
    assert len(lines) == 6                                               # (96 also with K-tile-contiguous weights)
    assert sum("== variant 98 bitwise" in l for l in lines) == 2      # 3-stage ring and 8-wave forms of the patch kernel


def test_register_operand_ring_kernels_match_the_lds_ring_kernels_bit_for_bit(emu):
    # (a subset of tools/cpu_emu/emu_gemm's list -- run the binary without arguments for all of it)
    lines = _run(emu, "variant 27 seven", "variant 27 two-source", "variant 27 zero rows", "variant 27 K-tile", "variant 45 (8",
                 "variant 85 (128", "variant 29", "variant 86", "variant 28", "variant 26", "variant 85 conv 3x3 s1",
                 "variant 45 conv", "variant 27 conv 3x3 +", "variant 85 conv 3x3 split")
    assert len(lines) == 14
    assert all("bitwise" in l for l in lines), "\n".join(lines)


def test_statistics_emitting_splitk_reduce_and_its_three_sweep_form(emu):
    """split-K reduction that also emits the GroupNorm statistics of what it stores: statistics == sums of the stored f16 values,
    and PFD_GN_PAR=1 (three row sweeps of loads in flight; round-5 candidate) gives the same output and statistics bit for bit"""
    lines = _run(emu, "statistics")
    assert len(lines) == 2 and all("PFD_GN_PAR=1: same output and statistics bitwise" in l for l in lines), "\n".join(lines)


def test_flag_handover_patch_kernel_matches_the_barrier_form(emu):
    lines = _run(emu, "variant 95")
    assert len(lines) == 2 and all("== variant 98 bitwise" in l for l in lines)


def test_attention_kernels_and_the_setprio_candidate(emu):
    """csrc/attention.hip on the same emulation (v_mfma_f32_32x32x16_f16, v_permlane16_swap): the default stage (PFD_ATTN=6)
    against a double-precision softmax(Q K^T) V at d = 40 / 80 / 96 / 160 with ragged query / key counts, and the round-5
    candidates PFD_ATTN=7 (s_setprio around the MFMA clusters) and PFD_ATTN=8 (16-byte clears of the LDS image), never run on
    hardware: the same bits as stage 6"""
    exe = os.path.join(os.path.dirname(emu), "emu_attn")
    out = {}
    for mode, extra in (("6", []), ("7", ["--quick"]), ("8", []), ("6w4", ["--quick", "--w4"]), ("8w4", ["--quick", "--w4"])):
        r = subprocess.run([exe, mode[0]] + extra, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
        assert r.returncode == 0 and lines and all(l.startswith("ok") for l in lines), r.stdout[-2000:] + r.stderr[-1000:]
        out[mode] = [l.split()[-1] for l in lines]
    assert len(out["6"]) == 5 and len(out["7"]) == 2
    assert out["7"] == out["6"][:2]
    # PFD_ATTN=8 (16-byte clears of the LDS image, every head dim; round-5 candidate): the bits of stage 6, 8-wave and 4-wave forms
    assert out["8"] == out["6"] and out["8w4"] == out["6w4"] and len(out["8w4"]) == 2


def test_areg_mask_redirects_the_automatic_choice(emu):
    """PFD_AREG=<mask>: the automatic tile choice launches the register-operand kernel exactly where the mask says (ring picks:
    bits 1 / 2 / 4, two-stage picks: bits 8 / 16) and the LDS kernels otherwise (default 0)"""
    def picks(mask):
        env = dict(os.environ, PFD_AREG=str(mask))
        r = subprocess.run([emu, "--auto"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        return {l.split()[1]: "gemm160ar_kernel" in l for l in r.stdout.splitlines() if l.startswith("auto")}
    assert picks(0) == {"ring23": False, "ring83": False, "two-stage22": False, "two-stage82": False}
    assert picks(31) == {"ring23": True, "ring83": True, "two-stage22": True, "two-stage82": True}
    assert picks(7) == {"ring23": True, "ring83": True, "two-stage22": False, "two-stage82": False}
    assert picks(24) == {"ring23": False, "ring83": False, "two-stage22": True, "two-stage82": True}


def test_fast_prologue_switch_changes_no_result(tmp_path_factory):
    """-DPFD_FAST_PROLOGUE (compile-time round-5 candidate: one argument-load batch at kernel entry, weight row pointers without
    a division per piece): the emulation built WITH the switch gives the same answers -- the K-tile-contiguous weight layout
    through every kernel family is where the pointer form differs"""
    out = str(tmp_path_factory.mktemp("pfd_cpu_emu_fast"))
    env = dict(os.environ, EMU_DEFINES="-DPFD_FAST_PROLOGUE", EMU_ONLY="emu_gemm")
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "cpu_emu", "build.py"), out], check=True, stdout=subprocess.DEVNULL, env=env)
    lines = _run(os.path.join(out, "emu_gemm"), "K-tile-contiguous", "variant 98")
    assert len(lines) == 5
