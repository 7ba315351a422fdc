#!/usr/bin/env python
"""Build the CPU emulation of csrc/gemm_glds.hip, csrc/norm.hip and csrc/attention.hip (tools/cpu_emu/emu_*.cpp): write
<file>_emu.inc = the kernel file with its gfx950 inline-asm statements replaced by their C meaning, then compile the
drivers for the host with clang++.
usage: [EMU_DEFINES="-DPFD_FAST_PROLOGUE ..."] build.py [outdir]   (default /tmp/pfd_cpu_emu)  -> <outdir>/emu_gemm, <outdir>/emu_norm, <outdir>/emu_attn"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "prompt-free-diffusion_amd", "csrc")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")

SUBST = [
    # a copy lands when it is issued; the wait is still a lockstep point of the wave (see hip_runtime.h)
    (r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', 'emu::wave_sync();'),
    # register pins / opaque values
    (r'asm volatile\("" : "\+v"\(([^;]*?)\)\);', '((void)0);'),
    (r'asm volatile\("" : "\+v"\(r\), "\+v"\(ch\)\);', '((void)0);'),
    (r'asm volatile\("" : "\+s"\(k\)::"memory"\);', '((void)0);'),
    (r'(?s)asm volatile\("" ::"s"\(p\.tiles_m\).*?\);', '((void)0);'),   # the argument-load batch at kernel entry (PFD_ARG_BATCH_*)
    (r'asm volatile\("" ::"v"\(t\)\);', '((void)0);'),
    # the opaque 16-byte LDS store of the GroupNorm prologue
    (r'asm volatile\("ds_write_b128 %0, %1" ::"v"\(addr\), "v"\(d\) : "memory"\);', 'memcpy(lds_dst, &d, 16); (void)addr;'),
]


def preprocess(name, out, required=True):
    src = open(os.path.join(CSRC, name + ".hip")).read()
    for pat, rep in SUBST:
        src, n = re.subn(pat, rep, src)
    # (the PFD_ARG_BATCH_* macro bodies are compiled only outside the emulation: #if ... && !defined(PFD_CPU_EMU))
    left = [l for l in src.splitlines() if "asm volatile" in l and 'asm volatile("" :::' not in l and '"s"((p).tiles_m)' not in l]
    if left:
        sys.exit(f"build.py: inline asm in {name}.hip the emulation does not translate:\n" + "\n".join(left))
    with open(os.path.join(out, name + "_emu.inc"), "w") as f:
        f.write(src)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pfd_cpu_emu"
    os.makedirs(out, exist_ok=True)
    procs = []
    only = os.environ.get("EMU_ONLY", "").split()     # e.g. EMU_ONLY=emu_gemm: just that driver
    for name, driver in (("gemm_glds", "emu_gemm"), ("norm", "emu_norm"), ("attention", "emu_attn")):
        if only and driver not in only:
            continue
        preprocess(name, out)
        exe = os.path.join(out, driver)
        cmd = [CXX, "-std=c++17", os.environ.get("EMU_OPT", "-O0"), "-pthread", "-w", f"-I{HERE}", f"-I{out}", f"-I{REPO}/include", f"-I{CSRC}",
               os.path.join(HERE, driver + ".cpp"), "-o", exe] + os.environ.get("EMU_DEFINES", "").split()
        procs.append((exe, cmd, subprocess.Popen(cmd)))     # the three drivers compile side by side
    for exe, cmd, p in procs:
        if p.wait() != 0:
            sys.exit("build.py: " + " ".join(cmd) + " failed")
        print(exe)


if __name__ == "__main__":
    main()
