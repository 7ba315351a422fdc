#!/usr/bin/env python
"""Build the CPU emulation of csrc/gemm_glds.hip (tools/cpu_emu/emu_gemm.cpp): write gemm_glds_emu.inc = the kernel file
with its gfx950 inline-asm statements replaced by their C meaning, then compile for the host with clang++.
usage: build.py [outdir]   (default /tmp/pfd_cpu_emu)  -> <outdir>/emu_gemm"""
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
    (r'asm volatile\("" : "\+s"\(k\)::"memory"\);', '((void)0);'),
    # the uncounted activation load of gemm160ar_kernel
    (r'asm volatile\("global_load_dwordx4 %0, %1, off" : "=v"\(d\) : "v"\(\(const __attribute__\(\(address_space\(1\)\)\) void\*\)src\) : "memory"\);',
     'memcpy(&d, src, 16);'),
    # the opaque 16-byte LDS store of the GroupNorm prologue
    (r'asm volatile\("ds_write_b128 %0, %1" ::"v"\(addr\), "v"\(d\) : "memory"\);', 'memcpy(lds_dst, &d, 16); (void)addr;'),
]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pfd_cpu_emu"
    os.makedirs(out, exist_ok=True)
    src = open(os.path.join(CSRC, "gemm_glds.hip")).read()
    for pat, rep in SUBST:
        src, n = re.subn(pat, rep, src)
        if n == 0:
            sys.exit(f"build.py: pattern not found any more: {pat}")
    left = [l for l in src.splitlines() if "asm volatile" in l and 'asm volatile("" :::' not in l.replace("  ", " ")]
    if left:
        sys.exit("build.py: inline asm the emulation does not translate:\n" + "\n".join(left))
    with open(os.path.join(out, "gemm_glds_emu.inc"), "w") as f:
        f.write(src)
    exe = os.path.join(out, "emu_gemm")
    cmd = [CXX, "-std=c++17", "-O1", "-pthread", "-w", f"-I{HERE}", f"-I{out}", f"-I{REPO}/include", f"-I{CSRC}",
           os.path.join(HERE, "emu_gemm.cpp"), "-o", exe]
    subprocess.run(cmd, check=True)
    print(exe)


if __name__ == "__main__":
    main()
