#!/usr/bin/env python
"""Static audit of the generated gfx950 ISA for the two compiler behaviours that cost this code base the most
(DESIGN.md 3.2), so that they cannot come back unnoticed.  No GPU needed: hipcc -S cross-compiles.

  1. a global load written inside an `if` is emitted as branch + load + `s_waitcnt vmcnt(0)`: however many loads the
     source issues "up front", one is in flight.  Metric: `s_waitcnt vmcnt(0)` within 2 lines of a global load that
     is the ONLY load since the previous VM wait (a wait behind a batch of loads is what we want); more than two of
     those in a kernel is a finding.
  2. `__syncthreads()` next to LDS-DMA drains the DMA queue (`s_waitcnt vmcnt(0)` directly in front of `s_barrier`):
     fatal inside the K loop of a ring kernel (NBUF > 2), where a counted wait must be the last VM wait before the
     barrier.  Metric, ring kernels only: `s_barrier` preceded (within 4 lines) by a `vmcnt(0)` wait more than twice
     (prologue-free kernels have exactly the tail wait of the last K tiles and the one before the epilogue).
  3. scratch (register spills) in any kernel.

usage: isa_audit.py [file.hip ...]   (default: gemm_glds.hip gemm_conv.hip norm.hip)   -> prints one line per kernel, exit 1 on a finding
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "prompt-free-diffusion_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{REPO}/include", f"-I{CSRC}", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only"]
# per-kernel override of the accepted count of single loads waited for immediately (default 2: a lone operand load in
# a prologue is fine).  swin_attn.hip (relative-position-bias gathers) and elementwise.hip (cfg_ddim tables) are not
# in the default file list: once per image / once per step, < 0.1 % of the loop (DESIGN.md work queue)
KNOWN = {}


def compile_asm(src):
    out = os.path.join(tempfile.gettempdir(), "pfd_isa_" + os.path.basename(src) + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src),
                                                               os.path.getmtime(os.path.join(CSRC, "pfd_common.h"))):
        subprocess.run([HIPCC] + FLAGS + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out


def audit(asm_path):
    """-> {kernel: dict(loads, serialised, drained_barriers, scratch, ring)}"""
    res, name = {}, None
    last_load, window, since = -10, [], 0
    for n, line in enumerate(open(asm_path)):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            name = m.group(1)
            res[name] = dict(loads=0, serialised=0, drained_barriers=0, scratch=0,
                             ring=bool(re.search(r"gemm160_kernelILi\dELi\dELb[01]ELi[3-9]E", name)))
            last_load, window, since = -10, [], 0
            continue
        if name is None:
            continue
        r = res[name]
        t = line.strip()
        if re.match(r"global_load_(dwordx\d|dword|ushort|ubyte|short)", t):
            r["loads"] += 1
            last_load = n
            since += 1
        elif t.startswith("s_waitcnt") and "vmcnt(" in t:
            if "vmcnt(0)" in t:
                if n - last_load <= 2 and since == 1:
                    r["serialised"] += 1
                window.append(n)
            since = 0
        elif t.startswith("s_barrier"):
            if window and n - window[-1] <= 4:
                r["drained_barriers"] += 1
        elif t.startswith("scratch_") or "buffer_store_dword" in t and "offen" in t:
            r["scratch"] += 1
        if t.startswith(".amdhsa_private_segment_fixed_size"):
            r["scratch"] += int(t.split()[-1]) > 0
    return {k: v for k, v in res.items() if "kernel" in k}


def findings(files=None):
    files = files or [os.path.join(CSRC, f) for f in ("gemm_glds.hip", "gemm_conv.hip", "norm.hip")]
    bad, rows = [], []
    for f in files:
        for k, v in sorted(audit(compile_asm(f)).items()):
            rows.append((os.path.basename(f), k, v))
            if v["serialised"] > KNOWN.get(k, 2):
                bad.append(f"{k}: {v['serialised']} of {v['loads']} global loads are waited for immediately")
            if v["scratch"]:
                bad.append(f"{k}: uses scratch")
            if v["ring"] and v["drained_barriers"] > 2:
                bad.append(f"{k}: {v['drained_barriers']} barriers behind a vmcnt(0) wait in a ring kernel")
    return bad, rows


if __name__ == "__main__":
    bad, rows = findings([os.path.abspath(a) for a in sys.argv[1:]] or None)
    for f, k, v in rows:
        print(f"{f:14s} {k[:84]:84s} loads={v['loads']:3d} serialised={v['serialised']:2d} "
              f"vmcnt0+barrier={v['drained_barriers']:2d} scratch={v['scratch']} {'ring' if v['ring'] else ''}")
    for b in bad:
        print("FINDING:", b)
    sys.exit(1 if bad else 0)
