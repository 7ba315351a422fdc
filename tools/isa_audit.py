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


def areg_loop_waits(asm_path):
    """gemm160ar_kernel (A fragments in registers, uncounted asm loads): {kernel: (KEEP, bad)} where `bad` lists every VM
    wait between the first and the last MFMA of the kernel that is neither the counted wait of the K loop (vmcnt(KEEP),
    KEEP = per-step VMEM instructions x (ring depth - 2)) nor the drain of the loop's tail (a bare vmcnt(0)); a
    compiler-inserted wait there (it shows as `vmcnt(N) lgkmcnt(M)` or another N) would serialise the ring."""
    res, name, body = {}, None, []
    for line in open(asm_path):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and "gemm160ar_kernel" in name:
            t = line.split(";")[0].strip()
            if t.startswith(".Lfunc_end"):
                tm = re.search(r"gemm160ar_kernelILi(\d)ELi(\d)ELi(\d)E", name)
                waves_m, wmb, nbuf = (int(x) for x in tm.groups())
                nw = 2 * waves_m
                keep = (2 * wmb + (20 + nw - 1) // nw) * (nbuf - 2)
                idx = [i for i, x in enumerate(body) if x.startswith("v_mfma")]
                loop = body[idx[0]:idx[-1] + 1]
                waits = [x for x in loop if x.startswith("s_waitcnt") and "vmcnt" in x]
                bad = [x for x in waits if x not in (f"s_waitcnt vmcnt({keep})", "s_waitcnt vmcnt(0)")]
                res[name] = (keep, bad, len([x for x in waits if x == f"s_waitcnt vmcnt({keep})"]), nbuf)
                name = None
            elif t:
                body.append(t)
    return res


def _regs(tok):
    """v[a:b] / vN operand -> set of VGPR numbers"""
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def areg_register_hygiene(asm_path):
    """gemm160ar_kernel: a register loaded by one of the uncounted asm loads (the activation ring) may be READ by nothing but
    the MFMAs that consume it -- a compiler copy / spill of such a register between the load and the counted wait would move
    garbage (guide 5.7 item 1).  hipcc does use the ring registers as address temporaries while they are dead (between the
    last MFMA of a slot and the load that refills it, often as the load's own address operand); those reads see a value
    written by an ordinary instruction and are fine.  Walks each kernel in program order with, per register, whether its last
    writer was an asm load.  -> {kernel: (offending instructions, ring registers)}"""
    res, name, body, in_asm = {}, None, [], False
    for line in open(asm_path):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and "gemm160ar_kernel" in name:
            raw = line.strip()
            if raw.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if raw.startswith(";;#ASMEND"):
                in_asm = False
                continue
            t = line.split(";")[0].strip()
            if t.startswith(".Lfunc_end"):
                ring = set()
                for asm, x in body:
                    if asm and x.startswith("global_load_dwordx4"):
                        ring |= _regs(x.split()[1].rstrip(","))
                loaded = set()      # ring registers whose last writer is an asm load
                bad = []
                for asm, x in body:
                    parts = x.split()
                    if len(parts) < 2 or x.endswith(":"):
                        continue
                    ops = [o.rstrip(",") for o in parts[1:]]
                    if asm and x.startswith("global_load_dwordx4"):
                        loaded |= _regs(ops[0])       # (its address operand may overlap: read at issue, before the write)
                        continue
                    store = parts[0].startswith(("global_store", "scratch_store", "ds_write", "buffer_store"))
                    srcs = ops if store else ops[1:]
                    dsts = [] if store else ops[:1]
                    if parts[0].startswith("v_mfma"):
                        srcs, dsts = [ops[3]], [ops[0]]          # A / B operands may read the ring; C / D must not be it
                    for o in srcs:
                        if _regs(o) & loaded:
                            bad.append(x)
                    for o in dsts:
                        loaded -= _regs(o)
                res[name] = (bad, len(ring))
                name = None
            elif t:
                body.append((in_asm, t))
    return res


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
