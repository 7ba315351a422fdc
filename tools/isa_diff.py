#!/usr/bin/env python
"""Per-kernel comparison of the generated gfx950 ISA between a git revision and the working tree (no GPU needed:
hipcc -S cross-compiles).  The last hardware-validated library of a round is a commit; edits after the GPU budget is
spent add forced variants / opt-in modes and claim "the ISA of every existing instantiation is unchanged" -- this is
the check behind that sentence.

usage: isa_diff.py <rev> [file.hip ...]      (default files: every kernel file of csrc/)
prints one line per kernel: same | CHANGED | new | gone; exit 1 when a kernel that exists in both differs
(unless it is listed with --allow <substring>, repeatable)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC_REL = os.path.join("prompt-free-diffusion_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILES = ("gemm_glds.hip", "gemm_conv.hip", "attention.hip", "swin_attn.hip", "norm.hip", "elementwise.hip")


def flags(root):
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{root}/include", f"-I{root}/{CSRC_REL}",
            "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only"]


def kernels(asm_path):
    """-> {symbol: [normalised instruction lines]} for every global function of the device assembly."""
    out, name = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if name is None:
            continue
        t = line.split(";")[0].strip()
        if t.startswith(".Lfunc_end"):
            name = None
            continue
        if not t or t.startswith(".") and not t.startswith(".LBB"):
            continue
        out[name].append(re.sub(r"\.LBB\d+_", ".LBB_", t))   # block labels carry the function's index in the file
    return out


def compile_tree(root, files, tag):
    res = {}
    for f in files:
        src = os.path.join(root, CSRC_REL, f)
        if not os.path.exists(src):
            continue
        asm = os.path.join(tempfile.gettempdir(), f"pfd_isadiff_{tag}_{f}.s")
        subprocess.run([HIPCC] + flags(root) + [src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
        for k, v in kernels(asm).items():
            res[(f, k)] = v
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rev")
    ap.add_argument("files", nargs="*", default=list(FILES))
    ap.add_argument("--allow", action="append", default=[])
    ap.add_argument("--map", action="append", default=[], metavar="OLD=NEW",
                    help="an old symbol that is gone is looked up again with this substring replaced (a template that "
                         "gained a defaulted parameter)")
    ap.add_argument("--quiet", action="store_true", help="only print kernels that are not 'same'")
    a = ap.parse_args()
    old_root = tempfile.mkdtemp(prefix="pfd_isadiff_")
    tar = subprocess.run(["git", "-C", REPO, "archive", a.rev, CSRC_REL, "include"], check=True, capture_output=True)
    subprocess.run(["tar", "-x", "-C", old_root], input=tar.stdout, check=True)
    old = compile_tree(old_root, a.files, "old")
    new = compile_tree(REPO, a.files, "new")
    for m in a.map:
        src, dst = m.split("=", 1)
        for (f, k) in list(old):
            if (f, k) not in new and src in k and (f, k.replace(src, dst)) in new:
                old[(f, k.replace(src, dst))] = old.pop((f, k))
    bad = 0
    demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()[:150]
    counts = dict(same=0, CHANGED=0, new=0, gone=0)
    for key in sorted(set(old) | set(new)):
        f, k = key
        if "kernel" not in k:
            continue
        st = ("new" if key not in old else "gone" if key not in new else
              "same" if old[key] == new[key] else "CHANGED")
        counts[st] += 1
        if st == "CHANGED" and not any(s in k or s in demangle(k) for s in a.allow):
            bad += 1
        if st != "same" or not a.quiet:
            extra = ""
            if st == "CHANGED":
                extra = f"  ({len(old[key])} -> {len(new[key])} instructions)"
            print(f"{st:8s} {f:16s} {demangle(k)}{extra}")
    print(f"# {a.rev} vs working tree: " + ", ".join(f"{v} {k}" for k, v in counts.items()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
