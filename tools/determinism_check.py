#!/usr/bin/env python
"""Two eager runs of the same request must give the same bits (tests/test_hip_parity.py::test_full_size_properties);
prints the max difference of the latents.  Used to bisect a nondeterminism by environment switch:
    PFD_GN_PSTATS=0 python tools/determinism_check.py"""
import contextlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "prompt-free-diffusion_amd"), os.path.join(REPO, "oracle")):
    sys.path.insert(0, p)
os.environ.setdefault("PFD_QUIET", "1")
import torch  # noqa: E402


def main():
    from lib.pipeline import PromptFreePipeline, build_model
    with contextlib.redirect_stdout(sys.stderr):
        net = build_model('pfd_seecoder', device='cuda:0', fp16=True)
    img = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(1234))
    pipe = PromptFreePipeline(net)
    outs = [pipe.generate(img, 4, 512, 512, steps=int(os.environ.get("DET_STEPS", "4")), scale=2.0, seed=20, decode=False)[0].float()
            for _ in range(3)]
    d = [float((outs[i] - outs[0]).abs().max()) for i in (1, 2)]
    sw = {k: v for k, v in os.environ.items() if k.startswith("PFD_") and k != "PFD_QUIET"}
    print(f"determinism {sw}: max|run1 - run0| = {d[0]:.3e}, max|run2 - run0| = {d[1]:.3e} -> {'SAME BITS' if max(d) == 0 else 'DIFFERENT'}")


if __name__ == "__main__":
    main()
