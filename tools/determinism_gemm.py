#!/usr/bin/env python
"""Isolated repeat-launch determinism of single GEMM configurations (same inputs, 40 launches each)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "prompt-free-diffusion_amd"))
os.environ.setdefault("PFD_QUIET", "1")
import torch  # noqa: E402


def main():
    from lib.hip import ops
    g = torch.Generator().manual_seed(1)
    M, K = 32768, 320
    x = torch.randn((M, K), generator=g).half().cuda()
    st = ops.ln_rowstats(x)
    cases = []
    for name, N, act, out_t in (("geglu", 2560, ops.ACT_GEGLU, False), ("qkv+Ct", 960, ops.ACT_NONE, True), ("plain", 320, ops.ACT_NONE, False)):
        w = (torch.randn((N, K), generator=g) * 0.05).half().cuda()
        b = torch.randn((N,), generator=g).half().cuda()
        cs = w.float().sum(1).contiguous()
        for ln in (False, True):
            for tile in (0,):
                cases.append((name, N, act, out_t, w, b, cs, ln, tile))
    for name, N, act, out_t, w, b, cs, ln, tile in cases:
        outs = []
        for _ in range(40):
            kw = dict(bias=b, act=act, tile=tile)
            if ln:
                kw["ln"] = (st, cs, 1e-5)
            if out_t:
                vt = torch.empty((N - 640, M), dtype=torch.float16, device='cuda')
                kw.update(out_t=vt, n_split=640)
            o = ops.gemm(x, w, **kw)
            outs.append(o.clone())
        torch.cuda.synchronize()
        same = sum(int(torch.equal(o, outs[0])) for o in outs)
        print(f"{name:8s} N{N} ln={ln}: {same}/40 launches equal to the first", flush=True)


if __name__ == "__main__":
    main()
