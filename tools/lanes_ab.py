#!/usr/bin/env python
"""Same-process A/B of the sampler's sub-batch lanes (DDIMSampler.lanes): the C2 batch (4 images, UNet batch 8) as
1, 2 or 4 independent hipGraphs replayed concurrently on their own HIP streams.  Interleaved rounds, one process, one
box -- every number below is comparable with its neighbours only.

  python tools/lanes_ab.py [--lanes 1,2,4] [--rounds 3] [--batch 4] [--config c2|c5]
"""
import argparse
import contextlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "prompt-free-diffusion_amd"), os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("PFD_QUIET", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="1,2,4")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    from lib.pipeline import PromptFreePipeline, build_model
    from lib.model_zoo.ddim import DDIMSampler
    with contextlib.redirect_stdout(sys.stderr):
        net = build_model('pfd_seecoder', device='cuda:0', fp16=True)
    lanes = [int(v) for v in args.lanes.split(",")]
    pipes = {}
    for n in lanes:
        s = DDIMSampler(net)
        s.lanes = n
        p = PromptFreePipeline(net, sampler=s)
        p.enable_graph(True)
        pipes[n] = p
    image = torch.rand((1, 3, args.size, args.size), generator=torch.Generator().manual_seed(1234))
    ref = None
    for n, p in pipes.items():          # capture + warm-up; the lanes must agree within fp16 noise
        _, lat = p.generate(image, args.batch, args.size, args.size, steps=args.ddim_steps, scale=2.0, seed=20)
        p.generate(image, args.batch, args.size, args.size, steps=args.ddim_steps, scale=2.0, seed=21)
        torch.cuda.synchronize()
        if ref is None:
            ref = lat.float()
        else:
            rel = float((lat.float() - ref).norm() / ref.norm())
            print(f"lanes {n}: latent rel-L2 vs lanes {lanes[0]} = {rel:.3e}", flush=True)
            assert rel < 1e-2
    res = {n: [] for n in lanes}
    for r in range(args.rounds):
        for n, p in pipes.items():
            tm = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(2):
                p.generate(image, args.batch, args.size, args.size, steps=args.ddim_steps, scale=2.0, seed=30 + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2 * 1e3
            p.generate(image, args.batch, args.size, args.size, steps=args.ddim_steps, scale=2.0, seed=40, timings=tm)
            res[n].append((dt, tm.get("ddim_loop_ms", 0.0)))
            print(f"round {r} lanes {n}: {dt:8.1f} ms per batch ({args.batch / dt * 1e3:.2f} images/s), "
                  f"DDIM loop {tm.get('ddim_loop_ms', 0.0):.1f} ms", flush=True)
    print(json.dumps({"batch": args.batch, "size": args.size, "ms_per_batch": {
        str(n): {"min": round(min(v[0] for v in vs), 1), "median": round(sorted(v[0] for v in vs)[len(vs) // 2], 1)}
        for n, vs in res.items()}}))


if __name__ == "__main__":
    main()
