#!/usr/bin/env python
"""Which launch is not deterministic?  Runs one CFG UNet step (C2 shape) twice with every tensor-level op of lib/hip/ops.py
wrapped so that its output is kept, and reports the first ops whose outputs differ between the two runs."""
import contextlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "prompt-free-diffusion_amd"), os.path.join(REPO, "oracle")):
    sys.path.insert(0, p)
os.environ.setdefault("PFD_QUIET", "1")
import torch  # noqa: E402


def main():
    from lib.hip import ops
    from lib.pipeline import build_model
    with contextlib.redirect_stdout(sys.stderr):
        net = build_model('pfd_seecoder', device='cuda:0', fp16=True)
    names = ["gemm", "conv", "groupnorm", "attention", "add", "add_rowvec", "layernorm", "ln_rowstats", "to_nhwc",
             "cfg_ddim_step", "activation", "im2col", "axpby", "groupnorm_table"]
    log = []
    orig = {n: getattr(ops, n) for n in names}

    def wrap(n):
        f = orig[n]

        def g(*a, **k):
            out = f(*a, **k)
            outs = out if isinstance(out, tuple) else (out,)
            shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)]
            log.append((n, shapes, {kk: (tuple(v.shape) if torch.is_tensor(v) else v) for kk, v in k.items()
                                    if kk in ("act", "zero_rows", "k", "rows_per_rv", "ups", "stride", "gn_out") or torch.is_tensor(v)},
                        [t.clone() for t in outs if torch.is_tensor(t)], (a, k)))
            return out
        return g
    for n in names:
        setattr(ops, n, wrap(n))
    g = torch.Generator().manual_seed(0)
    B = 4
    x = torch.randn((B, 4, 64, 64), generator=g).cuda()
    cond = torch.randn((1, 148, 768), generator=g).half().cuda().repeat(B, 1, 1)
    c = torch.cat([torch.zeros_like(cond), cond])
    t = torch.full((2 * B,), 621, dtype=torch.long, device='cuda')
    ctx = net.prepare_context(c)
    ctx.zero_lead = B
    unet = net.diffuser['image']
    emb_all, _ = unet.emb_projections(t[:1])

    def step():
        log.clear()
        xin = ops.to_nhwc(x, rep=1)
        net.apply_model_nhwc('image', xin, t, 'image', ctx, emb_table=emb_all[0:1], cfg_pair=True)
        torch.cuda.synchronize()
        return list(log)
    step()
    a, b = step(), step()
    assert len(a) == len(b)
    nbad = 0
    for i, (ra, rb) in enumerate(zip(a, b)):
        for j, (ta, tb) in enumerate(zip(ra[3], rb[3])):
            if not torch.equal(ta, tb):
                d = (ta.float() - tb.float()).abs()
                nbad += 1
                if nbad <= 3 and ta.dim() == 2:
                    idx = (d > 0).nonzero()
                    rows, cols = sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist()))
                    print(f"   rows {rows[:24]}{'...' if len(rows) > 24 else ''} ({len(rows)}), cols {cols[:40]}{'...' if len(cols) > 40 else ''} ({len(cols)})")
                if nbad <= 12:
                    print(f"op {i:4d} {ra[0]:12s} out{j} {tuple(ta.shape)} args {ra[1]} {ra[2]}: {int((d > 0).sum())} of {ta.numel()} "
                          f"elements differ, max {float(d.max()):.3e}", flush=True)
    print(f"{len(a)} ops per UNet step, {nbad} outputs differ between two runs")
    # the first GEGLU projection and the first fused q|k|v projection, alone, 40 times each on the inputs of run B
    for i, rb in enumerate(b):
        if rb[0] == "gemm" and (rb[2].get("act") == 4 or "out_t" in rb[2]) and i < 20:
            args, kw = rb[4]
            kw = {kk: v for kk, v in kw.items() if kk != "out"}
            outs = []
            for _ in range(40):
                o = orig[rb[0]](*args, **kw)
                o = o[0] if isinstance(o, tuple) else o
                outs.append(o.clone())
            torch.cuda.synchronize()
            same = sum(int(torch.equal(o, outs[0])) for o in outs)
            print(f"op {i} {rb[0]} {rb[1]} alone x40: {same}/40 equal to the first repeat")
    # the first op that differs, alone: the same call on the inputs of run B, ten times
    for i, (ra, rb) in enumerate(zip(a, b)):
        if any(not torch.equal(x_, y_) for x_, y_ in zip(ra[3], rb[3])):
            args, kw = rb[4]
            kw = {kk: v for kk, v in kw.items() if kk != "out"}
            outs = []
            for _ in range(10):
                o = orig[rb[0]](*args, **kw)
                o = o[0] if isinstance(o, tuple) else o
                outs.append(o.clone())
            torch.cuda.synchronize()
            same = sum(int(torch.equal(o, outs[0])) for o in outs)
            eq_b = sum(int(torch.equal(o, rb[3][0])) for o in outs)
            eq_a = sum(int(torch.equal(o, ra[3][0])) for o in outs)
            print(f"op {i} {rb[0]} alone x10 on run B's inputs: {same}/10 equal to the first repeat, {eq_b}/10 equal to run B's output, "
                  f"{eq_a}/10 equal to run A's output")
            d = (ra[3][0].float() - rb[3][0].float()).abs()
            idx = (d > 0).nonzero()[:6]
            for r_, c_ in idx.tolist():
                print(f"   [{r_},{c_}]: run A {float(ra[3][0][r_, c_]):+.5f}  run B {float(rb[3][0][r_, c_]):+.5f}  repeats "
                      f"{[round(float(o[r_, c_]), 5) for o in outs[:4]]}")
            break


if __name__ == "__main__":
    main()
