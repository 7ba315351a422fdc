"""TEST INFRASTRUCTURE: the long CPU-oracle trajectories of tests/test_hip_trajectory.py as stand-alone jobs.

The GPU suite's long poles are the fp32 CPU oracle runs of whole DDIM trajectories (C2: 50 CFG steps at 64x64, ~5 min;
C5: 31 CFG steps at 96x96, ~10 min; C3: 10 ControlNet-guided steps).  They depend on nothing the GPU computes, so
conftest.py starts them as CPU-only subprocesses at the beginning of a `-m gpu` session and the trajectory tests wait for
their results -- the oracle arithmetic (oracle/pfd_oracle.py, pinned by tests/test_oracle_golden.py) and the comparison are
unchanged, only the wall-clock overlaps.

    python tests/oracle_worker.py <c2|c5|c3> <out.pt> [threads]
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (os.path.join(REPO, "prompt-free-diffusion_amd"), os.path.join(REPO, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _param_shapes():
    with open(os.path.join(HERE, "golden", "state_spec.json")) as f:
        spec = json.load(f)
    return {k: v["shape"] for k, v in spec.items() if v["param"]}


def _sd(shapes, prefix):
    from weights import seeded_tensor
    return {k: seeded_tensor(k, s, 0) for k, s in shapes.items() if k.startswith(prefix)}


def _shard_xT(n, h, w, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)      # lib.pipeline.shard_xT for one rank (kept torch-only here)
    return torch.randn([n, 4, h // 8, w // 8], generator=g)


def trajectory(eps_fn, cond, uncond, xT, steps, scale=2.0, stop_after=None, last_from=None):
    """`steps` CFG DDIM steps of the CPU oracle from xT (ddim.py:107-172 restated by oracle.ddim_step);
    stop_after=k returns after the first k steps of that schedule, last_from=x runs ONLY the last step of the schedule
    from x (the two fixture pins of tests/test_oracle_golden.py).  -> (x, first step, penultimate latent, real steps)"""
    import pfd_oracle as O
    acp = O.schedule_buffers()["alphas_cumprod"]
    ts, a, ap, sg = O.ddim_tables(acp, steps, 0.0)
    x = xT.clone() if last_from is None else last_from.clone()
    first, penult = None, None
    for i, step in enumerate(np.flip(ts)):
        if last_from is not None and i < len(ts) - 1:
            continue
        idx = len(ts) - i - 1
        t = torch.full((x.shape[0],), int(step), dtype=torch.long)
        if i == len(ts) - 1:
            penult = x.clone()
        x, _ = O.ddim_step(eps_fn, x, t, cond, uncond, scale, float(a[idx]), float(ap[idx]), float(sg[idx]))
        if i == 0:
            first = x.clone()
        if stop_after is not None and i + 1 >= stop_after:
            break
    return x, first, penult, len(ts)


def _pins(x, x1, xp, n, stop_after, last_from):
    """what a pin run (stop_after / last_from) returns instead of the whole case"""
    if last_from is not None:
        return {"latent": x, "steps": n}
    return {"first_step": x1, "steps": n}


def case_c2(shapes, stop_after=None, last_from=None):
    import pfd_oracle as O
    img = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(1234))
    sd_u, sd_c, sd_v = _sd(shapes, "diffuser.image."), _sd(shapes, "ctx.image."), _sd(shapes, "vae.image.")
    cond = O.seecoder_encode(sd_c, "ctx.image.", img)
    eps_fn = lambda xx, tt, cc: O.unet_apply(sd_u, "diffuser.image.", xx, tt, cc)  # noqa: E731
    x, x1, xp, n = trajectory(eps_fn, cond, torch.zeros_like(cond), _shard_xT(4, 512, 512, 20)[:1], 50, stop_after=stop_after,
                              last_from=last_from)
    if stop_after is not None or last_from is not None:
        return _pins(x, x1, xp, n, stop_after, last_from)
    return {"latent": x, "first_step": x1, "penultimate": xp, "image": O.vae_decode(sd_v, "vae.image.", x), "steps": n}


def c5_uncond():
    g = torch.Generator().manual_seed(4321)
    ug = torch.zeros((1, 148, 768))
    ug[:, :77] = torch.randn((1, 77, 768), generator=g) - 0.1
    return ug.half().float()                                 # the fp16 values the GPU path is handed


def case_c5(shapes, stop_after=None, last_from=None):
    import pfd_oracle as O
    img = torch.rand((1, 3, 768, 768), generator=torch.Generator().manual_seed(77))
    sd_u = _sd(shapes, "diffuser.image.")
    cond = O.seecoder_encode(_sd(shapes, "ctx.image."), "ctx.image.", img)
    eps_fn = lambda xx, tt, cc: O.unet_apply(sd_u, "diffuser.image.", xx, tt, cc)  # noqa: E731
    x, x1, xp, n = trajectory(eps_fn, cond, c5_uncond(), _shard_xT(2, 768, 768, 31)[:1], 30, stop_after=stop_after,
                              last_from=last_from)
    if stop_after is not None or last_from is not None:
        return _pins(x, x1, xp, n, stop_after, last_from)
    return {"latent": x, "first_step": x1, "penultimate": xp, "steps": n}


def c3_pe_state():
    from weights import seeded_tensor
    spec = json.loads(str(np.load(os.path.join(HERE, "golden", "golden.npz"), allow_pickle=False)["seepa.spec"]))
    return {k: seeded_tensor(k, s, 0) for k, s in spec.items()}


def case_c3(shapes, stop_after=None, last_from=None):
    import pfd_oracle as O
    img = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(1234))
    hint16 = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(4321)).half().float()
    sd_c = _sd(shapes, "ctx.image.")
    sd_c.update(c3_pe_state())
    sd_u, sd_ctl = _sd(shapes, "diffuser.image."), _sd(shapes, "ctl.")
    cond = O.seecoder_encode(sd_c, "ctx.image.", img)

    def eps_fn(xx, tt, cc):   # pfd.py:466-528: ControlNet on the CFG-doubled batch with the same context
        res = O.controlnet_apply(sd_ctl, "ctl.", xx, hint16, tt, cc)
        return O.unet_apply(sd_u, "diffuser.image.", xx, tt, cc, control=res)
    xT = _shard_xT(4, 512, 512, 20)[:1]
    x, x1, xp, n = trajectory(eps_fn, cond, torch.zeros_like(cond), xT, 10, stop_after=stop_after, last_from=last_from)
    if stop_after is not None or last_from is not None:
        return _pins(x, x1, xp, n, stop_after, last_from)
    plain = lambda xx, tt, cc: O.unet_apply(sd_u, "diffuser.image.", xx, tt, cc)  # noqa: E731
    acp = O.schedule_buffers()["alphas_cumprod"]
    ts, a, ap, sg = O.ddim_tables(acp, 10, 0.0)
    xu, _ = O.ddim_step(plain, xT, torch.full((1,), int(ts[-1]), dtype=torch.long), cond, torch.zeros_like(cond), 2.0,
                        float(a[-1]), float(ap[-1]), float(sg[-1]))
    return {"latent": x, "first_step": x1, "penultimate": xp, "first_step_uncontrolled": xu, "steps": n}


CASES = {"c2": case_c2, "c5": case_c5, "c3": case_c3}
# the sources whose arithmetic the fixture holds: their digest is stored in the fixture's meta and asserted by the CPU suite
ORACLE_SOURCES = ("oracle/pfd_oracle.py", "oracle/weights.py", "tests/oracle_worker.py")


def oracle_digest():
    import hashlib
    h = hashlib.sha256()
    for rel in ORACLE_SOURCES:
        with open(os.path.join(REPO, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def main():
    case, out = sys.argv[1], sys.argv[2]
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, min(64, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    t0 = time.time()
    with torch.no_grad():
        res = CASES[case](_param_shapes())
    res["seconds"], res["threads"] = time.time() - t0, threads
    torch.save(res, out + ".tmp")
    os.replace(out + ".tmp", out)


if __name__ == "__main__":
    main()
