"""CPU: the oracle (oracle/pfd_oracle.py) against fixtures produced by the reference's own
modules (oracle/make_golden.py).  This is what pins the oracle; GPU parity tests then compare the
HIP path with the oracle and with the same fixtures."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import rel_err, seeded_sd

import pfd_oracle as O

T = torch.from_numpy


def test_schedule_buffers(golden):
    buf = O.schedule_buffers()
    for k, v in buf.items():
        np.testing.assert_allclose(v.numpy(), golden["sched." + k], rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("steps,nreal", [(50, 50), (10, 10), (30, 31)])
@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_ddim_tables(golden, steps, nreal, eta):
    acp = O.schedule_buffers()["alphas_cumprod"]
    ts, a, ap, sg = O.ddim_tables(acp, steps, eta)
    tag = f"ddim.s{steps}.eta{eta}."
    assert len(ts) == nreal  # S=30 -> 31 real steps (stride 1000//30 = 33)
    np.testing.assert_array_equal(ts, golden[tag + "timesteps"])
    np.testing.assert_allclose(a, golden[tag + "alphas"], rtol=1e-6)
    np.testing.assert_allclose(ap, golden[tag + "alphas_prev"], rtol=1e-6)
    np.testing.assert_allclose(sg, golden[tag + "sigmas"], rtol=1e-5, atol=1e-12)


def test_timestep_embedding(golden):
    out = O.timestep_embedding(T(golden["temb.t"]), 320)
    np.testing.assert_allclose(out.numpy(), golden["temb.out"], atol=1e-6)


def test_unet_eps(golden, param_shapes):
    sd = seeded_sd(param_shapes, "diffuser.image.")
    eps = O.unet_apply(sd, "diffuser.image.", T(golden["unet.x"]), T(golden["unet.t"]), T(golden["unet.c"]))
    assert rel_err(eps, golden["unet.eps"]) < 2e-4


def test_controlnet(golden, param_shapes):
    sd = seeded_sd(param_shapes, "ctl.")
    sd.update(seeded_sd(param_shapes, "diffuser.image."))
    x, t, c = T(golden["unet.x"]), T(golden["unet.t"]), T(golden["unet.c"])
    ccs = O.controlnet_apply(sd, "ctl.", x, T(golden["ctl.hint"]), t, c)
    assert len(ccs) == 13
    for i, o in enumerate(ccs):
        assert list(o.shape) == list(golden[f"ctl.res{i}.shape"])
        flat = o.flatten()
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        assert rel_err(flat[idx], golden[f"ctl.res{i}.sample"]) < 5e-4
    eps = O.unet_apply(sd, "diffuser.image.", x, t, c, control=ccs)
    assert rel_err(eps, golden["ctl.eps"]) < 5e-4


def test_seecoder(golden, param_shapes):
    sd = seeded_sd(param_shapes, "ctx.image.")
    img = T(golden["see.img"])
    fea = O.swin_forward(sd, "ctx.image.imencoder.", img)
    for k in ("res3", "res4", "res5"):
        assert list(fea[k].shape) == list(golden[f"see.swin.{k}.shape"])
        flat = fea[k].flatten()
        idx = torch.linspace(0, flat.numel() - 1, 256).long()
        assert rel_err(flat[idx], golden[f"see.swin.{k}.sample"]) < 5e-4
    dec = O.seecoder_decoder(sd, "ctx.image.imdecoder.", {k: fea[k] for k in ("res3", "res4", "res5")})
    for k in ("res3", "res4", "res5"):
        flat = dec[k].flatten()
        idx = torch.linspace(0, flat.numel() - 1, 256).long()
        assert rel_err(flat[idx], golden[f"see.dec.{k}.sample"]) < 5e-4
    ctx = O.seecoder_qtransformer(sd, "ctx.image.qtransformer.", [dec["res3"], dec["res4"], dec["res5"]])
    assert rel_err(ctx, golden["see.ctx"]) < 1e-3


def test_seecoder_small_odd(golden, param_shapes):
    sd = seeded_sd(param_shapes, "ctx.image.")
    ctx = O.seecoder_encode(sd, "ctx.image.", T(golden["see2.img"]))
    assert rel_err(ctx, golden["see2.ctx"]) < 1e-3


def test_seecoder_position_aware(golden, param_shapes):
    from weights import seeded_tensor
    sd = seeded_sd(param_shapes, "ctx.image.")
    for k, s in json.loads(str(golden["seepa.spec"])).items():
        sd[k] = seeded_tensor(k, s, 0)
    ctx = O.seecoder_encode(sd, "ctx.image.", T(golden["see.img"]))
    assert rel_err(ctx, golden["seepa.ctx"]) < 1e-3
    assert rel_err(golden["see.ctx"], golden["seepa.ctx"]) > 1e-2  # the positional term matters


def test_vae_decode(golden, param_shapes):
    sd = seeded_sd(param_shapes, "vae.image.")
    img = O.vae_decode(sd, "vae.image.", T(golden["vae.z"]))
    assert float((img - T(golden["vae.img"])).abs().max()) < 2e-4


def test_vae_encode(golden, param_shapes):
    sd = seeded_sd(param_shapes, "vae.image.")
    m = O.vae_encode_moments(sd, "vae.image.", T(golden["vaeenc.x"]))
    assert rel_err(m, golden["vaeenc.moments"]) < 2e-4


def test_end_to_end_trajectory(golden, param_shapes):
    """4 DDIM steps with CFG 2.0 (zero unconditional context), then decode"""
    sd = seeded_sd(param_shapes, "diffuser.image.")
    sdv = seeded_sd(param_shapes, "vae.image.")
    cond = T(golden["see.ctx"])
    acp = O.schedule_buffers()["alphas_cumprod"]
    ts, a, ap, sg = O.ddim_tables(acp, 4, 0.0)
    x = T(golden["e2e.xT"])
    eps_fn = lambda xx, tt, cc: O.unet_apply(sd, "diffuser.image.", xx, tt, cc)  # noqa: E731
    for i, step in enumerate(np.flip(ts)):
        idx = len(ts) - i - 1
        t = torch.full((1,), int(step), dtype=torch.long)
        x, _ = O.ddim_step(eps_fn, x, t, cond, torch.zeros_like(cond), 2.0, float(a[idx]), float(ap[idx]),
                           float(sg[idx]))
        assert rel_err(x, golden["e2e.traj"][i]) < 1e-3
    img = O.vae_decode(sdv, "vae.image.", x)
    assert float((img - T(golden["e2e.img"])).abs().max()) < 2e-3


def test_img2img(golden, param_shapes):
    """x0 branch of the sampler (ddim.py:97-103): 8-step schedule, start from ddim index 5"""
    sd = seeded_sd(param_shapes, "diffuser.image.")
    cond = T(golden["see.ctx"])
    eps_fn = lambda xx, tt, cc: O.unet_apply(sd, "diffuser.image.", xx, tt, cc)  # noqa: E731
    x, pred = O.img2img(eps_fn, T(golden["i2i.x0"]), T(golden["i2i.noise"]), cond, torch.zeros_like(cond), 2.0, 8, 5)
    assert rel_err(x, golden["i2i.out"]) < 1e-3
    assert rel_err(pred, golden["i2i.pred_x0_last"]) < 1e-3


def test_multicontext_sampling(golden, param_shapes):
    """two contexts mixed 0.7 / 0.3 at every context layer, CFG 2.0, 4 DDIM steps (ddim.py:174-299)"""
    sd = seeded_sd(param_shapes, "diffuser.image.")
    conds = [T(golden["see.ctx"]), T(golden["see2.ctx"])]
    unconds = [torch.zeros_like(c) for c in conds]
    acp = O.schedule_buffers()["alphas_cumprod"]
    ts, a, ap, sg = O.ddim_tables(acp, 4, 0.0)
    x = T(golden["mc.xT"])
    for i, step in enumerate(np.flip(ts)):
        idx = len(ts) - i - 1
        t = torch.full((1,), int(step), dtype=torch.long)
        x, _ = O.ddim_step_multicontext(sd, "diffuser.image.", x, t, conds, unconds, list(golden["mc.ratios"]), 2.0,
                                        float(a[idx]), float(ap[idx]), float(sg[idx]))
    assert rel_err(x, golden["mc.out"]) < 1e-3


@pytest.mark.parametrize("case", ["c2", "c3", "c5"])
def test_trajectory_fixture_first_and_last_step(case, monkeypatch):
    """tests/golden/trajectories.npz (the oracle trajectories the GPU suite compares whole DDIM runs with at the BASELINE
    shapes, oracle/make_trajectory_golden.py) is what THIS oracle computes: the first DDIM step of every case -- context
    encode (SeeCoder / SeeCoder-PA at 512x512 / 768x768), one CFG UNet evaluation (64x64 / 96x96 latent; with the
    ControlNet residuals for c3), the DDIM update -- is recomputed here from the same seeds and compared, and so is the LAST
    step from the stored penultimate latent (the end of the schedule: a late indexing / schedule change in the oracle
    cannot hide behind a correct first step).  The fixture's meta carries the digest of the oracle sources it was
    written from: a changed oracle fails here until oracle/make_trajectory_golden.py is re-run.
    (PFD_ORACLE_LIVE=1 in a GPU session recomputes the whole trajectories.)"""
    import oracle_worker as OW
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectories.npz"), allow_pickle=False))
    meta = json.loads(str(fx["meta"]))
    assert meta.get("oracle_sha256") == OW.oracle_digest(), \
        f"tests/golden/trajectories.npz was written from other oracle sources ({meta.get('oracle_sources')}): re-run oracle/make_trajectory_golden.py"
    want_steps = {"c2": 50, "c3": 10, "c5": 31}[case]
    assert meta["cases"][case]["steps"] == want_steps
    # the two pin runs share their synthetic weights and their context encode (same seeds, same image): computed once
    import pfd_oracle as O
    sd_memo, enc_memo, make_sd, encode = {}, {}, OW._sd, O.seecoder_encode

    def sd_once(shapes, prefix):
        if prefix not in sd_memo:
            sd_memo[prefix] = make_sd(shapes, prefix)
        return dict(sd_memo[prefix])

    def encode_once(sd, prefix, img):
        key = (prefix, tuple(img.shape), float(img.double().sum()), len(sd))
        if key not in enc_memo:
            enc_memo[key] = encode(sd, prefix, img)
        return enc_memo[key].clone()
    monkeypatch.setattr(OW, "_sd", sd_once)
    monkeypatch.setattr(O, "seecoder_encode", encode_once)
    with torch.no_grad():
        got = OW.CASES[case](OW._param_shapes(), stop_after=1)
        last = OW.CASES[case](OW._param_shapes(), last_from=T(fx[f"{case}.penultimate"]))
    assert got["steps"] == last["steps"] == want_steps
    ref = T(fx[f"{case}.first_step"])
    assert got["first_step"].shape == ref.shape == fx[f"{case}.latent"].shape
    # same arithmetic on another host / thread count: summation-order noise only
    assert rel_err(got["first_step"], ref) < 1e-4
    assert rel_err(last["latent"], fx[f"{case}.latent"]) < 1e-4
    assert np.isfinite(fx[f"{case}.latent"]).all() and float(np.abs(fx[f"{case}.latent"]).max()) > 1.0
