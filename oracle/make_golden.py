"""TEST INFRASTRUCTURE: generate tests/golden/* by running the REFERENCE ITSELF.

The reference (SHI-Labs/Prompt-Free-Diffusion, mounted read-only at /root/reference) has no
tests and no golden vectors, so parity is pinned by importing its own modules in
this container (fp32, CPU), filling them with the deterministic weights of oracle/weights.py and
recording small input/output pairs.  The reference tree is NOT present on the GPU box: only the
fixtures written here (committed) travel.  Run:  python oracle/make_golden.py

Harness-side shims only, the reference sources are untouched:
  stub modules torchvision(.models/.transforms), easydict, omegaconf.listconfig, tqdm-silencer;
  torch.cuda.device_count() -> 1 so lib/sync.get_rank does not divide by zero;
  DDIMSampler.register_buffer -> plain setattr (it hard-codes .to("cuda"), ddim.py:17-21);
  vae cfg `pth` -> None (the checkpoint is not in the container).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("PFD_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, HERE)
from weights import fill_module_, param_spec, seeded_tensor  # noqa: E402


def install_shims():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvt = types.ModuleType("torchvision.transforms")
    tvm.VGG16_Weights = type("VGG16_Weights", (), {"IMAGENET1K_V1": None})
    tvm.vgg16 = lambda *a, **k: None
    tv.models, tv.transforms = tvm, tvt
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt})

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            d = {} if d is None else dict(d)
            d.update(kw)
            for k, v in d.items():
                self[k] = v

        @classmethod
        def _w(cls, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                return cls(v)
            if isinstance(v, (list, tuple)):
                return type(v)(cls._w(i) for i in v)
            return v

        def __setitem__(self, k, v):
            super().__setitem__(k, self._w(v))

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def update(self, other=None, **kw):
            d = {} if other is None else dict(other)
            d.update(kw)
            for k, v in d.items():
                self[k] = v

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    oc = types.ModuleType("omegaconf")
    ocl = types.ModuleType("omegaconf.listconfig")
    ocl.ListConfig = type("ListConfig", (list,), {})
    oc.listconfig = ocl
    sys.modules.update({"omegaconf": oc, "omegaconf.listconfig": ocl})
    if torch.cuda.device_count() == 0:
        torch.cuda.device_count = lambda: 1


def rnd(name, shape, scale=1.0):
    return seeded_tensor("input." + name, shape, seed=1) * scale if len(shape) > 1 else None


def main():
    install_shims()
    os.makedirs(OUT, exist_ok=True)
    os.chdir(REF)
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    from lib.model_zoo.ddim import DDIMSampler
    from lib.model_zoo.seecoder import PPE_MLP

    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)

    cfgm = model_cfg_bank()('pfd_seecoder_with_controlnet')
    cfgm.args.vae_cfg_list[0][1].pth = None
    # resolved config (plain json) -> pins lib/cfg_helper.py semantics
    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        return o
    with open(os.path.join(OUT, "cfg_pfd_seecoder_with_controlnet.json"), "w") as f:
        json.dump(plain(cfgm), f, indent=1, sort_keys=True)

    net = get_model()(cfgm, verbose=False)
    net.to('cpu')
    net.eval()
    fill_module_(net, seed=0)

    # ---- state-dict surface: every key, shape, dtype; which are parameters ----
    sd = net.state_dict()
    pnames = set(n for n, _ in net.named_parameters())
    spec = {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "param": k in pnames}
            for k, v in sd.items()}
    with open(os.path.join(OUT, "state_spec.json"), "w") as f:
        json.dump(spec, f, sort_keys=True)
    print("state dict keys:", len(spec))

    G = {}

    # ---- schedule buffers + DDIM tables ----
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_mean_coef1", "posterior_mean_coef2"):
        G["sched." + k] = sd[k].numpy()
    sampler = DDIMSampler(net)
    for steps in (50, 10, 30):
        for eta in (0.0, 0.5):
            sampler.make_schedule(steps, ddim_eta=eta, verbose=False)
            tag = f"ddim.s{steps}.eta{eta}."
            G[tag + "timesteps"] = np.asarray(sampler.ddim_timesteps)
            G[tag + "alphas"] = np.asarray(sampler.ddim_alphas, dtype=np.float64)
            G[tag + "alphas_prev"] = np.asarray(sampler.ddim_alphas_prev, dtype=np.float64)
            G[tag + "sigmas"] = np.asarray(sampler.ddim_sigmas, dtype=np.float64)

    # ---- UNet: one apply_model call (CFG pair: two samples, different t, different contexts) ----
    x = seeded_tensor("input.unet.x", (2, 4, 16, 24), 1)
    t = torch.tensor([981, 421], dtype=torch.long)
    c = seeded_tensor("input.unet.c", (2, 148, 768), 1)
    eps = net.apply_model({'type': 'image', 'x': x.clone()}, t, {'type': 'image', 'c': c.clone()})
    G["unet.x"], G["unet.t"], G["unet.c"], G["unet.eps"] = x.numpy(), t.numpy(), c.numpy(), eps.numpy()
    print("unet eps", float(eps.abs().mean()), float(eps.std()))

    # timestep embedding + time MLP on their own
    from lib.model_zoo.diffusion_utils import timestep_embedding
    temb = timestep_embedding(torch.tensor([1, 21, 500, 981]), 320)
    G["temb.t"], G["temb.out"] = np.array([1, 21, 500, 981]), temb.numpy()

    # ---- ControlNet: residuals + the controlled eps ----
    hint = torch.rand((1, 3, 128, 192), generator=torch.Generator().manual_seed(77))
    ccs = net.ctl(x.clone(), hint=hint, timesteps=t, context=c)
    G["ctl.hint"] = hint.numpy()
    for i, o in enumerate(ccs):
        flat = o.flatten()
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        G[f"ctl.res{i}.shape"] = np.array(o.shape)
        G[f"ctl.res{i}.stats"] = np.array([float(o.mean()), float(o.std()), float(o.abs().max())])
        G[f"ctl.res{i}.sample"] = flat[idx].numpy()
    eps_ctl = net.apply_model({'type': 'image', 'x': x.clone()}, t,
                              {'type': 'image', 'c': c.clone(), 'control': hint})
    G["ctl.eps"] = eps_ctl.numpy()
    print("ctl eps", float(eps_ctl.std()))

    # ---- SeeCoder: context of one image whose patch grid (32x40) is not a multiple of 12 ----
    img = torch.rand((1, 3, 128, 160), generator=torch.Generator().manual_seed(1234))
    see = net.ctx['image']
    fea = see.imencoder(img)
    for k in ('res3', 'res4', 'res5'):
        flat = fea[k].flatten()
        idx = torch.linspace(0, flat.numel() - 1, 256).long()
        G[f"see.swin.{k}.shape"] = np.array(fea[k].shape)
        G[f"see.swin.{k}.stats"] = np.array([float(fea[k].mean()), float(fea[k].std()), float(fea[k].abs().max())])
        G[f"see.swin.{k}.sample"] = flat[idx].numpy()
    dec = see.imdecoder({k: fea[k] for k in ('res3', 'res4', 'res5')})
    for k in ('res3', 'res4', 'res5'):
        flat = dec[k].flatten()
        idx = torch.linspace(0, flat.numel() - 1, 256).long()
        G[f"see.dec.{k}.stats"] = np.array([float(dec[k].mean()), float(dec[k].std()), float(dec[k].abs().max())])
        G[f"see.dec.{k}.sample"] = flat[idx].numpy()
    ctx = net.ctx_encode(img, 'image')
    G["see.img"], G["see.ctx"] = img.numpy(), ctx.numpy()
    print("seecoder ctx", tuple(ctx.shape), float(ctx.std()))
    # a second, square, smaller image (8x8 res5 < one window; odd PatchMerging sizes: 25x25 patches)
    img2 = torch.rand((1, 3, 100, 100), generator=torch.Generator().manual_seed(4321))
    G["see2.img"], G["see2.ctx"] = img2.numpy(), net.ctx_encode(img2, 'image').numpy()

    # SeeCoder-PA: attach a PPE_MLP exactly like app.py:166-177 does
    pe = PPE_MLP(freq_num=20, freq_max=None, out_channel=768, mlp_layer=3)
    fill_module_(pe, seed=0, prefix="ctx.image.qtransformer.pe_layer.")
    see.qtransformer.pe_layer = pe.eval()
    G["seepa.ctx"] = net.ctx_encode(img, 'image').numpy()
    G["seepa.spec"] = np.array(json.dumps(param_spec(pe, "ctx.image.qtransformer.pe_layer.")))
    see.qtransformer.pe_layer = None

    # ---- VAE decode / encode ----
    z = seeded_tensor("input.vae.z", (1, 4, 8, 16), 1)
    im = net.vae_decode(z, 'image')
    G["vae.z"], G["vae.img"] = z.numpy(), im.numpy()
    print("vae img", float(im.mean()), float(im.std()))
    xim = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(5))
    post = net.vae['image'].encode(xim, out_posterior=True)
    G["vaeenc.x"], G["vaeenc.moments"] = xim.numpy(), post.parameters.numpy()

    # ---- end to end: ctx -> 4-step DDIM (CFG 2.0, zero uncond) -> decode, x_T injected ----
    steps, shape = 4, [1, 4, 8, 8]
    xT = seeded_tensor("input.e2e.xT", shape, 1)
    cond = ctx
    sampler.make_schedule(steps, ddim_eta=0.0, verbose=False)
    ts = sampler.ddim_timesteps
    x_info = {'type': 'image', 'x': xT.clone()}
    c_info = {'type': 'image', 'conditioning': cond, 'unconditional_conditioning': torch.zeros_like(cond),
              'unconditional_guidance_scale': 2.0}
    traj = []
    for i, step in enumerate(np.flip(ts)):
        index = len(ts) - i - 1
        tt = torch.full((shape[0],), int(step), dtype=torch.long)
        x_prev, pred_x0 = sampler.p_sample_ddim(x_info, c_info, tt, index)
        x_info['x'] = x_prev
        traj.append(x_prev.numpy())
    G["e2e.xT"], G["e2e.traj"] = xT.numpy(), np.stack(traj)
    G["e2e.img"] = net.vae_decode(x_prev, 'image').numpy()
    print("e2e final latent std", float(x_prev.std()))

    # ---- img2img (ddim.py:97-103): x0 noised to ddim step k by q_sample, then k reverse steps.  The
    # reference draws the noise with torch.randn_like inside q_sample; the harness pins that draw.
    x0 = seeded_tensor("input.i2i.x0", shape, 1)
    noise = seeded_tensor("input.i2i.noise", shape, 1)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, *a, **k: noise.to(dtype=t.dtype)
    try:
        xi, inter = sampler.sample(steps=8, shape=shape, x_info={'type': 'image', 'x0': x0.clone(), 'x0_forward_timesteps': 5},
                                   c_info={'type': 'image', 'conditioning': cond,
                                           'unconditional_conditioning': torch.zeros_like(cond),
                                           'unconditional_guidance_scale': 2.0}, eta=0., verbose=False)
    finally:
        torch.randn_like = real_randn_like
    G["i2i.x0"], G["i2i.noise"], G["i2i.out"] = x0.numpy(), noise.numpy(), xi.numpy()
    G["i2i.pred_x0_last"] = inter['pred_x0'][-1].numpy()
    print("i2i final latent std", float(xi.std()), "intermediates", len(inter['pred_x0']))

    # ---- multi-context sampling (ddim.py:174-299, pfd.py:366-439): two contexts, ratios 0.7 / 0.3,
    # 'attention' mixing, 4 steps; the reference draws x_T with torch.randn, pinned by the harness.
    cond2 = net.ctx_encode(img2, 'image')
    xT2 = seeded_tensor("input.mc.xT", shape, 1)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: xT2.clone()
    try:
        c_list = [{'type': 'image', 'conditioning': cc, 'unconditional_conditioning': torch.zeros_like(cc),
                   'unconditional_guidance_scale': 2.0, 'ratio': rr} for cc, rr in ((cond, 0.7), (cond2, 0.3))]
        xm, inter = sampler.sample_multicontext(steps=4, shape=shape, x_info={'type': 'image'},
                                                c_info_list=c_list, eta=0., verbose=False)
    finally:
        torch.randn = real_randn
    G["mc.xT"], G["mc.out"], G["mc.ratios"] = xT2.numpy(), xm.numpy(), np.array([0.7, 0.3])
    print("multicontext final latent std", float(xm.std()))

    np.savez_compressed(os.path.join(OUT, "golden.npz"), **G)
    print("wrote", os.path.join(OUT, "golden.npz"), "entries:", len(G))


if __name__ == "__main__":
    main()
