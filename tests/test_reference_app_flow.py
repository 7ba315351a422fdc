"""CPU, build container only (needs /root/reference; skipped on the GPU box): the drop-in claim, executed.

The body of the reference's `prompt_free_diffusion` class (app.py:98-275: construction, the three weight
loaders, per-request hot swap by tag, `action_inference`) is exec'd UNMODIFIED against THIS package's `lib/`
-- with gradio / torchvision stubbed (UI and PIL conversion, out of scope) and the checkpoint files replaced by
in-memory state dicts (there are no checkpoints here).  The arithmetic entry points are the only thing faked
on this GPU-less host (`ctx_encode`, `DDIMSampler.sample`, `vae_decode` need the HIP library and a GPU; their
numerics are the subject of the -m gpu parity tests); every call the reference makes into them is bound against
the REAL signatures, so a renamed keyword, a missing attribute (`net.ctx['image'].fp16`, `qtransformer.pe_layer`,
`net.ctl.load_state_dict`, `net.device`), a state-dict key mismatch under `strict=True` or a wrong return type
fails here.
"""
import inspect
import os
import re
import types
from collections import OrderedDict

import numpy as np
import pytest
import torch
from PIL import Image

APP = "/root/reference/app.py"
pytestmark = pytest.mark.skipif(not os.path.exists(APP), reason="reference checkout not present (GPU box)")


def _class_source():
    src = open(APP).read()
    a = src.index("class prompt_free_diffusion(object):")
    b = src.index("pfd_inference = prompt_free_diffusion(")
    return src[a:b]


class _ToTensor:
    def __call__(self, pic):
        a = np.asarray(pic.convert("RGB"), dtype=np.float32) / 255.0
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()


class _ToPILImage:
    def __call__(self, t):
        a = t.detach().float().mul(255).byte().permute(1, 2, 0).cpu().numpy()
        return Image.fromarray(a)


def test_reference_app_class_runs_on_this_lib():
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    from lib.model_zoo.ddim import DDIMSampler

    files = {}                       # "path" -> state dict (stands in for the checkpoint files)

    def load_sd_from_file(target):   # app.py:38-48 reads .ckpt/.pth/.safetensors from disk
        return OrderedDict(files[target](None))

    tvtrans = types.SimpleNamespace(ToTensor=_ToTensor, ToPILImage=_ToPILImage)
    ns = dict(torch=torch, np=np, Image=Image, OrderedDict=OrderedDict, tvtrans=tvtrans, time=__import__("time"),
              model_cfg_bank=model_cfg_bank, get_model=get_model, DDIMSampler=DDIMSampler, n_sample_image=2,
              load_sd_from_file=load_sd_from_file,
              highlight_print=lambda info: None,
              ctxencoder_path={'SeeCoder': 'ctx.safetensors', 'SeeCoder-PA': 'ctx_pa.safetensors',
                               'SeeCoder-Anime': 'ctx.safetensors'},
              diffuser_path={'Deliberate-v2.0': 'diffuser_old_names.safetensors', 'SD-v1.5': 'diffuser.safetensors'},
              controlnet_path={'canny': ('canny', 'ctl.safetensors'), 'none': (None, None)})
    holder = {}

    def sub(prefix, strip=False, rename=None):
        def make(_):
            sd = holder['net'].state_dict()
            out = OrderedDict()
            for k, v in sd.items():
                if k.startswith(prefix):
                    k2 = k[len(prefix):] if strip else k
                    out[rename(k2) if rename else k2] = v
            return out
        return make
    files['ctx.safetensors'] = sub('ctx.')
    files['ctx_pa.safetensors'] = sub('ctx.')
    files['diffuser.safetensors'] = sub('diffuser.')
    # checkpoints converted before the rename carry `diffuser.text.context_blocks.*` (app.py:146-151)
    files['diffuser_old_names.safetensors'] = sub(
        'diffuser.', rename=lambda k: k.replace('diffuser.image.context_blocks.', 'diffuser.text.context_blocks.'))
    files['ctl.safetensors'] = sub('ctl.', strip=True)

    # the class builds its own net through OUR registry; hand the loaders a reference to it
    real_get_model = get_model

    def get_model_spy():
        g = real_get_model()

        def build(cfg, *a, **k):
            assert cfg.type == 'pfd_with_control'
            cfg.args.vae_cfg_list[0][1].pth = None          # the autokl checkpoint does not exist here
            holder['net'] = g(cfg, *a, **k)
            return holder['net']
        return build
    ns['get_model'] = get_model_spy
    exec(compile(_class_source(), APP, "exec"), ns)
    cls = ns['prompt_free_diffusion']

    app = cls(fp16=True, tag_ctx='SeeCoder', tag_diffuser='Deliberate-v2.0', tag_ctl='canny')
    net = app.net
    assert net is holder['net'] and app.dtype == torch.float16
    assert next(net.parameters()).dtype == torch.float16 and net.ctx['image'].fp16 is True
    assert not net.training and isinstance(app.sampler, DDIMSampler) and app.sampler.model is net
    assert net.ctx['image'].qtransformer.pe_layer is None
    assert (app.tag_ctx, app.tag_diffuser, app.tag_ctl) == ('SeeCoder', 'Deliberate-v2.0', 'canny')
    # GPU-less host: the reference sets `.device` only inside `.to('cuda')` (pfd.py:100-102)
    net.to('cpu')
    assert net.device == 'cpu'

    # ---- the three arithmetic entry points, faked on this host but bound against the real signatures ----
    calls = []
    real_sample_sig = inspect.signature(DDIMSampler.sample)
    real_ctx_sig = inspect.signature(type(net).ctx_encode)
    real_vae_sig = inspect.signature(type(net).vae_decode)

    def fake_ctx_encode(*a, **k):
        b = real_ctx_sig.bind(net, *a, **k)
        x = b.arguments['x']
        assert b.arguments['which'] == 'image' and x.dtype == torch.float16 and x.shape[:2] == (1, 3)
        assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0
        calls.append(('ctx_encode', tuple(x.shape)))
        return torch.ones((1, 148, 768), dtype=torch.float16)

    def fake_sample(*a, **k):
        b = real_sample_sig.bind(app.sampler, *a, **k)
        args = b.arguments
        assert args['x_info'] == {'type': 'image'} and args['eta'] == 0.0 and args['verbose'] is False
        ci = args['c_info']
        assert set(ci) == {'type', 'conditioning', 'unconditional_conditioning', 'unconditional_guidance_scale',
                           'control'}
        assert ci['conditioning'].shape == (2, 148, 768)
        # (SeeCoder-Anime hands over ONE unconditional context whatever n_samples is, app.py:239-241)
        assert ci['unconditional_conditioning'].shape in ((2, 148, 768), (1, 148, 768))
        calls.append(('sample', args['steps'], list(args['shape']), ci['control'] is not None,
                      bool(ci['unconditional_conditioning'].any()), float(ci['unconditional_guidance_scale'])))
        return torch.zeros(args['shape'], dtype=torch.float16), {'pred_xt': [], 'pred_x0': []}

    def fake_vae_decode(*a, **k):
        b = real_vae_sig.bind(net, *a, **k)
        z = b.arguments['z']
        assert b.arguments['which'] == 'image'
        calls.append(('vae_decode', tuple(z.shape)))
        return torch.full((z.shape[0], 3, z.shape[2] * 8, z.shape[3] * 8), 0.25, dtype=torch.float16)

    net.ctx_encode, net.vae_decode, app.sampler.sample = fake_ctx_encode, fake_vae_decode, fake_sample

    im = Image.fromarray((np.random.RandomState(0).rand(96, 80, 3) * 255).astype(np.uint8))
    imctl = Image.fromarray((np.random.RandomState(1).rand(600, 520, 3) * 255).astype(np.uint8))
    h, w = app.action_autoset_hw(imctl)
    assert (h, w) == (576, 512) and app.action_autoset_hw(None) == (512, 512)
    assert app.action_autoset_method('canny') == 'canny'

    # request 1: same tags -> no reload; ControlNet hint passed through un-preprocessed
    out = app.action_inference(im, imctl, 'canny', False, h, w, 2.0, 20, 'SeeCoder', 'Deliberate-v2.0', 'canny')
    assert calls == [('ctx_encode', (1, 3, 96, 80)), ('sample', 50, [2, 4, 72, 64], True, False, 2.0),
                     ('vae_decode', (2, 4, 72, 64))]
    assert len(out) == 3 and all(isinstance(o, Image.Image) for o in out)        # 2 samples + the control image
    assert out[0].size == (512, 576) and np.asarray(out[0])[0, 0, 0] == 63        # 0.25 * 255 truncated

    # request 2: every tag changes -> hot swap of ctx (+ PPE_MLP attach), diffuser (new key names), no control
    calls.clear()
    v_ctx = net.ctx['image'].qtransformer.level_embed.weight._version
    v_dif = net.diffuser['image'].time_embed[0].weight._version
    out = app.action_inference(im, None, None, False, 512, 512, 1.5, -3, 'SeeCoder-PA', 'SD-v1.5', 'none')
    from lib.model_zoo.seecoder import PPE_MLP
    pe = net.ctx['image'].qtransformer.pe_layer
    assert isinstance(pe, PPE_MLP) and next(pe.parameters()).dtype == torch.float16 and not pe.training
    assert (app.tag_ctx, app.tag_diffuser, app.tag_ctl) == ('SeeCoder-PA', 'SD-v1.5', 'none')
    assert net.ctx['image'].qtransformer.level_embed.weight._version > v_ctx          # strict in-place load ran
    assert net.diffuser['image'].time_embed[0].weight._version > v_dif
    assert calls[1] == ('sample', 50, [2, 4, 64, 64], False, False, 1.5) and len(out) == 2

    # request 3: SeeCoder-Anime -> pe_layer detached again; needs assets/anime_ug.pth relative to the CWD
    calls.clear()
    cwd = os.getcwd()
    os.chdir("/root/reference")
    real_load = torch.load
    # anime_ug.pth was pickled from a CUDA tensor (app.py:239 assumes a GPU); map it on this GPU-less host
    torch.load = lambda f, *a, **k: real_load(f, map_location='cpu')
    try:
        app.action_inference(im, None, None, False, 512, 512, 2.0, 1, 'SeeCoder-Anime', 'SD-v1.5', 'none')
    finally:
        torch.load = real_load
        os.chdir(cwd)
    assert net.ctx['image'].qtransformer.pe_layer is None
    assert calls[1][4] is True                                # a non-zero unconditional context reached the sampler

    # a checkpoint with a wrong key must fail the strict load, exactly like the reference
    files['ctx.safetensors'] = lambda _: OrderedDict(list(sub('ctx.')(None).items())[1:])
    with pytest.raises(RuntimeError, match="Missing key"):
        app.action_load_ctx('SeeCoder')
