"""CPU: host-side logic of the product package -- config bank, registry, state-dict surface,
C-ABI symbol table.  No kernel is launched here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO


def _plain(o):
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    return o


def test_cfg_bank_matches_reference():
    from lib.cfg_helper import model_cfg_bank
    cfg = model_cfg_bank()('pfd_seecoder_with_controlnet')
    with open(os.path.join(REPO, "tests", "golden", "cfg_pfd_seecoder_with_controlnet.json")) as f:
        ref = json.load(f)
    ref["args"]["vae_cfg_list"][0][1]["pth"] = "pretrained/pfd/vae/sd-v2-0-base-autokl.pth"  # harness nulled it
    assert _plain(cfg) == ref
    assert cfg.type == 'pfd_with_control' and cfg.args.ctl_cfg.type == 'controlnet'
    assert cfg.args.ctx_cfg_list[0][1].args.imencoder_cfg.args.window_size == 12


def test_cfg_bank_inheritance_and_broken_entries():
    from lib.cfg_helper import model_cfg_bank
    bank = model_cfg_bank()
    base = bank('pfd_seecoder')
    assert base.type == 'pfd' and base.args.timesteps == 1000 and base.args.latent_scale_factor.image == 0.18215
    qpa = bank('seecoder_query_transformer_position_aware')
    assert qpa.args.with_fea2d_pos is True and qpa.args.num_queries == [4, 144]
    with pytest.raises(ValueError):     # super_cfg: seet does not exist (kept broken like the reference)
        bank('seecoder_pa')
    with pytest.raises(ValueError):     # unknown prefix
        bank('nonexistent_model')
    # returned configs are copies
    base.args.timesteps = 5
    assert bank('pfd_seecoder').args.timesteps == 1000


def test_cfg_unique_holder_singleton():
    from lib.cfg_holder import cfg_unique_holder
    a, b = cfg_unique_holder(), cfg_unique_holder()
    assert a is b
    a.save_cfg({'x': [1, 2]})
    a.add_code('main')
    assert b.cfg == {'x': [1, 2]} and 'main' in b.code


def test_registry_names():
    from lib.model_zoo import get_model
    from lib.cfg_helper import model_cfg_bank
    import lib.model_zoo.pfd, lib.model_zoo.autokl, lib.model_zoo.openaimodel  # noqa: F401,E401
    import lib.model_zoo.controlnet, lib.model_zoo.seecoder, lib.model_zoo.swin  # noqa: F401,E401
    reg = get_model().model
    for name in ('pfd', 'pfd_with_control', 'autoencoderkl', 'openai_unet_2d_next', 'controlnet', 'seecoder',
                 'seecoder_decoder', 'seecoder_query_transformer', 'swin'):
        assert name in reg, name
    assert get_model() is get_model()
    assert get_model()(None) is None
    from lib.model_zoo.seecoder import PPE_MLP  # app.py imports this by name (app.py:166-177)
    assert PPE_MLP(freq_num=20, freq_max=None, out_channel=768, mlp_layer=3).mlp[0].in_features == 80


@pytest.fixture(scope="module")
def full_net():
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    cfg = model_cfg_bank()('pfd_seecoder_with_controlnet')
    cfg.args.vae_cfg_list[0][1].pth = None
    torch.manual_seed(0)
    return get_model()(cfg, verbose=False)


def test_state_dict_surface(full_net, state_spec):
    """same 1902 keys, shapes, dtypes and parameter/buffer split as the reference composite"""
    sd = full_net.state_dict()
    assert len(state_spec) == 1902
    assert set(sd.keys()) == set(state_spec.keys())
    pnames = set(n for n, _ in full_net.named_parameters())
    for k, v in sd.items():
        ref = state_spec[k]
        assert list(v.shape) == ref["shape"], k
        assert str(v.dtype).replace("torch.", "") == ref["dtype"], k
        assert (k in pnames) == ref["param"], k


def test_schedule_buffers_match_reference(full_net, golden):
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_allclose(getattr(full_net, k).numpy(), golden["sched." + k], rtol=2e-6, atol=1e-9)
    assert full_net.num_timesteps == 1000
    i = full_net.diffuser['image']
    assert (len(i.i_order), len(i.m_order), len(i.o_order)) == (30, 3, 37)
    assert len(i.data_blocks) == 30 and len(i.context_blocks) == 16


def test_composite_quirks(full_net):
    assert full_net.to('cpu') is None and full_net.device == 'cpu'     # pfd.py:100-102
    assert full_net.control_scales == [1.0] * 13
    assert full_net.ctx['image'].qtransformer.pe_layer is None
    full_net.ctx['image'].fp16 = True                                  # dead attribute app.py sets (:119)
    rpi = full_net.ctx['image'].imencoder.layers[0].blocks[0].attn.relative_position_index
    assert rpi.dtype == torch.int64 and rpi.shape == (144, 144) and int(rpi[0, 0]) == 11 * 23 + 11
    sd = full_net.state_dict()
    full_net.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("steps,nreal", [(50, 50), (10, 10), (30, 31)])
def test_ddim_schedule_host(full_net, golden, steps, nreal):
    from lib.model_zoo.ddim import DDIMSampler
    full_net.to('cpu')
    s = DDIMSampler(full_net)
    for eta in (0.0, 0.5):
        s.make_schedule(steps, ddim_eta=eta, verbose=False)
        tag = f"ddim.s{steps}.eta{eta}."
        assert len(s.ddim_timesteps) == nreal
        np.testing.assert_array_equal(s.ddim_timesteps, golden[tag + "timesteps"])
        np.testing.assert_allclose(s.ddim_alphas, golden[tag + "alphas"], rtol=1e-6)
        np.testing.assert_allclose(s.ddim_alphas_prev, golden[tag + "alphas_prev"], rtol=1e-6)
        np.testing.assert_allclose(s.ddim_sigmas, golden[tag + "sigmas"], rtol=1e-5, atol=1e-12)
    tab = s._coef_table(2.0)
    assert tab.shape == (nreal, 5) and float(tab[0, 4]) == 2.0


def test_product_path_has_no_cpu_fallback(full_net):
    """CPU tensors must be refused loudly, never computed with torch ops"""
    full_net.to('cpu')
    with pytest.raises(RuntimeError):
        full_net.ctx_encode(torch.rand(1, 3, 64, 64), 'image')
    with pytest.raises(RuntimeError):
        full_net.vae_decode(torch.randn(1, 4, 8, 8), 'image')
    with pytest.raises(RuntimeError):
        full_net.apply_model({'type': 'image', 'x': torch.randn(1, 4, 8, 8)}, torch.tensor([1]),
                             {'type': 'image', 'c': torch.randn(1, 148, 768)})


def test_oracle_is_not_imported_by_product():
    pkg = os.path.join(REPO, "prompt-free-diffusion_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(root, fn)).read()
                assert "pfd_oracle" not in src and "import oracle" not in src and "from oracle" not in src, fn


# ------------------------------------------------------------------------------------------------
# C ABI
# ------------------------------------------------------------------------------------------------
def _header_functions():
    src = open(os.path.join(REPO, "include", "pfd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pfd_[a-z0-9_]+)\s*\(", src)))


def test_cabi_exports_every_declared_symbol():
    from lib.hip import binding
    names = _header_functions()
    assert len(names) >= 18
    assert set(names) == set(binding.SIGNATURES.keys())
    lib = ctypes.CDLL(binding.lib_path())
    for n in names:
        assert hasattr(lib, n), n
    lib.pfd_abi_version.restype = ctypes.c_int32
    assert lib.pfd_abi_version() == binding.ABI_VERSION
    assert binding.load().pfd_groupnorm_ws_bytes(2, 320, 4096) > 0


def test_cabi_rejects_bad_arguments_without_a_gpu():
    """argument validation happens before any launch, so it is testable on CPU"""
    from lib.hip import binding
    lib = binding.load()
    d = binding.PfdGemmDesc()
    assert lib.pfd_gemm_f16(ctypes.byref(d), None) == -1           # null pointers -> PFD_EINVAL
    assert lib.pfd_gemm_f16(None, None) == -1
    a = binding.PfdAttnDesc()
    assert lib.pfd_attention_f16(ctypes.byref(a), None) == -1
    with pytest.raises(binding.PfdError):
        binding.check(-2, "unit")


# ---- checkpoint formats / converters (lib/weights_io.py) --------------------------------------------
def test_keymaps_match_the_reference_converters(state_spec):
    """derived sdwebui / diffusers key maps == what tools/model_conversion.py's mover classes emit
    (tests/golden/keymaps.json, dumped by oracle/make_keymap_golden.py from the reference itself)"""
    from lib import weights_io as W
    gold = json.load(open(os.path.join(REPO, "tests", "golden", "keymaps.json")))
    keys = list(state_spec.keys())
    uk = [k for k in keys if k.startswith('diffuser.image.')]
    assert set(map(tuple, W.ldm_unet_keymap(uk, context_name='text'))) == set(map(tuple, gold["sdwebui_unet"]))
    assert set(map(tuple, W.diffusers_unet_keymap(uk, context_name='text'))) == \
        set(map(tuple, gold["sdhuggingface_diffuser_to_pfd_mover"]))
    vk = [k[len('vae.image.'):] for k in keys if k.startswith('vae.image.')]
    mine = set((a, b, 'unsqueeze_hw') if f else (a, b) for a, b, f in W.diffusers_vae_keymap(vk))
    assert mine == set(map(tuple, gold["sdhuggingface_vae_to_pfd_mover"]))
    # the converters move tensors under those maps (and reshape the VAE attention linears to 1x1 convs)
    sd = {frm: torch.full((2, 2), float(i)) for i, (frm, _, _) in enumerate(W.diffusers_vae_keymap(vk))}
    out = W.convert_vae(sd, vk)
    assert set(out) == set(vk) and out['decoder.mid.attn_1.q.weight'].shape == (2, 2, 1, 1)


def test_invalidate_packed_covers_data_edits():
    """edits through `param.data` move neither the pointer nor `_version`: invalidate_packed() drops the
    kernel-layout copies and bumps the generation every packed-cache / hipGraph key includes"""
    from lib.hip import layers as L
    lin = L.Linear(4, 4)
    sig0 = L._sig(lin.weight)
    lin.weight.data.mul_(2.0)
    assert L._sig(lin.weight) == sig0                      # the blind spot
    lin.__dict__["_pk_cache"] = {"w": (sig0, "stale")}
    g = L.generation()
    L.invalidate_packed(lin)
    assert L.generation() == g + 1 and "_pk_cache" not in lin.__dict__ and L._sig(lin.weight) != sig0


def test_safetensors_hot_swap_is_strict_and_in_place(tmp_path):
    from safetensors.torch import save_file
    from lib import weights_io as W

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ctx = torch.nn.ModuleDict({'image': torch.nn.Linear(3, 2)})
            self.diffuser = torch.nn.ModuleDict({'image': torch.nn.ModuleDict(
                {'context_blocks': torch.nn.ModuleList([torch.nn.Linear(2, 2)]), 'data_blocks': torch.nn.Linear(2, 1)})})
            self.ctl = torch.nn.Linear(4, 4)
    net = Tiny().half()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    w = net.diffuser['image'].context_blocks[0].weight
    ptr, ver = w.data_ptr(), w._version
    # an old-style checkpoint: context blocks stored under diffuser.text (app.py:149-153), fp32 on disk
    sd = {k.replace('diffuser.image.context_blocks.', 'diffuser.text.context_blocks.'): torch.randn_like(v, dtype=torch.float32)
          for k, v in net.state_dict().items() if k.startswith('diffuser.')}
    f = str(tmp_path / "unet.safetensors")
    save_file(sd, f)
    assert W.load_diffuser(net, f) == len(sd)
    assert w.data_ptr() == ptr and w._version > ver and w.dtype == torch.float16      # in place, cache key moved
    assert torch.equal(w, sd['diffuser.text.context_blocks.0.weight'].half())
    assert all(torch.equal(net.state_dict()[k], before[k]) for k in before if not k.startswith('diffuser.'))
    # strictness: a missing key, an unexpected key and a wrong shape all raise and modify nothing
    snap = {k: v.clone() for k, v in net.state_dict().items()}
    for broken in ({k: v for k, v in list(sd.items())[1:]}, dict(sd, **{'diffuser.image.extra': torch.zeros(1)}),
                   dict(sd, **{'diffuser.image.data_blocks.bias': torch.zeros(3)})):
        save_file(broken, f)
        with pytest.raises(RuntimeError):
            W.load_diffuser(net, f)
        assert all(torch.equal(net.state_dict()[k], snap[k]) for k in snap)
    # a whole-composite checkpoint (keys outside `diffuser.` that the net knows) is tolerated like app.py:139-156
    # tolerates it (`sd.update(sd_extra)`): out-of-prefix keys are left alone, unknown ones still fail
    full = dict(sd, **{'ctl.weight': torch.full((4, 4), 7.0), 'ctx.image.bias': torch.zeros(2)})
    save_file(full, f)
    ctl_before = net.ctl.weight.detach().clone()
    assert W.load_diffuser(net, f) == len(sd) and torch.equal(net.ctl.weight, ctl_before)
    save_file(dict(full, **{'nonsense.key': torch.zeros(1)}), f)
    with pytest.raises(RuntimeError):
        W.load_diffuser(net, f)
    # .pth goes through the same path; unknown extensions fail like app.py:91
    torch.save({k: torch.ones_like(v) for k, v in net.ctl.state_dict().items()}, str(tmp_path / "ctl.pth"))
    W.load_ctl(net, str(tmp_path / "ctl.pth"))
    assert float(net.ctl.weight.detach().min()) == 1.0
    with pytest.raises(AssertionError):
        W.load_sd_from_file(str(tmp_path / "x.bin"))


def test_serving_validates_at_submit_and_isolates_failures(monkeypatch):
    """ADVICE r02 (serving.py): a malformed request is refused on the caller's thread before it can be merged into
    somebody else's batch; if a coalesced batch still fails, its requests are re-run one by one and only the culprit
    receives the exception; a request left behind the close() sentinel is failed, never left pending.  The device work
    is stubbed (host logic only)."""
    import threading
    import torch
    from lib import serving

    class _Pipe:
        def __init__(self, net):
            pass

        def enable_graph(self, on):
            pass

    monkeypatch.setattr(serving, "PromptFreePipeline", _Pipe)
    srv = serving.PromptFreeServer(object(), use_graph=False, max_batch=4)
    calls = []

    def fake_generate(batch):
        calls.append([r.seed for r in batch])
        if any(r.seed == 13 for r in batch):
            raise RuntimeError("bad request 13")
        return [torch.full((r.n, 1), float(r.seed)) for r in batch]

    srv._generate = fake_generate
    img = torch.rand(1, 3, 64, 64)
    for bad in (dict(image=torch.rand(3, 64, 64)), dict(image=img, uncond=torch.zeros(1, 77, 768)),
                dict(image=img, control=torch.zeros(1, 3, 32, 32)), dict(image=img, n_samples=9),
                dict(image=img.int()), dict(image=img, height=100)):
        kw = dict(height=64, width=64)
        kw.update(bad)
        with pytest.raises(ValueError):
            srv.submit(**kw)
    gate = threading.Event()
    srv.call(lambda n: gate.wait(30))
    f1 = srv.submit(img, 1, 64, 64, seed=1)
    f2 = srv.submit(img, 1, 64, 64, seed=13)
    f3 = srv.submit(img, 1, 64, 64, seed=3)
    f4 = srv.submit(img, 1, 64, 64, seed=4, eta=0.5)       # eta > 0 never shares a batch
    gate.set()
    assert float(f1.result(30)[0, 0]) == 1.0 and float(f3.result(30)[0, 0]) == 3.0 and float(f4.result(30)[0, 0]) == 4.0
    with pytest.raises(RuntimeError, match="bad request 13"):
        f2.result(30)
    assert calls[0] == [1, 13, 3] and sorted(calls[1:4]) == [[1], [3], [13]] and calls[4] == [4]
    # close(): anything still queued behind the sentinel is failed
    gate2 = threading.Event()
    srv.call(lambda n: gate2.wait(30))
    srv._q.put(None)
    late = serving._Request(image=img, n=1, height=64, width=64, steps=1, scale=2.0, eta=0.0, seed=5,
                            future=serving.Future())
    srv._q.put(late)
    gate2.set()
    srv.close()
    with pytest.raises(RuntimeError, match="closed"):
        late.future.result(5)


def test_tracked_launch_list_parses_under_the_gpu_tests_reader():
    """profiles/unet_c2_gemm_shapes.txt is read by GPU-only tests (tests/test_hip_kernels_fullsize.py) and by
    `selftest --replay`: run the SAME reader and the same record filters here, so a regenerated file (new fields, lost
    shapes) fails the CPU suite instead of silently breaking the GPU one (round 4: 19-field unpack of a 22-field file)."""
    import test_hip_kernels_fullsize as FS
    from lib.hip import ops
    recs = FS._records()
    assert len(recs) >= 55 and all(len(r) == len(ops.TRACE_FIELDS) for r in recs)
    raw_widths = {len(line.split()) for line in open(FS.SHAPES) if line.strip()}
    assert raw_widths <= {ops.TRACE_FIELDS_ABI7, ops.TRACE_FIELDS_ABI8, len(ops.TRACE_FIELDS)}, raw_widths
    src = open(os.path.join(REPO, "prompt-free-diffusion_amd", "csrc", "selftest.cpp")).read()
    assert f"std::array<long, {len(ops.TRACE_FIELDS)}>" in src and f"n != {len(ops.TRACE_FIELDS)}" in src, \
        "selftest --replay reads a different number of fields than ops.TRACE_FIELDS"
    for r in recs:     # geometry the rebuilt problems rely on
        assert r.M > 0 and r.N > 0 and r.K % 64 == 0 and not r.bias_per_row
        if r.ksize:
            assert r.M == r.B * r.Ho * r.Wo and r.K == r.ksize * r.ksize * r.Cin and not r.k_split and not r.zero_rows
        if r.k_split:
            assert r.k_split % 64 == 0 and 0 < r.k_split < r.K
        assert 0 <= r.zero_rows < r.M
        if r.gn_out:
            assert r.N in (320, 640, 1280) and r.M % 64 == 0 and r.act != FS.ACT_GEGLU
    assert sum(1 for r in recs if r.k_split) >= 5 and sum(1 for r in recs if r.zero_rows) >= 3 and \
        sum(1 for r in recs if r.gn_out) >= 10
    alias = {"ks": "ksize", "st": "stride"}
    for flt, variants in FS.FORCED + FS.FORCED_ABI8:    # every forced-variant filter still finds its record
        assert any(all(getattr(r, alias.get(k, k)) == v for k, v in flt.items()) for r in recs), flt
        assert variants
    assert all(FS._name(r) for r in recs)


def test_gn_partials_reference_layout():
    """the fp32 restatement of PfdGemmDesc.gn_out the GPU test compares against (include/pfd_hip.h, ABI 8): slot
    [(slab * (N / 160) + n / 160) * 16 + (n % 160) / (N / 32)] over 64-row slabs"""
    import test_hip_kernels_fullsize as FS
    for N in (320, 640, 1280):
        y = torch.randn(128, N, generator=torch.Generator().manual_seed(N)).half()
        ref, used = FS.gn_partials_ref(y, N)
        assert used == 160 // (N // 32) and tuple(ref.shape) == (2, N // 160, used, 2)
        slab, n = 1, N - 1
        g0 = n - n % (N // 32)
        blk = y[64:128, g0:g0 + N // 32].double()
        got = ref[slab, n // 160, (n % 160) // (N // 32)]
        assert abs(float(got[0] - blk.sum())) < 1e-9 and abs(float(got[1] - (blk * blk).sum())) < 1e-9


def test_stale_groupnorm_statistics_are_dropped_on_rewrite():
    """producer statistics ride on the tensor OBJECT (ops.set_gn_stats); any op that writes into that object again
    without fresh ones must drop them (ops._written), or groupnorm() would normalise with the previous tensor's sums"""
    from lib.hip import ops
    t = torch.zeros(64, 320, dtype=torch.float16)
    st = torch.zeros(1, 2, 16, 2)
    ops.set_gn_stats(t, st)
    assert ops.get_gn_stats(t) is st
    assert ops._written(t) is t and ops.get_gn_stats(t) is None            # rewritten without statistics
    st2 = torch.ones(1, 2, 16, 2)
    assert ops.get_gn_stats(ops._written(t, st2)) is st2                   # rewritten with fresh ones
    assert ops.get_gn_stats(t[:32]) is None                                # views do not inherit
    assert ops._written(None) is None
    # round 6: a write through a VIEW of the carrying tensor (ControlNet: add(h[b:b+1], g, out=h[b:b+1])) or a torch
    # in-place op invalidates statistics and a producer-computed normalised copy alike -- the shared version counter
    h = torch.zeros(128, 320, dtype=torch.float16)
    ops.set_gn_stats(h, st)
    ops.set_normed(h, ("k", 1e-5, True), torch.ones(128, 320))
    assert ops.get_gn_stats(h) is st and ops.get_normed(h, ("k", 1e-5, True)) is not None
    ops._written(h[0:64])                                                  # the library wrote through a view
    assert ops.get_gn_stats(h) is None and ops.get_normed(h, ("k", 1e-5, True)) is None
    ops.set_gn_stats(h, st)
    h.add_(1)                                                              # torch wrote in place
    assert ops.get_gn_stats(h) is None
    import time
    big = torch.zeros(32768, 320, dtype=torch.float16)
    t0 = time.perf_counter()
    for _ in range(100):
        ops._written(big)
    assert time.perf_counter() - t0 < 0.05, "ops._written must not touch the tensor's elements (it once iterated its rows)"


def test_unet_next_norm_hints(full_net):
    """UNetModel2D_Next._next_norms (round 5): which data layers are told the GroupNorm that reads their output ALONE, so that
    their last convolution's split-K reduction can also write that norm's result (PfdGemmDesc.gnf_y).  A ResBlock followed by a
    context layer gets SpatialTransformer.norm (eps 1e-6, no SiLU), one followed by a plain ResBlock gets in_layers[0] + SiLU
    (eps 1e-5); layers whose output enters a skip concat, convolution-only layers and -- with ControlNet residuals, which are
    added in between (pfd.py:515) -- the middle block's output get none."""
    from lib.model_zoo.openaimodel import ResBlock
    unet = full_net.diffuser['image']
    order = list(unet.i_order) + list(unet.m_order) + list(unet.o_order)
    hints = unet._next_norms(unet, False)
    assert len(hints) == 18 and all(order[i] == 'd' for i in hints)
    n_in, n_mid = len(unet.i_order), len(unet.m_order)
    d_pos = [i for i, t in enumerate(order) if t == 'd']
    blocks = dict(zip(d_pos, unet.data_blocks))
    for i, (norm, silu) in hints.items():
        assert isinstance(blocks[i][len(blocks[i]) - 1], ResBlock)          # only a trailing ResBlock can use the hint
        nxt = next(j for j in range(i + 1, len(order)) if order[j] in ('d', 'c', 'load_hidden_feature'))
        assert order[nxt] != 'load_hidden_feature'                            # never the half of a skip concat
        if order[nxt] == 'c':
            assert not silu and norm.eps == 1e-6 and norm.num_channels == blocks[i][len(blocks[i]) - 1].out_channels
        else:
            assert silu and norm.eps == 1e-5 and norm is blocks[nxt][0].in_layers[0]
    # 8 of them are at widths the fused reduction serves (1280 channels: the 16^2 / 8^2 levels)
    assert sum(1 for norm, _ in hints.values() if norm.num_channels == 1280) == 8
    with_ctl = unet._next_norms(unet, True)
    last_mid = n_in + n_mid - 1
    assert set(hints) - set(with_ctl) == ({last_mid} & set(hints)) and len(with_ctl) >= 17
    assert unet._next_norms(unet, False) is hints                             # cached
