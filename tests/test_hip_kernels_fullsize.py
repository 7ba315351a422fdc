"""GPU (-m gpu): parity at the shapes BASELINE.json quotes the headline number on (config C2: 512x512, batch 4
-> UNet batch 8, 64x64 latent; C3: + ControlNet; C5: 96x96 latent, non-zero unconditional context).

  * every distinct GEMM / convolution launch of one C2 UNet pass (profiles/unet_c2_gemm_shapes.txt, recorded
    from the real model with PFD_TRACE_GEMM) against an fp32 torch formula, at full size, through the C ABI --
    plus the same problems under every forced tile variant (256/128/64-row tiles, the 3x3 patch kernel) and
    split-K factor, i.e. the kernel instances that carry ~80 % of the bench's FLOPs;
  * whole stages at full size against the CPU oracle (oracle/pfd_oracle.py, pinned to the reference by
    tests/test_oracle_golden.py): UNet `apply_model` at [8,4,64,64] (pfd.py:314-365), ControlNet-guided eps
    (pfd.py:466-528, controlnet.py:302-324), VAE decode of a 64x64 latent (autokl_modules.py:535-568), SeeCoder
    at 512x512 (seecoder.py:567-575), a C5-shape UNet call with a non-zero unconditional context
    (app.py:238-241).  Samples are independent on this path, so the HIP side always runs the FULL batch (the
    tile / split-K choices depend on it) and the oracle checks a subset of the samples to keep the host time
    in seconds.

Tolerance (BASELINE.json north_star): fp16 path within 1e-2 of the fp32 CPU reference; gate = max-abs error
<= 1e-2 * max(1, max|ref|) and rel-L2 <= 1e-2; measured numbers are printed.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import REPO, seeded_sd

pytestmark = pytest.mark.gpu
SHAPES = os.path.join(REPO, "profiles", "unet_c2_gemm_shapes.txt")
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU, ACT_GEGLU = range(5)


def rel(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).abs().max() / max(1.0, float(ref.abs().max()))), float((a - ref).norm() / ref.norm())


def check(name, a, ref, tol=1e-2):
    e, l2 = rel(a, ref)
    raw = float((a.detach().double().cpu() - ref.detach().double().cpu()).abs().max())
    print(f"[fullsize] {name}: scaled max-abs {e:.3e}, rel-L2 {l2:.3e} (tol {tol:g}); unscaled max-abs {raw:.3e}, "
          f"max|ref| {float(ref.detach().double().abs().max()):.3f}")
    assert e <= tol and l2 <= tol, f"{name}: max-abs {e}, rel-L2 {l2} > {tol}"
    return e, l2


def _rand(shape, scale, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).half()


def _records():
    from lib.hip import ops
    return ops.parse_launch_records(SHAPES)


def gn_partials_ref(y, N):
    """PfdGemmDesc.gn_out of a stored [M, N] f16 tensor, from its own values in fp32/fp64: float2 [M / 64][N / 160][16]
    = (sum x, sum x^2) per 64-row slab and group of N / 32 channels; 160 / (N / 32) slots of a tile are used"""
    M = y.shape[0]
    cpg = N // 32
    used = 160 // cpg
    v = y.double().view(M // 64, 64, N // 160, used, cpg)
    return torch.stack([v.sum(dim=(1, 4)), (v * v).sum(dim=(1, 4))], dim=-1), used     # [M/64, N/160, used, 2]


def check_gn_partials(name, y, stats):
    """the statistics a gn_out launch emitted vs the sums of the tensor it stored (include/pfd_hip.h, ABI 8)"""
    M, N = y.shape
    assert stats is not None, f"{name}: the record says gn_out but the launch emitted no statistics"
    assert tuple(stats.shape) == (M // 64, N // 160, 16, 2) and stats.dtype == torch.float32
    ref, used = gn_partials_ref(y, N)
    got = stats[:, :, :used].double()
    # sums of 64 * N/32 f16 values in fp32: relative to the slab's sum of squares / its root
    scale_s = ref[..., 1].sqrt().clamp_min(1.0) * (64 * (N // 32)) ** 0.5
    e_sum = float(((got[..., 0] - ref[..., 0]).abs() / scale_s).max())
    e_sq = float(((got[..., 1] - ref[..., 1]).abs() / ref[..., 1].clamp_min(1.0)).max())
    assert e_sum <= 3e-4 and e_sq <= 3e-4, (name, "gn_out partials", e_sum, e_sq)
    return max(e_sum, e_sq)


class Problem:
    """one recorded launch rebuilt with seeded operands; `.run(tile)` -> HIP result, `.ref` -> fp32 formula.
    ABI-8 fields: k_split -> two-source operand [x | x2] (openaimodel.py:274 after pfd.py:356); zero_rows -> that many
    all-zero operand rows in front (app.py:236 zero unconditional context: result epi(0)); gn_out -> the launch also emits
    GroupNorm partials of what it stored (openaimodel.py:200-226), checked by check_gn_partials"""

    def __init__(self, rec, seed):
        from lib.hip import layers as L
        from lib.model_zoo.attention import GEGLU
        M, N, K, act, ks, rpr = rec.M, rec.N, rec.K, rec.act, rec.ksize, rec.rows_per_rv
        B, H, W, Cin, Ho, Wo = rec.B, rec.H, rec.W, rec.Cin, rec.Ho, rec.Wo
        assert not rec.bias_per_row
        self.rec, self.M, self.N, self.K, self.act, self.ks = rec, M, N, K, act, ks
        self.bias = _rand((N,), 0.5, seed + 1) if rec.has_bias else None
        self.rows_per_rv = rpr
        n_rv = (M + rpr - 1) // rpr if rec.has_rowvec else 0
        self.rowvec = _rand((max(n_rv, 1), N), 0.5, seed + 2) if rec.has_rowvec else None
        n_out = N // 2 if act == ACT_GEGLU else N
        self.res = _rand((M, n_out), 1.0, seed + 3) if rec.has_res else None
        self.x2, self.zero_rows, self.gn_out = None, rec.zero_rows, bool(rec.gn_out)
        if ks > 0:
            assert not rec.k_split and not rec.zero_rows
            self.x = _rand((B, H, W, Cin), 1.0, seed)                          # NHWC
            w4 = _rand((N, Cin, ks, ks), K ** -0.5, seed + 4)                   # torch conv layout
            self.w = L.pack_conv_weight(w4)
            self.geom = (rec.stride, rec.pad, bool(rec.ups), B, Ho, Wo)
            xr = self.x.float().permute(0, 3, 1, 2)
            if rec.ups:
                xr = F.interpolate(xr, scale_factor=2, mode="nearest")
            cols = F.unfold(xr, ks, padding=rec.pad, stride=rec.stride)         # [B, Cin*ks*ks, Ho*Wo], c-major
            y = cols.transpose(1, 2).reshape(M, K) @ w4.float().reshape(N, K).t()
        else:
            rows = M - rec.zero_rows
            xfull = _rand((rows, K), 1.0, seed)
            if rec.k_split:
                self.x = xfull[:, :rec.k_split].contiguous()
                self.x2 = xfull[:, rec.k_split:].contiguous()
            else:
                self.x = xfull
            w = _rand((N, K), K ** -0.5, seed + 4)
            if act == ACT_GEGLU:      # logical rows: x half then gate half; packed by the module that owns the layout
                m = GEGLU(K, N // 2).half().cuda()
                with torch.no_grad():
                    m.proj.weight.copy_(w)
                    m.proj.bias.copy_(self.bias if self.bias is not None else torch.zeros(N).cuda())
                self.w, self.bias_packed = m._pk()
            else:
                self.w = w
            y = xfull.float() @ w.float().t()
            if rec.zero_rows:
                y = torch.cat([torch.zeros((rec.zero_rows, N), device="cuda"), y])
        if self.bias is not None:
            y = y + self.bias.float()
        if self.rowvec is not None:
            idx = torch.arange(M, device="cuda") // rpr
            y = y + self.rowvec.float()[idx]
        if act == ACT_GEGLU:
            a, g = y.chunk(2, -1)
            y = a * F.gelu(g)
        elif act == ACT_SILU:
            y = F.silu(y)
        elif act == ACT_GELU:
            y = F.gelu(y)
        if self.res is not None:
            y = y + self.res.float()
        self.ref = y

    def run(self, tile=0, check_stats=True):
        from lib.hip import ops
        if self.ks > 0:
            st, pad, ups, B, Ho, Wo = self.geom
            r4 = None if self.res is None else self.res.view(B, Ho, Wo, -1)
            y = ops.conv(self.x, self.w, self.ks, stride=st, pad=pad, ups=ups, bias=self.bias, rowvec=self.rowvec,
                         res=r4, act=self.act, tile=tile, gn_out=self.gn_out,
                         rows_per_rv=self.rows_per_rv if self.rowvec is not None else None)
            stats = ops.get_gn_stats(y)
            y = y.view(self.M, -1)
        else:
            bias = self.bias_packed if self.act == ACT_GEGLU else self.bias
            y = ops.gemm(self.x, self.w, bias=bias, rowvec=self.rowvec, rows_per_rv=self.rows_per_rv, res=self.res,
                         act=self.act, tile=tile, a2=self.x2, zero_rows=self.zero_rows, gn_out=self.gn_out)
            stats = ops.get_gn_stats(y)
        self.stats_err = check_gn_partials(_name(self.rec), y, stats) if (self.gn_out and check_stats) else None
        return y


def _name(rec):
    kind = f"conv{rec.ksize}x{rec.ksize}/s{rec.stride}{'/ups' if rec.ups else ''}" if rec.ksize else "linear"
    return (f"{kind} M{rec.M} N{rec.N} K{rec.K} act{rec.act}{' +emb' if rec.has_rowvec else ''}{' +res' if rec.has_res else ''}"
            f"{' k_split%d' % rec.k_split if rec.k_split else ''}{' zero_rows%d' % rec.zero_rows if rec.zero_rows else ''}"
            f"{' +gn_out' if rec.gn_out else ''}")


def test_unet_c2_launch_list_vs_torch():
    """every distinct launch of a C2 UNet pass, heuristic tile choice (what the bench runs) -- including the ABI-8 forms:
    the two-source skip GEMMs, the zero-row out-projections and the GroupNorm partials of every gn_out launch"""
    recs = _records()
    assert len(recs) >= 55
    assert sum(1 for r in recs if r.k_split) >= 5 and sum(1 for r in recs if r.zero_rows) >= 3 and \
        sum(1 for r in recs if r.gn_out) >= 10, "the tracked launch list lost its ABI-8 records"
    worst, worst_st = 0.0, 0.0
    for i, rec in enumerate(recs):
        p = Problem(rec, 1000 + 17 * i)
        e, l2 = rel(p.run(), p.ref)
        worst = max(worst, e, l2)
        worst_st = max(worst_st, p.stats_err or 0.0)
        print(f"[fullsize] {_name(rec)}: scaled max-abs {e:.2e}, rel-L2 {l2:.2e}"
              + (f", gn_out partials {p.stats_err:.1e}" if p.stats_err is not None else ""))
        assert e <= 4e-3 and l2 <= 4e-3, (_name(rec), e, l2)
    print(f"[fullsize] {len(recs)} distinct launches, worst error {worst:.2e}, worst gn_out partial error {worst_st:.1e}")


def _tile(variant, splits=0):
    """pfd_gemm_f16_ex tile code of the wide-tile path: 1000 + 100 * variant + splits (include/pfd_hip.h)"""
    return 1000 + 100 * variant + splits


# (record filter, [(variant, splits)]): 44 = 256-row tiles, 24 = 128, 22 = 64, 99 = 3x3 patch kernel,
# 48 = 256-row tiles with 4 dedicated loader waves, 98 = the patch kernel with 4 loader waves
FORCED = [
    (dict(M=32768, N=320, K=2880, ks=3), [(99, 1), (99, 2), (98, 1), (98, 2), (44, 1), (44, 2), (24, 1), (22, 1), (48, 1), (48, 2)]),
    (dict(M=32768, N=320, K=8640, ks=3), [(99, 1), (99, 4), (98, 1), (44, 1)]),
    (dict(M=8192, N=640, K=5760, ks=3, ups=0), [(99, 1), (99, 2), (99, 4), (98, 2), (98, 4), (44, 2), (24, 2), (22, 1)]),
    (dict(M=2048, N=1280, K=11520, ks=3, ups=0), [(99, 1), (99, 4), (99, 8), (98, 1), (98, 8), (24, 4), (24, 8), (22, 4)]),
    (dict(M=2048, N=1280, K=23040, ks=3), [(99, 8), (24, 8)]),
    (dict(M=512, N=1280, K=11520, ks=3, st=1), [(24, 8), (22, 4), (22, 1)]),
    (dict(M=512, N=1280, K=11520, ks=3, st=2), [(24, 8), (22, 4), (48, 8)]),
    (dict(M=32768, N=640, K=5760, ks=3, ups=1), [(44, 1), (24, 1), (48, 1)]),
    (dict(M=8192, N=1280, K=11520, ks=3, ups=1), [(44, 2), (24, 2)]),
    (dict(M=32768, N=2560, K=320, ks=0), [(44, 1), (24, 1), (22, 1), (48, 1)]),   # GEGLU
    (dict(M=2048, N=10240, K=1280, ks=0), [(44, 1), (24, 1)]),                  # GEGLU
    (dict(M=32768, N=320, K=1280, ks=0), [(44, 1), (24, 1), (22, 1)]),
    (dict(M=32768, N=960, K=320, ks=0), [(44, 1), (24, 1), (22, 1), (48, 1)]),
    (dict(M=8192, N=640, K=2560, ks=0), [(44, 1), (44, 2), (24, 2), (22, 1), (48, 1), (48, 2)]),
    (dict(M=2048, N=1280, K=5120, ks=0), [(24, 4), (22, 2), (22, 4)]),
    (dict(M=512, N=1280, K=1280, ks=0), [(22, 1), (22, 4), (24, 2)]),
]


FORCED_ABI8 = [   # the two-source skip GEMMs, the zero-row out-projections and a gn_out linear under the forced tiles
    (dict(M=32768, N=320, K=960, ksize=0, k_split=640), [(44, 1), (24, 1), (22, 1)]),
    (dict(M=8192, N=640, K=1920, ksize=0, k_split=1280), [(44, 1), (24, 2), (22, 1)]),
    (dict(M=2048, N=1280, K=2560, ksize=0, k_split=1280), [(24, 4), (22, 2)]),
    (dict(M=32768, N=320, K=320, ksize=0, zero_rows=16384), [(44, 1), (24, 1), (22, 1)]),
    (dict(M=2048, N=1280, K=1280, ksize=0, zero_rows=1024), [(24, 2), (22, 1), (22, 4)]),
    (dict(M=8192, N=640, K=640, ksize=0, gn_out=1), [(44, 1), (24, 1), (22, 1)]),
]


def test_unet_c2_forced_tile_variants_and_split_k():
    """the same full-size problems under every tile variant / split-K factor the heuristic may pick"""
    recs = _records()
    alias = {"ks": "ksize", "st": "stride"}
    n = 0
    for i, (flt, variants) in enumerate(FORCED + FORCED_ABI8):
        match = [r for r in recs if all(getattr(r, alias.get(k, k)) == v for k, v in flt.items())]
        assert match, flt
        # prefer the richest epilogue (residual / embedding row vector / statistics) among the matching records
        rec = max(match, key=lambda r: (r.has_res, r.has_rowvec, r.has_bias, r.gn_out))
        p = Problem(rec, 5000 + 31 * i)
        base = p.run()
        for variant, splits in variants:
            y = p.run(_tile(variant, splits))
            e, l2 = rel(y, p.ref)
            print(f"[fullsize] {_name(rec)} variant {variant} splits {splits}: max-abs {e:.2e}, rel-L2 {l2:.2e}"
                  + (f", gn_out partials {p.stats_err:.1e}" if p.stats_err is not None else ""))
            assert e <= 4e-3 and l2 <= 4e-3, (_name(rec), variant, splits, e, l2)
            assert float((y.float() - base.float()).abs().max()) <= 4e-3 * max(1.0, float(p.ref.abs().max()))
            n += 1
    print(f"[fullsize] {n} forced (variant, split-K) launches checked")


# ------------------------------------------------------------------------------------------------------
# whole stages at BASELINE sizes against the CPU oracle
# ------------------------------------------------------------------------------------------------------
def _threads():
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))


def _cfg_inputs(n, hw, seed, uncond=None):
    """UNet batch of a CFG step: [uncond x n | cond x n] (ddim.py:145-149), shared timestep"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, 4, hw, hw), generator=g)
    cond = torch.randn((1, 148, 768), generator=g).repeat(n, 1, 1)
    uc = torch.zeros_like(cond) if uncond is None else uncond.repeat(n, 1, 1)
    return torch.cat([x, x]), torch.full((2 * n,), 621, dtype=torch.long), torch.cat([uc, cond])


def test_unet_c2_batch8_vs_oracle(net, param_shapes):
    """(a-5) one CFG UNet call at the C2 shape [8,4,64,64]: the reference-API path (`apply_model`) and the
    sampler's path (NHWC, precomputed embedding table, zero-unconditional shortcut) vs the oracle"""
    import pfd_oracle as O
    from lib.hip import ops
    _threads()
    x, t, c = _cfg_inputs(4, 64, 11)
    sd = seeded_sd(param_shapes, "diffuser.image.")
    eps = net.apply_model({'type': 'image', 'x': x.cuda().half()}, t.cuda(), {'type': 'image', 'c': c.cuda().half()})
    assert eps.shape == (8, 4, 64, 64) and eps.dtype == torch.float16
    ctx = net.prepare_context(c.cuda().half())
    ctx.zero_lead = 4
    unet = net.diffuser['image']
    emb_all, _ = unet.emb_projections(t[:1].cuda())
    eps2 = ops.to_nchw(net.apply_model_nhwc('image', ops.to_nhwc(x.cuda()), t.cuda(), 'image', ctx,
                                            emb_table=emb_all[0:1]), torch.float16)
    for s in (0, 5):    # an unconditional (all-zero context) and a conditional sample
        ref = O.unet_apply(sd, "diffuser.image.", x[s:s + 1], t[s:s + 1], c[s:s + 1])
        check(f"C2 UNet eps sample {s} (apply_model)", eps[s:s + 1], ref)
        check(f"C2 UNet eps sample {s} (sampler path, zero-uncond shortcut)", eps2[s:s + 1], ref)
    assert float((eps2.float() - eps.float()).abs().max()) < 2e-2     # two routes to the same numbers
    # the GroupNorm prologue of the patch convolution (optional path, 36 of the 44 ResBlock convolutions at this shape)
    # against standalone GroupNorm launches: same statistics, same affine map -> the same bits
    # (both sides with a statistics PASS per GroupNorm: the prologue's table comes from that pass, while the default path
    #  takes the producers' sums -- the same numbers in another summation order, compared below at fp16 level)
    from lib.model_zoo.openaimodel import ResBlock
    # (and without the GroupNorm inside the split-K reduction, which since round 6 also serves the 32^2 level the prologue lives
    #  on: its statistics are the same numbers in yet another summation order)
    was, was_ps, was_fr = ResBlock.fuse_groupnorm, ops.GN_PSTATS, ResBlock.fuse_reduce_groupnorm
    ops.GN_PSTATS = False
    ResBlock.fuse_reduce_groupnorm = False
    try:
        eps_pass = net.apply_model({'type': 'image', 'x': x.cuda().half()}, t.cuda(), {'type': 'image', 'c': c.cuda().half()})
        ResBlock.fuse_groupnorm = not was
        eps3 = net.apply_model({'type': 'image', 'x': x.cuda().half()}, t.cuda(), {'type': 'image', 'c': c.cuda().half()})
    finally:
        ResBlock.fuse_groupnorm, ops.GN_PSTATS, ResBlock.fuse_reduce_groupnorm = was, was_ps, was_fr
    assert torch.equal(eps3, eps_pass)
    if was_ps:   # producers' statistics (default) vs a statistics pass per GroupNorm
        d = float((eps.float() - eps_pass.float()).abs().max())
        print(f"[fullsize] C2 UNet eps, GroupNorm statistics from the producers vs statistics passes: max|diff| {d:.3e}")
        assert d <= 1e-2


def test_controlnet_c3_eps_vs_oracle(net, param_shapes):
    """(a-13) ControlNet residuals at 64x64 + the guided UNet eps (config C3 shape), full batch 8"""
    import pfd_oracle as O
    _threads()
    x, t, c = _cfg_inputs(4, 64, 12)
    hint = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(7))
    sd_u = seeded_sd(param_shapes, "diffuser.image.")
    sd_c = seeded_sd(param_shapes, "ctl.")
    eps = net.apply_model({'type': 'image', 'x': x.cuda().half()}, t.cuda(),
                          {'type': 'image', 'c': c.cuda().half(), 'control': hint.cuda().half()})
    s = 6
    res = O.controlnet_apply(sd_c, "ctl.", x[s:s + 1], hint, t[s:s + 1], c[s:s + 1])
    ref = O.unet_apply(sd_u, "diffuser.image.", x[s:s + 1], t[s:s + 1], c[s:s + 1], control=res)
    check(f"C3 ControlNet-guided eps sample {s}", eps[s:s + 1], ref)
    plain = O.unet_apply(sd_u, "diffuser.image.", x[s:s + 1], t[s:s + 1], c[s:s + 1])
    assert float((ref - plain).abs().max()) > 1e-2        # the control really changes the answer


def test_vae_decode_c2_vs_oracle(net, param_shapes):
    """(a-17) decode of a batch of four 64x64 latents -> 512x512 images, one of them vs the oracle"""
    import pfd_oracle as O
    _threads()
    z = torch.randn((4, 4, 64, 64), generator=torch.Generator().manual_seed(13)) * 0.9
    img = net.vae_decode(z.cuda().half(), 'image')
    assert img.shape == (4, 3, 512, 512)
    ref = O.vae_decode(seeded_sd(param_shapes, "vae.image."), "vae.image.", z[2:3])
    check("C2 VAE decode sample 2 (512x512)", img[2:3], ref)


def test_seecoder_512_vs_oracle(net, param_shapes):
    """(a-14..16) SeeCoder context of a 512x512 reference image"""
    import pfd_oracle as O
    _threads()
    img = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(1234))
    ctx = net.ctx_encode(img.cuda().half(), 'image')
    ref = O.seecoder_encode(seeded_sd(param_shapes, "ctx.image."), "ctx.image.", img)
    check("SeeCoder ctx 512x512", ctx, ref)


def test_unet_c5_shape_nonzero_uncond_vs_oracle(net, param_shapes):
    """config C5: 96x96 latent (768x768 output), batch 2 -> UNet batch 4, NON-zero unconditional context
    (SeeCoder-Anime pads a fixed [77,768] tensor to 148 tokens, app.py:238-241): the zero-context shortcut
    must not trigger and self-attention runs over 9216 tokens"""
    import pfd_oracle as O
    from lib.hip import ops
    _threads()
    g = torch.Generator().manual_seed(5)
    ug = torch.zeros((1, 148, 768))
    ug[:, :77] = torch.randn((1, 77, 768), generator=g) - 0.107
    x, t, c = _cfg_inputs(2, 96, 14, uncond=ug)
    ctx = net.prepare_context(c.cuda().half())
    ctx.zero_lead = 0                                       # what the sampler decides for a non-zero uncond
    eps = ops.to_nchw(net.apply_model_nhwc('image', ops.to_nhwc(x.cuda()), t.cuda(), 'image', ctx), torch.float16)
    sd = seeded_sd(param_shapes, "diffuser.image.")
    for s in (1, 2):
        ref = O.unet_apply(sd, "diffuser.image.", x[s:s + 1], t[s:s + 1], c[s:s + 1])
        check(f"C5 UNet eps sample {s} (96x96 latent)", eps[s:s + 1], ref)


def test_wide_512x768_unet_and_vae_vs_oracle(net, param_shapes):
    """app.py:197-207 lets H and W differ (any multiples of 64 in [512, 1536]): a 64 x 96 latent (non-square:
    6144 tokens, image width 96 is not a patch-kernel width at the 64^2 level) -- UNet eps and the decoded
    image against the oracle, not just finiteness"""
    import pfd_oracle as O
    _threads()
    g = torch.Generator().manual_seed(21)
    x = torch.randn((1, 4, 64, 96), generator=g)
    cond = torch.randn((1, 148, 768), generator=g)
    xx, t, c = torch.cat([x, x]), torch.full((2,), 401, dtype=torch.long), torch.cat([torch.zeros_like(cond), cond])
    eps = net.apply_model({'type': 'image', 'x': xx.cuda().half()}, t.cuda(), {'type': 'image', 'c': c.cuda().half()})
    ref = O.unet_apply(seeded_sd(param_shapes, "diffuser.image."), "diffuser.image.", xx[1:], t[1:], c[1:])
    check("UNet eps at a 64x96 latent (512x768 image)", eps[1:], ref)
    z = x * 0.9
    img = net.vae_decode(z.cuda().half(), 'image')
    assert img.shape == (1, 3, 512, 768)
    check("VAE decode 512x768", img, O.vae_decode(seeded_sd(param_shapes, "vae.image."), "vae.image.", z))


def test_1024x768_unet_eps_vs_oracle(net, param_shapes):
    """One numeric check above 768^2: apply_model at a 128 x 96 latent (1024 x 768 image; 12 288 tokens,
    192 whole 64-key tiles: the software-pipelined d = 40 attention kernel of round 6, patch tiles on 96-wide rows) against the
    oracle on the conditional sample -- test_tall_and_wide_resolutions checks shape / range / finiteness only"""
    import pfd_oracle as O
    _threads()
    g = torch.Generator().manual_seed(22)
    x = torch.randn((1, 4, 128, 96), generator=g)
    cond = torch.randn((1, 148, 768), generator=g)
    xx, t, c = torch.cat([x, x]), torch.full((2,), 601, dtype=torch.long), torch.cat([torch.zeros_like(cond), cond])
    eps = net.apply_model({'type': 'image', 'x': xx.cuda().half()}, t.cuda(), {'type': 'image', 'c': c.cuda().half()})
    ref = O.unet_apply(seeded_sd(param_shapes, "diffuser.image."), "diffuser.image.", xx[1:], t[1:], c[1:])
    check("UNet eps at a 128x96 latent (1024x768 image)", eps[1:], ref)
