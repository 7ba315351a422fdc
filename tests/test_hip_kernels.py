"""GPU (-m gpu): individual C-ABI entry points against plain fp32 torch formulas (these are
floating-point kernels), including the ragged / edge shapes the path produces."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(shape, generator=g) * scale).half().cuda()


def close(a, ref, tol=1e-2):
    e = float((a.float().cpu() - ref.float().cpu()).abs().max() / max(1.0, float(ref.abs().max())))
    assert e <= tol, e
    return e


@pytest.mark.parametrize("M,N,K", [(1, 320, 1280), (8, 1280, 320), (301, 200, 192), (4096, 640, 320), (148, 768, 768)])
def test_gemm_bias_act_residual(M, N, K):
    from lib.hip import ops
    a, w, b, r = _dev(M, K), _dev(N, K, scale=K ** -0.5), _dev(N), _dev(M, N)
    y = ops.gemm(a, w, bias=b, res=r, act=ops.ACT_SILU)
    ref = F.silu(a.float() @ w.float().t() + b.float()) + r.float()
    close(y, ref)


@pytest.mark.parametrize("M,N,K", [(1, 320, 1280), (301, 160, 192), (1184, 640, 768), (2500, 1280, 320),
                                   (5000, 384, 64), (8192, 1920, 640), (70000, 320, 320)])
def test_gemm_wave_specialised_variant(M, N, K):
    """variant 48: 8 MFMA waves + 4 loader waves per block (ragged M, one tile, one K step, many tiles)"""
    from lib.hip import ops
    a, w, b, r = _dev(M, K), _dev(N, K, scale=K ** -0.5), _dev(N), _dev(M, N)
    rv = _dev((M + 63) // 64, N)
    tile = 1000 + 100 * 48
    y = ops.gemm(a, w, bias=b, res=r, act=ops.ACT_SILU, rowvec=rv, rows_per_rv=64, tile=tile)
    idx = torch.arange(M, device="cuda") // 64
    ref = F.silu(a.float() @ w.float().t() + b.float() + rv.float()[idx]) + r.float()
    close(y, ref, 4e-3)
    y2 = ops.gemm(a, w, tile=tile)                       # no epilogue operands at all
    close(y2, a.float() @ w.float().t(), 4e-3)
    for _ in range(3):                                    # same launch again: no state left behind
        assert torch.equal(ops.gemm(a, w, tile=tile), y2)
    if K >= 256:
        close(ops.gemm(a, w, bias=b, tile=tile + 2), a.float() @ w.float().t() + b.float(), 4e-3)   # split-K 2


def test_gemm_geglu_and_strided_views():
    from lib.hip import ops
    from lib.model_zoo.attention import GEGLU
    m = GEGLU(320, 1280).half().cuda()
    x = _dev(77, 320)
    y = m.hip(x)
    h = F.linear(x.float(), m.proj.weight.float(), m.proj.bias.float())
    a, g = h.chunk(2, -1)
    close(y, a * F.gelu(g))
    # A operand = column slice of a wider matrix, W = column slice (two-source 1x1 skip conv)
    big, w = _dev(50, 640), _dev(96, 640, scale=0.05)
    y = ops.gemm(big[:, 320:], w[:, 320:], k=320)
    close(y, big[:, 320:].float() @ w[:, 320:].float().t())


@pytest.mark.parametrize("M,C,K", [(64, 320, 320), (200, 640, 640), (1024, 1280, 1280)])
def test_gemm_transposed_tail_and_fused_qkv(M, C, K):
    """ABI 3: q | k token-major + v transposed out of one launch == three separate projections."""
    from lib.hip import ops
    a, w, b = _dev(M, K), _dev(3 * C, K, scale=K ** -0.5), _dev(3 * C)
    vt = torch.zeros((C, M + 8), dtype=torch.float16, device="cuda")
    qk = ops.gemm(a, w, bias=b, out_t=vt[:, :M], n_split=2 * C)
    ref = a.float() @ w.float().t() + b.float()
    assert qk.shape == (M, 2 * C)
    close(qk, ref[:, :2 * C])
    close(vt[:, :M], ref[:, 2 * C:].t())
    assert float(vt[:, M:].abs().max()) == 0.0          # pad columns untouched
    with pytest.raises(Exception):                       # no slow path behind it: N % 160 and N % 128 != 0 is loud
        ops.gemm(a, w[:3 * 96], out_t=vt[:96, :M], n_split=192)
    qk2 = ops.gemm(a, w[:3 * 128], out_t=vt[:128, :M], n_split=256)   # 128-wide tiles serve N % 128 == 0
    close(qk2, (a.float() @ w[:3 * 128].float().t())[:, :256])
    close(vt[:128, :M], (a.float() @ w[256:384].float().t()).t())


def test_self_attention_fused_projection_matches_split_path():
    from lib.model_zoo.attention import CrossAttention
    torch.manual_seed(0)
    m = CrossAttention(320, heads=8, dim_head=40).half().cuda()
    x = _dev(2 * 64, 320)
    y = m.hip(x, 2, 64)
    xf = x.float().view(2, 64, 320)
    q, k, v = (F.linear(xf, getattr(m, n).weight.float()).view(2, 64, 8, 40).transpose(1, 2)
               for n in ("to_q", "to_k", "to_v"))
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2 * 64, 320)
    ref = F.linear(o, m.to_out[0].weight.float(), m.to_out[0].bias.float())
    close(y, ref)


@pytest.mark.parametrize("cin,cout,k,s,ups,hw", [(64, 96, 3, 1, False, (9, 7)), (128, 64, 3, 2, False, (10, 8)),
                                                  (64, 64, 3, 1, True, (5, 6)), (4, 320, 3, 1, False, (8, 8)),
                                                  (3, 192, 4, 4, False, (10, 13)), (320, 4, 3, 1, False, (8, 8))])
def test_conv_layers(cin, cout, k, s, ups, hw):
    from lib.hip import layers as L
    from lib.hip import ops
    torch.manual_seed(1)
    pad = 0 if k == 4 else 1
    conv = L.Conv2d(cin, cout, k, stride=s, padding=pad).half().cuda()
    x = _dev(2, cin, *hw)
    xin = ops.to_nhwc(x)
    if k == 4:  # patch-embed: right/bottom zero pad to a multiple of 4
        H, W = hw
        y = conv.hip(xin, out_hw=((H + 3) // 4, (W + 3) // 4))
        xr = F.pad(x.float(), (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))
    else:
        y = conv.hip(xin, ups=ups)
        xr = F.interpolate(x.float(), scale_factor=2, mode="nearest") if ups else x.float()
    ref = F.conv2d(xr, conv.weight.float(), conv.bias.float(), stride=s, padding=pad)
    close(ops.to_nchw(y, torch.float32), ref)


@pytest.mark.parametrize("B,hw,c1,c2,cout,silu", [(2, 32, 128, 64, 320, True), (1, 64, 64, 0, 160, True),
                                                  (3, 32, 320, 0, 320, False)])
def test_conv_groupnorm_prologue(B, hw, c1, c2, cout, silu):
    """GroupNorm32 -> SiLU -> conv3x3 (openaimodel.py:254-259) with the normalise / activate pass inside the patch
    kernel's input staging: identical to the two-launch form, and equal to torch fp32 within fp16 noise."""
    from lib.hip import layers as L
    from lib.hip import ops
    torch.manual_seed(4)
    C = c1 + c2
    gn = L.GroupNorm(32, C, eps=1e-5).half().cuda()
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.2)
        gn.bias.normal_(0.0, 0.2)
    conv = L.Conv2d(C, cout, 3, padding=1).half().cuda()
    x1 = _dev(B, hw, hw, c1) * 1.5 + 0.3
    x2 = _dev(B, hw, hw, c2, seed=5) if c2 else None
    res, e = _dev(B, hw, hw, cout, seed=6), _dev(B, cout, seed=7)
    assert ops.conv_gn_fusable(B, hw, hw, c1, c2, cout)
    table = gn.hip_table(x1, x2)
    assert table.shape == (B, 2, C) and table.dtype == torch.float32
    y = conv.hip(x1, rowvec=e, res=res, gn=(table, x2, silu))
    two = conv.hip(gn.hip(x1, x2, silu=silu), rowvec=e, res=res)
    assert torch.equal(y, two)
    cat = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    hn = F.group_norm(cat, 32, gn.weight.float(), gn.bias.float(), 1e-5)
    hn = F.silu(hn) if silu else hn
    ref = F.conv2d(hn, conv.weight.float(), conv.bias.float(), padding=1) + e.float()[:, :, None, None]
    close(y.float(), ref.permute(0, 2, 3, 1) + res.float())
    # shapes the patch kernel does not take are refused, never served by another kernel without the prologue
    x8 = _dev(1, 8, 8, 64)
    c8 = L.Conv2d(64, 160, 3, padding=1).half().cuda()
    g8 = L.GroupNorm(32, 64).half().cuda()
    assert not ops.conv_gn_fusable(1, 8, 8, 64, 0, 160)
    with pytest.raises(RuntimeError):
        c8.hip(x8, gn=(g8.hip_table(x8), None, True))


@pytest.mark.parametrize("B,hw,cin,silu,res", [(8, 16, 1280, True, False), (8, 8, 1280, True, False), (8, 8, 2560, True, False),
                                               (4, 16, 640, False, True)])
def test_conv_with_groupnorm_in_its_splitk_reduction(B, hw, cin, silu, res):
    """conv3x3 -> GroupNorm32 (-> SiLU) (`h = in_layers(x) + emb_out; h = out_layers(h)`, openaimodel.py:254-272) with the
    normalisation inside the convolution's split-K reduction launch (PfdGemmDesc.gnf_y, ABI 9): the same bits as the
    two-call form (raw tensor too, when kept), equal to torch fp32 within fp16 noise; shapes the library does not split
    are declined (None) without a launch."""
    from lib.hip import layers as L
    from lib.hip import ops
    torch.manual_seed(11)
    cout = 1280
    conv = L.Conv2d(cin, cout, 3, padding=1).half().cuda()
    gn = L.GroupNorm(32, cout, eps=1e-5).half().cuda()
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.2)
        gn.bias.normal_(0.0, 0.2)
    x = _dev(B, hw, hw, cin, scale=1.0)
    e = _dev(B, cout, seed=7)
    r = _dev(B, hw, hw, cout, seed=6) if res else None
    fused = conv.hip_gn(x, gn, silu=silu, keep_raw=res, rowvec=e, res=r)
    assert fused is not None, "the 8^2 / 16^2 convolutions split K: the fused reduction must serve them"
    raw, y = fused
    h = conv.hip(x, rowvec=e, res=r)
    two = gn.hip(h, silu=silu)
    assert torch.equal(y, two)
    assert (raw is None) == (not res) and (raw is None or torch.equal(raw, h))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), conv.weight.float(), conv.bias.float(), padding=1) + e.float()[:, :, None, None]
    if res:
        ref = ref + r.float().permute(0, 3, 1, 2)
    ref = F.group_norm(ref, 32, gn.weight.float(), gn.bias.float(), 1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1)
    close(y.float(), ref)
    # a convolution the library does not split (64^2) / a width the fused form is not built for (cpg 10): declined, no launch
    c64 = L.Conv2d(320, 320, 3, padding=1).half().cuda()
    g64 = L.GroupNorm(32, 320).half().cuda()
    assert c64.hip_gn(_dev(2, 64, 64, 320), g64) is None


@pytest.mark.parametrize("B,cin,res", [(8, 640, False), (8, 1280, True), (4, 320, False)])
def test_conv_with_groupnorm_in_its_splitk_reduction_at_20_channels_per_group(B, cin, res):
    """the same fusion for the 640-channel norms of the 32x32 level (round 6): served; the raw tensor (when kept) is bitwise the
    two-call form's; the normalised tensor agrees with it except for last-bit roundings (the two-call GroupNorm of this
    shape sums its statistics in another order) and with torch fp32 within fp16 noise"""
    from lib.hip import layers as L
    torch.manual_seed(12)
    cout, hw = 640, 32
    conv = L.Conv2d(cin, cout, 3, padding=1).half().cuda()
    gn = L.GroupNorm(32, cout, eps=1e-5).half().cuda()
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.2)
        gn.bias.normal_(0.0, 0.2)
    x = _dev(B, hw, hw, cin, scale=1.0)
    e = _dev(B, cout, seed=7)
    r = _dev(B, hw, hw, cout, seed=6) if res else None
    fused = conv.hip_gn(x, gn, silu=True, keep_raw=res, rowvec=e, res=r)
    assert fused is not None, "the 32^2 convolutions split K: the fused reduction must serve them"
    raw, y = fused
    h = conv.hip(x, rowvec=e, res=r)
    two = gn.hip(h, silu=True)
    assert float((y != two).float().mean()) < 1e-3 and float((y.float() - two.float()).abs().max()) < 1e-2
    assert (raw is None) == (not res) and (raw is None or torch.equal(raw, h))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), conv.weight.float(), conv.bias.float(), padding=1) + e.float()[:, :, None, None]
    if res:
        ref = ref + r.float().permute(0, 3, 1, 2)
    ref = F.silu(F.group_norm(ref, 32, gn.weight.float(), gn.bias.float(), 1e-5)).permute(0, 2, 3, 1)
    close(y.float(), ref)


def test_groupnorm_concat_and_layernorm():
    from lib.hip import ops
    x1, x2 = _dev(2, 6, 5, 320), _dev(2, 6, 5, 640, seed=3)
    g, b = _dev(960, scale=0.2) + 1, _dev(960, scale=0.1)
    y = ops.groupnorm(x1, g, b, 32, 1e-5, x2=x2, silu=True)
    cat = torch.cat([x1, x2], -1).float().permute(0, 3, 1, 2)
    ref = F.silu(F.group_norm(cat, 32, g.float(), b.float(), 1e-5)).permute(0, 2, 3, 1)
    close(y, ref)
    t = _dev(37, 1536)
    g, b = _dev(1536, scale=0.2) + 1, _dev(1536, scale=0.1)
    close(ops.layernorm(t, g, b), F.layer_norm(t.float(), (1536,), g.float(), b.float()))


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 8, 64, 64, 160), (2, 8, 300, 148, 40), (1, 8, 144, 1000, 96),
                                          (1, 8, 148, 148, 96), (3, 8, 256, 256, 80),
                                          (2, 1, 320, 256, 512)])   # the last: VAE mid-block attention (K18)
def test_attention(B, H, Nq, Nk, D):
    from lib.hip import ops
    Cd = H * D
    q, k, v = _dev(B, Nq, Cd), _dev(B, Nk, Cd, seed=1), _dev(B, Nk, Cd, seed=2)
    Nkp = (Nk + 7) // 8 * 8
    vt = torch.zeros((Cd, B, Nkp), dtype=torch.float16, device='cuda')
    vt[:, :, :Nk] = v.permute(2, 0, 1)
    o = ops.attention(q, k, vt, B, H, Nq, Nk, D, D ** -0.5, ldq=Cd, ldk=Cd, ldvt=B * Nkp, q_bs=Nq * Cd,
                      k_bs=Nk * Cd, vt_bs=Nkp)
    sp = lambda t: t.float().view(B, -1, H, D).permute(0, 2, 1, 3)  # noqa: E731
    ref = ((sp(q) @ sp(k).transpose(-1, -2)) * D ** -0.5).softmax(-1) @ sp(v)
    close(o.view(B, Nq, H, D), ref.permute(0, 2, 1, 3), 5e-3)


@pytest.mark.parametrize("slices", ["2", "4"])
def test_vae_attention512_both_slice_forms(slices, monkeypatch):
    """d = 512 fused VAE attention: the two-slice instantiation (default since round 4) and the four-slice one, at a ragged
    query tile and at a 64^2-token shape, vs fp32 torch (PFD_ATTN512_SLICES is read per launch)"""
    from lib.hip import ops
    monkeypatch.setenv("PFD_ATTN512_SLICES", slices)
    for (B, Nq, Nk) in ((2, 200, 96), (1, 4096, 4096)):
        D = Cd = 512
        q, k, v = _dev(B, Nq, Cd), _dev(B, Nk, Cd, seed=1), _dev(B, Nk, Cd, seed=2)
        vt = v.permute(2, 0, 1).contiguous()
        o = ops.attention(q, k, vt, B, 1, Nq, Nk, D, D ** -0.5, ldq=Cd, ldk=Cd, ldvt=B * Nk, q_bs=Nq * Cd,
                          k_bs=Nk * Cd, vt_bs=Nk)
        ref = ((q.float() @ k.float().transpose(-1, -2)) * D ** -0.5).softmax(-1) @ v.float()
        close(o.view(B, Nq, Cd), ref, 5e-3)


def test_self_attention_at_36864_tokens():
    """app.py:197-207 allows a 1536 x 1536 output: the UNet's first self-attention (attention.py:188-199) then runs over
    192 x 192 = 36 864 tokens with 8 heads of d = 40 -- the largest attention problem the path can be asked for
    (tests/test_hip_parity.py::test_tall_and_wide_resolutions only checks that such a request is finite).  Numeric check of
    the fused kernel at that size against an fp32 reference on the same device, computed in query chunks (the score
    matrix of one chunk is 8 x 2048 x 36 864 fp32 = 2.4 GB), and against itself on a permuted key order (softmax is
    permutation invariant: a size-independent property)."""
    from lib.hip import ops
    B, H, N, D = 1, 8, 36864, 40
    Cd = H * D
    q, k, v = _dev(B, N, Cd), _dev(B, N, Cd, seed=1), _dev(B, N, Cd, seed=2)
    q[:, 17] *= 6.0                                           # one query row with a peaked distribution
    vt = v.permute(2, 0, 1).contiguous()
    o = ops.attention(q, k, vt, B, H, N, N, D, D ** -0.5, ldq=Cd, ldk=Cd, ldvt=B * N, q_bs=N * Cd, k_bs=N * Cd, vt_bs=N)
    sp = lambda t: t.float().view(N, H, D).permute(1, 0, 2)   # noqa: E731
    qf, kf, vf = sp(q[0]), sp(k[0]), sp(v[0])
    worst = 0.0
    for c0 in range(0, N, 2048):
        ref = ((qf[:, c0:c0 + 2048] @ kf.transpose(-1, -2)) * D ** -0.5).softmax(-1) @ vf     # [H, 2048, D]
        got = o[c0:c0 + 2048].view(-1, H, D).permute(1, 0, 2).float()
        worst = max(worst, float((got - ref).abs().max() / max(1.0, float(ref.abs().max()))))
    print(f"[kernels] self-attention at 36 864 tokens (8 heads, d = 40): scaled max-abs error {worst:.3e} vs fp32")
    assert worst <= 5e-3
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(9)).cuda()
    o2 = ops.attention(q, k[:, perm].contiguous(), v[:, perm].permute(2, 0, 1).contiguous(), B, H, N, N, D, D ** -0.5,
                       ldq=Cd, ldk=Cd, ldvt=B * N, q_bs=N * Cd, k_bs=N * Cd, vt_bs=N)
    assert float((o2.float() - o.float()).abs().max()) <= 5e-3


def test_vae_attention_beyond_16k_tokens():
    """tall / wide outputs up to 1536 px (app.py:197-207): the VAE mid attention sees up to 36 864 tokens;
    rows longer than 16 384 take the streaming softmax.  Checked against torch fp32 on the same device."""
    from lib.hip import ops
    from lib.model_zoo.autokl_modules import AttnBlock
    s = _dev(3, 36864, scale=3.0)
    y = ops.softmax_rows(s, 0.044)
    close(y, torch.softmax(s.float() * 0.044, -1), 1e-3)
    torch.manual_seed(3)
    m = AttnBlock(512).half().cuda()
    H, W = 192, 128                                     # 24 576 tokens = a 1536 x 1024 image
    x = _dev(1, H, W, 512)
    y = m.hip(x)
    xf = x.float().permute(0, 3, 1, 2)
    hn = F.group_norm(xf, 32, m.norm.weight.float(), m.norm.bias.float(), 1e-6)
    q, k, v = (F.conv2d(hn, getattr(m, n).weight.float(), getattr(m, n).bias.float()).flatten(2) for n in "qkv")
    att = torch.softmax(torch.bmm(q.transpose(1, 2), k) * 512 ** -0.5, dim=2)
    o = torch.bmm(v, att.transpose(1, 2)).view(1, 512, H, W)
    ref = xf + F.conv2d(o, m.proj_out.weight.float(), m.proj_out.bias.float())
    close(y, ref.permute(0, 2, 3, 1))


def test_cfg_ddim_step_matches_formula():
    from lib.hip import ops
    B, C, h, w = 2, 4, 8, 8
    eps = _dev(2 * B, h, w, C)
    x = torch.randn(B, C, h, w, device='cuda')
    noise = torch.randn(B, C, h, w, device='cuda')
    a_t, a_prev, sig, scale = 0.4, 0.6, 0.1, 2.0
    coef = torch.tensor([a_t, a_prev, sig, math.sqrt(1 - a_t), scale], device='cuda')
    xp, p0, xin = ops.cfg_ddim_step(eps, 2, x, coef, noise=noise)
    e = eps.float().permute(0, 3, 1, 2)
    e = e[:B] + scale * (e[B:] - e[:B])
    r0 = (x - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)
    rp = math.sqrt(a_prev) * r0 + math.sqrt(1 - a_prev - sig ** 2) * e + sig * noise
    close(p0, r0, 1e-5)
    close(xp, rp, 1e-5)
    close(xin[:B].permute(0, 3, 1, 2), rp, 2e-3)
    close(xin[B:].permute(0, 3, 1, 2), rp, 2e-3)


def test_errors_are_loud():
    from lib.hip import binding, ops
    with pytest.raises(binding.PfdError):
        ops.gemm(_dev(8, 100), _dev(8, 100))          # K % 64 != 0 -> PFD_ESHAPE
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(8, 64, dtype=torch.float16), torch.zeros(8, 64, dtype=torch.float16))  # CPU tensors


def test_operand_validation_is_loud():
    """elementwise ops do not broadcast silently, caller-provided outputs are validated, the narrow
    convolution honours out=, a control hint that does not broadcast to the batch raises"""
    from lib.hip import layers as L
    from lib.hip import ops
    a, b = _dev(4, 64), _dev(2, 64)
    with pytest.raises(ValueError):
        ops.add(a, b)
    with pytest.raises(ValueError):
        ops.axpby(a, 1.0, b, 1.0)
    with pytest.raises(ValueError):
        ops.gemm(_dev(8, 64), _dev(160, 64), out=torch.empty((8, 128), dtype=torch.float16, device="cuda"))
    with pytest.raises(TypeError):
        ops.gemm(_dev(8, 64), _dev(160, 64), out=torch.empty((8, 160), dtype=torch.float32, device="cuda"))
    conv = L.Conv2d(4, 64, 3, padding=1).half().cuda()
    x = _dev(1, 8, 8, 4)
    out = torch.zeros((1, 8, 8, 64), dtype=torch.float16, device="cuda")
    y = conv.hip(x, out=out)
    assert y.data_ptr() == out.data_ptr() and float(out.abs().max()) > 0
    close(out, conv.hip(x))


def test_repeated_launches_give_the_same_bits():
    """Forty launches of the same problem must return the same bits (round 4: the LayerNorm-folded GEMMs did not --
    lanes 48-63 of the first accumulator the fold touched occasionally saw stale row statistics, depending on where the
    kernel happened to sit in the code object; profiles/r04_ln_fold_determinism.log).  Every epilogue family: plain,
    LayerNorm fold in (GEGLU / fused q|k|v with transposed tail / plain), statistics out (LayerNorm, GroupNorm), zero rows,
    two sources, split-K, and a 3x3 patch convolution with GroupNorm statistics."""
    from lib.hip import ops
    g = torch.Generator().manual_seed(1)
    M, K = 8192, 320
    x = torch.randn((M, K), generator=g).half().cuda()
    x2 = torch.randn((M, 320), generator=g).half().cuda()
    r = torch.randn((M, 320), generator=g).half().cuda()
    st = ops.ln_rowstats(x)

    def W(n, k=K):
        return (torch.randn((n, k), generator=g) * 0.05).half().cuda()
    w320, w960, w2560, w640k = W(320), W(960), W(2560), W(320, 640)
    b320, b960, b2560 = (torch.randn((n,), generator=g).half().cuda() for n in (320, 960, 2560))
    cs = {n: w.float().sum(1).contiguous() for n, w in ((320, w320), (960, w960), (2560, w2560))}
    img = torch.randn((2, 32, 32, 320), generator=g).half().cuda()
    wc = W(320, 9 * 320)
    cases = {
        "plain": lambda: ops.gemm(x, w320, bias=b320, res=r),
        "ln fold": lambda: ops.gemm(x, w320, bias=b320, ln=(st, cs[320], 1e-5)),
        "ln fold GEGLU": lambda: ops.gemm(x, w2560, bias=b2560, act=ops.ACT_GEGLU, ln=(st, cs[2560], 1e-5)),
        "ln fold q|k|v + transposed tail": lambda: ops.gemm(x, w960, bias=b960, ln=(st, cs[960], 1e-5), n_split=640,
                                                            out_t=torch.empty((320, M), dtype=torch.float16, device='cuda')),
        "ln statistics out": lambda: ops.gemm(x, w320, bias=b320, res=r, ln_out=True)[0],
        "GroupNorm statistics out": lambda: ops.gemm(x, w320, bias=b320, res=r, gn_out=True),
        "zero rows": lambda: ops.gemm(x[: M // 2], w320, bias=b320, res=r, zero_rows=M // 2),
        "two sources": lambda: ops.gemm(x, w640k, bias=b320, a2=x2),
        "split-K": lambda: ops.gemm(x[:512], w320, bias=b320, tile=1000 + 2200 + 2),
        "3x3 patch conv + GroupNorm statistics": lambda: ops.conv(img, wc, 3, bias=b320, gn_out=True),
    }
    for name, fn in cases.items():
        outs = [fn().clone() for _ in range(40)]
        torch.cuda.synchronize()
        same = sum(int(torch.equal(o, outs[0])) for o in outs)
        assert same == 40, f"{name}: only {same}/40 launches returned the same bits"
