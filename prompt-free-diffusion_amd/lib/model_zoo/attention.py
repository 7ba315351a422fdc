"""UNet transformer blocks on the HIP path.

Same module tree / state-dict keys / constructor kwargs as the live classes of the reference's
lib/model_zoo/attention.py (CrossAttention :159-201, GEGLU :44-51, FeedForward :54-71,
BasicTransformerBlock :277-306, SpatialTransformer :309-371), but the forward is token-major
fp16 end to end on hand-written gfx950 kernels:

  GN(eps 1e-6) -> 1x1 conv (GEMM) -> [LN -> fused QK GEMM + V^T GEMM -> flash attention ->
  out-proj GEMM (+bias +residual)] -> [LN -> Q GEMM -> flash attention over cached context K/V^T
  -> out-proj (+residual)] -> [LN -> GEGLU GEMM (x*gelu(gate) in the epilogue) -> GEMM (+residual)]
  -> 1x1 conv (+bias + block input)

No `b c h w <-> b (hw) c` rearranges exist (attention.py:361,368): NHWC *is* token-major.
The score matrix is never materialised (attention.py:188-199 does, 2.1 GB per layer at C2).
"""
import numpy as np
import numpy.random as npr
import torch
import torch.nn as nn

from ..hip import binding
from ..hip import layers as L
from ..hip import ops


def exists(v):
    return v is not None


def default(v, d):
    return v if v is not None else (d() if callable(d) else d)


def Normalize(in_channels):
    return L.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class ContextKV:
    """Per-request cache of the step-invariant cross-attention operands: for every
    CrossAttention layer that consumes `context` [B, Nk, Cctx], K = to_k(context) and
    V^T = to_v(context)^T are computed once (SURVEY §8a saving (ii)) instead of once per DDIM
    step.  Context rows are zero-padded to a multiple of 8 tokens so every V^T row is 16-byte
    aligned (pfd_attention_f16 contract)."""

    def __init__(self, context):
        if not context.is_cuda:
            raise RuntimeError("HIP path: context must be on the GPU (no CPU fallback)")
        B, Nk, Cd = context.shape
        self.B, self.Nk, self.Cd = B, Nk, Cd
        self.Nkp = (Nk + 7) // 8 * 8
        ctx = torch.zeros((B, self.Nkp, Cd), dtype=torch.float16, device=context.device)
        ctx[:, :Nk] = context.to(torch.float16)
        self.ctx2d = ctx.view(B * self.Nkp, Cd)
        self._kv = {}
        # number of LEADING batch entries whose context is all zero (set by the sampler for the
        # unconditional half of a CFG batch, app.py:236): for those rows K = V = 0, so cross-attention
        # returns exactly to_out.bias and the q / attention / out-projection work is skipped
        # (SURVEY 8a exact saving (i); verified bit-identical in tests/test_hip_parity.py)
        self.zero_lead = 0

    def get(self, attn):
        ent = self._kv.get(id(attn))
        if ent is None:
            k = attn.to_k.hip(self.ctx2d)          # [B*Nkp, inner]
            vt = attn.to_v.hip_t(self.ctx2d)       # [inner, B*Nkp]
            ent = self._kv[id(attn)] = (k, vt)
        return ent


class ContextMix:
    """Several contexts mixed at every context layer (pfd.py:366-386 `context_mixing`):
    'attention': h = sum_i ratio_i * SpatialTransformer_i(x, c_i), ratios normalised to 1;
    'layer':     one context per layer, drawn with numpy's global RNG `npr.choice(n, p=ratios)`.
    items: [(net that owns the context blocks, ContextKV, ratio)]."""

    def __init__(self, items, mixing_type='attention'):
        if mixing_type not in ('attention', 'layer'):
            raise ValueError(f"unknown mixing_type {mixing_type!r}")
        self.nets = [n for n, _, _ in items]
        self.contexts = [as_context_kv(c) for _, c, _ in items]
        r = np.array([float(r) for _, _, r in items])
        self.ratios = r / r.sum()
        self.mixing_type = mixing_type

    def mix(self, modules, h, emb):
        assert len(modules) == len(self.contexts)
        if self.mixing_type == 'layer':
            ni = npr.choice(len(modules), p=self.ratios)
            return modules[ni].hip(h, emb, self.contexts[ni])
        out = None
        for m, c, r in zip(modules, self.contexts, self.ratios):
            hi = m.hip(h, emb, c)
            out = ops.axpby(hi, r) if out is None else ops.axpby(hi, r, out, 1.0)
        return out


def as_context_kv(context):
    if isinstance(context, ContextMix):
        return context
    if context is None or isinstance(context, ContextKV):
        return context
    if isinstance(context, (list, tuple)):
        context = context[0]
    return ContextKV(context)


class GEGLU(nn.Module, L._Packed):
    """proj: Linear(dim_in, 2*dim_out); y = x * gelu(gate).  The packed weight interleaves x/gate
    rows (pairs for the wide-tile kernel: x0 x1 g0 g1 | x2 x3 g2 g3 ... = the four columns one lane of
    its accumulator owns; blocks of 32 otherwise) so both halves of an output column meet in the epilogue."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = L.Linear(dim_in, dim_out * 2)
        self.dim_out = dim_out

    def _pk(self):
        def build():
            n = self.dim_out
            w = L._dev16(self.proj.weight)
            b = L._dev16(self.proj.bias)
            gr = int(binding.load().pfd_gemm_geglu_group(2 * n))  # packing granularity of the kernel serving this N
            wi = torch.stack([w[:n].view(n // gr, gr, -1), w[n:].view(n // gr, gr, -1)], 1).reshape(2 * n, -1)
            bi = torch.stack([b[:n].view(n // gr, gr), b[n:].view(n // gr, gr)], 1).reshape(2 * n)
            return wi.contiguous(), bi.contiguous()
        return self._packed("geglu", build, self.proj.weight, self.proj.bias)

    def _pk_ln(self, norm):
        # fixed slot (a GEGLU is always folded with the one LayerNorm in front of it; its parameters are in the signature)
        return self._packed("geglu_ln", lambda: L.fold_layernorm(*self._pk(), norm.weight, norm.bias),
                            self.proj.weight, self.proj.bias, norm.weight, norm.bias)

    def hip(self, x2d, ln=None):
        """ln = (LayerNorm module, partial row sums of x2d): norm3 folded into the projection (x2d un-normalised)"""
        if ln is not None:
            norm, st = ln
            w, cs, b = self._pk_ln(norm)
            return ops.gemm(x2d, w, bias=b, act=ops.ACT_GEGLU, ln=(st, cs, norm.eps))
        w, b = self._pk()
        return ops.gemm(x2d, w, bias=b, act=ops.ACT_GEGLU)

    def forward(self, x):
        y = self.hip(x.to(torch.float16).reshape(-1, x.shape[-1]))
        return y.view(*x.shape[:-1], self.dim_out).to(x.dtype)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        self.glu = glu
        project_in = GEGLU(dim, inner_dim) if glu else nn.Sequential(L.Linear(dim, inner_dim), nn.GELU())
        self.net = nn.Sequential(project_in, nn.Dropout(dropout), L.Linear(inner_dim, dim_out))

    def hip(self, x2d, res=None, ln=None):
        if self.glu:
            h = self.net[0].hip(x2d, ln=ln)
        else:
            h = self.net[0][0].hip(x2d, act=ops.ACT_GELU, ln=ln)
        return self.net[2].hip(h, res=res)

    def forward(self, x):
        y = self.hip(x.to(torch.float16).reshape(-1, x.shape[-1]))
        return y.view(*x.shape[:-1], y.shape[-1]).to(x.dtype)


class CrossAttention(nn.Module, L._Packed):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.inner_dim = inner_dim
        self.to_q = L.Linear(query_dim, inner_dim, bias=False)
        self.to_k = L.Linear(context_dim, inner_dim, bias=False)
        self.to_v = L.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(L.Linear(inner_dim, query_dim), nn.Dropout(dropout))

    def _pk_qk(self):
        return self._packed(
            "qk", lambda: torch.cat([L._dev16(self.to_q.weight), L._dev16(self.to_k.weight)], 0).contiguous(),
            self.to_q.weight, self.to_k.weight)

    def _pk_qkv(self):
        return self._packed(
            "qkv", lambda: torch.cat([L._dev16(self.to_q.weight), L._dev16(self.to_k.weight),
                                      L._dev16(self.to_v.weight)], 0).contiguous(),
            self.to_q.weight, self.to_k.weight, self.to_v.weight)

    def _pk_qkv_ln(self, norm):
        # (one LayerNorm per attention, so the slot name is fixed; its parameters are part of the signature.  The stacked
        #  [Wq; Wk; Wv] is built here and dropped: with the fold on, the plain "qkv" pack is never requested.)
        def build():
            w = torch.cat([L._dev16(self.to_q.weight), L._dev16(self.to_k.weight), L._dev16(self.to_v.weight)], 0)
            return L.fold_layernorm(w.contiguous(), None, norm.weight, norm.bias)
        return self._packed("qkv_ln", build, self.to_q.weight, self.to_k.weight, self.to_v.weight, norm.weight, norm.bias)

    def ln_foldable(self, x, N, context):
        """can the LayerNorm in front of this attention be folded into its first projection?"""
        Cd = self.inner_dim
        if not ops.ln_fold_ok(x.shape[1]):
            return False
        if context is None:
            return N % 8 == 0 and Cd % 160 == 0 and x.shape[1] % 64 == 0
        return Cd % 160 == 0 or Cd % 128 == 0          # to_q on the wide-tile kernels (the fold lives there only)

    def hip(self, x, B, N, context=None, res=None, ln=None, stats_out=False, single=False):
        """x: [B*N, query_dim] tokens (already normalised, or UN-normalised with ln = (LayerNorm, partial row sums of
        x): the norm is then folded into the q | k | v / q projection); context: None (self-attention) or a ContextKV;
        res: residual added by the out-projection epilogue.  -> [B*N, query_dim]; with stats_out also the partial
        row sums of the result (for the next folded LayerNorm).
        single: x / res / the statistics hold ONE copy ([B/2 * N] rows) of a CFG pair whose halves are identical up to here and
        whose unconditional half has an all-zero context (zero_lead == B / 2): the query projection runs on that copy (it IS
        the conditional half), the out-projection re-joins the halves with the residual read once (ops.gemm res_rows) --
        no torch.cat of the tokens, the block input or the statistics.  Returns the full batch."""
        Cd, H, D = self.inner_dim, self.heads, self.dim_head
        if ln is not None and not self.ln_foldable(x, N, context):
            raise ValueError("CrossAttention.hip: ln= passed for a shape the fold does not serve (see ln_foldable)")
        if context is None:
            if N % 8 == 0 and Cd % 160 == 0 and x.shape[1] % 64 == 0:
                # one launch over the shared activation: q | k token-major, v transposed (ABI 3)
                Np = N
                vt = torch.empty((Cd, B * N), dtype=torch.float16, device=x.device)
                if ln is not None:
                    norm, st = ln
                    wq, cs, bq = self._pk_qkv_ln(norm)
                    qk = ops.gemm(x, wq, bias=bq, out_t=vt, n_split=2 * Cd, ln=(st, cs, norm.eps))
                else:
                    qk = ops.gemm(x, self._pk_qkv(), out_t=vt, n_split=2 * Cd)
            elif N % 8 == 0:
                Np = N
                qk = ops.gemm(x, self._pk_qk())             # [M, 2*inner]: q | k
                vt = self.to_v.hip_t(x)                      # [inner, B*N]
            else:  # odd token counts (tiny latents): pad every sample's V^T rows to 16-byte multiples
                qk = ops.gemm(x, self._pk_qk())
                Np = (N + 7) // 8 * 8
                vt = torch.zeros((Cd, B * Np), dtype=torch.float16, device=x.device)
                for b in range(B):
                    self.to_v.hip_t(x[b * N:(b + 1) * N], out=vt[:, b * Np:b * Np + N])
            o = ops.attention(qk, qk[:, Cd:], vt, B, H, N, N, D, self.scale, ldq=2 * Cd, ldk=2 * Cd,
                              ldvt=B * Np, q_bs=N * 2 * Cd, k_bs=N * 2 * Cd, vt_bs=Np)
        else:
            k, vt = context.get(self)
            if context.B != B:
                raise ValueError(f"context batch {context.B} != activation batch {B}")
            z = context.zero_lead
            if single:
                if not (res is not None and 2 * z == B and stats_out and ops.wide_tile_ok(self.to_out[0].out_features, Cd)):
                    raise ValueError("CrossAttention.hip(single=True) needs the zero-context shortcut on half the batch, a residual, "
                                     "statistics and the wide-tile out-projection (SpatialTransformer.hip checks before it asks)")
                w_o, b_o = self.to_out[0]._pk()
                out = torch.empty((B * N, res.shape[1]), dtype=torch.float16, device=res.device)
                st_o = torch.empty((B * N, out.shape[1] // 160, 2), dtype=torch.float32, device=out.device)
                q = self.to_q.hip(x, ln=ln)                       # the one copy = the conditional half
                o = ops.attention(q, k[z * context.Nkp:], vt[:, z * context.Nkp:], B - z, H, N, context.Nk, D,
                                  self.scale, ldq=Cd, ldk=Cd, ldvt=B * context.Nkp, q_bs=N * Cd,
                                  k_bs=context.Nkp * Cd, vt_bs=context.Nkp)
                ops.gemm(o, w_o, bias=b_o, res=res, out=out, zero_rows=z * N, ln_out=st_o, res_rows=z * N)
                return out, st_o
            if 0 < z < B and res is not None:
                # rows of the first z samples: x + bias; the remaining samples: the real thing
                Bc = B - z
                out = torch.empty_like(res)
                w_o, b_o = self.to_out[0]._pk()
                st_o = None
                if stats_out:   # the `x + bias` rows write their statistics with the add (same order as the GEMM epilogue)
                    st_o = torch.empty((B * N, out.shape[1] // 160, 2), dtype=torch.float32, device=out.device)
                q = self.to_q.hip(x[z * N:], ln=None if ln is None else (ln[0], ln[1][z * N:]))
                o = ops.attention(q, k[z * context.Nkp:], vt[:, z * context.Nkp:], Bc, H, N, context.Nk, D,
                                  self.scale, ldq=Cd, ldk=Cd, ldvt=B * context.Nkp, q_bs=N * Cd,
                                  k_bs=context.Nkp * Cd, vt_bs=context.Nkp)
                if ops.wide_tile_ok(w_o.shape[0], Cd):
                    # ONE out-projection launch for the whole batch: the zero-context rows are `zero_rows` of the
                    # operand (their tiles skip the K loop: bias + residual, statistics from the same store pass)
                    ops.gemm(o, w_o, bias=b_o, res=res, out=out, zero_rows=z * N, ln_out=st_o)
                    return (out, st_o) if stats_out else out
                ops.add_rowvec(res[:z * N], b_o, out=out[:z * N], ln_out=None if st_o is None else st_o[:z * N])
                if stats_out:
                    self.to_out[0].hip(o, res=res[z * N:], out=out[z * N:], ln_out=st_o[z * N:])
                    return out, st_o
                self.to_out[0].hip(o, res=res[z * N:], out=out[z * N:])
                return out
            q = self.to_q.hip(x, ln=ln)
            o = ops.attention(q, k, vt, B, H, N, context.Nk, D, self.scale, ldq=Cd, ldk=Cd,
                              ldvt=B * context.Nkp, q_bs=N * Cd, k_bs=context.Nkp * Cd, vt_bs=context.Nkp)
        return self.to_out[0].hip(o, res=res, ln_out=True if stats_out else None)

    def forward(self, x, context=None, mask=None):
        assert mask is None, "mask is not supported on the HIP path"
        B, N, _ = x.shape
        ctx = None if context is None else as_context_kv(context)
        y = self.hip(x.to(torch.float16).reshape(B * N, -1), B, N, ctx)
        return y.view(B, N, -1).to(x.dtype)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        self.disable_self_attn = disable_self_attn
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                                    context_dim=context_dim if disable_self_attn else None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = L.LayerNorm(dim)
        self.norm2 = L.LayerNorm(dim)
        self.norm3 = L.LayerNorm(dim)
        self.checkpoint = checkpoint  # inference only: never used

    def _fold(self, x, N):
        """LayerNorm fold (every LayerNorm of the block feeds a Linear with nothing in between, attention.py:302-306):
        the norm's affine map goes into the consumer GEMM's epilogue, its row statistics come from the epilogue of the
        launch that produced x (ops.gemm(ln_out=...)) -- no LayerNorm launch, no extra pass over the tokens."""
        ctx1 = None if not self.disable_self_attn else 1
        return ops.ln_fold_ok(x.shape[1]) and self.attn1.ln_foldable(x, N, ctx1) and \
            self.attn1.to_out[0].out_features % 160 == 0 and self.attn2.to_out[0].out_features % 160 == 0

    def hip_self(self, x, B, N, context, xs=None):
        """x + attn1(LN(x)): the part of the block that does not see the context (unless disable_self_attn).
        xs: partial row sums of x (ops.gemm(ln_out=...)) or None.  Returns y, or (y, partial row sums of y) when the
        block folds its LayerNorms."""
        c1 = context if self.disable_self_attn else None
        if self._fold(x, N) and (c1 is None or getattr(c1, 'zero_lead', 0) == 0):
            if xs is None:
                xs = ops.ln_rowstats(x)
            return self.attn1.hip(x, B, N, c1, res=x, ln=(self.norm1, xs), stats_out=True)
        return self.attn1.hip(self.norm1.hip(x), B, N, c1, res=x)

    def pair_single_ok(self, x, B, N, context, xs):
        """can hip_rest(single=True) take ONE copy of a CFG pair (B = the doubled batch)?  The folded block, the zero-context
        shortcut on exactly the unconditional half, the wide-tile out-projection."""
        return xs is not None and context is not None and not isinstance(context, ContextMix) and \
            2 * getattr(context, 'zero_lead', 0) == B and \
            ops.wide_tile_ok(self.attn2.to_out[0].out_features, self.attn2.inner_dim)

    def hip_rest(self, x, B, N, context, xs=None, single=False):
        """cross-attention and feed-forward residual branches; xs: partial row sums of x when the block folds.
        single: x / xs are ONE copy of a CFG pair (see CrossAttention.hip); B is the doubled batch, the result is doubled"""
        z = getattr(context, 'zero_lead', 0) if context is not None else 0
        if xs is not None and context is not None and not isinstance(context, ContextMix):
            x, xs = self.attn2.hip(x, B, N, context, res=x, ln=(self.norm2, xs), stats_out=True, single=single)
            return self.ff.hip(x, res=x, ln=(self.norm3, xs))
        if single:
            raise ValueError("hip_rest(single=True) without the folded block (see pair_single_ok)")
        if 0 < z < B:   # LayerNorm only feeds to_q: skip it for the zero-context samples too
            xn = torch.empty_like(x)
            self.norm2.hip(x[z * N:], out=xn[z * N:])
            x = self.attn2.hip(xn, B, N, context, res=x)
        else:
            x = self.attn2.hip(self.norm2.hip(x), B, N, context, res=x)
        return self.ff.hip(self.norm3.hip(x), res=x)

    def hip(self, x, B, N, context, xs=None):
        y = self.hip_self(x, B, N, context, xs)
        if isinstance(y, tuple):
            return self.hip_rest(y[0], B, N, context, y[1])
        return self.hip_rest(y, B, N, context)

    def forward(self, x, context=None):
        B, N, _ = x.shape
        y = self.hip(x.to(torch.float16).reshape(B * N, -1).contiguous(), B, N, as_context_kv(context))
        return y.view(B, N, -1).to(x.dtype)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None,
                 disable_self_attn=False, use_linear=False, use_checkpoint=True):
        super().__init__()
        if exists(context_dim) and not isinstance(context_dim, (list, tuple)):
            context_dim = [context_dim]
        elif context_dim is None:
            context_dim = [None] * depth
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.use_linear = use_linear
        if use_linear:
            self.proj_in = L.Linear(in_channels, inner_dim)
        else:
            self.proj_in = L.Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                  disable_self_attn=disable_self_attn, checkpoint=use_checkpoint)
            for d in range(depth)])
        if use_linear:
            self.proj_out = L.Linear(in_channels, inner_dim)
        else:
            self.proj_out = L.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0)
        for p in self.proj_out.parameters():  # zero-initialised in the reference (attention.py:343-347)
            p.detach().zero_()

    # the CFG-pair doubling without torch.cat (PfdGemmDesc.res_rows); tests switch it per call
    pair_without_copies = True

    def hip(self, x, context=None, cfg_pair=False):
        """x: NHWC fp16 [B,H,W,C]; context: ContextKV | None.
        cfg_pair: x holds ONE copy of a classifier-free-guidance batch whose unconditional and conditional halves
        are identical up to here (same latent, same timestep: ddim.py:145-149 `torch.cat([x] * 2)`); everything
        before the first cross-attention -- GroupNorm, proj_in, self-attention -- is computed once and the batch
        is doubled right before the context enters.  Returns the full (doubled) batch."""
        B, H, W_, Cc = x.shape
        N = H * W_
        blocks = list(self.transformer_blocks)
        hs = None
        inner = self.proj_in.out_features if self.use_linear else self.proj_in.out_channels
        if ops.ln_fold_ok(inner) and inner % 160 == 0 and Cc % 64 == 0:   # (a 1x1 conv on Cc % 64 != 0 goes through im2col: no statistics)
            # proj_in stores the tokens the first LayerNorm reads: it emits their partial row sums with them
            h, hs = self.proj_in.hip(self.norm.hip(x), ln_out=True)
            h = h.view(B * N, -1)
        else:
            h = self.proj_in.hip(self.norm.hip(x)).view(B * N, -1)
        if cfg_pair:
            if blocks[0].disable_self_attn:
                raise ValueError("cfg_pair needs a context-free self-attention in the first block")
            h = blocks[0].hip_self(h, B, N, context, hs)
            hs = None
            if isinstance(h, tuple):
                h, hs = h
            x_rows = None
            if self.pair_without_copies and not self.use_linear and blocks[0].pair_single_ok(h, 2 * B, N, context, hs) and \
                    ops.wide_tile_ok(self.proj_out.out_channels, inner):
                # round 5: no copy at all -- the cross-attention out-projection and proj_out re-join the halves and read
                # their residual (the single copy of the tokens / of the block input) twice (PfdGemmDesc.res_rows)
                B = 2 * B
                h = blocks[0].hip_rest(h, B, N, context, hs, single=True)
                x_rows = (B // 2) * N
            else:
                if hs is not None:
                    hs = torch.cat([hs, hs])
                h, x, B = torch.cat([h, h]), torch.cat([x, x]), 2 * B      # three copies (21 MB each at 64^2 for h and x)
                h = blocks[0].hip_rest(h, B, N, context, hs)
            hs = None
            blocks = blocks[1:]
        for blk in blocks:
            h = blk.hip(h, B, N, context, hs)
            hs = None
        if self.use_linear:
            return self.proj_out.hip(h.view(B, H, W_, -1), res=x)
        if cfg_pair and x_rows is not None:
            return self.proj_out.hip(h.view(B, H, W_, -1), res=x.view(-1, Cc), gn_out=True, res_rows=x_rows)
        return self.proj_out.hip(h.view(B, H, W_, -1), res=x, gn_out=True)   # read next by a GroupNorm (ResBlock / head)

    def forward(self, x, context=None):
        y = self.hip(ops.to_nhwc(x), as_context_kv(context))
        return ops.to_nchw(y, x.dtype)
