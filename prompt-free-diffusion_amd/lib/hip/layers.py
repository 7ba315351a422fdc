"""Leaf layers of the HIP product path.

Each class subclasses the torch parameter holder of the same name so that `state_dict()` keys,
shapes and default initialisation are those of the reference checkpoints (1902
keys for pfd_with_control), but none of them ever runs a torch compute op: `.hip(...)` works
on fp16 NHWC / token-major tensors through libpfd_hip.so, and `.forward(...)` is the reference
calling convention (NCHW / any float dtype) wrapped around `.hip`.

Canonical parameters stay the `nn.Parameter`s; kernel-layout fp16 copies are cached per layer
and rebuilt whenever the canonical tensor changes identity, version, dtype or device
(`load_state_dict`, `.half()`, `.to()`, in-place ops on the parameter itself) -- app.py hot-swaps
weights per request (app.py:139-177, 217-222).  NOT detected: edits made through `param.data`
(`p.data.copy_(w)`, `p.data.mul_()` -- `.data` carries its own version counter, so neither the
pointer nor `p._version` moves).  Code that edits weights that way (EMA / LoRA merging) must call
`invalidate_packed(module)` afterwards; it also bumps a global generation that the hipGraph keys of
DDIMSampler / the pipeline stages include, so stale graphs are re-captured rather than replayed.
"""
import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU  # noqa: F401


def _round_up(v, m):
    return (v + m - 1) // m * m


_GENERATION = [0]


def generation():
    """bumped by invalidate_packed(); part of every packed-cache and hipGraph key"""
    return _GENERATION[0]


def invalidate_packed(module=None):
    """Drop the kernel-layout weight copies of `module` (and its children; None: nothing to walk, only the
    generation moves) after weights were edited through `.data` / raw pointers, and invalidate every
    captured hipGraph that may have baked the old copies in."""
    _GENERATION[0] += 1
    if module is not None:
        for m in module.modules():
            m.__dict__.pop("_pk_cache", None)


def _sig(*ps):
    return (_GENERATION[0],) + tuple((p.data_ptr(), p._version, p.dtype, p.device) if p is not None else None
                                     for p in ps)


def _dev16(t):
    if not t.is_cuda:
        raise RuntimeError("parameters must be on the GPU before the HIP path runs (call net.to('cuda')); "
                           "there is no CPU fallback")
    return t.detach().to(torch.float16)


def pack_matrix(w2d, kpad_to=64):
    """[N, K] -> dense fp16 [N, round_up(K, 64)] (zero padded K)."""
    w = _dev16(w2d)
    N, K = w.shape
    Kp = _round_up(K, kpad_to)
    if Kp != K:
        wp = torch.zeros((N, Kp), dtype=torch.float16, device=w.device)
        wp[:, :K] = w
        return wp
    return w.contiguous()


def pack_conv_weight(w4d):
    """[Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin] (tap-major, channel-minor), K padded to 64."""
    w = _dev16(w4d)
    Cout = w.shape[0]
    return pack_matrix(w.permute(0, 2, 3, 1).reshape(Cout, -1))


def pack_vec(b):
    return None if b is None else _dev16(b).contiguous()


def fold_layernorm(w, b, gamma, beta):
    """Operands of a Linear whose input LayerNorm is folded into the GEMM (PfdGemmDesc.ln_stats): w [N, K] fp16 in
    KERNEL row order (after any GEGLU interleave), b [N] fp16 or None in the same order, gamma / beta the LayerNorm
    affine [K].  -> (W o gamma as fp16, its fp32 row sums s_n, b' = beta . W^T + b as fp16).  Runs once per weight
    version (load time), like every other packing step."""
    w32 = w.float()
    g32, be32 = gamma.detach().float(), beta.detach().float()
    wg = (w32 * g32[None, :]).to(torch.float16).contiguous()
    cs = wg.float().sum(1).contiguous()
    bp = (w32 * be32[None, :]).sum(1)     # (elementwise + row sum, not `w32 @ be32`: no rocBLAS gemv inside lib/)
    if b is not None:
        bp = bp + b.float()
    return wg, cs, bp.to(torch.float16).contiguous()


class _Packed:
    """mixin: self._packed(name, builder, *params) -> cached kernel-layout tensors"""

    def _packed(self, name, builder, *params):
        cache = self.__dict__.setdefault("_pk_cache", {})
        sig = _sig(*params)
        ent = cache.get(name)
        if ent is None or ent[0] != sig:
            with torch.no_grad():
                ent = (sig, builder())
            cache[name] = ent
        return ent[1]


def _io_wrap_nchw(mod, x, fn):
    """reference calling convention for image layers: NCHW in -> NCHW out in x.dtype"""
    y = fn(ops.to_nhwc(x))
    return ops.to_nchw(y, x.dtype if x.dtype in (torch.float16, torch.float32) else torch.float32)


class Conv2d(nn.Conv2d, _Packed):
    """nn.Conv2d parameter holder; kxk (stride 1|2) as implicit GEMM, narrow Cin via im2col."""

    def _pk(self):
        return self._packed("w", lambda: (pack_conv_weight(self.weight), pack_vec(self.bias)), self.weight, self.bias)

    def hip_gn(self, x, norm, *, silu=True, keep_raw=False, rowvec=None, res=None, rows_per_rv=None):
        """this convolution followed by `norm` (a GroupNorm(32) over its output, + SiLU) with the normalisation done inside the
        convolution's split-K reduction (ops.conv gn_fuse, PfdGemmDesc.gnf_y): -> (raw | None, normalised), or None when the
        library does not serve the shape (nothing launched; the caller runs the two calls)"""
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        if self.in_channels % 64 or k != 3 or norm.num_groups != 32 or norm.num_channels != self.out_channels:
            return None
        w, b = self._pk()
        g, be = norm._pk()
        return ops.conv(x, w, k, stride=s, pad=p, bias=b, rowvec=rowvec, res=res, rows_per_rv=rows_per_rv,
                        gn_fuse=(g, be, norm.eps, silu, keep_raw))

    def hip(self, x, *, ups=False, rowvec=None, res=None, act=ACT_NONE, out=None, out_hw=None, rows_per_rv=None,
            gn=None, ln_out=None, gn_out=False, res_rows=None):
        """ln_out (1x1 convolutions only): also return the partial row sums of the output, see ops.gemm.
        res_rows (1x1 convolutions only): the residual is stored once for a doubled batch, see ops.gemm.
        gn_out=True: this output will be read by a GroupNorm -- where that norm takes the two-launch form the launch also
        emits its statistics (PfdGemmDesc.gn_out); they ride on the returned tensor (ops.get_gn_stats)"""
        w, b = self._pk()
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        cin = self.in_channels
        if gn is not None:     # GroupNorm prologue: x (| gn[1]) is the un-normalised input (ops.conv)
            return ops.conv(x, w, k, stride=s, pad=p, bias=b, rowvec=rowvec, res=res, act=act, out=out,
                            rows_per_rv=rows_per_rv, gn=gn)
        if cin % 64 == 0:
            if k == 1 and s == 1 and not ups:
                B, H, W_, _ = x.shape
                o2 = None if out is None else out.view(-1, out.shape[-1])
                r2 = None if res is None else res.reshape(-1, res.shape[-1])
                want = bool(gn_out) and (ln_out is None or ln_out is False) and act != ACT_GEGLU and \
                    ops.gn_stats_wanted(B, H * W_, self.out_channels) and ops.wide_tile_ok(self.out_channels, cin)
                y = ops.gemm(x.reshape(-1, cin), w, bias=b, rowvec=rowvec,
                             rows_per_rv=H * W_ if rows_per_rv is None else rows_per_rv, res=r2, act=act,
                             out=o2, ln_out=ln_out, gn_out=want, res_rows=res_rows)
                if ln_out is not None and ln_out is not False:
                    return y[0].view(B, H, W_, self.out_channels), y[1]
                v = y.view(B, H, W_, self.out_channels)
                if want:
                    ops.set_gn_stats(v, ops.get_gn_stats(y))
                return v
            if (ln_out is not None and ln_out is not False) or res_rows is not None:
                raise ValueError("ln_out / res_rows are for 1x1 convolutions (token-wise linears)")
            return ops.conv(x, w, k, stride=s, pad=p, ups=ups, bias=b, rowvec=rowvec, res=res, act=act, out=out,
                            out_hw=out_hw, rows_per_rv=rows_per_rv, gn_out=gn_out)
        if ups:
            raise NotImplementedError("narrow-channel conv with fused upsample")
        r2 = None if res is None else res.reshape(-1, res.shape[-1])
        ho, wo = out_hw if out_hw is not None else (None, None)
        return ops.conv_narrow(x, w, k, stride=s, pad=p, bias=b, rowvec=rowvec, res=r2, act=act, ho=ho, wo=wo,
                               out=out, gn_out=gn_out)

    def forward(self, x):
        return _io_wrap_nchw(self, x, self.hip)


class Linear(nn.Linear, _Packed):
    def _pk(self):
        return self._packed("w", lambda: (pack_matrix(self.weight), pack_vec(self.bias)), self.weight, self.bias)

    def _pk_ln(self, norm):
        """(W o gamma, column sums, b') for `norm` folded into this Linear (fold_layernorm).  A Linear is always folded
        with the same LayerNorm (the one in front of it), so the cache slot is just "ln"; the signature covers the norm's
        parameters (a swapped module or a reloaded weight rebuilds).  Built from the parameters directly: the plain pack
        of a folded Linear is never used on the default path and is not kept alive by this one."""
        return self._packed("ln", lambda: fold_layernorm(pack_matrix(self.weight), pack_vec(self.bias),
                                                         norm.weight, norm.bias),
                            self.weight, self.bias, norm.weight, norm.bias)

    def hip(self, x, *, act=ACT_NONE, res=None, rowvec=None, rows_per_rv=1, out=None, ln=None, ln_out=None):
        """x: [..., K] fp16 token-major -> [..., N].
        ln = (LayerNorm module, partial row sums of x): the LayerNorm in front of this Linear is folded into the GEMM
        (x is the UN-normalised tensor).  ln_out: see ops.gemm (returns (y, stats))."""
        if ln is not None:
            norm, st = ln
            w, cs, b = self._pk_ln(norm)
            return ops.gemm(x, w, bias=b, act=act, res=res, rowvec=rowvec, rows_per_rv=rows_per_rv, out=out,
                            ln=(st, cs, norm.eps), ln_out=ln_out)
        w, b = self._pk()
        K = self.in_features
        x2 = x.reshape(-1, K) if x.dim() != 2 else x
        if w.shape[1] != K:  # K was padded (e.g. PPE 80 -> 128): pad the activation once
            xp = torch.zeros((x2.shape[0], w.shape[1]), dtype=torch.float16, device=x2.device)
            xp[:, :K] = x2
            x2 = xp
        r2 = None if res is None else (res.reshape(-1, res.shape[-1]) if res.dim() != 2 else res)
        y = ops.gemm(x2, w, bias=b, act=act, res=r2, rowvec=rowvec, rows_per_rv=rows_per_rv, out=out, ln_out=ln_out)
        if ln_out is not None and ln_out is not False:
            return y
        return y if x.dim() == 2 else y.view(*x.shape[:-1], y.shape[-1])

    def hip_t(self, x2d, out=None):
        """transposed product: returns W @ x^T (+ bias per row) as [N, M] -- the V^T operand of
        pfd_attention_f16 written directly by the GEMM."""
        w, b = self._pk()
        return ops.gemm(w, x2d, bias=b, bias_per_row=True, out=out, k=self.in_features)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("HIP path: input must be on the GPU (no CPU fallback)")
        return self.hip(x.to(torch.float16)).to(x.dtype)


class GroupNorm(nn.GroupNorm, _Packed):
    def _pk(self):
        return self._packed("w", lambda: (pack_vec(self.weight), pack_vec(self.bias)), self.weight, self.bias)

    def fuse_key(self, silu):
        """identifies (this norm's parameters, eps, activation) for a result its input's producer computed ahead"""
        g, b = self._pk()
        return (g.data_ptr(), b.data_ptr(), float(self.eps), bool(silu))

    def hip(self, x, x2=None, silu=False):
        g, b = self._pk()
        if x2 is None:
            y = ops.get_normed(x, (g.data_ptr(), b.data_ptr(), float(self.eps), bool(silu)))
            if y is not None:        # the launch that wrote x already normalised it (Conv2d.hip_gn, PfdGemmDesc.gnf_y)
                return y
        return ops.groupnorm(x, g, b, self.num_groups, self.eps, x2=x2, silu=silu)

    def hip_table(self, x, x2=None):
        """statistics only, as the affine table of the consumer convolution's prologue (ops.groupnorm_table)"""
        g, b = self._pk()
        return ops.groupnorm_table(x, g, b, self.num_groups, self.eps, x2=x2)

    def forward(self, x):
        return _io_wrap_nchw(self, x, self.hip)


class LayerNorm(nn.LayerNorm, _Packed):
    def _pk(self):
        return self._packed("w", lambda: (pack_vec(self.weight), pack_vec(self.bias)), self.weight, self.bias)

    def hip(self, x, out=None):
        g, b = self._pk()
        return ops.layernorm(x, g, b, self.eps, out=out)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("HIP path: input must be on the GPU (no CPU fallback)")
        return self.hip(x.to(torch.float16).contiguous()).to(x.dtype)


class MultiheadAttention(nn.MultiheadAttention, _Packed):
    """nn.MultiheadAttention parameter holder (packed in_proj_weight [3C, C], seq-first API in
    the reference).  `.hip` takes batch-free token matrices: q_in [Nq, C], k_in [Nk, C],
    v_in [Nk, C] for ONE image and returns out_proj(attn) [Nq, C] (+ res)."""

    def _pk(self):
        def build():
            Cd = self.embed_dim
            w = _dev16(self.in_proj_weight)
            b = _dev16(self.in_proj_bias)
            return dict(wq=w[:Cd].contiguous(), wk=w[Cd:2 * Cd].contiguous(), wv=w[2 * Cd:].contiguous(),
                        bq=b[:Cd].contiguous(), bk=b[Cd:2 * Cd].contiguous(), bv=b[2 * Cd:].contiguous(),
                        wo=pack_matrix(self.out_proj.weight), bo=pack_vec(self.out_proj.bias))
        return self._packed("w", build, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight,
                            self.out_proj.bias)

    def hip(self, q_in, k_in, v_in, res=None):
        p = self._pk()
        Cd, H = self.embed_dim, self.num_heads
        D = Cd // H
        Nq, Nk = q_in.shape[0], k_in.shape[0]
        q = ops.gemm(q_in, p["wq"], bias=p["bq"])
        k = ops.gemm(k_in, p["wk"], bias=p["bk"])
        Nk8 = _round_up(Nk, 8)
        vt = torch.zeros((Cd, Nk8), dtype=torch.float16, device=q.device)   # pad columns must be finite (pfd_hip.h)
        ops.gemm(p["wv"], v_in, bias=p["bv"], bias_per_row=True, out=vt[:, :Nk])
        o = ops.attention(q, k, vt, 1, H, Nq, Nk, D, D ** -0.5, ldq=Cd, ldk=Cd, ldvt=Nk8, q_bs=0, k_bs=0, vt_bs=0)
        return ops.gemm(o, p["wo"], bias=p["bo"], res=res)

    def hip_seq1(self, x, res=None):
        """sequence length 1 (softmax over a single key == 1): out_proj(v_proj(x))."""
        p = self._pk()
        v = ops.gemm(x, p["wv"], bias=p["bv"])
        return ops.gemm(v, p["wo"], bias=p["bo"], res=res)


class Embedding(nn.Embedding, _Packed):
    def hip_weight(self):
        return self._packed("w", lambda: pack_vec(self.weight), self.weight)
