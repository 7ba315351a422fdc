"""SD-v1.5 UNet (split time_embed / data_blocks / context_blocks form) on the HIP path.

Mirrors the live part of the reference's lib/model_zoo/openaimodel.py: `UNetModel2D_Next`
(:2575-2812, registered as 'openai_unet_2d_next'), `ResBlock` (:162-274), `Downsample`
(:133-159), `Upsample` (:89-117), `TimestepEmbedSequential` (:72-86) -- same constructor
kwargs, same module tree and state-dict keys, same `i_order/m_order/o_order` lists that
`PromptFreeDiffusion.apply_model` walks (pfd.py:328-363).  The legacy UNets in that file
(:277-2570, :2814-2975) are not instantiated by any shipped config and are out of scope.

What differs is everything numerical: activations are NHWC fp16; GroupNorm+SiLU is one fused
kernel that reads the skip-connection concat virtually (two source pointers, no torch.cat,
pfd.py:356); convolutions are MFMA implicit GEMMs with bias, the per-sample time-embedding
vector (`h + emb_out`, :272), the residual (`skip_connection(x) + h`, :274) and nearest-2x
upsampling (:114) fused into their prologue/epilogue.
"""
import copy
import os
from functools import partial

import torch
import torch.nn as nn

from ..hip import layers as L
from ..hip import ops
from .attention import ContextMix, SpatialTransformer, as_context_kv
from .common.get_model import register

symbol = 'openai'


def normalization(channels):
    """GroupNorm32(32, C), eps 1e-5 (reference diffusion_utils.py:175-191)"""
    return L.GroupNorm(32, channels)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[N] int -> [N, dim] fp16 sinusoidal embedding (cos half first), fp32 math on device."""
    if repeat_only:
        return timesteps[:, None].to(torch.float16).repeat(1, dim)
    return ops.timestep_embedding(timesteps, dim, float(max_period))


class TimestepBlock(nn.Module):
    """marker: forward takes (x, emb)"""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Children get the time embedding / the context according to their kind."""

    def first_norm(self):
        """(GroupNorm, silu) that reads this layer's input alone -- a ResBlock's in_layers[0] (+ SiLU), a SpatialTransformer's
        norm -- or None (convolutions, the head)"""
        first = self[0] if len(self) else None
        if isinstance(first, ResBlock):
            return first.in_layers[0], True
        if isinstance(first, SpatialTransformer):
            return first.norm, False
        return None

    def hip(self, x, semb, context=None, x2=None, emb=None, cfg_pair=False, next_norm=None):
        """semb: SiLU(time embedding) [B, 4C] or None when `emb` = (table [rows, sumC], {id(block): col}, shared)
        already holds every ResBlock's emb_layers output (UNetModel2D_Next.emb_projections).
        next_norm: see ResBlock.hip (only a trailing ResBlock uses it)"""
        last = len(self) - 1
        for li, layer in enumerate(self):
            if isinstance(layer, ResBlock):
                x, x2 = layer.hip(x, semb, x2=x2, emb=emb, next_norm=next_norm if li == last else None), None
            elif isinstance(layer, SpatialTransformer):
                x = layer.hip(x, context, cfg_pair=cfg_pair)
            elif isinstance(layer, nn.Sequential):  # UNet head: GN -> SiLU -> conv
                x = layer[2].hip(layer[0].hip(x, silu=True))
            elif isinstance(layer, nn.SiLU):
                x = ops.activation(x, ops.ACT_SILU)
            elif isinstance(layer, L.Conv2d):       # stem convolution: its output is the first GroupNorm's input and a skip
                x = layer.hip(x, gn_out=True)
            else:
                x = layer.hip(x)
        return x

    def forward(self, x, emb, context=None):
        semb = ops.activation(emb.to(torch.float16).contiguous(), ops.ACT_SILU) if emb is not None else None
        y = self.hip(ops.to_nhwc(x), semb, as_context_kv(context))
        return ops.to_nchw(y, x.dtype)


class Upsample(nn.Module):
    """nearest 2x then (optionally) conv3x3 -- the gather is inside the conv's A-operand load"""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        if use_conv:
            self.conv = L.Conv2d(self.channels, self.out_channels, 3, padding=padding)

    def hip(self, x):
        assert x.shape[-1] == self.channels
        if not self.use_conv:
            raise NotImplementedError("Upsample without conv is not on the hot path")
        return self.conv.hip(x, ups=True, gn_out=True)     # read next by a ResBlock's GroupNorm (skip concat)

    def forward(self, x):
        return ops.to_nchw(self.hip(ops.to_nhwc(x)), x.dtype)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2 and use_conv, "only the strided-conv form is on the hot path"
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.op = L.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)

    def hip(self, x):
        assert x.shape[-1] == self.channels
        return self.op.hip(x, gn_out=True)                   # read next by a ResBlock's GroupNorm, and kept as a skip

    def forward(self, x):
        return ops.to_nchw(self.hip(ops.to_nhwc(x)), x.dtype)


class ResBlock(TimestepBlock):
    # GroupNorm -> SiLU inside the consumer convolution's input staging (PfdGemmDesc.gn_table) instead of a standalone
    # launch.  Bit-identical, and measured SLOWER on MI355X (profiles/r02_gn_prologue_ab.log: the affine map + SiLU on
    # the patch kernel's loader waves costs the convolution 20-26 us, the apply pass it removes 13-18 us; end to end
    # 6.58 vs 6.68 images/s), so it is off unless PFD_GN_PROLOGUE=1 (tests switch it per call).
    fuse_groupnorm = False
    # GroupNorm 2 (+ SiLU) inside the split-K reduction of the first convolution (PfdGemmDesc.gnf_y; tests switch it per call)
    fuse_reduce_groupnorm = True

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        assert dims == 2 and not use_scale_shift_norm and not up and not down, \
            "configuration outside the SD-v1.5 / ControlNet hot path"
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.updown = False
        self.in_layers = nn.Sequential(
            normalization(channels), nn.SiLU(), L.Conv2d(channels, self.out_channels, 3, padding=1))
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), L.Linear(emb_channels, self.out_channels))
        out_conv = L.Conv2d(self.out_channels, self.out_channels, 3, padding=1)
        for p in out_conv.parameters():  # zero_module in the reference (:228-230)
            p.detach().zero_()
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout), out_conv)
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = L.Conv2d(channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = L.Conv2d(channels, self.out_channels, 1)

    def hip(self, x, semb, x2=None, emb=None, next_norm=None):
        """x (and optional x2, the skip tensor of a virtual channel concat [x | x2]): NHWC fp16;
        semb: SiLU(time embedding) [B, emb_channels] fp16; emb: precomputed projections (see
        TimestepEmbedSequential.hip).  next_norm = (GroupNorm module, silu) of the layer that reads this block's output
        alone (not as half of a skip concat): where the last convolution splits K its reduction also writes that norm's
        result, which rides on the returned tensor (ops.set_normed) -- the consumer's GroupNorm launch disappears."""
        C1 = x.shape[-1]
        C2 = 0 if x2 is None else x2.shape[-1]
        assert C1 + C2 == self.channels
        B, H, W_, _ = x.shape
        # GroupNorm -> SiLU -> conv3x3 (:254-259, 268-270): where the patch kernel serves the convolution the
        # normalise + activate pass runs inside its input staging (the normalised tensor is never written)
        fuse = self.fuse_groupnorm
        fuse_in = fuse and ops.conv_gn_fusable(B, H, W_, C1, C2, self.out_channels)
        fuse_out = fuse and ops.conv_gn_fusable(B, H, W_, self.out_channels, 0, self.out_channels)
        rows_per_rv = None
        h_norm = None
        if emb is not None:
            table, cols, shared = emb
            c0 = cols[id(self)]
            e = table[:, c0:c0 + self.out_channels]                        # view, row stride sumC
            if shared:                                                     # one timestep for the whole batch
                rows_per_rv = 1 << 30
        else:
            e = self.emb_layers[1].hip(semb)                               # [B, Cout]
        if fuse_in:
            h = self.in_layers[2].hip(x, rowvec=e, rows_per_rv=rows_per_rv,
                                      gn=(self.in_layers[0].hip_table(x, x2), x2, True))
        else:
            hn = self.in_layers[0].hip(x, x2, silu=True)                   # [B,H,W,C1+C2]
            # conv + bias + emb; its output is read by GroupNorm 2 only.  Where the convolution splits K (the 8^2 / 16^2
            # levels) GroupNorm 2 + SiLU happen inside its reduction launch and the raw tensor is never written (round 5,
            # PfdGemmDesc.gnf_y); elsewhere the statistics come with the store (ops.gemm gn_out)
            fused = None if fuse_out or not self.fuse_reduce_groupnorm else \
                self.in_layers[2].hip_gn(hn, self.out_layers[0], silu=True, rowvec=e, rows_per_rv=rows_per_rv)
            if fused is not None:
                h, h_norm = None, fused[1]
            else:
                h = self.in_layers[2].hip(hn, rowvec=e, rows_per_rv=rows_per_rv, gn_out=not fuse_out)
        skip = self.skip_connection
        if isinstance(skip, nn.Identity):
            assert x2 is None
            sk = x
        elif skip.kernel_size[0] == 1:
            w, b = skip._pk()
            B, H, W_, _ = x.shape
            if x2 is not None and ops.wide_tile_ok(w.shape[0], C1 + C2) and C1 % 64 == 0:
                # `skip_connection(torch.cat([h, skip], 1))` (:274 after pfd.py:356) as ONE contraction over two sources
                sk = ops.gemm(x.view(-1, C1), w, bias=b, a2=x2.view(-1, C2), k=C1 + C2)
            else:
                sk = ops.gemm(x.view(-1, C1), w[:, :C1], bias=b, k=C1)
                if x2 is not None:
                    sk = ops.gemm(x2.view(-1, C2), w[:, C1:], res=sk, k=C2, out=sk)
            sk = sk.view(B, H, W_, -1)
        else:
            assert x2 is None
            sk = skip.hip(x)
        if fuse_out:
            return self.out_layers[3].hip(h, res=sk, gn=(self.out_layers[0].hip_table(h), None, True))
        # (the block's output feeds the next GroupNorm -- a ResBlock's, a SpatialTransformer's, the head's -- possibly
        #  later, as a skip: statistics with the store)
        if h_norm is None:
            h_norm = self.out_layers[0].hip(h, silu=True)
        if next_norm is not None and self.fuse_reduce_groupnorm:
            norm, nsilu = next_norm
            fused = self.out_layers[3].hip_gn(h_norm, norm, silu=nsilu, keep_raw=True, res=sk)
            if fused is not None:
                return ops.set_normed(fused[0], norm.fuse_key(nsilu), fused[1])
        return self.out_layers[3].hip(h_norm, res=sk, gn_out=True)

    def forward(self, x, emb):
        semb = ops.activation(emb.to(torch.float16).contiguous(), ops.ACT_SILU)
        return ops.to_nchw(self.hip(ops.to_nhwc(x), semb), x.dtype)


@register('openai_unet_2d_next')
class UNetModel2D_Next(nn.Module, L._Packed):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 context_dim, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, use_checkpoint=False,
                 num_heads=8, num_head_channels=None, parts=['global', 'data', 'context']):
        super().__init__()
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        if isinstance(num_res_blocks, int):
            num_res_blocks = len(channel_mult) * [num_res_blocks]
        elif len(num_res_blocks) != len(channel_mult):
            raise ValueError("provide num_res_blocks either as an int (globally constant) or "
                             "as a list/tuple (per-level) with the same length as channel_mult")
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.context_dim = context_dim
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.use_checkpoint = use_checkpoint
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        assert (num_heads is None) + (num_head_channels is None) == 1, \
            "One of num_heads or num_head_channels need to be set"
        self.parts = parts if isinstance(parts, list) else [parts]
        self.glayer_included = 'global' in self.parts
        self.dlayer_included = 'data' in self.parts
        self.clayer_included = 'context' in self.parts

        time_embed_dim = model_channels * 4
        if self.glayer_included:
            self.time_embed = nn.Sequential(
                L.Linear(model_channels, time_embed_dim), nn.SiLU(), L.Linear(time_embed_dim, time_embed_dim))
        if self.dlayer_included:
            self.data_blocks = nn.ModuleList()
        if self.clayer_included:
            self.context_blocks = nn.ModuleList()

        def res(cin, cout):
            if not self.dlayer_included:
                return None
            return ResBlock(cin, time_embed_dim, dropout, out_channels=cout, dims=2,
                            use_checkpoint=use_checkpoint, use_scale_shift_norm=False)

        def xattn(ch):
            if not self.clayer_included:
                return None
            d_head, n_heads = self.get_d_head_n_heads(ch)
            return SpatialTransformer(ch, n_heads, d_head, context_dim=context_dim, disable_self_attn=False)

        order = []
        self._order = order

        # ---- input half ----
        self.add_data_layer(L.Conv2d(in_channels, model_channels, 3, padding=1) if self.dlayer_included else None)
        order.append('save_hidden_feature')
        skip_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks[level]):
                self.add_data_layer(res(ch, mult * model_channels))
                ch = mult * model_channels
                if ds in attention_resolutions:
                    self.add_context_layer(xattn(ch))
                skip_chans.append(ch)
                order.append('save_hidden_feature')
            if level != len(channel_mult) - 1:
                self.add_data_layer(Downsample(ch, True, dims=2, out_channels=ch) if self.dlayer_included else None)
                skip_chans.append(ch)
                order.append('save_hidden_feature')
                ds *= 2
        self.i_order = list(order)
        order.clear()

        # ---- middle ----
        self.add_data_layer(res(ch, ch))
        self.add_context_layer(xattn(ch))
        self.add_data_layer(res(ch, ch))
        self.m_order = list(order)
        order.clear()

        # ---- output half ----
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for _ in range(num_res_blocks[level] + 1):
                order.append('load_hidden_feature')
                self.add_data_layer(res(ch + skip_chans.pop(), model_channels * mult))
                ch = model_channels * mult
                if ds in attention_resolutions:
                    self.add_context_layer(xattn(ch))
            if level != 0:
                self.add_data_layer(Upsample(ch, conv_resample, dims=2, out_channels=ch)
                                    if self.dlayer_included else None)
                ds //= 2
        if self.dlayer_included:
            head_conv = L.Conv2d(model_channels, out_channels, 3, padding=1)
            for p in head_conv.parameters():  # zero_module in the reference (:2735)
                p.detach().zero_()
            head = nn.Sequential(normalization(ch), nn.SiLU(), head_conv)
        else:
            head = None
        self.add_data_layer(head)
        self.o_order = list(order)
        self.layer_order = copy.deepcopy(self.i_order + self.m_order + self.o_order)
        del self._order

        self.parameter_group = {}
        if self.glayer_included:
            self.parameter_group['global'] = self.time_embed
        if self.dlayer_included:
            self.parameter_group['data'] = self.data_blocks
        if self.clayer_included:
            self.parameter_group['context'] = self.context_blocks

    def get_d_head_n_heads(self, ch):
        if self.num_head_channels is None:
            return ch // self.num_heads, self.num_heads
        return self.num_head_channels, ch // self.num_head_channels

    def add_data_layer(self, layer):
        if self.dlayer_included:
            layers = layer if isinstance(layer, (list, tuple)) else [layer]
            self.data_blocks.append(TimestepEmbedSequential(*layers))
        self._order.append('d')

    def add_context_layer(self, layer):
        if self.clayer_included:
            layers = layer if isinstance(layer, (list, tuple)) else [layer]
            self.context_blocks.append(TimestepEmbedSequential(*layers))
        self._order.append('c')

    # -------------------------------------------------------------------------------------------
    def silu_time_embedding(self, timesteps):
        """SiLU(time_embed(timestep_embedding(t))) [B, 4*model_channels] fp16: every consumer of the
        embedding (the 22 ResBlock emb_layers) applies SiLU first (:217), so only this is kept."""
        t_emb = timestep_embedding(timesteps, self.model_channels)
        h = self.time_embed[0].hip(t_emb, act=ops.ACT_SILU)
        return self.time_embed[2].hip(h, act=ops.ACT_SILU)

    def _emb_pack(self):
        """all 22 ResBlock `emb_layers` Linears as ONE packed weight [sum Cout, 4C] (+ bias, column map)"""
        blocks = [m for seq in self.data_blocks for m in seq if isinstance(m, ResBlock)]
        params = [b.emb_layers[1].weight for b in blocks] + [b.emb_layers[1].bias for b in blocks]

        def build():
            w = torch.cat([L._dev16(b.emb_layers[1].weight) for b in blocks], 0).contiguous()
            bias = torch.cat([L._dev16(b.emb_layers[1].bias) for b in blocks], 0).contiguous()
            cols, off = {}, 0
            for b in blocks:
                cols[id(b)] = off
                off += b.out_channels
            return w, bias, cols
        return self._packed("emb_cat", build, *params)

    def emb_projections(self, timesteps):
        """[len(timesteps), sum Cout] = every ResBlock's `emb_layers(emb)` for the given timesteps in
        one GEMM (the reference runs 22 SiLU+Linear pairs per forward, openaimodel.py:217-223, :262)"""
        w, bias, cols = self._emb_pack()
        return ops.gemm(self.silu_time_embedding(timesteps), w, bias=bias), cols

    def hip(self, x, timesteps, context, control=None, context_net=None, emb_table=None, cfg_pair=False):
        """The forward that pfd.apply_model defines (pfd.py:314-365, :466-528), NHWC fp16 in/out.
        control: list of 13 NHWC residuals from ControlNet (popped from the end) or None.
        context_net: the UNet that owns the context blocks (defaults to self).
        emb_table: optional [1, sum Cout] row of `emb_projections` valid for EVERY sample of the
        batch (the sampler precomputes all steps at once); otherwise computed here per sample.
        cfg_pair: x is ONE copy [B/2, ...] of a classifier-free-guidance batch [x | x] with a shared timestep
        (what ddim.py:145-149 builds with torch.cat([x] * 2)): the layers in front of the first cross-attention
        (stem conv, first ResBlock, first transformer's GroupNorm / proj_in / self-attention) give identical
        results for both halves and run once; returns eps for the full batch [uncond | cond].  Exact."""
        cnet = self if context_net is None else context_net
        if emb_table is not None:
            emb = (emb_table, self._emb_pack()[2], True)
        else:
            table, cols = self.emb_projections(timesteps)     # one row per sample of the FULL batch
            emb = (table, cols, False)
        semb = None
        d_iter = iter(self.data_blocks)
        pair = [bool(cfg_pair)]          # still running on one copy of the pair
        if isinstance(context, ContextMix):      # multi-context: every context layer mixes n transformers
            if cfg_pair:
                raise NotImplementedError("cfg_pair with multi-context mixing")
            c_iters = [iter(n.context_blocks) for n in context.nets]
            ctx_layer = lambda hh: context.mix([next(it) for it in c_iters], hh, semb)  # noqa: E731
        else:
            c_iter = iter(cnet.context_blocks)

            def ctx_layer(hh):
                p, pair[0] = pair[0], False
                return next(c_iter).hip(hh, semb, context, cfg_pair=p)
        ccs = list(control) if control is not None else None
        hs = []
        h = x
        nn_of = self._next_norms(cnet, control is not None) if not isinstance(context, ContextMix) else {}
        step = [0]

        def data_layer(hh, x2=None):
            return next(d_iter).hip(hh, semb, x2=x2, emb=emb, next_norm=nn_of.get(step[0]))
        for ltype in self.i_order:
            if ltype == 'd':
                h = data_layer(h)
            elif ltype == 'c':
                h = ctx_layer(h)
            else:
                hs.append(ops.cat_pair(h) if pair[0] else h)   # a skip saved before the doubling: stored doubled
            step[0] += 1
        if pair[0]:
            raise ValueError("cfg_pair: no context layer in the input half of this UNet")
        for ltype in self.m_order:
            h = data_layer(h) if ltype == 'd' else ctx_layer(h)
            step[0] += 1
        if ccs is not None:
            h = ops.add(h, ccs.pop())
        skip = None
        for ltype in self.o_order:
            if ltype == 'load_hidden_feature':
                skip = hs.pop()
                if ccs is not None:
                    skip = ops.add(skip, ccs.pop())
            elif ltype == 'd':
                h, skip = data_layer(h, x2=skip), None
            else:
                h = ctx_layer(h)
            step[0] += 1
        return h

    def _next_norms(self, cnet, with_control):
        """{position in i_order + m_order + o_order of a data layer: (GroupNorm, silu)} -- the norm that reads that layer's
        output ALONE: the next layer is a context layer (SpatialTransformer.norm) or a data layer that starts with a ResBlock
        and takes no skip concat (in_layers[0] + SiLU).  With ControlNet residuals the middle block's output is modified
        before its consumer reads it: no hint there.  Cached per (context net, control) on the module."""
        #  keyed by a WEAK reference to the context net plus what the table depends on (block counts, orders): a
        # replaced / freed context net cannot alias an old entry through a reused id, and its modules are not kept alive here
        import weakref
        cache = self.__dict__.setdefault("_nn_cache", {})
        key = (bool(with_control), len(cnet.context_blocks), len(self.data_blocks), tuple(self.i_order), tuple(self.m_order),
               tuple(self.o_order))
        ent = cache.get(key)
        if ent is not None and ent[0]() is cnet:
            return ent[1]
        order = list(self.i_order) + list(self.m_order) + list(self.o_order)
        d_list, c_list = list(self.data_blocks), list(cnet.context_blocks)
        di = ci = 0
        seq = []                                   # (kind, module | None) per position
        for lt in order:
            if lt == 'd':
                seq.append(('d', d_list[di]))
                di += 1
            elif lt == 'c':
                seq.append(('c', c_list[ci]))
                ci += 1
            else:
                seq.append((lt, None))
        n_in, n_mid = len(self.i_order), len(self.m_order)
        out = {}
        for i, (kind, mod) in enumerate(seq):
            if kind != 'd' or not isinstance(mod[len(mod) - 1], ResBlock):
                continue
            j = i + 1
            while j < len(seq) and seq[j][0] not in ('d', 'c', 'load_hidden_feature'):
                j += 1                             # ('save_hidden_feature' keeps the tensor: the raw result is stored anyway)
            if j >= len(seq) or seq[j][0] == 'load_hidden_feature':
                continue                           # consumed as half of a skip concat (or not at all)
            if with_control and i == n_in + n_mid - 1:
                continue                           # `h = h + control.pop()` sits between (pfd.py:515)
            fn = seq[j][1].first_norm()
            if fn is not None:
                out[i] = fn
        cache[key] = (weakref.ref(cnet), out)
        return out

    def forward(self, x, timesteps, context):
        y = self.hip(ops.to_nhwc(x), timesteps, as_context_kv(context))
        return ops.to_nchw(y, x.dtype)
