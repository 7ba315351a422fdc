"""ControlNet ('controlnet') on the HIP path: hint encoder + a copy of the UNet's encoder half +
13 zero-convs, returning the 13 residuals pfd_with_control.apply_model injects (pfd.py:472-519).

Module tree / state-dict keys / constructor kwargs follow the network part of the reference's
lib/model_zoo/controlnet.py (`ControlNet` :65-297, `forward` :302-324, hint block :165-181,
`make_zero_conv` :299-300).  `preprocess` and the vendored annotators (:332-503,
controlnet_annotator/**) are CPU-side third-party image preprocessors, off by default in the
app (`do_preprocess=False`) and out of scope (SURVEY §2 #11b).

The hint encoder (8 convs) depends on neither the step nor the sample, so `prepare_hint` runs
it once per request; the reference recomputes it in every one of the 50 steps (:314).
"""
import torch
import torch.nn as nn

from ..hip import layers as L
from ..hip import ops
from .attention import SpatialTransformer, as_context_kv
from .common.get_model import register
from .openaimodel import Downsample, ResBlock, TimestepEmbedSequential, timestep_embedding

symbol = 'controlnet'


class PreparedHint:
    """output of the hint encoder for one request: NHWC fp16 [1|B, h, w, model_channels]"""

    def __init__(self, feat):
        self.feat = feat


def _zero(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


@register('controlnet')
class ControlNet(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, hint_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False):
        super().__init__()
        assert use_spatial_transformer and context_dim is not None, "only the cross-attention form is on the path"
        assert dims == 2 and not resblock_updown and not use_scale_shift_norm and n_embed is None
        assert disable_self_attentions is None and num_attention_blocks is None
        if isinstance(context_dim, (list, tuple)) or type(context_dim).__name__ == 'ListConfig':
            context_dim = list(context_dim)
        if num_heads == -1:
            assert num_head_channels != -1, 'Either num_heads or num_head_channels has to be set'
        if num_head_channels == -1:
            assert num_heads != -1, 'Either num_heads or num_head_channels has to be set'
        self.dims = dims
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        if isinstance(num_res_blocks, int):
            num_res_blocks = len(channel_mult) * [num_res_blocks]
        elif len(num_res_blocks) != len(channel_mult):
            raise ValueError("provide num_res_blocks either as an int (globally constant) or "
                             "as a list/tuple (per-level) with the same length as channel_mult")
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads if num_heads_upsample == -1 else num_heads_upsample
        self.predict_codebook_ids = False

        def heads_for(ch):
            if num_head_channels == -1:
                nh, dh = num_heads, ch // num_heads
            else:
                nh, dh = ch // num_head_channels, num_head_channels
            if legacy:
                dh = ch // nh
            return nh, dh

        def xattn(ch, no_self=False):
            nh, dh = heads_for(ch)
            return SpatialTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim,
                                      disable_self_attn=no_self, use_linear=use_linear_in_transformer,
                                      use_checkpoint=use_checkpoint)

        def res(cin, cout):
            return ResBlock(cin, time_embed_dim, dropout, out_channels=cout, dims=dims,
                            use_checkpoint=use_checkpoint, use_scale_shift_norm=False)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(
            L.Linear(model_channels, time_embed_dim), nn.SiLU(), L.Linear(time_embed_dim, time_embed_dim))
        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(L.Conv2d(in_channels, model_channels, 3, padding=1))])
        self.zero_convs = nn.ModuleList([self.make_zero_conv(model_channels)])
        hint_layers = []
        chans = [hint_channels, 16, 16, 32, 32, 96, 96, 256]
        strides = [1, 1, 2, 1, 2, 1, 2]
        for cin, cout, s in zip(chans[:-1], chans[1:], strides):
            hint_layers += [L.Conv2d(cin, cout, 3, padding=1, stride=s), nn.SiLU()]
        hint_layers.append(_zero(L.Conv2d(256, model_channels, 3, padding=1)))
        self.input_hint_block = TimestepEmbedSequential(*hint_layers)

        self._feature_size = model_channels
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks[level]):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(xattn(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self.zero_convs.append(self.make_zero_conv(ch))
                self._feature_size += ch
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims,
                                                                            out_channels=ch)))
                self.zero_convs.append(self.make_zero_conv(ch))
                ds *= 2
                self._feature_size += ch
        self.middle_block = TimestepEmbedSequential(res(ch, ch), xattn(ch, disable_middle_self_attn), res(ch, ch))
        self.middle_block_out = self.make_zero_conv(ch)
        self._feature_size += ch

    def make_zero_conv(self, channels):
        return TimestepEmbedSequential(_zero(L.Conv2d(channels, channels, 1, padding=0)))

    # -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def prepare_hint(self, hint):
        """hint: NCHW image in [0,1] ([1,3,H,W], broadcast over the batch) or a PreparedHint"""
        if hint is None or isinstance(hint, PreparedHint):
            return hint
        h = ops.to_nhwc(hint)
        layers = list(self.input_hint_block)
        for i, layer in enumerate(layers):
            if isinstance(layer, L.Conv2d):
                fuse_silu = i + 1 < len(layers) and isinstance(layers[i + 1], nn.SiLU)
                h = layer.hip(h, act=ops.ACT_SILU if fuse_silu else ops.ACT_NONE)
        return PreparedHint(h)

    def hip(self, x, hint, timesteps, context, cfg_pair=False):
        """x NHWC fp16 [B,h,w,4]; hint NCHW tensor | PreparedHint; context ContextKV.
        Returns the 13 residuals (NHWC fp16), to be popped from the end.
        cfg_pair: x is ONE copy [B/2,...] of a classifier-free-guidance batch [x | x] with a shared timestep: the
        stem conv (+ hint), the first ResBlock and the first transformer's self-attention part run once
        (UNetModel2D_Next.hip); `timesteps` and the returned residuals are for the full batch."""
        guided = self.prepare_hint(hint).feat
        t_emb = timestep_embedding(timesteps, self.model_channels)
        semb = self.time_embed[2].hip(self.time_embed[0].hip(t_emb, act=ops.ACT_SILU), act=ops.ACT_SILU)
        B = x.shape[0]
        outs = []
        h = x
        pair = bool(cfg_pair)
        for i, (module, zero_conv) in enumerate(zip(self.input_blocks, self.zero_convs)):
            if i == 0:
                conv = module[0]
                if guided.shape[0] == B:
                    h = conv.hip(h, res=guided)                       # h = conv(x) + guided_hint
                elif guided.shape[0] == 1:                            # one hint for the whole batch
                    h = conv.hip(h)
                    for b in range(B):
                        ops.add(h[b:b + 1], guided, out=h[b:b + 1])
                elif B % guided.shape[0] == 0:                        # per-sample hints under CFG: [u | c] halves
                    h = conv.hip(h)
                    k = guided.shape[0]
                    for b in range(0, B, k):
                        ops.add(h[b:b + k], guided, out=h[b:b + k])
                else:   # the reference's `h + guided_hint` raises a broadcast error here (controlnet.py:315)
                    raise ValueError(f"control hint batch {guided.shape[0]} does not broadcast to batch {B}")
                o = zero_conv[0].hip(h)
                outs.append(torch.cat([o, o]) if pair else o)
                continue
            if pair:
                if not any(isinstance(m, SpatialTransformer) for m in module):
                    raise ValueError("cfg_pair: the block after the stem has no transformer to double the batch in")
                h = module.hip(h, semb, context, cfg_pair=True)       # leaves with the full batch
                pair = False
            else:
                h = module.hip(h, semb, context)
            outs.append(zero_conv[0].hip(h))
        h = self.middle_block.hip(h, semb, context)
        outs.append(self.middle_block_out[0].hip(h))
        return outs

    def forward(self, x, hint, timesteps, context, **kwargs):
        outs = self.hip(ops.to_nhwc(x), hint, timesteps, as_context_kv(context))
        return [ops.to_nchw(o, x.dtype) for o in outs]

    def preprocess(self, *args, **kwargs):
        raise NotImplementedError(
            "control-hint preprocessors (canny/HED/MiDaS/...) are vendored CPU annotators outside the "
            "denoising hot path; pass an already prepared control image (app default do_preprocess=False)")

    def get_device(self):
        return self.time_embed[0].weight.device

    def get_dtype(self):
        return self.time_embed[0].weight.dtype
