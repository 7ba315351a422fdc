"""SeeCoder semantic-context encoder ('seecoder', 'seecoder_decoder', 'seecoder_query_transformer')
on the HIP path: Swin-L features -> 6-layer multi-scale token decoder (+ laterals) -> 9-layer
query transformer (4 global + 144 local queries) -> context [B, 148, 768].

Module tree / state-dict keys / constructor kwargs follow the reference's
lib/model_zoo/seecoder.py: `Decoder` (:328-428), `DecoderLayer` (:60-90), `QueryTransformer`
(:434-550), `SelfAttentionLayer` (:107-155), `CrossAttentionLayer` (:157-210),
`FeedForwardLayer` (:212-246), `PPE_MLP` (:262-310), `SemanticContextEncoder` (:556-578).

Reference behaviour that is kept on purpose:
  * `DecoderLayer.self_attn` is an `nn.MultiheadAttention` WITHOUT batch_first that is fed
    [B, L, C] (:70, :83): it attends across the batch axis.  For one image that is a softmax over
    a single key, i.e. `out_proj(v_proj(x))`; the q/k projections never influence the result.
    app.py encodes one image at a time (app.py:234-235) and so does this module: B > 1 is
    rejected instead of silently mixing images.
  * q/k get positional terms added before projection, v does not (:131-133, :183-186).
  * PPE_MLP positional features are a function of (h, w) only; they are evaluated on the host in
    fp32 (the reference evaluates them in x.dtype, :301-303) and run through the MLP on the GPU.
"""
import math

import torch
import torch.nn as nn

from ..hip import layers as L
from ..hip import ops
from .common.get_model import get_model, register

symbol = 'seecoder'


def _clones(make, n):
    return nn.ModuleList([make() for _ in range(n)])


class Conv2d_Convenience(L.Conv2d):
    """1x1 conv (no bias) followed by an optional norm, used for the decoder laterals"""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation
        assert activation is None

    def hip(self, x, **kw):
        y = super().hip(x, **kw)
        return self.norm.hip(y) if self.norm is not None else y


class DecoderLayer(nn.Module):
    def __init__(self, dim=256, feedforward_dim=1024, dropout=0.1, activation="relu", n_heads=8):
        super().__init__()
        assert activation == "relu"
        self.self_attn = L.MultiheadAttention(dim, n_heads, dropout=dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = L.LayerNorm(dim)
        self.linear1 = L.Linear(dim, feedforward_dim)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = L.Linear(feedforward_dim, dim)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = L.LayerNorm(dim)

    def hip(self, x):
        """x: [L, C] tokens of ONE image (sequence length 1 in the reference's seq-first MHA)"""
        h = self.norm1.hip(self.self_attn.hip_seq1(x, res=x))
        h2 = self.linear2.hip(self.linear1.hip(h, act=ops.ACT_RELU), res=h)
        return self.norm2.hip(h2)


class DecoderLayerStacked(nn.Module):
    def __init__(self, make_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _clones(make_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def hip(self, x):
        for layer in self.layers:
            x = layer.hip(x)
        return self.norm.hip(x) if self.norm is not None else x


class SelfAttentionLayer(nn.Module):
    def __init__(self, channels, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before, "pre-norm is deprecated in the reference (assert False, :137)"
        self.self_attn = L.MultiheadAttention(channels, nhead, dropout=dropout)
        self.norm = L.LayerNorm(channels)
        self.dropout = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def hip(self, qkv, qk_pos=None, out=None):
        qk = qkv if qk_pos is None else ops.add(qkv, qk_pos)
        return self.norm.hip(self.self_attn.hip(qk, qk, qkv, res=qkv), out=out)


class CrossAttentionLayer(nn.Module):
    def __init__(self, channels, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before
        self.multihead_attn = L.MultiheadAttention(channels, nhead, dropout=dropout)
        self.norm = L.LayerNorm(channels)
        self.dropout = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def hip(self, q, kv, q_pos=None, k_pos=None, out=None):
        q_in = q if q_pos is None else ops.add(q, q_pos)
        k_in = kv if k_pos is None else ops.add(kv, k_pos)
        return self.norm.hip(self.multihead_attn.hip(q_in, k_in, kv, res=q), out=out)


class FeedForwardLayer(nn.Module):
    def __init__(self, channels, hidden_channels=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert activation == "relu" and not normalize_before
        self.linear1 = L.Linear(channels, hidden_channels)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = L.Linear(hidden_channels, channels)
        self.norm = L.LayerNorm(channels)
        self.normalize_before = normalize_before

    def hip(self, x, out=None):
        return self.norm.hip(self.linear2.hip(self.linear1.hip(x, act=ops.ACT_RELU), res=x), out=out)


class PPE_MLP(nn.Module):
    """position-aware encoding of SeeCoder-PA: sin/cos of centred pixel coordinates at
    `freq_num` geometric frequencies -> 3-layer SiLU MLP.  app.py attaches an instance to
    `qtransformer.pe_layer` at run time (app.py:166-177)."""

    def __init__(self, freq_num=20, freq_max=None, out_channel=768, mlp_layer=3):
        super().__init__()
        self.freq_num = freq_num
        self.freq_max = freq_max
        self.out_channel = out_channel
        self.mlp_layer = mlp_layer
        self.twopi = 2 * math.pi
        mlp = []
        in_channel = freq_num * 4
        for idx in range(mlp_layer):
            linear = L.Linear(in_channel, out_channel, bias=True)
            nn.init.xavier_normal_(linear.weight)
            nn.init.constant_(linear.bias, 0)
            mlp.append(linear)
            if idx != mlp_layer - 1:
                mlp.append(nn.SiLU())
            in_channel = out_channel
        self.mlp = nn.Sequential(*mlp)
        nn.init.constant_(self.mlp[-1].weight, 0)

    def features(self, h, w, device):
        """[h*w, 4*freq_num] fp16 sinusoid features: a table that depends on the SHAPE only, built once per
        (h, w, device) with host fp32 math (like the DDIM schedule tables) and kept on the device -- so a
        hipGraph capture of the context stage never sees a pageable host->device copy"""
        cache = self.__dict__.setdefault("_feat_cache", {})
        key = (h, w, str(device), self.freq_num, self.freq_max)
        hit = cache.get(key)
        if hit is None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("PPE_MLP.features: table for a new shape requested during stream capture")
            if len(cache) >= 16:
                cache.clear()
            hit = cache[key] = self._features_host(h, w).to(device=device, dtype=torch.float16)
        return hit

    def _features_host(self, h, w):
        minlen = min(h, w)
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                                indexing='ij')
        ys = (ys + 0.5 - h / 2) / minlen * self.twopi
        xs = (xs + 0.5 - w / 2) / minlen * self.twopi
        freq_max = self.freq_max if self.freq_max is not None else minlen / 2
        dim_t = float(freq_max) ** torch.linspace(0, 1, self.freq_num, dtype=torch.float32)
        ph, pw = ys[:, :, None] * dim_t, xs[:, :, None] * dim_t
        pos = torch.cat((ph.sin(), ph.cos(), pw.sin(), pw.cos()), dim=-1)
        return pos.reshape(h * w, -1)

    def hip(self, h, w, device):
        x = self.features(h, w, device)
        n = len(self.mlp)
        for i, layer in enumerate(self.mlp):
            if isinstance(layer, L.Linear):
                x = layer.hip(x, act=ops.ACT_SILU if i != n - 1 else ops.ACT_NONE)
        return x  # [h*w, out_channel]

    def forward(self, x, mask=None):
        assert mask is None, "Mask not implemented"
        h, w = x.shape[-2:]
        pos = self.hip(h, w, x.device).view(1, h, w, self.out_channel)
        return ops.to_nchw(pos.contiguous(), x.dtype)


@register('seecoder_decoder')
class Decoder(nn.Module):
    def __init__(self, inchannels, trans_input_tags, trans_num_layers, trans_dim, trans_nheads, trans_dropout,
                 trans_feedforward_dim):
        super().__init__()
        trans_in = {k: v for k, v in inchannels.items() if k in trans_input_tags}
        fpn_in = {k: v for k, v in inchannels.items() if k not in trans_input_tags}
        self.trans_tags = sorted(trans_in.keys())
        self.fpn_tags = sorted(fpn_in.keys())
        self.all_tags = sorted(inchannels.keys())
        assert len(self.trans_tags) > 0
        if self.fpn_tags:
            raise NotImplementedError("FPN-only levels are not used by the shipped SeeCoder config")
        self.num_trans_lvls = len(self.trans_tags)
        self.trans_dim = trans_dim

        self.inproj_layers = nn.ModuleDict()
        for tag in self.trans_tags:
            layer = nn.Sequential(L.Conv2d(trans_in[tag], trans_dim, kernel_size=1), L.GroupNorm(32, trans_dim))
            nn.init.xavier_uniform_(layer[0].weight, gain=1)
            nn.init.constant_(layer[0].bias, 0)
            self.inproj_layers[tag] = layer
        self.transformer = DecoderLayerStacked(
            lambda: DecoderLayer(dim=trans_dim, n_heads=trans_nheads, dropout=trans_dropout,
                                 feedforward_dim=trans_feedforward_dim, activation='relu'), trans_num_layers)
        for p in self.transformer.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.level_embed = nn.Parameter(torch.Tensor(len(self.trans_tags), trans_dim))
        nn.init.normal_(self.level_embed)
        self.lateral_layers = nn.ModuleDict()
        self.output_layers = nn.ModuleDict()
        for tag in self.all_tags:
            lat = Conv2d_Convenience(inchannels[tag], trans_dim, kernel_size=1, bias=False,
                                     norm=L.GroupNorm(32, trans_dim))
            nn.init.kaiming_uniform_(lat.weight, a=1)
            self.lateral_layers[tag] = lat

    def hip(self, features):
        """features: {tag: NHWC fp16 [1,h,w,Cin]} -> {tag: NHWC fp16 [1,h,w,trans_dim]}"""
        order = self.trans_tags[::-1]  # coarsest level first, as the reference concatenates them
        sizes = {t: features[t].shape[1:3] for t in order}
        B = features[order[0]].shape[0]
        if B != 1:
            raise NotImplementedError(
                "SeeCoder's decoder attends across the batch axis in the reference (seecoder.py:70,83); "
                "encode one image at a time, as app.py does")
        lens = [sizes[t][0] * sizes[t][1] for t in order]
        dev = features[order[0]].device
        toks = torch.empty((sum(lens), self.trans_dim), dtype=torch.float16, device=dev)
        lvl = L._dev16(self.level_embed)
        off = 0
        for idx, tag in enumerate(order):
            h, w = sizes[tag]
            seg = toks[off:off + lens[idx]]
            proj, gn = self.inproj_layers[tag]
            g, b = gn._pk()
            ops.groupnorm(proj.hip(features[tag]), g, b, gn.num_groups, gn.eps, out=seg.view(1, h, w, -1))
            ops.add_rowvec(seg, lvl[idx].contiguous(), out=seg)
            off += lens[idx]
        y = self.transformer.hip(toks)
        out, off = {}, 0
        for idx, tag in enumerate(order):
            h, w = sizes[tag]
            yi = y[off:off + lens[idx]].view(1, h, w, -1)
            out[tag] = ops.add(yi, self.lateral_layers[tag].hip(features[tag]))
            off += lens[idx]
        return out

    def forward(self, features):
        x = next(iter(features.values()))
        outs = self.hip({k: ops.to_nhwc(v) for k, v in features.items()})
        return {k: ops.to_nchw(v, x.dtype) for k, v in outs.items()}


@register('seecoder_query_transformer')
class QueryTransformer(nn.Module):
    def __init__(self, in_channels, hidden_dim, num_queries=[8, 144], nheads=8, num_layers=9, feedforward_dim=2048,
                 mask_dim=256, pre_norm=False, num_feature_levels=3, enforce_input_project=False,
                 with_fea2d_pos=True):
        super().__init__()
        self.pe_layer = PPE_MLP(freq_num=20, freq_max=None, out_channel=hidden_dim, mlp_layer=3) \
            if with_fea2d_pos else None
        if in_channels != hidden_dim or enforce_input_project:
            self.input_proj = nn.ModuleList()
            for _ in range(num_feature_levels):
                conv = L.Conv2d(in_channels, hidden_dim, kernel_size=1)
                nn.init.kaiming_uniform_(conv.weight, a=1)
                nn.init.constant_(conv.bias, 0)
                self.input_proj.append(conv)
        else:
            self.input_proj = None
        self.num_heads = nheads
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.transformer_selfatt_layers = nn.ModuleList()
        self.transformer_crossatt_layers = nn.ModuleList()
        self.transformer_feedforward_layers = nn.ModuleList()
        for _ in range(num_layers):
            self.transformer_selfatt_layers.append(
                SelfAttentionLayer(channels=hidden_dim, nhead=nheads, dropout=0.0, normalize_before=pre_norm))
            self.transformer_crossatt_layers.append(
                CrossAttentionLayer(channels=hidden_dim, nhead=nheads, dropout=0.0, normalize_before=pre_norm))
            self.transformer_feedforward_layers.append(
                FeedForwardLayer(channels=hidden_dim, hidden_channels=feedforward_dim, dropout=0.0,
                                 normalize_before=pre_norm))
        for stack in (self.transformer_selfatt_layers, self.transformer_crossatt_layers,
                      self.transformer_feedforward_layers):
            for p in stack.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)
        self.num_queries = num_queries
        num_gq, num_lq = num_queries
        self.init_query = nn.Embedding(num_gq + num_lq, hidden_dim)
        self.query_pos_embedding = nn.Embedding(num_gq + num_lq, hidden_dim)
        self.num_feature_levels = num_feature_levels
        self.level_embed = nn.Embedding(num_feature_levels, hidden_dim)

    def hip(self, x):
        """x: list of NHWC fp16 feature maps [1,h,w,C] (res3, res4, res5) -> [148, hidden] fp16"""
        assert len(x) == self.num_feature_levels
        if x[0].shape[0] != 1:
            raise NotImplementedError("one image at a time (see module docstring)")
        dev = x[0].device
        lvl = L._dev16(self.level_embed.weight)
        fea, fea_pos = [], []
        for i in range(self.num_feature_levels):
            _, h, w, _ = x[i].shape
            xi = self.input_proj[i].hip(x[i]) if self.input_proj is not None else x[i]
            fea.append(ops.add_rowvec(xi.reshape(h * w, -1), lvl[i].contiguous()))
            fea_pos.append(self.pe_layer.hip(h, w, dev) if self.pe_layer is not None else None)
        num_gq, num_lq = self.num_queries
        q = L._dev16(self.init_query.weight).clone()                # [gq+lq, C]: rows 0..gq-1 global
        qpos = L._dev16(self.query_pos_embedding.weight).contiguous()
        lq, lq_pos = q[num_gq:], qpos[num_gq:]
        for i in range(self.num_layers):
            k = i % self.num_feature_levels
            # local queries attend to the feature level; the result overwrites them in place
            self.transformer_crossatt_layers[i].hip(lq, fea[k], q_pos=lq_pos, k_pos=fea_pos[k], out=lq)
            # all queries attend to each other, then the FFN
            self.transformer_selfatt_layers[i].hip(q, qk_pos=qpos, out=q)
            self.transformer_feedforward_layers[i].hip(q, out=q)
        return q

    def forward(self, x):
        q = self.hip([ops.to_nhwc(xi) for xi in x])
        return q[None].to(x[0].dtype)


@register('seecoder')
class SemanticContextEncoder(nn.Module):
    def __init__(self, imencoder_cfg, imdecoder_cfg, qtransformer_cfg):
        super().__init__()
        self.imencoder = get_model()(imencoder_cfg)
        self.imdecoder = get_model()(imdecoder_cfg)
        self.qtransformer = get_model()(qtransformer_cfg)

    def hip(self, x_nhwc):
        fea = self.imencoder.hip(x_nhwc, want=('res3', 'res4', 'res5'))
        hs = self.imdecoder.hip({k: fea[k] for k in ('res3', 'res4', 'res5')})
        return self.qtransformer.hip([hs['res3'], hs['res4'], hs['res5']])

    @torch.no_grad()
    def forward(self, x):
        if x.shape[0] != 1:
            raise NotImplementedError(
                "SeeCoder encodes one reference image at a time (app.py:234-235): in the reference a batch "
                "B > 1 is mixed across images by the decoder's seq-first MultiheadAttention (seecoder.py:70,83)")
        return self.hip(ops.to_nhwc(x))[None].to(x.dtype)

    def encode(self, x):
        return self(x)
