"""Swin Transformer backbone ('swin') on the HIP path -- SeeCoder's image encoder (Swin-L:
embed 192, depths 2/2/18/2, heads 6/12/24/48, window 12; configs/model/swin.yaml).

Module tree, state-dict keys (incl. the `relative_position_index` buffers) and constructor
kwargs follow the reference's lib/model_zoo/swin.py: `SwinTransformer` (:498-653), `BasicLayer`
(:354-453), `SwinTransformerBlock` (:213-310), `WindowAttention` (:132-210), `PatchMerging`
(:313-351), `PatchEmbed` (:456-495), `Mlp` (:82-101).

The reference materialises, per block: F.pad, torch.roll, window_partition, a [nW,144,144]
shift mask, a gathered [nH,144,144] bias, window_reverse, roll back and crop (:266-302,
:421-440, :192-195).  Here all of that is index arithmetic inside ONE kernel
(pfd_swin_window_attention_f16) that reads the token-major qkv matrix and writes the
token-major attention output; tokens never move.  PatchMerging's 2x2 gather is fused into its
LayerNorm.  Activations are [B*H*W, C] fp16 throughout.
"""
import torch
import torch.nn as nn

from ..hip import layers as L
from ..hip import ops
from .common.get_model import register


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = L.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = L.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def hip(self, x, res=None):
        return self.fc2.hip(self.fc1.hip(x, act=ops.ACT_GELU), res=res)


class WindowAttention(nn.Module, L._Packed):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.window_size = to_2tuple(window_size)
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        # index of (query token i, key token j) into the table: (dy + wh-1) * (2ww-1) + (dx + ww-1)
        ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing='ij')
        ys, xs = ys.flatten(), xs.flatten()
        rel = (ys[:, None] - ys[None, :] + wh - 1) * (2 * ww - 1) + (xs[:, None] - xs[None, :] + ww - 1)
        self.register_buffer("relative_position_index", rel)
        self.qkv = L.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = L.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def _pk_rpb(self):
        return self._packed("rpb", lambda: L._dev16(self.relative_position_bias_table).contiguous(),
                            self.relative_position_bias_table)

    def hip(self, xn, B, H, W_, shift, res):
        """xn: LayerNorm'ed tokens [B*H*W, C]; returns proj(window_attention) + res"""
        if self.window_size != (12, 12) or self.dim // self.num_heads != 32 or self.qkv.bias is None:
            raise NotImplementedError("pfd_swin_window_attention_f16 is built for window 12, head_dim 32")
        qkv = self.qkv.hip(xn)
        qkv_bias = self.qkv._pk()[1]
        a = ops.swin_window_attention(qkv, qkv_bias, self._pk_rpb(), B, H, W_, self.dim, self.num_heads, 12, shift,
                                      float(self.scale))
        return self.proj.hip(a, res=res)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = L.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size=to_2tuple(window_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()  # stochastic depth is training-only
        self.norm2 = L.LayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H = None
        self.W = None

    def hip(self, x, B, H, W_):
        x = self.attn.hip(self.norm1.hip(x), B, H, W_, self.shift_size, res=x)
        return self.mlp.hip(self.norm2.hip(x), res=x)


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = L.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = L.LayerNorm(4 * dim)

    def hip(self, x, B, H, W_):
        g, b = self.norm._pk()
        xm = ops.layernorm_patch_merge(x.view(B, H, W_, self.dim), g, b, self.norm.eps)
        return self.reduction.hip(xm)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size = window_size
        self.shift_size = window_size // 2
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if (i % 2 == 0) else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def hip(self, x, B, H, W_):
        for blk in self.blocks:
            blk.H, blk.W = H, W_
            x = blk.hip(x, B, H, W_)
        if self.downsample is not None:
            return x, H, W_, self.downsample.hip(x, B, H, W_), (H + 1) // 2, (W_ + 1) // 2
        return x, H, W_, x, H, W_


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = to_2tuple(patch_size)
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = L.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = L.LayerNorm(embed_dim) if norm_layer is not None else None

    def hip(self, x):
        """x NHWC fp16 [B,H,W,3] -> tokens [B*Wh*Ww, C], Wh, Ww.  Right/bottom zero padding to a
        multiple of the patch size (swin.py:481-485) = the conv's out-of-range taps reading 0."""
        B, H, W_, _ = x.shape
        p = self.patch_size[0]
        Wh, Ww = (H + p - 1) // p, (W_ + p - 1) // p
        y = self.proj.hip(x, out_hw=(Wh, Ww)).view(B * Wh * Ww, self.embed_dim)
        if self.norm is not None:
            y = self.norm.hip(y)
        return y, Wh, Ww


@register('swin')
class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        if ape:
            raise NotImplementedError("absolute position embedding (ape) is not used by SeeCoder")
        self.pretrain_img_size = pretrain_img_size
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.ape = ape
        self.patch_norm = patch_norm
        self.out_indices = out_indices
        self.frozen_stages = frozen_stages
        self.patch_embed = PatchEmbed(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if (i < self.num_layers - 1) else None, use_checkpoint=use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f'norm{i}', L.LayerNorm(self.num_features[i]))

    def hip(self, x, want=('res2', 'res3', 'res4', 'res5')):
        """x: NHWC fp16 image [B,H,W,3]; returns {tag: NHWC fp16 [B,h,w,C]} for the wanted stages"""
        B = x.shape[0]
        t, Wh, Ww = self.patch_embed.hip(x)
        outs = {}
        for i in range(self.num_layers):
            x_out, H, W_, t, Wh, Ww = self.layers[i].hip(t, B, Wh, Ww)
            tag = f'res{i + 2}'
            if i in self.out_indices and tag in want:
                outs[tag] = getattr(self, f'norm{i}').hip(x_out).view(B, H, W_, self.num_features[i])
        return outs

    def forward(self, x):
        outs = self.hip(ops.to_nhwc(x))
        return {k: ops.to_nchw(v, x.dtype) for k, v in outs.items()}
