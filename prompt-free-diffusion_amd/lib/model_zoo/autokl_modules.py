"""AutoencoderKL encoder / decoder stacks on the HIP path.

Module tree and state-dict keys follow the reference's lib/model_zoo/autokl_modules.py:
`ResnetBlock` (:82-141), `AttnBlock` (:150-202, single head, d = C, scale C^-1/2),
`Upsample` (:42-57), `Downsample` (:60-79, asymmetric bottom/right zero pad + stride-2 conv),
`Encoder` (:368-459), `Decoder` (:462-568).  Every norm is GroupNorm(32, eps 1e-6) followed by
swish; both are fused into one kernel, and nearest-2x upsampling is a gather inside the
following conv.  The unused classes of that file (:216-365, :571-835) are out of scope.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from ..hip import layers as L
from ..hip import ops


def Normalize(in_channels, num_groups=32):
    return L.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv, "only the conv form is on the hot path"
        self.with_conv = with_conv
        self.conv = L.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def hip(self, x):
        return self.conv.hip(x, ups=True)


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv, "only the conv form is on the hot path"
        self.with_conv = with_conv
        self.conv = L.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def hip(self, x):
        # F.pad(x, (0,1,0,1)) then conv(stride 2, pad 0): out = floor((H+1-3)/2)+1; the taps that
        # fall on the padded row/column read zero inside the conv's bounds check
        H, W_ = x.shape[1], x.shape[2]
        return self.conv.hip(x, out_hw=((H + 1 - 3) // 2 + 1, (W_ + 1 - 3) // 2 + 1))


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        assert temb_channels == 0 and not conv_shortcut, "VAE blocks have no time embedding"
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = L.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = L.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = L.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def hip(self, x, temb=None):
        h = self.conv1.hip(self.norm1.hip(x, silu=True))
        h = self.norm2.hip(h, silu=True)
        sk = self.nin_shortcut.hip(x) if self.in_channels != self.out_channels else x
        return self.conv2.hip(h, res=sk)


class AttnBlock(nn.Module):
    """single-head spatial self-attention with d = C = 512 (autokl_modules.py:186-197): ONE fused launch
    (pfd_attention_f16 with D = 512 -> attention512_kernel: QK^T, online softmax and PV per 128-query tile, four
    128-column slices of V per tile) on Q / K from the 1x1 convolutions and the transposed V the v-projection GEMM writes
    directly.  No [N, N] score matrix exists at any resolution (the reference materialises it in fp32: 2.7 GB per image
    at the 36 864 tokens of a 1536^2 output).  PFD_VAE_ATTN=gemm selects the round-1/2 form for A/B measurements:
    scores via GEMM in fp16 for `ROWS` query rows at a time, row softmax, PV via GEMM."""
    ROWS = 4096

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = L.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = L.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = L.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = L.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def hip(self, x):
        B, H, W_, Cc = x.shape
        N = H * W_
        if N % 64:
            raise NotImplementedError(f"VAE attention over {N} tokens (need a multiple of 64)")
        hn = self.norm.hip(x)
        q = self.q.hip(hn).view(B, N, Cc)
        k = self.k.hip(hn).view(B, N, Cc)
        wv, bv = self.v._pk()
        vt = ops.gemm(wv, hn.view(B * N, Cc), bias=bv, bias_per_row=True)      # [C, B*N]
        scale = float(int(Cc) ** (-0.5))
        if Cc == 512 and os.environ.get("PFD_VAE_ATTN", "fused") != "gemm":
            o = ops.attention(q, k, vt, B, 1, N, N, Cc, scale, ldq=Cc, ldk=Cc, ldvt=B * N, q_bs=N * Cc, k_bs=N * Cc,
                              vt_bs=N)
            return self.proj_out.hip(o.view(B, H, W_, Cc), res=x)
        o = torch.empty((B, N, Cc), dtype=torch.float16, device=x.device)
        rows = min(N, self.ROWS)
        s = torch.empty((rows, N), dtype=torch.float16, device=x.device)        # one scratch for all chunks
        for b in range(B):
            for r0 in range(0, N, rows):
                r1 = min(N, r0 + rows)
                sc = s[:r1 - r0]
                ops.gemm(q[b, r0:r1], k[b], out=sc)                              # scores of this row block
                ops.softmax_rows(sc, scale, out=sc)
                ops.gemm(sc, vt[:, b * N:(b + 1) * N], out=o[b, r0:r1])
        return self.proj_out.hip(o.view(B, H, W_, Cc), res=x)


def make_attn(in_channels, attn_type="vanilla"):
    assert attn_type in ("vanilla", "none"), f'attn_type {attn_type} is not on the hot path'
    return AttnBlock(in_channels) if attn_type == "vanilla" else nn.Identity()


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True,
                 use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        assert not use_linear_attn
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.conv_in = L.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0,
                                         dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = L.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1,
                                 padding=1)

    def hip(self, x):
        h = self.conv_in.hip(x)
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = lvl.block[i_block].hip(h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].hip(h)
            if i_level != self.num_resolutions - 1:
                h = lvl.downsample.hip(h)
        h = self.mid.block_2.hip(self.mid.attn_1.hip(self.mid.block_1.hip(h)))
        return self.conv_out.hip(self.norm_out.hip(h, silu=True))

    def forward(self, x):
        return ops.to_nchw(self.hip(ops.to_nhwc(x)), x.dtype)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        assert not use_linear_attn and not tanh_out
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.give_pre_end = give_pre_end
        self.tanh_out = tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        print("Working with z of shape {} = {} dimensions.".format(self.z_shape, np.prod(self.z_shape)))
        self.conv_in = L.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0,
                                         dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)  # keeps up[i] = resolution level i
        self.norm_out = Normalize(block_in)
        self.conv_out = L.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def hip(self, z):
        self.last_z_shape = z.shape
        h = self.conv_in.hip(z)
        h = self.mid.block_2.hip(self.mid.attn_1.hip(self.mid.block_1.hip(h)))
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = lvl.block[i_block].hip(h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block].hip(h)
            if i_level != 0:
                h = lvl.upsample.hip(h)
        if self.give_pre_end:
            return h
        return self.conv_out.hip(self.norm_out.hip(h, silu=True))

    def forward(self, z):
        return ops.to_nchw(self.hip(ops.to_nhwc(z)), z.dtype)
