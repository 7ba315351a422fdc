"""AutoencoderKL ('autoencoderkl') on the HIP path.

Surface of the reference's lib/model_zoo/autokl.py:14-60: ctor `(ddconfig, lossconfig,
embed_dim)`, sub-modules `encoder / decoder / quant_conv / post_quant_conv` (state-dict keys),
`encode(x, out_posterior=False)` (x in [0,1] -> x*2-1 -> moments -> sample) and `decode(z)`
((dec+1)/2 clamped to [0,1]).  The LPIPS/discriminator loss (autokl_utils.py) is training-only
and not built (`lossconfig: null`, autokl.yaml:21).

decode: the `1/scale * z` of pfd.vae_decode (pfd.py:277-281) rides on the NCHW->NHWC boundary
kernel and `(x+1)/2` + clamp on the NHWC->NCHW one, so the decoder body is 100 % fused
GN+swish / MFMA conv kernels.
"""
import torch
import torch.nn as nn

from ..hip import layers as L
from ..hip import ops
from .autokl_modules import Decoder, Encoder
from .common.get_model import register


class DiagonalGaussianDistribution(object):
    """posterior of the VAE encoder (reference distributions.py:24-62); [B, 2*zc, h, w] moments.
    A few KB of fp32 elementwise math on the host-visible result, not a kernel."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device,
                                                                      dtype=self.mean.dtype)

    def mode(self):
        return self.mean


@register('autoencoderkl')
class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig, embed_dim):
        super().__init__()
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        if lossconfig is not None:
            raise NotImplementedError("LPIPSWithDiscriminator is training-only and out of scope")
        assert ddconfig["double_z"]
        self.quant_conv = L.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = L.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim

    @torch.no_grad()
    def encode(self, x, out_posterior=False):
        return self.encode_trainable(x, out_posterior)

    def encode_trainable(self, x, out_posterior=False):
        h = self.encoder.hip(ops.to_nhwc(x, mul=2.0, add=-1.0))
        moments = ops.to_nchw(self.quant_conv.hip(h), x.dtype)
        posterior = DiagonalGaussianDistribution(moments)
        return posterior if out_posterior else posterior.sample()

    @torch.no_grad()
    def decode(self, z, in_scale=1.0, out_uint8=False):
        """z NCHW latent (already multiplied by in_scale inside) -> image NCHW in [0, 1];
        out_uint8: packed uint8 [B, H, W, 3] instead (the bytes `ToPILImage` would produce, app.py:273-275)"""
        h = self.decode_nhwc(ops.to_nhwc(z, mul=float(in_scale)))
        if out_uint8:
            return ops.image_u8(h, mul=0.5, add=0.5, f16_image=z.dtype == torch.float16)
        return ops.to_nchw(h, z.dtype, mul=0.5, add=0.5, lo=0.0, hi=1.0)

    def decode_nhwc(self, z_nhwc):
        return self.decoder.hip(self.post_quant_conv.hip(z_nhwc))

    def decode_trainable(self, z):
        h = self.decode_nhwc(ops.to_nhwc(z))
        return ops.to_nchw(h, z.dtype, mul=0.5, add=0.5)
