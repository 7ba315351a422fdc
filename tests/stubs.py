"""TEST INFRASTRUCTURE: per-sample stand-ins for the GPU compute of the composite, used by the world-size-2 gloo tests
(tests/test_distributed_cpu.py) and by `bench.py --stub` (the launcher / sharding / collective path without a GPU).
Samples are independent on the real path too; these are deterministic per-sample functions with the real shapes."""
import time

import torch


class StubNet:
    device = 'cpu'
    num_timesteps = 1000

    def ctx_encode(self, image, which):
        assert which == 'image' and image.shape[0] == 1
        return image.mean().reshape(1, 1, 1).expand(1, 148, 768).clone()

    def vae_decode(self, z, which, out_uint8=False):
        img = z[:, :3].repeat_interleave(8, -1).repeat_interleave(8, -2).mul(0.1).add(0.5).clamp(0, 1)
        return (img * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous() if out_uint8 else img


class StubSampler:
    def __init__(self, rank):
        self.rank, self.calls = rank, 0

    def enable_graph(self, on=True):
        pass

    def sample(self, steps, shape, x_info, c_info, eta=0., verbose=True):
        self.calls += 1
        x = x_info['xt']
        assert list(x.shape) == list(shape) and c_info['conditioning'].shape == (shape[0], 148, 768)
        assert not bool(c_info['unconditional_conditioning'].any())
        time.sleep(0.02 * (self.rank + 1))               # ranks finish at different times
        return x * 0.5 + c_info['conditioning'][:, :1, :1].reshape(-1, 1, 1, 1), {}
