"""CPU, world_size 2, gloo: the sharding + all-gather logic of the multi-GPU path (the compute
itself needs a GPU and is covered by -m gpu; the collective here is the same call on gloo)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "prompt-free-diffusion_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lib.pipeline import all_gather_batch, shard_xT
    n_global = 8
    xT = shard_xT(n_global, 64, 64, seed=20, rank=rank, world_size=world)
    # stand-in for the per-sample denoise+decode (samples are independent): f(x) = 2x + 1
    img = all_gather_batch(xT * 2 + 1, world)
    full = shard_xT(n_global, 64, 64, seed=20, rank=0, world_size=1)
    ok = bool(torch.equal(img, full * 2 + 1)) and xT.shape[0] == n_global // world
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, ok, float(t)))
    dist.destroy_process_group()


class _StubNet:
    """stand-in for the GPU compute of the composite (samples are independent on the real path too):
    deterministic, per-sample functions with the real shapes"""
    device = 'cpu'
    num_timesteps = 1000

    def ctx_encode(self, image, which):
        assert which == 'image' and image.shape[0] == 1
        return image.mean().reshape(1, 1, 1).expand(1, 148, 768).clone()

    def vae_decode(self, z, which, out_uint8=False):
        img = z[:, :3].repeat_interleave(8, -1).repeat_interleave(8, -2).mul(0.1).add(0.5).clamp(0, 1)
        return (img * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous() if out_uint8 else img


class _StubSampler:
    def __init__(self, rank):
        self.rank, self.calls = rank, 0

    def sample(self, steps, shape, x_info, c_info, eta=0., verbose=True):
        import time
        self.calls += 1
        x = x_info['xt']
        assert list(x.shape) == list(shape) and c_info['conditioning'].shape == (shape[0], 148, 768)
        assert not bool(c_info['unconditional_conditioning'].any())
        time.sleep(0.02 * (self.rank + 1))               # ranks finish at different times
        return x * 0.5 + c_info['conditioning'][:, :1, :1].reshape(-1, 1, 1, 1), {}


def _pipeline_worker(rank, world, port, q):
    """the REAL PromptFreePipeline.generate(gather=True) + bench.py's timing reduction under a process group"""
    import sys
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "prompt-free-diffusion_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lib.pipeline import PromptFreePipeline, max_over_ranks
    n_global, H, W = 8, 64, 96
    image = torch.rand((1, 3, H, W), generator=torch.Generator().manual_seed(1234))
    pipe = PromptFreePipeline(_StubNet(), rank=rank, world_size=world, sampler=_StubSampler(rank))
    timings = {}
    dist.barrier()
    t0 = time.perf_counter()
    img, lat = pipe.generate(image, n_global, H, W, steps=5, scale=2.0, seed=20, gather=True, timings=timings)
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, world)
    u8, _ = pipe.generate(image, n_global, H, W, steps=5, scale=2.0, seed=20, gather=True, as_uint8=True)
    # the single-process answer for the global batch
    one = PromptFreePipeline(_StubNet(), rank=0, world_size=1, sampler=_StubSampler(0))
    ref, ref_lat = one.generate(image, n_global, H, W, steps=5, scale=2.0, seed=20, gather=True)
    n = n_global // world
    ok = (img.shape == (n_global, 3, H, W) and torch.equal(img, ref) and lat.shape[0] == n
          and torch.equal(lat, ref_lat[rank * n:(rank + 1) * n]) and u8.dtype == torch.uint8
          and u8.shape == (n_global, H, W, 3) and set(timings) == {'ctx_encode_ms', 'ddim_loop_ms', 'vae_decode_ms'}
          and timings['ddim_loop_ms'] >= 15.0 * (rank + 1))
    dist.barrier()
    q.put((rank, bool(ok), dt, dt_local))
    dist.destroy_process_group()


def test_pipeline_generate_gather_world2():
    """rank slicing, the one all-gather, per-stage timings and the max-over-ranks step time run exactly as
    PromptFreePipeline / bench.py write them (compute replaced by a per-sample stand-in)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert abs(res[0][2] - res[1][2]) < 1e-9                       # every rank holds the same reduced time ...
    assert res[0][2] >= max(r[3] for r in res) - 1e-9              # ... the slowest rank's


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == 2.0 for r in res)


def test_sharding_is_independent_of_world_size():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "prompt-free-diffusion_amd"))
    from lib.pipeline import shard_xT
    full = shard_xT(8, 64, 96, 7, 0, 1)
    for P in (2, 4, 8):
        parts = [shard_xT(8, 64, 96, 7, r, P) for r in range(P)]
        assert torch.equal(torch.cat(parts), full)
    assert full.shape == (8, 4, 8, 12)
