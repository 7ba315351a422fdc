"""Request front-end of the hot path: reference image -> SeeCoder context -> DDIM/CFG loop ->
VAE decode, for one rank's shard of a (possibly multi-GPU) batch.

Restates the arithmetic-free glue of the reference's `prompt_free_diffusion.action_inference`
(app.py:229-274): ctx = ctx_encode(image).repeat(n); uncond = zeros_like(ctx) (or a padded fixed
tensor for SeeCoder-Anime, app.py:236-241); x_T ~ N(0, 1) of shape [n, 4, H/8, W/8];
DDIMSampler.sample(steps, eta, CFG scale, optional control); vae_decode.  The Gradio UI itself
is out of scope.

Multi-GPU (new, the reference is single-device): samples are independent, so rank r of P takes
samples [r*n/P, (r+1)*n/P) of the global batch.  x_T is drawn ONCE for the global shape from a
CPU generator and sliced, so results do not depend on P.  The only collective is one RCCL
all-gather of the decoded images after the loop (`gather=True`).
"""
import os

import torch

from .hip import ops
from .hip.ops import serialised


def build_model(name='pfd_seecoder', device='cuda', fp16=True, randomize_zero_init=True, seed=0, verbose=False):
    """Construct a composite from the config bank with synthetic weights (no checkpoints exist in
    the build/bench environment): default initialisation, plus re-randomised zero-initialised
    tensors so that no branch of the network is multiplied by zero."""
    from .cfg_helper import model_cfg_bank
    from .model_zoo import get_model
    cfg = model_cfg_bank()(name)
    cfg.args.vae_cfg_list[0][1].pth = None  # checkpoint not available: synthetic weights
    torch.manual_seed(seed)
    if device != 'cpu' and torch.cuda.is_available():
        with torch.device(device):
            net = get_model()(cfg, verbose=verbose)
    else:
        net = get_model()(cfg, verbose=verbose)
    if randomize_zero_init:
        g = torch.Generator(device='cpu').manual_seed(seed + 1)
        with torch.no_grad():
            for p in net.parameters():
                if p.is_floating_point() and p.numel() > 0 and float(p.detach().abs().max()) == 0.0:
                    fan_in = max(1, p.numel() // p.shape[0]) if p.dim() > 1 else 1
                    std = fan_in ** -0.5 if p.dim() > 1 else 0.02
                    p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device, p.dtype))
    if fp16:
        net.half()
    net.to(device)
    net.eval()
    return net


def shard_xT(n_global, height, width, seed, rank, world_size):
    """rank's slice of the x_T drawn once for the GLOBAL batch (CPU generator => independent of P)"""
    assert n_global % world_size == 0, "global batch must divide over the ranks"
    n = n_global // world_size
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randn([n_global, 4, height // 8, width // 8], generator=g)[rank * n:(rank + 1) * n]


def force_collective():
    """PFD_FORCE_COLLECTIVE=1 with an initialised process group: the collectives run even at world size 1 -- the
    one-GPU RCCL smoke (tests/test_hip_parity.py::test_rccl_one_rank_collectives, `bench.py --gpus 1` under that
    environment): communicator creation, all_gather and all_reduce on an MI355X without a multi-GPU node"""
    if os.environ.get("PFD_FORCE_COLLECTIVE") != "1":
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def all_gather_batch(local, world_size):
    """the one collective of the path: concatenate every rank's shard in rank order"""
    if world_size == 1 and not force_collective():
        return local
    import torch.distributed as dist
    out = [torch.empty_like(local) for _ in range(world_size)]
    dist.all_gather(out, local.contiguous())  # RCCL over xGMI on the GPU box; gloo in the CPU tests
    return torch.cat(out, 0)


class _GraphedStage:
    """One stage (context encode, VAE decode) replayed as a hipGraph: ~920 / ~105 kernel launches per call at C2
    (426 / 75 of them the library's, tools/stage_launches.py) whose issue cost is otherwise paid by the host every
    batch (SeeCoder: 5.8 ms of device time in launches of 4-16 us).  One
    graph per (input shape, dtype, weights identity+version of the sub-model); static input buffer owned here;
    the same kernels as the eager path.  A capture that fails RAISES (like DDIMSampler._capture): a stage
    that silently fell back to eager launches would hide exactly the kind of capture-illegal call (host sync,
    pageable H2D copy) that corrupts a replay."""

    def __init__(self, net, method, which, module):
        # (net, method name) instead of a closure over the pipeline: no reference cycle, so the graphs and
        # their private memory pools are released by refcount when the pipeline is dropped, not at some
        # later cyclic-GC pass (possibly in the middle of another stream capture)
        self.net, self.method, self.which, self.module = net, method, which, module
        self.graphs = {}

    def fn(self, x):
        return getattr(self.net, self.method)(x, self.which)

    def _signature(self):
        from .hip.layers import generation
        return hash((generation(),) + tuple((p.data_ptr(), p._version) for p in self.module.parameters()))

    @serialised
    def __call__(self, x):
        if not x.is_cuda:
            raise RuntimeError("HIP path: stage input must be on the GPU (no CPU fallback)")
        key = (tuple(x.shape), x.dtype, self._signature())
        ent = self.graphs.get(key)
        if ent is None:
            from .hip import binding
            binding.prof_enable(False)
            sx = x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):     # warm-up outside capture: packs weights, sizes the allocator
                self.fn(sx)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.fn(sx)
            if len(self.graphs) >= 4:
                torch.cuda.synchronize()      # never drop a graph whose replay may still be in flight
                self.graphs.clear()
            ent = self.graphs[key] = (g, sx, out)
        g, sx, out = ent
        sx.copy_(x)
        g.replay()
        return out.clone()


class _Marks:
    """stage boundaries of one request: HIP events on a GPU model, wall clock otherwise (CPU tests)"""

    def __init__(self, on_gpu):
        self.on_gpu, self.t = on_gpu, []

    def mark(self):
        if self.on_gpu:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.t.append(e)
        else:
            import time
            self.t.append(time.perf_counter())

    def ms(self, i, j):
        if self.on_gpu:
            return self.t[i].elapsed_time(self.t[j])
        return (self.t[j] - self.t[i]) * 1e3


def max_over_ranks(seconds, world_size, device='cpu'):
    """the timing reduction of bench.py: a step is as slow as its slowest rank"""
    if world_size == 1 and not force_collective():
        return float(seconds)
    import torch.distributed as dist
    t = torch.tensor([float(seconds)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class PromptFreePipeline:
    """One rank's view of a (possibly multi-GPU) request: shard -> encode -> DDIM loop -> decode -> gather.
    `sampler` is injectable (the gloo CPU tests drive this very code with a stand-in for the GPU compute)."""

    def __init__(self, net, rank=0, world_size=1, sampler=None):
        self.net = net
        if sampler is None:
            from .model_zoo.ddim import DDIMSampler
            sampler = DDIMSampler(net)
        self.sampler = sampler
        self.rank, self.world_size = rank, world_size
        self._ctx_stage = self._vae_stage = self._stages = None

    def enable_graph(self, on=True):
        """hipGraph replay for all three stages: the DDIM loop (DDIMSampler.enable_graph) and, here, the
        context encode and the VAE decode"""
        self.sampler.enable_graph(on)
        if on and self._stages is None:
            self._stages = (_GraphedStage(self.net, 'ctx_encode', 'image', self.net.ctx['image']),
                            _GraphedStage(self.net, 'vae_decode', 'image', self.net.vae['image']))
        self._ctx_stage, self._vae_stage = self._stages if on else (None, None)

    @serialised
    @torch.no_grad()
    def encode_reference(self, image, n):
        """image: [1,3,H,W] in [0,1] -> (cond [n,148,768], uncond zeros), app.py:234-236"""
        c = self._ctx_stage(image) if self._ctx_stage is not None else self.net.ctx_encode(image, 'image')
        cond = c.repeat(n, 1, 1)
        return cond, torch.zeros_like(cond)

    @serialised
    @torch.no_grad()
    def generate(self, image, n_global, height, width, steps=50, scale=2.0, eta=0.0, seed=20, control=None,
                 uncond=None, decode=True, gather=False, verbose=False, timings=None, as_uint8=False):
        """returns (images, latents [n_local,4,h,w]); images = [n_local | n_global (gather), 3, H, W] in [0,1]
        in the model dtype, or -- as_uint8 -- packed uint8 [n, H, W, 3] (the bytes ToPILImage would produce,
        app.py:273-275; 4x fewer bytes through the all-gather); decode=False returns the latents twice"""
        P, r = self.world_size, self.rank
        dev = self.net.device
        mk = _Marks(torch.device(dev).type == 'cuda') if timings is not None else None
        if mk:
            mk.mark()
        xT = shard_xT(n_global, height, width, seed, r, P)
        n = xT.shape[0]
        if image.shape[0] > 1:
            # one reference image PER SAMPLE (the "+737 GFLOP/img" variant): SeeCoder is run once per image
            # (its decoder MHA attends across the batch axis, seecoder.py:70,83 -- a batch of images would mix them)
            if image.shape[0] != n:
                raise ValueError(f"{image.shape[0]} reference images for {n} local samples")
            pairs = [self.encode_reference(image[i:i + 1].to(dev), 1) for i in range(n)]
            cond, zeros = torch.cat([c for c, _ in pairs]), torch.cat([z for _, z in pairs])
        else:
            cond, zeros = self.encode_reference(image.to(dev), n)
        if uncond is None:
            uncond = zeros
        if mk:
            mk.mark()
        x_info = {'type': 'image', 'xt': xT.to(dev)}
        c_info = {'type': 'image', 'conditioning': cond, 'unconditional_conditioning': uncond,
                  'unconditional_guidance_scale': scale}
        if control is not None:
            c_info['control'] = control.to(dev)
        x, _ = self.sampler.sample(steps=steps, shape=list(xT.shape), x_info=x_info, c_info=c_info, eta=eta,
                                   verbose=verbose)
        if mk:
            mk.mark()
        if not decode:
            return x, x
        if as_uint8:
            img = self.net.vae_decode(x, 'image', out_uint8=True)
        else:
            img = self._vae_stage(x) if self._vae_stage is not None else self.net.vae_decode(x, 'image')
        if mk:
            mk.mark()
            if mk.on_gpu:
                torch.cuda.synchronize()
            timings.update(ctx_encode_ms=mk.ms(0, 1), ddim_loop_ms=mk.ms(1, 2), vae_decode_ms=mk.ms(2, 3))
        if gather:
            img = all_gather_batch(img, P)
        return img, x
