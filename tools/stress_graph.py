"""Stress the eager <-> hipGraph pipeline sequence of the GPU suite inside ONE process (the model build is the
slow part of a pytest run): small-shape graphs, the C2 shapes eagerly and graphed, pipelines dropped and
garbage-collected between rounds.  Used to chase the round-1 driver-side SIGABRT in test_full_size_properties.

    python tools/stress_graph.py [rounds] [name]
"""
import faulthandler
import gc
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "prompt-free-diffusion_amd"))
os.environ.setdefault("PFD_QUIET", "1")
faulthandler.enable()

import torch  # noqa: E402

from lib.pipeline import PromptFreePipeline, build_model  # noqa: E402
from lib.model_zoo.ddim import DDIMSampler  # noqa: E402


def log(*a):
    print(f"[{time.time() - T0:7.1f}s]", *a, flush=True)


T0 = time.time()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
name = sys.argv[2] if len(sys.argv) > 2 else 'pfd_seecoder_with_controlnet'
net = build_model(name, device='cuda', fp16=True)
log("model built")
g = torch.Generator().manual_seed(1234)
img_small = torch.rand((1, 3, 64, 64), generator=g)
img_c1 = torch.rand((1, 3, 256, 256), generator=g)
img_c2 = torch.rand((1, 3, 512, 512), generator=g)
for r in range(rounds):
    # sampler-only graph at a tiny latent (tests/test_hip_parity.py::test_sampler_graph...)
    cond = torch.randn((2, 148, 768), generator=g).cuda().half()
    s = DDIMSampler(net)
    s.enable_graph(True)
    for seed in (1, 2):
        xT = torch.randn([2, 4, 8, 8], generator=torch.Generator().manual_seed(seed))
        s.sample(steps=4, shape=[2, 4, 8, 8], x_info={'type': 'image', 'xt': xT.cuda()},
                 c_info={'type': 'image', 'conditioning': cond, 'unconditional_conditioning': torch.zeros_like(cond),
                         'unconditional_guidance_scale': 2.0}, eta=0., verbose=False)
    torch.cuda.synchronize()
    log(f"round {r}: sampler graph ok")
    eager, graphed = PromptFreePipeline(net), PromptFreePipeline(net)
    graphed.enable_graph(True)
    for seed in (5, 6):
        ie, xe = eager.generate(img_small, 2, 64, 64, steps=4, scale=2.0, seed=seed)
        ig, xg = graphed.generate(img_small, 2, 64, 64, steps=4, scale=2.0, seed=seed)
        assert torch.equal(xe, xg) and torch.equal(ie, ig)
    log(f"round {r}: small pipeline graphs ok")
    PromptFreePipeline(net).generate(img_c1, 1, 256, 256, steps=10, scale=2.0, seed=20)
    torch.cuda.synchronize()
    log(f"round {r}: C1 eager ok")
    eager, graphed = PromptFreePipeline(net), PromptFreePipeline(net)
    graphed.enable_graph(True)
    i4, x4 = eager.generate(img_c2, 4, 512, 512, steps=4, scale=2.0, seed=20)
    i4b, x4b = eager.generate(img_c2, 4, 512, 512, steps=4, scale=2.0, seed=20)
    torch.cuda.synchronize()
    assert torch.equal(x4, x4b) and torch.equal(i4, i4b)
    log(f"round {r}: C2 eager x2 ok")
    ig, xg = graphed.generate(img_c2, 4, 512, 512, steps=4, scale=2.0, seed=20)
    torch.cuda.synchronize()
    assert torch.equal(x4, xg) and torch.equal(i4, ig)
    log(f"round {r}: C2 graphed ok")
    ig, xg = graphed.generate(img_c2, 4, 512, 512, steps=4, scale=2.0, seed=20)
    i1, x1 = eager.generate(img_c2, 1, 512, 512, steps=4, scale=2.0, seed=20)
    torch.cuda.synchronize()
    assert torch.equal(x4, xg)
    log(f"round {r}: C2 replay + B=1 ok; mem {torch.cuda.memory_allocated() >> 20} MiB alloc, "
        f"{torch.cuda.memory_reserved() >> 20} MiB reserved")
    del eager, graphed, s
    gc.collect()
log("STRESS OK")
