import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "prompt-free-diffusion_amd")
for p in (PKG, os.path.join(REPO, "oracle"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("PFD_QUIET", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(REPO, "tests", "golden", "golden.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def state_spec():
    with open(os.path.join(REPO, "tests", "golden", "state_spec.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def param_shapes(state_spec):
    """{key: shape} of the floating-point parameters of pfd_seecoder_with_controlnet"""
    return {k: v["shape"] for k, v in state_spec.items() if v["param"]}


@pytest.fixture(scope="session")
def net(param_shapes):
    """the full composite on the GPU, fp16 (like app.py:117-129), seeded weights; shared by every GPU test file"""
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    from weights import seeded_tensor
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    cfg = model_cfg_bank()('pfd_seecoder_with_controlnet')
    cfg.args.vae_cfg_list[0][1].pth = None
    n = get_model()(cfg, verbose=False)
    sd = n.state_dict()
    for k, s in param_shapes.items():
        sd[k] = seeded_tensor(k, s, 0)
    n.load_state_dict(sd, strict=True)
    n.half()
    n.to('cuda')
    n.eval()
    return n


def seeded_sd(param_shapes, prefix):
    """fp32 CPU state dict of the sub-model under `prefix` (keys keep the full composite name)"""
    from weights import seeded_tensor
    return {k: seeded_tensor(k, s, 0) for k, s in param_shapes.items() if k.startswith(prefix)}


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
