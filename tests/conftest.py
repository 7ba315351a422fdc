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


class _OracleJobs:
    """The long fp32 CPU-oracle trajectories (tests/oracle_worker.py: C2 50 steps, C5 31 steps, C3 10 steps) run as
    CPU-only subprocesses from the start of a GPU session, next to the rest of the suite; a trajectory test asks for its
    case and waits for it.  Same oracle code, same inputs, same comparison -- only the wall-clock overlaps (the suite was
    21 minutes with the three runs in line, 18 of them oracle time)."""

    def __init__(self, cases):
        import subprocess
        import tempfile
        self.dir = tempfile.mkdtemp(prefix="pfd_oracle_")
        ncpu = os.cpu_count() or 1
        threads = max(1, min(64, ncpu // max(1, len(cases) + 1)))
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
        self.procs = {}
        for c in cases:
            out = os.path.join(self.dir, c + ".pt")
            log = open(os.path.join(self.dir, c + ".log"), "w")
            self.procs[c] = (subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "oracle_worker.py"), c, out,
                                               str(threads)], stdout=log, stderr=subprocess.STDOUT, env=env), out, log)

    def get(self, case, timeout=3000):
        if case not in self.procs:     # not started at session begin: start it now
            self.procs.update(_OracleJobs([case]).procs)
        proc, out, log = self.procs[case]
        rc = proc.wait(timeout=timeout)
        log.close()
        if rc != 0 or not os.path.exists(out):
            raise RuntimeError(f"oracle job {case} failed (rc {rc}):\n" + open(log.name).read()[-3000:])
        return torch.load(out)

    def close(self):
        for proc, _, log in self.procs.values():
            if proc.poll() is None:
                proc.kill()
            if not log.closed:
                log.close()


@pytest.fixture(scope="session", autouse=True)
def oracle_jobs(request):
    """started at session begin iff trajectory tests are among the selected items (a `-m gpu` run)"""
    want = []
    for item in request.session.items:
        for case, name in (("c5", "test_config_c5_trajectory_all_31_steps"), ("c2", "test_config_c2_trajectory_vs_oracle"),
                           ("c3", "test_config_c3_trajectory_vs_oracle")):
            if item.name == name and case not in want:
                want.append(case)
    jobs = _OracleJobs(want) if (want and torch.cuda.is_available()) else _OracleJobs([])
    yield jobs
    jobs.close()


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
