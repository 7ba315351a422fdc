"""CPU, world_size 2, gloo: the sharding + all-gather logic of the multi-GPU path (the compute
itself needs a GPU and is covered by -m gpu; the collective here is the same call on gloo)."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


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


from stubs import StubNet as _StubNet, StubSampler as _StubSampler  # noqa: E402  (shared with `bench.py --stub`)


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


def _bench(*extra, timeout=300, env=None):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--stub", "--backend", "gloo", "--steps", "2",
                           "--warmup", "1", "--height", "64", "--width", "96", *extra],
                          capture_output=True, text=True, timeout=timeout, env=e)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE (how the driver runs --gpus 1) must not die at argument parsing: it
    spawns the two ranks under torch.distributed.run on 127.0.0.1, the ranks shard / all-gather / reduce their times
    (gloo + the stand-in compute here, RCCL + the HIP path on a GPU node) and ONE JSON line comes back."""
    r = _bench("--gpus", "2")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 8 and res["config"]["parallelism"] == "dp2"
    assert res["config"]["world_size_reported_by_backend"] == 2 and res["config"]["workload"].startswith("[c2-weak]")
    assert res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak" and res["value"] > 0
    assert res["ms_per_step"] >= 40.0            # the slowest rank's time (rank 1 sleeps 40 ms per batch)


def test_bench_launcher_fails_loudly_instead_of_hanging():
    """a job whose ranks cannot finish in time is killed and reported (exit code 124), not waited for forever"""
    r = _bench("--gpus", "2", "--launch-timeout", "0.5", timeout=120)
    assert r.returncode == 124 and "did not finish within" in r.stderr
    # a world size that contradicts --gpus is an error message, not a hang
    r = _bench("--gpus", "2", env={"WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_forced_collectives_at_world_size_one():
    """PFD_FORCE_COLLECTIVE=1: `bench.py --gpus 1` creates a 1-rank process group and the pipeline's all_gather /
    barrier / max-over-ranks all_reduce run through it (gloo + stub here; the same switch drives the one-GPU RCCL
    smoke, tests/test_hip_parity.py::test_rccl_one_rank_collectives)"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PFD_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--stub", "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "gloo" and d["config"]["collectives_forced_at_world_1"] is True
    assert d["config"]["world_size_reported_by_backend"] == 1 and d["n_gpus"] == 1
