#!/usr/bin/env python
"""Headline benchmark: images/s @512x512, 50-step DDIM (CFG 2.0), SD-v1.5 UNet + SeeCoder, fp16.

One "step" = one pass of the whole hot path over one batch: SeeCoder encode of the reference
image, 50 DDIM steps with classifier-free guidance (UNet batch 2B), AutoKL decode, and -- on
N > 1 GPUs -- the one RCCL all-gather of the decoded images.  N=1 runs BASELINE.json configs[1]
(batch 4); N>1 keeps 4 images per GPU (weak scaling, global batch 4N; configs[3] is the 8-GPU
point of the same family).  Synthetic data, synthetic weights of the exact architecture (no
checkpoints / datasets in the environment).

  python bench.py [--gpus N --steps K --warmup W]
N > 1 without WORLD_SIZE in the environment: this process starts the N ranks itself (one per GPU,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), relays rank 0's JSON line
and fails loudly -- non-zero exit, the ranks' stderr -- on a rank failure or after --launch-timeout seconds instead of
hanging; under torch.distributed.run (WORLD_SIZE set, what the driver does) it is simply one of the ranks.

Prints ONE JSON line (rank 0).  `roofline` is measured live inside this process with HIP-event pairs
around every launch of every kernel family on its launch stream (libpfd_hip's pfd_prof_*): with
--no-graph during the timed region itself; by default the timed region replays hipGraphs (events cannot
be captured), so the same kernels, shapes and data path are timed on ONE more, eagerly launched, batch
right after the timed region (`roofline.measured_on` says which; rocprofv3 summaries of the same command
are under profiles/).  Event pairs around eager launches also contain the idle time in front of each kernel
(the host enqueues ~20 k launches per batch); every kernel of the library is instrumented, so that gap is estimated
as (sum of the event pairs - graph-replayed wall time of the same batch) / launches and subtracted PER LAUNCH
(`roofline.eager_gap_us_per_launch`; un-corrected value in `avg_launch_ms_eager`) -- an estimate of the in-graph
duration rocprofv3 reports.
`roofline.traffic` = HBM bytes per launch from rocprofv3 PMC passes
(profiles/pmc_traffic.json, see tools/pmc_bucket.py).
`cpu_baseline` times the CPU oracle (a port of the reference's algorithm, oracle/pfd_oracle.py)
on a bounded sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "prompt-free-diffusion_amd"), os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("PFD_QUIET", "1")

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def _reference_vs_port():
    """the oracle port against the reference's own modules on one host (oracle/time_reference_vs_port.py, run in the
    build container where /root/reference exists; that script writes the tracked file read here)"""
    path = os.path.join(REPO, "profiles", "cpu_reference_vs_port.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return {"max_abs_diff": d["max_abs_diff"], "port_time_over_reference_time": d["port_time_over_reference_time"],
                "measured": d.get("measured"), "host_cpus": d.get("host_cpus"), "measured_in_this_run": False,
                "source": "profiles/cpu_reference_vs_port.json (oracle/time_reference_vs_port.py)"}
    except Exception as e:   # noqa: BLE001
        return {"measured_in_this_run": False, "source": f"unavailable: {e}"}


def cpu_baseline(net, height, width, ddim_steps, scale):
    """CPU path on the host cores, same workload, bounded sample: 1 SeeCoder encode + 1 CFG UNet step
    (batch 2) + 1 VAE decode for ONE image, extrapolated over the schedule (every step costs the same).
    kind = "port": the reference itself is Python over /root/reference, which does not exist on the GPU box;
    what runs here is oracle/pfd_oracle.py, the op-for-op restatement pinned to it by tests/test_oracle_golden.py
    (same torch CPU kernels: F.conv2d, materialised softmax(QK^T)V, F.group_norm ...).
    Thread count: the REAL oracle CFG UNet step (the 50x term of the estimate) is timed at up to three thread counts
    and the fastest is used for every stage -- all of a 256-thread host's cores is several times slower than 32-64."""
    import torch
    import pfd_oracle as O
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()
          if k.startswith(("diffuser.image.", "vae.image.", "ctx.image."))}
    g = torch.Generator().manual_seed(1234)
    img = torch.rand((1, 3, height, width), generator=g)
    x0 = torch.randn((1, 4, height // 8, width // 8), generator=g)
    tt = torch.tensor([981])
    eps_fn = lambda xx, ttt, cc: O.unet_apply(sd, "diffuser.image.", xx, ttt, cc)  # noqa: E731
    ncpu = os.cpu_count() or 1
    cands = sorted({n for n in (16, 32, 64) if n <= ncpu} or {ncpu})
    sweep, threads, t_step = {}, cands[0], float("inf")
    with torch.no_grad():
        ctx0 = torch.randn((1, 148, 768), generator=g)
        for n in cands:
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            O.ddim_step(eps_fn, x0, tt, ctx0, torch.zeros_like(ctx0), scale, 0.5, 0.6, 0.0)
            sweep[n] = round(time.perf_counter() - t0, 3)
            if sweep[n] < t_step:
                threads, t_step = n, sweep[n]
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        ctx = O.seecoder_encode(sd, "ctx.image.", img)
        t1 = time.perf_counter()
        x, _ = O.ddim_step(eps_fn, x0, tt, ctx, torch.zeros_like(ctx), scale, 0.5, 0.6, 0.0)
        t2 = time.perf_counter()
        O.vae_decode(sd, "vae.image.", x)
        t3 = time.perf_counter()
    t_ctx, t_step, t_vae = t1 - t0, min(t_step, t2 - t1), t3 - t2
    per_image = t_ctx + ddim_steps * t_step + t_vae
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(), "unet_step_s_by_threads": sweep,
            "reference_vs_port": _reference_vs_port(),
            "sample": f"1 image {height}x{width}: SeeCoder encode {t_ctx:.2f}s + 1 CFG UNet step (batch 2) "
                      f"{t_step:.2f}s x{ddim_steps} (extrapolated) + VAE decode {t_vae:.2f}s, fp32 torch CPU, "
                      f"{threads} of {os.cpu_count()} host threads (fastest of {cands} on the real UNet step)"}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a process group: start the N ranks (one per GPU) under torch.distributed.run on
    127.0.0.1, relay rank 0's stdout (the JSON line), return the launcher's exit code; kill the whole group and fail on
    --launch-timeout.  The ranks' stderr goes to this process's stderr."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        proc.wait()
        print(f"bench.py: the {args.gpus}-rank job did not finish within {args.launch_timeout} s -- killed "
              f"(a rank that never reaches the process group, or a collective that hangs)", file=sys.stderr, flush=True)
        return 124
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        print(f"bench.py: the {args.gpus}-rank job failed (exit code {proc.returncode}, "
              f"{len(lines)} result line(s)); rank output:\n{out[-4000:]}", file=sys.stderr, flush=True)
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--scale", type=float, default=2.0)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json config: c2 = headline (default; N > 1 keeps its 4 images per GPU); c3 = + ControlNet "
                         "+ SeeCoder-PA; c4 = 8 images per GPU (configs[3]: global batch 64 on 8 GPUs; on one GPU it is that "
                         "run's per-rank workload, UNet batch 16); c5 = 768x768, 30 (->31) steps, batch 2, non-zero "
                         "unconditional context")
    ap.add_argument("--per-sample-image", action="store_true",
                    help="one reference image (one SeeCoder encode) per sample instead of one per batch (SURVEY 8(d))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the DDIM loop eagerly instead of one hipGraph")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help='"nccl" is RCCL on ROCm; gloo: CPU tests')
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("PFD_BENCH_TIMEOUT", "3000")),
                    help="seconds before a self-started multi-rank job is killed")
    ap.add_argument("--pg-timeout", type=float, default=600.0, help="process-group timeout (rendezvous and collectives), seconds")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY: per-sample stand-ins for the GPU compute (tests/stubs.py) on the CPU, to drive the "
                         "launcher / sharding / collective path without a GPU (tests/test_distributed_cpu.py)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    if args.config == "c5":
        args.height = args.width = 768
        args.ddim_steps, args.batch = 30, 2
    if args.config == "c4":
        args.batch = 8

    import datetime
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` "
                         f"(it spawns the ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus")
    on_gpu = not args.stub
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
        if world > torch.cuda.device_count():
            raise SystemExit(f"bench.py: {world} ranks for {torch.cuda.device_count()} visible GPU(s): one process per GPU")
        torch.cuda.set_device(local)
    dev = f'cuda:{local}' if on_gpu else 'cpu'
    # PFD_FORCE_COLLECTIVE=1: a process group (and the path's collectives) at world size 1 too -- the one-GPU RCCL smoke
    forced = os.environ.get("PFD_FORCE_COLLECTIVE") == "1"
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced and world == 1:
            os.environ.setdefault("MASTER_PORT", "29531")
        try:
            dist.init_process_group(args.backend, rank=rank, world_size=world,   # "nccl" == RCCL on ROCm
                                    timeout=datetime.timedelta(seconds=args.pg_timeout))
        except Exception as e:   # noqa: BLE001
            raise SystemExit(f"bench.py rank {rank}: process group ({args.backend}, world {world}, "
                             f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}) failed: {e}")

    from lib.pipeline import PromptFreePipeline, max_over_ranks
    import contextlib
    if args.stub:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import stubs
        binding = None
        net = stubs.StubNet()
        pipe = PromptFreePipeline(net, rank=rank, world_size=world, sampler=stubs.StubSampler(rank))
        args.no_prof = args.no_cpu_baseline = True
    else:
        from lib.hip import binding
        from lib.pipeline import build_model
        with contextlib.redirect_stdout(sys.stderr):   # constructors print (like the reference's); keep stdout = the JSON line
            net = build_model('pfd_seecoder_with_controlnet' if args.config == "c3" else 'pfd_seecoder',
                              device=dev, fp16=True)
            if args.config == "c3":   # SeeCoder-PA: attach the position-aware MLP like app.py:166-177
                from lib.model_zoo.seecoder import PPE_MLP
                pe = PPE_MLP(freq_num=20, freq_max=None, out_channel=768, mlp_layer=3)
                torch.nn.init.normal_(pe.mlp[-1].weight, std=768 ** -0.5)
                net.ctx['image'].qtransformer.pe_layer = pe.half().to(dev)
        pipe = PromptFreePipeline(net, rank=rank, world_size=world)
        pipe.enable_graph(not args.no_graph)
    image = torch.rand((args.batch if args.per_sample_image else 1, 3, args.height, args.width),
                       generator=torch.Generator().manual_seed(1234))
    n_global = args.batch * world
    gen = torch.Generator().manual_seed(4321)
    control = torch.rand((1, 3, args.height, args.width), generator=gen) if args.config == "c3" else None
    uncond = None
    if args.config == "c5":  # SeeCoder-Anime: fixed [77,768] unconditional context zero-padded to 148 (app.py:238-241)
        ug = torch.zeros((1, 148, 768), dtype=torch.float16)
        ug[:, :77] = (torch.randn((1, 77, 768), generator=gen) - 0.1).half()
        uncond = ug.repeat(args.batch, 1, 1).to(dev)

    stage_ms = {}

    def step(i, gather=True, timings=None):
        img, _ = pipe.generate(image, n_global, args.height, args.width, steps=args.ddim_steps, scale=args.scale,
                               eta=0.0, seed=20 + i, gather=gather, control=control, uncond=uncond, timings=timings)
        return img

    for i in range(args.warmup):
        step(i)

    def barrier():
        if world > 1 or forced:
            if on_gpu:
                dist.barrier(device_ids=[local])   # (explicit device: no guess from the rank, no warning, no wrong-GPU context)
            else:
                dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    barrier()
    prof_live = (not args.no_prof) and args.no_graph   # event pairs cannot live inside a captured graph
    if prof_live:
        binding.prof_enable(True)
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = step(100 + i)
    barrier()
    dt = time.perf_counter() - t0
    prof, prof_steps, prof_where = [], args.steps, "timed region (eager launches)"
    if prof_live:
        prof = binding.prof_read()
    elif not args.no_prof and rank == 0:
        # the timed region replayed a hipGraph; time the same kernels on one more, eagerly launched,
        # instrumented batch (identical kernels, shapes and data path; not part of `value`)
        pipe.enable_graph(False)
        binding.prof_enable(True)
        step(999, gather=False)   # rank-0 only: must not enter the collective
        torch.cuda.synchronize()
        prof, prof_steps, prof_where = binding.prof_read(), 1, "instrumented eager replica of one timed step"
        pipe.enable_graph(True)
    if binding is not None:
        binding.prof_enable(False)
    if rank == 0:   # per-stage split of one more (graph-replayed) batch, outside the timed region
        step(998, gather=False, timings=stage_ms)
    dt = max_over_ranks(dt, world, device=dev)     # a step is as slow as its slowest rank
    assert out is not None and out.shape[0] == n_global and bool(torch.isfinite(out).all())

    if rank == 0:
        ddim_real = len(getattr(pipe.sampler, "ddim_timesteps", range(args.ddim_steps)))
        res = {
            "metric": f"images/sec @{args.height}x{args.width} {args.ddim_steps}-step DDIM, SD-v1.5+SeeCoder",
            "value": n_global * args.steps / dt,
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"[{args.config}{'-weak' if world > 1 and args.config == 'c2' else ''}] SD-v1.5 UNet + seecoder-v1-0"
                                   f"{' + ControlNet + PPE_MLP' if args.config == 'c3' else ''}, {args.height}x{args.width}, "
                                   f"{args.ddim_steps}-step DDIM ({ddim_real} real steps), CFG {args.scale}, fp16, "
                                   f"batch={args.batch}/GPU, {args.batch if args.per_sample_image else 1} SeeCoder encode(s) + VAE decode per batch",
                       "global_batch": n_global, "parallelism": f"dp{world}",
                       "world_size_reported_by_backend": dist.get_world_size() if (world > 1 or forced) else 1,
                       "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if (world > 1 or forced) else None,
                       "collectives_forced_at_world_1": bool(forced and world == 1)},
        }
        if args.stub:
            res["data"] = "stub (CPU stand-ins for the GPU compute: launcher / collective test only, not a measurement)"
        if prof:
            top = max(prof, key=lambda b: b["ms"])
            mfma = top["name"].startswith(("gemm", "conv3x3", "attention", "swin")) and \
                (top["bytes"] <= 0 or top["flops"] / top["bytes"] >= MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9))
            tot = sum(b["ms"] for b in prof)
            # Event pairs around EAGER launches also contain the idle time between the previous kernel's end and this
            # kernel's start (the host enqueues ~23 k launches per batch through ctypes; the hipGraph the timed region
            # replays has no such gaps; profiles/*_rocprof_kernel_stats.md hold the in-graph durations of the same command).
            wall = sum(stage_ms.get(k, 0.0) for k in ("ctx_encode_ms", "ddim_loop_ms", "vae_decode_ms"))
            # The idle time in front of an eagerly launched kernel is a roughly CONSTANT host enqueue gap per launch, not
            # a share of the kernel's duration: it is estimated as (sum of the event pairs - graph wall time) / launches
            # and subtracted per launch (a proportional factor would shrink long kernels too much and short ones too
            # little).  A bucket never goes below half of its eager time (guards the estimate on tiny kernels).
            n_launch = sum(b["launches"] for b in prof)
            gap_ms = 0.0
            if not args.no_graph and wall > 0 and tot > wall * prof_steps and n_launch > 0:
                gap_ms = (tot - wall * prof_steps) / n_launch
            for b in prof:
                b["ms_graph"] = max(b["ms"] - gap_ms * b["launches"], 0.5 * b["ms"])
            norm = top["ms_graph"] / top["ms"] if top["ms"] > 0 else 1.0
            secs = top["ms_graph"] / 1e3
            if mfma:
                ach = top["flops"] / secs / 1e12
                res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": ach / MFMA_PEAK_TFLOPS}
            else:
                ach = top["bytes"] / secs / 1e9
                res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS}
            pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
            traffic, traffic_from = None, None
            if os.path.exists(pmc):
                try:
                    tj = json.load(open(pmc))
                    traffic = tj.get(top["name"])
                    # (the PMC passes cannot ride in this process: rocprofv3 --pmc over torch crashes here; the tracked file
                    #  names the commit whose kernels it counted)
                    traffic_from = {"file": "profiles/pmc_traffic.json", "commit": tj.get("_commit"), "recorded": tj.get("_recorded")}
                except Exception:
                    traffic = None
            res["roofline"].update({"traffic": traffic, "traffic_from": traffic_from, "kernel": top["name"], "launches": top["launches"],
                                    "avg_launch_ms": top["ms_graph"] / top["launches"],
                                    "avg_launch_ms_eager": top["ms"] / top["launches"], "eager_to_graph": norm,
                                    "eager_gap_us_per_launch": gap_ms * 1e3,
                                    "measured_on": prof_where + " (eager event pairs minus the per-launch enqueue gap)",
                                    "alg_flops_per_launch": top["flops"] / top["launches"],
                                    "alg_bytes_per_launch": top["bytes"] / top["launches"]})
            # A bucket's binding roofline is HBM when its algorithmic intensity (flops per byte, summed over its launches) is
            # below the machine balance 2500 TFLOP/s / 8 TB/s = 312: GroupNorm / LayerNorm / glue / the split-K reductions
            # (no flops), and -- round 6 -- the short-K linears of the 64-row / 128-row tiles.  Those
            # MFMA-kernel buckets are reported BOTH ways: TFLOP/s in kernel_tflops, GB/s here.
            balance = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            mfma_named = lambda b: b["name"].startswith(("gemm", "conv3x3", "attention", "swin"))  # noqa: E731
            # (1.5 x the balance: the 64-row linears sit at 325 flops per byte -- weight- / activation-streaming launches of the
            #  16^2 / 8^2 levels that neither roof describes alone; they are listed against both)
            hbm_bound = lambda b: (not mfma_named(b)) or (b["bytes"] > 0 and b["flops"] / b["bytes"] < 1.5 * balance)  # noqa: E731
            res["kernel_time_ms_per_step"] = {b["name"]: round(b["ms_graph"] / prof_steps, 3) for b in prof}
            res["kernel_tflops"] = {b["name"]: round(b["flops"] / (b["ms_graph"] / 1e3) / 1e12, 1) for b in prof
                                    if b["flops"] > 0 and b["ms_graph"] > 0 and mfma_named(b)}
            res["kernel_flops_per_byte"] = {b["name"]: round(b["flops"] / b["bytes"], 1) for b in prof if b["bytes"] > 0 and b["flops"] > 0}
            # HBM-bound families (GroupNorm, LayerNorm, elementwise / glue, softmax): algorithmic bytes per second and
            # the fraction of the 8 TB/s peak
            res["kernel_gbps"] = {b["name"]: round(b["bytes"] / (b["ms_graph"] / 1e3) / 1e9, 1) for b in prof
                                  if hbm_bound(b) and b["bytes"] > 0 and b["ms_graph"] > 0}
            res["kernel_hbm_frac"] = {k: round(v / HBM_PEAK_GBS, 3) for k, v in res["kernel_gbps"].items()}
            res["launches_per_step"] = {"instrumented_library_launches": int(n_launch / prof_steps),
                                        "by_bucket": {b["name"]: int(b["launches"] / prof_steps) for b in prof}}
            res["instrumented_kernel_ms_per_step"] = tot / prof_steps          # eager, before the normalisation
            res["launch_mode"] = "eager" if args.no_graph else "hipGraph (DDIM loop)"
        res["stage_ms_per_batch"] = {k: round(v, 2) for k, v in stage_ms.items()}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(net, args.height, args.width, ddim_real, args.scale)
        print(json.dumps(res))
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
