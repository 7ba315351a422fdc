"""Library launches of the context encode and the VAE decode of one C2 batch, by profiling bucket (eager launches with the
library's event pairs: the durations contain the host's enqueue gaps, the COUNTS are what the stage graphs replay), plus
the torch kernels in between (torch profiler).  GPU only.   usage: python tools/stage_launches.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "prompt-free-diffusion_amd"))
from lib.hip import binding  # noqa: E402
from lib.pipeline import PromptFreePipeline, build_model  # noqa: E402

net = build_model()
pipe = PromptFreePipeline(net)
image = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(0)).to(net.device)
x = torch.randn((4, 4, 64, 64), generator=torch.Generator().manual_seed(1)).half().to(net.device)
stages = {"ctx_encode": lambda: net.ctx_encode(image, 'image'), "vae_decode": lambda: net.vae_decode(x, 'image')}
for name, fn in stages.items():
    with torch.no_grad():
        fn()                                   # packs weights
        torch.cuda.synchronize()
        binding.prof_enable(True)
        fn()
        torch.cuda.synchronize()
        prof = binding.prof_read()
        binding.prof_enable(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"== {name}: {sum(b['launches'] for b in prof)} library launches, {e0.elapsed_time(e1):.2f} ms eager wall")
        for b in sorted(prof, key=lambda b: -b["launches"]):
            print(f"   {b['launches']:5d} launches  {b['ms']:8.3f} ms (event pairs)  {b['name']}")
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as tp:
            fn()
            torch.cuda.synchronize()
        rows = [(e.key, e.count, e.self_device_time_total) for e in tp.key_averages() if e.self_device_time_total > 0]
        rows.sort(key=lambda r: -r[2])
        print(f"   -- device kernels by name (torch profiler): {sum(r[1] for r in rows)} launches, {sum(r[2] for r in rows) / 1e3:.2f} ms device time")
        for k, c, t in rows[:25]:
            print(f"   {c:5d}  {t / 1e3:8.3f} ms  {k[:110]}")
