#!/usr/bin/env python
"""Which GroupNorms of one CFG UNet step at the C2 shape find statistics from their producers (one apply launch) and which
take the statistics pass or the single-slab kernel?  Runs one eager apply_model under PFD_TRACE_GN and prints the table.
usage (GPU box): python tools/gn_paths.py [out.log]"""
import collections
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "prompt-free-diffusion_amd"))
trace = tempfile.mktemp(suffix=".gn")
os.environ["PFD_TRACE_GN"] = trace
os.environ.setdefault("PFD_QUIET", "1")
import torch  # noqa: E402
from lib.pipeline import build_model  # noqa: E402
from lib.model_zoo.ddim import DDIMSampler  # noqa: E402

net = build_model('pfd_seecoder', device='cuda', fp16=True)
g = torch.Generator().manual_seed(1)
cond = torch.randn((4, 148, 768), generator=g).cuda().half()
s = DDIMSampler(net)
xT = torch.randn([4, 4, 64, 64], generator=g)
c_info = {'type': 'image', 'conditioning': cond, 'unconditional_conditioning': torch.zeros_like(cond),
          'unconditional_guidance_scale': 2.0}
s.sample(steps=2, shape=[4, 4, 64, 64], x_info={'type': 'image', 'xt': xT.cuda()}, c_info=c_info, eta=0., verbose=False)   # warm-up + packing
open(trace, "w").close()
s.sample(steps=2, shape=[4, 4, 64, 64], x_info={'type': 'image', 'xt': xT.cuda()}, c_info=c_info, eta=0., verbose=False)
torch.cuda.synchronize()
rows = collections.Counter(tuple(int(v) for v in ln.split()) for ln in open(trace))
out = ["B HW C1 C2 | stats(x) stats(x2) shape_ok | count | path"]
for (B, HW, C1, C2, s1, s2, ok), n in sorted(rows.items(), key=lambda kv: (-kv[0][1], kv[0][2])):
    path = "one launch from producer statistics" if (s1 and s2 and ok) else ("single-slab kernel / statistics pass" if not ok else "STATISTICS PASS (producer emitted none)")
    out.append(f"{B} {HW} {C1} {C2} | {s1} {s2} {ok} | {n} | {path}")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
