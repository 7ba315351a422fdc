import os, sys, time, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, 'prompt-free-diffusion_amd')
from lib.model_zoo.autokl_modules import AttnBlock
torch.manual_seed(0)
m = AttnBlock(512).half().cuda()
for (B, H, W) in ((4, 64, 64), (2, 96, 96)):
    x = torch.randn(B, H, W, 512, device='cuda').half()
    outs = {}
    for mode in ("gemm", "fused"):
        os.environ["PFD_VAE_ATTN"] = mode
        y = m.hip(x); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10): y = m.hip(x)
        torch.cuda.synchronize()
        outs[mode] = y
        print(f"AttnBlock B{B} {H}x{W} {mode}: {(time.time()-t0)*100:.3f} ms per call", flush=True)
    d = (outs["gemm"].float() - outs["fused"].float()).abs().max().item()
    print(f"  max |gemm - fused| = {d:.3e} (max |y| {outs['fused'].float().abs().max().item():.2f})", flush=True)
