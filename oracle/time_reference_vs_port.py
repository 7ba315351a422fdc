"""TEST INFRASTRUCTURE: times the REFERENCE's own modules (imported from /root/reference, build container
only) against the oracle port on the same host, same threads, same input -- one CFG UNet step (batch 2) at
64x64, SeeCoder at 512x512, VAE decode of a 64x64 latent.  bench.py's `cpu_baseline` must use the port on the
GPU box (the reference tree does not travel); this script shows how the two compare where both can run.
    python oracle/time_reference_vs_port.py > profiles/rNN_cpu_reference_vs_port.log
Also writes profiles/cpu_reference_vs_port.json (tracked): bench.py's `cpu_baseline.reference_vs_port` is read from
that file, so the numbers in a bench line are the ones THIS script measured, with their provenance.
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import pfd_oracle as O  # noqa: E402


def timed(fn, n=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


def main():
    MG.install_shims()
    os.chdir(MG.REF)
    sys.path.insert(0, MG.REF)
    torch.set_grad_enabled(False)
    from lib.cfg_helper import model_cfg_bank
    from lib.model_zoo import get_model
    cfgm = model_cfg_bank()('pfd_seecoder')
    cfgm.args.vae_cfg_list[0][1].pth = None
    net = get_model()(cfgm)
    net.to('cpu')
    net.eval()
    sd = {k: v.detach().float() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 4, 64, 64), generator=g)
    t = torch.tensor([981, 981])
    c = torch.randn((2, 148, 768), generator=g)
    img = torch.rand((1, 3, 512, 512), generator=g)
    z = torch.randn((1, 4, 64, 64), generator=g)
    print(f"host cpus {os.cpu_count()}, torch threads {torch.get_num_threads()}")
    rows = [
        ("UNet CFG step [2,4,64,64]",
         lambda: net.apply_model({'type': 'image', 'x': x}, t, {'type': 'image', 'c': c}),
         lambda: O.unet_apply(sd, "diffuser.image.", x, t, c)),
        ("SeeCoder 512x512", lambda: net.ctx_encode(img, which='image'),
         lambda: O.seecoder_encode(sd, "ctx.image.", img)),
        ("VAE decode 64x64 latent", lambda: net.vae_decode(z, which='image'),
         lambda: O.vae_decode(sd, "vae.image.", z)),
    ]
    out = {"host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__,
           "measured": time.strftime("%Y-%m-%d"), "script": "oracle/time_reference_vs_port.py", "stages": {}}
    for name, ref, port in rows:
        a, b = ref(), port()
        err = float((a - b).abs().max())
        tr, tp = timed(ref), timed(port)
        print(f"{name}: reference {tr:.2f} s, port {tp:.2f} s (port/reference {tp / tr:.2f}), max|diff| {err:.2e}")
        out["stages"][name] = {"reference_s": round(tr, 3), "port_s": round(tp, 3), "port_over_reference": round(tp / tr, 3),
                               "max_abs_diff": err}
    ratios = [v["port_over_reference"] for v in out["stages"].values()]
    out["max_abs_diff"] = max(v["max_abs_diff"] for v in out["stages"].values())
    out["port_time_over_reference_time"] = [min(ratios), max(ratios)]
    with open(os.path.join(os.path.dirname(HERE), "profiles", "cpu_reference_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
